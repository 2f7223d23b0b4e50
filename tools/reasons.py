#!/usr/bin/env python3
"""Diagnostic: why episodes end in the bench workload (U(-1,1) actions, default parameters)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
n = 65536
env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0)
g = torch.Generator(device="cuda:0").manual_seed(1)
acts = [(torch.rand((n, 6), device="cuda:0", generator=g) * 2 - 1).contiguous() for _ in range(16)]
env.reset()
for t in range(2000):
    env.step(acts[t % 16])
s = env.get_stats()
print({k: s[k] for k in ("env_steps", "episodes", "reasons", "successes", "collisions")}, "mean length", s["sum_length"] / s["episodes"])
