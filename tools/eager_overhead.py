#!/usr/bin/env python3
"""Diagnostic: what a Python-driven (eager, no HIP graph) loop of RendezvousBatch.step() costs per call, against the kernel's own time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch

for n in (1024, 65536):
    env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [(torch.rand((n, 6), device="cuda:0", generator=g) * 2 - 1).contiguous() for _ in range(8)]
    env.reset()
    for t in range(200):
        env.step(acts[t % 8])
    torch.cuda.synchronize()
    K = 5000
    t0 = time.perf_counter()
    for t in range(K):
        env.step(acts[t % 8])
    t1 = time.perf_counter()          # host time to ENQUEUE K steps
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: enqueue {1e6 * (t1 - t0) / K:.2f} us per call, wall {1e6 * (t2 - t0) / K:.2f} us per step (eager)")
