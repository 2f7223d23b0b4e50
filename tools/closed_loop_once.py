#!/usr/bin/env python3
"""Diagnostic workload for rocprofv3 --pmc: the closed loop of BASELINE config 3 at 65,536 envs in its two forms —
rdv_rollout (64 steps per persistent launch) x6, and 64 x (rdv_policy_act + rdv_step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
from reinforcement_learning_rendezvous_amd.policy import MlpPolicy

n, T = 65536, 64
pol = MlpPolicy.from_npz(os.path.join(ROOT, "tests", "golden", "mlp_policy.npz")).to("cuda:0")
pol.backend = "hip"
env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0)
env.reset()
out = env.rollout(pol, T)
for _ in range(6):
    env.rollout(pol, T, out=out)
buf = torch.empty((n, 6), dtype=torch.float32, device="cuda:0")
for _ in range(T):
    env.step(env.act(pol, deterministic=False, out=buf))
torch.cuda.synchronize()
print("done", env.get_stats()["episodes"])
