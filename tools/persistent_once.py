#!/usr/bin/env python3
"""Diagnostic workload for rocprofv3 --pmc: a few launches of the persistent kernels (rdv_step_many, rdv_rollout) at 65,536 envs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
from reinforcement_learning_rendezvous_amd.policy import MlpPolicy

n, K = 65536, 64
pol = MlpPolicy.from_npz(os.path.join(ROOT, "tests", "golden", "mlp_policy.npz")).to("cuda:0")
env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0)
env.reset()
tape = (torch.rand((K, n, 6), device="cuda:0") * 2 - 1).contiguous()
o1 = env.step_many(tape)
o2 = env.rollout(pol, K)
for _ in range(6):
    env.step_many(tape, out=o1)
    env.rollout(pol, K, out=o2)
torch.cuda.synchronize()
print("done")
