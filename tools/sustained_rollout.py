#!/usr/bin/env python3
"""Diagnostic: the persistent rollout kernel under sustained load (~8 s), us per step per launch, to line up with a `rocm-smi` sampler."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
from reinforcement_learning_rendezvous_amd.policy import MlpPolicy
n, T = 65536, 64
det = "det" in sys.argv[1:]
storage = "f64" if "f64" in sys.argv[1:] else "f32"
blocks = 3 if "short" in sys.argv[1:] else 8
pol = MlpPolicy.from_npz(os.path.join(ROOT, "tests", "golden", "mlp_policy.npz")).to("cuda:0")
env = RendezvousBatch(n, device="cuda:0", storage=storage, seed=0)
env.reset()
out = env.rollout(pol, T, deterministic=det)
torch.cuda.synchronize()
time.sleep(1.5)
print(f"start {time.time():.2f} deterministic={det} storage={storage}", flush=True)
for block in range(blocks):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    ev[0].record()
    for c in range(40):
        for _ in range(32):
            env.rollout(pol, T, deterministic=det, out=out)
        ev[c + 1].record()
    torch.cuda.synchronize()
    us = [ev[c].elapsed_time(ev[c + 1]) * 1e3 / (32 * T) for c in range(40)]
    print(f"{time.time():.2f}: " + " ".join(f"{u:.2f}" for u in us[::4]), flush=True)
