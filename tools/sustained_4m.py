#!/usr/bin/env python3
"""Diagnostic: the fused kernel at 4.2 M envs under sustained load (~6 s), us per launch per 64 launches, with wall-clock stamps to line
up with a `rocm-smi` sampler running beside it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch

n, dev = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304, "cuda:0"
env = RendezvousBatch(n, device=dev, storage="f32", seed=0)
g = torch.Generator(device=dev).manual_seed(1)
acts = [(torch.rand((n, 6), device=dev, generator=g) * 2 - 1).contiguous() for _ in range(2)]
env.reset()
torch.cuda.synchronize()
time.sleep(2.0)
t_start = time.time()
print(f"start {t_start:.2f}", flush=True)
for block in range(6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(51)]
    ev[0].record()
    for c in range(50):
        for t in range(64):
            env.step(acts[t % 2])
        ev[c + 1].record()
    torch.cuda.synchronize()
    us = [ev[c].elapsed_time(ev[c + 1]) * 1e3 / 64 for c in range(50)]
    print(f"t+{time.time() - t_start:5.2f}s: " + " ".join(f"{u:.0f}" for u in us), flush=True)
