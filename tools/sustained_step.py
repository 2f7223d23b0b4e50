#!/usr/bin/env python3
"""Diagnostic: rdv_step at N envs under sustained load: a 256-step HIP graph replayed back to back for ~3 s, us per launch per replay
(does the launch period depend on how long the GPU has been busy?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0)
g0 = torch.Generator(device="cuda:0").manual_seed(1)
acts = [(torch.rand((n, 6), device="cuda:0", generator=g0) * 2 - 1).contiguous() for _ in range(8)]
env.reset()
for t in range(32):
    env.step(acts[t % 8])
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for t in range(256):
        env.step(acts[t % 8])
torch.cuda.synchronize()
time.sleep(2.0)                      # idle: clocks fall back
R = 1500
ev = [torch.cuda.Event(enable_timing=True) for _ in range(R + 1)]
ev[0].record()
for r in range(R):
    g.replay()
    ev[r + 1].record()
torch.cuda.synchronize()
us = [ev[r].elapsed_time(ev[r + 1]) * 1e3 / 256 for r in range(R)]
t_acc, marks = 0.0, []
for r, u in enumerate(us):
    t_acc += u * 256 * 1e-3
    if r in (0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, R - 1):
        marks.append(f"replay {r} (t={t_acc:.0f} ms): {u:.3f}")
print(f"n={n}: us per launch -> " + " | ".join(marks), flush=True)
