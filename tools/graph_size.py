#!/usr/bin/env python3
"""Diagnostic: us per launch of rdv_step at 65,536 envs against the number of launches captured per HIP graph (sustained load)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
n = 65536
env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0)
g0 = torch.Generator(device="cuda:0").manual_seed(1)
acts = [(torch.rand((n, 6), device="cuda:0", generator=g0) * 2 - 1).contiguous() for _ in range(8)]
env.reset()
for t in range(32):
    env.step(acts[t % 8])
for K in (20, 64, 256, 1024, 4096, 256, 20):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for t in range(K):
            env.step(acts[t % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        g.replay(); torch.cuda.synchronize()
    R = max(4, 40000 // K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{K:5d} launches per graph, {R} replays back to back: {e0.elapsed_time(e1) * 1e3 / (R * K):.3f} us per launch", flush=True)
