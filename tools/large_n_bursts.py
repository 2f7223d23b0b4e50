#!/usr/bin/env python3
"""Diagnostic: is the two-mode launch time of the fused kernel at 4.2 M envs (profiles/r02_large_n_placement.txt) a property of the
allocation or of the moment?  ONE batch, bursts of 16 launches separated by idle gaps of different lengths."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch

n, dev = 4194304, "cuda:0"
for trial in range(3):
    env = RendezvousBatch(n, device=dev, storage="f32", seed=0)
    g = torch.Generator(device=dev).manual_seed(1)
    acts = [(torch.rand((n, 6), device=dev, generator=g) * 2 - 1).contiguous() for _ in range(2)]
    env.reset()
    for t in range(24):
        env.step(acts[t % 2])
    torch.cuda.synchronize()
    out = []
    for gap in (0.0, 0.0, 0.001, 0.01, 0.1, 0.5, 0.0, 1.0, 0.0, 0.05, 2.0, 0.0):
        time.sleep(gap)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(16):
            env.step(acts[t % 2])
        e1.record()
        torch.cuda.synchronize()
        out.append((gap, e0.elapsed_time(e1) * 1e3 / 16))
    print(f"allocation {trial}: " + "  ".join(f"[{gp:g}s] {us:6.1f}" for gp, us in out), flush=True)
    env.close()
    del env, acts
    torch.cuda.empty_cache()

# ---- sustained load: 960 launches back to back (events every 16), once from a busy GPU and once after 1 s of idling
env = RendezvousBatch(n, device=dev, storage="f32", seed=0)
g = torch.Generator(device=dev).manual_seed(1)
acts = [(torch.rand((n, 6), device=dev, generator=g) * 2 - 1).contiguous() for _ in range(2)]
env.reset()
for label, gap in (("from a busy GPU", 0.0), ("after 1 s idle", 1.0), ("after 5 s idle", 5.0)):
    time.sleep(gap)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    ev[0].record()
    for c in range(60):
        for t in range(16):
            env.step(acts[t % 2])
        ev[c + 1].record()
    torch.cuda.synchronize()
    us = [ev[c].elapsed_time(ev[c + 1]) * 1e3 / 16 for c in range(60)]
    print(f"sustained, {label}: " + " ".join(f"{u:.0f}" for u in us), flush=True)
