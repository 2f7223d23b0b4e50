#!/usr/bin/env python3
"""Diagnostic: where the time of RendezvousVecEnv.step (NumPy boundary) goes at N envs."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from reinforcement_learning_rendezvous_amd.vec_env import RendezvousVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
vec = RendezvousVecEnv(n, device="cuda:0")
rng = np.random.default_rng(0)
acts = [rng.uniform(-1, 1, (n, 6)).astype(np.float32) for _ in range(8)]
vec.reset()
for k in range(30):
    vec.step(acts[k % 8])
import gc
per = []
for k in range(60):
    t0 = time.perf_counter()
    vec.step(acts[k % 8])
    per.append(time.perf_counter() - t0)
per_ms = np.array(per) * 1e3
dt = per_ms.mean() * 1e-3
print(f"{dt * 1e3:.2f} ms per step (median {np.median(per_ms):.2f}, min {per_ms.min():.2f}, max {per_ms.max():.2f}), {n / dt * 1e-6:.2f} M env steps/s; gc counts {gc.get_count()}, thresholds {gc.get_threshold()}")
print("per-step ms:", " ".join(f"{x:.1f}" for x in per_ms[:40]))
gc.disable()
per = []
for k in range(40):
    t0 = time.perf_counter()
    vec.step(acts[k % 8])
    per.append(time.perf_counter() - t0)
print(f"with the garbage collector off: {np.mean(per) * 1e3:.2f} ms per step (median {np.median(per) * 1e3:.2f})")
gc.enable()
pr = cProfile.Profile()
pr.enable()
for k in range(20):
    vec.step(acts[k % 8])
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
