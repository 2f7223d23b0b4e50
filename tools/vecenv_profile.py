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
t0 = time.perf_counter()
for k in range(20):
    vec.step(acts[k % 8])
dt = (time.perf_counter() - t0) / 20
print(f"{dt * 1e3:.2f} ms per step, {n / dt * 1e-6:.2f} M env steps/s")
pr = cProfile.Profile()
pr.enable()
for k in range(20):
    vec.step(acts[k % 8])
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
