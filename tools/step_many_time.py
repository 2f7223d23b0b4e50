#!/usr/bin/env python3
"""Diagnostic: us per env step of rdv_step_many (open-loop action tape, K steps per launch) beside rdv_step under a HIP graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
for storage in ("f32", "f64"):
    env = RendezvousBatch(n, device="cuda:0", storage=storage, seed=0)
    env.reset()
    tape = (torch.rand((K, n, 6), device="cuda:0") * 2 - 1).contiguous()
    out = env.step_many(tape)
    for _ in range(3):
        env.step_many(tape, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 32
    e0.record()
    for _ in range(reps):
        env.step_many(tape, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * K)
    print(f"{storage} step_many K={K}: {us:6.2f} us/step  {n / us * 1e-3:7.3f} G env steps/s", flush=True)
    env.close()
