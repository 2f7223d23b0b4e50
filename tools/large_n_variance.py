#!/usr/bin/env python3
"""Diagnostic: run-to-run spread of the fused step kernel beyond the Infinity Cache — the same batch size re-created several times
in one process (fresh allocations), and sizes just off the power of two (chunk arrays no longer 2^k bytes apart)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch


def time_one(n, reps=5, steps=16):
    env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [(torch.rand((n, 6), device="cuda:0", generator=g) * 2 - 1).contiguous() for _ in range(2)]
    env.reset()
    for t in range(24):
        env.step(acts[t % 2])
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(steps):
            env.step(acts[t % 2])
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / steps)
    ptrs = [env._ws.data_ptr(), env.obs.data_ptr(), env.reward.data_ptr(), env.done.data_ptr(), env.terminal_obs.data_ptr(),
            env.episode_return.data_ptr(), env.episode_length.data_ptr(), env.done_reason.data_ptr(), acts[0].data_ptr(), acts[1].data_ptr()]
    print("   ws obs rew done tobs eret elen reason a0 a1 [MiB from ws]: " + " ".join(f"{(q - ptrs[0]) / 1048576:.2f}" for q in ptrs) + f"  ws at {ptrs[0]:#x}")
    env.close()
    del env, acts
    torch.cuda.empty_cache()
    return out


for n in [4194304] * 8:
    us = time_one(n)
    best = min(us)
    print(f"n={n:9d}: " + " ".join(f"{u:7.1f}" for u in us) + f"  us per launch | best {best / n * 1e3:7.4f} ns per env-step, "
          f"{293 * n / (best * 1e-6) / 8e12:.3f} of 8 TB/s", flush=True)
