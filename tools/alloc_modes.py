#!/usr/bin/env python3
"""Diagnostic: rdv_step at 4.2 M envs runs in one of two modes per ALLOCATION (~285 or ~320-340 us per launch, profiles/r04_alloc_modes.txt).
Prints, for a series of fresh batches in one process, the device addresses of the batch's buffers next to the time per launch; between
batches a pad tensor of a chosen size stays allocated, so that the next batch lands elsewhere.
    python tools/alloc_modes.py [n_envs] [pad MiB, ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch

class ArenaBatch(RendezvousBatch):
    """every buffer of the batch carved out of ONE device allocation (2 MiB-aligned pieces)"""
    arena, used = None, 0
    def _alloc(self, name, shape, dtype):
        nbytes = int(torch.tensor([], dtype=dtype).element_size())
        for d in shape: nbytes *= d
        off = ArenaBatch.used
        ArenaBatch.used = (off + nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20) + int(os.environ.get("ARENA_SKEW", "0"))
        t = ArenaBatch.arena[off:off + nbytes].view(dtype).view(shape)
        t.zero_()
        return t

MODE = os.environ.get("ALLOC_MODE", "separate")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
pads = [int(x) for x in sys.argv[2:]] or [0, 0, 0, 1, 2, 3, 64, 0, 0]
g0 = torch.Generator(device="cuda:0").manual_seed(1)
acts = [(torch.rand((n, 6), device="cuda:0", generator=g0) * 2 - 1).contiguous() for _ in range(2)]
print("actions at", [hex(a.data_ptr()) for a in acts])
keep = []
for trial, pad in enumerate(pads):
    if pad:
        keep.append(torch.empty((pad << 20,), dtype=torch.uint8, device="cuda:0"))
    if MODE == "arena":
        ArenaBatch.arena, ArenaBatch.used = torch.empty((3 << 30,), dtype=torch.uint8, device="cuda:0"), 0
        env = ArenaBatch(n, device="cuda:0", storage="f32", seed=0)
    else:
        env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0)
    env.reset()
    for t in range(24): env.step(acts[t % 2])
    steps = 16
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for t in range(steps): env.step(acts[t % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        g.replay(); torch.cuda.synchronize()
    R = 9
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(R + 1)]
    ev[0].record()
    for r in range(R):
        g.replay(); ev[r + 1].record()
    torch.cuda.synchronize()
    us = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(R))[R // 2] * 1e3 / steps
    addr = {k: getattr(env, k).data_ptr() for k in ("_ws", "_obs", "_reward", "_done", "terminal_obs", "episode_return", "episode_length", "done_reason")}
    print(f"trial {trial} pad {pad:4d} MiB: {us:6.1f} us   " + " ".join(f"{k.strip('_')}={v:#x}" for k, v in addr.items()), flush=True)
    env.close(); del env, g
    ArenaBatch.arena = None
    if os.environ.get("ALLOC_KEEP_CACHE") != "1":
        torch.cuda.empty_cache()
