#!/usr/bin/env python3
"""Diagnostic: how the fused step kernel's launch time beyond the Infinity Cache depends on WHERE its arrays lie relative to each
other.  Everything the kernel touches (workspace, outputs, actions) is carved from one arena at chosen offsets (base 2 MiB-aligned +
a skew), so the placement is the experiment's variable instead of the allocator's accident.

    python tools/placement_sweep.py [n_envs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from reinforcement_learning_rendezvous_amd import batch as B

MiB = 1 << 20
dev = torch.device("cuda:0")
arena = torch.zeros(6 * 1024 * MiB, dtype=torch.uint8, device=dev)
base0 = (-arena.data_ptr()) % (2 * MiB)            # first 2 MiB-aligned byte of the arena
print(f"arena at {arena.data_ptr():#x}, first 2 MiB boundary at +{base0}", flush=True)


class Carver:
    def __init__(self, skews):
        self.cur = base0
        self.skews = skews                          # name -> skew in bytes (default 0)
        self.placed = {}

    def take(self, name, nbytes):
        start = self.cur + self.skews.get(name, 0)
        self.placed[name] = start - base0
        if start + nbytes > arena.numel():          # (a slice past the end would be silently shorter: r2's last sweep rows died in .view)
            raise MemoryError(f"{name}: {nbytes} bytes at +{start} do not fit the {arena.numel() >> 20} MiB arena")
        self.cur = start + nbytes
        self.cur += (-self.cur) % (2 * MiB)         # next array starts on a 2 MiB boundary (+ its skew)
        return arena[start:start + nbytes]


def run(n, skews, label, steps=16, reps=3):
    carver = Carver(skews)

    def alloc(self, name, shape, dtype):
        nb = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        t = carver.take(name, nb)
        t.zero_()
        return t.view(dtype).view(*shape)
    B.RendezvousBatch._alloc = alloc
    try:
        env = B.RendezvousBatch(n, device="cuda:0", storage="f32", seed=0)
        acts = [carver.take(f"actions{k}", n * 24).view(torch.float32).view(n, 6) for k in range(2)]
    except MemoryError as exc:
        print(f"{label:58s} n={n:8d}: skipped ({exc})", flush=True)
        return None
    g = torch.Generator(device="cuda:0").manual_seed(1)
    for a in acts:
        a.copy_(torch.rand((n, 6), device=dev, generator=g) * 2 - 1)
    env.reset()
    for t in range(24):
        env.step(acts[t % 2])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(steps):
            env.step(acts[t % 2])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
    env.close()
    print(f"{label:58s} n={n:8d}: {best:7.1f} us per launch, {293 * n / (best * 1e-6) / 8e12:.3f} of 8 TB/s", flush=True)
    return best


n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
names = ["workspace", "obs", "reward", "done", "terminal_obs", "episode_return", "episode_length", "done_reason", "actions0", "actions1"]
run(n0, {}, "all arrays on 2 MiB boundaries")
for mib in (2, 4, 6, 10, 14, 18, 34, 66, 130):
    d = mib * MiB // 16
    run(n0 + d, {}, f"chunk arrays 2^26 + {mib} MiB apart (n = 2^22 + {d})")
run(n0, {}, "all arrays on 2 MiB boundaries (again)")
