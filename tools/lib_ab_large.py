#!/usr/bin/env python3
"""Diagnostic: A/B of builds of the library on rdv_step at sizes beyond the Infinity Cache, where the launch time of ONE build varies
by +-10 % with the allocation and the moment: every build runs in its own child process, the processes alternate,
and inside a process the batch is created FRESH several times — all samples are printed, so that builds are compared by their
distributions (min / median), not by one draw each.
    AB_SIZES=524288,4194304 AB_ALLOCS=5 AB_ROUNDS=3 python tools/lib_ab_large.py tools/_a.so tools/_b.so
An argument may carry environment settings for its child: "tools/_a.so,AB_VARIANT=fused_tiles,RDV_TILES_GRID=512" ("-" = the product
library; AB_VARIANT = the batch's kernel variant)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from reinforcement_learning_rendezvous_amd import _native
if sys.argv[1] != "-":
    _native.LIB_PATH = sys.argv[1]
    _native.STRICT = False
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
for n in [int(x) for x in os.environ.get("AB_SIZES", "4194304").split(",")]:
    g0 = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [(torch.rand((n, 6), device="cuda:0", generator=g0) * 2 - 1).contiguous() for _ in range(2)]
    out = []
    for trial in range(int(os.environ.get("AB_ALLOCS", "5"))):
        env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0, variant=os.environ.get("AB_VARIANT", "auto"))
        env.reset()
        for t in range(24): env.step(acts[t %% 2])
        steps = 16 if n > 1000000 else 64
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(steps): env.step(acts[t %% 2])
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
        out.append(sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(R))[R // 2] * 1e3 / steps)
        env.close(); del env, g
        torch.cuda.empty_cache()
    print(f"{n}: " + " ".join(f"{x:.1f}" for x in out), end="   ")
print()
''' % ROOT
libs = [("product", "-", {})]
for o in sys.argv[1:]:
    parts = o.split(",")
    libs.append((os.path.basename(parts[0]) + ("," + ",".join(parts[1:]) if parts[1:] else ""), parts[0], dict(kv.split("=", 1) for kv in parts[1:])))
for rep in range(int(os.environ.get("AB_ROUNDS", "2"))):
    for label, lib, env in libs:
        r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True, timeout=600, env={**os.environ, **env})
        print(f"{label:34s} us per launch, one value per fresh allocation   {r.stdout.strip()}" + ("" if r.returncode == 0 else " FAILED " + r.stderr[-300:]), flush=True)
