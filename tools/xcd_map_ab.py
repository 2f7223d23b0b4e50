#!/usr/bin/env python3
"""Diagnostic: plain block order against the XCD-contiguous one for the fused kernel (RDV_XCD_ORDER=0|1 at rdv_create), two handles
per size, timed alternately; every step of the warm-up checks that the two give identical outputs.  (Beyond the cache the launch
time also depends on the allocation, profiles/r02_large_n_placement.txt: the run recorded in profiles/r02_xcd_order_ab.txt toggled
the order per launch on ONE handle.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4194304,4194304,4194304,1048576,16777216,262144,131072").split(",")]
dev = "cuda:0"


def timed(env, acts, steps=16, reps=3):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(steps):
            env.step(acts[t % 2])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
    return best


for n in sizes:
    envs = {}
    for mode in ("0", "1"):
        os.environ["RDV_XCD_ORDER"] = mode
        envs[mode] = RendezvousBatch(n, device=dev, storage="f32", seed=0, variant="fused")
    os.environ.pop("RDV_XCD_ORDER")
    g = torch.Generator(device=dev).manual_seed(1)
    acts = [(torch.rand((n, 6), device=dev, generator=g) * 2 - 1).contiguous() for _ in range(2)]
    a, b = envs["0"], envs["1"]
    assert torch.equal(a.reset(), b.reset())
    for t in range(24):
        a.step(acts[t % 2]); b.step(acts[t % 2])
        assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done), t
    assert torch.equal(a.get_state(), b.get_state()) and a.get_stats() == b.get_stats()
    out = []
    for rep in range(2):
        for mode in ("0", "1"):
            out.append((mode, timed(envs[mode], acts, steps=16 if n >= 1048576 else 128)))
    a.close(); b.close()
    del envs, a, b, acts
    torch.cuda.empty_cache()
    print(f"n={n:9d}: " + "  ".join(f"{'xcd' if m == '1' else 'plain'} {u:8.2f} us" for m, u in out), flush=True)
