#!/usr/bin/env python3
"""Diagnostic (not product): where a wave of the fused by-part step kernel (step_kernel_parts, N > 65,536) spends its life.

    python tools/stamp_profile_parts.py 4194304

Builds a -DRDV_STAMPS copy of the library into tools/_stamps.so.  Stamps per wave (shader cycles):
  0 entry | 1 inputs in (the stamped build waits for them explicitly) | 2 transition done | 3 statistics, outputs, state stored (at
  barrier 1) | 4 past it | 5 reset parts done (at barrier 2) | 6 past it | 7 rows stored
plus s_memrealtime at entry and exit: wave lifetimes, how many waves are resident over the launch, the launch as the waves see it.
Never quote this build's run time."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
    lib = os.path.join(ROOT, "tools", "_stamps.so")
    from _build import build_variant
    build_variant(lib, ["-DRDV_STAMPS"] + os.environ.get("RDV_EXTRA_FLAGS", "").split())
    import torch
    from reinforcement_learning_rendezvous_amd import _native
    _native.LIB_PATH = lib
    from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
    env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0, variant="fused")
    L = _native.lib()
    L.rdv_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    waves = ((n + 255) // 256 + 8) * 4
    stamps = torch.zeros((waves, 10), dtype=torch.int64, device="cuda:0")
    gen = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [(torch.rand((n, 6), device="cuda:0", generator=gen) * 2 - 1).contiguous() for _ in range(2)]
    env.reset()
    for t in range(40):
        env.step(acts[t % 2])
    _native.check(L.rdv_debug_set_stamps(env._h, stamps.data_ptr()))
    rows = []
    for t in range(6):
        for _ in range(3):
            env.step(acts[t % 2])
        torch.cuda.synchronize()
        rows.append(stamps.cpu().numpy().copy())
    s = np.stack(rows).astype(np.float64)
    s = s[:, s[0, :, 7] != 0, :]                      # waves that ran (padding workgroups write nothing)
    cyc = s[:, :, 7] - s[:, :, 0]
    real = (s[:, :, 9] - s[:, :, 8]) * 10.0          # ns
    print(f"{n} envs, {s.shape[1]} waves; shader clock while the kernel runs: {np.median(cyc / real):.2f} GHz (median over waves)")
    d = np.diff(s[:, :, :8], axis=2)
    names = ["inputs in", "transition", "statistics + outputs + state store", "barrier 1 wait", "reset parts", "barrier 2 wait", "row stores"]
    med, p90 = np.median(d, axis=(0, 1)), np.percentile(d, 90, axis=(0, 1))
    for k, nm in enumerate(names):
        print(f"  {nm:38s} median {med[k]:8.0f}  p90 {p90[k]:8.0f} cycles")
    print(f"  wave lifetime: median {np.median(real):.0f} ns, p90 {np.percentile(real, 90):.0f} ns")
    span = (s[:, :, 9].max(axis=1) - s[:, :, 8].min(axis=1)) * 10.0
    print(f"  launch as the waves see it (first entry -> last exit): {np.median(span) / 1e3:.1f} us; "
          f"resident waves on average: {np.median(real.sum(axis=1) / span):.0f} of {1024 * 4} slots at four per SIMD")


if __name__ == "__main__":
    main()
