#!/usr/bin/env python3
"""Diagnostic (not product): where a launch of the split step kernel spends its time.

Builds a -DRDV_STAMPS copy of the library into tools/_stamps.so, runs a few hundred steps at N envs and prints, per role
(step waves / service waves), the median shader-cycle counts between the stamps
  0 entry | 1 inputs staged | 2 step / next-state computed | 3 at the barrier | 4 past it | 5 | 6 | 7 end
plus, from s_memrealtime (100 MHz, one counter for the whole chip): the shader clock, when waves enter and leave
relative to the first wave of the launch, and the span of the launch as the waves see it.
Never quote this build's run time: the stamps fence the scheduler."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    svc = 1          # service waves per step wave (round 3 also stamped a 12-wave workgroup with two: profiles/r03_split_service_waves_stamps.txt)
    wpg = 4 * (1 + svc)
    lib = os.path.join(ROOT, "tools", "_stamps.so")
    from _build import build_variant
    build_variant(lib, ["-DRDV_STAMPS"] + os.environ.get("RDV_EXTRA_FLAGS", "").split())
    import torch
    from reinforcement_learning_rendezvous_amd import _native
    _native.LIB_PATH = lib
    from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
    env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0, variant="split")
    L = _native.lib()
    L.rdv_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    waves = (n + 255) // 256 * wpg
    stamps = torch.zeros((waves, 10), dtype=torch.int64, device="cuda:0")
    gen = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [(torch.rand((n, 6), device="cuda:0", generator=gen) * 2 - 1).contiguous() for _ in range(8)]
    env.reset()
    for t in range(100):
        env.step(acts[t % 8])
    _native.check(L.rdv_debug_set_stamps(env._h, stamps.data_ptr()))
    rows = []
    for t in range(50):
        for _ in range(4):                 # back-to-back launches; the last one's stamps are read
            env.step(acts[t % 8])
        torch.cuda.synchronize()
        rows.append(stamps.cpu().numpy().copy())
    s = np.stack(rows).astype(np.float64)            # [iter, wave, 10]
    role = (np.arange(waves) % wpg) // 4          # 0 step, 1 service (chaser half if svc = 2), 2 service (target half)
    cyc = s[:, :, 7] - s[:, :, 0]
    real = (s[:, :, 9] - s[:, :, 8]) * 10.0          # ns
    print(f"shader clock while the kernel runs: {np.median(cyc / real):.2f} GHz (median over waves)")
    t0 = s[:, :, 8].min(axis=1, keepdims=True)
    names = ["step waves", "service waves" if svc == 1 else "service waves (chaser half)", "service waves (target half)"]
    for name, sel in [(names[k], role == k) for k in range(1 + svc)]:
        x = s[:, sel, :]
        d = np.diff(x[:, :, :8], axis=2)
        ent = (x[:, :, 8] - t0) * 10.0
        ext = (x[:, :, 9] - t0) * 10.0
        print(f"{name}: median cycles per phase 0>1>..>7 {np.median(d, axis=(0, 1)).astype(int).tolist()} | wave lifetime "
              f"{np.median(real[:, sel]):.0f} ns | enters {np.median(ent):.0f} ns (p95 {np.percentile(ent, 95):.0f}) and leaves "
              f"{np.median(ext):.0f} ns (p95 {np.percentile(ext, 95):.0f}, max {np.median(ext.max(axis=1)):.0f}) after the first wave")
    at_barrier = s[:, :, 3] - s[:, :, 0]
    print("cycles from entry to the barrier, median by role: " + ", ".join(f"{names[k]} {np.median(at_barrier[:, role == k]):.0f}" for k in range(1 + svc)))
    print(f"launch span seen by the waves (first entry -> last exit): {np.median((s[:, :, 9].max(axis=1) - s[:, :, 8].min(axis=1)) * 10):.0f} ns")


if __name__ == "__main__":
    main()
