#!/usr/bin/env python3
"""Diagnostic (not product): where a wave of the tile-loop step kernel (step_kernel_tiles, csrc/rdv_tiles.hip) spends its life.

    python tools/stamp_profile_tiles.py 4194304 [grid]

Builds a -DRDV_STAMPS copy of the library into tools/_stamps.so and runs the batch with variant="fused_tiles".  Per wave the kernel sums the
shader cycles of each phase over its tiles:
  1 waiting for the FIRST tile's inputs | 2 transition (the look-ahead fetch is issued inside it) | 3 statistics, outputs, state stores |
  4 barrier 1 | 5 reset parts | 6 barrier 2 | 11 waiting for the look-ahead (next tile's inputs) | 7 row stores
plus the number of tiles and s_memrealtime at entry and exit.  Never quote this build's run time."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
    if len(sys.argv) > 2:
        os.environ["RDV_TILES_GRID"] = sys.argv[2]
    lib = os.path.join(ROOT, "tools", "_stamps.so")
    from _build import build_variant
    build_variant(lib, ["-DRDV_STAMPS"])
    import torch
    from reinforcement_learning_rendezvous_amd import _native
    _native.LIB_PATH = lib
    from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
    env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0, variant="fused_tiles")
    L = _native.lib()
    L.rdv_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    waves = ((n + 255) // 256 + 8) * 4
    stamps = torch.zeros((waves, 12), dtype=torch.int64, device="cuda:0")
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
    s = s[:, s[0, :, 10] != 0, :]                    # waves that ran
    tiles = s[:, :, 10]
    real = (s[:, :, 9] - s[:, :, 8]) * 10.0          # ns
    cyc = s[:, :, [1, 2, 3, 4, 5, 6, 7, 11]].sum(axis=2)
    print(f"{n} envs, {s.shape[1]} waves of the tile loop, {np.median(tiles):.0f} tiles per wave (median); shader clock while the kernel runs: "
          f"{np.median(cyc / real):.2f} GHz")
    names = {1: "waiting for the first tile's inputs (per wave)", 2: "transition (look-ahead issued inside)", 3: "statistics + outputs + state store",
             4: "barrier 1 wait", 5: "reset parts", 6: "barrier 2 wait", 11: "waiting for the look-ahead", 7: "row stores"}
    for k in (1, 2, 3, 4, 5, 6, 11, 7):
        per = s[:, :, k] / (1.0 if k == 1 else tiles)
        print(f"  {names[k]:48s} median {np.median(per):8.0f}  p90 {np.percentile(per, 90):8.0f} cycles" + ("" if k == 1 else " per tile"))
    per_tile = (cyc - s[:, :, 1]) / tiles
    print(f"  a tile takes a wave {np.median(per_tile):.0f} cycles (median), wave lifetime: median {np.median(real):.0f} ns")
    span = (s[:, :, 9].max(axis=1) - s[:, :, 8].min(axis=1)) * 10.0
    print(f"  launch as the waves see it (first entry -> last exit): {np.median(span) / 1e3:.1f} us; resident waves on average: "
          f"{np.median(real.sum(axis=1) / span):.0f}")


if __name__ == "__main__":
    main()
