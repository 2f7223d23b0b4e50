#!/usr/bin/env python3
"""Diagnostic: launch time of rdv_step against the spacing of the seven chunk arrays of the workspace (RDV_CHUNK_ALIGN / RDV_CHUNK_SKEW,
read once per process: every configuration runs tools/n_sweep.py in its own child process; this parent never touches the GPU)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = sys.argv[1] if len(sys.argv) > 1 else "65536,131072,262144,524288,1048576,4194304"
configs = [(256, 0, "chunk arrays back to back (the layout until now)"), (65536, 0, "padded to 64 KiB, no skew"),
           (65536, 4096, "64 KiB + 4 KiB"), (65536, 8192, "64 KiB + 8 KiB"), (65536, 12288, "64 KiB + 12 KiB"), (65536, 20480, "64 KiB + 20 KiB"),
           (65536, 36864, "64 KiB + 36 KiB"), (2097152, 4096, "2 MiB + 4 KiB"), (2097152, 266240, "2 MiB + 260 KiB"), (256, 0, "back to back (again)")]
for align, skew, label in configs:
    env = dict(os.environ, RDV_CHUNK_ALIGN=str(align), RDV_CHUNK_SKEW=str(skew))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "n_sweep.py"), "--sizes", sizes, "--variants", "auto"], env=env,
                       capture_output=True, text=True, timeout=600)
    rows = [l.split(",") for l in r.stdout.splitlines() if l and l[0].isdigit()]
    print(f"{label:52s} " + "  ".join(f"{int(x[0]) // 1024}k:{float(x[2]):8.2f}" for x in rows) + ("" if rows else " FAILED " + r.stderr[-300:]), flush=True)
