#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table from hipcc's resource remarks (CPU-only: hipcc cross-compiles gfx950).

    python tools/resource_table.py [extra hipcc flags...]        # e.g. -DRDV_STAMPS
"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reinforcement_learning_rendezvous_amd", "csrc")


def table(extra=()):
    # the product's own Makefile (`make resource` compiles both translation units with their flags)
    p = subprocess.run(["make", "-C", CSRC, "resource", "EXTRA=" + " ".join(extra)], capture_output=True, text=True)
    if p.returncode:
        sys.exit(p.stderr[-4000:])
    rows, cur = [], None
    for line in (p.stdout + p.stderr).splitlines():
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z \[\]/]*): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip(), "SGPR spill": "0"}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


if __name__ == "__main__":
    rows = table(sys.argv[1:])
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch B/lane':>15s} {'occupancy':>9s} {'LDS B':>7s} {'SGPR spill':>10s}")
    for r in rows:
        name = re.sub(r"\(.*", "", r["name"]).replace("void ", "")
        print(f"{name[:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>15s} {r.get('Occupancy [waves/SIMD]', '?'):>9s} {r.get('LDS Size [bytes/block]', '?'):>7s} {r.get('SGPRs Spill', '?'):>10s}")
