#!/usr/bin/env python3
"""Where a kernel's VGPR pressure peaks, from its gfx950 assembly (CPU-only: hipcc cross-compiles).

    python tools/vgpr_pressure.py 'step_kernel_partsIfEE' [-DFLAG ...]

Compiles csrc/rdv_hip.hip with -save-temps -gline-tables-only into /tmp, cuts the kernel out of the .s, runs a LINEAR backward
liveness over its instructions (branches ignored: divergent regions are emitted in line with exec masking, which is what the
allocator sees too) and prints the pressure profile by source line: the top source lines by live VGPRs, and the live count at
every .loc change.  An estimate — the allocator's own number is the `VGPRs:` remark (tools/resource_table.py) — but it shows WHERE
the peak is and what is live there.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "reinforcement_learning_rendezvous_amd", "csrc")
OUT = "/tmp/rdv_isa"

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
ALL_USE = ("v_cmp", "v_cmpx", "global_store", "ds_write", "buffer_store", "scratch_store", "flat_store", "v_readlane", "v_readfirstlane",
           "ds_store", "global_atomic", "s_", "v_nop", "ds_bpermute_dummy")
DEF_AND_USE = ("v_fmac", "v_mac", "v_writelane", "v_swap", "v_pk_fmac", "v_dot")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def main():
    pat = sys.argv[1]
    extra = sys.argv[2:]
    os.makedirs(OUT, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc"] + extra + ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-mllvm",
                                             "-amdgpu-kernarg-preload-count=16", "-gline-tables-only", "-save-temps", "-c", "-o", "/dev/null",
                                             os.path.join(CSRC, "rdv_hip.hip")]
    subprocess.run(cmd, cwd=OUT, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(os.path.join(OUT, "rdv_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    files = {}
    start = end = None
    for i, ln in enumerate(text):
        m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", ln)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
        if start is None and re.match(r"^_ZN\S*" + re.escape(pat) + r"\S*:", ln):
            start = i
        if start is not None and end is None and ".amdhsa_kernel" in ln and pat in ln:
            end = i
    if start is None:
        sys.exit(f"no kernel matching {pat}")
    ins = []      # (loc, mnemonic, defs, uses, text)
    loc = ("?", 0)
    for ln in text[start:end]:
        s = ln.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((";", ".", "_Z")) or s.endswith(":"):
            continue
        s = s.split(";")[0].strip()
        parts = s.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if mn.startswith(ALL_USE):
            d, u = [], [r for o in ops for r in regs(o)]
        else:
            d = regs(ops[0]) if ops else []
            u = [r for o in ops[1:] for r in regs(o)]
            if mn.startswith(DEF_AND_USE) or "dpp" in s or "row_" in s:
                u += d
        ins.append((loc, mn, d, u, s))
    live = set()
    pressure = [0] * len(ins)
    live_at = [None] * len(ins)
    for i in range(len(ins) - 1, -1, -1):
        _, _, d, u, _ = ins[i]
        live -= set(d)
        live |= set(u)
        pressure[i] = len(live)
        live_at[i] = frozenset(live)
    by_line = collections.OrderedDict()
    for (loc, _, _, _, _), p in zip(ins, pressure):
        by_line[loc] = max(by_line.get(loc, 0), p)
    peak = max(pressure)
    print(f"{len(ins)} instructions, estimated peak {peak} live VGPRs")
    print("top source lines by live VGPRs:")
    for loc, p in sorted(by_line.items(), key=lambda kv: -kv[1])[:25]:
        print(f"  {p:4d}  {loc[0]}:{loc[1]}")
    # what is live at the peak: for each live register, where it was last written (source line of the defining instruction)
    at = os.environ.get("AT")
    ipk = int(at) if at else max(range(len(ins)), key=lambda k: pressure[k])
    last_def = {}
    for i in range(ipk):
        for r in ins[i][2]:
            last_def[r] = i
    groups = collections.defaultdict(list)
    for r in sorted(live_at[ipk]):
        j = last_def.get(r)
        groups[(ins[j][0], ins[j][4][:50]) if j is not None else (("entry", 0), "")].append(r)
    print(f"live before instruction {ipk} ({ins[ipk][0][0]}:{ins[ipk][0][1]}  {ins[ipk][4][:60]}), by defining instruction:")
    for (loc, txt), rs in sorted(groups.items(), key=lambda kv: (kv[0][0][0], kv[0][0][1])):
        print(f"  {loc[0]}:{loc[1]:<5d} {len(rs):3d}  v{rs}  <- {txt}")
    print("profile (instruction index, live, source line) at every change of source line:")
    last = None
    for i, ((loc, mn, _, _, s), p) in enumerate(zip(ins, pressure)):
        if loc != last:
            print(f"  {i:5d} {p:4d}  {loc[0]}:{loc[1]}   {s[:60]}")
            last = loc


if __name__ == "__main__":
    main()
