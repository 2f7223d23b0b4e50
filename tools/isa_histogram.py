#!/usr/bin/env python3
"""Instruction histogram of one kernel from its gfx950 assembly (CPU-only: hipcc cross-compiles), by instruction class and by the
source function each instruction was inlined from, weighted with the issue costs tools/ubench_issue.hip measured on an MI355X.

    python tools/isa_histogram.py step_kernel_splitIfEE [--tu rdv_hip.hip] [--range A:B] [--marks] [-DFLAG ...]

--marks prints the positions of labels, branches and barriers (to choose --range, e.g. the step role of the split kernel);
--range restricts the histogram to instruction indices [A, B).
Cost columns: `lone` = shader cycles per instruction for ONE wave on its SIMD (the split kernel at 65,536 envs), `x4` = per instruction
per SIMD with four waves issuing (the fused kernels at large N).  Classes and weights: see COST below (profiles/r04_ubench_issue.txt)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "reinforcement_learning_rendezvous_amd", "csrc")
OUT = "/tmp/rdv_isa_hist"

# class -> (lone-wave cycles, four-waves-per-SIMD cycles) per instruction: profiles/r04_ubench_issue.txt
COST = {
    "f64 arith": (5.3, 2.7), "f64 rsq/rcp/sqrt": (20.1, 10.0), "f64 cvt/rndne/cmp": (5.4, 3.2), "f32/i32 alu": (5.0, 1.8),
    "v_mov": (5.0, 2.1), "int mul (quarter)": (5.5, 2.8), "mad_u64_u32": (9.1, 3.3), "f32 transcendental": (12.1, 6.0),
    "dpp/readlane/writelane": (8.9, 3.2), "v_cmp 32": (8.8, 3.2), "v_cndmask": (5.0, 1.8), "salu": (8.8, 3.3), "s_waitcnt/nop": (4.9, 1.2),
    "smem": (8.8, 3.3), "vmem": (8.8, 3.3), "lds": (8.8, 3.3), "branch": (8.8, 3.3), "other": (5.0, 2.0),
}


def classify(mn):
    if mn.startswith(("v_rsq_f64", "v_rcp_f64", "v_sqrt_f64")):
        return "f64 rsq/rcp/sqrt"
    if mn.startswith(("v_cvt_f64", "v_cvt_f32_f64", "v_cvt_i32_f64", "v_cvt_u32_f64", "v_rndne_f64", "v_cmp") ) and "f64" in mn:
        return "f64 cvt/rndne/cmp"
    if mn.startswith("v_") and "f64" in mn:
        return "f64 arith"
    if mn.startswith("v_mad_u64_u32") or mn.startswith("v_mad_i64_i32"):
        return "mad_u64_u32"
    if mn.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_lo_i32", "v_mul_hi_i32")):
        return "int mul (quarter)"
    if mn.startswith(("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag")):
        return "f32 transcendental"
    if mn.startswith(("v_readlane", "v_writelane", "v_readfirstlane")) or "dpp" in mn:
        return "dpp/readlane/writelane"
    if mn.startswith("v_mov") or mn.startswith("v_accvgpr"):
        return "v_mov"
    if mn.startswith("v_cndmask"):
        return "v_cndmask"
    if mn.startswith("v_cmp"):
        return "v_cmp 32"
    if mn.startswith("v_"):
        return "f32/i32 alu"
    if mn.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio")):
        return "s_waitcnt/nop"
    if mn.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache", "s_memtime", "s_memrealtime")):
        return "smem"
    if mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mn.startswith("ds_"):
        return "lds"
    return "other"


def function_map(path):
    """line -> name of the (device) function whose definition contains it: the last header `... name(` at column 0 above the line"""
    out, cur = {}, "?"
    hdr = re.compile(r"^(?:template\s*<[^>]*>\s*)?(?:static\s+|inline\s+|constexpr\s+|__device__\s+|__forceinline__\s+|__global__\s+|__host__\s+|__launch_bounds__\([^)]*\)\s+|__attribute__\(\(.*?\)\)\s+)*[\w:<>\*&\s]+?\b(\w+)\s*\(")
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return out
    for i, ln in enumerate(lines, 1):
        if ln and not ln[0].isspace() and not ln.startswith(("//", "#", "}", "{", "using", "namespace", "enum", "struct", "typedef", "extern")):
            m = hdr.match(ln)
            if m and m.group(1) not in ("if", "for", "while", "switch", "return", "static_assert", "asm"):
                cur = m.group(1)
        out[i] = cur
    return out


def main():
    args = sys.argv[1:]
    pat = args.pop(0)
    tu, rng, marks, extra = "rdv_hip.hip", None, False, []
    while args:
        a = args.pop(0)
        if a == "--tu":
            tu = args.pop(0)
        elif a == "--range":
            lo, hi = args.pop(0).split(":")
            rng = (int(lo), int(hi))
        elif a == "--marks":
            marks = True
        else:
            extra.append(a)
    os.makedirs(OUT, exist_ok=True)
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-mllvm", "-amdgpu-kernarg-preload-count=16"]
    if tu in ("rdv_tiles.hip", "rdv_general.hip"):
        flags += ["-mllvm", "-disable-machine-licm"]
    subprocess.run(["/opt/rocm/bin/hipcc"] + extra + flags + ["-gline-tables-only", "-save-temps", "-c", "-o", "/dev/null", os.path.join(CSRC, tu)],
                   cwd=OUT, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(os.path.join(OUT, tu.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    files, start, end = {}, None, None
    for i, ln in enumerate(text):
        m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", ln)
        if m:
            d, f = (m.group(2), m.group(3)) if m.group(3) else ("", m.group(2))
            files[int(m.group(1))] = os.path.join(d, f) if not os.path.isabs(f) else f
        if start is None and re.match(r"^_ZN\S*" + re.escape(pat) + r"\S*:", ln):
            start = i
        if start is not None and end is None and ".amdhsa_kernel" in ln and pat in ln:
            end = i
    if start is None:
        sys.exit(f"no kernel matching {pat}")
    fmaps = {}
    ins, loc = [], ("?", 0)
    for ln in text[start:end]:
        s = ln.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if s.endswith(":") and not s.startswith(";"):
            ins.append((loc, "LABEL", s))
            continue
        if not s or s.startswith((";", ".", "_Z")):
            continue
        s = s.split(";")[0].strip()
        ins.append((loc, s.split(None, 1)[0], s))
    real = [x for x in ins if x[1] != "LABEL"]
    if marks:
        k = 0
        for loc, mn, s in ins:
            if mn == "LABEL":
                if s.startswith(".LBB"):
                    print(f"{k:6d}  {s}")
                continue
            if mn.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm")):
                print(f"{k:6d}      {s}    [{os.path.basename(loc[0])}:{loc[1]}]")
            k += 1
        return
    if rng:
        real = real[rng[0]:rng[1]]
    by_class = collections.Counter()
    by_func = collections.defaultdict(collections.Counter)
    for loc, mn, s in real:
        c = classify(mn)
        by_class[c] += 1
        if loc[0] not in fmaps:
            fmaps[loc[0]] = function_map(loc[0])
        fn = fmaps[loc[0]].get(loc[1], "?")
        by_func[f"{os.path.basename(loc[0])}:{fn}"][c] += 1
    tot = sum(by_class.values())
    lone = sum(COST[c][0] * k for c, k in by_class.items())
    x4 = sum(COST[c][1] * k for c, k in by_class.items())
    print(f"{pat}" + (f" instructions [{rng[0]}, {rng[1]})" if rng else "") + f": {tot} instructions; estimated issue cycles: lone wave {lone:.0f}, per SIMD at four waves {x4:.0f}")
    print(f"{'class':26s} {'count':>6s} {'lone cyc':>9s} {'x4 cyc':>8s}")
    for c, k in by_class.most_common():
        print(f"{c:26s} {k:6d} {COST[c][0] * k:9.0f} {COST[c][1] * k:8.0f}")
    valu = sum(k for c, k in by_class.items() if c not in ("salu", "s_waitcnt/nop", "smem", "vmem", "lds", "branch", "other"))
    print(f"VALU instructions: {valu}")
    print()
    print(f"{'source function':44s} {'count':>6s} {'lone cyc':>9s}   classes")
    rows = sorted(by_func.items(), key=lambda kv: -sum(COST[c][0] * k for c, k in kv[1].items()))
    for fn, cnt in rows:
        n = sum(cnt.values())
        cyc = sum(COST[c][0] * k for c, k in cnt.items())
        print(f"{fn:44s} {n:6d} {cyc:9.0f}   " + ", ".join(f"{c} {k}" for c, k in cnt.most_common(5)))


if __name__ == "__main__":
    main()
