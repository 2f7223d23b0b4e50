#!/bin/bash
# Runs on the GPU box: EXECUTED instruction counts of the step kernel (rocprofv3 --pmc, kernel trace only), per wave.
#   tools/inst_counters.sh <n_envs> <variant> [label]     (RDV_LIB=<other .so> for another build)
set -o pipefail
N=${1:-65536}; V=${2:-auto}; L=${3:-run}
R=$(pwd); OUT=$R/gpurun_out/inst_$L; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  d="$OUT/$(echo $c | tr ' ' '+' | cut -c1-60)"
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o p -- python3 "$R/tools/step_once.py" $N $V 6 f32 > "$d.log" 2>&1 || { tail -5 "$d.log"; }
done
python3 - <<PY
import csv, glob, collections, os
by = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            by[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
waves = {k: sum(v[-4:]) / len(v[-4:]) for (k, c), v in by.items() if c == "SQ_WAVES"}
print("kernel,counter,mean_of_last_4_dispatches,per_wave")
for (k, c), v in sorted(by.items()):
    m = sum(v[-4:]) / len(v[-4:])
    print(f'"{k}",{c},{m:.6g},{m / waves.get(k, 1):.1f}')
PY
