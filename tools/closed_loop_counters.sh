#!/bin/bash
# Runs on the GPU box: SQ instruction-mix / busy / wait counters of the closed loop (rollout_kernel, policy_act_kernel, step_kernel_split) at
# 65,536 envs; separate --pmc passes of three counters each, kernel trace only, the program directly after `--`.
#    bash tools/closed_loop_counters.sh     -> gpurun_out/clc/summary.csv  (per kernel: counter per dispatch, per wave, per env-step)
set -o pipefail
R=$(pwd); OUT=$R/gpurun_out/clc; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for c in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS" \
         "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F32" \
         "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F64" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVES"; do
  d="$OUT/$(echo $c | tr ' ' '+')"
  timeout -k 10 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o p -- python3 "$R/tools/closed_loop_once.py" > "$d.log" 2>&1 || { echo "pass $c failed"; tail -5 "$d.log"; }
  echo "pass $c done"
done
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in sorted(glob.glob("$OUT/*/p_counter_collection.csv")):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("rdv::rollout_kernel") or k.startswith("rdv::policy_act_kernel") or k.startswith("rdv::step_kernel_split"):
            by[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in by.items():
        v = v[2:] if len(v) > 4 else v
        rows[k][c] = sum(v) / len(v)
n = 65536
per = {"rdv::rollout_kernel<float, false>": n * 64, "rdv::policy_act_kernel": n, "rdv::step_kernel_split<float, 1>": n}
with open("$OUT/summary.csv", "w") as fh:
    fh.write("kernel,counter,mean_per_dispatch,per_env_step\n")
    for k in sorted(rows):
        for c in sorted(rows[k]):
            fh.write(f'"{k}",{c},{rows[k][c]:.6g},{rows[k][c] / per.get(k, n):.5g}\n')
print(open("$OUT/summary.csv").read())
PY
