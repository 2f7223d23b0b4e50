#!/bin/bash
# Runs on the GPU box: SQ instruction / wait counters of the two fused kernels at 4.2 M envs (separate --pmc passes, kernel trace only).
set -o pipefail
R=$(pwd); OUT=$R/gpurun_out/sq; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for v in fused fused_inlane; do
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    d="$OUT/${v}_$(echo $c | tr ' ' '+')"
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o p -- python3 "$R/tools/step_once.py" 4194304 $v 6 f32 > "$d.log" 2>&1 || exit 1
  done
done
python3 - <<PY
import csv, glob, collections, os
rows = []
for f in sorted(glob.glob("$OUT/*/p_counter_collection.csv")):
    variant = os.path.basename(os.path.dirname(f)).split("_SQ")[0]
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            by[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(by.items()):
        rows.append((variant, k, c, sum(v[-4:]) / len(v[-4:])))
with open("$OUT/summary.csv", "w") as fh:
    fh.write("variant,kernel,counter,mean_of_last_4_dispatches\n")
    for r in rows:
        fh.write(f'{r[0]},"{r[1]}",{r[2]},{r[3]:.6g}\n')
print(open("$OUT/summary.csv").read())
PY
