#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh'): the measurements profiles/ is built from.
#   1. bench.py (default flags, and the driver's --steps 20 --warmup 5)             -> gpurun_out/refresh/bench_*.json
#   2. rocprofv3 --kernel-trace --stats of the default bench command                 -> gpurun_out/refresh/kstats/
#      and of `bench.py --only-large-n N` for N = 524,288 and 4,194,304 (one size per run) -> gpurun_out/refresh/kstats_large_N/
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per workload        -> gpurun_out/refresh/pmc/<key>_<COUNTER>/
# Counters are collected without any other trace domain (only --kernel-trace), the program after `--` is python3 itself.
# Two parts (a gpurun call is limited to 20 minutes): `refresh_profiles.sh bench` = 1 + 2, `refresh_profiles.sh pmc` = 3; no argument = both.
set -o pipefail
PART=${1:-all}
R=$(pwd)
OUT=$R/gpurun_out/refresh
mkdir -p "$OUT/pmc"
export TMPDIR=/tmp
if [ "$PART" = all ] || [ "$PART" = bench ]; then
timeout -k 10 400 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || exit 1
echo "bench default done"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err" || exit 1
echo "bench driver flags done"
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kstats" -o p -- python3 "$R/bench.py" > "$OUT/bench_profiled.json" 2> "$OUT/bench_profiled.err" || exit 1
echo "kernel stats done"
# one row of step_kernel_parts PER SIZE: the large-batch legs alone, each in its own traced run (bench.py --only-large-n)
for N in 524288 4194304; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kstats_large_$N" -o p -- python3 "$R/bench.py" --only-large-n $N > "$OUT/bench_large_$N.json" 2> "$OUT/bench_large_$N.err" || exit 1
  echo "kernel stats, $N envs alone, done"
done
cd "$R"
fi
if [ "$PART" = all ] || [ "$PART" = pmc ]; then
cd /tmp
pmc() {   # key, script, args...
  local key=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc/${key}_$c" -o p -- python3 "$@" > "$OUT/pmc/${key}_$c.log" 2>&1 || return 1
  done
  echo "pmc $key done"
}
pmc f32_65536 "$R/tools/step_once.py" 65536 auto 8 f32 &&
pmc f64_65536 "$R/tools/step_once.py" 65536 auto 8 f64 &&
pmc f32_524288 "$R/tools/step_once.py" 524288 auto 8 f32 &&
pmc f32_4194304 "$R/tools/step_once.py" 4194304 auto 8 f32 &&
pmc persist "$R/tools/persistent_once.py"
# one run holds both persistent kernels: tools/pmc_to_json.py picks each by name from a directory per key
for c in FETCH_SIZE WRITE_SIZE; do
  cp -r "$OUT/pmc/persist_$c" "$OUT/pmc/step_many_f32_65536_K64_$c" && mv "$OUT/pmc/persist_$c" "$OUT/pmc/rollout_f32_65536_T64_$c" || exit 1
done
fi
find "$OUT" -name "*.csv" -size +20M -delete      # kernel traces of the long bench run are not needed, the stats are
echo "refresh $PART complete"
