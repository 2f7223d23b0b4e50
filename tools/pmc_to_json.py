#!/usr/bin/env python3
"""Builds profiles/pmc_traffic.json from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as
MI355X_MICROARCH.md prescribes: the two counters do not fit one pass; FETCH_SIZE x2 for 16-byte-per-lane streams on gfx950).

    python tools/pmc_to_json.py gpurun_out/refresh/pmc [profiles/rNN_pmc_summary.csv]      # expects <dir>/<key>_FETCH_SIZE/p_counter_collection.csv, ..._WRITE_SIZE/...

key = <storage>_<n_envs> for rdv_step, step_many_<storage>_<n>_K64, rollout_<storage>_<n>_T64 (tools/step_once.py, persistent_once.py)."""
import collections
import csv
import datetime
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(path, kernel_substr, skip=2):
    rows = [r for r in csv.DictReader(open(path)) if kernel_substr in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows][skip:]
    return (sum(vals) / len(vals), rows[0]["Kernel_Name"].split("(")[0]) if vals else (None, None)


def main():
    d = sys.argv[1]
    out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    keys = sorted({re.sub(r"_(FETCH|WRITE)_SIZE$", "", x) for x in os.listdir(d) if x.endswith("_SIZE")})
    for key in keys:
        if key.startswith("step_many"):
            sub, per = "step_many_kernel", 64
        elif key.startswith("rollout"):
            sub, per = "rollout_kernel", 64
        else:
            sub, per = "step_kernel", 1
        f, name = mean_counter(os.path.join(d, key + "_FETCH_SIZE", "p_counter_collection.csv"), sub)
        w, _ = mean_counter(os.path.join(d, key + "_WRITE_SIZE", "p_counter_collection.csv"), sub)
        if f is None or w is None:
            continue
        n = int(re.search(r"_(\d+)", key).group(1))
        rec = {"kernel": name, "fetch_size_kib": f, "write_size_kib": w, "fetch_correction": 2.0, "read_bytes": f * 2 * 1024, "write_bytes": w * 1024,
               "bytes_per_launch": f * 2 * 1024 + w * 1024, "bytes_per_env_step": (f * 2 * 1024 + w * 1024) / (n * per),
               "algorithmic_bytes": 293 * n * per if "f32" in key else None,
               "note": "MI355X_MICROARCH.md HBM section: counters are KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a 16-B-per-lane "
                       "coalesced read stream (x2 applied); WRITE_SIZE is exact for 16-B-per-lane stores",
               "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python3 tools/step_once.py | persistent_once.py "
                          "(two separate passes; mean over the dispatches after the first two)",
               "collected": datetime.date.today().isoformat(), "round": 4}
        out[key] = rec
        print(key, name, f"{rec['bytes_per_launch'] / 1e6:.2f} MB per launch, {rec['bytes_per_env_step']:.1f} B per env-step")
    json.dump(out, open(out_path, "w"), indent=1)
    # the per-pass summary that goes with it (profiles/rNN_pmc_summary.csv): every kernel of every pass, mean after the first two
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as fh:
            fh.write("workload,kernel,counter,dispatches,mean_KiB_after_first_two\n")
            for sub in sorted(x for x in os.listdir(d) if x.endswith("_SIZE")):
                counter = sub.rsplit("_", 2)[1] + "_SIZE"
                by = collections.defaultdict(list)
                for r in csv.DictReader(open(os.path.join(d, sub, "p_counter_collection.csv"))):
                    if "rdv::" in r["Kernel_Name"]:
                        by[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
                for k, v in sorted(by.items()):
                    if len(v) > 2:
                        fh.write(f'{sub},"{k}",{counter},{len(v)},{sum(v[2:]) / len(v[2:]):.3f}\n')


if __name__ == "__main__":
    main()
