#!/usr/bin/env python3
"""Append the counters of the sim step's kernels outside the solve (profiles/run_step_pmc.sh) to
profiles/pmc_traffic.json under "step_entries", stamped with a hash of the sources those kernels are compiled from
(advect_tiled.hip, stencil_kernels.hip, advect_math.h) -- bench.py quotes an entry only while that hash matches.
usage: update_step_table.py <gpurun_out/prof_step_<tag>> <summary file under profiles/> <round> [dim_x dim_y]"""
import csv
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from kernel_names import short  # noqa: E402  (the row of the summary the numbers can be read from)
from bench import HBM_PEAK_GBS, STEP_KERNELS, step_kernel_source_hash  # noqa: E402


def steady(v):
    n = max(1, len(v) // 3)
    return v[-n:]


def which(name):
    for key, rec in STEP_KERNELS.items():
        if rec["match"](name):
            return key
    return None


def main():
    root, summary, rnd = sys.argv[1], sys.argv[2], int(sys.argv[3])
    dim_x, dim_y = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (8192, 8192)
    counters, dur, rows_named = {}, {}, {}
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        by = {}
        for r in rows:
            k = which(r["Kernel_Name"])
            if k:
                by.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
                rows_named[k] = short(r["Kernel_Name"])
        for (k, c), v in by.items():
            counters.setdefault(k, {})[c] = sum(steady(v)) / len(steady(v))
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True):
        for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
            k = which(r["Kernel_Name"])
            if k:
                dur.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    path = os.path.join(HERE, "pmc_traffic.json")
    table = json.load(open(path))
    table.setdefault("step_entries", [])
    cells = dim_x * dim_y
    for k, rec in STEP_KERNELS.items():
        c, d = counters.get(k, {}), steady(dur.get(k, []))
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or not d:
            print(f"{k}: incomplete ({sorted(c)} / {len(d)} trace rows), skipped")
            continue
        read_b = c["FETCH_SIZE"] * 1024 * 2      # gfx950: FETCH_SIZE reports half (MI355X_MICROARCH.md)
        write_b = c["WRITE_SIZE"] * 1024
        us = sum(d) / len(d) / 1e3
        entry = {"round": rnd, "grid": [dim_x, dim_y], "kernel": k, "summary_row": rows_named.get(k), "does": rec["does"],
                 "kernel_source_sha16": step_kernel_source_hash(),
                 "algorithmic_bytes_per_cell": rec["bytes_per_cell"], "algorithmic_bytes_per_launch": rec["bytes_per_cell"] * cells,
                 "read_bytes_per_launch": int(read_b), "write_bytes_per_launch": int(write_b),
                 "traffic_bytes_per_launch": int(read_b + write_b), "avg_launch_us_rocprof": us,
                 "steady_state_launches": len(d),
                 "tcc_hit_rate": (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])) if "TCC_HIT_sum" in c else None,
                 "frac_of_hbm_peak": (read_b + write_b) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                 "algorithmic_frac_of_hbm_peak": rec["bytes_per_cell"] * cells / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                 "source": summary}
        table["step_entries"].append(entry)
        print(json.dumps(entry, indent=1))
    json.dump(table, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
