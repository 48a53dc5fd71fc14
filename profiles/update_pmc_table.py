#!/usr/bin/env python3
"""Append the dominant kernel's counters of a profile directory (profiles/run_profile.sh) to
profiles/pmc_traffic.json, stamped with a hash of the kernel sources they were measured on -- bench.py quotes an
entry only while that hash matches the sources it runs (otherwise: compulsory bytes, "traffic_source": "stale").
usage: update_pmc_table.py <gpurun_out/prof_<tag>> <summary file under profiles/> <dim_x> <dim_y> <fuse> <round> [note]"""
import csv
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from bench import kernel_source_hash  # noqa: E402


def steady(v):
    n = max(1, len(v) // 3)
    return v[-n:]


def main():
    root, summary, dim_x, dim_y, fuse, rnd = sys.argv[1], sys.argv[2], *map(int, sys.argv[3:7])
    note = sys.argv[7] if len(sys.argv) > 7 else ""
    from kernel_names import short   # the row of the summary the numbers can be read from
    counters, dur, row = {}, [], None
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "sor_fused_kernel" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        by = {}
        for r in rows:
            if is_continuing(r["Kernel_Name"], fuse):
                by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                row = short(r["Kernel_Name"])
        for k, v in by.items():
            counters[k] = sum(steady(v)) / len(steady(v))
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows
               if "sor_fused_kernel" in r["Kernel_Name"] and is_continuing(r["Kernel_Name"], fuse)]
    if "FETCH_SIZE" not in counters or "WRITE_SIZE" not in counters or not dur:
        sys.exit(f"incomplete profile in {root}: {sorted(counters)} / {len(dur)} trace rows")
    d = steady(dur)
    read_b = counters["FETCH_SIZE"] * 1024 * 2       # gfx950: FETCH_SIZE reports half (MI355X_MICROARCH.md)
    write_b = counters["WRITE_SIZE"] * 1024
    entry = {"round": rnd, "grid": [dim_x, dim_y], "fuse": fuse, "n_gpus": 1,
             "kernel": row or f"sor_fused_kernel<Lane2, NS={fuse}, dx1=true, zero_in=false>", "note": note,
             "recompute": "traffic_bytes_per_launch = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 of this kernel's row in `source`; "
                          "avg_launch_us_rocprof = its steady-state avg_us in the kernel trace section",
             "kernel_source_sha16": kernel_source_hash(),
             "fetch_size_kib_raw": counters["FETCH_SIZE"], "read_bytes_per_launch": int(read_b),
             "write_bytes_per_launch": int(write_b), "traffic_bytes_per_launch": int(read_b + write_b),
             "tcc_ea0_rdreq": counters.get("TCC_EA0_RDREQ_sum"), "tcc_ea0_wrreq": counters.get("TCC_EA0_WRREQ_sum"),
             "valu_wave_insts_per_launch": counters.get("SQ_INSTS_VALU"), "waves_per_launch": counters.get("SQ_WAVES"),
             "grbm_gui_active": counters.get("GRBM_GUI_ACTIVE"), "sq_busy_cycles": counters.get("SQ_BUSY_CYCLES"),
             "sq_wave_cycles": counters.get("SQ_WAVE_CYCLES"),
             "avg_launch_us_rocprof": sum(d) / len(d) / 1e3, "steady_state_launches": len(d),
             "source": summary}
    path = os.path.join(HERE, "pmc_traffic.json")
    table = json.load(open(path))
    table["entries"].append(entry)
    json.dump(table, open(path, "w"), indent=1)
    print(json.dumps(entry, indent=1))


def is_continuing(name, fuse):
    import re
    m = re.search(r"Lane2<(\d+), (true|false), (true|false)", name)
    if m:
        return int(m.group(1)) == fuse and m.group(3) == "false"
    m = re.search(r"Lane2ILi(\d+)ELb([01])ELb([01])", name)
    return bool(m) and int(m.group(1)) == fuse and m.group(3) == "0"


if __name__ == "__main__":
    main()
