#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (rocprofv3 CSVs) into a short text summary:
per-kernel launch count / average duration from the kernel trace, and per-kernel average PMC
values from each counter pass.  FETCH_SIZE / WRITE_SIZE are reported raw (KiB) and converted to
bytes per launch with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE x2)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
# Steady state: bench.py first runs an untimed solve, a cold W + K region and ~40 priming solves while the clocks
# climb; only the LAST W + K solves are the timed region.  `tail` = how many trailing dispatches of each kernel
# count as steady state (default: a third of them); averages are printed for all dispatches and for the tail,
# and everything quoted in DESIGN.md / pmc_traffic.json is the tail.
tail_arg = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def tail_of(v):
    n = tail_arg if tail_arg > 0 else max(1, len(v) // 3)
    return v[-n:] if len(v) > n else v


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short  # noqa: E402  (one naming of kernels for the summaries and for pmc_traffic.json)


print(f"# profile summary of {os.path.basename(root)}")
for f in sorted(glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True)):
    dur = defaultdict(list)
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for row in rows:
        dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print("\n## kernel trace (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, root))
    print(f"{'kernel':90s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} | steady state (last n): {'n':>5s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        t = tail_of(v)
        print(f"{k:90s} {len(v):6d} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f} {sum(v)/1e6:10.3f} | "
              f"{'':23s}{len(t):5d} {sum(t)/len(t)/1e3:10.2f} {min(t)/1e3:10.2f} {max(t)/1e3:10.2f}")

for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        rows = list(csv.DictReader(open(f)))
        if rows and "Dispatch_Id" in rows[0]:
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        for row in rows:
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("\n## PMC pass:", os.path.relpath(f, root))
        for k, cs in sorted(acc.items()):
            for c, v in sorted(cs.items()):
                v = tail_of(v)   # steady state only
                avg = sum(v) / len(v)
                extra = ""
                if c == "FETCH_SIZE":
                    extra = f"  -> {avg * 1024 * 2 / 1e6:.1f} MB/launch read (x2 gfx950 correction), raw {avg * 1024 / 1e6:.1f} MB"
                if c == "WRITE_SIZE":
                    extra = f"  -> {avg * 1024 / 1e6:.1f} MB/launch written (uncalibrated)"
                print(f"{k:90s} {c:28s} n={len(v):4d} avg={avg:16.1f}{extra}")
