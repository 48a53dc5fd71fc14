#!/usr/bin/env python3
"""Timeline of the LAST solve in a rocprofv3 kernel trace (csv): start offset, duration, queue, grid size of
every dispatch, and the gaps between consecutive dispatches of the busiest queue.
usage: timeline.py <kernel_trace.csv> [launches per solve on the main queue, default: auto]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sor = [r for r in rows if "sor_fused" in r["Kernel_Name"]]
if not sor:
    sys.exit("no sor_fused dispatches")
# a solve starts with the zero_in instantiation (template argument ..., true>) -- find the last one
starts = [i for i, r in enumerate(rows) if "sor_fused" in r["Kernel_Name"] and r["Kernel_Name"].rstrip(">) ").endswith("true")]
def is_first(r):
    n = r["Kernel_Name"]
    return "sor_fused" in n and ("ELb1EEEvPf" in n or n.replace(" ", "").endswith("true>(float*,floatconst*,floatconst*,sfl::Slab,sfl::sor::Tiling,sfl::sor::Tiling,sfl::SorParams)"))
firsts = [i for i, r in enumerate(rows) if "sor_fused" in r["Kernel_Name"] and "true>" in r["Kernel_Name"].replace(" ", "").split("(")[0][-8:]]
if len(firsts) < 2:
    firsts = starts
# use the second to last solve (complete for sure)
a = firsts[-2] if len(firsts) >= 2 else 0
b = firsts[-1] if len(firsts) >= 2 else len(rows)
sel = rows[a:b]
t0 = int(sel[0]["Start_Timestamp"])
print(f"solve: {len(sel)} dispatches, {(int(sel[-1]['End_Timestamp']) - t0) / 1e3:.1f} us from first start to last end; "
      f"next solve starts at {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us" if b < len(rows) else "")
prev_end = {}
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    name = r["Kernel_Name"]
    short = "sor_fused" + ("(zero_in)" if r is sel[0] else "") if "sor_fused" in name else name.split("(")[0][-40:]
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else float("nan")
    wg = r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))
    grid = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    print(f"  +{(s - t0) / 1e3:8.1f} us  {((e - s) / 1e3):7.1f} us  queue {q:>3}  gap on queue {gap:6.1f}  grid {grid:>8}  {short}")
    prev_end[q] = e
