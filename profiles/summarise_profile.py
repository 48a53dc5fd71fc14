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


def short(name):
    import re
    # sor_fused_kernel<Lane2<NS, VEC, ZERO_IN[, NT]> | Lane4<NS, ZERO_IN>, NS, DX1, ZERO_IN>
    # (the store policy ST is an int since round 4 -- 0 plain, 2 nt, 16 sc1 -- and was a bool NT before)
    # (... and a load policy LD since the chained launch: Lane2<NS, VEC, ZERO_IN, ST, LD>.  Every template argument that tells two
    # instantiations apart is kept: NS, dx1, zero_in and the cache policies -- VERDICT r04: a truncated name had merged the
    # zero_in = true and zero_in = false kernels into one row)
    m = re.search(r"(sor_fused_kernel|sor_chain_kernel)<.*?Lane(\d)<(\d+)((?:, (?:true|false))+)((?:, \d+)*)>, \d+, (true|false)(?:, (true|false))?>", name)
    if m:
        flags = [f == "true" for f in re.findall(r"true|false", m.group(4))]
        pol = re.findall(r"\d+", m.group(5))
        st = {"2": "nt", "16": "sc1"}.get(pol[0] if pol else "", "nt" if (m.group(2) == "2" and len(flags) >= 3 and flags[2]) else "")
        ld = {"16": "+ldsc1"}.get(pol[1] if len(pol) > 1 else "", "")
        zero = f", zero_in={m.group(7)}" if m.group(7) else ""
        return f"{m.group(1)}<Lane{m.group(2)}{st}{ld}, NS={m.group(3)}, dx1={m.group(6)}{zero}>"
    m = re.search(r"(advect_divergence_tiled_kernel|advect_vec2f_tiled_kernel|advect_vec3uq32_tiled_kernel)<([^>]*)>", name)
    if m:
        flags = re.findall(r"true|false", m.group(2))
        label = {"advect_divergence_tiled_kernel": ["no_slip"], "advect_vec2f_tiled_kernel": ["no_slip", "self"],
                 "advect_vec3uq32_tiled_kernel": ["no_slip", "fuse_grad", "reach"]}[m.group(1)]
        return m.group(1) + "<" + ", ".join(f"{a}={b}" for a, b in zip(label, flags)) + ">"
    for key in ("seam_tiled_kernel", "advect_channels_kernel", "copy_bands_kernel", "signal_arrival_kernel"):
        if key in name:
            return key
    for key in ("divergence_tiled_kernel", "gradient_tiled_kernel"):
        if key in name:
            return key
    for key in ("divergence_stream_kernel", "gradient_stream_kernel", "sor_half_sweep_kernel", "advect_vec2f_kernel", "advect_vec3uq32_kernel",
                "divergence_kernel", "subtract_gradient_kernel", "zero_rows_kernel", "apply_forces_kernel"):
        if key in name:
            return key
    # anything else: the demangled name without its argument list, template arguments kept (abbreviated, never cut off in
    # the middle of what distinguishes two kernels)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"\b(void|sfl::|\(anonymous namespace\)::)", "", name).strip()
    return name if len(name) <= 110 else name[:107] + "..."


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
