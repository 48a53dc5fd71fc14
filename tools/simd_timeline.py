#!/usr/bin/env python3
"""Per-SIMD view of a sor_clock_probe CSV: when does the k-th wave of each SIMD finish, and how many
shader cycles did a wave of each finishing rank live?  usage: simd_timeline.py trace.csv [...]"""
import collections
import csv
import statistics as st
import sys

for fn in sys.argv[1:]:
    rows = list(csv.DictReader(open(fn)))
    simd = collections.defaultdict(list)
    for r in rows:
        hw, xcc = int(r["hwid"]), int(r["xcc"])
        key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3)
        simd[key].append((int(r["w1"]) / 100.0, int(r["t1"]) - int(r["t0"]), int(r["kind"])))
    ends = collections.defaultdict(list)
    cyc = collections.defaultdict(list)
    for v in simd.values():
        v.sort()
        for i, x in enumerate(v):
            ends[(len(v), i)].append(x[0])
            cyc[(len(v), i)].append(x[1])
    print(f"{fn}: {len(rows)} waves on {len(simd)} SIMDs")
    for k in sorted(ends):
        print(f"  SIMDs holding {k[0]} waves, {k[1] + 1}. to finish (n={len(ends[k])}): end us min {min(ends[k]):.1f} "
              f"median {st.median(ends[k]):.1f} max {max(ends[k]):.1f}; lifetime {st.median(cyc[k]):.0f} cycles")
