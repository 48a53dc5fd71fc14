#!/usr/bin/env python3
"""Kernel timeline of ONE whole sim step of a rocprofv3 kernel trace (csv): from the end of one dye advection to the end of the next.
usage: step_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.findall(r'(\w+_kernel|\w+Buffer\w*)', n)
    return m[0] if m else n[:40]
names = [short(r['Kernel_Name']) for r in rows]
dye = [i for i, n in enumerate(names) if n.startswith('advect_vec3uq32')]
full = [k for k in range(len(dye) - 1) if dye[k + 1] - dye[k] > 20]
k = full[-2]
a, b = dye[k] + 1, dye[k + 1] + 1
t0 = int(rows[dye[k]]['End_Timestamp'])
prev, nsor = {}, 0
for r, n in zip(rows[a:b], names[a:b]):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r['Queue_Id']
    gap = (s - prev[q]) / 1e3 if q in prev else float('nan')
    if n == 'sor_fused_kernel':
        nsor += 1
    if n != 'sor_fused_kernel' or gap > 1 or nsor == 1:
        print(f"+{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} us q{q} gap {gap:6.1f}  {n}")
    prev[q] = e
print("sor launches", nsor, "; from the end of one dye advection to the end of the next:", (int(rows[b - 1]['End_Timestamp']) - t0) / 1e3, "us")
