#!/bin/bash
# r03: HIP API calls issued per sfl_step on slabs (no hipStreamSynchronize inside a step)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03_step_api
rm -rf $O; mkdir -p $O
for K in 5 25; do
  rocprofv3 --hip-trace --stats --output-format csv -d $O/k$K -o t -- python3 tools/r03/step_api_probe.py $K > $O/k$K.log 2>&1
done
python3 - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
root = sys.argv[1]
cnt = {}
for K in (5, 25):
    c = collections.Counter()
    for f in glob.glob(f"{root}/k{K}/**/*hip_api_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            c[r["Function"]] += 1
    cnt[K] = c
print("# HIP API calls per sfl_step on 4 virtual ranks (8192 x 2048, 80 iterations, automatic advection halo):")
print("# (calls in a run of 25 steps - calls in a run of 5 steps) / 20; tools/r03/step_api_trace.sh")
for fn in sorted(set(cnt[5]) | set(cnt[25])):
    d = (cnt[25][fn] - cnt[5][fn]) / 20.0
    if d or "ynchronize" in fn:
        print(f"{fn:40s} {d:8.2f} per step   ({cnt[5][fn]} / {cnt[25][fn]} calls in the two runs)")
PY
