#!/bin/bash
# Kernel trace of full sim steps (ino:252-287 order) at the headline size.  Runs on the GPU box.
set -u
OUT=$PWD/gpurun_out/prof_step
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --sim-steps 5 > $OUT/stats.log 2>&1
python3 profiles/summarise_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
