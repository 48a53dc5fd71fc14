#!/bin/bash
# PMC passes (HBM traffic) over full sim steps at the headline size.  Runs on the GPU box.
set -u
OUT=$PWD/gpurun_out/prof_step_pmc
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --sim-steps 3"
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$N -o pmc -- python3 bench.py $ARGS > $OUT/pmc_$N.log 2>&1
done
python3 profiles/summarise_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
