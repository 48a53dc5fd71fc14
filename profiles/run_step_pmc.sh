#!/bin/bash
# Kernel trace + PMC passes (HBM traffic) over full sim steps at the headline size -- the kernels of the step outside the
# solve (bench.py times the steps once as n x sfl_step and once as sfl_step_n, so the two per-step kernels and the seam
# kernel all appear).  Runs on the GPU box: bash profiles/run_step_pmc.sh [tag]; then, with the summary copied to
# profiles/, python3 profiles/update_step_table.py gpurun_out/prof_step_<tag> profiles/<summary> <round>
set -u
OUT=$PWD/gpurun_out/prof_step_${1:-pmc}
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-fold-leg --sim-steps ${SIM_STEPS:-8}"   # sfl_step_n(n): n - 1 launches of the seam kernel per pass
# LIB=<path>: a variant build of the library (tools/recipes/build_variant.sh) instead of the product's, through tools/with_lib.py
RUN="python3 bench.py"; [ -n "${LIB:-}" ] && RUN="python3 tools/with_lib.py $LIB bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o trace -- $RUN $ARGS > $OUT/stats.log 2>&1
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$N -o pmc -- $RUN $ARGS > $OUT/pmc_$N.log 2>&1
done
python3 profiles/summarise_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
