#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel trace + PMC passes of the headline bench command.
# Usage: bash profiles/run_profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,pmc_*}/...; copy the summaries you want judged into profiles/.
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --sim-steps 0 --no-fold-leg $*"
# 1. kernel trace + stats (no counters in this pass)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o trace -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
# 2. counters, one block of counters per pass (kernel-trace only, as the guide prescribes)
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$N -o pmc -- python3 bench.py $ARGS > $OUT/pmc_$N.log 2>&1
done
# summarise: per-kernel averages
python3 profiles/summarise_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
