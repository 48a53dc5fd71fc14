#!/bin/bash
# LDS / VALU counters of the sim-step kernels (tiled advection, fused SOR)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_step_lds
mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --sim-steps 3"
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$N -o pmc -- python3 bench.py $ARGS > $OUT/pmc_$N.log 2>&1
done
python3 profiles/summarise_profile.py $OUT > gpurun_out/r02_run42_step_lds.txt 2>&1
grep -E "tiled|sor_fused" gpurun_out/r02_run42_step_lds.txt | grep -E "BANK_CONFLICT|IDX_ACTIVE|INSTS_LDS|INSTS_VALU " | head -40
