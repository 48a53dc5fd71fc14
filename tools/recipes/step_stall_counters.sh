set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_seam_stalls
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-fold-leg --sim-steps 16"
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$N -o pmc -- python3 bench.py $ARGS > $OUT/pmc_$N.log 2>&1
done
python3 profiles/summarise_profile.py $OUT > $OUT/summary.txt 2>&1
grep -E "^seam|^advect_vec3uq32_tiled_kernel<no_slip=false, fuse_grad=true|^advect_divergence" $OUT/summary.txt | cut -c1-60,91-160
