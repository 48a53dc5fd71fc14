#!/bin/bash
# final round-2 profiles: default config (kernel trace + PMC passes), the 8-GPU share, the sim step
set -u
export TMPDIR=/tmp
bash profiles/run_profile.sh r02_final > gpurun_out/r02_run37_default.txt 2>&1
bash profiles/run_profile.sh r02_slab1024 --dim-y 1024 > gpurun_out/r02_run37_slab1024.txt 2>&1
bash profiles/run_step_trace.sh > gpurun_out/r02_run37_step_trace.txt 2>&1
bash profiles/run_step_pmc.sh > gpurun_out/r02_run37_step_pmc.txt 2>&1
cp gpurun_out/prof_r02_final/stats/*kernel_stats.csv gpurun_out/r02_run37_kernel_stats.csv 2>/dev/null || find gpurun_out/prof_r02_final/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_run37_kernel_stats.csv \;
python bench.py > gpurun_out/r02_run37_bench.json 2> gpurun_out/r02_run37_bench.err
head -12 gpurun_out/r02_run37_default.txt; grep -E "FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU |SQ_WAVES " gpurun_out/r02_run37_default.txt | grep sor_fused; grep -E "FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU |SQ_WAVES |avg_us" gpurun_out/r02_run37_slab1024.txt | grep -E "sor_fused|avg_us"
