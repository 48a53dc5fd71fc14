#!/bin/bash
# The round's evidence in one call ON THE GPU BOX (profiles of the headline kernel, the thin share and the step kernels, the
# three bench lines, timelines of an emulated rank with both transports, the rank tables):  gpurun -- bash tools/recipes/evidence.sh
# Copy what should be judged from gpurun_out/ into profiles/ (profiles/README_$R.md says what came from where).
set -u
export TMPDIR=/tmp
R=${ROUND:-r06}   # the round the files are named after
mkdir -p gpurun_out/$R
bash profiles/run_profile.sh ${R}_default > /dev/null 2>&1
bash profiles/run_profile.sh ${R}_slab1024 --size 8192 --dim-y 1024 > /dev/null 2>&1
SIM_STEPS=40 bash profiles/run_step_pmc.sh $R > /dev/null 2>&1
python bench.py > gpurun_out/$R/bench_default.json 2> gpurun_out/$R/bench_default.err
python bench.py --size 61 --dim-y 81 --iters 20 --steps 50 --warmup 10 > gpurun_out/$R/bench_c1.json 2>/dev/null
python bench.py --size 2048 --iters 40 --steps 30 --warmup 5 > gpurun_out/$R/bench_c2.json 2>/dev/null
bash tools/recipes/trace_timeline.sh rank3_rccl --emulate-rank 3 --of 8 --via-rccl > /dev/null 2>&1
bash tools/recipes/trace_timeline.sh rank3_copy --emulate-rank 3 --of 8 > /dev/null 2>&1
bash tools/recipes/emulate_ranks.sh ${R}_c4 8192 80 8 "0 3 7" "copy rccl" "0 25" 2 > /dev/null 2>&1
bash tools/recipes/emulate_ranks.sh ${R}_c5 16384 200 8 "3" "copy rccl" "0 25" 2 > /dev/null 2>&1
tail -3 gpurun_out/emulate_${R}_c4.txt
