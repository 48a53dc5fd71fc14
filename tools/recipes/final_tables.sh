#!/bin/bash
# The tables of the round after the profiles (tools/recipes/evidence.sh): the three bench lines with the committed PMC entries in them,
# the emulated ranks with both transports and a wire, the halo-depth sweeps.  ON THE GPU BOX:  gpurun -- bash tools/recipes/final_tables.sh
set -u
export TMPDIR=/tmp
R=${ROUND:-r06}   # the round the files are named after
mkdir -p gpurun_out/$R
python bench.py > gpurun_out/$R/bench_default.json 2> gpurun_out/$R/bench_default.err
python bench.py --size 61 --dim-y 81 --iters 20 --steps 50 --warmup 10 > gpurun_out/$R/bench_c1.json 2>/dev/null
python bench.py --size 2048 --iters 40 --steps 30 --warmup 5 > gpurun_out/$R/bench_c2.json 2>/dev/null
bash tools/recipes/emulate_ranks.sh ${R}_c4 8192 80 8 "0 3 7" "copy rccl" "0 25" 2 > /dev/null 2>&1
bash tools/recipes/emulate_ranks.sh ${R}_c5 16384 200 8 "3" "copy rccl" "0 25" 2 > /dev/null 2>&1
bash tools/recipes/halo_sweep.sh ${R}_c4 8192 80 8 3 "0 64 96 160" "0 25" "copy rccl" 2 > /dev/null 2>&1
bash tools/recipes/halo_sweep.sh ${R}_c5 16384 200 8 3 "0 64 96 160" "0 25" "copy rccl" 1 > /dev/null 2>&1
