#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final4; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "slab_steps or automatic_advection or config4_whole or overlapped_exchange_inside or emulated or both_advection_kernels_on_slabs or step or early_interior" ) 2>&1 | grep -E "^E  |passed|failed" | head
bash tools/r04/run40.sh | tail -8
head -12 $O/../r04_final3/step_timeline_new.txt
