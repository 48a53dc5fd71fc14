#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
echo "all tiles wait:" | tee -a $O/unaligned_diag.txt
SFL_DEBUG_ALL_WAIT=1 timeout 600 python tools/r04/unaligned_stress.py 240 1 0 2>&1 | tail -4 | tee -a $O/unaligned_diag.txt
echo "no urgent priority:" | tee -a $O/unaligned_diag.txt
SFL_DEBUG_NO_URGENT=1 timeout 600 python tools/r04/unaligned_stress.py 240 1 0 2>&1 | tail -4 | tee -a $O/unaligned_diag.txt
echo "as is:" | tee -a $O/unaligned_diag.txt
timeout 600 python tools/r04/unaligned_stress.py 240 1 0 2>&1 | tail -4 | tee -a $O/unaligned_diag.txt
