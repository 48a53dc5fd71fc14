#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
for cfg in "1 0" "1 1" "1 0"; do
  timeout 600 python tools/r04/unaligned_stress.py ${1:-300} $cfg 2>&1 | tail -6 | tee -a $O/unaligned_stress_fixed.txt
done
