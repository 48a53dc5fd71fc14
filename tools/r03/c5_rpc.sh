#!/bin/bash
# r03: rows per tile for the multi-round launches of BASELINE config 5 (16384^2 and its 2048-row share), NS = 16
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_c5
for grid in "16384 2048" "16384 16384"; do
set -- $grid
for rpc in 0 90 100 114 128 150 171 205 256 300 340 400 455 520; do
  a=$(./tools/sor_clock_probe_ns16 $1 $2 12 $rpc | grep -E "waves traced" | sed 's/.*rows_per_chunk [0-9]*: //')
  echo "$1 x $2 rpc $rpc: $a" | tee -a gpurun_out/r03_c5/sweep.txt
done
done
