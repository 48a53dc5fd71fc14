#!/bin/bash
# r03: rotating issue priority on / off across slab sizes (NS = 16 kernel)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_prio_sizes
for rep in 1 2 3; do
for grid in "16384 16384" "8192 4096" "8192 2048" "16384 2048"; do
  set -- $grid
  a=$(./tools/sor_clock_probe_ns16_noprio $1 $2 15 0 | grep -E "waves traced" | sed 's/.*rows_per_chunk [0-9]*: //')
  b=$(PROBE_ALTERNATE=0 ./tools/sor_clock_probe_ns16 $1 $2 15 0 | grep -E "waves traced" | sed 's/.*events //')
  echo "rep $rep $1x$2: no rotation: $a | rotation: $b" | tee -a gpurun_out/r03_prio_sizes/summary.txt
done
done
