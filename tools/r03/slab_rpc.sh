#!/bin/bash
# r03: the 8192 x 1024 share, NS = 10 / 8 / 12: rows per tile around two and three waves per SIMD
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_slab_rpc
for ns in 10 8 12; do
for rpc in 0 22 24 25 26 27 28 30 34 38 43 48; do
  ./tools/sor_clock_probe_ns$ns 8192 1024 60 $rpc > gpurun_out/r03_slab_rpc/x.txt
  a=$(grep -E "waves traced" gpurun_out/r03_slab_rpc/x.txt | sed 's/.*rows_per_chunk [0-9]*: //')
  h=$(grep -E "histogram" gpurun_out/r03_slab_rpc/x.txt | sed 's/.*histogram://')
  echo "NS $ns rpc $rpc: $a |$h" | tee -a gpurun_out/r03_slab_rpc/sweep.txt
done
done
