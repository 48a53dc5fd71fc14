#!/bin/bash
# r03: BASELINE config 2 (2048^2, 40 iterations): fuse depth x rows per tile, us per launch and per colour pass
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_c2
for ns in 8 10 12 14 16; do
for rpc in 0 12 16 24 32 48 64 96; do
  a=$(./tools/sor_clock_probe_ns$ns 2048 2048 60 $rpc | grep -E "waves traced" | sed 's/.*rows_per_chunk [0-9]*: //')
  us=$(echo "$a" | sed 's/.*events \([0-9.]*\) us/\1/')
  echo "NS $ns rpc $rpc: $a  -> $(python3 -c "print('%.3f us per pass, %.1f us per 80 passes' % ($us / $ns, $us / $ns * 80))")" | tee -a gpurun_out/r03_c2/sweep.txt
done
done
