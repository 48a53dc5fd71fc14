#!/bin/bash
# r03: rows per tile (-> waves per SIMD) at 8192^2, NS = 16, with and without the rotating priority
set -u
export TMPDIR=/tmp
O=gpurun_out/r03_rpc
mkdir -p $O
for rep in 1 2; do
for v in L1R6 L4R2; do
for rpc in 0 200 280 344 400 520 683; do
  a=$(./tools/sor_clock_probe_ns16_$v 8192 8192 30 $rpc $O/ns16_${v}_$rpc.csv | tee $O/ns16_${v}_$rpc.txt | grep -E "waves traced|shader clock" | sed 's/.*rows_per_chunk [0-9]*: //;s/shader clock per wave.*median/median GHz/;s/p90.*//' | tr '\n' ' ')
  echo "rep $rep $v rpc $rpc: $a"
done
done
done
