#!/bin/bash
# r03: rotating issue priority -- levels x rows per turn, NS = 16 at 8192^2 and NS = 10 on the 8192 x 1024 slab
set -u
export TMPDIR=/tmp
O=gpurun_out/r03_prio_sweep
mkdir -p $O
for rep in 1 2 3; do
for v in L1R6 L4R6 L4R2 L4R1 L3R6 L3R2 L3R1 L2R3; do
  a=$(./tools/sor_clock_probe_ns16_$v 8192 8192 30 0 $O/ns16_$v.csv | tee $O/ns16_$v.txt | grep -E "waves traced" | sed 's/.*events //')
  b=$(./tools/sor_clock_probe_ns10_$v 8192 1024 60 0 $O/ns10_$v.csv | tee $O/ns10_$v.txt | grep -E "waves traced" | sed 's/.*events //')
  c=$(./tools/sor_clock_probe_ns10_$v 8192 1024 60 30 $O/ns10r30_$v.csv | tee $O/ns10r30_$v.txt | grep -E "waves traced" | sed 's/.*: \([0-9]*\) waves.*events /\1 waves /')
  echo "rep $rep $v: ns16 8192^2 $a | ns10 slab auto $b | ns10 slab rpc30 $c"
done
done
