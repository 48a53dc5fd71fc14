#!/bin/bash
# r04: does a 2-level issue-priority rotation equalise the two waves of a SIMD on the thin share?  (r03 only tried 4 levels there:
# with wave slots 0 and 1 the younger wave is then favoured in three of four phases -- the order is reversed, not levelled)
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( time python -m pytest tests -m gpu -x -q -k "generic or external or dropin or domain_for_each" ) > $O/pytest_gpu_generic.log 2>&1
tail -3 $O/pytest_gpu_generic.log
for rep in 1 2 3; do
for v in ns10 ns10_L2R1 ns10_L2R2 ns10_L2R3 ns10_L2R6 ns10_L4R2; do
  ./tools/sor_clock_probe_$v 8192 1024 40 0 $O/probe_$v.csv > $O/probe_$v.txt 2>&1
  echo "$v: $(head -1 $O/probe_$v.txt | sed 's/.*events //')  $(grep 'lifetime, shader' $O/probe_$v.txt)" | tee -a $O/prio2_probe.txt
done; done
