#!/bin/bash
# r03: rotating issue priority (SFL_PRIO_LEVELS = 1 off, 2, 3, 4), NS = 16 at 8192^2
set -u
export TMPDIR=/tmp
O=gpurun_out/r03_prio
mkdir -p $O
for rep in 1 2; do
for L in 1 2 3 4; do
  ./tools/sor_clock_probe_ns16_prio$L 8192 8192 30 0 $O/prio$L.csv > $O/prio$L.txt 2>&1
  echo "== levels $L"; grep -E "waves traced|shader clock|lifetime, shader" $O/prio$L.txt
  python3 tools/r03/simd_timeline.py $O/prio$L.csv | tail -n +2 | grep "holding 3"
done
done
