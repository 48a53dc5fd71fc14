#!/bin/bash
# r03: the NS = 16 kernel at 1 / 2 / 3 waves per SIMD, as shipped / without lane shifts / with row shifts
set -u
export TMPDIR=/tmp
O=gpurun_out/r03_occ
mkdir -p $O
for v in "" _noshift _rowshift; do
  for rows in 2808 5616 8192; do
    rpc=234; [ $rows = 8192 ] && rpc=0
    ./tools/sor_clock_probe_ns16$v 8192 $rows 30 $rpc $O/ns16${v}_$rows.csv > $O/ns16${v}_$rows.txt 2>&1
    echo "== variant '$v' rows $rows"; grep -E "waves traced|shader clock|lifetime, shader|SIMDs seen" $O/ns16${v}_$rows.txt
    python3 tools/r03/simd_timeline.py $O/ns16${v}_$rows.csv | tail -n +2
  done
done
