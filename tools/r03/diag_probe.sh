#!/bin/bash
# r03: which ingredient costs what when the waves of a SIMD share it fairly (PRIO_LEVELS = 4)?
# (binaries: bash tools/r03/build_probes.sh diag)
# d1 no lane shifts, d2 no rhs ring (LDS), d3 no global loads, d4 all three, d5 no boundary path; prio4 = everything
set -u
export TMPDIR=/tmp
O=gpurun_out/r03_diag
mkdir -p $O
for v in prio4 d1 d2 d3 d4 d5 prio4; do
  for rows in 2808 8192; do
    rpc=234; [ $rows = 8192 ] && rpc=0
    ./tools/sor_clock_probe_ns16_$v 8192 $rows 30 $rpc $O/${v}_$rows.csv > $O/${v}_$rows.txt 2>&1
    echo "== $v rows $rows: $(grep -E 'waves traced' $O/${v}_$rows.txt | sed 's/.*launch by/launch by/') $(grep -E 'shader clock' $O/${v}_$rows.txt | sed 's/.*median/median GHz/;s/p90.*//') $(grep 'lifetime, shader' $O/${v}_$rows.txt)"
  done
done
