#!/bin/bash
# seam kernel variants: per-kernel durations from a kernel trace
set -u
export TMPDIR=/tmp
for v in product seam_late6 seam_late8 seam_late4; do
  O=$PWD/gpurun_out/r04/trace_seam_$v
  rm -rf $O; mkdir -p $O
  if [ $v = product ]; then prog="bench.py"; else prog="tools/with_lib.py build/variants/$v/libsfl_hip.so bench.py"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $prog --steps 2 --warmup 1 --no-cpu-baseline --sim-steps 6 > $O/run.log 2>&1
  F=$(find $O -name "*kernel_stats.csv" | head -1)
  python3 - "$F" $v <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('seam','advect_divergence','advect_vec3uq32_tiled_kernel<false, true')):
        print('%-14s %-50s calls %4s  avg %9.1f us' % (sys.argv[2], n.split('::')[-1][:50], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
