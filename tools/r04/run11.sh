#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04/trace_stepn
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sim-steps 6 > $O/run.log 2>&1
F=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('seam','advect','divergence','gradient')):
        print('%-60s calls %4s  avg %9.1f us  min %9.1f  max %9.1f' % (n[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
tail -2 $O/run.log | cut -c1-300
