#!/bin/bash
# r04: kernel timeline of one emulated rank's solve (rocprofv3 --kernel-trace only); usage: trace_emulate.sh <tag> [bench.py options]
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04/trace_${1:-x}
rm -rf $O; mkdir -p $O
shift
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 bench.py --steps 6 --warmup 3 --no-priming --sim-steps 0 "$@" > $O/run.log 2>&1
F=$(find $O -name "*kernel_trace.csv" | head -1)
python3 tools/r03/timeline.py $F > $O/timeline.txt
head -45 $O/timeline.txt
