#!/bin/bash
# Kernel timeline of a bench.py run (rocprofv3 --kernel-trace only: start, duration, gap of every dispatch), ON THE GPU BOX:
#   gpurun -- 'bash tools/recipes/trace_timeline.sh <tag> [bench.py options]'
# e.g.  bash tools/recipes/trace_timeline.sh rank3_rccl --emulate-rank 3 --of 8 --via-rccl
# -> gpurun_out/trace_<tag>/timeline.txt (tools/timeline.py) and per-kernel durations (profiles/summarise_profile.py)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/trace_${1:-x}
rm -rf $O; mkdir -p $O/stats
shift
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o trace -- python3 bench.py --steps 6 --warmup 3 --no-priming --sim-steps 0 "$@" > $O/run.log 2>&1
F=$(find $O -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $F > $O/timeline.txt
python3 profiles/summarise_profile.py $O 12 > $O/summary.txt 2>&1
head -60 $O/summary.txt
