#!/bin/bash
# Soak / fuzz / stress of the library as built, ON THE GPU BOX:  gpurun -- 'bash tools/recipes/soak.sh <tag> <seconds per leg> [legs]'
#   legs (default: all):  fuzz      tests/fuzz_vs_oracle.py: random shapes / options / dx / omega / dt, a solve and two steps each
#                         overlap   tools/soak_overlap.py: the overlapped slab executor on random virtual-rank configurations
#                         rccl      tools/soak_transport.py ... rccl: one rank's program with RCCL-to-self as the transport
#                         copy      tools/soak_transport.py ... copy: ... with self-copies
#                         unaligned tools/unaligned_stress.py: pitches that are not whole cache lines, both arrival modes
#                         ranks     tests/rccl_rank_worker.py --spawn N soak: 2 / 3 / 4 RANK PROCESSES, one real RCCL communicator on
#                                   device 0 (NCCL_HOSTID per rank, socket transport), random slab groups against the oracle
# Result lines: gpurun_out/soak_<tag>.txt (every leg ends with its count of mismatches; the script's status is non-zero on any)
set -u
export TMPDIR=/tmp
TAG=$1 SECS=$2; shift 2
LEGS=${*:-"fuzz overlap rccl copy unaligned ranks"}
OUT=gpurun_out/soak_$TAG.txt; mkdir -p gpurun_out; : > $OUT
rc=0
run() { echo "## $*" >> $OUT; "$@" 2>/dev/null | grep -v "^$" | tail -6 >> $OUT || rc=1; }
for leg in $LEGS; do case $leg in
  fuzz)      run python tests/fuzz_vs_oracle.py 61 $SECS;;
  overlap)   run python tools/soak_overlap.py 62 $SECS;;
  rccl)      run python tools/soak_transport.py 63 $SECS rccl;;
  copy)      run python tools/soak_transport.py 64 $SECS copy;;
  unaligned) run python tools/unaligned_stress.py $SECS 1; run python tools/unaligned_stress.py $((SECS / 2)) 0;;
  ranks)     for w in 2 3 4; do run python tests/rccl_rank_worker.py --spawn $w soak $((70 + w)) $((SECS / 2)); done
             run python tests/rccl_rank_worker.py --spawn 3 soak 79 $((SECS / 2)) big;;
esac; done
cat $OUT
exit $rc
