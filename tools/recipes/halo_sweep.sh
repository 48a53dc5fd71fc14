#!/bin/bash
# Is the AUTOMATIC halo depth of a slab solve (chosen from the measured exchange, sor_executor.cpp effective_halo) as good as the
# best fixed one?  One emulated rank, halo depths 0 (automatic) and a list, over wire delays and transports.  ON THE GPU BOX:
#   gpurun -- 'bash tools/recipes/halo_sweep.sh <tag> <size> <iters> <of> <rank> "<halos>" "<wire_us list>" "<transports>" [reps]'
# e.g.  bash tools/recipes/halo_sweep.sh c4 8192 80 8 3 "0 32 48 64 80 96 128 160" "0 25 100" "copy rccl" 2
# Table: gpurun_out/halo_<tag>.txt
set -u
export TMPDIR=/tmp
TAG=$1 SIZE=$2 ITERS=$3 OF=$4 RANK=$5 HALOS=$6 WIRES=$7 TRANSPORTS=$8 REPS=${9:-2}
OUT=gpurun_out/halo_$TAG.txt; mkdir -p gpurun_out; : > $OUT
echo "# bench.py --emulate-rank $RANK --of $OF --size $SIZE --iters $ITERS --sor-halo H --wire-us D [--via-rccl] ; $(date -u +%FT%TZ)" | tee -a $OUT
for t in $TRANSPORTS; do for w in $WIRES; do for rep in $(seq $REPS); do for h in $HALOS; do
  flag=""; [ "$t" = rccl ] && flag="--via-rccl"
  python bench.py --emulate-rank $RANK --of $OF --size $SIZE --iters $ITERS --steps 30 --warmup 5 --sim-steps 0 --wire-us $w --sor-halo $h $flag \
      > gpurun_out/halo_$TAG.json 2>> gpurun_out/halo_$TAG.err || { echo "FAILED $t wire $w halo $h" | tee -a $OUT; continue; }
  python - gpurun_out/halo_$TAG.json $rep $h <<'PY' | tee -a $OUT
import json, sys
d = json.load(open(sys.argv[1]))
print("%-12s wire %3d us  rep %s  halo asked %3s used %3d  %.4f ms per solve  %d launches %d exchanges  measured exchange latency %d us" % (
    d["transport"], d["emulated_wire_us"], sys.argv[2], sys.argv[3] if sys.argv[3] != "0" else "auto", d["halo_rows_per_superstep"],
    d["ms_per_solve"], d["sor_launches_per_solve"], d["halo_exchanges_per_solve"], d["measured_exchange_latency_us"]))
PY
done; done; done; done
