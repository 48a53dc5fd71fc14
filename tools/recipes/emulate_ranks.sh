#!/bin/bash
# One rank's program of an N-GPU solve on ONE GPU (bench.py --emulate-rank), over transports / ranks / wire delays,
# interleaved so that box drift shows.  Runs ON THE GPU BOX:
#   gpurun -- 'bash tools/recipes/emulate_ranks.sh <tag> <size> <iters> <of> "<ranks>" "<transports>" "<wire_us list>" [reps] [extra bench args]'
#   transports: copy (sfl_comm_emulate: self-copies)  rccl (sfl_comm_emulate_rccl: ncclSend/ncclRecv to self)
# e.g.  bash tools/recipes/emulate_ranks.sh c4 8192 80 8 "0 3 7" "copy rccl" "0" 3
# Table: gpurun_out/emulate_<tag>.txt (ms per solve by wall clock and HIP events, us per sim step, exchanges, schedule).
set -u
export TMPDIR=/tmp
TAG=$1 SIZE=$2 ITERS=$3 OF=$4 RANKS=$5 TRANSPORTS=$6 WIRES=$7 REPS=${8:-2}; shift 8 2>/dev/null || shift $#
EXTRA="$*"
OUT=gpurun_out/emulate_$TAG.txt; mkdir -p gpurun_out; : > $OUT
echo "# bench.py --emulate-rank R --of $OF --size $SIZE --iters $ITERS $EXTRA ; $(date -u +%FT%TZ)" | tee -a $OUT
# the one-GPU solve of the same grid on the same box: what the speed-ups are quoted against
python bench.py --size $SIZE --iters $ITERS --steps 10 --warmup 3 --no-cpu-baseline --sim-steps 6 > gpurun_out/emulate_$TAG.one.json 2> gpurun_out/emulate_$TAG.err \
  && python - gpurun_out/emulate_$TAG.one.json <<'PY' | tee -a $OUT
import json, sys
d = json.load(open(sys.argv[1]))
print("one GPU, whole grid: %.4f ms per solve, %.1f us per sim step" % (d["ms_per_step"], d["sim_step_us"]))
PY
for rep in $(seq $REPS); do
  for r in $RANKS; do for w in $WIRES; do for t in $TRANSPORTS; do
    flag=""; [ "$t" = rccl ] && flag="--via-rccl"
    python bench.py --emulate-rank $r --of $OF --size $SIZE --iters $ITERS --steps 30 --warmup 5 --wire-us $w $flag $EXTRA \
        > gpurun_out/emulate_$TAG.json 2>> gpurun_out/emulate_$TAG.err || { echo "FAILED rank $r $t wire $w" | tee -a $OUT; tail -2 gpurun_out/emulate_$TAG.err; continue; }
    python - gpurun_out/emulate_$TAG.json $rep <<'PY' | tee -a $OUT
import json, sys
d = json.load(open(sys.argv[1]))
print("rep %s rank %d of %d  %-12s wire %3d us  %.4f ms per solve (events %.4f)  %7.1f us per sim step  %d launches %d exchanges  %s" % (
    sys.argv[2], d["emulated_rank"], d["of"], d["transport"], d["emulated_wire_us"], d["ms_per_solve"], d["ms_per_solve_hip_events"],
    d["sim_step_us"] or float("nan"), d["sor_launches_per_solve"], d["halo_exchanges_per_solve"], d["exchange_schedule"]))
PY
  done; done; done
done
