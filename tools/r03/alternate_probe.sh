#!/bin/bash
# r03: does reversing every tile's stream direction from one launch to the next (the next launch begins on
# the rows the previous one touched last) pay through the Infinity Cache?  NS = 16, nt / plain stores
set -u
export TMPDIR=/tmp
O=gpurun_out/r03_alt
mkdir -p $O
for rep in 1 2 3; do
for v in "" _nont; do
for alt in 0 1; do
for grid in "8192 8192" "8192 4096" "16384 16384"; do
  set -- $grid
  a=$(PROBE_ALTERNATE=$alt ./tools/sor_clock_probe_ns16$v $1 $2 20 0 | grep -E "waves traced|shader clock" | sed 's/.*rows_per_chunk [0-9]*: //;s/shader clock per wave.*median/median GHz/;s/p90.*//' | tr '\n' ' ')
  echo "rep $rep stores${v:-_nt} alternate $alt grid $1x$2: $a"
done
done
done
done
