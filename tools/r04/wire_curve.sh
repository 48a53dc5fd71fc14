#!/bin/bash
# ms per solve of ONE emulated rank against the latency D of its (emulated) halo messages,
# overlap on and off: up to which xGMI latency does the schedule hide the wire?
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
out=$O/wire_curve.txt
echo "# bench.py --emulate-rank R --of 8 --wire-us D [--no-overlap]   (ms per solve | us per sim step)" >> $out
for cfg in "C4 3 --size 8192 --iters 80" "C4 0 --size 8192 --iters 80" "C5 3 --size 16384 --iters 200 --steps 4 --warmup 2"; do
  set -- $cfg; name=$1; rank=$2; shift 2
  for ov in "" "--no-overlap"; do
    line="$name rank $rank of 8 ${ov:-overlap    }"
    for D in 0 10 25 50 100; do
      python bench.py --emulate-rank $rank --of 8 --wire-us $D $ov --steps 20 --warmup 3 --sim-steps 4 "$@" > $O/wire_run.json 2> $O/wire_run.err || tail -3 $O/wire_run.err
      line="$line  D=$D: $(python -c "import json;d=json.load(open('$O/wire_run.json'));print('%.4f | %.0f' % (d['ms_per_solve'], d['sim_step_us'] or 0))")"
    done
    echo "$line" | tee -a $out
  done
done
