#!/bin/bash
# r04: the automatic tile height memoised per thread (host time per launch), against the library without it; interleaved, one box
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final4; mkdir -p $O; rm -f $O/memo_ab.txt
for rep in 1 2 3; do
for cfg in "--size 2048 --iters 40 --no-cpu-baseline" "--size 1024 --iters 20 --no-cpu-baseline" "--emulate-rank 3 --of 8" "--no-cpu-baseline"; do
for v in product nomemo; do
  if [ $v = product ]; then run="python bench.py"; else run="python tools/with_lib.py build/variants/$v/libsfl_hip.so bench.py"; fi
  $run --sim-steps 12 $cfg > $O/run.json 2>$O/run.err || tail -3 $O/run.err
  python - "$v $cfg" $O/run.json <<'PY' | tee -a $O/memo_ab.txt
import json, sys
d = json.load(open(sys.argv[2]))
ms = d.get("ms_per_solve") or d["ms_per_step"]
print("%-60s %.4f ms per solve  %8.1f us per sim step" % (sys.argv[1][-60:], ms, d["sim_step_us"] or 0))
PY
done; done; done
