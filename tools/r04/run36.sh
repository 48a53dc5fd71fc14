#!/bin/bash
# r04: regression check of the non-chained path after the chain work: the library of commit 7612811 (build/variants/r24) against the
# product, interleaved on one box
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final3
mkdir -p $O; rm -f $O/regress_ab.txt
for rep in 1 2; do
for cfg in "--emulate-rank 1 --of 2" "--emulate-rank 1 --of 4" "--emulate-rank 3 --of 8" "--no-cpu-baseline --sim-steps 0" "--size 16384 --iters 200 --steps 4 --warmup 1 --sim-steps 0 --emulate-rank 3 --of 8"; do
for v in product r24; do
  if [ $v = product ]; then run="python bench.py"; else run="python tools/with_lib.py build/variants/$v/libsfl_hip.so bench.py"; fi
  $run --steps 20 --warmup 5 --sim-steps 0 $cfg > $O/run.json 2>$O/run.err || tail -3 $O/run.err
  python - "$v $cfg" $O/run.json <<'PY' | tee -a $O/regress_ab.txt
import json, sys
d = json.load(open(sys.argv[2]))
print("%-75s %.4f ms per solve" % (sys.argv[1][-75:], d.get("ms_per_solve") or d["ms_per_step"]))
PY
done; done; done
