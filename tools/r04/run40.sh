#!/bin/bash
# r04: the slab sim step before / after (reach measured by the dye's kernel, early interior advection, exchange order): the library of
# the commit before (build/variants/prev) against the product, interleaved on one box; then a kernel timeline of one step of each
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final3; mkdir -p $O; rm -f $O/step_ab.txt
for rep in 1 2; do
for cfg in "--emulate-rank 3 --of 8" "--emulate-rank 1 --of 4"; do
for v in product prev; do
  if [ $v = product ]; then run="python bench.py"; else run="python tools/with_lib.py build/variants/$v/libsfl_hip.so bench.py"; fi
  $run --steps 10 --warmup 3 --sim-steps 12 $cfg > $O/run.json 2>$O/run.err || tail -3 $O/run.err
  python - "$v $cfg" $O/run.json <<'PY' | tee -a $O/step_ab.txt
import json, sys
d = json.load(open(sys.argv[2]))
print("%-40s %.4f ms per solve  %7.1f us per sim step" % (sys.argv[1], d["ms_per_solve"], d["sim_step_us"] or 0))
PY
done; done; done
bash tools/r04/trace_emulate.sh step_new --emulate-rank 3 --of 8 --sim-steps 6 > /dev/null
python3 tools/r04/step_timeline.py $(find gpurun_out/r04/trace_step_new -name "*kernel_trace.csv" | head -1) > $O/step_timeline_new.txt
