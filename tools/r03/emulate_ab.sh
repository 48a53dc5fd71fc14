#!/bin/bash
# r03: cross-stream dependencies by HIP events (SFL_STREAM_EVENTS=1) vs stream memory operations (default)
set -u
export TMPDIR=/tmp
O=gpurun_out/r03_emulate
mkdir -p $O
for rep in 1 2 3; do
for ev in 1 0; do
for cfg in "--emulate-rank 3 --of 8" "--emulate-rank 1 --of 4"; do
  SFL_STREAM_EVENTS=$ev python bench.py --steps 30 --warmup 5 --sim-steps 10 $cfg > $O/ab_run.json 2>$O/ab_run.err || tail -3 $O/ab_run.err
  python - "$cfg" $ev $O/ab_run.json <<'PY' | tee -a $O/ab_summary.txt
import json, sys
d = json.load(open(sys.argv[3]))
print("%-26s %s  %.4f ms per solve  %.1f us per sim step" % (sys.argv[1], "HIP events        " if sys.argv[2] == "1" else "stream memory ops ", d["ms_per_solve"], d["sim_step_us"] or 0))
PY
done
done
done
