#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final3
mkdir -p $O
rm -f $O/emulate_ab.txt
for rep in 1 2; do
for cfg in "--emulate-rank 3 --of 8" "--emulate-rank 3 --of 8 --chain 0" "--emulate-rank 0 --of 8" "--emulate-rank 0 --of 8 --chain 0" "--emulate-rank 7 --of 8" "--emulate-rank 7 --of 8 --chain 0"; do
  python bench.py --steps 30 --warmup 5 --sim-steps 8 $cfg > $O/run.json 2>$O/run.err || tail -3 $O/run.err
  python - "$cfg" $O/run.json <<'PY' | tee -a $O/emulate_ab.txt
import json, sys
d = json.load(open(sys.argv[2]))
print("%-60s %.4f ms per solve  %8.1f us per sim step  supersteps %d (%d chained) exchanges %d" % (sys.argv[1][-60:], d["ms_per_solve"], d["sim_step_us"] or 0, d["sor_launches_per_solve"], d["supersteps_in_chained_launches"], d["halo_exchanges_per_solve"]))
PY
done; done
