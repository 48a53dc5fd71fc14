#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final4; mkdir -p $O; rm -f $O/emulate_c4.txt $O/emulate_c5.txt
bash profiles/run_step_pmc.sh r04 > $O/profile_step.log 2>&1; tail -3 $O/profile_step.log
python bench.py --no-cpu-baseline > $O/run0.json 2>$O/run0.err
for cfg in "--emulate-rank 1 --of 2" "--emulate-rank 1 --of 4" "--emulate-rank 3 --of 8" "--emulate-rank 0 --of 8" "--emulate-rank 7 --of 8"; do
  python bench.py --steps 30 --warmup 5 --sim-steps 12 $cfg > $O/run.json 2>$O/run.err || tail -3 $O/run.err
  python - "$cfg" $O/run.json <<'PY' | tee -a $O/emulate_c4.txt
import json, sys
d = json.load(open(sys.argv[2]))
print("%-45s %.4f ms per solve  %8.1f us per sim step  supersteps %d exchanges %d fuse %d" % (sys.argv[1], d["ms_per_solve"], d["sim_step_us"] or 0, d["sor_launches_per_solve"], d["halo_exchanges_per_solve"], d["half_sweeps_fused_per_launch"]))
PY
done
python - $O/run0.json <<'PY' | tee -a $O/emulate_c4.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("%-45s %.4f ms per solve  %8.1f us per sim step  (1 GPU, the same box)" % ("bench.py (8192^2 x 80, 1 GPU)", d["ms_per_step"], d["sim_step_us"]))
PY
