#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final4; mkdir -p $O; rm -f $O/emulate_final.txt
python bench.py --no-cpu-baseline > $O/run0.json 2>$O/run0.err
for cfg in "--emulate-rank 1 --of 2" "--emulate-rank 1 --of 4" "--emulate-rank 3 --of 8" "--emulate-rank 0 --of 8" "--emulate-rank 7 --of 8" "--emulate-rank 3 --of 8 --chain -1" "--size 16384 --iters 200 --steps 4 --warmup 1 --sim-steps 2 --emulate-rank 3 --of 8"; do
  python bench.py --steps 30 --warmup 5 --sim-steps 12 $cfg > $O/run.json 2>$O/run.err || tail -3 $O/run.err
  python - "$cfg" $O/run.json <<'PY' | tee -a $O/emulate_final.txt
import json, sys
d = json.load(open(sys.argv[2]))
print("%-60s %.4f ms per solve  %8.1f us per sim step  supersteps %d (%d chained) exchanges %d fuse %d" % (sys.argv[1][-60:], d["ms_per_solve"], d["sim_step_us"] or 0, d["sor_launches_per_solve"], d["supersteps_in_chained_launches"], d["halo_exchanges_per_solve"], d["half_sweeps_fused_per_launch"]))
PY
done
python - $O/run0.json <<'PY' | tee -a $O/emulate_final.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("%-60s %.4f ms per solve  %8.1f us per sim step  (1 GPU, the same box)" % ("bench.py (8192^2 x 80, 1 GPU)", d["ms_per_step"], d["sim_step_us"]))
PY
python bench.py --size 16384 --iters 200 --steps 4 --warmup 1 --sim-steps 2 --no-cpu-baseline > $O/run5.json 2>$O/run5.err
python -c "import json;d=json.load(open('$O/run5.json'));print('%-60s %.4f ms per solve  %8.1f us per sim step  (1 GPU, the same box)' % ('16384^2 x 200, 1 GPU', d['ms_per_step'], d['sim_step_us']))" | tee -a $O/emulate_final.txt
