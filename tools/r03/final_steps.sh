#!/bin/bash
# r03: one box -- 1 GPU and the emulated ranks, solve and whole sim step, final library
set -u
export TMPDIR=/tmp
O=gpurun_out/r03_final
mkdir -p $O
: > $O/emulate_c4.txt
python bench.py --no-cpu-baseline > $O/bench_1gpu.json 2> $O/bench_1gpu.err
python - $O/bench_1gpu.json <<'PY' | tee -a $O/emulate_c4.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("%-45s %.4f ms per solve  %8.1f us per sim step" % ("bench.py (8192^2 x 80, 1 GPU)", d["ms_per_step"], d["sim_step_us"]))
PY
for cfg in "--emulate-rank 1 --of 2" "--emulate-rank 1 --of 4" "--emulate-rank 3 --of 8" "--emulate-rank 0 --of 8" "--dim-y 1024 --no-cpu-baseline --sim-steps 0"; do
  python bench.py --steps 30 --warmup 5 --sim-steps 30 $cfg > $O/run.json 2>$O/run.err || tail -3 $O/run.err
  python - "$cfg" $O/run.json <<'PY' | tee -a $O/emulate_c4.txt
import json, sys
d = json.load(open(sys.argv[2]))
if "ms_per_solve" in d:
    print("%-45s %.4f ms per solve  %8.1f us per sim step  launches %d exchanges %d fuse %d" % (sys.argv[1], d["ms_per_solve"], d["sim_step_us"] or 0, d["sor_launches_per_solve"], d["halo_exchanges_per_solve"], d["half_sweeps_fused_per_launch"]))
else:
    print("%-45s %.4f ms per solve  launches %d fuse %d" % (sys.argv[1], d["ms_per_step"], d["config"]["sor_launches_per_solve"], d["config"]["half_sweeps_fused_per_launch"]))
PY
done
