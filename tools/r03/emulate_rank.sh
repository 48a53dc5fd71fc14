#!/bin/bash
# r03: the per-GPU critical path of BASELINE config 4 (8192^2 x 80 on 8 GPUs), one rank's program alone
set -u
export TMPDIR=/tmp
O=gpurun_out/r03_emulate
mkdir -p $O
TAG=${1:-base}
for rep in 1 2; do
for cfg in "--emulate-rank 3 --of 8" "--emulate-rank 0 --of 8" "--emulate-rank 3 --of 8 --no-overlap" "--dim-y 1024 --no-cpu-baseline --sim-steps 0" "--emulate-rank 1 --of 4" "--emulate-rank 1 --of 2"; do
  python bench.py --steps 30 --warmup 5 $cfg > $O/${TAG}_run.json 2>$O/${TAG}_run.err || tail -3 $O/${TAG}_run.err
  python - "$cfg" $O/${TAG}_run.json <<'PY' | tee -a $O/${TAG}_summary.txt
import json, sys
d = json.load(open(sys.argv[2]))
if "ms_per_solve" in d:
    print("%-45s %.4f ms per solve (events %.4f, unprimed %.4f)  launches %d exchanges %d fuse %d" % (sys.argv[1], d["ms_per_solve"], d["ms_per_solve_hip_events"], d["ms_per_solve_unprimed"] or 0, d["sor_launches_per_solve"], d["halo_exchanges_per_solve"], d["half_sweeps_fused_per_launch"]))
else:
    print("%-45s %.4f ms per solve  launches %d fuse %d" % (sys.argv[1], d["ms_per_step"], d["config"]["sor_launches_per_solve"], d["config"]["half_sweeps_fused_per_launch"]))
PY
done
done
