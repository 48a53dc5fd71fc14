#!/bin/bash
# r03: fuse depth x halo for one emulated rank of 8 (8192^2 x 80)
set -u
export TMPDIR=/tmp
O=gpurun_out/r03_emulate
mkdir -p $O
for cfg in "--fuse 10 --sor-halo 64" "--fuse 8 --sor-halo 64" "--fuse 12 --sor-halo 64" "--fuse 10 --sor-halo 40" "--fuse 10 --sor-halo 50" "--fuse 8 --sor-halo 48" "--fuse 14 --sor-halo 64" "--fuse 10 --sor-halo 64"; do
  python bench.py --steps 30 --warmup 5 --emulate-rank 3 --of 8 $cfg > $O/sweep_run.json 2>$O/sweep_run.err || tail -3 $O/sweep_run.err
  python - "$cfg" $O/sweep_run.json <<'PY' | tee -a $O/sweep_summary.txt
import json, sys
d = json.load(open(sys.argv[2]))
print("%-28s %.4f ms per solve  launches %d exchanges %d" % (sys.argv[1], d["ms_per_solve"], d["sor_launches_per_solve"], d["halo_exchanges_per_solve"]))
PY
done
