#!/bin/bash
# r04: final evidence of the tree with the chained launch in it (default off): whole GPU suite, smoke, bench lines, profiles + PMC passes
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final2
mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -1 | tee $O/smoke.log
bash profiles/run_profile.sh r04_final > $O/profile_8192.log 2>&1
bash profiles/run_profile.sh r04_slab1024 --dim-y 1024 > $O/profile_slab.log 2>&1
bash profiles/run_step_pmc.sh r04 > $O/profile_step.log 2>&1
tail -2 $O/profile_step.log
for rep in 1 2 3; do
for cfg in "--emulate-rank 3 --of 8 --chain 0" "--emulate-rank 3 --of 8 --chain 1" "--emulate-rank 0 --of 8 --chain 0" "--emulate-rank 0 --of 8 --chain 1" "--dim-y 1024 --no-cpu-baseline --sim-steps 0 --chain 0" "--dim-y 1024 --no-cpu-baseline --sim-steps 0 --chain 1"; do
  python bench.py --steps 30 --warmup 5 --sim-steps 8 $cfg > $O/run.json 2>$O/run.err || tail -3 $O/run.err
  python - "$cfg" $O/run.json <<'PY' | tee -a $O/chain_final.txt
import json, sys
d = json.load(open(sys.argv[2]))
if "ms_per_solve" in d:
    print("%-60s %.4f ms per solve  %8.1f us per sim step" % (sys.argv[1], d["ms_per_solve"], d["sim_step_us"] or 0))
else:
    print("%-60s %.4f ms per solve" % (sys.argv[1], d["ms_per_step"]))
PY
done; done
