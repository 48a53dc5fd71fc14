#!/bin/bash
# r04: the numbers DESIGN.md quotes, from one box: 1 GPU, the emulated ranks of configs 4 and 5, profiles + PMC passes of the solve's
# kernel (8192^2 and the thin share) and of the step's kernels
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final3
rm -f $O/emulate_c4.txt $O/emulate_c5.txt
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for cfg in "--emulate-rank 1 --of 2" "--emulate-rank 1 --of 4" "--emulate-rank 3 --of 8" "--emulate-rank 3 --of 8 --chain 0" "--emulate-rank 0 --of 8" "--emulate-rank 0 --of 8 --chain 0" "--dim-y 1024 --no-cpu-baseline --sim-steps 0"; do
  python bench.py --steps 30 --warmup 5 --sim-steps 10 $cfg > $O/run.json 2>$O/run.err || tail -3 $O/run.err
  python - "$cfg" $O/run.json <<'PY' | tee -a $O/emulate_c4.txt
import json, sys
d = json.load(open(sys.argv[2]))
if "ms_per_solve" in d:
    print("%-45s %.4f ms per solve  %8.1f us per sim step  supersteps %d (%d chained) exchanges %d fuse %d" % (sys.argv[1], d["ms_per_solve"], d["sim_step_us"] or 0, d["sor_launches_per_solve"], d["supersteps_in_chained_launches"], d["halo_exchanges_per_solve"], d["half_sweeps_fused_per_launch"]))
else:
    print("%-45s %.4f ms per solve  launches %d fuse %d" % (sys.argv[1], d["ms_per_step"], d["config"]["sor_launches_per_solve"], d["config"]["half_sweeps_fused_per_launch"]))
PY
done
python - $O/bench_default.json <<'PY' | tee -a $O/emulate_c4.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("%-45s %.4f ms per solve  %8.1f us per sim step  (1 GPU, the same box)" % ("bench.py (8192^2 x 80, 1 GPU)", d["ms_per_step"], d["sim_step_us"]))
PY
for cfg in "--emulate-rank 3 --of 8" "--dim-y 2048 --no-cpu-baseline --sim-steps 0" "--no-cpu-baseline --sim-steps 0"; do
  python bench.py --size 16384 --iters 200 --steps 4 --warmup 1 --sim-steps 0 $cfg > $O/run.json 2>$O/run.err || tail -3 $O/run.err
  python - "$cfg" $O/run.json <<'PY' | tee -a $O/emulate_c5.txt
import json, sys
d = json.load(open(sys.argv[2]))
if "ms_per_solve" in d:
    print("16384 x 200  %-45s %.3f ms per solve  launches %d exchanges %d fuse %d" % (sys.argv[1], d["ms_per_solve"], d["sor_launches_per_solve"], d["halo_exchanges_per_solve"], d["half_sweeps_fused_per_launch"]))
else:
    print("16384 x 200  %-45s %.3f ms per solve  launches %d fuse %d" % (sys.argv[1], d["ms_per_step"], d["config"]["sor_launches_per_solve"], d["config"]["half_sweeps_fused_per_launch"]))
PY
done
python bench.py --size 2048 --iters 40 > $O/bench_c2_2048.json 2> $O/bench_c2.err
python bench.py --size 61 --dim-y 81 --iters 20 > $O/bench_c1_61x81.json 2> $O/bench_c1.err
bash profiles/run_profile.sh r04_final > $O/profile_8192.log 2>&1
bash profiles/run_profile.sh r04_slab1024 --dim-y 1024 > $O/profile_slab.log 2>&1
tail -3 $O/profile_slab.log
bash profiles/run_step_pmc.sh r04 > $O/profile_step.log 2>&1
tail -5 $O/profile_step.log
