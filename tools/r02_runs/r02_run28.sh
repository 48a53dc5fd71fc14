#!/bin/bash
# every second launch of a solve walks its tiles last to first (what the previous launch touched last is
# still in the memory-side cache): A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run28
mkdir -p $O
B="python bench.py --no-cpu-baseline --sim-steps 0 --steps 30 --warmup 5"
run() { n=$1; shift; $B "$@" > $O/$n.json 2>> $O/bench.err; python - <<PY
import json
try:
    d=json.load(open("$O/$n.json"))
    print("%-22s %8.4f ms  fuse %2d launches %2d  %7.2f us/launch  parity %s" % ("$n", d["ms_per_step"], d["config"]["half_sweeps_fused_per_launch"], d["config"]["sor_launches_per_solve"], d["roofline"]["avg_launch_us"], d.get("parity")))
except Exception as e:
    print("$n", "FAILED", e)
PY
}
for rep in 1 2; do
for cfg in "full:" "s4096:--dim-y 4096" "s2048:--dim-y 2048" "s1024:--dim-y 1024" "c2:--size 2048 --iters 40" "c5s:--size 16384 --dim-y 2048 --iters 200 --steps 8"; do
  n=${cfg%%:*}; a=${cfg#*:}
  for f in 1 5; do SFL_SOR_FLIP=$f run order${f}_${n}_$rep $a; done
done; done
SFL_SOR_FLIP=5 python bench.py --steps 5 --warmup 2 --sim-steps 0 > $O/parity_full.json 2>> $O/bench.err
python -c "
import json; d=json.load(open('$O/parity_full.json')); print('reverse order parity:', d['parity'], d['ms_per_step'])"
for f in 1 5; do
  SFL_SOR_FLIP=$f rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_flip$f -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --sim-steps 0 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
rows=[r for f in glob.glob("$O/pmc_flip$f/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
v=[float(r["Counter_Value"]) for r in rows if "sor_fused" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("order $f full domain FETCH_SIZE avg per launch (raw):", sum(v)/max(len(v),1), "n", len(v))
PY
done
