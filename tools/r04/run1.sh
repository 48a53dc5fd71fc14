#!/bin/bash
# r04 first GPU call: the whole GPU suite, the default bench line, sc1 cache-policy probes of the fused SOR kernel,
# and the wire-delay curve of the emulated rank (tools/r04/wire_curve.sh).
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -3 $O/bench_default.err
for v in product p_ld_sc1 p_st_sc1 p_both_sc1 product; do
  for cfg in "" "--dim-y 1024" "--size 16384 --iters 200 --steps 3 --warmup 1"; do
    if [ $v = product ]; then run="python bench.py"; else run="python tools/with_lib.py build/variants/$v/libsfl_hip.so bench.py"; fi
    $run --steps 20 --warmup 3 --sim-steps 0 --no-cpu-baseline $cfg > $O/sc1_run.json 2> $O/sc1_run.err || tail -3 $O/sc1_run.err
    python - "$v" "$cfg" $O/sc1_run.json <<'PY' | tee -a $O/sc1_probe.txt
import json, sys
d = json.load(open(sys.argv[3]))
print("%-12s %-46s %.4f ms per solve  %.1f us per launch" % (sys.argv[1], sys.argv[2] or "8192^2 x 80", d["ms_per_step"], d["roofline"]["avg_launch_us"]))
PY
  done
done
bash tools/r04/wire_curve.sh
# per-wave trace of the thin share (CSV: tile, start / end on both clocks, hardware slot, XCD, path) for the balance study
./tools/sor_clock_probe_ns10 8192 1024 40 0 $O/probe_ns10_slab1024.csv > $O/probe_ns10_slab1024.txt 2>&1
cat $O/probe_ns10_slab1024.txt
