export TMPDIR=/tmp
for cfg in "512 20" "1024 20" "2048 40"; do set -- $cfg
  O=gpurun_out/gaps_$1; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 bench.py --size $1 --iters $2 --steps 1 --warmup 0 --no-cpu-baseline --no-fold-leg --sim-steps 200 > $O/line.json 2> $O/err.log
  python3 - $O $1 <<'PY'
import csv, glob, json, sys
o, size = sys.argv[1], sys.argv[2]
d = json.loads([l for l in open(o + "/line.json") if l.startswith("{")][-1])
rows = sorted(csv.DictReader(open(glob.glob(o + "/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
# the last 200 steps = sfl_step_n: take the last third of the dispatches as steady state
tail = rows[-len(rows) // 3:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
print("size %s: %.1f us per step (step_n), %.1f as separate calls; last third of the trace: %d kernels, busy %.1f %% of the span, mean kernel %.1f us, mean gap %.2f us" % (
    size, d["sim_step_us"], 1e6 / d["sim_steps_per_sec_as_separate_calls"], len(tail), 100.0 * busy / span, busy / len(tail) / 1e3, (span - busy) / len(tail) / 1e3))
PY
done
