#!/bin/bash
# One parametrised bench sweep (replaces round 2's 47 one-shot tools/r02_runs/r02_run*.sh, which are in the git
# history up to commit 99421be).  Runs ON THE GPU BOX:  gpurun -- 'bash tools/recipes/bench_sweep.sh <tag> <common args> -- <variant> [-- <variant> ...]'
#   tag            results go to gpurun_out/sweep_<tag>.txt
#   common args    passed to every bench.py run (e.g. --steps 30 --warmup 5 --no-cpu-baseline --sim-steps 0)
#   variant        extra bench.py arguments of one run; an item LIB=<path> selects another build of the library
# Every variant runs twice, interleaved, so that box-to-box and warm-up drift shows.
set -u
export TMPDIR=/tmp
TAG=$1; shift
COMMON=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do COMMON+=("$1"); shift; done
VARIANTS=(); cur=""
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then [ -n "$cur" ] && VARIANTS+=("$cur"); cur=" "; else cur="$cur $1"; fi; shift
done
[ -n "$cur" ] && VARIANTS+=("$cur")
OUT=gpurun_out/sweep_$TAG.txt; mkdir -p gpurun_out; : > $OUT
for rep in 1 2; do
  for v in "${VARIANTS[@]}"; do
    lib=""; args=""
    for w in $v; do case $w in LIB=*) lib=${w#LIB=};; *) args="$args $w";; esac; done
    # (another build of the library -- tools/recipes/build_variant.sh -- is selected through tools/with_lib.py: the product binding
    # knows one path only, and an SFL_LIB variable it ignores would silently benchmark the product library: ADVICE r04)
    if [ -n "$lib" ]; then run=(python tools/with_lib.py "$lib" bench.py); else run=(python bench.py); fi
    "${run[@]}" "${COMMON[@]}" $args > gpurun_out/sweep_$TAG.json 2> gpurun_out/sweep_$TAG.err || { echo "FAILED:$v" | tee -a $OUT; tail -3 gpurun_out/sweep_$TAG.err; continue; }
    python - "$v" gpurun_out/sweep_$TAG.json <<'PY' | tee -a $OUT
import json, sys
d = json.load(open(sys.argv[2]))
ms = d.get("ms_per_step", d.get("ms_per_solve"))
us = d.get("roofline", {}).get("avg_launch_us")
print("%-60s %.4f ms per solve%s" % (sys.argv[1].strip(), ms, "  %.2f us per launch" % us if us else ""))
PY
  done
done
