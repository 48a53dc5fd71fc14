#!/bin/bash
# N RCCL ranks as N processes on ONE GPU (bench.py --share-device 0: every rank its own NCCL_HOSTID, RCCL's socket transport over
# loopback between them), every exchange schedule, parity of every rank's rows.  ON THE GPU BOX:
#   gpurun -- bash tools/recipes/shared_device_ranks.sh <tag> "<N size iters halo mode>" ["<N size iters halo mode>" ...]
#   mode: chain | in-time | by-event | in-line   (chain = the launcher as the driver runs it: library default, in line as the fallback,
#         the in-time experiment behind the headline)
# Result lines: gpurun_out/shared_<tag>.txt ; the last run's RCCL log (NCCL_DEBUG=INFO) in gpurun_out/shared_<tag>_rccl.log
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/shared_$TAG.txt; mkdir -p gpurun_out; : > $OUT
for spec in "$@"; do
  set -- $spec; n=$1 size=$2 iters=$3 halo=$4 mode=$5
  case $mode in chain) m="";; in-time) m="--arrival-in-time --halo-timeout-ms 15000";; by-event) m="--arrival-by-event";; in-line) m="--no-overlap";; esac
  h=""; [ "$halo" != 0 ] && h="--sor-halo $halo"
  t0=$(date +%s.%N)
  NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,NET timeout 900 python bench.py --gpus $n --share-device 0 --size $size --iters $iters --steps 6 --warmup 1 --sim-steps 1 --no-priming $m $h \
      > gpurun_out/shared_$TAG.json 2> gpurun_out/shared_${TAG}_rccl.log
  rc=$?
  t1=$(date +%s.%N)
  python - "$spec" $rc gpurun_out/shared_$TAG.json gpurun_out/shared_${TAG}_rccl.log $t0 $t1 <<'PY' | tee -a $OUT
import json, re, sys
spec, rc, path, log, t0, t1 = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], float(sys.argv[5]), float(sys.argv[6])
txt = open(log, errors="replace").read()
comms = len(re.findall(r"Init COMPLETE|ncclCommInitRank comm \S+ rank \d+ nranks \d+.*- Init COMPLETE", txt))
nr = sorted(set(re.findall(r"nranks (\d+)", txt)))
net = sorted(set(re.findall(r"via NET/(\w+)", txt)))
try:
    d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    par = (d.get("parity") or {})
    print(f"{spec:34s} rc {rc}  {t1 - t0:5.1f} s  ranks {d['n_gpus']} on {d['config'].get('physical_gpus', '?')} GPU  schedule '{d.get('exchange_mode')}' "
          f"fallback_from {[f.get('mode') for f in d.get('fallback_from', [])]}  solve vs reference CPU loop bit-exact {par.get('bit_exact')} ({par.get('cells')} cells)  "
          f"sim step on slabs vs whole domain bit-exact {(d.get('sim_step_parity') or {}).get('bit_exact')}  exchanges per solve {d['config']['halo_exchanges_per_solve']}  "
          f"halo {d['config'].get('halo_rows_per_superstep')}  measured exchange {d['config'].get('measured_exchange_latency_us')} us  ms per solve {d['ms_per_step']:.3f}  | RCCL log: nranks {nr}, transports {net}")
except Exception as e:
    print(f"{spec:34s} rc {rc}  {t1 - t0:5.1f} s  NO RESULT LINE ({e}); last log lines:", [l for l in txt.splitlines() if 'bench.py' in l][-3:])
PY
done
