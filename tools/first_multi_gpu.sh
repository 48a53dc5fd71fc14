#!/bin/bash
# First thing to run on a node with more than one MI355X: the RCCL path with N > 1 has only ever been
# exercised on one-GPU boxes (virtual ranks, an emulated rank, a one-rank communicator).
#   1. the multi-GPU tests (BASELINE configs 4 and 5 exactly, 2- and 4-rank points, with and without overlap);
#   2. bench.py at 1 / 2 / 4 / 8 GPUs as the driver runs it: the headline from the library's own schedule (behind events between
#      processes; fresh ranks with every exchange in line should that fail) + the in-time schedule as `in_time_experiment`; then
#      every schedule forced: exchanges in time, early exchanges behind events, in line --
#      the ranks are started as fresh child processes by bench.py's own launcher (its parent never touches a GPU);
#   3. RCCL's own report of the ranks (NCCL_DEBUG=INFO of the 2-GPU run: "comm ... nranks 2" per rank);
#   4. a rocprofv3 kernel trace of rank 0 at the largest N (the launcher's children are traced through torch.distributed.run's
#      per-rank command; rocprofv3 must see `python3` itself, not a shell).
# Results under gpurun_out/multi_gpu/ ; copy what should be judged into profiles/.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
O=gpurun_out/multi_gpu
mkdir -p $O
NG=$(python3 -c "import importlib; print(importlib.import_module('esp32-fluid-simulation_amd').device_count())")
echo "visible GPUs: $NG" | tee $O/summary.txt
if [ "$NG" -lt 2 ]; then echo "needs at least 2 GPUs" | tee -a $O/summary.txt; exit 1; fi
# (0. the same N processes and N-rank communicator on GPU 0 alone, over RCCL's socket transport -- what the one-GPU boxes could show,
#  profiles/r05_rccl_ranks_on_one_device.txt: if THIS fails here the trouble is not the wire)
timeout 600 python3 bench.py --gpus 4 --share-device 0 --size 2048 --iters 40 --steps 6 --warmup 1 --sim-steps 1 --no-priming > $O/bench_shared_device_n4.json 2> $O/bench_shared_device_n4.err \
    && echo "4 ranks on device 0 over sockets: ok" | tee -a $O/summary.txt || echo "4 ranks on device 0 over sockets: FAILED (see $O/bench_shared_device_n4.err)" | tee -a $O/summary.txt
( time SFL_SLOW_MULTI_GPU=1 python3 -m pytest tests/test_multi_gpu.py -m gpu -x -q ) > $O/pytest_multi_gpu.log 2>&1
tail -3 $O/pytest_multi_gpu.log | tee -a $O/summary.txt
for n in 1 2 4 8; do
  [ $n -le $NG ] || continue
  # "" = the launcher as the driver runs it (the line's exchange_mode / fallback_from / in_time_experiment say what ran).  The others
  # force one schedule (no fallback, no experiment).
  for mode in "" "--arrival-in-time" "--arrival-by-event" "--no-overlap"; do
    tag=n${n}$(echo "$mode" | tr -d ' -')
    timeout 900 python3 bench.py --gpus $n --steps 10 --warmup 3 $mode > $O/bench_$tag.json 2> $O/bench_$tag.err || tail -3 $O/bench_$tag.err
    python3 -c "
import json; d = json.load(open('$O/bench_$tag.json'))
print('%-34s %d GPU(s): %.3e cell-iters/s  %.4f ms per solve  %s sim steps/s  parity %s  exchanges per solve %s  halo %s rows  measured exchange %s us  schedule %s  fallback_from %s' % ('${mode:-as the driver runs it}', d['n_gpus'], d['value'], d['ms_per_step'], d['sim_steps_per_sec'], (d.get('parity') or {}).get('bit_exact'), d['config']['halo_exchanges_per_solve'], d['config'].get('halo_rows_per_superstep'), d['config'].get('measured_exchange_latency_us'), d.get('exchange_mode', d['config'].get('exchange_schedule')), [f.get('mode') for f in d.get('fallback_from', [])]) + ('  in-time experiment: %s' % json.dumps(d['in_time_experiment']) if 'in_time_experiment' in d else ''))" | tee -a $O/summary.txt
  done
done
NCCL_DEBUG=INFO timeout 600 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --sim-steps 0 > /dev/null 2> $O/rccl_info_n2.log
grep -c "nranks 2" $O/rccl_info_n2.log | sed 's/^/RCCL communicators reporting nranks 2: /' | tee -a $O/summary.txt
N=$NG; [ $N -gt 8 ] && N=8
# (SFL_BENCH_WORKER=1: every process torch.distributed.run starts IS a rank -- without it bench.py would be its rank's supervisor and
# do the GPU work in a child the profiler does not follow)
SFL_BENCH_WORKER=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29531 --no-python \
    rocprofv3 --kernel-trace --output-format csv -d $O/trace_n$N -o t -- python3 bench.py --gpus $N --steps 6 --warmup 3 --no-priming --no-cpu-baseline --sim-steps 0 \
    > $O/trace_n$N.log 2>&1 || tail -3 $O/trace_n$N.log
for f in $(find $O/trace_n$N -name "*kernel_trace.csv" | head -2); do python3 tools/timeline.py $f > ${f%.csv}_timeline.txt 2>/dev/null; done
cat $O/summary.txt
