set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "quiescent or folded or sor_fused_vs_oracle or baseline_config3 or signed_zero" > gpurun_out/r06_t1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_t1.log
tail -5 gpurun_out/r06_t1.log
bash tools/recipes/bench_sweep.sh r06_fold_c3 --steps 30 --warmup 5 --no-cpu-baseline --sim-steps 0 --no-fold-leg -- -- --sor-fold
bash tools/recipes/bench_sweep.sh r06_fold_share --size 8192 --dim-y 1024 --steps 60 --warmup 10 --no-cpu-baseline --sim-steps 0 --no-fold-leg -- -- --sor-fold
bash tools/recipes/bench_sweep.sh r06_fold_c2 --size 2048 --iters 40 --steps 60 --warmup 10 --no-cpu-baseline --sim-steps 0 --no-fold-leg -- -- --sor-fold
bash tools/recipes/bench_sweep.sh r06_fold_c5 --size 16384 --iters 200 --steps 4 --warmup 1 --no-cpu-baseline --sim-steps 0 --no-fold-leg -- -- --sor-fold
bash tools/recipes/bench_sweep.sh r06_fold_rank3of8 --emulate-rank 3 --of 8 --steps 30 --warmup 5 -- -- --sor-fold
bash tools/recipes/bench_sweep.sh r06_fold_c5rank3 --emulate-rank 3 --of 8 --size 16384 --iters 200 --steps 6 --warmup 2 -- -- --sor-fold
python bench.py > gpurun_out/r06_bench_default_a.json 2> gpurun_out/r06_bench_default_a.err; tail -c 1500 gpurun_out/r06_bench_default_a.json
