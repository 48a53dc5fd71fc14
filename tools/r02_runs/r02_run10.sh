#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run10
mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overlap or virtual or randomised or config3 or fused_vs_oracle" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^E " $O/pytest_gpu.log | head -5
python bench.py --no-cpu-baseline --sim-steps 0 --steps 20 --warmup 5 > $O/full.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/full.json')); print('full', d['ms_per_step'], d['roofline']['avg_launch_us'])"
python bench.py --no-cpu-baseline --sim-steps 0 --steps 20 --warmup 5 --dim-y 1024 > $O/slab.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/slab.json')); print('slab1024', d['ms_per_step'], d['roofline']['avg_launch_us'])"
python - <<'PY'
import importlib, numpy as np, time
sfl = importlib.import_module("esp32-fluid-simulation_amd")
cap = sfl.capi
om = np.float32(1.96)
dim, nranks, iters = 8192, 8, 80
for overlap in (0, 1):
    slabs = [sfl.Solver(dim, dim, 0, r, nranks) for r in range(nranks)]
    sfl.Solver.link_group(slabs)
    slabs[0].set_option(cap.OPT_SOR_OVERLAP, overlap)
    d = (np.random.default_rng(2).standard_normal((dim // nranks, dim)) * 0.1).astype(np.float32)
    for s in slabs:
        s.upload(cap.FIELD_DIVERGENCE, d)
    for _ in range(5):
        slabs[0].poisson_solve(1.0, iters, om)
    slabs[0].synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        slabs[0].poisson_solve(1.0, iters, om)
    slabs[0].synchronize()
    print("virtual 8 ranks on one GPU, overlap", overlap, (time.perf_counter() - t0) / 10 * 1e3, "ms per solve", slabs[3].last_solve_info())
    for s in slabs:
        s.close()
PY
