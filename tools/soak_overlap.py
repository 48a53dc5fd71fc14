"""Soak test of the overlapped halo-exchange executor with virtual ranks on one GPU: many solves of random depth /
halo / iteration count, each compared bit for bit with the same solve on a whole-domain context.  An ordering bug
between the compute stream and the exchange stream (events, buffer swaps) would show up intermittently."""
import sys, time, importlib
import numpy as np
sys.path.insert(0, ".")
sfl = importlib.import_module("esp32-fluid-simulation_amd")
capi = sfl.capi

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0, cases, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    nranks = int(rng.choice([2, 3, 4, 8]))
    dim_x = int(rng.choice([512, 1024, 2048, 3000]))
    dim_y = int(rng.integers(nranks * 70, nranks * 400))
    iters = int(rng.integers(3, 70))
    fuse = int(rng.choice([0, 4, 8, 10, 12, 16]))
    halo = int(rng.choice([0, 16, 32, 64]))
    if halo and fuse and halo < fuse:
        halo = 0
    d = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
    with sfl.Solver(dim_x, dim_y) as one:
        if fuse:
            one.set_option(capi.OPT_SOR_KERNEL, 2)
            one.set_option(capi.OPT_SOR_FUSE, fuse)
        one.upload(capi.FIELD_DIVERGENCE, d)
        one.poisson_solve(1.0, iters, 1.96)
        one.synchronize()
        want = one.download(capi.FIELD_PRESSURE)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        if min(s.row_end - s.row_begin for s in slabs) < max(fuse, halo, 16):
            continue
        sfl.Solver.link_group(slabs)
        slabs[0].set_option(capi.OPT_SOR_KERNEL, 2)
        slabs[0].set_option(capi.OPT_SOR_FUSE, fuse)
        slabs[0].set_option(capi.OPT_SOR_HALO, halo)
        slabs[0].set_option(capi.OPT_EXCHANGE_SCHEDULE, 3 if int(rng.integers(0, 4)) > 0 else 2)   # mostly the device-side arrival count
        for rep in range(3):          # back-to-back solves on the same contexts: stale ghost rows, swapped buffers
            for s in slabs:
                s.upload(capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
            slabs[0].poisson_solve(1.0, iters, 1.96)
            slabs[0].synchronize()
            got = np.concatenate([s.download(capi.FIELD_PRESSURE) for s in slabs], axis=0)
            cases += 1
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                bad += 1
                print(f"MISMATCH nranks {nranks} {dim_x}x{dim_y} iters {iters} fuse {fuse} halo {halo} rep {rep}: "
                      f"{int(np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))} cells", flush=True)
    finally:
        for s in slabs:
            s.close()
print(f"{cases} overlapped solves on virtual ranks in {time.time() - t0:.0f} s: {bad} mismatches", flush=True)

# whole steps: slabs (automatic advection halo, overlapped solve) against a whole-domain context, several steps in a row
t1, steps = time.time(), 0
while time.time() - t1 < budget / 2:
    nranks = int(rng.choice([2, 3, 4]))
    dim_x = int(rng.choice([256, 1000, 2048]))
    dim_y = int(rng.integers(nranks * 80, nranks * 300))
    iters = int(rng.integers(2, 30))
    v = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * float(rng.choice([30.0, 100.0, 400.0]))).astype(np.float32)
    c = rng.integers(0, 2 ** 31, (dim_y, dim_x, 3), dtype=np.uint32)
    with sfl.Solver(dim_x, dim_y) as one:
        one.upload(capi.FIELD_VELOCITY, v); one.upload(capi.FIELD_COLOR, c)
        for _ in range(3):
            one.step(np.float32(1 / 30), 1.0, iters, 1.96)
        one.synchronize()
        want = [one.download(f) for f in (capi.FIELD_VELOCITY, capi.FIELD_COLOR, capi.FIELD_PRESSURE)]
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        slabs[0].set_option(capi.OPT_ADVECT_HALO, 0)
        for s in slabs:
            s.upload(capi.FIELD_VELOCITY, v[s.row_begin:s.row_end]); s.upload(capi.FIELD_COLOR, c[s.row_begin:s.row_end])
        for _ in range(3):
            slabs[0].step(np.float32(1 / 30), 1.0, iters, 1.96)
        slabs[0].synchronize()
        steps += 3
        for f, w in zip((capi.FIELD_VELOCITY, capi.FIELD_COLOR, capi.FIELD_PRESSURE), want):
            got = np.concatenate([s.download(f) for s in slabs], axis=0)
            if not np.array_equal(got.view(np.uint32), w.view(np.uint32)):
                bad += 1
                print(f"STEP MISMATCH field {f} nranks {nranks} {dim_x}x{dim_y} iters {iters}", flush=True)
    finally:
        for s in slabs:
            s.close()
print(f"{steps} slab steps against whole-domain steps: {bad} mismatches in all")
sys.exit(1 if bad else 0)
