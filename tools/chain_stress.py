"""r04: stress of the chained launch (SFL_OPT_SOR_CHAIN): whole domains and two / three virtual ranks whose chains run side by
side with in-time halo exchanges inside them; pitches that are and are not whole cache lines, every chainable fuse depth,
halo depths that put one to five exchanges into a chain, one or many tiles per wave; every solve bit for bit against the
same solve by single launches on a whole-domain context.
usage: chain_stress.py <seconds> [seed]"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # three chains side by side need a hardware queue each (tests/conftest.py)
import sys, time, importlib
import numpy as np
sys.path.insert(0, ".")
sfl = importlib.import_module("esp32-fluid-simulation_amd")
capi = sfl.capi
budget = float(sys.argv[1])
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
t0, cases, bad, chained_total = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    nranks = int(rng.choice([1, 2, 2, 3, 3]))
    dim_x = int(rng.choice([3000, 1000, 2998, 1030, 2048, 4096, 8192]))
    dim_y = int(rng.integers(nranks * 120, nranks * 420))
    fuse = int(rng.choice([8, 10, 12, 16]))
    iters = int(rng.integers(2, 9)) * fuse // 2 + int(rng.integers(0, 2)) * (fuse // 2)
    halo = int(rng.choice([0, 16, 24, 32, 48, 64]))
    if halo and halo < fuse:
        halo = 0
    waves = int(rng.choice([1, 1, 64, 256]))
    d = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
    with sfl.Solver(dim_x, dim_y) as one:
        one.set_option(capi.OPT_SOR_KERNEL, 2)
        one.set_option(capi.OPT_SOR_FUSE, fuse)
        one.upload(capi.FIELD_DIVERGENCE, d)
        one.poisson_solve(1.0, iters, 1.96)
        one.synchronize()
        want = one.download(capi.FIELD_PRESSURE)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        if nranks > 1:
            if min(s.row_end - s.row_begin for s in slabs) < max(fuse, halo, 16):
                continue
            sfl.Solver.link_group(slabs)
            slabs[0].set_option(capi.OPT_SOR_HALO, halo)
        slabs[0].set_option(capi.OPT_SOR_KERNEL, 2)
        slabs[0].set_option(capi.OPT_SOR_FUSE, fuse)
        slabs[0].set_option(capi.OPT_SOR_CHAIN, waves)
        for s in slabs:
            s.upload(capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
        for rep in range(6):
            try:
                slabs[0].poisson_solve(1.0, iters, 1.96)
                slabs[0].synchronize()
            except sfl.SflError as e:
                bad += 1
                print(f"ERROR nranks {nranks} {dim_x}x{dim_y} iters {iters} fuse {fuse} halo {halo} waves {waves} rep {rep}: {e}", flush=True)
                break
            chained_total += slabs[-1].get_option(capi.OPT_LAST_CHAINED)
            got = np.concatenate([s.download(capi.FIELD_PRESSURE) for s in slabs], axis=0)
            cases += 1
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                bad += 1
                rows = np.unique(np.argwhere(got.view(np.uint32) != want.view(np.uint32))[:, 0])
                print(f"MISMATCH nranks {nranks} {dim_x}x{dim_y} iters {iters} fuse {fuse} halo {halo} waves {waves} rep {rep}: rows "
                      f"{rows.min()}..{rows.max()} cuts {[s.row_begin for s in slabs[1:]]}", flush=True)
    finally:
        for s in slabs:
            s.close()
print(f"chained launches: {cases} solves ({chained_total} supersteps inside chains) on whole domains and 2 / 3 virtual ranks in "
      f"{time.time() - t0:.0f} s: {bad} mismatches or errors", flush=True)
sys.exit(1 if bad else 0)
