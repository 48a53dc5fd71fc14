"""r04: stress of the overlapped slab executor on row pitches that are NOT a multiple of the 128-byte cache line (a line then holds
the end of a ghost row and the beginning of an owned row), short slabs, shallow fuse depths: many exchanges per solve.
usage: unaligned_stress.py <seconds> <0 = exchanges behind events | 1 = in time>"""
import sys, time, importlib
import numpy as np
sys.path.insert(0, ".")
sfl = importlib.import_module("esp32-fluid-simulation_amd")
capi = sfl.capi
budget, arrival = float(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(7)
t0, cases, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    nranks = int(rng.choice([2, 2, 3, 4]))
    dim_x = int(rng.choice([3000, 3000, 1000, 2999, 1030]))
    dim_y = int(rng.integers(nranks * 100, nranks * 200))
    iters = int(rng.integers(20, 60))
    fuse = int(rng.choice([4, 4, 6, 8, 10]))
    halo = int(rng.choice([0, 16, 32]))
    if halo and halo < fuse:
        halo = 0
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
        sfl.Solver.link_group(slabs)
        slabs[0].set_option(capi.OPT_SOR_KERNEL, 2)
        slabs[0].set_option(capi.OPT_SOR_FUSE, fuse)
        slabs[0].set_option(capi.OPT_SOR_HALO, halo)
        slabs[0].set_option(capi.OPT_EXCHANGE_SCHEDULE, 3 if arrival else 2)
        for s in slabs:
            s.upload(capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
        for rep in range(8):
            slabs[0].poisson_solve(1.0, iters, 1.96)
            slabs[0].synchronize()
            got = np.concatenate([s.download(capi.FIELD_PRESSURE) for s in slabs], axis=0)
            cases += 1
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                bad += 1
                rows = np.unique(np.argwhere(got.view(np.uint32) != want.view(np.uint32))[:, 0])
                print(f"MISMATCH nranks {nranks} {dim_x}x{dim_y} iters {iters} fuse {fuse} halo {halo} rep {rep}: rows {rows.min()}..{rows.max()} "
                      f"cuts {[s.row_begin for s in slabs[1:]]}", flush=True)
    finally:
        for s in slabs:
            s.close()
print(f"exchanges {'in time' if arrival else 'behind events'}: {cases} solves on unaligned pitches in {time.time() - t0:.0f} s: {bad} mismatches", flush=True)
