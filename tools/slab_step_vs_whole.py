"""Whole sim steps on 2-4 virtual ranks against a whole-domain context, field by field after EVERY step, with the rows that differ
and the cuts printed (the soak in tools/soak_overlap.py only says that something differs): 40 random configurations,
per-cell velocity noise of up to 13 rows per step.  Found the halo bug of the extended velocity advection in round 3."""
import sys, importlib, os
import numpy as np
sys.path.insert(0, ".")
sfl = importlib.import_module("esp32-fluid-simulation_amd")
capi = sfl.capi
rng = np.random.default_rng(77)
bad = 0
for case in range(40):
    nranks = int(rng.choice([2, 3, 4]))
    dim_x = int(rng.choice([256, 1000, 2048]))
    dim_y = int(rng.integers(nranks * 80, nranks * 300))
    iters = int(rng.integers(2, 30))
    vamp = float(rng.choice([30.0, 100.0, 400.0]))
    v = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * vamp).astype(np.float32)
    c = rng.integers(0, 2 ** 31, (dim_y, dim_x, 3), dtype=np.uint32)
    nsteps = 3
    with sfl.Solver(dim_x, dim_y) as one:
        one.upload(capi.FIELD_VELOCITY, v); one.upload(capi.FIELD_COLOR, c)
        wants = []
        for _ in range(nsteps):
            one.step(np.float32(1 / 30), 1.0, iters, 1.96)
            one.synchronize()
            wants.append([one.download(f) for f in (capi.FIELD_VELOCITY, capi.FIELD_COLOR, capi.FIELD_PRESSURE, capi.FIELD_DIVERGENCE)])
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        for s in slabs:
            s.upload(capi.FIELD_VELOCITY, v[s.row_begin:s.row_end]); s.upload(capi.FIELD_COLOR, c[s.row_begin:s.row_end])
        for k in range(nsteps):
            slabs[0].step(np.float32(1 / 30), 1.0, iters, 1.96)
            slabs[0].synchronize()
            for name, f, w in zip("v c p d".split(), (capi.FIELD_VELOCITY, capi.FIELD_COLOR, capi.FIELD_PRESSURE, capi.FIELD_DIVERGENCE), wants[k]):
                got = np.concatenate([s.download(f) for s in slabs], axis=0)
                neq = got.view(np.uint32) != w.view(np.uint32)
                if neq.any():
                    bad += 1
                    rows = np.unique(np.nonzero(neq)[0])
                    print(f"case {case} step {k} field {name}: nranks {nranks} {dim_x}x{dim_y} iters {iters} vamp {vamp}: {int(neq.sum())} values, rows {rows[:6]}..{rows[-3:]} cuts {[s.row_begin for s in slabs]} fuse {slabs[0].last_solve_info()}", flush=True)
                    break
            else:
                continue
            break
    finally:
        for s in slabs:
            s.close()
print("bad", bad)
