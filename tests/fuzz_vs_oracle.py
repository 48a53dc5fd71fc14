"""Test infrastructure, run by hand on a GPU box (python tests/fuzz_vs_oracle.py <seed> <seconds>): fuzz of the
single-GPU product path against the oracle (checker only): random grid shapes, kernel options, fuse
depths, dx / omega / dt, velocity scales, and -- since round 6 -- field TEXTURES: dense noise, a quiescent field with sparse forcing
at 60..130 iterations (the solution's front decays through the denormals: where round 5's folded product left the reference's bits),
and fields scaled down to the bottom of the float range; whole steps through sfl_step_n (all four fields), stand-alone solves and
advect<T, float> of a random element type, bit for bit."""
import sys, time, importlib
import numpy as np
sys.path.insert(0, ".")
sfl = importlib.import_module("esp32-fluid-simulation_amd")
from oracle import loader
capi, orc = sfl.capi, loader.port()
hp = sfl.HostPath()

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
big = len(sys.argv) > 3 and sys.argv[3] == "big"      # grids of 700 .. 3000 cells per side (many tiles per launch)
t0, cases, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    kind = rng.integers(0, 4)
    if big:
        dim_x, dim_y = int(rng.integers(700, 3000)), int(rng.integers(700, 3000))
        if rng.integers(0, 3) == 0:
            dim_x = int(rng.choice([1024, 2048, 1536, 2560]))
    elif kind == 0:
        dim_x, dim_y = int(rng.integers(2, 90)), int(rng.integers(2, 90))
    elif kind == 1:
        dim_x, dim_y = int(rng.integers(2, 700)), int(rng.integers(2, 700))
    elif kind == 2:
        dim_x, dim_y = int(rng.choice([64, 128, 192, 256, 320, 1024])), int(rng.integers(2, 400))
    else:
        dim_x, dim_y = int(rng.integers(2, 12)), int(rng.integers(200, 3000))
    iters = int(rng.integers(0, 26)) if not big else int(rng.integers(1, 20))
    dx = float(rng.choice([1.0, 1.0, 0.5, 1.37]))
    omega = np.float32(rng.choice([1.96, 1.0, 1.5]))
    dt = np.float32(rng.choice([1 / 30.0, 0.1, 0.004]))
    vamp = float(rng.choice([0.0, 20.0, 100.0, 1500.0]))
    opts = {capi.OPT_ADVECT_KERNEL: int(rng.choice([0, 1, 2])), capi.OPT_SOR_FUSE: int(rng.choice([0, 0, 2, 6, 10, 14, 16])),
            capi.OPT_SMALL_GRID: int(rng.choice([1, 1, 0])), capi.OPT_FUSE_DIVERGENCE: int(rng.choice([1, 0])),
            capi.OPT_FUSE_PROJECTION: int(rng.choice([1, 0])), capi.OPT_SOR_KERNEL: int(rng.choice([0, 0, 2, 1])),
            capi.OPT_STEP_SEAMS: int(rng.choice([1, 1, 0]))}
    n_steps = int(rng.choice([2, 2, 3]))          # through sfl_step_n: the seam kernel between the steps where it applies
    channels, uq = int(rng.integers(1, 4)), bool(rng.integers(0, 2))   # advect<T, float> of another element type
    texture = str(rng.choice(["dense", "dense", "sparse", "tiny"]))
    v = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * vamp).astype(np.float32)
    c = rng.integers(0, 2 ** 31, (dim_y, dim_x, 3), dtype=np.uint32)
    d = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
    if texture == "sparse":      # zero but for a few cells (ino:199, 264-269), enough iterations for the front to reach the denormals
        keep = rng.random((dim_y, dim_x)) < 3.0 / (dim_x * dim_y)
        keep[rng.integers(0, dim_y), rng.integers(0, dim_x)] = True
        d = np.where(keep, d * np.float32(rng.choice([1.0, 40.0, 1e-20])), np.float32(0.0)).astype(np.float32)
        v = np.where(keep[..., None], v, np.float32(0.0)).astype(np.float32)
        if not big:
            iters = int(rng.integers(60, 130))
    elif texture == "tiny":      # the whole field at the bottom of the float range
        scale = np.float32(2.0 ** -int(rng.integers(100, 146)))
        d, v = (d * scale).astype(np.float32), (v * scale).astype(np.float32)
    with sfl.Solver(dim_x, dim_y) as s:
        for k, val in opts.items():
            s.set_option(k, val)
        s.upload(capi.FIELD_DIVERGENCE, d)
        s.poisson_solve(dx, iters, omega)
        s.synchronize()
        got_p = s.download(capi.FIELD_PRESSURE)
        s.upload(capi.FIELD_VELOCITY, v); s.upload(capi.FIELD_COLOR, c)
        s.step_n(n_steps, dt, dx, iters, omega)
        s.synchronize()
        got = [s.download(f) for f in (capi.FIELD_VELOCITY, capi.FIELD_DIVERGENCE, capi.FIELD_PRESSURE, capi.FIELD_COLOR)]
    want = (v, None, None, c)
    for _ in range(n_steps):
        want = orc.step(want[0], want[3], dt, dx, iters, omega)
    ok = np.array_equal(got_p.view(np.uint32), orc.poisson_solve(d, dx, iters, omega).view(np.uint32))
    shape = (dim_y, dim_x) if channels == 1 else (dim_y, dim_x, channels)
    q = rng.integers(0, 2 ** 31, shape, dtype=np.uint32) if uq else (rng.standard_normal(shape) * 30).astype(np.float32)
    no_slip = bool(rng.integers(0, 2))
    ok = ok and np.array_equal(hp.advect_channels(q, v, dt, no_slip).view(np.uint32),
                               orc.advect_channels(q, v, dt, no_slip).view(np.uint32))
    for a, b in zip(got, want):
        ok = ok and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    cases += 1
    if not ok:
        bad += 1
        print(f"MISMATCH {dim_x}x{dim_y} {texture} iters {iters} dx {dx} omega {omega} dt {dt} vamp {vamp} options {opts}", flush=True)
print(f"{cases} random configurations (dense, sparse and denormal-range fields; solve + 2-3 steps through sfl_step_n + one generic advection each) against the oracle in {time.time() - t0:.0f} s: {bad} mismatches")
sys.exit(1 if bad else 0)
