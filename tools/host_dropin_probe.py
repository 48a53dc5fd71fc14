"""Wall time of the host-pointer drop-ins (upload + kernels + download), headline size and the
sketch's own 61 x 81 grid."""
import ctypes as C, importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
sfl = importlib.import_module("esp32-fluid-simulation_amd")
lib = sfl.capi.lib()
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
n, iters = 8192, 80
d = (np.random.default_rng(1).standard_normal((n, n)) * 0.1).astype(np.float32)
p = np.empty_like(d)
for rep in range(3):
    t0 = time.perf_counter()
    rc = lib.sfl_host_poisson_solve(fp(p), fp(d), n, n, C.c_float(1.0), iters, C.c_float(1.96))
    dt = time.perf_counter() - t0
    assert rc == 0
    print(f"sfl_host_poisson_solve 8192^2 x {iters} iters: {dt*1e3:.1f} ms wall "
          f"({n*n*iters/dt:.3e} cell-iters/s incl. 2 x 256 MiB over PCIe from pageable memory)")
hp = sfl.HostPath()
rng = np.random.default_rng(2)
dim_x, dim_y = 61, 81
v = (rng.standard_normal((dim_y, dim_x, 2)) * 3).astype(np.float32)
c = rng.integers(0, 2**31, (dim_y, dim_x, 3), dtype=np.uint32)
up = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
vt, ct = np.empty_like(v), np.empty_like(c)
div, p = np.empty((dim_y, dim_x), np.float32), np.empty((dim_y, dim_x), np.float32)
dt, one, omega = C.c_float(1 / 30), C.c_float(1.0), C.c_float(1.96)


def loop_body():  # the operator calls of loop(), ino:252-287
    global v, vt, c, ct
    assert lib.sfl_host_advect_vec2f(fp(vt), fp(v), fp(v), dim_x, dim_y, dt, 1) == 0
    v, vt = vt, v
    assert lib.sfl_host_calculate_divergence(fp(div), fp(v), dim_x, dim_y, one) == 0
    assert lib.sfl_host_poisson_solve(fp(p), fp(div), dim_x, dim_y, one, 10, omega) == 0
    assert lib.sfl_host_subtract_gradient(fp(v), fp(p), dim_x, dim_y, one) == 0
    assert lib.sfl_host_advect_vec3uq32(up(ct), up(c), fp(v), dim_x, dim_y, dt, 0) == 0
    c, ct = ct, c


for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(50):
        loop_body()
    dt_s = (time.perf_counter() - t0) / 50
    print(f"loop() body through the five host-pointer drop-ins, 61 x 81, 10 iters: {dt_s*1e3:.3f} ms per frame")
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(20):
        hp.step(v, c, np.float32(1 / 30), 1.0, 10, np.float32(1.96))
    dt_s = (time.perf_counter() - t0) / 20
    print(f"one context per frame (create, upload, sfl_step, download x 4, destroy), 61 x 81: {dt_s*1e3:.3f} ms")
