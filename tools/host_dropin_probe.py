"""Wall time of the host-pointer drop-in (upload + kernels + download) at the headline size."""
import ctypes as C, importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
sfl = importlib.import_module("esp32-fluid-simulation_amd")
lib = sfl.capi.lib()
n, iters = 8192, 80
d = (np.random.default_rng(1).standard_normal((n, n)) * 0.1).astype(np.float32)
p = np.empty_like(d)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
for rep in range(3):
    t0 = time.perf_counter()
    rc = lib.sfl_host_poisson_solve(fp(p), fp(d), n, n, C.c_float(1.0), iters, C.c_float(1.96))
    dt = time.perf_counter() - t0
    assert rc == 0
    print(f"sfl_host_poisson_solve 8192^2 x {iters} iters: {dt*1e3:.1f} ms wall "
          f"({n*n*iters/dt:.3e} cell-iters/s incl. context creation, 2 x 256 MiB over PCIe from pageable memory)")
