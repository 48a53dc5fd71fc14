"""Which side loses denormals?  The fused kernel, the one-pass kernel and the oracle on right-hand sides scaled to the bottom of
the float range, against a float64-free restatement in numpy (float32 operations one by one, host FPU).

    python tests/denormal_probe.py [path/to/other/libsfl_hip.so]
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    capi = importlib.import_module("esp32-fluid-simulation_amd._capi")
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
sfl = importlib.import_module("esp32-fluid-simulation_amd")
from oracle import loader  # noqa: E402  (test infrastructure: this script lives under tests/)

oracle = loader.port()
a, b = np.float32(2.0 ** -130), np.float32(0.5)
print("host FPU keeps denormals:", float(a * b) != 0.0, float(a * b))
rng = np.random.default_rng(21)
d = rng.standard_normal((150, 300)).astype(np.float32)
OMEGA = np.float32(1.96)
for e in (-100, -110, -118, -120, -122, -126, -130, -140):
    tiny = (d * np.float32(2.0 ** e)).astype(np.float32)
    fused = sfl.HostPath(sor_kernel=2, sor_fuse=8).poisson_solve(tiny, 1.0, 6, OMEGA)
    one = sfl.HostPath(sor_kernel=1).poisson_solve(tiny, 1.0, 6, OMEGA)
    line = f"2^{e}: fused vs one-pass {np.max(np.abs(fused.astype(np.float64) - one.astype(np.float64))) / 2.0 ** -149:.0f} units of 2^-149"
    if oracle is not None:
        want = oracle.poisson_solve(tiny, 1.0, 6, OMEGA)
        line += f" | fused vs oracle {np.max(np.abs(fused.astype(np.float64) - want.astype(np.float64))) / 2.0 ** -149:.0f} | one-pass vs oracle {np.max(np.abs(one.astype(np.float64) - want.astype(np.float64))) / 2.0 ** -149:.0f}"
    print(line, flush=True)
