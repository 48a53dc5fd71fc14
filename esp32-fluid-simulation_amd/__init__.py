"""MI355X-native stable-fluids hot path (the sim task of colonelwatch/ESP32-fluid-simulation).

The product is ``lib/libsfl_hip.so`` -- hand-written HIP kernels for gfx950 behind the C ABI of
``include/sfl.h`` -- plus the C++ drop-in headers in ``include/sfl/``.  This Python package is
only the binding used by the tests and by ``bench.py``:

* :mod:`._capi`   ctypes declarations, 1:1 with include/sfl.h
* :mod:`.solver`  ``Solver`` (a context) and ``HostPath`` (reference-style operators on host
  arrays, executed on the GPU)

Import it with ``importlib.import_module("esp32-fluid-simulation_amd")`` (the directory name is
not a Python identifier).  Importing never touches the GPU and never builds anything; the first
call that needs the library fails loudly if it is missing -- there is no CPU fallback.
"""
from . import _capi as capi
from ._capi import LIB_PATH, SflError, build_library
from .solver import (HostPath, Solver, comm_unique_id, device_count, device_info, plan_poisson,
                     slab_rows, sor_pass_plan, stdout_to_stderr)

__all__ = ["capi", "LIB_PATH", "SflError", "build_library", "HostPath", "Solver",
           "comm_unique_id", "device_count", "device_info", "plan_poisson", "slab_rows",
           "sor_pass_plan", "stdout_to_stderr"]
