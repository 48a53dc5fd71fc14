#!/usr/bin/env python3
"""Experiment aid (never used by the product, the tests or the driver's bench run): run a script of this
repo against an ALTERNATIVE build of the library -- compile-time variants for A/B measurements --

    python tools/with_lib.py path/to/libsfl_variant.so bench.py --steps 20 ...

by pointing the binding at that file before anything loads it.  The product binding itself
(esp32-fluid-simulation_amd/_capi.py) knows one path only."""
import importlib
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    lib, script = os.path.abspath(sys.argv[1]), sys.argv[2]
    if not os.path.exists(lib):
        sys.exit(f"with_lib.py: {lib} does not exist")
    capi = importlib.import_module("esp32-fluid-simulation_amd._capi")
    assert capi._lib is None, "with_lib.py: the product library is already loaded (something loaded it at import time)"
    capi.LIB_PATH = lib
    os.environ["SFL_WITH_LIB"] = lib     # bench.py's launchers start their rank processes through this script too (worker_argv)
    loaded = capi.lib()
    assert os.path.samefile(loaded._name, lib), f"with_lib.py: loaded {loaded._name}, asked for {lib}"
    sys.argv = [script] + sys.argv[3:]
    runpy.run_path(os.path.join(ROOT, script) if not os.path.isabs(script) else script, run_name="__main__")
