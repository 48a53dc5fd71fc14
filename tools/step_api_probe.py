#!/usr/bin/env python3
"""K whole sim steps on four virtual ranks (8192 x 2048 domain, automatic advection halo, every option default) and
nothing else between the two markers -- run under `rocprofv3 --hip-trace` twice (K = 5, K = 25): the difference of the
HIP API call counts divided by 20 is what ONE step issues (rocprofv3 --hip-trace --stats -- python3 tools/step_api_probe.py)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sfl = importlib.import_module("esp32-fluid-simulation_amd")
K = int(sys.argv[1])
dim_x, dim_y, n = 8192, 2048, 4
slabs = [sfl.Solver(dim_x, dim_y, 0, r, n) for r in range(n)]
sfl.Solver.link_group(slabs)
for s in slabs:
    s.upload(sfl.capi.FIELD_VELOCITY, bench.synthetic_velocity(dim_x, s.row_begin, s.row_end))
    s.upload(sfl.capi.FIELD_COLOR, bench.synthetic_color(dim_x, s.row_begin, s.row_end))
dt, om = np.float32(1 / 30.0), np.float32(1.96)
for _ in range(3):          # first step measures (the velocity came from outside); buffers get allocated
    slabs[0].step(dt, 1.0, 80, om)
slabs[0].synchronize()
for _ in range(K):
    slabs[0].step(dt, 1.0, 80, om)
slabs[0].synchronize()
for s in slabs:
    s.close()
