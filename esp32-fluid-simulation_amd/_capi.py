"""ctypes binding of the C ABI (include/sfl.h).  Mirrors the header one to one; no logic."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libsfl_hip.so")   # the one product library (A/B builds: tools/with_lib.py)

OK, ERR_INVALID, ERR_HIP, ERR_RCCL, ERR_NOMEM, ERR_STATE, ERR_HALO = 0, -1, -2, -3, -4, -5, -6
FIELD_VELOCITY, FIELD_COLOR, FIELD_DIVERGENCE, FIELD_PRESSURE = 0, 1, 2, 3
OPT_SOR_KERNEL, OPT_SOR_FUSE, OPT_ADVECT_HALO, OPT_SOR_ROWS, OPT_TRANSPORT = 0, 1, 2, 3, 4
OPT_SOR_LANE_CELLS = 5
OPT_SOR_HALO = 6
OPT_FUSE_PROJECTION = 7
OPT_ADVECT_KERNEL = 9
OPT_FUSE_DIVERGENCE = 10
OPT_SMALL_GRID = 11
OPT_EMULATE_WIRE_US = 12
OPT_STEP_SEAMS = 14
OPT_LAST_EARLY_ROWS = 17
OPT_HALO_TIMEOUT_MS = 18
OPT_EXCHANGE_SCHEDULE = 19
SCHEDULE_AUTO, SCHEDULE_IN_LINE, SCHEDULE_BY_EVENT, SCHEDULE_IN_TIME = 0, 1, 2, 3   # values of OPT_EXCHANGE_SCHEDULE
OPT_MEASURED_WIRE_US = 20
OPT_LAST_HALO = 21
OPT_SOR_FOLD = 22
CHANNEL_F32, CHANNEL_UQ32 = 0, 1
STEP_EXCHANGE, STEP_SOR, STEP_ZERO = 1, 2, 3
UNIQUE_ID_BYTES = 128


class PlanStep(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "field", "rows", "g_begin", "g_end", "nsweeps",
                                         "first_colour", "from_zero")]


class Drag(C.Structure):
    """struct drag of the sketch (ino:45-48): graphics coordinates."""
    _fields_ = [("coord_x", C.c_uint16), ("coord_y", C.c_uint16), ("vel_x", C.c_float), ("vel_y", C.c_float)]


class SflError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"sfl error {code}: {message}")
        self.code = code


def build_library(verbose: bool = False) -> str:
    """Compile the HIP extension for gfx950 in-tree (csrc/Makefile); returns the .so path."""
    subprocess.run(["make", "-C", os.path.join(_PKG, "csrc"), "-j4"], check=True,
                   stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """The loaded product library.  Fails loudly when it has not been built: there is no
    Python or CPU substitute for it."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                f"(python -c 'import __graft_entry__ as g; g.build()' or make -C "
                f"{os.path.join(_PKG, 'csrc')}).  There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def check(rc: int) -> None:
    if rc != OK:
        raise SflError(rc, lib().sfl_last_error().decode())


_ctx = C.c_void_p
_i, _f, _sz = C.c_int, C.c_float, C.c_size_t
_pi, _pf, _pu = C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_uint32)

# name -> (restype, argtypes); every symbol include/sfl.h declares is listed here and checked
# against the built library by tests/test_capi_symbols.py
SIGNATURES = {
    "sfl_abi_version": (_i, []),
    "sfl_last_error": (C.c_char_p, []),
    "sfl_device_count": (_i, [_pi]),
    "sfl_device_info": (_i, [_i, C.c_char_p, _sz, _pi, C.POINTER(_sz)]),
    "sfl_slab_rows": (_i, [_i, _i, _i, _pi, _pi]),
    "sfl_plan_poisson": (_i, [_i, _i, _i, _i, _i, _i, _i, C.POINTER(PlanStep), _i, _pi]),
    "sfl_plan_poisson_tail": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(PlanStep), _i, _pi]),
    "sfl_sor_pass_plan": (_i, [_i, _i, _pi, _pi, _i]),
    "sfl_host_advect_vec2f": (_i, [_pf, _pf, _pf, _i, _i, _f, _i]),
    "sfl_host_advect_vec3uq32": (_i, [_pu, _pu, _pf, _i, _i, _f, _i]),
    "sfl_host_advect_channels": (_i, [C.c_void_p, C.c_void_p, _pf, _i, _i, _f, _i, _i, _i]),
    "sfl_host_calculate_divergence": (_i, [_pf, _pf, _i, _i, _f]),
    "sfl_host_subtract_gradient": (_i, [_pf, _pf, _i, _i, _f]),
    "sfl_host_poisson_solve": (_i, [_pf, _pf, _i, _i, _f, _i, _f]),
    "sfl_host_release": (_i, []),
    "sfl_create": (_i, [C.POINTER(_ctx), _i, _i, _i]),
    "sfl_create_slab": (_i, [C.POINTER(_ctx), _i, _i, _i, _i, _i]),
    "sfl_destroy": (_i, [_ctx]),
    "sfl_set_option": (_i, [_ctx, _i, _i]),
    "sfl_get_option": (_i, [_ctx, _i, _pi]),
    "sfl_slab_of": (_i, [_ctx, _pi, _pi, _pi, _pi]),
    "sfl_comm_unique_id": (_i, [C.c_void_p, _sz]),
    "sfl_comm_attach": (_i, [_ctx, C.c_void_p, _sz]),
    "sfl_comm_check_options": (_i, [_ctx]),
    "sfl_comm_loopback": (_i, [_ctx, _i]),
    "sfl_comm_emulate": (_i, [_ctx]),
    "sfl_comm_emulate_rccl": (_i, [_ctx]),
    "sfl_group_link": (_i, [C.POINTER(_ctx), _i]),
    "sfl_upload": (_i, [_ctx, _i, C.c_void_p, _sz]),
    "sfl_download": (_i, [_ctx, _i, C.c_void_p, _sz]),
    "sfl_field_device_ptr": (_i, [_ctx, _i, C.POINTER(C.c_void_p)]),
    "sfl_advect_velocity": (_i, [_ctx, _f, _i]),
    "sfl_advect_color": (_i, [_ctx, _f, _i]),
    "sfl_advect_external": (_i, [_ctx, C.c_void_p, C.c_void_p, _i, _i, _f, _i]),
    "sfl_calculate_divergence": (_i, [_ctx, _f]),
    "sfl_poisson_solve": (_i, [_ctx, _f, _i, _f]),
    "sfl_subtract_gradient": (_i, [_ctx, _f]),
    "sfl_step": (_i, [_ctx, _f, _f, _i, _f]),
    "sfl_step_n": (_i, [_ctx, _i, _f, _f, _i, _f]),
    "sfl_queue_forces": (_i, [_ctx, _pi, _pf, _i]),
    "sfl_queue_drags": (_i, [_ctx, C.c_void_p, _i]),
    "sfl_setup_sketch_fields": (_i, [_ctx]),
    "sfl_render_rgb565": (_i, [_ctx, _i, _i, C.POINTER(C.c_uint16), _sz]),
    "sfl_synchronize": (_i, [_ctx]),
    "sfl_timer_start": (_i, [_ctx]),
    "sfl_timer_stop": (_i, [_ctx, _pf]),
    "sfl_last_solve_info": (_i, [_ctx, _pi, _pi, _pi]),
}


def _declare(l: C.CDLL) -> None:
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args
