"""numpy-level front-end over the C ABI.

Two classes:

* :class:`Solver` -- one context (= one GPU's row slab, fields resident in HBM); thin, explicit.
* :class:`HostPath` -- the reference's operator interface on host arrays (same names, argument
  meaning and result layout as the oracle's ``CpuPath``), each call going through the
  ``sfl_host_*`` drop-ins or a temporary :class:`Solver`.  This is what the parity tests drive.

Array conventions (identical to the reference, operations.h:7-9): C-contiguous, row j major,
velocity ``float32[rows, dim_x, 2]``, dye ``uint32[rows, dim_x, 3]``, scalars ``float32[rows, dim_x]``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as capi

_FIELD_SPEC = {
    capi.FIELD_VELOCITY: (np.float32, 2),
    capi.FIELD_COLOR: (np.uint32, 3),
    capi.FIELD_DIVERGENCE: (np.float32, 1),
    capi.FIELD_PRESSURE: (np.float32, 1),
}


def device_count() -> int:
    n = C.c_int(0)
    rc = capi.lib().sfl_device_count(C.byref(n))
    return n.value if rc == capi.OK else 0


def device_info(device: int = 0):
    name = C.create_string_buffer(256)
    cus, mem = C.c_int(0), C.c_size_t(0)
    capi.check(capi.lib().sfl_device_info(device, name, 256, C.byref(cus), C.byref(mem)))
    return name.value.decode(), cus.value, mem.value


def slab_rows(dim_y: int, nranks: int, rank: int):
    b, e = C.c_int(), C.c_int()
    capi.check(capi.lib().sfl_slab_rows(dim_y, nranks, rank, C.byref(b), C.byref(e)))
    return b.value, e.value


def sor_pass_plan(iters: int, fuse: int):
    n = C.c_int()
    capi.check(capi.lib().sfl_sor_pass_plan(iters, fuse, C.byref(n), None, 0))
    arr = (C.c_int * max(n.value, 1))()
    capi.check(capi.lib().sfl_sor_pass_plan(iters, fuse, C.byref(n), arr, n.value))
    return list(arr[: n.value])


def plan_poisson(dim_y: int, nranks: int, rank: int, iters: int, fuse: int = 8, kernel: int = 2,
                 halo: int = 0, tail: int = 0):
    """The launch / halo-exchange program of one poisson_solve for one rank (pure arithmetic).
    halo = rows of p exchanged per superstep (0: exchange before every launch); tail = ghost rows of p left
    exact at the end (early-exchange plans only)."""
    n = C.c_int()
    capi.check(capi.lib().sfl_plan_poisson_tail(dim_y, nranks, rank, iters, fuse, kernel, halo, tail, None, 0,
                                                C.byref(n)))
    steps = (capi.PlanStep * max(n.value, 1))()
    capi.check(capi.lib().sfl_plan_poisson_tail(dim_y, nranks, rank, iters, fuse, kernel, halo, tail, steps,
                                                n.value, C.byref(n)))
    return [steps[k] for k in range(n.value)]


class stdout_to_stderr:
    """RCCL prints a version banner on the C library's stdout while a communicator comes up (buffered: it would surface when
    the process exits).  A program whose stdout carries ONE JSON line wraps communicator creation in this: file descriptor 1
    points at stderr for the duration, and the C library's buffers are flushed before it is restored."""

    def __enter__(self):
        import os
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import os
        try:
            C.CDLL(None).fflush(None)
        finally:
            os.dup2(self._saved, 1)
            os.close(self._saved)


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(capi.UNIQUE_ID_BYTES)
    capi.check(capi.lib().sfl_comm_unique_id(buf, capi.UNIQUE_ID_BYTES))
    return buf.raw


class Solver:
    """One solver context: rows [row_begin, row_end) of a dim_x * dim_y domain on one device."""

    def __init__(self, dim_x: int, dim_y: int, device: int = 0, rank: int = 0, nranks: int = 1):
        self._h = C.c_void_p()
        self._lib = capi.lib()
        capi.check(self._lib.sfl_create_slab(C.byref(self._h), device, dim_x, dim_y, rank, nranks))
        self.dim_x, self.dim_y, self.rank, self.nranks = dim_x, dim_y, rank, nranks
        self.row_begin, self.row_end = slab_rows(dim_y, nranks, rank)

    # -- lifetime ------------------------------------------------------------------------
    def close(self):
        if self._h:
            self._lib.sfl_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def rows(self) -> int:
        return self.row_end - self.row_begin

    # -- configuration -------------------------------------------------------------------
    def set_option(self, option: int, value: int):
        capi.check(self._lib.sfl_set_option(self._h, option, value))

    def get_option(self, option: int) -> int:
        v = C.c_int()
        capi.check(self._lib.sfl_get_option(self._h, option, C.byref(v)))
        return v.value

    def comm_attach(self, unique_id: bytes):
        capi.check(self._lib.sfl_comm_attach(self._h, unique_id, len(unique_id)))

    def comm_check_options(self):
        """Collective: every rank of the communicator must carry the same domain and options (sfl_comm_check_options)."""
        capi.check(self._lib.sfl_comm_check_options(self._h))

    def comm_loopback(self, rows: int):
        capi.check(self._lib.sfl_comm_loopback(self._h, rows))

    def comm_emulate(self):
        """This rank's program alone, halo messages as self-copies (timing only; sfl_comm_emulate)."""
        capi.check(self._lib.sfl_comm_emulate(self._h))

    def comm_emulate_rccl(self):
        """This rank's program alone with real RCCL messages to itself as the transport (timing only; sfl_comm_emulate_rccl)."""
        capi.check(self._lib.sfl_comm_emulate_rccl(self._h))

    @staticmethod
    def link_group(solvers):
        """Join slabs living on one device into an in-process group (virtual ranks)."""
        arr = (C.c_void_p * len(solvers))(*[s._h for s in solvers])
        capi.check(capi.lib().sfl_group_link(arr, len(solvers)))

    # -- field I/O (owned rows) --------------------------------------------------------------
    def _shape(self, field):
        dt, nc = _FIELD_SPEC[field]
        return (self.rows, self.dim_x, nc) if nc > 1 else (self.rows, self.dim_x)

    def upload(self, field: int, a: np.ndarray):
        dt, _ = _FIELD_SPEC[field]
        a = np.ascontiguousarray(a, dtype=dt)
        if a.shape != self._shape(field):
            raise ValueError(f"field {field}: shape {a.shape}, slab wants {self._shape(field)}")
        capi.check(self._lib.sfl_upload(self._h, field, a.ctypes.data, a.nbytes))

    def download(self, field: int) -> np.ndarray:
        dt, _ = _FIELD_SPEC[field]
        a = np.empty(self._shape(field), dt)
        capi.check(self._lib.sfl_download(self._h, field, a.ctypes.data, a.nbytes))
        return a

    def device_ptr(self, field: int) -> int:
        p = C.c_void_p()
        capi.check(self._lib.sfl_field_device_ptr(self._h, field, C.byref(p)))
        return p.value

    # -- operators (asynchronous on the context's stream) -----------------------------------
    def advect_velocity(self, dt, no_slip=True):
        capi.check(self._lib.sfl_advect_velocity(self._h, dt, int(no_slip)))

    def advect_color(self, dt, no_slip=False):
        capi.check(self._lib.sfl_advect_color(self._h, dt, int(no_slip)))

    def advect_external(self, next_p_dev: int, p_dev: int, channels: int, kind: int, dt, no_slip):
        """A field of the caller's (device pointers) advected with the resident velocity (sfl_advect_external)."""
        capi.check(self._lib.sfl_advect_external(self._h, next_p_dev, p_dev, channels, kind, dt, int(no_slip)))

    def calculate_divergence(self, dx=1.0):
        capi.check(self._lib.sfl_calculate_divergence(self._h, dx))

    def poisson_solve(self, dx=1.0, iters=10, omega=1.96):
        capi.check(self._lib.sfl_poisson_solve(self._h, dx, iters, omega))

    def subtract_gradient(self, dx=1.0):
        capi.check(self._lib.sfl_subtract_gradient(self._h, dx))

    def step(self, dt, dx=1.0, iters=10, omega=1.96):
        capi.check(self._lib.sfl_step(self._h, dt, dx, iters, omega))

    def step_n(self, n, dt, dx=1.0, iters=10, omega=1.96):
        """n steps in one call (sfl_step_n): the same results as n x step(), fused across the step boundaries."""
        capi.check(self._lib.sfl_step_n(self._h, n, dt, dx, iters, omega))

    def queue_forces(self, cells_ij, vel_xy):
        cells = np.ascontiguousarray(cells_ij, np.int32).reshape(-1, 2)
        vel = np.ascontiguousarray(vel_xy, np.float32).reshape(-1, 2)
        capi.check(self._lib.sfl_queue_forces(
            self._h, cells.ctypes.data_as(C.POINTER(C.c_int)),
            vel.ctypes.data_as(C.POINTER(C.c_float)), len(cells)))

    def queue_drags(self, drags):
        """drags: iterable of (coords_x, coords_y, velocity_x, velocity_y) in the sketch's graphics coordinates
        (struct drag, ino:45-48); transformed like ino:264-269 by the library."""
        drags = list(drags)
        arr = (capi.Drag * max(len(drags), 1))(*[capi.Drag(int(a), int(b), float(c), float(d)) for a, b, c, d in drags])
        capi.check(self._lib.sfl_queue_drags(self._h, C.cast(arr, C.c_void_p), len(drags)))

    def setup_sketch_fields(self):
        """Velocity = 0, dye = the sketch's blurred three-sector pattern (setup(), ino:196-241)."""
        capi.check(self._lib.sfl_setup_sketch_fields(self._h))

    def render_rgb565(self, scaling: int = 4, byteswap: bool = True) -> np.ndarray:
        """Dye field -> RGB565 image, uint16[scaling*(dim_x-1), scaling*(dim_y-1)] (ino:116-176)."""
        img = np.empty((scaling * (self.dim_x - 1), scaling * (self.dim_y - 1)), np.uint16)
        capi.check(self._lib.sfl_render_rgb565(self._h, scaling, int(byteswap),
                                               img.ctypes.data_as(C.POINTER(C.c_uint16)), img.nbytes))
        return img

    def synchronize(self):
        capi.check(self._lib.sfl_synchronize(self._h))

    def timer_start(self):
        capi.check(self._lib.sfl_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        capi.check(self._lib.sfl_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def last_solve_info(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        capi.check(self._lib.sfl_last_solve_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"launches": a.value, "exchanges": b.value, "fuse": c.value, "halo": self.get_option(capi.OPT_LAST_HALO)}


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class HostPath:
    """The reference's operators on host arrays, executed by the HIP kernels.

    Same method names / argument order / return conventions as ``oracle.loader.CpuPath`` so the
    parity tests read identically for the checker and the product.  ``sor_kernel`` / ``sor_fuse``
    select the SOR implementation (0 = auto)."""

    kind = "hip"

    def __init__(self, device: int = 0, sor_kernel: int = 0, sor_fuse: int = 0, sor_rows: int = 0,
                 sor_lane_cells: int = 0, sor_fold: int = 0):
        self._lib = capi.lib()
        self.device, self.sor_kernel, self.sor_fuse, self.sor_rows = device, sor_kernel, sor_fuse, sor_rows
        self.sor_lane_cells, self.sor_fold = sor_lane_cells, sor_fold

    @staticmethod
    def _dims(a):
        return int(a.shape[1]), int(a.shape[0])

    def _solver(self, dim_x, dim_y) -> Solver:
        s = Solver(dim_x, dim_y, self.device)
        if self.sor_kernel:
            s.set_option(capi.OPT_SOR_KERNEL, self.sor_kernel)
        if self.sor_fuse:
            s.set_option(capi.OPT_SOR_FUSE, self.sor_fuse)
        if self.sor_rows:
            s.set_option(capi.OPT_SOR_ROWS, self.sor_rows)
        if self.sor_lane_cells:
            s.set_option(capi.OPT_SOR_LANE_CELLS, self.sor_lane_cells)
        if self.sor_fold:
            s.set_option(capi.OPT_SOR_FOLD, self.sor_fold)
        return s

    def advect_vec2f(self, p, vel, dt, no_slip=True):
        dim_x, dim_y = self._dims(vel)
        p, vel = np.ascontiguousarray(p, np.float32), np.ascontiguousarray(vel, np.float32)
        out = np.empty_like(p)
        # keep the aliasing information: self-advection passes the same pointer twice (ino:253)
        pp = _fp(vel) if p is vel or (p.ctypes.data == vel.ctypes.data) else _fp(p)
        capi.check(self._lib.sfl_host_advect_vec2f(_fp(out), pp, _fp(vel), dim_x, dim_y, dt,
                                                   int(no_slip)))
        return out

    def advect_vec3uq32(self, p, vel, dt, no_slip=False):
        dim_x, dim_y = self._dims(vel)
        p, vel = np.ascontiguousarray(p, np.uint32), np.ascontiguousarray(vel, np.float32)
        out = np.empty_like(p)
        capi.check(self._lib.sfl_host_advect_vec3uq32(_up(out), _up(p), _fp(vel), dim_x, dim_y, dt,
                                                      int(no_slip)))
        return out

    def advect_channels(self, p, vel, dt, no_slip):
        """advect<T, float> for T = float / UQ32 / Vector2 / Vector3 of either (sfl_host_advect_channels): `p` is
        float32 or uint32 (UQ32 raw), shape [dim_y, dim_x] or [dim_y, dim_x, 2 | 3]."""
        dim_x, dim_y = self._dims(vel)
        if p.dtype not in (np.float32, np.uint32):
            raise TypeError("element channels are float32 or uint32 (UQ32 raw)")
        p, vel = np.ascontiguousarray(p), np.ascontiguousarray(vel, np.float32)
        channels = 1 if p.ndim == 2 else int(p.shape[2])
        out = np.empty_like(p)
        capi.check(self._lib.sfl_host_advect_channels(out.ctypes.data, p.ctypes.data, _fp(vel), dim_x, dim_y, dt,
                                                      int(no_slip), channels,
                                                      capi.CHANNEL_UQ32 if p.dtype == np.uint32 else capi.CHANNEL_F32))
        return out

    def divergence(self, v, dx=1.0):
        dim_x, dim_y = self._dims(v)
        v = np.ascontiguousarray(v, np.float32)
        out = np.empty((dim_y, dim_x), np.float32)
        capi.check(self._lib.sfl_host_calculate_divergence(_fp(out), _fp(v), dim_x, dim_y, dx))
        return out

    def subtract_gradient(self, v, p, dx=1.0):
        dim_x, dim_y = self._dims(v)
        out = np.array(v, np.float32, order="C", copy=True)
        p = np.ascontiguousarray(p, np.float32)
        capi.check(self._lib.sfl_host_subtract_gradient(_fp(out), _fp(p), dim_x, dim_y, dx))
        return out

    def poisson_solve(self, div, dx=1.0, iters=10, omega=1.96):
        dim_x, dim_y = self._dims(div)
        with self._solver(dim_x, dim_y) as s:
            s.upload(capi.FIELD_DIVERGENCE, div)
            s.poisson_solve(dx, iters, omega)
            s.synchronize()
            return s.download(capi.FIELD_PRESSURE)

    def step(self, v, colour, dt, dx=1.0, iters=10, omega=1.96):
        """One sim step (ino:252-287 order).  Returns (v, div, p, colour)."""
        dim_x, dim_y = self._dims(v)
        with self._solver(dim_x, dim_y) as s:
            s.upload(capi.FIELD_VELOCITY, v)
            s.upload(capi.FIELD_COLOR, colour)
            s.step(dt, dx, iters, omega)
            s.synchronize()
            return (s.download(capi.FIELD_VELOCITY), s.download(capi.FIELD_DIVERGENCE),
                    s.download(capi.FIELD_PRESSURE), s.download(capi.FIELD_COLOR))
