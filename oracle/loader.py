"""ctypes access to the CPU checker libraries.  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module (see oracle/sf_oracle.c header).  Two libraries:

* ``port()``      -> oracle/libsf_oracle.so   this repo's C restatement (always buildable)
* ``reference()`` -> oracle/_ref/libsf_ref.so the unmodified reference sources compiled in
                     place by oracle/Makefile (present wherever it was built; the GPU box
                     receives the prebuilt file, it cannot rebuild it)

Both expose the same numpy-level API through :class:`CpuPath`:
fields are C-contiguous arrays, velocity ``float32[dim_y, dim_x, 2]``, dye
``uint32[dim_y, dim_x, 3]``, scalars ``float32[dim_y, dim_x]`` -- i.e. element
``(i, j)`` at ``dim_x*j + i`` exactly like the reference (operations.h:7-9).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = C.POINTER(C.c_float)
_U = C.POINTER(C.c_uint32)


def build(quiet: bool = True) -> None:
    """(Re)build the checker libraries with oracle/Makefile."""
    subprocess.run(["make", "-C", _HERE] + (["-s"] if quiet else []), check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(_F)


def _up(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags.c_contiguous
    return a.ctypes.data_as(_U)


class CpuPath:
    """numpy front-end over one of the two checker libraries."""

    def __init__(self, lib: C.CDLL, prefix: str, kind: str):
        self.lib, self.kind = lib, kind
        g = lambda name: getattr(lib, prefix + name)
        self._adv2 = g("advect_vec2f")
        self._adv2.argtypes = [_F, _F, _F, C.c_int, C.c_int, C.c_float, C.c_int]
        self._adv3 = g("advect_vec3uq32")
        self._adv3.argtypes = [_U, _U, _F, C.c_int, C.c_int, C.c_float, C.c_int]
        self._div = g("divergence")
        self._div.argtypes = [_F, _F, C.c_int, C.c_int, C.c_float]
        self._grad = g("subtract_gradient")
        self._grad.argtypes = [_F, _F, C.c_int, C.c_int, C.c_float]
        self._pois = g("poisson_solve")
        self._pois.argtypes = [_F, _F, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float]
        self._step = g("step")
        self._step.argtypes = [_F, _U, _F, _F, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                               C.c_float]
        self._step.restype = C.c_int
        for f in (self._adv2, self._adv3, self._div, self._grad, self._pois):
            f.restype = None
        self._advc = g("advect_channels")
        self._advc.argtypes = [C.c_void_p, C.c_void_p, _F, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        self._advc.restype = C.c_int if prefix == "ref_" else None

    @staticmethod
    def _dims(a: np.ndarray):
        return int(a.shape[1]), int(a.shape[0])  # dim_x, dim_y

    def advect_vec2f(self, p, vel, dt, no_slip=True):
        dim_x, dim_y = self._dims(vel)
        out = np.empty_like(p)
        self._adv2(_fp(out), _fp(p), _fp(vel), dim_x, dim_y, dt, int(no_slip))
        return out

    def advect_vec3uq32(self, p, vel, dt, no_slip=False):
        dim_x, dim_y = self._dims(vel)
        out = np.empty_like(p)
        self._adv3(_up(out), _up(p), _fp(vel), dim_x, dim_y, dt, int(no_slip))
        return out

    def advect_channels(self, p, vel, dt, no_slip):
        """advect<T, float> for T = float / UQ32 / Vector2 / Vector3 of either: `p` is float32 or uint32 (UQ32 raw)
        of shape [dim_y, dim_x] (scalar) or [dim_y, dim_x, 2 | 3]."""
        dim_x, dim_y = self._dims(vel)
        assert p.dtype in (np.float32, np.uint32) and p.flags.c_contiguous and p.shape[:2] == vel.shape[:2]
        channels = 1 if p.ndim == 2 else int(p.shape[2])
        out = np.empty_like(p)
        rc = self._advc(out.ctypes.data, p.ctypes.data, _fp(vel), dim_x, dim_y, dt, int(no_slip), channels,
                        1 if p.dtype == np.uint32 else 0)
        assert not rc
        return out

    def divergence(self, v, dx=1.0):
        dim_x, dim_y = self._dims(v)
        out = np.empty((dim_y, dim_x), np.float32)
        self._div(_fp(out), _fp(v), dim_x, dim_y, dx)
        return out

    def subtract_gradient(self, v, p, dx=1.0):
        """Returns the projected velocity (the input array is not modified)."""
        dim_x, dim_y = self._dims(v)
        out = v.copy()
        self._grad(_fp(out), _fp(p), dim_x, dim_y, dx)
        return out

    def poisson_solve(self, div, dx=1.0, iters=10, omega=1.96):
        dim_x, dim_y = self._dims(div)
        p = np.empty((dim_y, dim_x), np.float32)
        self._pois(_fp(p), _fp(div), dim_x, dim_y, dx, iters, omega)
        return p

    def step(self, v, colour, dt, dx=1.0, iters=10, omega=1.96):
        """One sim step (ino:252-287 order).  Returns (v, div, p, colour) copies."""
        dim_x, dim_y = self._dims(v)
        v, colour = v.copy(), colour.copy()
        div = np.empty((dim_y, dim_x), np.float32)
        p = np.empty((dim_y, dim_x), np.float32)
        rc = self._step(_fp(v), _up(colour), _fp(div), _fp(p), dim_x, dim_y, dt, dx, iters, omega)
        if rc != 0:
            raise MemoryError("checker step failed")
        return v, div, p, colour


class OraclePort(CpuPath):
    """The C restatement; also carries the helpers only it has."""

    def __init__(self, lib):
        super().__init__(lib, "orc_", "port")
        self._half = lib.orc_sor_half_sweep_rows
        self._half.argtypes = [_F, _F, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                               C.c_int, C.c_int]
        self._half.restype = None
        self._iter = lib.orc_sor_iterate
        self._iter.argtypes = [_F, _F, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float]
        self._iter.restype = None
        self._lcg = lib.orc_lcg_fields
        self._lcg.argtypes = [_F, _U, C.c_int, C.c_int, C.c_uint32, C.c_float]
        self._lcg.restype = None
        self._render = lib.orc_render_rgb565
        self._render.argtypes = [C.POINTER(C.c_uint16), _U, C.c_int, C.c_int, C.c_int, C.c_int]
        self._render.restype = None
        self._setup = lib.orc_setup_fields
        self._setup.argtypes = [_F, _U, C.c_int, C.c_int]
        self._setup.restype = None
        self._fnv = lib.orc_fnv1a64
        self._fnv.argtypes = [C.c_void_p, C.c_size_t]
        self._fnv.restype = C.c_uint64

    def sor_half_sweep_rows(self, p, d, gdim_y, colour, row_begin, row_end, grow0, dx=1.0,
                            omega=1.96):
        """In place on local array ``p`` (rows = global rows grow0..)."""
        dim_x = int(p.shape[1])
        self._half(_fp(p), _fp(d), dim_x, gdim_y, dx, omega, colour, row_begin, row_end, grow0)

    def sor_iterate(self, p, div, dx=1.0, iters=1, omega=1.96):
        dim_x, dim_y = self._dims(div)
        p = p.copy()
        self._iter(_fp(p), _fp(div), dim_x, dim_y, dx, iters, omega)
        return p

    def render_rgb565(self, colour, scaling=4, byteswap=True):
        """Draw-task arithmetic (ino:116-176; unpinned).  Returns uint16[scaling*(dim_x-1), scaling*(dim_y-1)]."""
        dim_x, dim_y = self._dims(colour)
        img = np.empty((scaling * (dim_x - 1), scaling * (dim_y - 1)), np.uint16)
        self._render(img.ctypes.data_as(C.POINTER(C.c_uint16)), _up(colour), dim_x, dim_y, scaling,
                     int(byteswap))
        return img

    def setup_fields(self, dim_x, dim_y):
        """Initial condition of the sketch's setup() (ino:196-241; unpinned).  Returns (v, colour)."""
        v = np.empty((dim_y, dim_x, 2), np.float32)
        c = np.empty((dim_y, dim_x, 3), np.uint32)
        self._setup(_fp(v), _up(c), dim_x, dim_y)
        return v, c

    def lcg_fields(self, dim_x, dim_y, seed, vamp):
        v = np.empty((dim_y, dim_x, 2), np.float32)
        c = np.empty((dim_y, dim_x, 3), np.uint32)
        self._lcg(_fp(v), _up(c), dim_x, dim_y, seed, vamp)
        return v, c

    def fnv1a64(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a)
        return int(self._fnv(a.ctypes.data, a.nbytes))


_port = None
_ref = None


def port() -> OraclePort:
    global _port
    if _port is None:
        path = os.path.join(_HERE, "libsf_oracle.so")
        if not os.path.exists(path):
            build()
        _port = OraclePort(C.CDLL(path))
    return _port


def reference_available() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libsf_ref.so"))


def reference() -> CpuPath:
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libsf_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (build it with `make -C oracle ref` where "
                                    "/root/reference exists)")
        _ref = CpuPath(C.CDLL(path), "ref_", "reference")
    return _ref
