"""CPU probe (test infrastructure, not collected by pytest): what SFL_OPT_SOR_FOLD = 1 costs on the sketch's own scenario -- a quiescent
field, a handful of drag forces, whole sim steps at 80 SOR iterations -- by running the PRODUCT's pipeline header lane by lane
(tests/cpp/libsor_stream_emu.so) with both arithmetics inside the oracle's step.  Prints, per step and field, the cells that differ,
the largest absolute difference and the largest reference value among the differing cells; then traces one solve iteration by
iteration.  Output committed as profiles/r06_numerics.txt.      python tests/fold_numerics_probe.py [steps]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_fields  # noqa: E402
from oracle import loader  # noqa: E402

o = loader.port()
OMEGA, DT = np.float32(1.96), np.float32(1 / 30.0)
lib = C.CDLL(os.path.join(ROOT, "tests", "cpp", "libsor_stream_emu.so"))
_F = C.POINTER(C.c_float)
lib.emu_sor_fused.argtypes = [_F, _F, _F] + [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int, C.c_int]


def launch(p_in, d, ns, fold, k):
    lrows, dim_x = d.shape
    out = np.full_like(d, np.nan)
    fp = lambda a: None if a is None else a.ctypes.data_as(_F)
    flags = 1 | (32 if k & 1 else 16) | (64 if fold else 0)
    assert lib.emu_sor_fused(fp(out), fp(p_in), fp(d), dim_x, lrows, 0, lrows, 0, lrows, ns, 1.0, OMEGA, 48, flags) >= 0
    return out


def solve(d, iters, fold, ns=16):
    p = None
    for k in range(2 * iters // ns):
        p = launch(p, d, ns, fold, k)
    return p


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dim_x, dim_y, iters = 1536, 1024, 80
    _, c, _ = random_fields(dim_x, dim_y, 77, 0.0)
    v = np.zeros((dim_y, dim_x, 2), np.float32)
    drags = [(512, 700, 35.0, -20.0), (513, 700, 30.0, -25.0), (100, 90, -60.0, 12.0), (900, 1400, 8.0, 90.0), (1023, 1535, 5.0, 5.0)]
    forces = ([(y, x) for x, y, _, _ in drags], [(vy, vx) for _, _, vx, vy in drags])
    res = {}
    for fold in (0, 1):
        vo, co = v, c
        for k in range(steps):
            va = o.advect_vec2f(vo, vo, DT, True)
            if k == 0:
                for (i, j), u in zip(*forces):
                    va[j, i] = u
            dd = o.divergence(va, 1.0)
            po = solve(dd, iters, fold)
            if not fold:
                assert np.array_equal(po.view(np.uint32), o.poisson_solve(dd, 1.0, iters, OMEGA).view(np.uint32)), "default arithmetic != oracle"
            vo = o.subtract_gradient(va, po, 1.0)
            co = o.advect_vec3uq32(co, vo, DT, False)
            res[fold, k] = (vo.copy(), dd.copy(), po.copy(), co.copy())
    print(f"# {dim_x} x {dim_y}, {iters} iterations, {steps} steps, fuse 16: default arithmetic (== oracle, asserted) against SFL_OPT_SOR_FOLD = 1")
    for k in range(steps):
        for n, a, b in zip(("velocity", "divergence", "pressure", "dye"), res[0, k], res[1, k]):
            df = a.view(np.uint32) != b.view(np.uint32)
            if n == "dye":
                print(f"step {k + 1} {n:10s} cells differing {int(df.sum())}")
                continue
            ad = np.abs(a.astype(np.float64) - b.astype(np.float64))
            print(f"step {k + 1} {n:10s} cells differing {int(df.sum()):7d}  max abs difference {ad.max():.3e}  largest |reference| among them "
                  f"{(np.abs(a[df]).max() if df.any() else 0):.3e}  field maximum {np.abs(a).max():.3e}")
    if steps >= 2:
        print("# the second step's solve, iteration by iteration: where the largest difference sits and how large the reference is there")
        dd = res[0, 1][1]
        pe = pf = None
        for it in range(iters):
            pe, pf = launch(pe, dd, 2, 0, 0), launch(pf, dd, 2, 1, 0)
            ad = np.abs(pe.astype(np.float64) - pf.astype(np.float64))
            j, i = np.unravel_index(ad.argmax(), ad.shape)
            if it < 4 or it % 8 == 7:
                print(f"iteration {it + 1:2d}  max abs difference {ad.max():.3e} at ({j}, {i})  reference {pe[j, i]:.7e}  folded {pf[j, i]:.7e}  rhs {dd[j, i]:.3e}")


if __name__ == "__main__":
    main()
