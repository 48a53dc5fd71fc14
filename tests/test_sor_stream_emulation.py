"""CPU suite: the PRODUCT's fused SOR pipeline (csrc/sor_stream_core.h, the header the GPU kernel
is compiled from) executed lane-by-lane by tests/cpp/sor_stream_emu.cpp and compared with the
oracle bit for bit.  Pipeline registers, the LDS ring and every clamped load are NaN-poisoned
in the emulator, so a stale or out-of-tile read cannot go unnoticed."""
import ctypes as C
import fcntl
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_bit_equal

OMEGA = np.float32(1.96)
_F = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def emu():
    d = os.path.join(ROOT, "tests", "cpp")
    so = os.environ.get("SFL_EMU_LIB")
    if not so:   # (make every time: a library older than sor_stream_core.h would test yesterday's pipeline)
        so = os.path.join(d, "libsor_stream_emu.so")
        with open(os.path.join(d, ".emu_build.lock"), "w") as lock:   # (pytest -n: one worker builds, the others wait)
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.run(["make", "-C", d, "-j8", so], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)
    lib.emu_sor_fused.argtypes = [_F, _F, _F] + [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int, C.c_int]
    lib.emu_sor_fused.restype = C.c_int

    lib.emu_tiling_cover.argtypes = [C.c_int] * 9 + [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.emu_tiling_cover.restype = C.c_int
    lib.emu_tile_order.argtypes = [C.c_int] * 13
    lib.emu_tile_order.restype = C.c_int

    def run(p_in, d, ns, *, dx=1.0, omega=OMEGA, rows=32, vec2=False, force_edge=False,
            gdim_y=None, grow0=0, g_begin=0, g_end=None, out=None, uniform=False, flip=True, fold=False):
        lrows, dim_x = d.shape
        gdim_y = lrows if gdim_y is None else gdim_y
        g_end = gdim_y if g_end is None else g_end
        out = np.full_like(d, np.nan) if out is None else out
        fp = lambda a: None if a is None else a.ctypes.data_as(_F)
        flags = (1 if vec2 else 0) | 2 | (4 if force_edge else 0) | (8 if uniform else 0) | (32 if flip == 2 else 16 if flip else 0) | (64 if fold else 0)
        rc = lib.emu_sor_fused(fp(out), fp(p_in), fp(d), dim_x, gdim_y, grow0, lrows, g_begin, g_end,
                               ns, dx, omega, rows, flags)
        assert rc >= 0
        run.flipped_tiles = rc
        return out
    run.lib = lib
    return run


SHAPES = [(2, 2), (3, 3), (5, 4), (61, 81), (64, 48), (130, 70), (300, 41), (257, 100)]


@pytest.mark.parametrize("dim_x,dim_y", SHAPES)
@pytest.mark.parametrize("ns", [2, 4, 6, 8, 12, 16])
def test_fused_passes_from_zero(emu, oracle, dim_x, dim_y, ns):
    d = np.random.default_rng(dim_x * 1000 + dim_y).standard_normal((dim_y, dim_x)).astype(np.float32)
    want = oracle.poisson_solve(d, 1.0, ns // 2, OMEGA)
    assert_bit_equal(emu(None, d, ns, rows=16), want, "auto edge")
    assert_bit_equal(emu(None, d, ns, rows=16, flip=False), want, "every tile bottom-up")
    assert_bit_equal(emu(None, d, ns, rows=40, force_edge=True), want, "forced edge path")
    assert_bit_equal(emu(None, d, ns, rows=16, uniform=True), want, "uniform tiling")
    if dim_x % 2 == 0:
        assert_bit_equal(emu(None, d, ns, rows=16, vec2=True), want, "vec2 access variant")


@pytest.mark.parametrize("dim_x,dim_y", SHAPES)
@pytest.mark.parametrize("ns", [2, 8, 10, 12, 14, 16])
def test_fused_passes_continue(emu, oracle, dim_x, dim_y, ns):
    rng = np.random.default_rng(7)
    d = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
    p0 = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
    assert_bit_equal(emu(p0, d, ns, rows=24), oracle.sor_iterate(p0, d, 1.0, ns // 2, OMEGA), "continue")
    assert_bit_equal(emu(p0, d, ns, rows=24, flip=False), oracle.sor_iterate(p0, d, 1.0, ns // 2, OMEGA),
                     "continue, every tile bottom-up")
    assert_bit_equal(emu(p0, d, ns, rows=24, dx=0.5, omega=np.float32(1.4)),
                     oracle.sor_iterate(p0, d, 0.5, ns // 2, np.float32(1.4)), "dx, omega")


def test_signed_zero_fields(emu, oracle):
    for fill in (0.0, -0.0):
        d = np.full((30, 140), fill, np.float32)
        d[5, 7] = 2.0
        assert_bit_equal(emu(None, d, 4), oracle.poisson_solve(d, 1.0, 2, OMEGA), f"fill {fill}")


@pytest.mark.parametrize("nranks", [2, 3])
@pytest.mark.parametrize("ns", [4, 8])
def test_slab_launch_matches_whole_domain(emu, oracle, nranks, ns):
    """One fused launch on a slab (local array with ghost rows, global row offset) produces the
    slab's rows of the whole-domain result, given valid input on own +- ns rows."""
    dim_x, dim_y, ghost = 150, 96, 16
    rng = np.random.default_rng(3)
    d = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
    p0 = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
    want = oracle.sor_iterate(p0, d, 1.0, ns // 2, OMEGA)
    for r in range(nranks):
        g0, g1 = dim_y * r // nranks, dim_y * (r + 1) // nranks
        grow0 = g0 - ghost
        lo, hi = max(grow0, 0), min(g1 + ghost, dim_y)
        # local arrays: NaN everywhere except the rows a halo exchange would have delivered
        def local(a, halo):
            l = np.full((g1 - g0 + 2 * ghost, dim_x), np.nan, np.float32)
            a0, a1 = max(g0 - halo, 0), min(g1 + halo, dim_y)
            l[a0 - grow0:a1 - grow0] = a[a0:a1]
            return l
        out = np.full((g1 - g0 + 2 * ghost, dim_x), np.nan, np.float32)
        emu(local(p0, ns), local(d, ns - 1), ns, rows=20, gdim_y=dim_y, grow0=grow0, g_begin=g0,
            g_end=g1, out=out)
        assert_bit_equal(out[g0 - grow0:g1 - grow0], want[g0:g1], f"slab {r}/{nranks}")


def test_tilings_partition_the_row_range(emu):
    """Property of the product's tiling arithmetic (uniform and boundary-balanced): every cell of the launch's row range is stored by exactly one tile, nothing outside
    it is touched, and balancing only ever shortens boundary tiles."""
    rng = np.random.default_rng(11)
    cases = [(16, 8192, 8192, 0, 8192, 234), (8, 8192, 8192, 1024, 2048, 38), (16, 300, 200, 0, 200, 200)]
    for _ in range(150):
        ns = int(rng.choice([2, 4, 8, 12, 16]))
        dim_x, gdim_y = int(rng.integers(2, 700)), int(rng.integers(2, 400))
        g_begin = int(rng.integers(0, gdim_y))
        g_end = int(rng.integers(g_begin + 1, gdim_y + 1))
        if rng.random() < 0.4:
            g_begin, g_end = 0, gdim_y
        cases.append((ns, dim_x, gdim_y, g_begin, g_end, int(rng.integers(1, g_end - g_begin + 1))))
    for ns, dim_x, gdim_y, g_begin, g_end, rpc in cases:
        for tile_cols, align in ((128, 2),):
            counts = {}
            for balance in (0, 10, 7, 13):
                cover = np.zeros((gdim_y, dim_x), np.int32)
                n_edge = C.c_int(0)
                n = emu.lib.emu_tiling_cover(ns, tile_cols, align, dim_x, gdim_y, g_begin, g_end, rpc, balance,
                                             cover.ctypes.data_as(C.POINTER(C.c_int)), C.byref(n_edge))
                tag = f"ns {ns} {dim_x}x{gdim_y} rows [{g_begin},{g_end}) rpc {rpc} cols {tile_cols} balance {balance}"
                assert n > 0, tag
                assert (cover[g_begin:g_end] == 1).all(), tag
                assert cover[:g_begin].sum() == 0 and cover[g_end:].sum() == 0, tag
                counts[balance] = n
            assert counts[10] >= counts[0]


def test_dispatch_order_is_a_bijection_with_the_free_tiles_first(emu):
    """A launch hands out the tiles that will wait for a halo message inside the kernel LAST (sor::tile_of_position: otherwise
    the first XCD fills up with waiting tiles and the kernel that delivers the message finds no room there).  The order is
    scheduling only -- but it must visit every tile exactly once, whatever chunk ranges the launcher marks free: the thin-slab
    and whole-grid shapes of the BASELINE configurations and random ones, with ranges that are empty, full, or partial."""
    rng = np.random.default_rng(5)
    cases = [(10, 8192, 8192, 960, 2112, 41), (16, 16384, 16384, 5984, 8352, 100), (16, 8192, 8192, 0, 8192, 234),
             (8, 300, 200, 0, 200, 30), (10, 130, 900, 100, 800, 9)]
    for _ in range(300):
        ns = int(rng.choice([2, 4, 8, 10, 12, 16]))
        dim_x, gdim_y = int(rng.integers(2, 1200)), int(rng.integers(2, 700))
        b = int(rng.integers(0, gdim_y))
        e = int(rng.integers(b + 1, gdim_y + 1))
        cases.append((ns, dim_x, gdim_y, b, e, int(rng.integers(1, e - b + 1))))
    total = 0
    for ns, dim_x, gdim_y, b, e, rpc in cases:
        for balance in (0, 10):
            for _ in range(6):
                c0, c1, e0, e1 = (int(v) for v in rng.integers(-2, 40, 4))
                n = emu.lib.emu_tile_order(ns, 128, 2, dim_x, gdim_y, b, e, rpc, balance, c0, c1, e0, e1)
                assert n > 0, (ns, dim_x, gdim_y, b, e, rpc, balance, c0, c1, e0, e1)
                total += n
    assert total > 100000


@pytest.mark.parametrize("ns", [2, 4, 8, 10, 12, 16])
@pytest.mark.parametrize("dim_x,dim_y,rows", [(420, 400, 23), (258, 300, 16), (640, 250, 9), (130, 500, 40)])
def test_alternating_stream_direction(emu, oracle, ns, dim_x, dim_y, rows):
    """Every second chunk of an inner strip is streamed TOP-DOWN (pipeline row index = -row; the S / N
    operands of a relaxation swap places so that ((W + E) + S) + N keeps its order): vertically adjacent
    tiles then read their shared halo rows at the same moment.  Tall grids with many chunks, odd and even
    tile heights: same bits as the oracle, and top-down tiles really occur."""
    rng = np.random.default_rng(ns * 13 + dim_y)
    d = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
    p0 = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
    assert_bit_equal(emu(p0, d, ns, rows=rows), oracle.sor_iterate(p0, d, 1.0, ns // 2, OMEGA), "continue")
    flipped = emu.flipped_tiles
    assert_bit_equal(emu(None, d, ns, rows=rows), oracle.poisson_solve(d, 1.0, ns // 2, OMEGA), "from zero")
    assert_bit_equal(emu(p0, d, ns, rows=rows, dx=0.5, omega=np.float32(1.4)),
                     oracle.sor_iterate(p0, d, 0.5, ns // 2, np.float32(1.4)), "dx and omega")
    if dim_x > 300:
        assert flipped > 0, "no tile was streamed top-down"
    # the odd launches of a solve on a big slab flip the EVEN chunks instead (successive launches alternate)
    assert_bit_equal(emu(p0, d, ns, rows=rows, flip=2), oracle.sor_iterate(p0, d, 1.0, ns // 2, OMEGA), "even chunks flipped")
    if dim_x > 300:
        assert emu.flipped_tiles > 0


# ---- the input class the reference's own demo produces: a quiescent field with sparse forcing ----------
def sparse_rhs(dim_x, dim_y, amplitude, touches=((0.5, 0.5),)):
    """Divergence of a zero velocity field (ino:199) after a few touch-like single-cell force writes (ino:264-269): zero except for
    the +-dipoles calculate_divergence (finitediff.cpp:29-30) makes of each written cell."""
    d = np.zeros((dim_y, dim_x), np.float32)
    for fx, fy in touches:
        i, j = int(dim_x * fx), int(dim_y * fy)
        d[j, i - 1] += np.float32(0.5 * amplitude)
        d[j, i + 1] -= np.float32(0.5 * amplitude)
        d[j - 1, i] += np.float32(0.25 * amplitude)
        d[j + 1, i] -= np.float32(0.25 * amplitude)
    return d


def solve_by_launches(emu, d, iters, ns, **kw):
    """poisson_solve as the executor issues it: launches of ns passes, the first from zero, even and odd launches flipping the
    other chunks (sor_executor.cpp / sor_fused.hip launch_variant)."""
    p, left, k = None, 2 * iters, 0
    while left > 0:
        n = min(ns, left)
        p = emu(p, d, n, rows=48, flip=1 + (k & 1), **kw)
        left -= n
        k += 1
    return p


TINY = np.float32(2.0 ** -124)   # below this a nonzero operand makes -0.25f * t round (sor_stream_core.h relax)


@pytest.mark.parametrize("dim,iters,amplitude", [(384, 80, 1.0), (384, 80, 30.0), (512, 120, 5.0)])
def test_quiescent_field_with_sparse_forcing_is_bit_exact(emu, oracle, dim, iters, amplitude):
    """A zero field with one touch-like dipole in the right-hand side, at BASELINE iteration counts (poisson.cpp:114-125 on the
    state ino:199,264-276 produce): SOR at omega = 1.96 spreads the source by ~0.49 per cell and pass, so from ~63 iterations on
    the solution's front decays THROUGH the denormal range on its way into the region that is still exactly zero.  The library's
    default arithmetic (two products, as poisson.cpp:109-111 writes them) gives the reference's bits there too."""
    d = sparse_rhs(dim, dim, amplitude)
    want = oracle.poisson_solve(d, 1.0, iters, OMEGA)
    tiny = (want != 0) & (np.abs(want) < TINY)
    assert tiny.sum() > 50, "the scenario must reach the range where the products differ"
    assert (want == 0).sum() > 1000, "... and keep a quiescent region beyond the front"
    assert_bit_equal(solve_by_launches(emu, d, iters, 16), want, "exact arithmetic, fuse 16")
    assert_bit_equal(solve_by_launches(emu, d, iters, 10), want, "exact arithmetic, fuse 10")


def test_folded_quarter_omega_differs_only_in_the_decaying_front(emu, oracle):
    """SFL_OPT_SOR_FOLD = 1 (opt-in): (1 - omega) * p + (-0.25f * omega) * t.  On the sparse scenario it is NOT the reference's bits --
    this test fails if the folded arithmetic is made the default without anybody noticing -- and what it costs is bounded as the
    documentation says (include/sfl.h SFL_OPT_SOR_FOLD): only cells of the front differ, by absolute amounts that are denormal-sized;
    on a dense right-hand side the two arithmetics are the same bits."""
    d = sparse_rhs(384, 384, 1.0)
    want = oracle.poisson_solve(d, 1.0, 80, OMEGA)
    got = solve_by_launches(emu, d, 80, 16, fold=True)
    differs = bits_differ(got, want)
    assert differs.any(), "the folded product should differ from the reference on this input"
    assert np.abs(want[differs]).max() < 2.0 ** -110, "only the decaying front may differ"
    assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() < 1e-40
    dense = sparse_rhs(384, 384, 1.0) + np.float32(1e-3) * np.random.default_rng(5).standard_normal((384, 384)).astype(np.float32)
    assert_bit_equal(solve_by_launches(emu, dense, 80, 16, fold=True), oracle.poisson_solve(dense, 1.0, 80, OMEGA),
                     "folded arithmetic on a dense right-hand side")


def bits_differ(a, b):
    return np.ascontiguousarray(a).view(np.uint32) != np.ascontiguousarray(b).view(np.uint32)
