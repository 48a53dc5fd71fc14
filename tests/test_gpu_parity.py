"""GPU parity tests proper: the HIP path (through the C ABI) against the reference-produced
golden fixtures and against the oracle on seeded inputs.  Bit-exact everywhere: the kernels are
built with -ffp-contract=off, so the 1e-5 relative tolerance north_star allows for fp32 fields is
met with 0 ulp; UQ32 / index / interpolation arithmetic must be bit-exact by contract."""
import os

import numpy as np
import pytest

import golden_cases as G
from conftest import ROOT, assert_bit_equal, random_fields

pytestmark = pytest.mark.gpu

DT = np.float32(1 / 30.0)
OMEGA = np.float32(1.96)


def plan_exchanges(sfl, dim_y, nranks, iters, fuse, halo=0, kernel=3):
    """Halo exchanges of one fused-kernel solve as the library plans it with an automatic (0) or given halo:
    counted on the program sfl_plan_poisson returns (csrc/sor_executor.cpp legacy_halo restated; the automatic depth is a timed choice: pass the solve's own).  kernel 3 = exchanges in
    time (the default executor), 2 = early exchanges where the halo is deep enough."""
    rows = min(b - a for a, b in (sfl.slab_rows(dim_y, nranks, r) for r in range(nranks)))
    h = halo or (64 if rows >= 1024 else 32)      # (callers pass last_solve_info()["halo"]: the automatic depth is a measured choice)
    h = max(min(h, rows, 160), fuse)
    return sum(st.kind == sfl.capi.STEP_EXCHANGE for st in sfl.plan_poisson(dim_y, nranks, 0, iters, fuse, kernel, h))


@pytest.fixture(scope="module")
def hip(sfl):
    assert sfl.device_count() >= 1, "no GPU visible: the product path has no CPU fallback"
    return sfl.HostPath()


@pytest.fixture(scope="module")
def hip_baseline(sfl):
    return sfl.HostPath(sor_kernel=1)


@pytest.mark.parametrize("path", G.ops_files(), ids=lambda p: p.split("/")[-1])
def test_ops_match_golden(hip, path):
    G.check_ops(hip, path)


@pytest.mark.parametrize("path", G.ops_files(), ids=lambda p: p.split("/")[-1])
def test_ops_match_golden_baseline_sor(hip_baseline, path):
    G.check_ops(hip_baseline, path)


@pytest.mark.parametrize("path", G.step_files(), ids=lambda p: p.split("/")[-1])
def test_steps_match_golden(hip, path):
    G.check_steps(hip, path)


@pytest.mark.parametrize("case", list(G.KAT), ids=lambda c: f"{c[0]}x{c[1]}_i{c[2]}")
def test_known_answer_hashes(hip, oracle, case):
    """SURVEY.md 8(c) hash table, computed from the GPU results."""
    dim_x, dim_y, iters, vamp, seed = case
    _, _, rows = G.KAT[case]
    h = lambda a: "%016x" % oracle.fnv1a64(a)  # hashing only
    v, c = G.lcg_fields(dim_x, dim_y, seed, vamp)
    for want in rows:
        v, d, p, c = hip.step(v, c, DT, 1.0, iters, OMEGA)
        assert (h(v), h(d), h(p), h(c)) == want


@pytest.mark.parametrize("path", G.channel_files(), ids=lambda p: p.split("/")[-1])
def test_generic_advect_matches_golden(hip, path):
    """advect<T, float> for T = float, UQ32, Vector2<UQ32>, Vector3<float> against fixtures the reference wrote."""
    G.check_channels(hip, path)


@pytest.mark.parametrize("dim_x,dim_y", [(2, 2), (3, 7), (64, 4), (65, 5), (257, 100), (1000, 333), (2048, 1030)])
@pytest.mark.parametrize("channels,uq", [(1, False), (1, True), (2, False), (2, True), (3, False), (3, True)])
def test_generic_advect_every_element_type_vs_oracle(hip, oracle, dim_x, dim_y, channels, uq):
    """Every element type the reference's headers can express (advect.h:74-85 over vector.h / uq32.h), slow and fast
    velocity fields, both wall rules; the sketch's two types take the tile kernels through the same entry point."""
    rng = np.random.default_rng(dim_x * 3 + dim_y + 10 * channels + uq)
    shape = (dim_y, dim_x) if channels == 1 else (dim_y, dim_x, channels)
    for vamp in (0.0, 40.0, 900.0):
        v = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * vamp).astype(np.float32)
        q = rng.integers(0, 2 ** 31, shape, dtype=np.uint32) if uq else (rng.standard_normal(shape) * 20).astype(np.float32)
        for no_slip in (True, False):
            assert_bit_equal(hip.advect_channels(q, v, DT, no_slip), oracle.advect_channels(q, v, DT, no_slip),
                             f"{channels} x {'uq32' if uq else 'f32'}, vamp {vamp}, no_slip {no_slip}")


def test_generic_advect_rejects_what_it_cannot_be(sfl, hip):
    v = np.zeros((4, 4, 2), np.float32)
    with pytest.raises(sfl.SflError):
        hip.advect_channels(np.zeros((4, 4, 4), np.float32), v, DT, True)       # four channels: no such reference type
    with pytest.raises(TypeError):
        hip.advect_channels(np.zeros((4, 4), np.float64), v, DT, True)


def test_external_device_field_is_advected_with_the_resident_velocity(sfl, oracle):
    """sfl_advect_external: a scalar of the caller's, resident on the device (here: the context's own divergence /
    pressure buffers stand in for it), carried by the context's velocity; whole-domain contexts only."""
    dim_x, dim_y = 200, 120
    v, _, t = random_fields(dim_x, dim_y, 21, 70.0)
    with sfl.Solver(dim_x, dim_y) as s:
        s.upload(sfl.capi.FIELD_VELOCITY, v)
        s.upload(sfl.capi.FIELD_DIVERGENCE, t)
        s.upload(sfl.capi.FIELD_PRESSURE, np.zeros_like(t))
        src, dst = s.device_ptr(sfl.capi.FIELD_DIVERGENCE), s.device_ptr(sfl.capi.FIELD_PRESSURE)
        s.advect_external(dst, src, 1, sfl.capi.CHANNEL_F32, DT, False)
        s.synchronize()
        assert_bit_equal(s.download(sfl.capi.FIELD_PRESSURE), oracle.advect_channels(t, v, DT, False), "external scalar")
        with pytest.raises(sfl.SflError):
            s.advect_external(src, src, 1, sfl.capi.CHANNEL_F32, DT, False)     # next_p == p
    with sfl.Solver(dim_x, dim_y, rank=0, nranks=2) as slab:
        with pytest.raises(sfl.SflError):
            slab.advect_external(1, 2, 1, 0, DT, False)


SHAPES = [(2, 2), (2, 7), (7, 2), (3, 3), (4, 5), (61, 81), (64, 48), (127, 33), (128, 64),
          (130, 70), (257, 100), (300, 41), (512, 96)]


@pytest.mark.parametrize("dim_x,dim_y", SHAPES)
def test_sor_fused_vs_oracle(sfl, oracle, dim_x, dim_y):
    _, _, d = random_fields(dim_x, dim_y, 3)
    for fuse, iters, rows in [(2, 1, 0), (4, 3, 16), (8, 4, 0), (8, 9, 24), (16, 8, 0), (6, 7, 0)]:
        hp = sfl.HostPath(sor_kernel=2, sor_fuse=fuse, sor_rows=rows)
        assert_bit_equal(hp.poisson_solve(d, 1.0, iters, OMEGA),
                         oracle.poisson_solve(d, 1.0, iters, OMEGA),
                         f"{dim_x}x{dim_y} fuse {fuse} iters {iters}")
    hp = sfl.HostPath(sor_kernel=2, sor_fuse=8)
    assert_bit_equal(hp.poisson_solve(d, 0.5, 5, np.float32(1.3)),
                     oracle.poisson_solve(d, 0.5, 5, np.float32(1.3)), "dx 0.5 omega 1.3")


@pytest.mark.parametrize("dim_x,dim_y", SHAPES)
def test_operators_vs_oracle(hip, oracle, dim_x, dim_y):
    for seed, vamp in [(1, 100.0), (2, 1000.0), (3, 0.0)]:
        v, c, s = random_fields(dim_x, dim_y, seed, vamp)
        for ns in (True, False):
            assert_bit_equal(hip.advect_vec2f(v, v, DT, ns), oracle.advect_vec2f(v, v, DT, ns), "adv2")
            assert_bit_equal(hip.advect_vec3uq32(c, v, DT, ns), oracle.advect_vec3uq32(c, v, DT, ns),
                             "adv3")
        assert_bit_equal(hip.divergence(v, 1.0), oracle.divergence(v, 1.0), "div")
        assert_bit_equal(hip.subtract_gradient(v, s, 1.0), oracle.subtract_gradient(v, s, 1.0), "grad")


def test_signed_zero_and_zero_rhs(sfl, oracle):
    for fill in (0.0, -0.0):
        d = np.full((40, 200), fill, np.float32)
        d[3, 4] = 1.0
        for k, f in ((1, 0), (2, 4), (2, 8)):
            hp = sfl.HostPath(sor_kernel=k, sor_fuse=f)
            assert_bit_equal(hp.poisson_solve(d, 1.0, 4, OMEGA), oracle.poisson_solve(d, 1.0, 4, OMEGA),
                             f"fill {fill} kernel {k}")


def test_folded_quarter_omega_at_the_edges_of_float(sfl, oracle):
    """poisson.cpp:109-111 multiplies by -0.25f and then by omega; so does the fused kernel (csrc/sor_stream_core.h relax), and
    is therefore bit-exact down to the last denormal.  SFL_OPT_SOR_FOLD = 1 multiplies once, by -0.25f * omega: the same bits
    unless a quarter underflows INEXACTLY -- an omega below 2^-124 is solved unfolded whatever the option says; a residual
    t = dx * d - sum has bits below 2^-147 only if one of its operands is a nonzero number below 2^-124 (4.7e-38), then -0.25f * t
    is rounded once instead of twice, one unit of 2^-149 apart, which later passes amplify like any perturbation.  Folded, fields
    that live at the very bottom of the float range agree within north_star's tolerance (1e-5 of the field's maximum) instead
    of bit for bit; everything above stays bit-exact."""
    _, _, d = random_fields(300, 150, 21)
    for fold in (0, 1):
        for omega in (np.float32(1e-39), np.float32(3e-38), np.float32(2.0 ** -124), np.float32(0.0), np.float32(-1.5)):
            hp = sfl.HostPath(sor_kernel=2, sor_fuse=8, sor_fold=fold)
            assert_bit_equal(hp.poisson_solve(d, 1.0, 5, omega), oracle.poisson_solve(d, 1.0, 5, omega), f"omega {omega} fold {fold}")
        for exp in (-60, -100):   # (no operand below 2^-124: a Gaussian sample is not 2^-24 small)
            small = (d * np.float32(2.0 ** exp)).astype(np.float32)
            assert_bit_equal(sfl.HostPath(sor_kernel=2, sor_fuse=8, sor_fold=fold).poisson_solve(small, 1.0, 6, OMEGA),
                             oracle.poisson_solve(small, 1.0, 6, OMEGA), f"right-hand side scaled by 2^{exp}, fold {fold}")
    for exp in (-118, -122, -126, -140):
        tiny = (d * np.float32(2.0 ** exp)).astype(np.float32)
        want = oracle.poisson_solve(tiny, 1.0, 6, OMEGA)
        # the default arithmetic, fused and one pass per launch: the reference's bits
        assert_bit_equal(sfl.HostPath(sor_kernel=2, sor_fuse=8).poisson_solve(tiny, 1.0, 6, OMEGA), want, f"fused kernel, 2^{exp}")
        assert_bit_equal(sfl.HostPath(sor_kernel=2, sor_fuse=16).poisson_solve(tiny, 1.0, 6, OMEGA), want, f"fused kernel, 2^{exp}")
        assert_bit_equal(sfl.HostPath(sor_kernel=1).poisson_solve(tiny, 1.0, 6, OMEGA), want, f"one-pass kernel, 2^{exp}")
        got = sfl.HostPath(sor_kernel=2, sor_fuse=8, sor_fold=1).poisson_solve(tiny, 1.0, 6, OMEGA).astype(np.float64)
        apart = np.max(np.abs(got - want.astype(np.float64)))
        # 1e-5 relative (north_star); where the whole field is denormal a unit of 2^-149 is already more than that: a few units
        assert apart <= max(1e-5 * np.max(np.abs(want)), 64 * 2.0 ** -149), f"2^{exp}: {apart / 2.0 ** -149} units of 2^-149 apart"


# ---- the input class the reference's own demo produces: a quiescent field, sparse forcing, BASELINE iteration counts --
def touch_dipoles(dim_x, dim_y, amplitude, touches):
    """calculate_divergence (finitediff.cpp:29-30) of a zero velocity field (ino:199) after single-cell force writes (ino:264-269)."""
    d = np.zeros((dim_y, dim_x), np.float32)
    for fx, fy in touches:
        i, j = int(dim_x * fx), int(dim_y * fy)
        d[j, i - 1] += np.float32(0.5 * amplitude)
        d[j, i + 1] -= np.float32(0.5 * amplitude)
        d[j - 1, i] += np.float32(0.25 * amplitude)
        d[j + 1, i] -= np.float32(0.25 * amplitude)
    return d


def differing(a, b):
    return np.ascontiguousarray(a).view(np.uint32) != np.ascontiguousarray(b).view(np.uint32)


TINY = 2.0 ** -124
_SPARSE_HEADLINE = {}


@pytest.mark.parametrize("fold", [0, 1])
def test_quiescent_sparse_right_hand_side_at_the_headline_size(sfl, oracle, fold):
    """poisson_solve (poisson.cpp:114-125) at BASELINE configuration 3's size and iteration count, 8192^2 x 80, on the right-hand
    side the sketch's own start produces: zero (ino:199) but for a few touch dipoles (ino:264-276).  SOR at omega = 1.96 carries
    a point source ~0.49 per cell and pass, so the solution's front -- 160 cells out after 80 iterations -- decays through the
    denormals into cells that are still exactly zero.  Default arithmetic: every one of the 67 M cells has the reference's bits.
    SFL_OPT_SOR_FOLD = 1: only cells of that front differ, by denormal-sized amounts (the bound include/sfl.h states)."""
    dim, iters = 8192, 80
    d = touch_dipoles(dim, dim, 20.0, [(0.5, 0.5), (0.12, 0.8), (0.97, 0.03), (0.6, 0.31)])
    if "want" not in _SPARSE_HEADLINE:   # (one oracle solve for both arithmetics: ~5 s of a host core)
        _SPARSE_HEADLINE["want"] = oracle.poisson_solve(d, 1.0, iters, OMEGA)
    want = _SPARSE_HEADLINE["want"]
    front = (want != 0) & (np.abs(want) < TINY)
    assert front.sum() > 500 and (want == 0).mean() > 0.9, "the scenario must have a denormal front and a quiescent region"
    with sfl.Solver(dim, dim) as s:
        s.set_option(sfl.capi.OPT_SOR_FOLD, fold)
        s.upload(sfl.capi.FIELD_DIVERGENCE, d)
        s.poisson_solve(1.0, iters, OMEGA)
        s.synchronize()
        assert s.last_solve_info()["fuse"] == 16
        got = s.download(sfl.capi.FIELD_PRESSURE)
    if not fold:
        assert_bit_equal(got, want, "8192^2 x 80 on a sparse right-hand side")
        return
    diff = differing(got, want)
    assert diff.any(), "the folded product is expected to differ from the reference on this input (else: drop the option's caveat)"
    assert np.abs(want[diff]).max() < 2.0 ** -100, "only cells of the decaying front may differ"
    assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= 1e-38


@pytest.mark.parametrize("fold", [0, 1])
@pytest.mark.parametrize("nranks", [1, 4])
def test_quiescent_field_with_drags_over_whole_sim_steps(sfl, oracle, nranks, fold):
    """The sketch's scenario end to end (ino:199, 249-289): velocity zero, a handful of drag messages (sfl_queue_drags), then
    sim steps at 80 SOR iterations on 1536 x 1024 -- three steps, so that the projected velocity of one step, which carries the
    pressure front's denormals over most of the domain, is advected, differenced and solved for again in the next.  Whole domain
    (sfl_step_n: seam kernels) and four virtual ranks.  Default arithmetic: velocity, divergence, pressure and dye bit for bit.
    SFL_OPT_SOR_FOLD = 1 is NOT the reference's bits here, and not only in the denormal front: a field that holds every magnitude
    between 2^-149 and 1 hands a one-unit difference at the bottom up the scales (a difference of 2^-149 flips the rounding of a
    value 64 times larger once in 64 relaxations, and so on), so that after three steps pressure and velocity differ from the
    reference by ordinary rounding noise -- measured 7.5e-9 absolute on a field of maximum 0.86, one to a few units in the last
    place of the cells concerned (profiles/r06_numerics.txt).  The bound asserted: 1e-5 of each field's maximum (north_star's
    tolerance read against the field's scale); the dye within one raw unit of UQ32."""
    dim_x, dim_y, iters, steps = 1536, 1024, 80, 3
    _, c, _ = random_fields(dim_x, dim_y, 77, 0.0)
    v = np.zeros((dim_y, dim_x, 2), np.float32)
    drags = [(512, 700, 35.0, -20.0), (513, 700, 30.0, -25.0), (100, 90, -60.0, 12.0), (900, 1400, 8.0, 90.0), (1023, 1535, 5.0, 5.0)]
    forces = ([(y, x) for x, y, _, _ in drags], [(vy, vx) for _, _, vx, vy in drags])
    vo, co = v, c
    for k in range(steps):
        vo, do, po, co = oracle_step(oracle, vo, co, iters, forces if k == 0 else None)
    assert ((po != 0) & (np.abs(po) < TINY)).sum() > 100, "the scenario must reach the denormal range"
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        if nranks > 1:
            sfl.Solver.link_group(slabs)
        for s in slabs:
            s.set_option(sfl.capi.OPT_SOR_FOLD, fold)
            s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
            s.upload(sfl.capi.FIELD_COLOR, c[s.row_begin:s.row_end])
        slabs[0].queue_drags(drags)
        slabs[0].step_n(steps, DT, 1.0, iters, OMEGA)
        slabs[0].synchronize()
        cat = lambda f: np.concatenate([s.download(f) for s in slabs], axis=0)
        gv, gd, gp, gc = (cat(f) for f in (sfl.capi.FIELD_VELOCITY, sfl.capi.FIELD_DIVERGENCE, sfl.capi.FIELD_PRESSURE,
                                           sfl.capi.FIELD_COLOR))
    finally:
        for s in slabs:
            s.close()
    if not fold:
        for name, a, b in zip(("velocity", "divergence", "pressure", "dye"), (gv, gd, gp, gc), (vo, do, po, co)):
            assert_bit_equal(a, b, f"{name}, {nranks} rank(s)")
        return
    # (the dye: the same bits after these three steps, 3 of 1.5 M cells one raw unit apart after a fourth -- tests/fold_numerics_probe.py)
    assert np.abs(gc.astype(np.int64) - co.astype(np.int64)).max() <= 1, "dye"
    assert differing(gp, po).any(), "the folded product is expected to differ from the reference on this input"
    for name, a, b in zip(("velocity", "divergence", "pressure"), (gv, gd, gp), (vo, do, po)):
        assert np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= 1e-5 * np.abs(b).max(), name


def test_iters_zero_gives_zero_pressure(hip):
    d = np.ones((9, 12), np.float32)
    p = hip.poisson_solve(d, 1.0, 0, OMEGA)
    assert_bit_equal(p, np.zeros_like(d), "iters = 0")


def test_rejects_degenerate_dims(sfl):
    with pytest.raises(sfl.SflError):
        sfl.HostPath().poisson_solve(np.zeros((1, 8), np.float32), 1.0, 1, OMEGA)
    with pytest.raises(sfl.SflError):
        sfl.HostPath().divergence(np.zeros((8, 1, 2), np.float32), 1.0)


def test_forces_are_injected_between_advect_and_divergence(sfl, oracle):
    dim_x, dim_y, iters = 61, 81, 10
    v, c, _ = random_fields(dim_x, dim_y, 5, 40.0)
    cells = np.array([[10, 20], [30, 40], [10, 20]], np.int32)      # last write wins
    vel = np.array([[5.0, -3.0], [1.5, 2.5], [-7.0, 9.0]], np.float32)
    with sfl.Solver(dim_x, dim_y) as s:
        s.upload(sfl.capi.FIELD_VELOCITY, v)
        s.upload(sfl.capi.FIELD_COLOR, c)
        s.queue_forces(cells, vel)
        s.step(DT, 1.0, iters, OMEGA)
        s.synchronize()
        got_v, got_c = s.download(sfl.capi.FIELD_VELOCITY), s.download(sfl.capi.FIELD_COLOR)
    va = oracle.advect_vec2f(v, v, DT, True)
    for (i, j), u in zip(cells, vel):
        va[j, i] = u
    d = oracle.divergence(va, 1.0)
    p = oracle.poisson_solve(d, 1.0, iters, OMEGA)
    want_v = oracle.subtract_gradient(va, p, 1.0)
    assert_bit_equal(got_v, want_v, "velocity with forces")
    assert_bit_equal(got_c, oracle.advect_vec3uq32(c, want_v, DT, False), "colour with forces")


@pytest.mark.parametrize("nranks", [2, 3, 4])
@pytest.mark.parametrize("kernel,fuse,halo", [(1, 2, 0), (2, 4, 4), (2, 8, 0), (2, 8, 32), (2, 16, 32),
                                              (2, 6, 20)])
def test_virtual_slabs_match_single_context(sfl, oracle, nranks, kernel, fuse, halo):
    """Row-slab decomposition on ONE device (in-process halo copies instead of RCCL): the slab
    executor, halo bookkeeping and slab-aware kernels must reproduce the whole-domain result
    bit for bit."""
    dim_x, dim_y, iters = 96, 160, 7
    v, c, _ = random_fields(dim_x, dim_y, 21, 60.0)
    want = oracle.step(v, c, DT, 1.0, iters, OMEGA)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        for s in slabs:
            s.set_option(sfl.capi.OPT_SOR_KERNEL, kernel)
            s.set_option(sfl.capi.OPT_SOR_FUSE, fuse)
            s.set_option(sfl.capi.OPT_SOR_HALO, halo)
            s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
            s.upload(sfl.capi.FIELD_COLOR, c[s.row_begin:s.row_end])
        slabs[0].step(DT, 1.0, iters, OMEGA)     # collective over the linked group
        slabs[0].synchronize()
        cat = lambda f: np.concatenate([s.download(f) for s in slabs], axis=0)
        got = (cat(sfl.capi.FIELD_VELOCITY), cat(sfl.capi.FIELD_DIVERGENCE),
               cat(sfl.capi.FIELD_PRESSURE), cat(sfl.capi.FIELD_COLOR))
    finally:
        for s in slabs:
            s.close()
    for name, a, b in zip(("v", "div", "p", "colour"), got, want):
        assert_bit_equal(a, b, f"{nranks} slabs: {name}")


def test_slab_advect_halo_overflow_is_reported(sfl):
    dim_x, dim_y = 64, 128
    v = np.zeros((dim_y, dim_x, 2), np.float32)
    v[..., 1] = 30.0 * 20      # back-trace of 20 rows > 4-row halo
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, 2) for r in range(2)]
    try:
        sfl.Solver.link_group(slabs)
        assert slabs[1].get_option(sfl.capi.OPT_ADVECT_HALO) == 0      # the default is the automatic halo
        slabs[0].set_option(sfl.capi.OPT_ADVECT_HALO, 4)               # a FIXED halo reports what leaves it
        for s in slabs:
            s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
        slabs[0].advect_velocity(DT, True)
        with pytest.raises(sfl.SflError) as e:
            slabs[0].synchronize()
        assert e.value.code == sfl.capi.ERR_HALO
    finally:
        for s in slabs:
            s.close()


def test_rccl_single_rank_communicator(sfl, oracle):
    """RCCL bring-up on the one GPU available to tests: a 1-rank communicator attaches and a
    solve runs through the RCCL-transport code path (no neighbours => no sends)."""
    dim_x, dim_y = 64, 64
    _, _, d = random_fields(dim_x, dim_y, 8)
    with sfl.Solver(dim_x, dim_y) as s:
        s.comm_attach(sfl.comm_unique_id())
        s.upload(sfl.capi.FIELD_DIVERGENCE, d)
        s.poisson_solve(1.0, 6, OMEGA)
        s.synchronize()
        assert_bit_equal(s.download(sfl.capi.FIELD_PRESSURE), oracle.poisson_solve(d, 1.0, 6, OMEGA),
                         "rccl 1-rank")
        # real ncclSend / ncclRecv on the solver's stream (to itself: one GPU cannot host 2 ranks)
        s.upload(sfl.capi.FIELD_PRESSURE, np.zeros_like(d))
        s.comm_loopback(17)
        s.synchronize()
        got = s.download(sfl.capi.FIELD_PRESSURE)
        assert_bit_equal(got[:17], d[:17], "loopback rows")
        assert not got[17:].any()
        # the option block travels through a real ncclAllGather (attach did it once already): still in agreement
        s.set_option(sfl.capi.OPT_SOR_FUSE, 6)
        s.comm_check_options()
        assert s.get_option(sfl.capi.OPT_TRANSPORT) == 1


def bench_rhs(sfl, size, dim_y=None):
    """The right-hand side bench.py solves for: divergence of its seeded velocity field."""
    import bench
    dim_y = dim_y or size
    with sfl.Solver(size, dim_y) as s:
        s.upload(sfl.capi.FIELD_VELOCITY, bench.synthetic_velocity(size, 0, dim_y))
        s.calculate_divergence(1.0)
        s.synchronize()
        return s.download(sfl.capi.FIELD_DIVERGENCE)


def test_headline_size_spot_check_vs_oracle(sfl, oracle):
    """8192 x 8192 (BASELINE config 3 grid), 16 iterations = 32 colour passes: two launches of the
    NS = 16 kernel (the second continues from a given p: the instantiation bench.py times) and two
    shallower depths, every cell against the oracle (SURVEY 8d parity gate)."""
    n, iters = 8192, 16
    rng = np.random.default_rng(2026)
    d = (rng.standard_normal((n, n)) * 0.05).astype(np.float32)
    want = oracle.poisson_solve(d, 1.0, iters, OMEGA)
    for lane, fuse in ((2, 16), (2, 12), (0, 8)):
        with sfl.Solver(n, n) as s:
            s.set_option(sfl.capi.OPT_SOR_KERNEL, 2)
            s.set_option(sfl.capi.OPT_SOR_FUSE, fuse)
            s.set_option(sfl.capi.OPT_SOR_LANE_CELLS, lane)
            s.upload(sfl.capi.FIELD_DIVERGENCE, d)
            s.poisson_solve(1.0, iters, OMEGA)
            s.synchronize()
            info = s.last_solve_info()
            got = s.download(sfl.capi.FIELD_PRESSURE)
        assert info["fuse"] == fuse and info["launches"] == -(-2 * iters // fuse)
        assert_bit_equal(got, want, f"8192^2 lane{lane} fuse{fuse}")


def test_baseline_config3_exactly_as_benchmarked(sfl, oracle):
    """BASELINE config 3 as bench.py times it: 8192 x 8192, 80 iterations, omega 1.96, dx 1, rhs =
    divergence of the bench's seeded velocity, EVERY option on auto -- which must resolve to ten
    launches of the NS = 16 kernel.  Every cell against the oracle (poisson.cpp:114-125; the oracle
    needs ~10 s on one core).  Tolerance north_star allows: 1e-5 relative; asserted: 0 ulp."""
    n, iters = 8192, 80
    d = bench_rhs(sfl, n)
    with sfl.Solver(n, n) as s:
        s.upload(sfl.capi.FIELD_DIVERGENCE, d)
        s.poisson_solve(1.0, iters, OMEGA)
        s.synchronize()
        info = s.last_solve_info()
        s.poisson_solve(1.0, iters, OMEGA)      # a second solve on the same context restarts from zero
        s.synchronize()
        got = s.download(sfl.capi.FIELD_PRESSURE)
    assert info["fuse"] == 16 and info["launches"] == 10 and info["exchanges"] == 0
    assert_bit_equal(got, oracle.poisson_solve(d, 1.0, iters, OMEGA), "C3: 8192^2 x 80 iterations, auto")


def test_baseline_config2_vs_oracle(sfl, oracle):
    """BASELINE config 2: 2048 x 2048, 40 SOR iterations per step, one GPU, every option on auto --
    the stand-alone solve on the bench's rhs and one whole sim step, every cell of every field."""
    import bench
    n, iters = 2048, 40
    d = bench_rhs(sfl, n)
    hp = sfl.HostPath()
    assert_bit_equal(hp.poisson_solve(d, 1.0, iters, OMEGA), oracle.poisson_solve(d, 1.0, iters, OMEGA),
                     "C2: 2048^2 x 40 iterations, auto")
    v, c = bench.synthetic_velocity(n, 0, n), bench.synthetic_color(n, 0, n)
    got = hp.step(v, c, DT, 1.0, iters, OMEGA)
    want = oracle.step(v, c, DT, 1.0, iters, OMEGA)
    for name, a, b in zip(("v", "div", "p", "colour"), got, want):
        assert_bit_equal(a, b, f"C2 step: {name}")


def test_large_grid_properties(sfl):
    """16384 x 16384 (BASELINE config 5 grid): size-independent properties instead of an oracle
    run -- (1) the fused kernel (any fuse depth) and the one-launch-per-pass
    baseline kernel are different programs that must agree bit for bit; (2) a zero right-hand
    side keeps the pressure at (signed) zero; (3) the solve is deterministic."""
    n, iters = 16384, 3
    with sfl.Solver(n, n) as s:
        rng = np.random.default_rng(7)
        d = (rng.standard_normal((n, n)) * 0.1).astype(np.float32)
        s.upload(sfl.capi.FIELD_DIVERGENCE, d)
        results = []
        for kernel, fuse, lane in ((1, 0, 0), (2, 6, 2), (2, 16, 2), (2, 4, 0), (2, 6, 2)):
            s.set_option(sfl.capi.OPT_SOR_KERNEL, kernel)
            if fuse:
                s.set_option(sfl.capi.OPT_SOR_FUSE, fuse)
            s.set_option(sfl.capi.OPT_SOR_LANE_CELLS, lane)
            s.poisson_solve(1.0, iters, OMEGA)
            s.synchronize()
            results.append(s.download(sfl.capi.FIELD_PRESSURE))
        for k, r in enumerate(results[1:], 1):
            assert_bit_equal(r, results[0], f"16384^2 variant {k} vs baseline kernel")
        assert np.isfinite(results[0]).all() and np.abs(results[0]).max() > 0
        s.upload(sfl.capi.FIELD_DIVERGENCE, np.zeros((n, n), np.float32))
        s.poisson_solve(1.0, iters, OMEGA)
        s.synchronize()
        assert not np.abs(s.download(sfl.capi.FIELD_PRESSURE)).any()


@pytest.mark.parametrize("dim_x,dim_y,scaling", [(61, 81, 4), (5, 4, 4), (33, 17, 2), (40, 50, 3), (2, 2, 8)])
def test_dye_visualiser_matches_oracle(sfl, oracle, dim_x, dim_y, scaling):
    """draw-task arithmetic (ino:116-176) as a HIP kernel, bit-exact against the
    oracle's restatement (itself unpinned: the .ino cannot be compiled here)."""
    _, c, _ = random_fields(dim_x, dim_y, 12)
    # use the top bits the 565 pack keeps, stay below the float -> uint32 UB range
    c = np.minimum(c.astype(np.uint64) * 2, 0xFE000000).astype(np.uint32)
    with sfl.Solver(dim_x, dim_y) as s:
        s.upload(sfl.capi.FIELD_COLOR, c)
        for swap in (True, False):
            got = s.render_rgb565(scaling, swap)
            want = oracle.render_rgb565(c, scaling, swap)
            assert got.shape == (scaling * (dim_x - 1), scaling * (dim_y - 1))
            assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} pixels differ"


def test_api_error_paths(sfl):
    """Status codes instead of crashes: options, sizes, states (the reference checks nothing)."""
    cap = sfl.capi
    with sfl.Solver(32, 16) as s:
        for opt, bad in ((cap.OPT_SOR_KERNEL, 3), (cap.OPT_SOR_FUSE, 7), (cap.OPT_SOR_FUSE, 18),
                         (cap.OPT_SOR_LANE_CELLS, 3), (cap.OPT_SOR_LANE_CELLS, 4), (cap.OPT_ADVECT_HALO, -1), (cap.OPT_ADVECT_HALO, 65), (cap.OPT_SOR_HALO, 1),
                         (99, 0)):
            with pytest.raises(sfl.SflError) as e:
                s.set_option(opt, bad)
            assert e.value.code == cap.ERR_INVALID
        with pytest.raises(ValueError):
            s.upload(cap.FIELD_PRESSURE, np.zeros((16, 31), np.float32))
        with pytest.raises(sfl.SflError) as e:
            s.poisson_solve(1.0, -1, OMEGA)
        assert e.value.code == cap.ERR_INVALID
        with pytest.raises(sfl.SflError) as e:      # no communicator on a whole-domain context
            s.comm_loopback(4)
        assert e.value.code == cap.ERR_STATE
    # a slab without a transport cannot exchange halos: clear state error, not a hang
    with sfl.Solver(32, 64, 0, 0, 2) as slab:
        slab.upload(cap.FIELD_DIVERGENCE, np.ones((32, 32), np.float32))
        with pytest.raises(sfl.SflError) as e:
            slab.poisson_solve(1.0, 4, OMEGA)
        assert e.value.code == cap.ERR_STATE
        with pytest.raises(sfl.SflError) as e:      # render needs the whole domain
            slab.render_rgb565(2)
        assert e.value.code == cap.ERR_STATE
    with pytest.raises(sfl.SflError):               # more slabs than rows
        sfl.Solver(8, 4, 0, 0, 5)


def test_two_contexts_are_independent(sfl, oracle):
    """No global mutable state: two contexts of different shapes interleave freely."""
    _, _, d1 = random_fields(64, 40, 1)
    _, _, d2 = random_fields(130, 33, 2)
    with sfl.Solver(64, 40) as a, sfl.Solver(130, 33) as b:
        a.upload(sfl.capi.FIELD_DIVERGENCE, d1)
        b.upload(sfl.capi.FIELD_DIVERGENCE, d2)
        a.poisson_solve(1.0, 5, OMEGA)
        b.poisson_solve(1.0, 9, OMEGA)
        a.synchronize(); b.synchronize()
        assert_bit_equal(b.download(sfl.capi.FIELD_PRESSURE), oracle.poisson_solve(d2, 1.0, 9, OMEGA), "b")
        assert_bit_equal(a.download(sfl.capi.FIELD_PRESSURE), oracle.poisson_solve(d1, 1.0, 5, OMEGA), "a")


def test_randomised_configurations_vs_oracle(sfl, oracle):
    """Seeded fuzz over shapes x fuse depth x rows-per-tile x lane flavour x slab count x halo
    depth x dx / omega: every combination must reproduce the oracle's poisson_solve bit for bit."""
    rng = np.random.default_rng(20261002)
    for case in range(40):
        dim_x = int(rng.integers(2, 400))
        dim_y = int(rng.integers(2, 300))
        if case % 4 == 0:
            dim_x = int(rng.choice([4, 8, 128, 256, 260, 512, 640, 1500]))
        iters = int(rng.integers(1, 14))
        fuse = int(rng.choice([2, 4, 6, 8, 10, 12, 14, 16]))
        rows = int(rng.choice([0, 0, 8, 17, 40]))
        lane = int(rng.choice([0, 2]))
        dx = float(rng.choice([1.0, 1.0, 0.5, 2.0]))
        omega = np.float32(rng.choice([1.96, 1.0, 1.7]))
        nranks = int(rng.choice([1, 1, 2, 3]))
        halo = int(rng.choice([0, 16, 32, 64]))
        d = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
        want = oracle.poisson_solve(d, dx, iters, omega)
        tag = f"case {case}: {dim_x}x{dim_y} iters {iters} fuse {fuse} rows {rows} lane {lane} dx {dx} " \
              f"omega {omega} ranks {nranks} halo {halo}"
        if nranks > 1 and min(sfl.slab_rows(dim_y, nranks, r)[1] - sfl.slab_rows(dim_y, nranks, r)[0]
                              for r in range(nranks)) < max(fuse, halo):
            nranks = 1      # slabs thinner than the halo are rejected by the API (tested elsewhere)
        slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
        try:
            if nranks > 1:
                sfl.Solver.link_group(slabs)
            for s in slabs:
                s.set_option(sfl.capi.OPT_SOR_KERNEL, 2)
                s.set_option(sfl.capi.OPT_SOR_FUSE, fuse)
                s.set_option(sfl.capi.OPT_SOR_ROWS, rows)
                s.set_option(sfl.capi.OPT_SOR_LANE_CELLS, lane)
                s.set_option(sfl.capi.OPT_SOR_HALO, halo)
                s.upload(sfl.capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
            slabs[0].poisson_solve(dx, iters, omega)
            slabs[0].synchronize()
            got = np.concatenate([s.download(sfl.capi.FIELD_PRESSURE) for s in slabs], axis=0)
        finally:
            for s in slabs:
                s.close()
        assert_bit_equal(got, want, tag)


def test_step_with_and_without_fused_projection(sfl, oracle):
    """sfl_step applies subtract_gradient inside the dye-advection kernel by default; the
    two-kernel form must give the same bits (both equal the oracle)."""
    v, c, _ = random_fields(150, 90, 31, 70.0)
    want = oracle.step(v, c, DT, 1.0, 6, OMEGA)
    for fuse in (1, 0):
        with sfl.Solver(150, 90) as s:
            s.set_option(sfl.capi.OPT_FUSE_PROJECTION, fuse)
            s.upload(sfl.capi.FIELD_VELOCITY, v)
            s.upload(sfl.capi.FIELD_COLOR, c)
            s.step(DT, 1.0, 6, OMEGA)
            s.synchronize()
            got = (s.download(sfl.capi.FIELD_VELOCITY), s.download(sfl.capi.FIELD_DIVERGENCE),
                   s.download(sfl.capi.FIELD_PRESSURE), s.download(sfl.capi.FIELD_COLOR))
        for name, a, b in zip(("v", "div", "p", "colour"), got, want):
            assert_bit_equal(a, b, f"fuse_projection={fuse}: {name}")


def test_long_run_stays_bit_identical(hip, oracle):
    """30 consecutive sim steps (the sketch's parameters: 10 iterations, omega 1.96, dt 1/30) on the
    ESP32 domain: no drift between the GPU path and the oracle, bit for bit, at every step."""
    v, c = G.lcg_fields(61, 81, 31337, 40.0)
    vo, co = v, c
    for step in range(30):
        v, d, p, c = hip.step(v, c, DT, 1.0, 10, OMEGA)
        vo, do, po, co = oracle.step(vo, co, DT, 1.0, 10, OMEGA)
        assert_bit_equal(v, vo, f"step {step} v")
        assert_bit_equal(c, co, f"step {step} colour")


@pytest.mark.parametrize("dim_x,dim_y", [(5000, 3000), (8191, 1025), (4098, 4097), (16384, 520), (130, 9000)])
def test_large_irregular_shapes_vs_oracle(sfl, oracle, dim_x, dim_y):
    """Large grids whose sides are odd / not tile multiples / very flat / very tall: full sim
    step and a deeper solve against the oracle (auto fuse depth, auto tiling, both stencil forms)."""
    rng = np.random.default_rng(dim_x * 7 + dim_y)
    v = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * 60).astype(np.float32)
    c = rng.integers(0, 2 ** 31, (dim_y, dim_x, 3), dtype=np.uint32)
    hp = sfl.HostPath()
    got = hp.step(v, c, DT, 1.0, 3, OMEGA)
    want = oracle.step(v, c, DT, 1.0, 3, OMEGA)
    for name, a, b in zip(("v", "div", "p", "colour"), got, want):
        assert_bit_equal(a, b, f"{dim_x}x{dim_y} step: {name}")
    d = want[1]
    assert_bit_equal(hp.poisson_solve(d, 1.0, 9, OMEGA), oracle.poisson_solve(d, 1.0, 9, OMEGA),
                     f"{dim_x}x{dim_y} 9-iteration solve")


@pytest.mark.parametrize("nranks", [2, 4])
def test_virtual_slabs_with_auto_settings_at_realistic_size(sfl, oracle, nranks):
    """1024 x 2048 split into 2 / 4 slabs with every option on auto (fuse 8, halo 64 resp. 32,
    supersteps of several launches, extended output ranges): the production multi-GPU schedule
    with in-process copies in place of RCCL, against the oracle."""
    dim_x, dim_y, iters = 1024, 2048, 20
    rng = np.random.default_rng(5)
    d = (rng.standard_normal((dim_y, dim_x)) * 0.1).astype(np.float32)
    want = oracle.poisson_solve(d, 1.0, iters, OMEGA)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        for s in slabs:
            s.upload(sfl.capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
        slabs[0].poisson_solve(1.0, iters, OMEGA)
        slabs[0].synchronize()
        info = slabs[0].last_solve_info()
        got = np.concatenate([s.download(sfl.capi.FIELD_PRESSURE) for s in slabs], axis=0)
        assert slabs[0].get_option(sfl.capi.OPT_TRANSPORT) == 2
    finally:
        for s in slabs:
            s.close()
    assert_bit_equal(got, want, f"{nranks} slabs, auto settings")
    assert info["fuse"] == 8 and info["launches"] == 5
    assert info["exchanges"] == plan_exchanges(sfl, dim_y, nranks, iters, 8, info["halo"]) >= 1   # the rhs once (+ p)


def test_virtual_slabs_eight_ranks_on_the_headline_grid(sfl, oracle):
    """BASELINE config 4 exactly as bench.py runs it at --gpus 8 (8192^2 in eight 1024-row slabs, 80 iterations,
    every option on auto: fuse 10, halo 64 -> 16 launches, the rhs exchange and two early p exchanges with their
    ghost-row launches on the exchange stream), executed by eight virtual ranks on one GPU, against the oracle."""
    dim, iters, nranks = 8192, 80, 8
    rng = np.random.default_rng(88)
    d = (rng.standard_normal((dim, dim)) * 0.1).astype(np.float32)
    want = oracle.poisson_solve(d, 1.0, iters, OMEGA)
    slabs = [sfl.Solver(dim, dim, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        for s in slabs:
            s.upload(sfl.capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
        slabs[0].poisson_solve(1.0, iters, OMEGA)
        slabs[0].synchronize()
        info = slabs[3].last_solve_info()
        got = np.concatenate([s.download(sfl.capi.FIELD_PRESSURE) for s in slabs], axis=0)
    finally:
        for s in slabs:
            s.close()
    assert_bit_equal(got, want, "8192^2 in 8 slabs, auto settings")
    # (the halo depth is chosen from the measured exchange: 64 rows = the rhs + two p exchanges, deeper = fewer)
    assert info["fuse"] == 10 and info["launches"] == 16 and info["halo"] >= 64
    assert 1 <= info["exchanges"] == plan_exchanges(sfl, dim, nranks, iters, 10, info["halo"]) <= 3


@pytest.mark.parametrize("dim_y,fuse", [(1600, 10), (3200, 16)])
def test_virtual_slabs_auto_fuse_depths_at_bench_width(sfl, oracle, dim_y, fuse):
    """8192-wide slabs big enough for the deeper auto fuse depths (10 from 3 M cells per slab, 16
    from 12 M): two virtual ranks, everything on auto, against the oracle."""
    dim_x, iters, nranks = 8192, 20, 2
    rng = np.random.default_rng(dim_y)
    d = (rng.standard_normal((dim_y, dim_x)) * 0.1).astype(np.float32)
    want = oracle.poisson_solve(d, 1.0, iters, OMEGA)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        for s in slabs:
            s.upload(sfl.capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
        slabs[0].poisson_solve(1.0, iters, OMEGA)
        slabs[0].synchronize()
        info = slabs[0].last_solve_info()
        got = np.concatenate([s.download(sfl.capi.FIELD_PRESSURE) for s in slabs], axis=0)
    finally:
        for s in slabs:
            s.close()
    assert_bit_equal(got, want, f"8192 x {dim_y} in 2 slabs, auto settings")
    assert info["fuse"] == fuse and info["launches"] == -(-2 * iters // fuse)


@pytest.mark.parametrize("dim_x,dim_y", [(61, 81), (2, 2), (40, 7), (257, 130)])
def test_sketch_initial_condition_matches_oracle(sfl, oracle, dim_x, dim_y):
    """setup() (ino:196-241) on the GPU -- sectors by atan2f, then the two in-place
    sequential UQ32 blurs -- bit-exact against the oracle's restatement (unpinned; saturating)."""
    v, c = oracle.setup_fields(dim_x, dim_y)
    with sfl.Solver(dim_x, dim_y) as s:
        s.upload(sfl.capi.FIELD_VELOCITY, np.ones((dim_y, dim_x, 2), np.float32))
        s.setup_sketch_fields()
        s.synchronize()
        assert_bit_equal(s.download(sfl.capi.FIELD_VELOCITY), v, "velocity")
        assert_bit_equal(s.download(sfl.capi.FIELD_COLOR), c, "dye")
    if min(dim_x, dim_y) > 8:   # sector interiors stay saturated / empty after the blurs
        assert c.max() == 0xFFFFFFFF and c.min() == 0


def test_headline_size_full_step_vs_oracle(hip, oracle):
    """One whole sim step (all five operators, ino:252-287) on the 8192 x 8192 headline grid against
    the oracle, every cell of every field (the oracle needs ~20 s for it on one core)."""
    n = 8192
    rng = np.random.default_rng(8192)
    v = (rng.uniform(-1, 1, (n, n, 2)) * 100).astype(np.float32)
    c = rng.integers(0, 2 ** 31, (n, n, 3), dtype=np.uint32)
    got = hip.step(v, c, DT, 1.0, 2, OMEGA)
    want = oracle.step(v, c, DT, 1.0, 2, OMEGA)
    for name, a, b in zip(("v", "div", "p", "colour"), got, want):
        assert_bit_equal(a, b, f"8192^2 step: {name}")


def test_host_dropins_keep_a_working_context_per_thread(sfl, oracle):
    """The host-pointer drop-ins reuse their device-side context between calls (same thread, same
    grid shape), replace it when the shape changes and free it on sfl_host_release(); results are
    those of independent calls."""
    import ctypes as C
    lib = sfl.capi.lib()
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    rng = np.random.default_rng(12)
    for dim_x, dim_y in [(61, 81), (61, 81), (40, 7), (61, 81), (130, 33)]:
        v = (rng.standard_normal((dim_y, dim_x, 2)) * 2).astype(np.float32)
        d = np.empty((dim_y, dim_x), np.float32)
        p = np.empty((dim_y, dim_x), np.float32)
        assert lib.sfl_host_calculate_divergence(fp(d), fp(v), dim_x, dim_y, C.c_float(1.0)) == 0
        assert_bit_equal(d, oracle.divergence(v, 1.0), f"divergence {dim_x}x{dim_y}")
        for iters in (3, 7):   # a second solve on the reused context starts from zero again
            assert lib.sfl_host_poisson_solve(fp(p), fp(d), dim_x, dim_y, C.c_float(1.0), iters,
                                              C.c_float(1.96)) == 0
            assert_bit_equal(p, oracle.poisson_solve(d, 1.0, iters, OMEGA), f"solve {dim_x}x{dim_y} x{iters}")
        got = v.copy()
        assert lib.sfl_host_subtract_gradient(fp(got), fp(p), dim_x, dim_y, C.c_float(1.0)) == 0
        assert_bit_equal(got, oracle.subtract_gradient(v, p, 1.0), f"gradient {dim_x}x{dim_y}")
    assert lib.sfl_host_release() == 0
    assert lib.sfl_host_release() == 0   # nothing left: still fine
    assert lib.sfl_host_calculate_divergence(fp(d), fp(v), dim_x, dim_y, C.c_float(1.0)) == 0
    assert_bit_equal(d, oracle.divergence(v, 1.0), "after release")
    assert lib.sfl_host_release() == 0


def test_group_options_are_group_wide(sfl, oracle):
    """Options of linked slabs are one set: a value given to ANY member (before or after linking) is
    what every member exchanges and trusts -- a peer can never rely on ghost rows nobody sent."""
    dim_x, dim_y, nranks = 64, 120, 3
    v = np.zeros((dim_y, dim_x, 2), np.float32)
    v[..., 1] = 30.0 * 9          # back-trace of 9 rows: beyond the default 4-row advection halo
    v[..., 0] = 7.0
    want = oracle.advect_vec2f(v, v, DT, True)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        slabs[0].set_option(sfl.capi.OPT_SOR_FUSE, 6)      # before linking: slab 0's values win
        sfl.Solver.link_group(slabs)
        slabs[2].set_option(sfl.capi.OPT_ADVECT_HALO, 12)  # after linking: any member sets all
        for s in slabs:
            assert s.get_option(sfl.capi.OPT_ADVECT_HALO) == 12
            assert s.get_option(sfl.capi.OPT_SOR_FUSE) == 6
            s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
        slabs[1].advect_velocity(DT, True)
        slabs[1].synchronize()                              # no SFL_ERR_HALO
        got = np.concatenate([s.download(sfl.capi.FIELD_VELOCITY) for s in slabs], axis=0)
    finally:
        for s in slabs:
            s.close()
    assert_bit_equal(got, want, "advection with a group-wide 12-row halo")


@pytest.mark.parametrize("nranks,dim_y,fuse,halo,iters", [(2, 400, 8, 16, 21), (3, 600, 12, 24, 30), (4, 512, 6, 6, 9),
                                                          (4, 1024, 8, 64, 40), (8, 1024, 12, 36, 40), (2, 96, 16, 32, 20)])
def test_overlapped_halo_exchange_gives_the_same_bits(sfl, oracle, nranks, dim_y, fuse, halo, iters):
    """SURVEY 8e overlap: the halo exchanges of a solve run on a second stream while the rows of the
    neighbouring launches that do not depend on them are relaxed (cut-adjacent rows first / last).
    Virtual ranks on one GPU (in-process copies on the exchange stream): overlapped and in-line
    execution must both reproduce the oracle, with the same launch / exchange counts."""
    dim_x = 640
    rng = np.random.default_rng(nranks * 100 + fuse)
    d = (rng.standard_normal((dim_y, dim_x)) * 0.1).astype(np.float32)
    want = oracle.poisson_solve(d, 1.0, iters, OMEGA)
    infos = []
    # overlapped with the halo's arrival signalled on the device (the default: cut-adjacent tiles wait inside the
    # launch), overlapped with a cross-stream event in front of the launch, in line
    # in time, counted on the device (the default); early behind cross-stream events; in line
    for schedule in (sfl.capi.SCHEDULE_IN_TIME, sfl.capi.SCHEDULE_BY_EVENT, sfl.capi.SCHEDULE_IN_LINE):
        slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
        try:
            sfl.Solver.link_group(slabs)
            slabs[0].set_option(sfl.capi.OPT_EXCHANGE_SCHEDULE, schedule)
            assert slabs[-1].get_option(sfl.capi.OPT_EXCHANGE_SCHEDULE) == schedule     # group-wide, and resolved to itself
            slabs[0].set_option(sfl.capi.OPT_SOR_FUSE, fuse)
            slabs[0].set_option(sfl.capi.OPT_SOR_HALO, halo)
            for s in slabs:
                s.upload(sfl.capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
            for _ in range(2):      # twice: the second solve starts while nothing of the first is pending
                slabs[0].poisson_solve(1.0, iters, OMEGA)
            slabs[0].synchronize()
            infos.append(slabs[nranks // 2].last_solve_info())
            got = np.concatenate([s.download(sfl.capi.FIELD_PRESSURE) for s in slabs], axis=0)
        finally:
            for s in slabs:
                s.close()
        assert_bit_equal(got, want, f"{nranks} slabs, exchange schedule {schedule}")
    # (in time and in line walk the same kernel-3 program; the early program has the same launches)
    assert infos[0]["launches"] == infos[1]["launches"] == infos[2]["launches"] and all(i["exchanges"] > 0 for i in infos)


def test_halo_arrival_inside_the_launch_under_load(sfl, oracle):
    """SFL_OPT_EXCHANGE_SCHEDULE = 3: exchanges in time -- the message leaves on a device-side count of the launch's sender tiles, the
    next launch is queued without a cross-stream event and its cut-adjacent tiles poll an arrival count and acquire.  Many solves back to back on 8 virtual ranks of a grid wide enough
    that the launches fill the chip (the polling tiles' CUs are busy and L1-warm from the previous launch, the exchange
    stream's copies and ghost-row launches run beside them), every solve's result against the oracle."""
    dim_x, dim_y, nranks, iters = 4096, 2048, 8, 31
    rng = np.random.default_rng(2026)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        slabs[0].set_option(sfl.capi.OPT_SOR_FUSE, 10)
        slabs[0].set_option(sfl.capi.OPT_SOR_HALO, 32)     # (the automatic depth is a timed choice: this test wants its exchanges)
        assert slabs[3].get_option(sfl.capi.OPT_EXCHANGE_SCHEDULE) == 3   # automatic on virtual ranks: in time
        for rep in range(3):
            slabs[0].set_option(sfl.capi.OPT_EXCHANGE_SCHEDULE, 2 if rep == 1 else 3)   # (early exchanges behind events in the middle repetition)
            d = (rng.standard_normal((dim_y, dim_x)) * 0.1).astype(np.float32)
            for s in slabs:
                s.upload(sfl.capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
            for _ in range(6):       # the same right-hand side again and again: ghost rows keep being rewritten
                slabs[0].poisson_solve(1.0, iters, OMEGA)
            slabs[0].synchronize()
            got = np.concatenate([s.download(sfl.capi.FIELD_PRESSURE) for s in slabs], axis=0)
            assert_bit_equal(got, oracle.poisson_solve(d, 1.0, iters, OMEGA), f"repetition {rep}")
        assert slabs[3].last_solve_info()["exchanges"] >= 2
    finally:
        for s in slabs:
            s.close()


def test_overlapped_exchange_inside_a_full_step(sfl, oracle):
    """The whole sim step on four virtual ranks with overlapped solve exchanges, several steps in a
    row (buffers ping-pong across steps), against the oracle."""
    dim_x, dim_y, iters, nranks = 256, 480, 14, 4
    v, c, _ = random_fields(dim_x, dim_y, 77, 50.0)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        slabs[0].set_option(sfl.capi.OPT_SOR_FUSE, 6)
        slabs[0].set_option(sfl.capi.OPT_SOR_HALO, 12)
        slabs[0].set_option(sfl.capi.OPT_ADVECT_HALO, 8)
        for s in slabs:
            s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
            s.upload(sfl.capi.FIELD_COLOR, c[s.row_begin:s.row_end])
        vo, co = v, c
        for step in range(3):
            slabs[0].step(DT, 1.0, iters, OMEGA)
            vo, do, po, co = oracle.step(vo, co, DT, 1.0, iters, OMEGA)
        slabs[0].synchronize()
        cat = lambda f: np.concatenate([s.download(f) for s in slabs], axis=0)
        assert_bit_equal(cat(sfl.capi.FIELD_VELOCITY), vo, "velocity after 3 steps")
        assert_bit_equal(cat(sfl.capi.FIELD_PRESSURE), po, "pressure after 3 steps")
        assert_bit_equal(cat(sfl.capi.FIELD_COLOR), co, "colour after 3 steps")
    finally:
        for s in slabs:
            s.close()


def test_baseline_config5_slab_program_vs_oracle(sfl, oracle):
    """BASELINE config 5's per-GPU program (16384 columns, 2048-row slabs, 200 SOR iterations, every
    option on auto: fuse 16, 64-row halo, seven early p exchanges + the rhs exchange) on
    two neighbouring virtual ranks -- a 16384 x 4096 domain -- against the oracle, every cell."""
    dim_x, dim_y, iters, nranks = 16384, 4096, 200, 2
    rng = np.random.default_rng(55)
    d = (rng.standard_normal((dim_y, dim_x)) * 0.1).astype(np.float32)
    want = oracle.poisson_solve(d, 1.0, iters, OMEGA)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        for s in slabs:
            s.upload(sfl.capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
        slabs[0].poisson_solve(1.0, iters, OMEGA)
        slabs[0].synchronize()
        info = slabs[1].last_solve_info()
        got = np.concatenate([s.download(sfl.capi.FIELD_PRESSURE) for s in slabs], axis=0)
    finally:
        for s in slabs:
            s.close()
    # the rhs + six in-time p exchanges (supersteps of four launches: 64 passes per 64-row halo)
    assert info["fuse"] == 16 and info["launches"] == 25 and info["halo"] >= 64
    assert 1 <= info["exchanges"] == plan_exchanges(sfl, dim_y, nranks, iters, 16, info["halo"]) <= 7
    assert_bit_equal(got, want, "C5 slab program: 16384 x 4096 in two slabs, 200 iterations")


@pytest.mark.parametrize("nranks,dim_y,vy,expect", [(2, 200, 9.0, "halo"), (3, 300, 40.5, "halo"), (4, 400, 130.0, "gather"),
                                                    (8, 96, 20.0, "gather"), (2, 64, -700.0, "gather")])
def test_automatic_advection_halo_and_gather_fallback(sfl, oracle, nranks, dim_y, vy, expect):
    """SFL_OPT_ADVECT_HALO = 0: the reach of the back-traces is measured before every advection; it is
    exchanged exactly when it fits the ghost rows and the thinnest slab, otherwise the whole field is
    gathered on every (virtual) GPU (all-gather fallback).  Vertical speeds of `vy` rows per
    step, far beyond the default 4-row halo: velocity self-advection, dye advection and a whole step
    against the oracle; never an SFL_ERR_HALO."""
    dim_x, iters = 96, 5
    rng = np.random.default_rng(int(abs(vy)) + nranks)
    v = np.empty((dim_y, dim_x, 2), np.float32)
    v[..., 0] = rng.uniform(-60, 60, (dim_y, dim_x))
    v[..., 1] = (vy * 30.0) * rng.uniform(0.2, 1.0, (dim_y, dim_x))     # rows per step = v_y * dt
    c = rng.integers(0, 2 ** 31, (dim_y, dim_x, 3), dtype=np.uint32)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        slabs[0].set_option(sfl.capi.OPT_ADVECT_HALO, 0)
        cat = lambda f: np.concatenate([s.download(f) for s in slabs], axis=0)

        def load():
            for s in slabs:
                s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
                s.upload(sfl.capi.FIELD_COLOR, c[s.row_begin:s.row_end])
        load()
        slabs[0].advect_color(DT, False)
        slabs[0].synchronize()
        assert_bit_equal(cat(sfl.capi.FIELD_COLOR), oracle.advect_vec3uq32(c, v, DT, False), "dye advection")
        slabs[0].advect_velocity(DT, True)
        slabs[0].synchronize()
        assert_bit_equal(cat(sfl.capi.FIELD_VELOCITY), oracle.advect_vec2f(v, v, DT, True), "velocity advection")
        load()
        slabs[0].step(DT, 1.0, iters, OMEGA)
        slabs[0].synchronize()
        want = oracle.step(v, c, DT, 1.0, iters, OMEGA)
        assert_bit_equal(cat(sfl.capi.FIELD_VELOCITY), want[0], "step: velocity")
        assert_bit_equal(cat(sfl.capi.FIELD_COLOR), want[3], "step: colour")
        # the same fields with the default fixed halo must be REPORTED, not silently wrong
        slabs[0].set_option(sfl.capi.OPT_ADVECT_HALO, 4)
        load()
        slabs[0].advect_velocity(DT, True)
        with pytest.raises(sfl.SflError) as e:
            slabs[0].synchronize()
        assert e.value.code == sfl.capi.ERR_HALO
    finally:
        for s in slabs:
            s.close()
    reach = abs(vy) + 1
    thinnest = dim_y // nranks
    assert (expect == "gather") == (reach > 64 or reach > thinnest)


@pytest.mark.parametrize("nranks,fuse_projection,vamp", [(4, 1, 45.0), (4, 0, 45.0), (2, 1, 45.0), (7, 1, 45.0), (3, 1, 400.0),
                                                         (4, 1, 900.0)])
def test_slab_steps_on_the_automatic_halo_without_mid_step_round_trips(sfl, oracle, nranks, fuse_projection, vamp):
    """SFL_OPT_ADVECT_HALO = 0 inside sfl_step (the default): from the second step on nothing is measured before an
    advection -- the velocity advection takes the reach reported at the end of the previous step, the dye advection
    runs on that reach plus a margin and is CHECKED afterwards, and repeated from the untouched old buffer when a
    back-trace left the guess.  Six steps in a row against the oracle, every field after every step: a calm start,
    then a drag force (ino:264-269) that throws a jet of 30 rows per step across a cut between advection and
    projection (the guess of that step is short: the repeat path), the jet spreading in the following steps (reach
    beyond the 64 ghost rows / the thinnest slab: the gathered path), an upload from outside in between.
    The velocity advection also covers the ghost row next to each cut (calculate_divergence then exchanges nothing):
    with per-cell noise of 13 or 30 rows per step (vamp 400 / 900) the neighbour's edge row traces deep into the
    neighbour's own slab, which the halo has to cover (reach_extended)."""
    dim_x, dim_y, iters = 80, 224, 4
    v, c, _ = random_fields(dim_x, dim_y, 77 + nranks, vamp)         # |v dt| <= 1.5 rows at vamp 45
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    cut = slabs[1].row_begin
    forces = {2: (np.array([[i, cut + 1] for i in range(20, 60)], np.int32),
                  np.array([[0.0, 900.0]] * 40, np.float32)),          # 30 rows per step, upwards out of slab 1
              4: (np.array([[7, 5]], np.int32), np.array([[30.0, -2400.0]], np.float32))}   # 80 rows per step
    try:
        sfl.Solver.link_group(slabs)
        slabs[0].set_option(sfl.capi.OPT_FUSE_PROJECTION, fuse_projection)
        cat = lambda f: np.concatenate([s.download(f) for s in slabs], axis=0)
        for s in slabs:
            s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
            s.upload(sfl.capi.FIELD_COLOR, c[s.row_begin:s.row_end])
        for k in range(6):
            if k == 3:   # the velocity is replaced from outside: the reported reach no longer describes it
                v = (v * np.float32(0.5)).astype(np.float32)
                for s in slabs:
                    s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
            va = oracle.advect_vec2f(v, v, DT, True)
            if k in forces:
                cells, vel = forces[k]
                slabs[0].queue_forces(cells, vel)
                for (ci, cj), f in zip(cells, vel):
                    va[cj, ci] = f
            slabs[0].step(DT, 1.0, iters, OMEGA)
            if k % 2:    # sometimes the next step settles the previous one, sometimes synchronize / download do
                slabs[0].synchronize()
            p = oracle.poisson_solve(oracle.divergence(va, 1.0), 1.0, iters, OMEGA)
            v = oracle.subtract_gradient(va, p, 1.0)
            c = oracle.advect_vec3uq32(c, v, DT, False)
            if k in (1, 2, 4, 5):
                assert_bit_equal(cat(sfl.capi.FIELD_COLOR), c, f"{nranks} slabs, step {k}: colour")
                assert_bit_equal(cat(sfl.capi.FIELD_VELOCITY), v, f"{nranks} slabs, step {k}: velocity")
        slabs[0].synchronize()
        assert_bit_equal(cat(sfl.capi.FIELD_COLOR), c, f"{nranks} slabs, final colour")
        assert_bit_equal(cat(sfl.capi.FIELD_PRESSURE), p, f"{nranks} slabs, final pressure")
    finally:
        for s in slabs:
            s.close()


@pytest.mark.parametrize("nranks,jet", [(2, 0.0), (2, 80.0), (3, 70.0), (3, 0.0)])
def test_early_interior_advection_is_kept_only_where_it_holds(sfl, oracle, nranks, jet):
    """sfl_step on slabs queues the velocity advection of the rows further than 64 from both cuts BEFORE the host reads the
    previous step's report (those rows need no halo); the report's word [3] -- how far from its own row any cell's sources lie,
    measured by the dye's kernel together with the reach -- says whether that held.  Calm flow: kept from the second step on.
    With a jet of 70 / 80 rows per step just inside that zone (its back-traces cross the cut; the reach, 4 - 14 rows, would have
    let it pass) the early rows are dropped and everything is advected after the report.  Four steps back to back, every field
    against the oracle."""
    dim_x, rows, iters = 256, 320, 6
    dim_y = rows * nranks
    v, c, _ = random_fields(dim_x, dim_y, 500 + nranks, 40.0)          # |v dt| <= 1.4 rows
    if jet:
        for r in range(1, nranks):
            v[r * rows + 66:r * rows + 76, 40:200, 1] = jet / float(DT)     # upwards: sources `jet` rows below, across the cut
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    kept = []
    try:
        sfl.Solver.link_group(slabs)
        cat = lambda f: np.concatenate([s.download(f) for s in slabs], axis=0)
        for s in slabs:
            s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
            s.upload(sfl.capi.FIELD_COLOR, c[s.row_begin:s.row_end])
        for k in range(4):      # back to back: nothing touches the contexts in between, every step finds a report pending
            slabs[0].step(DT, 1.0, iters, OMEGA)
            kept.append(slabs[1].get_option(sfl.capi.OPT_LAST_EARLY_ROWS))
            v, d, p, c = oracle.step(v, c, DT, 1.0, iters, OMEGA)
        slabs[0].synchronize()
        assert_bit_equal(cat(sfl.capi.FIELD_VELOCITY), v, f"{nranks} slabs, jet {jet}: velocity after 4 steps")
        assert_bit_equal(cat(sfl.capi.FIELD_COLOR), c, f"{nranks} slabs, jet {jet}: colour after 4 steps")
        assert_bit_equal(cat(sfl.capi.FIELD_PRESSURE), p, f"{nranks} slabs, jet {jet}: pressure after 4 steps")
    finally:
        for s in slabs:
            s.close()
    assert kept[0] == 0                      # the first step measures before it advects
    if jet:
        assert kept[1] == 0, kept            # the jet's sources lie further than 64 rows from their cells
    else:
        assert kept[1:] == [64, 64, 64], kept


def _smooth_velocity(dim_x, dim_y, amp):
    """Solid-body swirl + a shear: neighbouring cells back-trace to neighbouring texels (what a
    simulation has, as opposed to random_fields' per-cell noise)."""
    j, i = np.mgrid[0:dim_y, 0:dim_x].astype(np.float32)
    v = np.empty((dim_y, dim_x, 2), np.float32)
    v[..., 0] = amp * (-(j - dim_y / 2) / max(dim_y, 2) + 0.3 * np.sin(i / 17.0))
    v[..., 1] = amp * ((i - dim_x / 2) / max(dim_x, 2) + 0.3 * np.cos(j / 13.0))
    return v


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("dim_x,dim_y", SHAPES + [(1000, 300), (129, 517), (2048, 65)])
def test_both_advection_kernels_vs_oracle(sfl, oracle, kernel, dim_x, dim_y):
    """SFL_OPT_ADVECT_KERNEL: the one-thread-per-cell gather (1) and the LDS-staged tiles (2) against the
    oracle on noise (back-traces scattered over the window, many leaving it), on a smooth field (all inside
    the window), on speeds far beyond the window (every sample from memory) and through a whole step."""
    fields = [("noise", random_fields(dim_x, dim_y, 11, 100.0)[0]),
              ("smooth", _smooth_velocity(dim_x, dim_y, 110.0)),
              ("fast", random_fields(dim_x, dim_y, 12, 1500.0)[0]),
              ("still", np.zeros((dim_y, dim_x, 2), np.float32))]
    _, c, pr = random_fields(dim_x, dim_y, 13)
    with sfl.Solver(dim_x, dim_y) as s:
        s.set_option(sfl.capi.OPT_ADVECT_KERNEL, kernel)
        assert s.get_option(sfl.capi.OPT_ADVECT_KERNEL) == kernel
        for name, v in fields:
            for ns in (True, False):
                s.upload(sfl.capi.FIELD_VELOCITY, v)
                s.upload(sfl.capi.FIELD_COLOR, c)
                s.advect_color(DT, ns)
                s.advect_velocity(DT, ns)
                s.synchronize()
                assert_bit_equal(s.download(sfl.capi.FIELD_COLOR), oracle.advect_vec3uq32(c, v, DT, ns),
                                 f"{name}: dye, no_slip {ns}")
                assert_bit_equal(s.download(sfl.capi.FIELD_VELOCITY), oracle.advect_vec2f(v, v, DT, ns),
                                 f"{name}: velocity, no_slip {ns}")
            # the finite-difference operators follow the same option (tiled / one thread per cell)
            s.upload(sfl.capi.FIELD_VELOCITY, v)
            s.upload(sfl.capi.FIELD_PRESSURE, pr)
            s.calculate_divergence(0.5)
            s.subtract_gradient(0.5)
            s.synchronize()
            assert_bit_equal(s.download(sfl.capi.FIELD_DIVERGENCE), oracle.divergence(v, 0.5), f"{name}: divergence")
            assert_bit_equal(s.download(sfl.capi.FIELD_VELOCITY), oracle.subtract_gradient(v, pr, 0.5),
                             f"{name}: gradient")
            for fused in (1, 0):
                s.set_option(sfl.capi.OPT_FUSE_PROJECTION, fused)
                s.upload(sfl.capi.FIELD_VELOCITY, v)
                s.upload(sfl.capi.FIELD_COLOR, c)
                s.step(DT, 1.0, 3, OMEGA)
                s.synchronize()
                want = oracle.step(v, c, DT, 1.0, 3, OMEGA)
                assert_bit_equal(s.download(sfl.capi.FIELD_VELOCITY), want[0], f"{name}: step velocity ({fused})")
                assert_bit_equal(s.download(sfl.capi.FIELD_COLOR), want[3], f"{name}: step colour ({fused})")


def test_host_advect_of_another_field_on_a_tiled_size(hip, oracle):
    """advect(next, q, v) with q != v (advect.h:74-76 allows it) at a size the automatic choice gives to the
    tiled kernel: the cell's own velocity then comes from memory, the window holds q."""
    v, _, _ = random_fields(320, 200, 21, 90.0)
    q, _, _ = random_fields(320, 200, 22, 5.0)
    for ns in (True, False):
        assert_bit_equal(hip.advect_vec2f(q, v, DT, ns), oracle.advect_vec2f(q, v, DT, ns), f"other field, {ns}")


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("nranks,dim_y,halo", [(2, 100, 4), (3, 131, 7), (4, 64, 0)])
def test_both_advection_kernels_on_slabs(sfl, oracle, kernel, nranks, dim_y, halo):
    """The tiled kernel's window on a slab stops at the exchanged ghost rows (halo 0 = measured reach,
    including the gathered whole field when the reach outruns a slab)."""
    dim_x = 200
    rng = np.random.default_rng(nranks)
    v = np.empty((dim_y, dim_x, 2), np.float32)
    v[..., 0] = rng.uniform(-100, 100, (dim_y, dim_x))
    v[..., 1] = rng.uniform(-1, 1, (dim_y, dim_x)) * (30.0 * (halo - 1) if halo else 600.0)
    _, c, _ = random_fields(dim_x, dim_y, 5)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        slabs[0].set_option(sfl.capi.OPT_ADVECT_KERNEL, kernel)
        slabs[0].set_option(sfl.capi.OPT_ADVECT_HALO, halo)
        cat = lambda f: np.concatenate([s.download(f) for s in slabs], axis=0)
        for s in slabs:
            s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
            s.upload(sfl.capi.FIELD_COLOR, c[s.row_begin:s.row_end])
        slabs[0].advect_color(DT, False)
        slabs[0].advect_velocity(DT, True)
        slabs[0].synchronize()
        assert_bit_equal(cat(sfl.capi.FIELD_COLOR), oracle.advect_vec3uq32(c, v, DT, False), "dye")
        assert_bit_equal(cat(sfl.capi.FIELD_VELOCITY), oracle.advect_vec2f(v, v, DT, True), "velocity")
        for s in slabs:
            s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
            s.upload(sfl.capi.FIELD_COLOR, c[s.row_begin:s.row_end])
        slabs[0].set_option(sfl.capi.OPT_ADVECT_HALO, 0)   # the projected velocity may outrun a fixed halo
        slabs[0].step(DT, 1.0, 4, OMEGA)
        slabs[0].synchronize()
        want = oracle.step(v, c, DT, 1.0, 4, OMEGA)
        assert_bit_equal(cat(sfl.capi.FIELD_VELOCITY), want[0], "step: velocity")
        assert_bit_equal(cat(sfl.capi.FIELD_COLOR), want[3], "step: colour")
    finally:
        for s in slabs:
            s.close()


@pytest.mark.parametrize("dim_x,dim_y", SHAPES + [(1000, 300), (129, 517), (2048, 65), (64, 32), (65, 33), (63, 31)])
def test_step_with_fused_advection_and_divergence(sfl, oracle, dim_x, dim_y):
    """SFL_OPT_FUSE_DIVERGENCE: sfl_step's velocity advection + calculate_divergence as ONE kernel (tile +
    ring advected, differenced in LDS) against the oracle's step -- divergence field included -- on noise, a
    smooth field and speeds beyond the window; with a queued force the two operators run separately (the
    force lands between them, ino:264-269) and the step must still match."""
    _, c, _ = random_fields(dim_x, dim_y, 31)
    fields = [("noise", random_fields(dim_x, dim_y, 32, 100.0)[0]), ("smooth", _smooth_velocity(dim_x, dim_y, 120.0)),
              ("fast", random_fields(dim_x, dim_y, 33, 900.0)[0])]
    with sfl.Solver(dim_x, dim_y) as s:
        s.set_option(sfl.capi.OPT_ADVECT_KERNEL, 2)
        assert s.get_option(sfl.capi.OPT_FUSE_DIVERGENCE) == 1
        for name, v in fields:
            want = oracle.step(v, c, DT, 1.0, 3, OMEGA)
            for fused in (1, 0):
                s.set_option(sfl.capi.OPT_FUSE_DIVERGENCE, fused)
                s.upload(sfl.capi.FIELD_VELOCITY, v)
                s.upload(sfl.capi.FIELD_COLOR, c)
                s.step(DT, 1.0, 3, OMEGA)
                s.synchronize()
                for field, k in ((sfl.capi.FIELD_DIVERGENCE, 1), (sfl.capi.FIELD_VELOCITY, 0),
                                 (sfl.capi.FIELD_PRESSURE, 2), (sfl.capi.FIELD_COLOR, 3)):
                    assert_bit_equal(s.download(field), want[k], f"{name}: field {field}, fused {fused}")
        # a drag force between advection and divergence
        s.set_option(sfl.capi.OPT_FUSE_DIVERGENCE, 1)
        v = fields[0][1]
        cells = np.array([[dim_x // 2, dim_y // 2], [0, 0]], np.int32)
        vel = np.array([[55.0, -35.0], [-20.0, 10.0]], np.float32)
        s.upload(sfl.capi.FIELD_VELOCITY, v)
        s.upload(sfl.capi.FIELD_COLOR, c)
        s.queue_forces(cells, vel)
        s.step(DT, 1.0, 3, OMEGA)
        s.step(DT, 1.0, 3, OMEGA)      # and a fused one right after it
        s.synchronize()
        v1 = oracle.advect_vec2f(v, v, DT, True)
        for (ci, cj), f in zip(cells, vel):
            v1[cj, ci] = f
        d = oracle.divergence(v1, 1.0)
        p = oracle.poisson_solve(d, 1.0, 3, OMEGA)
        v1 = oracle.subtract_gradient(v1, p, 1.0)
        c1 = oracle.advect_vec3uq32(c, v1, DT, False)
        want = oracle.step(v1, c1, DT, 1.0, 3, OMEGA)
        assert_bit_equal(s.download(sfl.capi.FIELD_VELOCITY), want[0], "after a forced step + a fused step: velocity")
        assert_bit_equal(s.download(sfl.capi.FIELD_COLOR), want[3], "after a forced step + a fused step: colour")


@pytest.mark.parametrize("fuse", [2, 4, 6, 8, 10, 12, 14, 16])
def test_every_fuse_depth_from_zero_and_continuing(sfl, oracle, fuse):
    """Each compiled depth of the fused kernel, first launch (p implicitly zero) and continuing launches, interior /
    flipped / boundary tiles, dx = 1 and dx != 1, on a grid large enough for several tiles per strip (the register
    allocation differs per depth: NS = 14 is held at three waves per SIMD with a few spilled registers)."""
    dim_x, dim_y = 1500, 1100
    _, _, d = random_fields(dim_x, dim_y, 40 + fuse)
    iters = fuse + fuse // 2          # three launches of this depth
    for dx in (1.0, 0.75):
        with sfl.Solver(dim_x, dim_y) as s:
            s.set_option(sfl.capi.OPT_SOR_KERNEL, 2)
            s.set_option(sfl.capi.OPT_SOR_FUSE, fuse)
            s.upload(sfl.capi.FIELD_DIVERGENCE, d)
            s.poisson_solve(dx, iters, OMEGA)
            s.synchronize()
            info = s.last_solve_info()
            got = s.download(sfl.capi.FIELD_PRESSURE)
        assert info["fuse"] == fuse and info["launches"] == 3
        assert_bit_equal(got, oracle.poisson_solve(d, dx, iters, OMEGA), f"fuse {fuse} dx {dx}")


SMALL_SHAPES = [(2, 2), (2, 7), (7, 2), (3, 3), (4, 5), (61, 81), (80, 60), (64, 48), (127, 33), (128, 48), (78, 78),
                (3, 2048), (2047, 3), (2, 3072), (3072, 2), (257, 23), (128, 80), (101, 101)]


@pytest.mark.parametrize("dim_x,dim_y", SMALL_SHAPES)
def test_small_grid_one_launch_step_and_solve(sfl, oracle, dim_x, dim_y):
    """SFL_OPT_SMALL_GRID (default on): grids of <= 6144 cells run sfl_poisson_solve and sfl_step as ONE launch of
    one workgroup with the fields in LDS (the sketch's 61 x 81 among them).  Against the oracle and against the
    general kernels: every field of the step, forces between advection and divergence, iters 0 / 1 / 9, dx != 1,
    shapes down to 2 x 2, up to exactly 6144 cells and beyond (general kernels), including widths where one colour holds 2/3 of a row (3 x 3413: too many of one
    colour per thread, general kernels)."""
    v, c, d = random_fields(dim_x, dim_y, 70 + dim_x, 90.0)
    cells = np.array([[dim_x // 2, dim_y // 2], [0, 0], [dim_x - 1, dim_y - 1], [dim_x // 2, dim_y // 2]], np.int32)
    vel = np.array([[55.0, -35.0], [-20.0, 10.0], [3.0, 4.0], [-8.0, 6.0]], np.float32)
    with sfl.Solver(dim_x, dim_y) as s:
        assert s.get_option(sfl.capi.OPT_SMALL_GRID) == 1
        for small in (1, 0):
            s.set_option(sfl.capi.OPT_SMALL_GRID, small)
            for iters, dx in ((9, 1.0), (1, 0.5), (0, 1.0)):
                s.upload(sfl.capi.FIELD_DIVERGENCE, d)
                s.poisson_solve(dx, iters, OMEGA)
                s.synchronize()
                fits = dim_x * dim_y <= 6144 and dim_y * ((dim_x + 1) // 2) <= 3072   # cells of one colour per thread
                if small and fits:
                    assert s.last_solve_info()["launches"] == 1
                assert_bit_equal(s.download(sfl.capi.FIELD_PRESSURE), oracle.poisson_solve(d, dx, iters, OMEGA),
                                 f"solve, small {small}, iters {iters}, dx {dx}")
                s.upload(sfl.capi.FIELD_VELOCITY, v)
                s.upload(sfl.capi.FIELD_COLOR, c)
                s.step(DT, dx, iters, OMEGA)
                s.synchronize()
                want = oracle.step(v, c, DT, dx, iters, OMEGA)
                for field, k in ((sfl.capi.FIELD_VELOCITY, 0), (sfl.capi.FIELD_DIVERGENCE, 1),
                                 (sfl.capi.FIELD_PRESSURE, 2), (sfl.capi.FIELD_COLOR, 3)):
                    assert_bit_equal(s.download(field), want[k], f"step field {field}, small {small}, iters {iters}, dx {dx}")
            # two steps, the first with drag forces between advection and divergence (last write wins)
            s.upload(sfl.capi.FIELD_VELOCITY, v)
            s.upload(sfl.capi.FIELD_COLOR, c)
            s.queue_forces(cells, vel)
            s.step(DT, 1.0, 5, OMEGA)
            s.step(DT, 1.0, 5, OMEGA)
            s.synchronize()
            v1 = oracle.advect_vec2f(v, v, DT, True)
            for (ci, cj), f in zip(cells, vel):
                v1[cj, ci] = f
            p1 = oracle.poisson_solve(oracle.divergence(v1, 1.0), 1.0, 5, OMEGA)
            v1 = oracle.subtract_gradient(v1, p1, 1.0)
            c1 = oracle.advect_vec3uq32(c, v1, DT, False)
            want = oracle.step(v1, c1, DT, 1.0, 5, OMEGA)
            assert_bit_equal(s.download(sfl.capi.FIELD_VELOCITY), want[0], f"forced + plain step: velocity, small {small}")
            assert_bit_equal(s.download(sfl.capi.FIELD_COLOR), want[3], f"forced + plain step: colour, small {small}")


def test_small_grid_limit_and_explicit_options(sfl, oracle):
    """6145 cells take the general kernels; so does a small grid with an explicit kernel option (honoured as given)."""
    _, _, d = random_fields(1229, 5, 5)   # 6145 cells
    with sfl.Solver(1229, 5) as s:
        s.upload(sfl.capi.FIELD_DIVERGENCE, d)
        s.poisson_solve(1.0, 12, OMEGA)
        s.synchronize()
        assert s.last_solve_info()["launches"] > 1
        assert_bit_equal(s.download(sfl.capi.FIELD_PRESSURE), oracle.poisson_solve(d, 1.0, 12, OMEGA), "6145 cells")
    _, _, d = random_fields(61, 81, 6)
    with sfl.Solver(61, 81) as s:
        s.set_option(sfl.capi.OPT_SOR_FUSE, 8)
        s.upload(sfl.capi.FIELD_DIVERGENCE, d)
        s.poisson_solve(1.0, 12, OMEGA)
        s.synchronize()
        assert s.last_solve_info()["launches"] == 3
        assert_bit_equal(s.download(sfl.capi.FIELD_PRESSURE), oracle.poisson_solve(d, 1.0, 12, OMEGA), "explicit fuse")


def test_emulated_rank_runs_its_program_alone(sfl):
    """sfl_comm_emulate (bench.py --emulate-rank): ONE rank of an 8-slab solve alone on the GPU, every halo message a
    self-copy of the same size on the exchange stream.  Values next to the cuts are meaningless by construction, but
    the program is the rank's own: launch / exchange counts equal the plan's, rows further than the solve's reach
    from the cuts equal the whole-domain result bit for bit, and a whole step runs through (automatic advection
    halo, report, settle) without error."""
    dim_x, dim_y, nranks, rank, iters = 640, 2048, 8, 3, 12
    import bench
    v = bench.synthetic_velocity(dim_x, 0, dim_y)
    with sfl.Solver(dim_x, dim_y) as one:
        one.upload(sfl.capi.FIELD_VELOCITY, v)
        one.calculate_divergence(1.0)
        one.poisson_solve(1.0, iters, OMEGA)
        one.synchronize()
        d, want = one.download(sfl.capi.FIELD_DIVERGENCE), one.download(sfl.capi.FIELD_PRESSURE)
    with sfl.Solver(dim_x, dim_y, 0, rank, nranks) as s:
        with pytest.raises(sfl.SflError):
            s.poisson_solve(1.0, iters, OMEGA)          # a slab without a transport cannot exchange
    with sfl.Solver(dim_x, dim_y, 0, rank, nranks) as s:
        s.comm_emulate()
        assert s.get_option(sfl.capi.OPT_TRANSPORT) == 3
        s.upload(sfl.capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
        s.poisson_solve(1.0, iters, OMEGA)
        s.synchronize()
        info = s.last_solve_info()
        got = s.download(sfl.capi.FIELD_PRESSURE)
        assert info["exchanges"] == plan_exchanges(sfl, dim_y, nranks, iters, info["fuse"], info["halo"]) and info["launches"] == -(-2 * iters // info["fuse"])
        # information travels one row per colour pass: 2 * iters rows from each cut are tainted by the fake halos
        reach = 2 * iters
        inner = slice(reach, (s.row_end - s.row_begin) - reach)
        assert_bit_equal(got[inner], want[s.row_begin:s.row_end][inner], "emulated rank: rows out of the cuts' reach")
        s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
        s.upload(sfl.capi.FIELD_COLOR, bench.synthetic_color(dim_x, s.row_begin, s.row_end))
        for _ in range(3):
            s.step(DT, 1.0, iters, OMEGA)
        s.synchronize()


@pytest.mark.parametrize("schedule,halo", [(2, 0), (2, 32), (1, 0), (3, 0)])
def test_zero_iterations_on_slabs(sfl, schedule, halo):
    """poisson_solve with iters == 0 still zero-fills p (poisson.cpp:117-119) -- on slabs too.  The early-exchange plan used to
    index its empty tables at n - 1 for it (found by UBSan on the host side, tests/cpp/host_san_driver.cpp)."""
    dim_x, dim_y, nranks = 256, 900, 3
    _, _, d = random_fields(dim_x, dim_y, 4)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        slabs[0].set_option(sfl.capi.OPT_EXCHANGE_SCHEDULE, schedule)
        slabs[0].set_option(sfl.capi.OPT_SOR_HALO, halo)
        for s in slabs:
            s.upload(sfl.capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
            s.upload(sfl.capi.FIELD_PRESSURE, d[s.row_begin:s.row_end])
        slabs[0].poisson_solve(1.0, 0, OMEGA)
        slabs[0].synchronize()
        got = np.concatenate([s.download(sfl.capi.FIELD_PRESSURE) for s in slabs], axis=0)
    finally:
        for s in slabs:
            s.close()
    assert not got.any()


@pytest.mark.parametrize("rank,dim_x,dim_y,iters,arrival", [
    (3, 640, 2048, 12, -1), (0, 640, 2048, 12, -1), (7, 1030, 4096, 25, -1), (3, 8192, 8192, 30, -1), (3, 640, 2048, 12, 0),
    (1, 2048, 1024, 20, 1)])
def test_emulated_rank_with_rccl_as_transport(sfl, rank, dim_x, dim_y, iters, arrival):
    """sfl_comm_emulate_rccl (bench.py --emulate-rank R --of 8 --via-rccl): one rank's program with every halo message a REAL
    ncclSend / ncclRecv to the rank itself on a one-rank communicator, issued through the branch a rank of a real communicator
    takes -- the sender count in front of the ncclGroup, the arrival count behind it (exchanges in time, the transport's default;
    arrival 0 = early exchanges behind events) -- and the step's reductions as ncclAllReduce.  Exchange and launch counts equal
    the plan's, the schedule is the one asked for, rows out of the cuts' reach equal the whole-domain solve bit for bit, no
    wait gives up (sfl_synchronize would say so), whole steps run through.  Ref: the loop being sharded, poisson.cpp:121-124."""
    nranks = 8
    import bench
    v = bench.synthetic_velocity(dim_x, 0, dim_y)
    with sfl.Solver(dim_x, dim_y) as one:
        one.upload(sfl.capi.FIELD_VELOCITY, v)
        one.calculate_divergence(1.0)
        one.poisson_solve(1.0, iters, OMEGA)
        one.synchronize()
        d, want = one.download(sfl.capi.FIELD_DIVERGENCE), one.download(sfl.capi.FIELD_PRESSURE)
    del v
    with sfl.Solver(dim_x, dim_y, 0, rank, nranks) as s:
        s.comm_emulate_rccl()
        assert s.get_option(sfl.capi.OPT_TRANSPORT) == 4
        if arrival >= 0:
            s.set_option(sfl.capi.OPT_EXCHANGE_SCHEDULE, 3 if arrival else 2)
        assert s.get_option(sfl.capi.OPT_EXCHANGE_SCHEDULE) == (2 if arrival == 0 else 3)
        s.upload(sfl.capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
        for _ in range(3):
            s.poisson_solve(1.0, iters, OMEGA)
        s.synchronize()
        info = s.last_solve_info()
        got = s.download(sfl.capi.FIELD_PRESSURE)
        in_time = arrival != 0
        assert info["launches"] == -(-2 * iters // info["fuse"])
        assert info["exchanges"] == plan_exchanges(sfl, dim_y, nranks, iters, info["fuse"], info["halo"], kernel=3 if in_time else 2)
        reach = 2 * iters
        rows = s.row_end - s.row_begin
        inner = slice(reach if rank > 0 else 0, rows - (reach if rank < nranks - 1 else 0))
        assert_bit_equal(got[inner], want[s.row_begin:s.row_end][inner], "RCCL-to-self rank: rows out of the cuts' reach")
        s.upload(sfl.capi.FIELD_VELOCITY, bench.synthetic_velocity(dim_x, s.row_begin, s.row_end))
        s.upload(sfl.capi.FIELD_COLOR, bench.synthetic_color(dim_x, s.row_begin, s.row_end))
        for _ in range(3):
            s.step(DT, 1.0, iters, OMEGA)
        s.synchronize()


def test_a_wait_that_gives_up_fails_the_download_and_the_next_operator(sfl):
    """a halo wait inside a launch that gives up leaves an invalid pressure field; it used to be reported by
    sfl_synchronize only.  Here the message is held back longer than the (shortened) limit: the solve's launches give up,
    sfl_download refuses to hand the field out, the next operator refuses to build on it, sfl_synchronize reports and clears
    the condition, and the context works again afterwards."""
    dim_x, dim_y, nranks, rank, iters = 640, 2048, 8, 3, 12
    _, _, d = random_fields(dim_x, dim_y, 91)
    with sfl.Solver(dim_x, dim_y, 0, rank, nranks) as s:
        s.comm_emulate()
        s.upload(sfl.capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
        s.poisson_solve(1.0, iters, OMEGA)
        s.synchronize()
        good = s.download(sfl.capi.FIELD_PRESSURE)
        s.set_option(sfl.capi.OPT_HALO_TIMEOUT_MS, 2)
        s.set_option(sfl.capi.OPT_EMULATE_WIRE_US, 10000)     # every message 10 ms late, the waits give up after 2 ms
        s.poisson_solve(1.0, iters, OMEGA)
        with pytest.raises(sfl.SflError, match="gave up"):
            s.download(sfl.capi.FIELD_PRESSURE)
        with pytest.raises(sfl.SflError, match="gave up"):
            s.poisson_solve(1.0, iters, OMEGA)
        with pytest.raises(sfl.SflError, match="lasted longer"):
            s.synchronize()
        s.set_option(sfl.capi.OPT_EMULATE_WIRE_US, 0)
        s.set_option(sfl.capi.OPT_HALO_TIMEOUT_MS, 0)
        s.poisson_solve(1.0, iters, OMEGA)
        s.synchronize()
        assert_bit_equal(s.download(sfl.capi.FIELD_PRESSURE), good, "after the reported time-out the context solves as before")


def test_in_time_exchanges_under_the_runtimes_default_queues_with_many_live_streams(sfl, oracle):
    """the runtime folds its streams onto GPU_MAX_HW_QUEUES (4 by default) hardware queues; a launch that waits for a
    halo message inside the kernel must not share a queue with the stream that carries the message.  Twelve other contexts
    (a compute and an exchange stream each) are alive and have been used when the tested group is created and solves; whatever
    the streams' placement, the library either finds its two streams running side by side (exchanges in time) or falls back to
    events -- the result is the undivided solve's, and no wait gives up."""
    dim_x, dim_y, nranks, iters = 1030, 1200, 3, 30
    _, _, d = random_fields(dim_x, dim_y, 17)
    want = oracle.poisson_solve(d, 1.0, iters, OMEGA)
    crowd = [sfl.Solver(256, 512, 0, r % 4, 4) for r in range(12)]
    try:
        for c in crowd:
            c.comm_emulate()
            c.upload(sfl.capi.FIELD_DIVERGENCE, np.zeros((c.row_end - c.row_begin, 256), np.float32))
            c.poisson_solve(1.0, 4, OMEGA)
        slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
        try:
            sfl.Solver.link_group(slabs)
            schedule = slabs[0].get_option(sfl.capi.OPT_EXCHANGE_SCHEDULE)
            assert schedule in (2, 3)
            for s in slabs:
                s.upload(sfl.capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
            for _ in range(4):
                slabs[0].poisson_solve(1.0, iters, OMEGA)
                for c in crowd[:4]:
                    c.poisson_solve(1.0, 4, OMEGA)
            slabs[0].synchronize()
            got = np.concatenate([s.download(sfl.capi.FIELD_PRESSURE) for s in slabs], axis=0)
        finally:
            for s in slabs:
                s.close()
        for c in crowd:
            c.synchronize()
    finally:
        for c in crowd:
            c.close()
    assert_bit_equal(got, want, f"3 virtual ranks beside 24 live streams (schedule {schedule})")


def test_a_short_dye_guess_with_the_early_rows_queued_and_overlap_off(sfl, oracle):
    """sfl_step queues the early interior advection (and records "velocity and dye are final") BEFORE it examines the
    last step's report; when that report says the dye's guessed halo was short, the dye is advected again -- after the event.  The
    dye's halo of the new step must not leave behind the stale event.  Forced here: large forces in step k (the guess made
    before them is short), none in step k + 1 (the early rows are queued), exchanges in line (SFL_OPT_EXCHANGE_SCHEDULE = 1) and the baseline kernel
    (no exchange of the solve on the exchange stream orders anything by accident).  Every field of every step against the
    oracle on the whole domain."""
    dim_x, dim_y, nranks, iters = 640, 1536, 2, 6
    v, c, _ = random_fields(dim_x, dim_y, 23, vamp=20.0)
    for kernel, overlap in ((0, 0), (1, 1)):
        slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
        try:
            sfl.Solver.link_group(slabs)
            slabs[0].set_option(sfl.capi.OPT_EXCHANGE_SCHEDULE, 0 if overlap else 1)
            if kernel:
                slabs[0].set_option(sfl.capi.OPT_SOR_KERNEL, kernel)
            for s in slabs:
                s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
                s.upload(sfl.capi.FIELD_COLOR, c[s.row_begin:s.row_end])
            wv, wc = v, c
            for k in range(5):
                forces = None
                if k in (1, 3):   # a jet of 900 cells / s next to the cut: a reach of 30 rows where 2 - 3 were guessed
                    cells = np.array([[i, dim_y // 2 + dj] for i in range(100, 540, 4) for dj in (-3, 2)], np.int32)
                    vel = np.tile(np.array([[0.0, 900.0]], np.float32), (len(cells), 1)) * np.where(cells[:, 1:] < dim_y // 2, 1, -1)
                    forces = (cells, vel.astype(np.float32))
                    slabs[0].queue_forces(cells, vel)
                slabs[0].step(DT, 1.0, iters, OMEGA)
                wv, _, wp, wc = oracle_step(oracle, wv, wc, iters, forces)
                if k in (2, 4):
                    assert slabs[0].get_option(sfl.capi.OPT_LAST_EARLY_ROWS) >= 0
            slabs[0].synchronize()
            got_c = np.concatenate([s.download(sfl.capi.FIELD_COLOR) for s in slabs], axis=0)
            got_v = np.concatenate([s.download(sfl.capi.FIELD_VELOCITY) for s in slabs], axis=0)
        finally:
            for s in slabs:
                s.close()
        assert_bit_equal(got_v, wv, f"velocity after 5 steps (kernel {kernel}, overlap {overlap})")
        assert_bit_equal(got_c, wc, f"dye after 5 steps with two short guesses (kernel {kernel}, overlap {overlap})")


def test_bench_checks_the_sim_steps_fields_across_ranks():
    """bench.py's check of the sim step's fields for N > 1 (every rank's rows against the same steps on one whole-domain context
    on rank 0's GPU, by checksums), run here with its one-GPU switch: the replay takes the same number of steps, the checksums of
    equal fields agree, the line carries the verdict."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "640", "--dim-y", "512", "--iters", "10", "--steps", "2",
                        "--warmup", "1", "--sim-steps", "3", "--no-priming", "--check-sim-step-parity"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    sp = out.get("sim_step_parity")
    assert sp and sp.get("bit_exact") is True and sp["ranks_differing"] == [], sp
    assert "after 7 sim steps" in sp["what"]
    assert out["parity"]["bit_exact"]


# ---- round 4: the BASELINE multi-GPU configurations in full on eight virtual ranks -----------------------------
def oracle_step(oracle, v, c, iters, forces=None):
    """One sim step of the checker in the order of ino:252-287, drag forces (cells (i, j), velocities) written
    between the velocity advection and the divergence (ino:264-269).  Returns (v, div, p, colour)."""
    va = oracle.advect_vec2f(v, v, DT, True)
    if forces:
        for (i, j), u in zip(*forces):
            va[j, i] = u
    d = oracle.divergence(va, 1.0)
    p = oracle.poisson_solve(d, 1.0, iters, OMEGA)
    vp = oracle.subtract_gradient(va, p, 1.0)
    return vp, d, p, oracle.advect_vec3uq32(c, vp, DT, False)


def test_baseline_config5_in_full_on_eight_virtual_ranks(sfl, oracle):
    """BASELINE config 5 at FULL size -- 16384 x 16384, 200 SOR iterations, eight 2048-row slabs, every option on
    auto (fuse 16, 64-row halo: 25 launches, the rhs exchange and seven early p exchanges with their ghost-row
    launches) -- executed by eight virtual ranks on one GPU, every cell against the oracle (poisson.cpp:114-125;
    ~45 s of one host core)."""
    dim, iters, nranks = 16384, 200, 8
    rng = np.random.default_rng(1605)
    d = (rng.standard_normal((dim, dim), dtype=np.float32) * np.float32(0.1))
    want = oracle.poisson_solve(d, 1.0, iters, OMEGA)
    slabs = [sfl.Solver(dim, dim, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        for s in slabs:
            s.upload(sfl.capi.FIELD_DIVERGENCE, d[s.row_begin:s.row_end])
        slabs[0].poisson_solve(1.0, iters, OMEGA)
        slabs[0].synchronize()
        info = slabs[5].last_solve_info()
        for s in slabs:     # slab by slab: no second 1 GiB copy of the field on the host
            assert_bit_equal(s.download(sfl.capi.FIELD_PRESSURE), want[s.row_begin:s.row_end],
                             f"C5 in full: slab {s.rank} of 8")
    finally:
        for s in slabs:
            s.close()
    assert info["fuse"] == 16 and info["launches"] == 25
    assert info["exchanges"] == plan_exchanges(sfl, dim, nranks, iters, 16, info["halo"])


def test_baseline_config4_whole_sim_step_on_eight_virtual_ranks(sfl, oracle):
    """BASELINE config 4's WHOLE sim step (ino:252-287) at full size: 8192 x 8192, 80 SOR iterations, eight 1024-row
    slabs on eight virtual ranks, everything on auto (automatic advection halo: measured for the first step, known
    for the second; dye halo sent early; solve with a one-row tail), two steps, with drag forces thrown onto and
    next to two cuts before the second one (ino:264-269).  Every cell of every field against the oracle."""
    dim, iters, nranks = 8192, 80, 8
    rng = np.random.default_rng(84)
    v = (rng.uniform(-1, 1, (dim, dim, 2)) * 100).astype(np.float32)
    c = rng.integers(0, 2 ** 31, (dim, dim, 3), dtype=np.uint32)
    # drags in the sketch's graphics coordinates: (coords.x = j, coords.y = i, velocity.x -> v.y, velocity.y -> v.x)
    drags = [(1023, 77, 250.0, -40.0), (1024, 78, -300.0, 55.0), (1025, 4000, 90.0, 90.0), (6143, 8191, -120.0, 10.0),
             (6144, 0, 400.0, -400.0), (5000, 5000, 33.0, -66.0)]
    slabs = [sfl.Solver(dim, dim, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        for s in slabs:
            s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
            s.upload(sfl.capi.FIELD_COLOR, c[s.row_begin:s.row_end])
        vo, co = v, c
        for step in range(2):
            forces = None
            if step == 1:
                slabs[0].queue_drags(drags)     # (a linked group shares one queue; RCCL ranks each queue the same list)
                forces = ([(y, x) for x, y, _, _ in drags], [(vy, vx) for _, _, vx, vy in drags])
            slabs[0].step(DT, 1.0, iters, OMEGA)
            vo, do, po, co = oracle_step(oracle, vo, co, iters, forces)
        slabs[0].synchronize()
        for s in slabs:
            rows = slice(s.row_begin, s.row_end)
            assert_bit_equal(s.download(sfl.capi.FIELD_VELOCITY), vo[rows], f"velocity, slab {s.rank}")
            assert_bit_equal(s.download(sfl.capi.FIELD_DIVERGENCE), do[rows], f"divergence, slab {s.rank}")
            assert_bit_equal(s.download(sfl.capi.FIELD_PRESSURE), po[rows], f"pressure, slab {s.rank}")
            assert_bit_equal(s.download(sfl.capi.FIELD_COLOR), co[rows], f"colour, slab {s.rank}")
    finally:
        for s in slabs:
            s.close()


def test_drag_messages_take_the_sketchs_transform(sfl, oracle):
    """sfl_queue_drags: the sketch's struct drag (ino:45-48) with the x / y swap of ino:264-269 done by the library --
    identical to sfl_queue_forces with the swap done by hand, and to the oracle; out-of-domain coordinates refused."""
    dim_x, dim_y, iters = 61, 81, 6
    v, c, _ = random_fields(dim_x, dim_y, 45, 40.0)
    drags = [(10, 3, 5.5, -7.25), (80, 60, -100.0, 3.0), (0, 0, 1.0, 2.0), (10, 3, 9.0, 9.5)]   # (coords.x, coords.y, vel.x, vel.y)
    cells = [(y, x) for x, y, _, _ in drags]
    vels = [(vy, vx) for _, _, vx, vy in drags]
    want = oracle_step(oracle, v, c, iters, (cells, vels))
    for small in (1, 0):
        with sfl.Solver(dim_x, dim_y) as s:
            s.set_option(sfl.capi.OPT_SMALL_GRID, small)
            s.upload(sfl.capi.FIELD_VELOCITY, v)
            s.upload(sfl.capi.FIELD_COLOR, c)
            s.queue_drags(drags)
            s.step(DT, 1.0, iters, OMEGA)
            s.synchronize()
            assert_bit_equal(s.download(sfl.capi.FIELD_VELOCITY), want[0], f"velocity (small grid {small})")
            assert_bit_equal(s.download(sfl.capi.FIELD_COLOR), want[3], f"colour (small grid {small})")
            with pytest.raises(sfl.SflError) as e:
                s.queue_drags([(dim_y, 0, 1.0, 1.0)])       # coords.x addresses j: one past the last row
            assert e.value.code == sfl.capi.ERR_INVALID
            with pytest.raises(sfl.SflError):
                s.queue_drags([(0, dim_x, 1.0, 1.0)])


def test_device_pointer_counts_as_a_write_from_outside(sfl, oracle):
    """sfl_field_device_ptr hands out a WRITABLE pointer.  A velocity written through it must not be
    advected on the halo / reach the library knew for the previous field: four virtual ranks step once (the reach of
    the slow field becomes 'known'), then a jet crossing the cuts is written through the pointers, and the next step
    must still match the oracle."""
    import ctypes as C
    hipr = C.CDLL("libamdhip64.so")
    hipr.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    dim_x, dim_y, iters, nranks = 96, 256, 5, 4
    v, c, _ = random_fields(dim_x, dim_y, 46, 20.0)
    jet = v.copy()
    jet[..., 1] = 30.0 * 11.5          # 11.5 rows per step: far beyond the reach of the first field (< 1 row)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        for s in slabs:
            s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
            s.upload(sfl.capi.FIELD_COLOR, c[s.row_begin:s.row_end])
        slabs[0].step(DT, 1.0, iters, OMEGA)
        vo, do, po, co = oracle.step(v, c, DT, 1.0, iters, OMEGA)
        slabs[0].synchronize()
        for s in slabs:
            part = np.ascontiguousarray(jet[s.row_begin:s.row_end])
            assert hipr.hipMemcpy(s.device_ptr(sfl.capi.FIELD_VELOCITY), part.ctypes.data, part.nbytes, 1) == 0   # H2D
        slabs[0].step(DT, 1.0, iters, OMEGA)
        slabs[0].synchronize()         # no SFL_ERR_HALO either
        want = oracle.step(jet, co, DT, 1.0, iters, OMEGA)
        cat = lambda f: np.concatenate([s.download(f) for s in slabs], axis=0)
        assert_bit_equal(cat(sfl.capi.FIELD_VELOCITY), want[0], "velocity after the write through the pointer")
        assert_bit_equal(cat(sfl.capi.FIELD_COLOR), want[3], "colour after the write through the pointer")
    finally:
        for s in slabs:
            s.close()


@pytest.mark.parametrize("dim_x,dim_y,vamp,n", [(256, 192, 60.0, 3), (300, 141, 900.0, 4), (1000, 333, 150.0, 2), (129, 130, 5.0, 5)])
def test_step_n_fuses_across_step_boundaries_with_the_same_bits(sfl, oracle, dim_x, dim_y, vamp, n):
    """sfl_step_n (the sim task's loop, ino:249-289): between two steps subtract_gradient + dye advection of one and
    velocity advection + divergence of the next run as ONE kernel; the projected velocity in between never reaches
    memory.  Must equal n calls of sfl_step and the oracle's n steps bit for bit -- slow and fast fields (back-traces that
    leave the LDS windows project their texels on the fly), ragged tile edges, a drag force in the first step."""
    v, c, _ = random_fields(dim_x, dim_y, dim_x + n, vamp)
    iters = 7
    cells = np.array([[dim_x // 3, dim_y // 2], [dim_x - 1, 0]], np.int32)
    fv = np.array([[35.0, -22.0], [4.0, 9.0]], np.float32)
    fields = (sfl.capi.FIELD_VELOCITY, sfl.capi.FIELD_DIVERGENCE, sfl.capi.FIELD_PRESSURE, sfl.capi.FIELD_COLOR)
    got = {}
    for mode in ("step_n", "separate", "seams_off"):
        with sfl.Solver(dim_x, dim_y) as s:
            assert s.get_option(sfl.capi.OPT_STEP_SEAMS) == 1
            if mode == "seams_off":
                s.set_option(sfl.capi.OPT_STEP_SEAMS, 0)
            s.upload(sfl.capi.FIELD_VELOCITY, v)
            s.upload(sfl.capi.FIELD_COLOR, c)
            s.queue_forces(cells, fv)
            if mode == "separate":
                for _ in range(n):
                    s.step(DT, 1.0, iters, OMEGA)
            else:
                s.step_n(n, DT, 1.0, iters, OMEGA)
            s.synchronize()
            got[mode] = [s.download(f) for f in fields]
    for name, a, b, cc in zip(("v", "div", "p", "colour"), got["step_n"], got["separate"], got["seams_off"]):
        assert_bit_equal(a, b, f"step_n vs {n} x step: {name}")
        assert_bit_equal(a, cc, f"step_n with and without seams: {name}")
    vo, co = v, c
    for k in range(n):
        va = oracle.advect_vec2f(vo, vo, DT, True)
        if k == 0:
            for (i, j), u in zip(cells, fv):
                va[j, i] = u
        do = oracle.divergence(va, 1.0)
        po = oracle.poisson_solve(do, 1.0, iters, OMEGA)
        vo = oracle.subtract_gradient(va, po, 1.0)
        co = oracle.advect_vec3uq32(co, vo, DT, False)
    for name, a, b in zip(("v", "div", "p", "colour"), got["step_n"], (vo, do, po, co)):
        assert_bit_equal(a, b, f"step_n vs oracle: {name}")


def test_step_n_where_there_is_nothing_to_fuse(sfl, oracle):
    """n = 0 / 1, the one-workgroup small-grid path and slab groups: sfl_step_n is n times sfl_step."""
    v, c, _ = random_fields(61, 81, 9, 40.0)
    with sfl.Solver(61, 81) as s:
        s.upload(sfl.capi.FIELD_VELOCITY, v)
        s.upload(sfl.capi.FIELD_COLOR, c)
        s.step_n(0, DT, 1.0, 5, OMEGA)
        s.synchronize()
        assert_bit_equal(s.download(sfl.capi.FIELD_VELOCITY), v, "n = 0 leaves the fields alone")
        s.step_n(3, DT, 1.0, 5, OMEGA)
        s.synchronize()
        vo, co = v, c
        for _ in range(3):
            vo, do, po, co = oracle.step(vo, co, DT, 1.0, 5, OMEGA)
        assert_bit_equal(s.download(sfl.capi.FIELD_VELOCITY), vo, "small grid, 3 steps")
        assert_bit_equal(s.download(sfl.capi.FIELD_COLOR), co, "small grid, 3 steps: dye")
        with pytest.raises(sfl.SflError):
            s.step_n(-1, DT, 1.0, 5, OMEGA)
    dim_x, dim_y, nranks = 192, 256, 2
    v, c, _ = random_fields(dim_x, dim_y, 10, 40.0)
    slabs = [sfl.Solver(dim_x, dim_y, 0, r, nranks) for r in range(nranks)]
    try:
        sfl.Solver.link_group(slabs)
        for s in slabs:
            s.upload(sfl.capi.FIELD_VELOCITY, v[s.row_begin:s.row_end])
            s.upload(sfl.capi.FIELD_COLOR, c[s.row_begin:s.row_end])
        slabs[0].step_n(2, DT, 1.0, 6, OMEGA)
        slabs[0].synchronize()
        vo, co = v, c
        for _ in range(2):
            vo, do, po, co = oracle.step(vo, co, DT, 1.0, 6, OMEGA)
        assert_bit_equal(np.concatenate([s.download(sfl.capi.FIELD_VELOCITY) for s in slabs], axis=0), vo, "slabs, 2 steps")
        assert_bit_equal(np.concatenate([s.download(sfl.capi.FIELD_COLOR) for s in slabs], axis=0), co, "slabs, 2 steps: dye")
    finally:
        for s in slabs:
            s.close()
