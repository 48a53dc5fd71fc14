"""Pin the oracle: oracle/sf_oracle.c must agree BIT FOR BIT with the unmodified reference
sources compiled in place (oracle/_ref, built by oracle/Makefile from /root/reference).

Runs only where oracle/_ref exists (this container, or a box that received the prebuilt
file); the committed fixtures in tests/golden/ carry the same pin everywhere else
(tests/test_golden.py).
"""
import numpy as np
import pytest

from conftest import assert_bit_equal, random_fields

SHAPES = [(2, 2), (2, 3), (3, 2), (3, 3), (4, 4), (5, 4), (17, 33), (33, 17), (61, 81), (64, 48),
          (128, 7), (7, 128)]
DT = np.float32(1 / 30.0)


@pytest.mark.parametrize("dim_x,dim_y", SHAPES)
@pytest.mark.parametrize("no_slip", [True, False])
def test_advect_vec2f(oracle, reference, dim_x, dim_y, no_slip):
    for seed, vamp in [(1, 100.0), (2, 5.0), (3, 1000.0), (4, 0.0)]:
        v, _, _ = random_fields(dim_x, dim_y, seed, vamp)
        q, _, _ = random_fields(dim_x, dim_y, seed + 100, 3.0)
        assert_bit_equal(oracle.advect_vec2f(v, v, DT, no_slip),
                         reference.advect_vec2f(v, v, DT, no_slip), "self-advection")
        assert_bit_equal(oracle.advect_vec2f(q, v, DT, no_slip),
                         reference.advect_vec2f(q, v, DT, no_slip), "advect other field")


@pytest.mark.parametrize("dim_x,dim_y", SHAPES)
@pytest.mark.parametrize("no_slip", [True, False])
def test_advect_vec3uq32(oracle, reference, dim_x, dim_y, no_slip):
    # raw < 2^31 keeps every float->uint32 conversion inside defined behaviour
    for seed, vamp, cmax in [(1, 100.0, 2 ** 31), (2, 5.0, 2 ** 24 + 7), (3, 300.0, 1000),
                             (4, 0.0, 2 ** 31)]:
        v, c, _ = random_fields(dim_x, dim_y, seed, vamp, cmax)
        assert_bit_equal(oracle.advect_vec3uq32(c, v, DT, no_slip),
                         reference.advect_vec3uq32(c, v, DT, no_slip), "dye advection")


def channel_field(rng, dim_x, dim_y, channels, uq, cmax=2 ** 31):
    """A field of one of the element types the reference's headers can express (float / UQ32, x 1..3)."""
    shape = (dim_y, dim_x) if channels == 1 else (dim_y, dim_x, channels)
    if uq:
        return rng.integers(0, cmax, shape, dtype=np.uint32)
    return (rng.standard_normal(shape) * 50).astype(np.float32)


@pytest.mark.parametrize("dim_x,dim_y", SHAPES)
@pytest.mark.parametrize("channels,uq", [(1, False), (1, True), (2, False), (2, True), (3, False), (3, True)])
def test_advect_every_element_type(oracle, reference, dim_x, dim_y, channels, uq):
    """advect<T, float> (advect.h:74-85) for T = float, UQ32, Vector2 / Vector3 of either: the oracle's
    channel-wise restatement against the reference's own template instantiations."""
    rng = np.random.default_rng(dim_x * 131 + dim_y * 7 + channels * 2 + uq)
    for vamp, cmax in [(100.0, 2 ** 31), (5.0, 2 ** 24 + 7), (1000.0, 1000), (0.0, 2 ** 31)]:
        v = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * vamp).astype(np.float32)
        q = channel_field(rng, dim_x, dim_y, channels, uq, cmax)
        for no_slip in (True, False):
            assert_bit_equal(oracle.advect_channels(q, v, DT, no_slip), reference.advect_channels(q, v, DT, no_slip),
                             f"{channels} x {'uq32' if uq else 'f32'}, no_slip {no_slip}")
    # the two instantiations of the sketch through the generic entry = through their own
    v, c, _ = random_fields(dim_x, dim_y, 3, 60.0)
    if channels == 2 and not uq:
        assert_bit_equal(oracle.advect_channels(v, v, DT, True), oracle.advect_vec2f(v, v, DT, True), "vec2f")
    if channels == 3 and uq:
        assert_bit_equal(oracle.advect_channels(c, v, DT, False), oracle.advect_vec3uq32(c, v, DT, False), "vec3uq32")


@pytest.mark.parametrize("dim_x,dim_y", SHAPES)
@pytest.mark.parametrize("dx", [1.0, 0.5, 3.0])
def test_divergence_and_gradient(oracle, reference, dim_x, dim_y, dx):
    v, _, s = random_fields(dim_x, dim_y, 11, 50.0)
    assert_bit_equal(oracle.divergence(v, dx), reference.divergence(v, dx), "divergence")
    assert_bit_equal(oracle.subtract_gradient(v, s, dx), reference.subtract_gradient(v, s, dx),
                     "subtract_gradient")


@pytest.mark.parametrize("dim_x,dim_y", SHAPES)
@pytest.mark.parametrize("iters,omega,dx", [(1, 1.96, 1.0), (2, 1.0, 1.0), (7, 1.5, 0.25),
                                            (20, 1.96, 1.0)])
def test_poisson_solve(oracle, reference, dim_x, dim_y, iters, omega, dx):
    _, _, s = random_fields(dim_x, dim_y, 5)
    assert_bit_equal(oracle.poisson_solve(s, dx, iters, np.float32(omega)),
                     reference.poisson_solve(s, dx, iters, np.float32(omega)), "poisson_solve")


def test_poisson_signed_zero_rhs(oracle, reference):
    # +0 / -0 right-hand sides exercise the 0 + x vs x association difference
    for fill in (0.0, -0.0):
        s = np.full((9, 8), fill, np.float32)
        s[3, 4] = 1.0
        assert_bit_equal(oracle.poisson_solve(s, 1.0, 3, np.float32(1.96)),
                         reference.poisson_solve(s, 1.0, 3, np.float32(1.96)), "signed zero")


@pytest.mark.parametrize("dim_x,dim_y,iters", [(61, 81, 10), (33, 17, 4), (2, 2, 3), (96, 64, 6)])
def test_full_step_sequence(oracle, reference, dim_x, dim_y, iters):
    v, c, _ = random_fields(dim_x, dim_y, 77, 100.0)
    vo, co = v, c
    vr, cr = v, c
    for _ in range(3):
        vo, do, po, co = oracle.step(vo, co, DT, 1.0, iters, np.float32(1.96))
        vr, dr, pr, cr = reference.step(vr, cr, DT, 1.0, iters, np.float32(1.96))
        for name, a, b in (("v", vo, vr), ("div", do, dr), ("p", po, pr), ("colour", co, cr)):
            assert_bit_equal(a, b, name)


def test_half_sweep_rows_compose_to_solve(oracle, reference):
    """The row-restricted colour pass (used by the slab tests) composes to poisson_solve."""
    dim_x, dim_y, iters = 19, 23, 5
    _, _, d = random_fields(dim_x, dim_y, 9)
    p = np.zeros((dim_y, dim_x), np.float32)
    for _ in range(iters):
        for colour in (0, 1):
            # three arbitrary row chunks, visited out of order: order inside a colour is free
            for a, b in ((10, 23), (0, 4), (4, 10)):
                oracle.sor_half_sweep_rows(p, d, dim_y, colour, a, b, 0)
    assert_bit_equal(p, reference.poisson_solve(d, 1.0, iters, np.float32(1.96)), "composed")
