"""Golden-fixture access shared by the CPU (oracle) and GPU (HIP path) tests.

The fixtures under tests/golden/ were produced by the unmodified reference
(tests/golden/make_golden.py).  ``check_ops`` / ``check_steps`` run any implementation that
offers the CpuPath-style API (oracle.loader.CpuPath, or the product's host front-end) against
them, bit for bit.
"""
import glob
import os

import numpy as np

from conftest import GOLDEN, assert_bit_equal

OMEGA = np.float32(1.96)

# Known-answer table of SURVEY.md 8(c): FNV-1a-64 of v, div, p, colour after each step.
KAT = {
    (61, 81, 10, 100.0, 12345): ("cd1e03d6c9a0a0b8", "6665e58c0dd0dab8", [
        ("baa18c8a8c34b4d7", "805a21e6a784945d", "987f8336744d2a0a", "6c81d2eb22789652"),
        ("7d4d7efd9e76d762", "9feaea12937002be", "21c60aa076c17f11", "5c501141f058679f"),
        ("3a97d4fdc99527bc", "acd1fef290d79009", "cc82e7f546737583", "67b8891591384c00")]),
    (61, 81, 20, 100.0, 12345): ("cd1e03d6c9a0a0b8", "6665e58c0dd0dab8", [
        ("92d0fd5249368860", "805a21e6a784945d", "54f8b1f5d1e039f8", "efce214d8e8d6b86"),
        ("bc0c094dd73f2677", "df7c4ec281dadffc", "da361bfc821670ef", "bf56e9fc4e31109d"),
        ("ef23b6b5c12425a6", "fd05f31d832673e5", "6f4d896a89d2be75", "8b060d4ad36d1678")]),
    (256, 192, 40, 400.0, 2024): ("ca4ceb89a8ba480d", "44b5f5096fe5bf4f", [
        ("e9cbd207ac140157", "d18b037a4155cda1", "92c5142f8fc6a221", "150ce2bcf78048be"),
        ("a435dbadb914473c", "b0a8a6330c0478a2", "e6aeecfc41227351", "9dd34bdeeb08d065")]),
}


def fnv1a64(a: np.ndarray) -> str:
    """FNV-1a-64 of the raw bytes (pure numpy/Python; small arrays only)."""
    h = 1469598103934665603
    for b in np.ascontiguousarray(a).view(np.uint8).ravel().tolist():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def lcg_fields(dim_x, dim_y, seed, vamp):
    """Input recipe of SURVEY.md 8(c), vectorised in numpy (no oracle needed)."""
    n = dim_x * dim_y
    state = seed
    # 5 draws per cell, strictly sequential LCG
    a, c, m = 1664525, 1013904223, 0xFFFFFFFF
    out = []
    for _ in range(5 * n):
        state = (state * a + c) & m
        out.append(state)
    s = np.array(out, np.uint64).reshape(n, 5)
    comp = lambda col: (((s[:, col] >> np.uint64(8)) % np.uint64(2001)).astype(np.int64) - 1000
                        ).astype(np.float32) / np.float32(1000.0) * np.float32(vamp)
    v = np.stack([comp(0), comp(1)], axis=1).astype(np.float32).reshape(dim_y, dim_x, 2)
    col = (s[:, 2:5] >> np.uint64(1)).astype(np.uint32).reshape(dim_y, dim_x, 3)
    return v, col


def step_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "step_*.npz")))


def ops_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "ops_*.npz")))


def channel_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "advc_*.npz")))


def check_channels(impl, path):
    """advect<T, float> for the element types besides the sketch's two (fixture written by the reference)."""
    z = np.load(path)
    tag = os.path.basename(path)
    n = 0
    for key in z.files:
        if not key.startswith("in_"):
            continue
        for ns in (0, 1):
            assert_bit_equal(impl.advect_channels(z[key], z["v"], z["dt"], bool(ns)), z[f"out_{key[3:]}_ns{ns}"],
                             f"{tag} {key[3:]} ns{ns}")
            n += 1
    assert n == 8


def check_steps(impl, path):
    z = np.load(path)
    dim_x, dim_y, iters, nsteps = (int(x) for x in z["meta"])
    v, c = z["v0"], z["c0"]
    assert v.shape == (dim_y, dim_x, 2)
    for s in range(1, nsteps + 1):
        v, d, p, c = impl.step(v, c, z["dt"], 1.0, iters, z["omega"])
        tag = f"{os.path.basename(path)} step {s}"
        assert_bit_equal(v, z[f"v{s}"], tag + " v")
        assert_bit_equal(d, z[f"div{s}"], tag + " div")
        assert_bit_equal(p, z[f"p{s}"], tag + " p")
        assert_bit_equal(c, z[f"c{s}"], tag + " colour")


def check_ops(impl, path):
    z = np.load(path)
    v, q, c, s, dt = z["v"], z["q"], z["c"], z["s"], z["dt"]
    tag = os.path.basename(path)
    for ns in (0, 1):
        assert_bit_equal(impl.advect_vec2f(v, v, dt, bool(ns)), z[f"adv2_self_ns{ns}"],
                         f"{tag} adv2 self ns{ns}")
        assert_bit_equal(impl.advect_vec2f(q, v, dt, bool(ns)), z[f"adv2_other_ns{ns}"],
                         f"{tag} adv2 other ns{ns}")
        assert_bit_equal(impl.advect_vec3uq32(c, v, dt, bool(ns)), z[f"adv3_ns{ns}"],
                         f"{tag} adv3 ns{ns}")
    for t, dx in (("dx1", 1.0), ("dx05", 0.5)):
        assert_bit_equal(impl.divergence(v, dx), z[f"div_{t}"], f"{tag} div {t}")
        assert_bit_equal(impl.subtract_gradient(v, s, dx), z[f"grad_{t}"], f"{tag} grad {t}")
    assert_bit_equal(impl.poisson_solve(s, 1.0, 1, OMEGA), z["pois_i1"], f"{tag} pois i1")
    assert_bit_equal(impl.poisson_solve(s, 1.0, 8, OMEGA), z["pois_i8"], f"{tag} pois i8")
    assert_bit_equal(impl.poisson_solve(s, 0.5, 5, np.float32(1.5)), z["pois_i5_w15_dx05"],
                     f"{tag} pois i5 w1.5 dx0.5")
