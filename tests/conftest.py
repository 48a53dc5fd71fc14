"""Shared test plumbing.

Markers: ``gpu`` = needs a real MI355X (run with ``-m gpu`` on the GPU box); everything
else must pass on a CPU-only container (``-m "not gpu"``).
"""
import importlib
import os
import sys

import numpy as np
import pytest

# (The suite runs under the runtime's own defaults -- GPU_MAX_HW_QUEUES = 4 in particular, as the product does.)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real AMD GPU (MI355X)")


def _ensure_built():
    """Build whatever is missing (fresh clone): the product library (hipcc cross-compiles gfx950
    without a GPU), the C++ host layer and the oracle.  No-ops when present.  (The CPU wave emulator is built by
    the fixture of tests/test_sor_stream_emulation.py: it does not travel to the GPU box, .gpurunignore.)"""
    import subprocess
    pkg = os.path.join(ROOT, "esp32-fluid-simulation_amd")
    jobs = [
        (os.path.join(pkg, "lib", "libsfl_hip.so"), ["make", "-C", os.path.join(pkg, "csrc"), "-j6"]),
        (os.path.join(pkg, "lib", "libsfl_dropin.so"), ["make", "-C", os.path.join(pkg, "host")]),
        (os.path.join(ROOT, "oracle", "libsf_oracle.so"), ["make", "-C", os.path.join(ROOT, "oracle")]),
    ]
    for artefact, cmd in jobs:
        if not os.path.exists(artefact):
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)


_ensure_built()


@pytest.fixture(scope="session")
def oracle():
    """The C restatement of the reference CPU path (checker only)."""
    from oracle import loader
    return loader.port()


@pytest.fixture(scope="session")
def reference():
    """The unmodified reference compiled in place; skipped where it was never built."""
    from oracle import loader
    if not loader.reference_available():
        pytest.skip("oracle/_ref/libsf_ref.so not built (needs /root/reference)")
    return loader.reference()


@pytest.fixture(scope="session")
def sfl():
    """The product package (HIP path behind the C ABI)."""
    return importlib.import_module("esp32-fluid-simulation_amd")


def bits(a: np.ndarray) -> np.ndarray:
    """View a float32 / uint32 array as uint32 for bit-exact comparison."""
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_bit_equal(got: np.ndarray, want: np.ndarray, what: str = ""):
    assert got.shape == want.shape, (what, got.shape, want.shape)
    g, w = bits(got), bits(want)
    if not np.array_equal(g, w):
        bad = np.argwhere(g != w)
        first = tuple(bad[0])
        raise AssertionError(
            f"{what}: {len(bad)} of {g.size} elements differ bitwise; first at {first}: "
            f"got {got[first]!r} (0x{int(g[first]):08x}) want {want[first]!r} (0x{int(w[first]):08x})")


def random_fields(dim_x, dim_y, seed, vamp=100.0, cmax=2 ** 31):
    """Seeded velocity / dye / scalar fields (numpy RNG; not the LCG of SURVEY 8c)."""
    rng = np.random.default_rng(seed)
    v = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * vamp).astype(np.float32)
    c = rng.integers(0, cmax, (dim_y, dim_x, 3), dtype=np.uint32)
    s = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
    return v, c, s
