"""The C++ drop-in surface (include/sfl/*.h + libsfl_dropin.so).

CPU: the headers compile stand-alone with g++, keep the reference's element layout, reject
element types without a GPU kernel, and the reference-style caller (tests/cpp/dropin_loop.cpp,
written like ino:249-289) builds against them.
GPU: that caller runs and reproduces the oracle bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_bit_equal

INC = os.path.join(ROOT, "include")
CPP = os.path.join(ROOT, "tests", "cpp")


def _gxx(src, tmp_path, ok=True):
    f = tmp_path / "t.cpp"
    f.write_text(src)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", INC, str(f)],
                       capture_output=True, text=True)
    assert (r.returncode == 0) == ok, r.stderr
    return r.stderr


def test_headers_compile_and_keep_reference_layout(tmp_path):
    _gxx('''
#include "sfl/vector.h"
#include "sfl/uq32.h"
#include "sfl/operations.h"
#include "sfl/advect.h"
#include "sfl/finitediff.h"
#include "sfl/poisson.h"
static_assert(sizeof(Vector2<float>) == 8 && sizeof(Vector3<UQ32>) == 12 && sizeof(Vector3<float>) == 12, "");
static float twice(float *c, int, int, int, int, void *) { return 2 * *c; }
int main() {
    Vector2<float> a(1, 2), b(a * 2.0f);            // promotion + converting ctor
    Vector3<UQ32> c(Vector3<float>(1.4f, 1.5f, 2.6f));
    Vector3<float> w = 0.25f * c + 0.5f * c;        // ino:227 style
    a -= b; a = -a + b / 2.0f;
    float f[6] = {1, 2, 3, 4, 5, 6}, g[6];
    domain_iter<float, float>(twice, twice, g, f, 3, 2, nullptr);
    kernel_func_t<float, float> k = twice; (void)k; (void)w;
    return index(1, 1, 3) == 4 && c.y.raw == 2 ? 0 : 1;
}''', tmp_path)


def test_advect_accepts_every_reference_element_type_and_rejects_unknown_ones(tmp_path):
    """advect<T, U> (advect.h:74-85): every element type the reference's headers can express compiles with a plain
    C++ compiler (kernels inside the library); a type nobody has seen needs hipcc (a kernel instantiated from the
    header) and is a compile error -- never a CPU loop -- elsewhere."""
    _gxx('''
#include "sfl/advect.h"
void f(Vector2<float> *v, float *a, UQ32 *b, Vector2<float> *c, Vector2<UQ32> *d, Vector3<float> *e, Vector3<UQ32> *g) {
    advect(a, a + 16, v, 4, 4, 0.1f, true);  advect(b, b + 16, v, 4, 4, 0.1f, false);
    advect(c, c + 16, v, 4, 4, 0.1f, true);  advect(d, d + 16, v, 4, 4, 0.1f, false);
    advect(e, e + 16, v, 4, 4, 0.1f, true);  advect(g, g + 16, v, 4, 4, 0.1f, false);
}''', tmp_path)
    err = _gxx('''
#include "sfl/advect.h"
void f(Vector2<double> *a, Vector2<double> *b, Vector2<float> *v) { advect(a, b, v, 4, 4, 0.1f, true); }
''', tmp_path, ok=False)
    assert "compile the" in err and "no CPU" in err
    err = _gxx('''
#include "sfl/advect.h"
void f(float *a, float *b, Vector2<double> *v) { advect(a, b, v, 4, 4, 0.1f, true); }
''', tmp_path, ok=False)
    assert "compile the" in err


def _build_advect_driver(tmp_path, compiler):
    lib = os.path.join(ROOT, "esp32-fluid-simulation_amd", "lib")
    subprocess.run(["make", "-C", os.path.join(ROOT, "esp32-fluid-simulation_amd", "host")], check=True,
                   stdout=subprocess.DEVNULL)
    exe = tmp_path / ("advect_generic_" + compiler)
    src = os.path.join(CPP, "advect_generic_driver.cpp")
    tail = ["-std=c++17", "-ffp-contract=off", "-I", os.path.join(INC, "sfl"), "-I", INC, src, "-o", str(exe),
            "-L", lib, "-lsfl_dropin", "-lsfl_hip", "-Wl,-rpath," + lib]
    if compiler == "hipcc":
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-x", "hip", "-O2", "-DDRIVER_ANY_TYPE"] + tail
    else:
        cmd = ["g++", "-O1"] + tail
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return exe


def test_generic_advect_callers_build(tmp_path):
    """The reference-style caller of advect<T, U> for other element types links against the drop-in library from a
    plain C++ compiler, and -- with types of its own -- from hipcc."""
    _build_advect_driver(tmp_path, "g++")
    _build_advect_driver(tmp_path, "hipcc")


def test_generic_advect_fixture_is_what_the_reference_prints():
    m = _header_goldens()
    if not os.path.isdir(m.REF):
        pytest.skip("needs /root/reference")
    want = open(os.path.join(ROOT, "tests", "golden", "advect_generic_reference.txt")).read()
    assert m.run_driver(m.REF, "advect_generic_driver.cpp", ("DRIVER_ANY_TYPE",)) == want


@pytest.mark.gpu
@pytest.mark.parametrize("compiler", ["g++", "hipcc"])
def test_generic_advect_runs_on_gpu_from_reference_style_callers(tmp_path, compiler):
    """advect<T, U> for T = float, UQ32, Vector2<UQ32>, Vector3<float> from a plain C++ caller (library kernels), and
    additionally Vector2<double> elements / a Vector2<double> velocity from a hipcc caller (kernel instantiated from
    include/sfl/advect.h): every printed bit equals what the reference's own template printed as a host loop
    (tests/golden/advect_generic_reference.txt, written by make_header_goldens.py)."""
    exe = _build_advect_driver(tmp_path, compiler)
    got = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines()
    want = open(os.path.join(ROOT, "tests", "golden", "advect_generic_reference.txt")).read().splitlines()
    if compiler == "hipcc":
        assert got == want
    else:
        kept = [l for l in want if not (l.startswith("Vector2<double>") or l.startswith("float|"))]
        assert len(kept) == 3 * (1 + 8) and got == kept


def test_uq32_rounding_matches_reference_semantics(tmp_path):
    src = tmp_path / "u.cpp"
    src.write_text('''
#include <cstdio>
#include "sfl/uq32.h"
int main() { float xs[] = {0.f, 0.49f, 0.5f, 1.5f, 2.5f, 8388609.f, 16777216.f, 2147483648.f};
  for (float x : xs) { UQ32 u(x); std::printf("%u %.1f\\n", u.raw, (double)float(u)); } }''')
    exe = tmp_path / "u"
    subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", INC, str(src), "-o", str(exe)], check=True)
    got = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    raws = [int(x) for x in got[0::2]]
    # +0.5f then truncate: 2.5 -> 3, 8388609 + 0.5 ties to even 8388610
    assert raws == [0, 0, 1, 2, 3, 8388610, 16777216, 2147483648]


def test_reference_style_caller_builds():
    subprocess.run(["make", "-C", os.path.join(ROOT, "esp32-fluid-simulation_amd", "host")], check=True,
                   stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CPP, "dropin"], check=True, stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(CPP, "dropin_loop"))


@pytest.mark.gpu
@pytest.mark.parametrize("dim_x,dim_y,iters,steps", [(61, 81, 10, 3), (128, 96, 7, 2)])
def test_reference_style_caller_runs_on_gpu(tmp_path, oracle, dim_x, dim_y, iters, steps):
    subprocess.run(["make", "-C", os.path.join(ROOT, "esp32-fluid-simulation_amd", "host")], check=True,
                   stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CPP, "dropin"], check=True, stdout=subprocess.DEVNULL)
    v, c = oracle.lcg_fields(dim_x, dim_y, 4321, 90.0)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("3i", dim_x, dim_y, iters))
        f.write(v.tobytes())
        f.write(c.tobytes())
    subprocess.run([os.path.join(CPP, "dropin_loop"), str(fin), str(steps), str(fout)], check=True)
    raw = open(fout, "rb").read()
    n = dim_x * dim_y
    got_v = np.frombuffer(raw, np.float32, 2 * n, 0).reshape(dim_y, dim_x, 2)
    got_d = np.frombuffer(raw, np.float32, n, 8 * n).reshape(dim_y, dim_x)
    got_p = np.frombuffer(raw, np.float32, n, 12 * n).reshape(dim_y, dim_x)
    got_c = np.frombuffer(raw, np.uint32, 3 * n, 16 * n).reshape(dim_y, dim_x, 3)
    dt, omega = np.float32(1 / 30.0), np.float32(1.96)
    for _ in range(steps):
        v, d, p, c = oracle.step(v, c, dt, 1.0, iters, omega)
    for name, a, b in (("v", got_v, v), ("div", got_d, d), ("p", got_p, p), ("colour", got_c, c)):
        assert_bit_equal(a, b, f"drop-in caller: {name}")


def _build_domain_for_each(tmp_path):
    exe = tmp_path / "dfe"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17",
                    "-ffp-contract=off", "-I", INC, os.path.join(CPP, "domain_for_each_test.hip"),
                    "-o", str(exe)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return exe


def test_domain_for_each_caller_builds(tmp_path):
    _build_domain_for_each(tmp_path)


@pytest.mark.gpu
def test_domain_for_each_runs_user_expressions_on_the_device(tmp_path, oracle):
    """user safe/fast functors through sfl/operations.h::domain_for_each (out of place, T != U)."""
    exe = _build_domain_for_each(tmp_path)
    dim_x, dim_y = 97, 45
    v = (np.random.default_rng(1).uniform(-1, 1, (dim_y, dim_x, 2)) * 30).astype(np.float32)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("2i", dim_x, dim_y))
        f.write(v.tobytes())
    subprocess.run([str(exe), "div", str(fin), str(fout)], check=True)
    got = np.fromfile(fout, np.float32).reshape(dim_y, dim_x)
    assert_bit_equal(got, oracle.divergence(v, 1.0), "domain_for_each divergence")


@pytest.mark.gpu
def test_domain_for_each_in_place_leaves_the_reference_orders_bits(tmp_path):
    """operations.h:11-38 allows wrt == rd (finitediff.cpp:80 uses it).  Order-SENSITIVE expressions -- every cell reads
    its four neighbours and overwrites the centre -- run in place on the device and must leave exactly what the
    reference's visiting order leaves: case A of tests/golden/domain_iter_reference.txt (written by the reference's own
    operations.h), then larger grids against the host domain_iter of include/sfl (itself pinned to that fixture)."""
    exe = _build_domain_for_each(tmp_path)
    got = subprocess.run([str(exe), "inplace"], capture_output=True, text=True, check=True).stdout.splitlines()
    want = [l for l in open(os.path.join(ROOT, "tests", "golden", "domain_iter_reference.txt")).read().splitlines()
            if l.startswith("A ")]
    strip_calls = lambda l: " ".join(t for t in l.split() if not t.startswith("calls="))
    a_lines = [l for l in got if l.startswith("A ")]
    assert len(a_lines) == len(want) == 6 and a_lines == [strip_calls(l) for l in want]
    b_lines = [l for l in got if l.startswith("B ")]
    assert len(b_lines) == 3 and all(l.endswith(" same") for l in b_lines), b_lines


@pytest.mark.gpu
def test_domain_for_each_pointwise_in_place_and_red_black(tmp_path, oracle):
    """The two in-place uses the reference itself makes of its drivers, as USER functors on the device:
    subtract_gradient's expressions over the velocity in place (finitediff.cpp:80; they read only the centre of the field
    they rewrite: every cell at once), and the SOR expressions through the colour-split driver (poisson.cpp:14-61,
    :121-124), both against the oracle."""
    exe = _build_domain_for_each(tmp_path)
    dim_x, dim_y, iters = 131, 77, 9
    rng = np.random.default_rng(5)
    v = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * 30).astype(np.float32)
    p = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("2i", dim_x, dim_y))
        f.write(v.tobytes())
        f.write(p.tobytes())
    subprocess.run([str(exe), "pointwise", str(fin), str(fout)], check=True)
    assert_bit_equal(np.fromfile(fout, np.float32).reshape(dim_y, dim_x, 2), oracle.subtract_gradient(v, p, 1.0),
                     "in-place gradient subtraction by user functors")
    for dx_, dy_ in ((dim_x, dim_y), (2, 2), (3, 2), (64, 5)):
        d = rng.standard_normal((dy_, dx_)).astype(np.float32)
        with open(fin, "wb") as f:
            f.write(struct.pack("3i", dx_, dy_, iters))
            f.write(d.tobytes())
        subprocess.run([str(exe), "redblack", str(fin), str(fout)], check=True)
        assert_bit_equal(np.fromfile(fout, np.float32).reshape(dy_, dx_), oracle.poisson_solve(d, 1.0, iters, np.float32(1.96)),
                         f"red-black SOR by user functors {dx_}x{dy_}")


@pytest.mark.gpu
def test_c_abi_context_loop_with_forces(tmp_path, oracle):
    """INTEGRATION.md section 3 as a plain C-ABI program: resident fields, queued forces, sfl_step."""
    subprocess.run(["make", "-C", CPP, "dropin"], check=True, stdout=subprocess.DEVNULL)
    dim_x, dim_y, iters, steps = 96, 72, 9, 3
    v, c = oracle.lcg_fields(dim_x, dim_y, 99, 50.0)
    cells = np.array([[5, 6], [50, 40], [95, 71]], np.int32)
    fv = np.array([[30.0, -20.0], [-15.5, 8.25], [3.0, 4.0]], np.float32)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("4i", dim_x, dim_y, iters, len(cells)))
        f.write(v.tobytes()); f.write(c.tobytes()); f.write(cells.tobytes()); f.write(fv.tobytes())
    subprocess.run([os.path.join(CPP, "context_loop"), str(fin), str(steps), str(fout)], check=True)
    raw = open(fout, "rb").read()
    n = dim_x * dim_y
    got_v = np.frombuffer(raw, np.float32, 2 * n, 0).reshape(dim_y, dim_x, 2)
    got_p = np.frombuffer(raw, np.float32, n, 8 * n).reshape(dim_y, dim_x)
    got_c = np.frombuffer(raw, np.uint32, 3 * n, 12 * n).reshape(dim_y, dim_x, 3)
    dt, omega = np.float32(1 / 30.0), np.float32(1.96)
    for s in range(steps):
        va = oracle.advect_vec2f(v, v, dt, True)
        if s == 0:
            for (i, j), u in zip(cells, fv):
                va[j, i] = u
        d = oracle.divergence(va, 1.0)
        p = oracle.poisson_solve(d, 1.0, iters, omega)
        v = oracle.subtract_gradient(va, p, 1.0)
        c = oracle.advect_vec3uq32(c, v, dt, False)
    for name, a, b in (("v", got_v, v), ("p", got_p, p), ("colour", got_c, c)):
        assert_bit_equal(a, b, f"C-ABI context loop: {name}")


def _header_goldens():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_header_goldens",
                                                  os.path.join(ROOT, "tests", "golden", "make_header_goldens.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("driver,fixture", [("domain_iter_driver.cpp", "domain_iter_reference.txt"),
                                            ("sample_driver.cpp", "sample_reference.txt")])
def test_header_templates_executed_against_reference_bits(driver, fixture):
    """`domain_iter` (operations.h:11-38; in place with order-sensitive expressions,
    wrt == rd, and T != U) and the per-point helpers lerp / billinear_interpolate / sample
    (advect.h:10-72) are EXECUTED from include/sfl and must leave exactly the bits the reference's
    own headers leave (fixture generated from /root/reference by tests/golden/make_header_goldens.py;
    where the reference is present the comparison is also made live)."""
    m = _header_goldens()
    ours = m.run_driver(os.path.join(INC, "sfl"), driver)
    want = open(os.path.join(ROOT, "tests", "golden", fixture)).read()
    assert ours == want
    if os.path.isdir(m.REF):
        assert m.run_driver(m.REF, driver) == want
