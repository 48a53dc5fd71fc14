"""CPU suite: the HOST side of the product library under AddressSanitizer + UBSan.

`make -C tests/cpp san_host` compiles csrc/slab_plan.cpp and the host units of the C ABI (contexts, options, plans,
transports, executors, drop-ins) as plain C++ with -fsanitize=address,undefined, links them with stubs for the gfx950 kernels
(tests/cpp/launch_stubs.cpp: every launcher reports "no device") and runs tests/cpp/host_san_driver.cpp: a seeded fuzz of
sfl_plan_poisson / sfl_plan_poisson_tail / sfl_slab_rows / sfl_sor_pass_plan over (dim_y, nranks, iters, fuse, kernel, halo,
tail) on every rank -- slabs partition the rows, every rank's program has the same shape, launches cover the owned rows and
stay inside the ghost rows, no launch reads ghost rows that no exchange or earlier launch left exact, capacity-limited writes
stay inside the caller's array -- and the argument checks of the ABI.  The first run found a real one: the early-exchange
plan indexed empty tables at n - 1 for iters == 0 (csrc/slab_plan.cpp)."""
import os
import subprocess

from conftest import ROOT

CPP = os.path.join(ROOT, "tests", "cpp")


def test_host_side_under_asan_and_ubsan():
    subprocess.run(["make", "-C", CPP, "-j4", "san_host"], check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(CPP, "host_san_driver"), "2500"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "0 failed checks" in r.stdout and "launches checked" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    launches = int(r.stdout.split("(")[1].split()[0])
    assert launches > 10000      # the ghost-row validity simulation really ran


def test_launch_stubs_match_the_launch_interface():
    """tests/cpp/launch_stubs.cpp is generated from csrc/kernels.h: regenerating it must change nothing."""
    before = open(os.path.join(CPP, "launch_stubs.cpp")).read()
    subprocess.run(["python3", os.path.join(CPP, "make_launch_stubs.py")], check=True, stdout=subprocess.DEVNULL)
    assert open(os.path.join(CPP, "launch_stubs.cpp")).read() == before
