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
    """tests/cpp/launch_stubs.cpp and launch_stubs_ok.cpp are generated from csrc/kernels.h: regenerating them must change nothing."""
    before = [open(os.path.join(CPP, f)).read() for f in ("launch_stubs.cpp", "launch_stubs_ok.cpp")]
    subprocess.run(["python3", os.path.join(CPP, "make_launch_stubs.py")], check=True, stdout=subprocess.DEVNULL)
    assert [open(os.path.join(CPP, f)).read() for f in ("launch_stubs.cpp", "launch_stubs_ok.cpp")] == before


def test_host_side_from_four_threads_under_tsan():
    """`make -C tests/cpp tsan_host` -- the host units over a HIP / RCCL runtime that lives on the host
    (tests/cpp/fake_hip.cpp) and kernels that do nothing (launch_stubs_ok.cpp), RUN from four threads at once under ThreadSanitizer:
    per thread a whole-domain context (options, I/O, steps, force queue), a linked group of 2 - 4 virtual ranks through every
    exchange schedule, one rank's program over the one-rank RCCL transport, and the host-pointer drop-ins with their per-thread
    context cache; all threads share the error state and the plan queries.  Clean -- and the harness is shown to see a race
    when there is one (two threads, one plain int)."""
    subprocess.run(["make", "-C", CPP, "tsan_host"], check=True, stdout=subprocess.DEVNULL)
    exe = os.path.join(CPP, "host_tsan_driver")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:exitcode=66:second_deadlock_stack=1")
    r = subprocess.run([exe, "4", "6"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert "0 failed checks, 0 allocations left" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
    r = subprocess.run([exe, "race"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 66 and "ThreadSanitizer: data race" in r.stderr


def test_host_side_runs_through_under_asan_over_the_fake_runtime():
    """The same driver under AddressSanitizer + UBSan: here the executors, groups and transports really execute (the launch stubs
    of test_host_side_under_asan_and_ubsan stop every operator at its first launch), and every fake-device allocation must be freed."""
    subprocess.run(["make", "-C", CPP, "asan_fake_host"], check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(CPP, "host_asan_fake_driver"), "2", "4"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert "0 failed checks, 0 allocations left" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-6000:]
