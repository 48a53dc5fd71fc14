"""CPU suite: the multi-rank plumbing of bench.py -- the self-launcher (`python bench.py --gpus N`
as typed: the parent touches no GPU and starts N fresh rank processes), the torch.distributed.run
entry, and the TCP rendezvous the ranks use instead of torch.distributed (broadcast of the RCCL
unique id, barriers, max over ranks).  `--dry-run` ranks do everything except the GPU work."""
import importlib
import json
import multiprocessing as mp
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _rank_main(rank, world, key, q):
    sys.path.insert(0, ROOT)
    rdzv_mod = importlib.import_module("esp32-fluid-simulation_amd.rendezvous")
    r = rdzv_mod.Rendezvous(rank, world, key=key, timeout_s=60)
    try:
        blob = r.broadcast_bytes(bytes(range(128)) if rank == 0 else None)
        gathered = r.all_gather({"rank": rank, "sq": rank * rank})
        r.barrier()
        top = r.max([float(rank), -float(rank), 7.5])
        q.put((rank, blob == bytes(range(128)), gathered, top))
    finally:
        r.close()


@pytest.mark.parametrize("world", [1, 2, 4])
def test_rendezvous_all_gather_broadcast_max(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = f"pytest_{os.getpid()}_{world}"
    procs = [ctx.Process(target=_rank_main, args=(r, world, key, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, blob_ok, gathered, top in results:
        assert blob_ok
        assert gathered == [{"rank": r, "sq": r * r} for r in range(world)]
        assert top == [float(world - 1), 0.0, 7.5]
    rdzv_mod = importlib.import_module("esp32-fluid-simulation_amd.rendezvous")
    assert not os.path.exists(rdzv_mod.rendezvous_file(key))    # rank 0 removes the port file


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_self_launch_as_typed():
    """`python bench.py --gpus 3` without any launcher: three rank processes, ONE JSON line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run"],
                       capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr
    out = _json_line(r.stdout)
    # the headline is the LIBRARY'S OWN schedule (no flag); the in-time schedule runs afterwards as a labelled experiment
    assert out == {"dry_run": True, "n_gpus": 3, "max_rank": 2.0, "mode": "library default", "tokens_agree": True, "ranks": [0, 1, 2],
                   "devices": [0, 1, 2], "rccl_host_ids": [None, None, None], "exchange_mode": "library default", "fallback_from": [],
                   "in_time_experiment": {"dry_run": True, "mode": "in-time"}}


def test_share_device_gives_every_rank_the_device_and_a_host_id_of_its_own():
    """`--share-device D`: N rank processes on ONE device form a real N-rank RCCL communicator only if RCCL takes them for ranks
    on different hosts (it refuses duplicate devices of one host): every rank must carry its own NCCL_HOSTID and use device D."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--share-device", "0", "--dry-run"],
                       capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NCCL_HOSTID")})
    assert r.returncode == 0, r.stderr
    out = _json_line(r.stdout)
    assert out["devices"] == [0, 0, 0] and out["ranks"] == [0, 1, 2]
    assert len(set(out["rccl_host_ids"])) == 3 and None not in out["rccl_host_ids"]


def test_bench_under_torch_distributed_run():
    """The driver's multi-GPU command line: torch.distributed.run starts the ranks, bench.py finds
    WORLD_SIZE in its environment and does not start children of its own."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["tokens_agree"] and out["ranks"] == [0, 1]


def test_bench_parent_reports_a_failing_rank():
    """A rank that dies makes the launcher exit non-zero (here: no GPU in the container, the ranks
    refuse to run because the product path has no CPU fallback)."""
    import importlib
    sfl = importlib.import_module("esp32-fluid-simulation_amd")
    if sfl.device_count() > 0:
        pytest.skip("needs a box without GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64",
                        "--iters", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr


def test_self_launcher_stops_the_other_ranks_when_one_dies():
    """RCCL send / recv has no timeout: the launcher supervises its children, and a rank that dies takes the others
    (here: waiting in a barrier for ever) down with it -- non-zero exit within seconds, no JSON line."""
    import time
    env = dict(os.environ, SFL_BENCH_TEST_FAIL_RANK="1")
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run"],
                       capture_output=True, text=True, timeout=120, env=env)
    # (the survivors may notice the dead socket and fail on their own before the launcher stops them: either way
    # the launcher reports a failure, promptly)
    assert r.returncode != 0, (r.returncode, r.stderr[-2000:])
    assert time.monotonic() - t0 < 60
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "bench.py launcher: a rank exited with status" in r.stderr


def test_self_launcher_deadline():
    """--launch-timeout: ranks that never finish are stopped and the launcher reports 124."""
    env = dict(os.environ, SFL_BENCH_TEST_FAIL_RANK="none", SFL_BENCH_TEST_HANG="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--launch-timeout", "3"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])


# ---- rounds 5 / 6: the headline is the library's default schedule; fresh rank processes per attempt; in line as the fallback; the
# in-time schedule as a separately labelled experiment behind a successful headline
def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SFL_BENCH_WORKER")}
    env.update(extra)
    return env


def _bench(*argv, timeout=300, **env):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True,
                          timeout=timeout, env=_clean_env(**env))


def test_one_gpu_line_has_no_launcher_fields():
    r = _bench("--gpus", "1", "--dry-run", timeout=120)
    assert r.returncode == 0, r.stderr
    out = _json_line(r.stdout)
    assert "exchange_mode" not in out and "fallback_from" not in out and "in_time_experiment" not in out and out["mode"] == "library default"


def test_headline_is_the_library_default_and_the_experiment_rides_along():
    """the first multi-GPU line must show what a caller of sfl_poisson_solve gets.  Both keys in ONE line."""
    r = _bench("--gpus", "2", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["exchange_mode"] == "library default" and out["mode"] == "library default" and out["fallback_from"] == []
    assert out["in_time_experiment"] == {"dry_run": True, "mode": "in-time"}
    r = _bench("--gpus", "2", "--dry-run", "--no-experiment")
    assert r.returncode == 0 and "in_time_experiment" not in _json_line(r.stdout)


@pytest.mark.parametrize("broken", ["fail", "hang"])
def test_a_failing_experiment_costs_the_headline_nothing(broken):
    env = {"SFL_BENCH_TEST_FAIL_MODES": "in-time"} if broken == "fail" else {"SFL_BENCH_TEST_HANG_MODES": "in-time"}
    r = _bench("--gpus", "2", "--dry-run", "--launch-timeout", "4", **env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["exchange_mode"] == "library default" and out["fallback_from"] == []
    assert ("made to fail by the test" if broken == "fail" else "no result after 4 s") in out["in_time_experiment"]["failed"]


def test_self_launcher_falls_back_to_exchanges_in_line():
    """The library's own schedule fails on a rank: fresh ranks are started with every exchange in line, the line says which
    schedule produced it and why the first did not; the experiment still follows the headline it can be compared with."""
    r = _bench("--gpus", "3", "--dry-run", SFL_BENCH_TEST_FAIL_MODES="library default")
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["exchange_mode"] == "in-line" and out["mode"] == "in-line" and out["n_gpus"] == 3
    assert [f["mode"] for f in out["fallback_from"]] == ["library default"]
    assert "made to fail by the test" in out["fallback_from"][0]["why"]
    assert out["in_time_experiment"] == {"dry_run": True, "mode": "in-time"}


def test_self_launcher_falls_back_past_a_schedule_that_hangs():
    """... or never finishes: the per-attempt deadline stops its ranks, the next schedule gets fresh ones."""
    r = _bench("--gpus", "2", "--dry-run", "--launch-timeout", "4", "--no-experiment", SFL_BENCH_TEST_HANG_MODES="library default")
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["exchange_mode"] == "in-line"
    assert [f["mode"] for f in out["fallback_from"]] == ["library default"]
    assert "no result after 4 s" in out["fallback_from"][0]["why"] and out["fallback_from"][0]["status"] == 124


def test_self_launcher_reports_every_schedules_reason_when_all_fail():
    r = _bench("--gpus", "2", "--dry-run", SFL_BENCH_TEST_FAIL_MODES="library default,in-line")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    tail = [l for l in r.stderr.splitlines() if "every exchange schedule failed" in l]
    assert len(tail) == 1
    reasons = json.loads(tail[0].split("failed: ", 1)[1])
    assert [f["mode"] for f in reasons] == ["library default", "in-line"]     # (no experiment without a headline)
    assert all("made to fail by the test" in f["why"] for f in reasons)


def test_a_schedule_asked_for_is_the_only_one_tried():
    r = _bench("--gpus", "2", "--dry-run", "--arrival-by-event", SFL_BENCH_TEST_FAIL_MODES="by-event")
    assert r.returncode != 0 and "starting fresh ranks" not in r.stderr
    r = _bench("--gpus", "2", "--dry-run", "--no-overlap", SFL_BENCH_TEST_FAIL_MODES="library default,in-time")
    out = _json_line(r.stdout)
    assert r.returncode == 0 and out["exchange_mode"] == "in-line" and "in_time_experiment" not in out
    r = _bench("--gpus", "2", "--dry-run", "--arrival-in-time")
    assert r.returncode == 0 and _json_line(r.stdout)["exchange_mode"] == "in-time"


def test_a_variant_library_travels_to_the_rank_processes():
    """under tools/with_lib.py (SFL_WITH_LIB) the rank processes are started through with_lib.py as well."""
    sys.path.insert(0, ROOT)
    import bench
    old = os.environ.pop("SFL_WITH_LIB", None)
    try:
        assert bench.worker_argv(["--x"])[1].endswith("bench.py")
        os.environ["SFL_WITH_LIB"] = "/somewhere/libsfl_variant.so"
        argv = bench.worker_argv(["--x"])
        assert argv[1].endswith(os.path.join("tools", "with_lib.py")) and argv[2] == "/somewhere/libsfl_variant.so" and argv[3].endswith("bench.py")
    finally:
        os.environ.pop("SFL_WITH_LIB", None)
        if old is not None:
            os.environ["SFL_WITH_LIB"] = old


def _torchrun(port, *argv, **env):
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2",
                           "--dry-run"] + list(argv), capture_output=True, text=True, timeout=600, env=_clean_env(**env))


@pytest.mark.parametrize("broken", ["fail", "hang"])
def test_fallback_under_torch_distributed_run(broken):
    """The driver's command line: torch.distributed.run starts one bench.py per rank; each is its rank's supervisor, starts a
    fresh worker per attempt, and the supervisors agree on failing over to the next schedule together -- none of them exits
    non-zero in between (the elastic agent would end the run)."""
    env = {"SFL_BENCH_TEST_FAIL_MODES": "library default"} if broken == "fail" else {"SFL_BENCH_TEST_HANG_MODES": "library default"}
    r = _torchrun(29641 if broken == "fail" else 29643, "--launch-timeout", "5", **env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["exchange_mode"] == "in-line" and out["tokens_agree"]
    assert [f["mode"] for f in out["fallback_from"]] == ["library default"]
    assert ("made to fail by the test" if broken == "fail" else "no result after 5 s") in out["fallback_from"][0]["why"]
    assert out["in_time_experiment"] == {"dry_run": True, "mode": "in-time"}


def test_headline_and_experiment_under_torch_distributed_run():
    r = _torchrun(29647)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["exchange_mode"] == "library default" and out["fallback_from"] == []
    assert out["in_time_experiment"] == {"dry_run": True, "mode": "in-time"}
    r = _torchrun(29649, "--launch-timeout", "5", SFL_BENCH_TEST_HANG_MODES="in-time")
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["exchange_mode"] == "library default" and "no result after 5 s" in out["in_time_experiment"]["failed"]


def test_all_schedules_failing_under_torch_distributed_run():
    r = _torchrun(29645, SFL_BENCH_TEST_FAIL_MODES="library default,in-line")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "every exchange schedule failed" in r.stderr


def test_the_multi_gpu_suite_fits_the_drivers_limit():
    """on a node with 8 GPUs the one-rank-per-GPU cases un-skip on top of the one-GPU suite (~4 minutes) under
    the driver's 1200 s limit: at most 10 of them in the default selection, configuration 5 (a 43 s reference solve) once."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    m = importlib.import_module("test_multi_gpu")
    assert len(m.REAL_GPU_CASES) <= 10
    assert sum(1 for c in m.REAL_GPU_CASES if c[1] == 16384) == 1
    assert {(c[0], c[1], c[2]) for c in m.REAL_GPU_CASES} >= {(2, 8192, 80), (4, 8192, 80), (8, 8192, 80), (8, 16384, 200)}
    assert len(m.REAL_GPU_CASES_SLOW) >= 5
