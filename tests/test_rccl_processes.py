"""GPU suite: the product's C ABI driven by N RANK PROCESSES that form ONE real N-rank RCCL communicator -- on a box with one GPU.
Every rank sits on device 0 and carries its own NCCL_HOSTID, so RCCL takes them for ranks on different hosts (it refuses duplicate
devices of one host) and moves every halo between the processes over its socket transport on the loopback interface: matched
ncclSend / ncclRecv pairs, the option check's all-gather, the measured exchange's and the step report's all-reduce, the all-pairs
gather, the in-time protocol's device-side counts -- all between real processes.  tests/rccl_rank_worker.py is the
rank; each checks its own rows against the oracle bit for bit.  (tests/test_multi_gpu.py runs bench.py the same way.)"""
import importlib
import json
import os
import subprocess
import sys
import tempfile

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_ranks(world, scenario, seed, seconds=None, timeout=600):
    if importlib.import_module("esp32-fluid-simulation_amd").device_count() < 1:
        pytest.skip("needs a GPU")
    import socket
    if "lo" not in [name for _, name in socket.if_nameindex()]:   # (NCCL_SOCKET_IFNAME=lo: RCCL's socket transport between the ranks)
        pytest.skip("no loopback interface for RCCL's socket transport")
    private = tempfile.mkdtemp(prefix="sfl_rccl_test_")
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NCCL_HOSTID")}
    base.update({"WORLD_SIZE": str(world), "SFL_RDZV_KEY": f"pytest_{os.getpid()}_{scenario}_{world}", "SFL_RDZV_DIR": private})
    cmd = [sys.executable, os.path.join(ROOT, "tests", "rccl_rank_worker.py"), scenario, str(seed)] + ([str(seconds)] if seconds else [])
    kids = [subprocess.Popen(cmd, env=dict(base, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            for r in range(world)]
    results = []
    try:
        for k in kids:
            out, err = k.communicate(timeout=timeout)
            lines = [l for l in out.splitlines() if l.startswith("{")]
            assert lines, f"a rank printed no result (status {k.returncode}): {err[-3000:]}"
            results.append(json.loads(lines[-1]))
            assert k.returncode == 0, (results[-1], err[-2000:])
    finally:
        for k in kids:          # (our own children, by PID)
            if k.poll() is None:
                k.kill()
    assert sorted(r["rank"] for r in results) == list(range(world))
    assert all(r["ok"] for r in results), results
    return results


@pytest.mark.parametrize("world", [2, 3])
def test_random_slab_groups_over_real_rccl_ranks(world):
    """Seeded random shapes / iterations / fuse and halo depths / schedules (behind events, in time,
    in line) / dx / omega / dt / velocity scales: a solve and 2-3 sim steps per configuration, all four fields of every rank."""
    results = run_ranks(world, "soak", 100 + world, seconds=10)   # (minutes of it: tools/recipes/soak.sh, leg "ranks")
    assert results[0]["cases"] >= 3, results
    print(f"{world} RCCL rank processes on one device: {results[0]['cases']} random configurations, 0 mismatches")


@pytest.mark.parametrize("world", [2, 4])
def test_rank_created_with_other_options_is_refused_by_every_rank(world):
    results = run_ranks(world, "mismatch", 1)
    assert all(r["refused"] and "disagree" in r["why"] for r in results)


def test_gather_fallback_and_late_checked_dye_halo_between_processes():
    run_ranks(3, "gather", 7)


@pytest.mark.parametrize("world", [2, 4])
def test_forces_on_both_sides_of_a_cut_between_processes(world):
    run_ranks(world, "forces", 11)


def test_in_time_exchanges_with_a_peer_that_is_late_to_every_solve():
    results = run_ranks(3, "late_peer", 5)
    assert all(r["schedule"] == 3 and r["exchanges"] > 1 for r in results), results


def test_quiescent_field_with_drags_between_processes():
    """The input class of round 6 (a quiescent field, sparse forcing, 80 iterations, two whole sim steps) on three real RCCL rank
    processes, every exchange schedule: all four fields of every rank bit for bit; the scenario is checked to reach the denormals."""
    results = run_ranks(3, "quiescent", 21)
    assert all(r["denormal_front_cells"] > 50 for r in results), results
