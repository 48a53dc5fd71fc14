"""RCCL with more than one rank.  The ranks are fresh processes started by bench.py's own launcher; each solves its slab with
RCCL halo exchanges and compares the result of the timed solve bit for bit with the reference CPU loop run on the whole domain
(bench.py's `parity` block); the full sim step runs as well.
  * test_rccl_slab_solve_matches_reference: one rank per GPU -- needs as many GPUs as ranks (skipped on the 1-GPU test boxes);
  * test_rccl_ranks_as_processes_on_one_device: the same N processes and the same N-rank communicator on ONE GPU, over RCCL's
    socket transport (bench.py --share-device): runs wherever there is a GPU.
(The in-process virtual-rank tests of test_gpu_parity.py cover the slab program itself without RCCL.)"""
import importlib
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _needs_loopback():
    """the ranks of a shared device talk over RCCL's socket transport on `lo` (NCCL_SOCKET_IFNAME): no such interface, no test"""
    import socket
    if "lo" not in [name for _, name in socket.if_nameindex()]:
        pytest.skip("no loopback interface for RCCL's socket transport")


def _devices():
    return importlib.import_module("esp32-fluid-simulation_amd").device_count()


# (8, 8192, 80) is BASELINE config 4, (2 / 4, 8192, 80) its other scaling points, (8, 16384, 200) config 5 -- exactly as
# bench.py runs them; the smaller grids exercise classic (halo < 2 x fuse) and early exchanges, with and without overlap.
# mode: "" = the launcher's own chain of schedules (exchanges in time first; fresh ranks with early exchanges behind events, then
# in line, should an attempt fail: the line says which schedule produced it); the others force ONE schedule, no fallback:
# "in-time" (counted on the device), "by-event" (the library's own default on RCCL ranks), "in-line", and in-time with chained
# launches where the slabs are thin ("chain-auto") / wherever they can run ("chain")
MODES = {"": [], "in-time": ["--arrival-in-time"], "by-event": ["--arrival-by-event"], "in-line": ["--no-overlap"],
         "chain-auto": ["--arrival-in-time", "--chain", "-1"], "chain": ["--arrival-in-time", "--chain", "1"]}


def _run_bench(nranks, size, iters, halo, mode, steps, extra=(), env_extra=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--size", str(size),
           "--iters", str(iters), "--steps", str(steps), "--warmup", "1", "--sim-steps", "1", "--no-priming"] + MODES[mode] + list(extra)
    if halo:
        cmd += ["--sor-halo", str(halo)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SFL_BENCH_WORKER")}
    env.update(env_extra or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == nranks
    assert out["parity"]["bit_exact"] and out["parity"]["cells"] == size * size
    assert out["config"]["halo_exchanges_per_solve"] > 0
    assert out["sim_steps_per_sec"] is not None, out.get("sim_steps_note")
    # the sim step's fields: every rank's rows against the same steps on one whole-domain context
    assert out.get("sim_step_parity", {}).get("bit_exact") is True, out.get("sim_step_parity")
    # which schedule produced the line; a forced schedule is the one that ran
    assert out["exchange_mode"] in ("in-time", "by-event", "in-line")
    if mode in ("in-time", "by-event", "in-line"):
        assert out["exchange_mode"] == mode and out["fallback_from"] == []
    if mode == "":
        print("launcher chain:", out["exchange_mode"], out["fallback_from"])
    return out, r.stderr


@pytest.mark.parametrize("nranks,size,iters,halo,mode", [
    (2, 2048, 40, 0, ""), (2, 2048, 40, 0, "in-time"), (2, 2048, 40, 0, "by-event"), (2, 1024, 24, 16, "in-time"), (2, 1024, 24, 12, "in-line"),
    (4, 2048, 40, 0, ""), (8, 4096, 30, 0, "in-time"), (2, 8192, 80, 0, ""), (4, 8192, 80, 0, ""), (8, 8192, 80, 0, ""),
    (8, 8192, 80, 0, "in-time"), (8, 8192, 80, 0, "by-event"), (8, 8192, 80, 0, "in-line"), (8, 8192, 80, 160, "in-time"),
    (8, 8192, 80, 0, "chain-auto"), (2, 2048, 40, 0, "chain"), (4, 8192, 80, 0, "chain"), (8, 16384, 200, 0, ""),
    (8, 16384, 200, 160, "in-time")])
def test_rccl_slab_solve_matches_reference(nranks, size, iters, halo, mode):
    if _devices() < nranks:
        pytest.skip(f"needs {nranks} GPUs, {_devices()} visible")
    _run_bench(nranks, size, iters, halo, mode, steps=14)


# The SAME path -- N rank processes, a real N-rank RCCL communicator, matched ncclSend / ncclRecv between processes, the collective
# decisions (option check, measured exchange, halo tuner), every exchange schedule -- on a box with ONE GPU: bench.py --share-device 0
# puts every rank on device 0 and gives each its own NCCL_HOSTID, so RCCL takes them for ranks on different hosts ("Duplicate GPU
# detected" otherwise) and moves the halos over its socket transport on the loopback interface.  What this proves: the multi-process
# program is right (every cell of the solve against the reference CPU loop, every field of a sim step against a whole-domain
# context), with RCCL's own log as the witness of the communicator's size.  What it cannot show: xGMI, or any timing.
# (8, 8192, 80) is BASELINE configuration 4 exactly as the driver's 8-GPU run would execute it, but for the wire (17 s; its other
# schedules, a 160-row halo and configuration 5 were run by hand: profiles/r05_rccl_ranks_on_one_device.txt).
@pytest.mark.parametrize("nranks,size,iters,halo,mode", [
    (2, 2048, 40, 0, ""), (2, 2048, 40, 0, "in-time"), (2, 2048, 40, 0, "by-event"), (2, 1024, 24, 12, "in-line"), (2, 1024, 24, 16, "in-time"),
    (2, 2048, 40, 0, "chain"), (4, 2048, 40, 0, ""), (8, 4096, 30, 0, "by-event"), (8, 8192, 80, 0, "")])
def test_rccl_ranks_as_processes_on_one_device(nranks, size, iters, halo, mode):
    if _devices() < 1:
        pytest.skip("needs a GPU")
    _needs_loopback()
    extra = ["--share-device", "0"]
    if "--arrival-in-time" in MODES[mode]:
        extra += ["--halo-timeout-ms", "15000"]   # (a lost message is to end as an error, not as a kernel that spins for minutes)
    out, log = _run_bench(nranks, size, iters, halo, mode, steps=6, extra=extra,
                          env_extra={"NCCL_DEBUG": "INFO", "NCCL_DEBUG_SUBSYS": "INIT,NET"})
    assert out["config"]["physical_gpus"] == 1 and out["config"]["ranks_share_device"] == 0
    assert "ON DEVICE 0" in out["config"]["parallelism"]
    # RCCL's own word for it: a communicator of `nranks` ranks, reached over the socket transport
    assert f"nranks {nranks}" in log or f"nRanks {nranks:02d}" in log, log[-3000:]
    assert "NET/Socket" in log, log[-3000:]


def test_a_real_time_out_makes_the_launcher_fall_back_with_fresh_rccl_ranks():
    """End to end, nothing simulated: four rank processes on one device, exchanges in time with a limit (1 ms) no message between
    time-sliced processes can meet -- a wait inside a solve gives up on a real RCCL rank, that rank fails loudly, the launcher stops
    the others and starts FRESH ranks behind events, whose result is the reference's."""
    if _devices() < 1:
        pytest.skip("needs a GPU")
    _needs_loopback()
    out, log = _run_bench(4, 2048, 40, 0, "", steps=6, extra=["--share-device", "0", "--halo-timeout-ms", "1"])
    assert out["exchange_mode"] == "by-event", (out["exchange_mode"], out["fallback_from"])
    assert [f["mode"] for f in out["fallback_from"]] == ["in-time"]
    assert "a wait inside a solve lasted longer than" in out["fallback_from"][0]["why"]
