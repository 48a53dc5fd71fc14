"""RCCL with more than one rank: needs >= 2 GPUs in the box (skipped on the 1-GPU test boxes; the
in-process virtual-rank tests of test_gpu_parity.py cover the same slab program on one GPU).  The
ranks are fresh processes started by bench.py's own launcher; each solves its slab with RCCL halo
exchanges and compares the result of the timed solve bit for bit with the reference CPU loop run
on the whole domain (bench.py's `parity` block); the full sim step runs as well."""
import importlib
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("nranks,size,iters,halo,mode", [
    (2, 2048, 40, 0, ""), (2, 2048, 40, 0, "in-time"), (2, 2048, 40, 0, "by-event"), (2, 1024, 24, 16, "in-time"), (2, 1024, 24, 12, "in-line"),
    (4, 2048, 40, 0, ""), (8, 4096, 30, 0, "in-time"), (2, 8192, 80, 0, ""), (4, 8192, 80, 0, ""), (8, 8192, 80, 0, ""),
    (8, 8192, 80, 0, "in-time"), (8, 8192, 80, 0, "by-event"), (8, 8192, 80, 0, "in-line"), (8, 8192, 80, 160, "in-time"),
    (8, 8192, 80, 0, "chain-auto"), (2, 2048, 40, 0, "chain"), (4, 8192, 80, 0, "chain"), (8, 16384, 200, 0, ""),
    (8, 16384, 200, 160, "in-time")])
def test_rccl_slab_solve_matches_reference(nranks, size, iters, halo, mode):
    if _devices() < nranks:
        pytest.skip(f"needs {nranks} GPUs, {_devices()} visible")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--size", str(size),
           "--iters", str(iters), "--steps", "14", "--warmup", "1", "--sim-steps", "1", "--no-priming"] + MODES[mode]
    if halo:
        cmd += ["--sor-halo", str(halo)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SFL_BENCH_WORKER")})
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
