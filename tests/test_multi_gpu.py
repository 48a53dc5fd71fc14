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
# bench.py runs them; the smaller grids exercise classic (halo < 2 x fuse) and early exchanges, with and without overlap
# overlap: 1 = exchanges on the second stream, the halo's arrival counted on the device (the default), 2 = the same with a
# cross-stream event in front of the launch that needs it (SFL_OPT_SOR_ARRIVAL = 0), 0 = in line; 3 = as 1 with chained launches where the slabs
# are thin (--chain -1), 4 = as 1 with chained launches wherever they can run (--chain 1)
@pytest.mark.parametrize("nranks,size,iters,halo,overlap", [
    (2, 2048, 40, 0, 1), (2, 2048, 40, 0, 2), (2, 1024, 24, 16, 1), (2, 1024, 24, 12, 0), (4, 2048, 40, 0, 1), (8, 4096, 30, 0, 1),
    (2, 8192, 80, 0, 1), (4, 8192, 80, 0, 1), (8, 8192, 80, 0, 1), (8, 8192, 80, 0, 2), (8, 8192, 80, 0, 0), (8, 8192, 80, 0, 3),
    (2, 2048, 40, 0, 4), (4, 8192, 80, 0, 4), (8, 16384, 200, 0, 1)])
def test_rccl_slab_solve_matches_reference(nranks, size, iters, halo, overlap):
    if _devices() < nranks:
        pytest.skip(f"needs {nranks} GPUs, {_devices()} visible")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--size", str(size),
           "--iters", str(iters), "--steps", "2", "--warmup", "1", "--sim-steps", "1", "--no-priming"]
    if halo:
        cmd += ["--sor-halo", str(halo)]
    if not overlap:
        cmd += ["--no-overlap"]
    if overlap == 2:
        cmd += ["--arrival-by-event"]
    if overlap in (3, 4):
        cmd += ["--chain", "-1" if overlap == 3 else "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == nranks
    assert out["parity"]["bit_exact"] and out["parity"]["cells"] == size * size
    assert out["config"]["halo_exchanges_per_solve"] > 0
    assert out["sim_steps_per_sec"] is not None, out.get("sim_steps_note")
    # the sim step's fields: every rank's rows against the same steps on one whole-domain context
    assert out.get("sim_step_parity", {}).get("bit_exact") is True, out.get("sim_step_parity")
