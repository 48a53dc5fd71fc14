"""CPU suite: guards of the build and of the source layout.

* The diagnostic switches of the fused SOR kernel (SFL_PROBE_*: ablations and timing mocks that compute WRONG results) are legal
  only in the trace harness tools/sor_clock_probe.hip; a product build that carries one -- `make EXTRA_FLAGS=-DSFL_PROBE_NO_LOAD=1`
  -- must stop in the compiler.
* No translation unit or header under csrc/ grows past 900 lines again (round 4's sfl_api.cpp had reached 2710)."""
import glob
import os
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "esp32-fluid-simulation_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("flag", ["-DSFL_PROBE_NO_LOAD=1", "-DSFL_PROBE_SHIFT=1", "-DSFL_PROBE_COOP=3", "-DSFL_PROBE_NO_STORE=1", "-DSEAM_MOCK_NO_P=1"])
def test_a_product_build_refuses_the_diagnostic_switches(flag):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    source = "advect_tiled.hip" if "SEAM" in flag else "sor_fused.hip"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fsyntax-only", "-DSFL_NS_GROUP=2",
                        "-DSFL_DX_PART=0", "-DSFL_FOLD_PART=0", flag, os.path.join(CSRC, source)], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "diagnostic builds only" in r.stderr or "only with SFL_SOR_TRACE" in r.stderr, r.stderr[-1500:]


def test_no_source_file_outgrows_900_lines():
    long = {os.path.basename(f): sum(1 for _ in open(f)) for f in glob.glob(os.path.join(CSRC, "*")) if os.path.isfile(f)}
    assert long and max(long.values()) <= 900, {k: v for k, v in long.items() if v > 900}
