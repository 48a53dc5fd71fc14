#!/usr/bin/env python3
"""Regenerates tests/golden/domain_iter_reference.txt, sample_reference.txt and advect_generic_reference.txt: the output of
tests/cpp/domain_iter_driver.cpp / sample_driver.cpp / advect_generic_driver.cpp compiled against the REFERENCE's operations.h / advect.h (/root/reference, build container only).  The
fixture is data: the bits `domain_iter` (operations.h:11-38) leaves behind for the driver's
order-sensitive expressions.  Nothing of the reference's source is stored."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/ESP32-fluid-simulation"


def run_driver(include_dir: str, driver: str = "domain_iter_driver.cpp", defines=()) -> str:
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "drv")
        subprocess.run(["g++", "-std=gnu++17", "-O1", "-ffp-contract=off", "-Wno-unused-parameter", "-I",
                        include_dir, os.path.join(ROOT, "tests", "cpp", driver), "-o", exe] +
                       ["-D" + d for d in defines], check=True)
        return subprocess.run([exe], capture_output=True, text=True, check=True).stdout


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference")
    # advect_generic_driver.cpp: the reference's advect<T, U> template as a host loop, for element / velocity types
    # other than the sketch's two (all of the driver's cases: DRIVER_ANY_TYPE)
    for driver, fixture, defines in (("domain_iter_driver.cpp", "domain_iter_reference.txt", ()),
                                     ("sample_driver.cpp", "sample_reference.txt", ()),
                                     ("advect_generic_driver.cpp", "advect_generic_reference.txt", ("DRIVER_ANY_TYPE",))):
        out = run_driver(REF, driver, defines)
        with open(os.path.join(HERE, fixture), "w") as f:
            f.write(out)
        print(f"{fixture}: {len(out.splitlines())} lines written")
