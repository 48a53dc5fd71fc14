#!/usr/bin/env python3
"""Regenerates tests/cpp/launch_stubs.cpp from csrc/kernels.h (run after the launch interface changes)."""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, "..", "..", "esp32-fluid-simulation_amd", "csrc", "kernels.h")).read()
code = re.sub(r"//[^\n]*", "", src)
decls = re.findall(r"\b(hipError_t|bool)\s+(\w+)\s*\(([^;{]*?)\)\s*;", code, re.S)
out = ['''// launch_stubs.cpp -- TEST HARNESS ONLY (tests/cpp, `make san_host`): every launcher of csrc/kernels.h as a stub that reports
// "no device", so that the HOST side of the library (contexts, options, plans, transports, executors) links without the
// gfx950 kernels and can be run under AddressSanitizer + UBSan on the CPU.  Generated from kernels.h by
// tests/cpp/make_launch_stubs.py; never part of the product.
#include "kernels.h"

namespace sfl {
''']
for ret, name, args in decls:
    a = re.sub(r"\s*=\s*[^,()]+(\([^)]*\))?", "", args)   # drop default arguments
    a = re.sub(r"\s+", " ", a).strip()
    out.append(f"{ret} {name}({a}) {{ {'return false;' if ret == 'bool' else 'return hipErrorNoDevice;'} }}\n")
out.append("\n}  // namespace sfl\n")
open(os.path.join(HERE, "launch_stubs.cpp"), "w").write("".join(out))
print(f"{len(decls)} stubs")
