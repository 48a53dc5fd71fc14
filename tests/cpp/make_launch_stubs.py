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
ok = [out[0].replace("launch_stubs.cpp", "launch_stubs_ok.cpp").replace('as a stub that reports\n// "no device"', 'as a stub that does NOTHING\n// and reports success')
      .replace("can be run under AddressSanitizer + UBSan on the CPU", "can be RUN THROUGH on the CPU over the fake runtime of fake_hip.cpp (ThreadSanitizer: `make tsan_host`)")]
for ret, name, args in decls:
    a = re.sub(r"\s*=\s*[^,()]+(\([^)]*\))?", "", args)   # drop default arguments
    a = re.sub(r"\s+", " ", a).strip()
    out.append(f"{ret} {name}({a}) {{ {'return false;' if ret == 'bool' else 'return hipErrorNoDevice;'} }}\n")
    body = "if (senders) *senders = 0; " if re.search(r"int \*senders", a) else ""
    ok.append(f"{ret} {name}({a}) {{ {body}{'return false;' if ret == 'bool' else 'return hipSuccess;'} }}\n")
out.append("\n}  // namespace sfl\n")
ok.append("\n}  // namespace sfl\n")
open(os.path.join(HERE, "launch_stubs.cpp"), "w").write("".join(out))
open(os.path.join(HERE, "launch_stubs_ok.cpp"), "w").write("".join(ok))
print(f"{len(decls)} stubs")
