"""CPU suite: the C-ABI library loads without a GPU, exports every symbol include/sfl.h
declares, and its GPU-free entry points (version, slab arithmetic, plans, error reporting)
behave.  No compute call is made here."""
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sfl.h")).read()
    return sorted(set(re.findall(r"SFL_API\s+[\w\s\*]+?\b(sfl_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(sfl):
    lib = sfl.capi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/sfl.h but not exported"
    assert sorted(sfl.capi.SIGNATURES) == declared, "ctypes table out of sync with include/sfl.h"


def test_option_numbers_of_the_binding_match_the_header(sfl):
    """Every SFL_OPT_* of include/sfl.h has the same number in the ctypes binding and vice versa -- round 6 retired four
    options (8, 13, 15, 16) and added one: nobody may go on using a retired number under an old name."""
    text = open(os.path.join(ROOT, "include", "sfl.h")).read()
    header = {name: int(num) for name, num in re.findall(r"#define SFL_(OPT_\w+)\s+(\d+)", text)}
    binding = {k: v for k, v in vars(sfl.capi).items() if k.startswith("OPT_") and isinstance(v, int)}
    assert header == binding
    assert len(set(header.values())) == len(header) and not {8, 13, 15, 16} & set(header.values())
    assert header["OPT_EXCHANGE_SCHEDULE"] == 19 and header["OPT_SOR_FOLD"] == 22


def test_version_and_error_reporting_without_gpu(sfl):
    lib = sfl.capi.lib()
    assert lib.sfl_abi_version() == 1
    if sfl.device_count() == 0:
        with pytest.raises(sfl.SflError) as e:
            sfl.Solver(16, 16)
        assert e.value.code == sfl.capi.ERR_HIP  # fails loudly: no CPU fallback


def test_slab_rows_partition():
    import importlib
    sfl = importlib.import_module("esp32-fluid-simulation_amd")
    for dim_y, n in [(8192, 8), (81, 2), (81, 3), (100, 7), (16384, 8)]:
        cuts = [sfl.slab_rows(dim_y, n, r) for r in range(n)]
        assert cuts[0][0] == 0 and cuts[-1][1] == dim_y
        assert all(cuts[r][1] == cuts[r + 1][0] for r in range(n - 1))
        assert max(e - b for b, e in cuts) - min(e - b for b, e in cuts) <= 1
    assert sfl.slab_rows(8192, 8, 3) == (3072, 4096)


def test_pass_plan(sfl):
    assert sfl.sor_pass_plan(80, 8) == [8] * 20
    assert sfl.sor_pass_plan(10, 8) == [8, 8, 4]
    assert sfl.sor_pass_plan(1, 16) == [2]
    assert sfl.sor_pass_plan(0, 8) == []
    with pytest.raises(sfl.SflError):
        sfl.sor_pass_plan(4, 3)


def test_poisson_program_shape(sfl):
    E, S, Z = sfl.capi.STEP_EXCHANGE, sfl.capi.STEP_SOR, sfl.capi.STEP_ZERO
    # single rank: no exchanges at all
    prog = sfl.plan_poisson(8192, 1, 0, 80, 8, 2)
    assert [s.kind for s in prog] == [S] * 20 and prog[0].from_zero == 1 and prog[1].from_zero == 0
    # 8 ranks, fused: one rhs exchange + one p exchange before every launch but the first
    progs = [sfl.plan_poisson(8192, 8, r, 80, 8, 2) for r in range(8)]
    kinds = [s.kind for s in progs[0]]
    assert kinds == [E, S] + [E, S] * 19
    assert progs[0][0].field == sfl.capi.FIELD_DIVERGENCE and progs[0][0].rows == 7
    assert all(s.rows == 8 and s.field == sfl.capi.FIELD_PRESSURE for s in progs[0][2::2])
    for r in range(8):  # same skeleton on every rank, own rows as output
        assert [s.kind for s in progs[r]] == kinds
        assert all((s.g_begin, s.g_end) == sfl.slab_rows(8192, 8, r) for s in progs[r] if s.kind == S)
    # classic supersteps (halo < 2 * fuse): halo 12 at fuse 8 -> one launch per superstep, rhs 7 rows
    prog = sfl.plan_poisson(8192, 8, 3, 80, 8, 2, 12)
    assert [s.kind for s in prog] == [E, S] + [E, S] * 19 and all(s.g_begin == 0 for s in prog if s.kind == E)
    # EARLY exchanges (halo >= 2 * fuse): halo 32 at fuse 8.  The first superstep holds 4 launches (32 passes);
    # the exchange for the next one travels before its LAST launch, while the ghost rows are still valid 8 deep:
    # rows at depth [8, 32) only, and that launch's output extends 24 rows into the ghost rows (its passes are
    # repeated on what was received).  Afterwards 24 rows are left: supersteps of 3 launches.
    prog = sfl.plan_poisson(8192, 8, 3, 80, 8, 2, 32)
    assert [s.kind for s in prog] == [E] + [S] * 3 + ([E] + [S] * 3) * 5 + [E, S, S]
    assert prog[0].field == sfl.capi.FIELD_DIVERGENCE and prog[0].rows == 31 and prog[0].g_begin == 0
    assert all(s.rows == 24 and s.g_begin == 8 and s.field == sfl.capi.FIELD_PRESSURE for s in prog[1:] if s.kind == E)
    g0, g1 = sfl.slab_rows(8192, 8, 3)
    assert [(s.g_begin, s.g_end) for s in prog[1:4]] == [(g0 - 24, g1 + 24), (g0 - 16, g1 + 16), (g0 - 8, g1 + 8)]
    assert [(s.g_begin, s.g_end) for s in prog[5:8]] == [(g0 - 24, g1 + 24), (g0 - 16, g1 + 16), (g0 - 8, g1 + 8)]
    assert (prog[-1].g_begin, prog[-1].g_end) == (g0, g1)
    assert [s.from_zero for s in prog if s.kind == S] == [1] + [0] * 19
    # BASELINE config 4 as the library runs it (fuse 10, halo 64): 16 launches, the rhs + two early exchanges
    prog = sfl.plan_poisson(8192, 8, 3, 80, 10, 2, 64)
    assert [(s.rows, s.g_begin) for s in prog if s.kind == E] == [(59, 0), (54, 10), (54, 10)]
    assert [s.kind for s in prog].index(E, 1) == 6 and sum(s.kind == S for s in prog) == 16
    # the domain's bottom / top slabs never reach outside the domain
    lo, hi = sfl.plan_poisson(8192, 8, 0, 80, 8, 2, 32), sfl.plan_poisson(8192, 8, 7, 80, 8, 2, 32)
    assert min(s.g_begin for s in lo if s.kind == S) == 0
    assert max(s.g_end for s in hi if s.kind == S) == 8192
    # a short last superstep exchanges only what it needs: 10 iters at fuse 8 = launches 8, 8, 4
    prog = sfl.plan_poisson(400, 2, 0, 10, 8, 2, 12)   # classic: launches 8 | 8 + 4
    assert [(s.kind, s.rows or s.nsweeps) for s in prog] == [(E, 11), (S, 8), (E, 12), (S, 8), (S, 4)]
    prog = sfl.plan_poisson(400, 2, 0, 10, 8, 2, 16)   # early: the 4-pass launch needs 4 ghost rows after launch 2
    assert [(s.kind, s.rows or s.nsweeps, s.g_begin if s.kind == E else s.g_end - 200) for s in prog] == [
        (E, 15, 0), (S, 8, 8), (E, 8, 8), (S, 8, 4), (S, 4, 0)]
    # baseline: zero fill, then exchange before every colour pass but the first
    prog = sfl.plan_poisson(100, 2, 1, 2, 8, 1)
    assert [s.kind for s in prog] == [Z, S, E, S, E, S, E, S]
    assert [s.first_colour for s in prog if s.kind == S] == [0, 1, 0, 1]


def test_context_size_limits_are_rejected_before_any_gpu_is_touched(sfl):
    """The kernels address a context's local arrays with 32-bit byte offsets (8-byte velocity
    elements): a context may hold at most 2^28 cells, a domain 2^30; both limits come back as
    SFL_ERR_INVALID -- on a box without a GPU too, i.e. before the device is even looked at."""
    import ctypes as C
    lib, cap = sfl.capi.lib(), sfl.capi
    h = C.c_void_p()
    for dim_x, dim_y, rank, nranks, code in [
            (32768, 16384, 0, 1, cap.ERR_INVALID),     # 2^29 cells in one context
            (16384, 16385, 0, 1, cap.ERR_INVALID),     # one row too many
            (65536, 32768, 0, 8, cap.ERR_INVALID),     # domain beyond 2^30 cells
            (16384, 32768, 0, 2, cap.ERR_INVALID),     # slab of 2^28 cells + ghost rows
            (1, 8, 0, 1, cap.ERR_INVALID), (8, 1, 0, 1, cap.ERR_INVALID)]:
        rc = lib.sfl_create_slab(C.byref(h), 0, dim_x, dim_y, rank, nranks)
        assert rc == code, (dim_x, dim_y, rank, nranks, rc, lib.sfl_last_error())
        assert not h.value
    # within the limits the only possible failure is the missing device
    rc = lib.sfl_create_slab(C.byref(h), 0, 16384, 32768, 1, 4)
    assert rc in (cap.OK, cap.ERR_HIP)
    if rc == cap.OK:
        lib.sfl_destroy(h)
