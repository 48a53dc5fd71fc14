"""CPU suite, multi-process: the PRODUCT's slab program (sfl_plan_poisson through the C ABI: which
halo to exchange when, how many rows, which launches) is executed by `world_size` gloo ranks --
gloo send/recv standing in for RCCL, the oracle's row-restricted colour pass standing in for the
kernels -- and must reproduce the whole-domain reference solve bit for bit.  Ghost rows start as
NaN, so an exchange that is missing, too shallow or too late poisons the result."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_bit_equal

OMEGA = np.float32(1.96)
GHOST = 160   # ghost rows per side of a slab (csrc/context.h kGhostRows)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _exchange(dist, torch, rank, world, arr, grow0, g0, g1, rows, skip=0):
    """Matched neighbour exchange of `rows` owned rows per side, those at depth [skip, skip + rows) from the
    cuts, into the ghost rows at the same depth (skip > 0: the early exchanges of slab_plan.cpp -- the ghost
    rows nearer the cut must still be valid, or the NaNs they start with reach the result)."""
    ops, bufs = [], []
    lo, hi = g0 - grow0, g1 - grow0
    def send(block, peer):
        t = torch.from_numpy(np.ascontiguousarray(block))
        bufs.append(t)
        ops.append(dist.P2POp(dist.isend, t, peer))
    def recv(slc, peer):
        t = torch.empty((rows, arr.shape[1]), dtype=torch.float32)
        bufs.append((t, slc))
        ops.append(dist.P2POp(dist.irecv, t, peer))
    if rank > 0:
        send(arr[lo + skip:lo + skip + rows], rank - 1)
        recv(slice(lo - skip - rows, lo - skip), rank - 1)
    if rank < world - 1:
        send(arr[hi - skip - rows:hi - skip], rank + 1)
        recv(slice(hi + skip, hi + skip + rows), rank + 1)
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    for b in bufs:
        if isinstance(b, tuple):
            arr[b[1]] = b[0].numpy()


def _worker(rank, world, port, dim_x, dim_y, iters, fuse, kernel, halo, outdir, tail=0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import loader
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        sfl = importlib.import_module("esp32-fluid-simulation_amd")
        orc = loader.port()
        cap = sfl.capi
        g0, g1 = sfl.slab_rows(dim_y, world, rank)
        grow0 = g0 - GHOST
        d_full = np.random.default_rng(99).standard_normal((dim_y, dim_x)).astype(np.float32)
        lrows = g1 - g0 + 2 * GHOST
        d = np.full((lrows, dim_x), np.nan, np.float32)
        d[GHOST:GHOST + g1 - g0] = d_full[g0:g1]
        p = np.full((lrows, dim_x), np.nan, np.float32)
        fields = {cap.FIELD_PRESSURE: p, cap.FIELD_DIVERGENCE: d}
        dom_lo, dom_hi = max(grow0, 0) - grow0, min(grow0 + lrows, dim_y) - grow0
        n_exchanges = 0
        for st in sfl.plan_poisson(dim_y, world, rank, iters, fuse, kernel, halo, tail):
            if st.kind == cap.STEP_EXCHANGE:
                _exchange(dist, torch, rank, world, fields[st.field], grow0, g0, g1, st.rows, st.g_begin)
                n_exchanges += 1
            elif st.kind == cap.STEP_ZERO:
                p[dom_lo:dom_hi] = 0.0
            elif st.kind == cap.STEP_SOR:
                if st.from_zero:
                    p[dom_lo:dom_hi] = 0.0
                n = st.nsweeps
                for j in range(1, n + 1):
                    a = max(st.g_begin - (n - j), 0)
                    b = min(st.g_end + (n - j), dim_y)
                    orc.sor_half_sweep_rows(p, d, dim_y, (st.first_colour + j - 1) & 1, a - grow0,
                                            b - grow0, grow0, 1.0, OMEGA)
        np.save(os.path.join(outdir, f"p_{rank}.npy"), p[GHOST:GHOST + g1 - g0])
        # owned rows with `tail` ghost rows on each side that has a neighbour (NaN elsewhere: never compared)
        lo_t = tail if rank > 0 else 0
        hi_t = tail if rank < world - 1 else 0
        np.save(os.path.join(outdir, f"pt_{rank}.npy"), p[GHOST - lo_t:GHOST + g1 - g0 + hi_t])
        np.save(os.path.join(outdir, f"n_{rank}.npy"), np.array([n_exchanges]))
    finally:
        dist.destroy_process_group()


def _expected_exchanges(iters, fuse, halo):
    """Exchanges of a fused-kernel solve: the rhs once, then p -- classic (halo < 2 fuse): before every superstep
    but the first; early (halo >= 2 fuse): before the last launch of every superstep but the last."""
    passes = [min(fuse, 2 * iters - k) for k in range(0, 2 * iters, fuse)]
    halo = max(halo, fuse)
    if halo < 2 * fuse:
        per_group = halo // fuse
        return (-(-len(passes) // per_group) - 1) + 1
    n, budget = 1, halo
    for j, p in enumerate(passes):
        if p > budget:
            n += 1
            budget = halo - passes[j - 1]
        budget -= p
    return n


@pytest.mark.parametrize("world,kernel,fuse,iters,halo", [(2, 2, 8, 10, 0), (2, 2, 16, 9, 32), (2, 1, 2, 3, 0),
                                                         (3, 2, 4, 7, 16), (2, 2, 8, 13, 32), (3, 2, 6, 20, 30),
                                                         (2, 2, 10, 40, 64), (2, 2, 8, 12, 12)])
def test_slab_program_over_gloo(tmp_path, oracle, world, kernel, fuse, iters, halo):
    import torch.multiprocessing as mp
    dim_x, dim_y = 37, 140
    port = _free_port()
    mp.spawn(_worker, args=(world, port, dim_x, dim_y, iters, fuse, kernel, halo, str(tmp_path)),
             nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"p_{r}.npy") for r in range(world)], axis=0)
    d_full = np.random.default_rng(99).standard_normal((dim_y, dim_x)).astype(np.float32)
    assert_bit_equal(got, oracle.poisson_solve(d_full, 1.0, iters, OMEGA), f"{world} gloo ranks")
    n = int(np.load(tmp_path / "n_0.npy")[0])
    if kernel == 2:   # one rhs exchange + one of p per superstep after the first
        assert n == _expected_exchanges(iters, fuse, halo)
    else:
        assert n == 2 * iters - 1


@pytest.mark.parametrize("world,fuse,iters,halo,tail", [(2, 8, 13, 32, 1), (3, 4, 7, 16, 1), (2, 10, 40, 64, 1), (3, 6, 20, 30, 3),
                                                       (2, 8, 12, 16, 1)])
def test_slab_program_with_a_tail_over_gloo(tmp_path, oracle, world, fuse, iters, halo, tail):
    """Early-exchange plans with a TAIL (sfl_plan_poisson_tail: what sfl_step on slabs runs, tail 1): besides the
    owned rows, the `tail` ghost rows next to every cut must hold the reference's values when the solve ends --
    the row subtract_gradient reads without an exchange.  halo 16 at fuse 8 has no room for a tail: the plan
    then carries none (classic result, ghost rows not claimed)."""
    import torch.multiprocessing as mp
    dim_x, dim_y = 37, 140
    port = _free_port()
    mp.spawn(_worker, args=(world, port, dim_x, dim_y, iters, fuse, 2, halo, str(tmp_path), tail),
             nprocs=world, join=True)
    d_full = np.random.default_rng(99).standard_normal((dim_y, dim_x)).astype(np.float32)
    want = oracle.poisson_solve(d_full, 1.0, iters, OMEGA)
    sfl = importlib.import_module("esp32-fluid-simulation_amd")
    has_tail = halo >= 2 * fuse + tail
    for r in range(world):
        g0, g1 = sfl.slab_rows(dim_y, world, r)
        assert_bit_equal(np.load(tmp_path / f"p_{r}.npy"), want[g0:g1], f"rank {r}: owned rows")
        if has_tail:
            lo_t, hi_t = (tail if r > 0 else 0), (tail if r < world - 1 else 0)
            assert_bit_equal(np.load(tmp_path / f"pt_{r}.npy"), want[g0 - lo_t:g1 + hi_t], f"rank {r}: owned rows + tail")
    prog = sfl.plan_poisson(dim_y, world, 0, iters, fuse, 2, halo, tail)
    last = [s for s in prog if s.kind == sfl.capi.STEP_SOR][-1]
    assert last.g_end - sfl.slab_rows(dim_y, world, 0)[1] == (tail if has_tail else 0)


@pytest.mark.parametrize("world,fuse,iters,halo,tail", [(2, 10, 40, 64, 0), (2, 10, 40, 64, 1), (3, 6, 20, 30, 3), (2, 8, 12, 16, 1),
                                                       (3, 4, 7, 4, 0), (2, 16, 24, 64, 1)])
def test_in_time_slab_program_over_gloo(tmp_path, oracle, world, fuse, iters, halo, tail):
    """sfl_plan_poisson kernel 3 (what SFL_OPT_EXCHANGE_SCHEDULE = 3 runs): the fused launches with in-time exchanges at EVERY halo depth --
    the exchange of a superstep follows the launch that produces its rows, never precedes it -- with and without a tail
    (every superstep then holds halo - tail passes, the exchange skips the tail rows that are still exact).  Executed by
    gloo ranks with NaN ghosts: owned rows and tail rows must hold the reference's values."""
    import torch.multiprocessing as mp
    dim_x, dim_y = 37, 140
    port = _free_port()
    mp.spawn(_worker, args=(world, port, dim_x, dim_y, iters, fuse, 3, halo, str(tmp_path), tail),
             nprocs=world, join=True)
    d_full = np.random.default_rng(99).standard_normal((dim_y, dim_x)).astype(np.float32)
    want = oracle.poisson_solve(d_full, 1.0, iters, OMEGA)
    sfl = importlib.import_module("esp32-fluid-simulation_amd")
    has_tail = max(halo, fuse) >= fuse + tail
    for r in range(world):
        g0, g1 = sfl.slab_rows(dim_y, world, r)
        assert_bit_equal(np.load(tmp_path / f"p_{r}.npy"), want[g0:g1], f"rank {r}: owned rows")
        if has_tail and tail:
            lo_t, hi_t = (tail if r > 0 else 0), (tail if r < world - 1 else 0)
            assert_bit_equal(np.load(tmp_path / f"pt_{r}.npy"), want[g0 - lo_t:g1 + hi_t], f"rank {r}: owned rows + tail")
    prog = sfl.plan_poisson(dim_y, world, 0, iters, fuse, 3, halo, tail)
    cap = sfl.capi
    kinds = [s.kind for s in prog]
    for k, st in enumerate(prog):   # never early: every p exchange starts at the cut (or behind the tail) and follows a launch
        if st.kind == cap.STEP_EXCHANGE and st.field == cap.FIELD_PRESSURE:
            assert st.g_begin == (tail if has_tail else 0) and kinds[k - 1] == cap.STEP_SOR and kinds[k + 1] == cap.STEP_SOR


@pytest.mark.parametrize("world,kernel,fuse,iters,halo,tail", [(2, 3, 10, 80, 160, 0), (2, 3, 10, 80, 160, 1), (2, 3, 16, 40, 96, 1),
                                                              (3, 3, 8, 30, 128, 0), (2, 2, 10, 80, 160, 1), (3, 2, 16, 45, 128, 0),
                                                              (2, 3, 10, 80, 80, 1)])
def test_deep_halos_over_gloo(tmp_path, oracle, world, kernel, fuse, iters, halo, tail):
    """Round 5: the halo depth of a solve is a measured choice of up to 160 rows (ghost rows per side 64 -> 160): configuration 4's
    plan with ONE exchange (160 rows: the right-hand side's, no p exchange at all), with two (80), and deep early-exchange plans,
    executed by gloo ranks with NaN ghosts against the undivided solve -- owned rows and the tail."""
    import torch.multiprocessing as mp
    dim_x, dim_y = 23, 420
    port = _free_port()
    mp.spawn(_worker, args=(world, port, dim_x, dim_y, iters, fuse, kernel, halo, str(tmp_path), tail), nprocs=world, join=True)
    d_full = np.random.default_rng(99).standard_normal((dim_y, dim_x)).astype(np.float32)
    want = oracle.poisson_solve(d_full, 1.0, iters, OMEGA)
    sfl = importlib.import_module("esp32-fluid-simulation_amd")
    for r in range(world):
        g0, g1 = sfl.slab_rows(dim_y, world, r)
        assert_bit_equal(np.load(tmp_path / f"p_{r}.npy"), want[g0:g1], f"rank {r}: owned rows at halo {halo}")
    prog = sfl.plan_poisson(dim_y, world, 0, iters, fuse, kernel, halo, tail)
    cap = sfl.capi
    n_p = sum(st.kind == cap.STEP_EXCHANGE and st.field == cap.FIELD_PRESSURE for st in prog)
    if kernel == 3 and halo >= 2 * iters + tail:
        assert n_p == 0      # the whole solve on one halo: only the right-hand side travels
    assert int(np.load(tmp_path / "n_0.npy")[0]) == sum(st.kind == cap.STEP_EXCHANGE for st in prog)
