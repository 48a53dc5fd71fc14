#!/usr/bin/env python3
"""bench.py -- headline benchmark of the stable-fluids hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--size 8192] [--iters 80]

A "step" is ONE poisson_solve (zero fill + `iters` red-black SOR iterations, poisson.cpp:114-125)
on a `size` x `size` fp32 grid held in HBM -- BASELINE.json's headline metric "cell-iters/sec (SOR
sweep)", config[2] "8192x8192 fp32, 80 SOR iters/step" at N = 1 and config[3] (same grid, row-slab
split with RCCL halo exchange) at N > 1, i.e. STRONG scaling.  After the timed region the full
sim step (advect, divergence, solve, gradient, dye advect: ino:252-287) is timed separately and
reported as `sim_steps_per_sec` (the metric's "+ steps/sec" half).

Launch.  `python bench.py --gpus N` works as typed: for N > 1 this process touches no GPU, starts
N child processes (one per GPU; RANK / LOCAL_RANK / WORLD_SIZE in their environment), relays
rank 0's JSON line and exits with the worst child's status.  Started by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` each process finds
WORLD_SIZE in its environment and is a rank directly.  The ranks never import torch: host-side
coordination (RCCL unique id, barriers, max over ranks) is a small TCP all-gather
(esp32-fluid-simulation_amd/rendezvous.py); every halo byte moves through RCCL send/recv issued by
the C++ library.

Output: ONE JSON line on rank 0 (contract in the task statement), including
  parity        the pressure field left by the LAST TIMED solve, downloaded and compared bit for
                bit with the reference CPU loop run on the very same right-hand side (downloaded
                from the GPU); plus one whole sim step at the two small BASELINE configs (61 x 81 and
                2048^2), all four fields; the process exits non-zero on a mismatch
  roofline      the bound that binds the temporally blocked kernel (the pass over HBM: bytes really
                moved, from the committed PMC passes, / launch duration by HIP events), the VALU
                fraction beside it, and the SURVEY 8d algorithmic-bytes figure (labelled as a ratio)
  cpu_baseline  the reference's own CPU loop (oracle/_ref, or the oracle port) on this host,
                1 thread: the parity solve itself is the timed sample; N = 1 only
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "esp32-fluid-simulation_amd"
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
SOR_BYTES_PER_CELL_ITER = 16
SOR_FLOPS_PER_CELL_ITER = 8    # one relaxation of an interior cell: 3 neighbour adds, rhs subtract,
                               # scale by -1/4, two omega products and their sum (never fused);
                               # SURVEY 8a13 counts 10: dx*d is exact at dx = 1, 1-omega is hoisted
VALU_CLOCK_GHZ = 2.4           # MI355X max engine clock (MI355X_MICROARCH.md)
VALU_LANES_PER_SIMD_CLK = 32   # a plain fp32 wave64 VALU instruction issues every 2 cycles per SIMD (measured:
                               # 2.35-2.5 "cycles at 2.4 GHz" at the sustained clock, profiles/r02_ubench_pk_chain.log)
# algorithmic bytes per cell of the streaming operators (SURVEY.md 8d / BASELINE.md 3)
OP_BYTES_PER_CELL = {"advect_velocity": 16, "calculate_divergence": 12, "subtract_gradient": 20,
                     "advect_color": 32}


def synthetic_velocity(dim_x, row_begin, row_end, seed=12345, vamp=100.0):
    """Seeded per-cell hash -> velocity in [-vamp, vamp]; independent of the slab split."""
    j = np.arange(row_begin, row_end, dtype=np.uint64)[:, None]
    i = np.arange(dim_x, dtype=np.uint64)[None, :]
    out = np.empty((row_end - row_begin, dim_x, 2), np.float32)
    for comp in (0, 1):
        h = (j * np.uint64(dim_x) + i) * np.uint64(2) + np.uint64(comp) + np.uint64(seed) * np.uint64(0x9E3779B9)
        h = (h * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        h ^= h >> np.uint64(31)
        h = (h * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        h ^= h >> np.uint64(29)
        out[..., comp] = ((h >> np.uint64(40)) % np.uint64(2001)).astype(np.float32)
    out -= 1000.0
    out *= np.float32(vamp / 1000.0)
    return out


def slab_checksums(fields):
    """Two 64-bit sums (wrapping) per field over its 4-byte words: plain, and weighted by (position mod 65521) + 1 -- what the ranks
    of a multi-GPU run compare instead of shipping 1.5 GB of fields."""
    out = []
    for a in fields:
        w = np.ascontiguousarray(a).view(np.uint32).ravel().astype(np.uint64)
        k = (np.arange(w.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
        out.append([int(w.sum(dtype=np.uint64)), int((w * k).sum(dtype=np.uint64))])
    return out


def synthetic_color(dim_x, row_begin, row_end, seed=777):
    j = np.arange(row_begin, row_end, dtype=np.uint64)[:, None, None]
    i = np.arange(dim_x, dtype=np.uint64)[None, :, None]
    k = np.arange(3, dtype=np.uint64)[None, None, :]
    h = ((j * np.uint64(dim_x) + i) * np.uint64(3) + k + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    h ^= h >> np.uint64(32)
    return ((h >> np.uint64(20)) & np.uint64(0x7FFFFFFF)).astype(np.uint32)   # raw < 2^31


def cpu_path():
    from oracle import loader  # checker / baseline only
    return loader.reference() if loader.reference_available() else loader.port()


def cpu_reference_solve(d, iters, min_seconds=8.0):
    """The reference CPU loop (1 thread) on the SAME right-hand side the GPU solved: returns its
    pressure field (the parity check's expectation) and the cpu_baseline record.  The first call is
    both; it is repeated (timing only) until ~min_seconds of CPU work have been spent."""
    path = cpu_path()
    dim_y, dim_x = d.shape
    om = np.float32(1.96)
    t0 = time.perf_counter()
    want = path.poisson_solve(d, 1.0, iters, om)
    spent, reps = time.perf_counter() - t0, 1
    while spent < min_seconds and reps < 50:
        t0 = time.perf_counter()
        path.poisson_solve(d, 1.0, iters, om)
        spent += time.perf_counter() - t0
        reps += 1
    what = "unmodified reference sources" if path.kind == "reference" else "oracle C port"
    return want, {"value": dim_x * dim_y * iters * reps / spent, "unit": "cell-iters/s", "cores": 1,
                  "kind": path.kind,
                  "sample": f"{reps} x poisson_solve {dim_x}x{dim_y} fp32, {iters} iters each, on the GPU "
                            f"run's own right-hand side, 1 thread, {spent:.1f} s ({what}, g++/gcc -O2 "
                            f"-ffp-contract=off); the first of them is the parity expectation"}


def sparse_forcing_check(sfl, device, fold):
    """The input class the reference's own demo produces (ino:199, 264-276), checked inside the bench run: a
    quiescent 2048 x 2048 field, three touch dipoles in the right-hand side, 80 iterations -- the front of the solution decays
    through the denormals into cells that are still zero.  The HIP solve against the reference CPU loop, bit for bit (with
    --sor-fold: reported, not required -- that arithmetic differs there by design)."""
    dim, iters, om = 2048, 80, np.float32(1.96)
    d = np.zeros((dim, dim), np.float32)
    for fx, fy, amp in ((0.5, 0.5, 20.0), (0.13, 0.8, -7.0), (0.9, 0.07, 1.0)):
        i, j = int(dim * fx), int(dim * fy)
        d[j, i - 1] += np.float32(0.5 * amp)
        d[j, i + 1] -= np.float32(0.5 * amp)
        d[j - 1, i] += np.float32(0.25 * amp)
        d[j + 1, i] -= np.float32(0.25 * amp)
    want = cpu_path().poisson_solve(d, 1.0, iters, om)
    with sfl.Solver(dim, dim, device=device) as s:
        if fold:
            s.set_option(sfl.capi.OPT_SOR_FOLD, 1)
        s.upload(sfl.capi.FIELD_DIVERGENCE, d)
        s.poisson_solve(1.0, iters, om)
        s.synchronize()
        got = s.download(sfl.capi.FIELD_PRESSURE)
    bad = int(np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))
    return {"what": f"poisson_solve {dim}x{dim}, {iters} iters on a zero right-hand side with three touch dipoles, vs the reference CPU "
                    "loop (poisson.cpp:114-125)",
            "bit_exact": bad == 0, "mismatching_cells": bad,
            "reference_cells_nonzero_below_2^-124": int(np.count_nonzero((want != 0) & (np.abs(want) < np.float32(2.0 ** -124)))),
            "reference_cells_still_zero": int(np.count_nonzero(want == 0))}


def cpu_operator_times(dim_x, dim_y, iters, sfl=None):
    """Per-operator wall time of the reference CPU path for ONE sim step (1 thread), ms -- and, with
    `sfl` given, the same step on the GPU compared bit for bit with what the reference produced
    (parity gate on the small BASELINE configs in the same run)."""
    path = cpu_path()
    v = synthetic_velocity(dim_x, 0, dim_y)
    c = synthetic_color(dim_x, 0, dim_y)
    dt, om = np.float32(1 / 30.0), np.float32(1.96)
    out = {}

    def timed(name, fn):
        t0 = time.perf_counter()
        r = fn()
        out[name] = (time.perf_counter() - t0) * 1e3
        return r
    va = timed("advect_velocity", lambda: path.advect_vec2f(v, v, dt, True))
    d = timed("calculate_divergence", lambda: path.divergence(va, 1.0))
    p = timed("poisson_solve", lambda: path.poisson_solve(d, 1.0, iters, om))
    vp = timed("subtract_gradient", lambda: path.subtract_gradient(va, p, 1.0))
    cp = timed("advect_color", lambda: path.advect_vec3uq32(c, vp, dt, False))
    out["step"] = sum(out.values())
    rec = {"grid": [dim_x, dim_y], "iters": iters, "kind": path.kind, "ms": out}
    if sfl is not None:
        with sfl.Solver(dim_x, dim_y) as g:
            g.upload(sfl.capi.FIELD_VELOCITY, v)
            g.upload(sfl.capi.FIELD_COLOR, c)
            g.step(dt, 1.0, iters, om)
            g.synchronize()
            same = all(np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))
                       for a, b in ((g.download(sfl.capi.FIELD_VELOCITY), vp), (g.download(sfl.capi.FIELD_DIVERGENCE), d),
                                    (g.download(sfl.capi.FIELD_PRESSURE), p), (g.download(sfl.capi.FIELD_COLOR), cp)))
        rec["gpu_step_bit_exact"] = bool(same)
    return rec


def gpu_operator_times(s, iters, cells, reps=3):
    """Per-operator HIP-event time of one sim step on this rank's slab (best of `reps`), with the
    streaming operators' algorithmic bytes / time against the HBM peak next to it."""
    dt, om = np.float32(1 / 30.0), np.float32(1.96)
    ops = [("advect_velocity", lambda: s.advect_velocity(dt, True)),
           ("calculate_divergence", lambda: s.calculate_divergence(1.0)),
           ("poisson_solve", lambda: s.poisson_solve(1.0, iters, om)),
           ("subtract_gradient", lambda: s.subtract_gradient(1.0)),
           ("advect_color", lambda: s.advect_color(dt, False))]
    out = {}
    for name, fn in ops:
        fn()
        best = None
        for _ in range(reps):
            s.timer_start()
            fn()
            ms = s.timer_stop()
            best = ms if best is None else min(best, ms)
        rec = {"us": best * 1e3}
        if name in OP_BYTES_PER_CELL:
            gbs = OP_BYTES_PER_CELL[name] * cells / (best * 1e-3) / 1e9
            rec.update({"algorithmic_bytes_per_cell": OP_BYTES_PER_CELL[name], "algorithmic_GBps": gbs,
                        "frac_of_hbm_peak": gbs / HBM_PEAK_GBS})
        out[name] = rec
    return out


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def kernel_source_hash():
    """sha256 (first 16 hex digits) of the sources the fused SOR kernel is compiled from: counters measured on
    another version of them are not this kernel's."""
    import hashlib
    h = hashlib.sha256()
    for name in ("sor_fused.hip", "sor_lane.h", "sor_stream_core.h"):
        with open(os.path.join(ROOT, PKG, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# The sim step's kernels outside the solve (inside sfl_step / sfl_step_n), with SURVEY 8d's algorithmic bytes per cell of
# what each fuses: K1 + K2 = v in, v out, div out; K5 + K6 = p, v, dye in, v, dye out; the seam = p, v, dye in, dye, v', div out
STEP_KERNELS = {
    "advect_divergence_tiled_kernel": {
        "does": "advect velocity + calculate_divergence (ino:252-256 + ino:274)", "bytes_per_cell": 20,
        "match": lambda n: "advect_divergence_tiled_kernel" in n},
    "advect_vec3uq32_tiled_kernel<fuse_grad>": {
        "does": "subtract_gradient + advect dye (ino:276 + ino:281-287)", "bytes_per_cell": 44,
        "match": lambda n: "advect_vec3uq32_tiled_kernel" in n and ("<false, true" in n or "ILb0ELb1E" in n)},
    "seam_tiled_kernel": {
        "does": "sfl_step_n between two steps: subtract_gradient + advect dye of one, advect velocity + calculate_divergence "
                "of the next (the projected velocity in between is never stored)", "bytes_per_cell": 48,
        "match": lambda n: "seam_tiled_kernel" in n},
}


def step_kernel_source_hash():
    """sha256 (first 16 hex digits) of the sources the step's kernels outside the solve are compiled from."""
    import hashlib
    h = hashlib.sha256()
    for name in ("advect_tiled.hip", "advect_seam.h", "stencil_kernels.hip", "advect_math.h"):
        with open(os.path.join(ROOT, PKG, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def step_kernel_records(grid):
    """Per-kernel HBM traffic / duration of the step's kernels from the committed PMC passes (profiles/pmc_traffic.json
    "step_entries"), only those measured on the kernel sources this run uses."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("step_entries", [])
    except Exception:
        return None
    now, out = step_kernel_source_hash(), {}
    for e in table:
        if e["grid"] == list(grid) and e.get("kernel_source_sha16") == now:
            out[e["kernel"]] = {k: e[k] for k in ("does", "algorithmic_bytes_per_launch", "traffic_bytes_per_launch",
                                                  "read_bytes_per_launch", "write_bytes_per_launch", "avg_launch_us_rocprof",
                                                  "frac_of_hbm_peak", "algorithmic_frac_of_hbm_peak", "source",
                                                  "kernel_source_sha16")}   # later entries win
    return out or None


def pmc_record(grid, fuse, world):
    """Counters of the dominant kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json):
    (entry, fresh) for the last entry that matches this exact configuration -- fresh = it was measured on the
    kernel sources this run uses (kernel_source_sha16) -- or (None, False)."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["entries"]
    except Exception:
        return None, False
    best = None
    for e in table:
        if e["grid"] == list(grid) and e["fuse"] == fuse and e["n_gpus"] == world:
            best = e    # later entries (later rounds) win
    return best, bool(best) and best.get("kernel_source_sha16") == kernel_source_hash()


# untimed sim steps in front of the timed ones: the GPU's clocks, and -- on slabs -- the first 12 solves of a kind, which run on
# candidate halo depths between events (csrc/sor_executor.cpp choose_halo; the 13th decides, collectively on RCCL ranks)
SIM_PRIMING_STEPS = 16
SCHEDULES = {0: "none", 1: "in line", 2: "one launch early, behind events", 3: "in time, counted on the device"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=80)
    ap.add_argument("--dim-y", type=int, default=0, help="rows (default: size); experiments only")
    ap.add_argument("--fuse", type=int, default=0, help="SOR half-sweeps fused per launch (0 = library default)")
    ap.add_argument("--sor-kernel", type=int, default=0)
    ap.add_argument("--sor-fold", action="store_true",
                    help="SFL_OPT_SOR_FOLD = 1 for the TIMED solves: the interior relaxation's one product by -0.25f * omega (opt-in "
                         "arithmetic, not the reference's bits on sparse fields; the line's `numerics` says so).  Without it the "
                         "folded arithmetic is still measured once, after everything else, as numerics.value_with_fold")
    ap.add_argument("--no-fold-leg", action="store_true", help="skip that extra measurement")
    ap.add_argument("--sor-rows", type=int, default=0)
    ap.add_argument("--sor-halo", type=int, default=0, help="rows of p exchanged per superstep (0 = auto)")
    ap.add_argument("--lane-cells", type=int, default=0, help="cells per lane of the fused kernel (0 auto, 2)")
    ap.add_argument("--sim-steps", type=int, default=8, help="full sim steps timed after the main region")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the reference CPU solve (and with it the parity check)")
    ap.add_argument("--no-parity", action="store_true", help="N > 1: skip the reference solve on rank 0")
    ap.add_argument("--check-sim-step-parity", action="store_true",
                    help="N = 1: run the sim-step field check of the multi-GPU path as well (rank 0 replays the steps on a second, "
                         "whole-domain context; every rank compares checksums of its rows) -- always on for N > 1")
    ap.add_argument("--no-priming", action="store_true",
                    help="skip the ~80 ms of untimed solves that bring the GPU to its sustained clocks")
    ap.add_argument("--advect-kernel", type=int, default=0,
                    help="advection kernels: 0 auto, 1 one thread per cell, 2 LDS-staged tiles (A/B)")
    ap.add_argument("--no-fuse-projection", action="store_true",
                    help="sim step: separate subtract_gradient and dye-advection kernels (A/B)")
    ap.add_argument("--emulate-rank", type=int, default=-1,
                    help="with --of N: run rank R's program of an N-GPU solve ALONE on one GPU, every halo message "
                         "replaced by a self-copy of the same size on the exchange stream (sfl_comm_emulate); reports "
                         "ms per solve of that rank = the per-GPU critical path without the wire; results next to "
                         "the cuts are meaningless, so parity and the CPU baseline are skipped")
    ap.add_argument("--of", type=int, default=8, help="group size for --emulate-rank")
    ap.add_argument("--via-rccl", action="store_true",
                    help="--emulate-rank: every halo message is a real ncclSend / ncclRecv of the rank to ITSELF on a one-rank "
                         "communicator, issued through the code path a real rank takes, and the step's reductions are "
                         "ncclAllReduce on it (sfl_comm_emulate_rccl): RCCL's own kernels beside the solve's launches")
    ap.add_argument("--halo-timeout-ms", type=int, default=0, help="SFL_OPT_HALO_TIMEOUT_MS (0 = the transport's default)")
    ap.add_argument("--wire-us", type=int, default=0,
                    help="--emulate-rank: hold every emulated halo message back by this many microseconds on the "
                         "exchange stream (SFL_OPT_EMULATE_WIRE_US): how much xGMI latency does the schedule hide?")
    ap.add_argument("--no-overlap", action="store_true", help="SFL_OPT_EXCHANGE_SCHEDULE = 1: every halo exchange in line (A/B)")
    ap.add_argument("--arrival-by-event", action="store_true",
                    help="SFL_OPT_EXCHANGE_SCHEDULE = 2: early halo exchanges behind cross-stream events (round 3's scheme; the "
                         "library's own choice on RCCL ranks whose peers are other processes)")
    ap.add_argument("--arrival-in-time", action="store_true",
                    help="SFL_OPT_EXCHANGE_SCHEDULE = 3: exchanges in time, counted on the device, also where the library would not "
                         "choose them by itself (RCCL ranks whose peers are other processes)")
    ap.add_argument("--no-experiment", action="store_true",
                    help="multi-GPU runs: skip the extra attempt that times the in-time schedule beside the headline (in_time_experiment)")
    ap.add_argument("--share-device", type=int, default=None, metavar="D",
                    help="multi-rank runs on a box with FEWER GPUs than ranks: every rank process uses device D and tells RCCL it "
                         "is a host of its own (NCCL_HOSTID), so the N processes form a real N-rank communicator over RCCL's "
                         "socket transport on the loopback interface.  Proves the multi-process path (rendezvous, "
                         "ncclCommInitRank, matched send / recv between processes, the collective decisions, parity of every "
                         "rank's rows); its timings mean nothing (one GPU, host-staged messages) and the line says so")
    ap.add_argument("--launch-timeout", type=float, default=400.0,
                    help="multi-GPU runs: seconds after which the rank processes of ONE attempt are stopped (the launcher then "
                         "starts fresh ranks with the next exchange schedule; all attempts together stay under 1500 s)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous plumbing only: the ranks touch no GPU (CPU test)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# multi-GPU launch: fresh rank processes per ATTEMPT, a chain of exchange schedules to fall back through
# ---------------------------------------------------------------------------------------------
# The three schedules of a slab solve's halo exchanges give the same bits.  The HEADLINE of a multi-GPU run is what a caller of
# sfl_poisson_solve gets: the library's own choice of schedule, no flag (on RCCL ranks whose peers are other processes: one launch
# early, behind events -- ADVICE r05 / VERDICT r05 item 3).  Should that attempt fail, FRESH rank processes try every exchange in
# line (a process that touched a GPU is never re-executed; a failed attempt's processes are stopped, by PID), and the line says
# so.  The schedule the library does not pick by itself yet -- exchanges in time, counted on the device: it waits for messages
# INSIDE kernels and has never run on more than one real GPU -- is timed AFTER a successful headline as a separately labelled
# experiment (`in_time_experiment`: short, no reference solve, its pressure compared with the headline's by checksum); its
# failure costs the line nothing.
PLAN = [("library default", [], "headline"),
        ("in-line", ["--no-overlap"], "fallback"),
        ("in-time", ["--arrival-in-time", "--no-cpu-baseline", "--sim-steps", "0"], "experiment")]
TOTAL_BUDGET_S = 1500.0
EXPERIMENT_BUDGET_S = 150.0


def attempt_plan(args):
    """(mode, extra flags, role) of the attempts a multi-GPU run goes through; a schedule asked for on the command line is the
    only one tried."""
    forced = "in-time" if args.arrival_in_time else "by-event" if args.arrival_by_event else "in-line" if args.no_overlap else None
    if forced:
        return [(forced, [], "headline")]
    plan = [p for p in PLAN if not (p[2] == "experiment" and args.no_experiment)]
    if not args.halo_timeout_ms:   # a lost message of the experiment is to end as an error line, not as a stopped process
        plan = [(m, f + ["--halo-timeout-ms", "20000"] if r == "experiment" else f, r) for m, f, r in plan]
    return plan


def mode_of(args):
    return "in-time" if args.arrival_in_time else "by-event" if args.arrival_by_event else "in-line" if args.no_overlap else "library default"


class Attempts:
    """The bookkeeping both launchers share: which attempt comes next, what the line will say."""

    def __init__(self, args):
        self.plan, self.k = attempt_plan(args), -1
        self.t_begin, self.failures, self.headline, self.experiment = time.monotonic(), [], None, None
        self.args = args

    def next(self):
        """(mode, flags, role, seconds allowed) of the next attempt to run, or None when there is nothing left to try."""
        while self.k + 1 < len(self.plan):
            self.k += 1
            mode, flags, role = self.plan[self.k]
            if role == "fallback" and self.headline is not None:
                continue
            if role == "experiment" and self.headline is None:
                continue
            left = TOTAL_BUDGET_S - (time.monotonic() - self.t_begin)
            if left < 30.0:
                if role != "experiment":
                    self.failures.append({"mode": mode, "why": "not tried: the total budget of the launcher was spent"})
                else:
                    self.experiment = {"failed": "not tried: the total budget of the launcher was spent"}
                continue
            limit = min(self.args.launch_timeout, left - 10.0)
            if role == "experiment":
                limit = min(limit, EXPERIMENT_BUDGET_S)
            return mode, list(flags), role, limit
        return None

    def done(self, status, line, why):
        """Record the outcome of the attempt next() handed out."""
        mode, _, role = self.plan[self.k]
        if role == "experiment":
            self.experiment = summarise_experiment(line, self.headline) if status == 0 and line else {"failed": why, "status": status}
            return
        if status == 0 and line:
            self.headline = annotate(line, mode, list(self.failures))
        else:
            self.failures.append({"mode": mode, "status": status, "why": why})

    def upcoming(self):
        for mode, _, role in self.plan[self.k + 1:]:
            if role == "fallback" and self.headline is None:
                return mode
        return None

    def result(self):
        """The ONE JSON line of the run, or None when no attempt produced a headline."""
        if self.headline is None:
            return None
        if self.experiment is None:
            return self.headline
        try:
            d = json.loads(self.headline)
        except ValueError:      # (a line that is not JSON is still the run's line: never lose the headline to the experiment)
            return self.headline
        d["in_time_experiment"] = self.experiment
        return json.dumps(d)


def summarise_experiment(line, headline):
    """What the experiment's own line says, next to the headline it is compared with (same grid, same right-hand side)."""
    try:
        e, h = json.loads(line), json.loads(headline)
    except ValueError:
        return {"failed": "the experiment's line could not be read"}
    if e.get("dry_run"):
        return {"dry_run": True, "mode": e.get("mode")}
    same = e.get("pressure_checksums") is not None and e.get("pressure_checksums") == h.get("pressure_checksums")
    return {"what": "the same solves with exchanges IN TIME, counted on the device (SFL_OPT_EXCHANGE_SCHEDULE = 3) -- not what the "
                    "library picks by itself on RCCL ranks whose peers are other processes; fresh rank processes, no reference solve",
            "value": e.get("value"), "unit": e.get("unit"), "ms_per_step": e.get("ms_per_step"),
            "exchange_schedule": (e.get("config") or {}).get("exchange_schedule"),
            "halo_exchanges_per_solve": (e.get("config") or {}).get("halo_exchanges_per_solve"),
            "pressure_matches_headline_bit_for_bit": bool(same),
            "vs_headline": (e["value"] / h["value"]) if e.get("value") and h.get("value") else None}


class Child:
    """One rank process of one attempt: stdout kept (the JSON line travels on rank 0's), stderr relayed to ours as it comes,
    its last non-empty line remembered (the reason, should the rank fail)."""

    def __init__(self, cmd, env, keep_stdout):
        import threading
        self.p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if keep_stdout else sys.stderr, stderr=subprocess.PIPE,
                                  text=True)
        self.lines, self.last_err = [], ""
        self._threads = [threading.Thread(target=self._relay, daemon=True)]
        if keep_stdout:
            self._threads.append(threading.Thread(target=lambda: self.lines.extend(self.p.stdout), daemon=True))
        for t in self._threads:
            t.start()

    def _relay(self):
        for line in self.p.stderr:
            sys.stderr.write(line)
            if line.strip():
                self.last_err = line.strip()[-300:]

    def poll(self):
        return self.p.poll()

    def stop(self):
        if self.p.poll() is None:
            self.p.terminate()
            t_kill = time.monotonic() + 5.0
            while self.p.poll() is None and time.monotonic() < t_kill:
                time.sleep(0.05)
            if self.p.poll() is None:
                self.p.kill()
        self.p.wait()

    def finish(self):
        for t in self._threads:
            t.join(timeout=5.0)

    def json_line(self):
        out = None
        for l in self.lines:
            l = l.rstrip("\n")
            if l.startswith("{"):
                out = l
            elif l:
                print(l, file=sys.stderr)
        return out


def worker_argv(extra):
    """The rank processes run this file again; under tools/with_lib.py (a variant build of the library, SFL_WITH_LIB) they run it
    through with_lib.py as well, so that a multi-rank A/B run does not silently measure the product library."""
    me = os.path.abspath(__file__)
    variant = os.environ.get("SFL_WITH_LIB")
    head = [sys.executable, os.path.join(os.path.dirname(me), "tools", "with_lib.py"), variant, me] if variant else [sys.executable, me]
    return head + sys.argv[1:] + list(extra)


def launch_ranks(args, extra=(), deadline_s=None):
    """ONE attempt of `python bench.py --gpus N` without a launcher around it: start one child per GPU and SUPERVISE them.
    This process never touches a GPU (no HIP call, not even a device count): the children are fresh processes, nothing that
    has initialised a GPU is ever replaced or forked.  RCCL send / recv has no timeout: a rank that dies after the communicator
    is up would leave its neighbours waiting for ever, so all children are polled; the first one that fails (or the deadline)
    takes the others down with it -- they are this process's own children, addressed by PID.
    Returns (status, json line or None, reason or None): status 0 = every rank finished, 124 = deadline."""
    import shutil
    import socket
    import tempfile
    with socket.socket() as s:      # a free port for MASTER_PORT (RCCL's bootstrap picks its own)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n = args.gpus
    private = tempfile.mkdtemp(prefix="sfl_bench_")     # mode 0700: rendezvous file + parity arrays live here
    base = dict(os.environ)
    base.update({"WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                 "LOCAL_WORLD_SIZE": str(n), "SFL_RDZV_KEY": f"{os.getpid()}_{port}", "SFL_RDZV_DIR": private,
                 "SFL_BENCH_WORKER": "1",
                 "HSA_ENABLE_IPC_MODE_LEGACY": base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    cmd = worker_argv(extra)
    deadline = time.monotonic() + (deadline_s if deadline_s is not None else args.launch_timeout)
    kids, worst, why = [], 0, None
    try:
        for r in range(n):
            kids.append(Child(cmd, dict(base, RANK=str(r), LOCAL_RANK=str(r)), keep_stdout=(r == 0)))
        failed_at = None
        while True:
            codes = [k.poll() for k in kids]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad and failed_at is None:
                worst, failed_at = bad[0][1], time.monotonic()
                print(f"bench.py launcher: a rank exited with status {worst}; stopping the others", file=sys.stderr)
            if all(c is not None for c in codes):
                break
            if (failed_at is not None and time.monotonic() - failed_at > 2.0) or time.monotonic() > deadline:
                if failed_at is None:
                    worst = 124
                    why = f"no result after {deadline_s if deadline_s is not None else args.launch_timeout:.0f} s"
                    print(f"bench.py launcher: {why}; stopping the ranks", file=sys.stderr)
                for k in kids:
                    k.stop()
                break
            time.sleep(0.05)
        for k in kids:
            k.finish()
        for r, k in enumerate(kids):
            rc = k.p.returncode
            if rc not in (0, None) and (worst == 0 or (why is None and worst != 124)):
                worst = worst or rc
            if rc not in (0, None) and why is None and rc > 0:
                why = f"rank {r} exited with status {rc}: {k.last_err or 'no message'}"
        if worst != 0 and why is None:
            why = next((f"rank {r} ended with status {k.p.returncode}: {k.last_err or 'no message'}"
                        for r, k in enumerate(kids) if k.p.returncode not in (0, None)), "a rank failed")
    finally:
        for k in kids:
            if k.poll() is None:
                k.stop()
        shutil.rmtree(private, ignore_errors=True)
    line = kids[0].json_line() if kids else None
    if worst == 0 and not line:
        worst, why = 1, "rank 0 printed no JSON line"
    return worst, line, why


def annotate(line, mode, failures):
    """The winning attempt's JSON line with the schedule that produced it and the attempts that did not."""
    try:
        d = json.loads(line)
    except ValueError:
        return line
    d["exchange_mode"] = mode
    d["fallback_from"] = failures
    return json.dumps(d)


def launch_with_fallback(args):
    """`python bench.py --gpus N`, N > 1, as typed: the attempts of attempt_plan() within TOTAL_BUDGET_S; non-zero with every
    attempt's reason when none gives a headline."""
    run = Attempts(args)
    status = 1
    while True:
        nxt = run.next()
        if nxt is None:
            break
        mode, flags, role, limit = nxt
        status, line, why = launch_ranks(args, flags if len(run.plan) > 1 else [], limit)
        run.done(status, line, why)
        if role != "experiment" and run.headline is None and run.upcoming():
            print(f"bench.py launcher: schedule '{mode}' failed ({why}); starting fresh ranks with '{run.upcoming()}'", file=sys.stderr)
        if role == "experiment" and "failed" in (run.experiment or {}):
            print(f"bench.py launcher: the in-time experiment failed ({why}); the headline stands", file=sys.stderr)
    line = run.result()
    if line:
        print(line, flush=True)
        return 0
    print("bench.py launcher: every exchange schedule failed: " + json.dumps(run.failures), file=sys.stderr)
    return status or 1


def supervise_rank(args):
    """Started per rank by `python -m torch.distributed.run ... bench.py --gpus N` (the driver's command line): this process
    is rank R's SUPERVISOR.  It never touches a GPU; per attempt it starts ONE fresh worker (the same command line + the
    schedule's flag), and the N supervisors agree -- over the TCP rendezvous, twice a second -- on whether the attempt is
    still running, has succeeded on every rank, or has failed on any (then every supervisor stops its worker and all go on to
    the next schedule together).  The launcher around us would end the whole run at the first non-zero exit: supervisors only
    exit non-zero when every schedule has failed."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"])
    from importlib import import_module
    Rendezvous = import_module(PKG + ".rendezvous").Rendezvous
    base_key = os.environ.get("SFL_RDZV_KEY") or f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}"
    sup = Rendezvous(rank, world, key=base_key + "_sup")
    run = Attempts(args)
    try:
        while True:
            nxt = run.next()
            nxt = sup.all_gather(nxt)[0]          # rank 0's clock decides for everybody
            if nxt is None:
                break
            mode, flags, role, limit = nxt
            if rank != 0:                          # (keep every rank's bookkeeping on the attempt rank 0 named)
                run.k = next(i for i, p in enumerate(run.plan) if p[0] == mode)
            extra = flags if len(run.plan) > 1 else []
            env = dict(os.environ, SFL_BENCH_WORKER="1", SFL_RDZV_KEY=f"{base_key}_w{run.k}",
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            kid = Child(worker_argv(extra), env, keep_stdout=(rank == 0))
            deadline = time.monotonic() + limit
            verdict = None
            while verdict is None:
                time.sleep(0.25)
                rc = kid.poll()
                mine = "running" if rc is None else ("ok" if rc == 0 else f"rank {rank} exited with status {rc}: {kid.last_err or 'no message'}")
                if rank == 0 and rc is None and time.monotonic() > deadline:
                    mine = f"no result after {limit:.0f} s"
                states = sup.all_gather(mine)
                broken = [st for st in states if st not in ("running", "ok")]
                if broken:
                    verdict = broken[0]
                elif all(st == "ok" for st in states):
                    verdict = "ok"
            kid.stop()
            kid.finish()
            line = kid.json_line() if (rank == 0 and verdict == "ok") else None
            if verdict == "ok" and rank == 0 and not line:
                verdict = "rank 0 printed no JSON line"
            verdict = sup.all_gather(verdict)[0]   # (rank 0 may have found its line missing)
            ok = verdict == "ok"
            # every rank keeps the same bookkeeping; only rank 0 holds the lines themselves
            run.done(0 if ok else 1, line if rank == 0 else ("{}" if ok else None), None if ok else verdict)
            if rank == 0 and role != "experiment" and not ok and run.upcoming():
                print(f"bench.py: schedule '{mode}' failed ({verdict}); starting fresh ranks with '{run.upcoming()}'", file=sys.stderr)
            if rank == 0 and role == "experiment" and not ok:
                print(f"bench.py: the in-time experiment failed ({verdict}); the headline stands", file=sys.stderr)
        if run.headline is not None:
            if rank == 0:
                print(run.result(), flush=True)
            sup.barrier()
            return 0
        if rank == 0:
            print("bench.py: every exchange schedule failed: " + json.dumps(run.failures), file=sys.stderr)
        return 1
    finally:
        sup.close()


# ---------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------
def run_rank(args):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    shared_device = args.share_device if args.share_device is not None else os.environ.get("SFL_BENCH_DEVICE")
    if shared_device is not None:   # --share-device: several ranks on one device
        local_rank = int(shared_device)
        # RCCL refuses two ranks of a communicator on one device OF ONE HOST ("Duplicate GPU detected"): every rank says it is
        # a host of its own, and the ranks talk through RCCL's socket transport over the loopback interface -- a real N-rank
        # communicator of N processes on a one-GPU box.  Correctness of the multi-process path, not its speed.
        os.environ.setdefault("NCCL_HOSTID", f"sfl-shared-device-rank-{rank}")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    args.gpus = world

    sfl = importlib.import_module(PKG)
    from importlib import import_module
    rdzv = import_module(PKG + ".rendezvous").Rendezvous(rank, world)
    capi = sfl.capi

    if args.dry_run:   # plumbing check without a GPU: broadcast, barrier, max, one JSON line
        token = rdzv.broadcast_bytes(os.urandom(16) if rank == 0 else None)
        if os.environ.get("SFL_BENCH_TEST_FAIL_RANK") == str(rank):   # launcher test: a rank dies mid-run
            os._exit(7)
        if os.environ.get("SFL_BENCH_TEST_HANG"):                      # launcher test: ranks that never finish
            time.sleep(3600)
        if mode_of(args) in os.environ.get("SFL_BENCH_TEST_FAIL_MODES", "").split(","):   # launcher test: a schedule that fails
            print(f"bench.py rank {rank}: schedule '{mode_of(args)}' made to fail by the test", file=sys.stderr)
            os._exit(9)
        if mode_of(args) in os.environ.get("SFL_BENCH_TEST_HANG_MODES", "").split(","):   # ... or never finishes
            time.sleep(3600)
        rdzv.barrier()
        top = rdzv.max([float(rank), 1.0])
        ranks = rdzv.all_gather({"rank": rank, "token": token.hex(), "device": local_rank, "hostid": os.environ.get("NCCL_HOSTID")})
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "max_rank": top[0], "mode": mode_of(args),
                              "tokens_agree": len({r["token"] for r in ranks}) == 1,
                              "ranks": [r["rank"] for r in ranks], "devices": [r["device"] for r in ranks],
                              "rccl_host_ids": [r["hostid"] for r in ranks]}), flush=True)
        rdzv.barrier()
        rdzv.close()
        return 0

    if sfl.device_count() <= local_rank:
        sys.exit(f"bench.py rank {rank}: device {local_rank} not visible ({sfl.device_count()} devices): "
                 "the product path has no CPU fallback")

    size, iters = args.size, args.iters
    dim_y = args.dim_y or size
    emulate = args.emulate_rank >= 0
    if emulate:
        if world != 1:
            sys.exit("--emulate-rank runs on ONE GPU")
        s = sfl.Solver(size, dim_y, device=local_rank, rank=args.emulate_rank, nranks=args.of)
        if args.via_rccl:
            with sfl.stdout_to_stderr():   # (RCCL's banner goes to stdout: keep it off the JSON line)
                s.comm_emulate_rccl()
        else:
            s.comm_emulate()
        if args.wire_us:
            s.set_option(capi.OPT_EMULATE_WIRE_US, args.wire_us)
        args.no_cpu_baseline = True
    else:
        s = sfl.Solver(size, dim_y, device=local_rank, rank=rank, nranks=world)
    if args.no_overlap:
        s.set_option(capi.OPT_EXCHANGE_SCHEDULE, capi.SCHEDULE_IN_LINE)
    if args.arrival_by_event:
        s.set_option(capi.OPT_EXCHANGE_SCHEDULE, capi.SCHEDULE_BY_EVENT)
    if args.arrival_in_time:
        s.set_option(capi.OPT_EXCHANGE_SCHEDULE, capi.SCHEDULE_IN_TIME)
    if args.halo_timeout_ms:
        s.set_option(capi.OPT_HALO_TIMEOUT_MS, args.halo_timeout_ms)
    for opt, val in ((capi.OPT_SOR_FUSE, args.fuse), (capi.OPT_SOR_KERNEL, args.sor_kernel),
                     (capi.OPT_SOR_ROWS, args.sor_rows), (capi.OPT_SOR_LANE_CELLS, args.lane_cells),
                     (capi.OPT_SOR_HALO, args.sor_halo), (capi.OPT_ADVECT_KERNEL, args.advect_kernel)):
        if val:
            s.set_option(opt, val)
    if args.no_fuse_projection:
        s.set_option(capi.OPT_FUSE_PROJECTION, 0)
    if args.sor_fold:
        s.set_option(capi.OPT_SOR_FOLD, 1)
    if world > 1:
        uid = rdzv.broadcast_bytes(sfl.comm_unique_id() if rank == 0 else None)
        # RCCL prints a version banner on stdout while the communicator comes up; keep stdout
        # clean for the ONE JSON line by pointing fd 1 at stderr for the duration of the call
        with sfl.stdout_to_stderr():
            s.comm_attach(uid)

    # synthetic inputs, resident in HBM before anything is timed
    s.upload(capi.FIELD_VELOCITY, synthetic_velocity(size, s.row_begin, s.row_end))
    s.upload(capi.FIELD_COLOR, synthetic_color(size, s.row_begin, s.row_end))
    s.calculate_divergence(1.0)     # right-hand side = divergence of the velocity
    s.synchronize()

    omega = np.float32(1.96)
    cells = size * dim_y
    if emulate:
        cells = size * (s.row_end - s.row_begin)   # what this rank relaxes per iteration

    def timed_region():
        """W untimed + K timed solves, barrier + device sync on both sides, max over ranks."""
        for _ in range(args.warmup):
            s.poisson_solve(1.0, iters, omega)
        s.synchronize()
        rdzv.barrier()
        t0 = time.perf_counter()
        s.timer_start()
        for _ in range(args.steps):
            s.poisson_solve(1.0, iters, omega)
        ev_ms = s.timer_stop()
        s.synchronize()
        rdzv.barrier()
        elapsed = time.perf_counter() - t0
        return rdzv.max([elapsed, ev_ms])

    s.poisson_solve(1.0, iters, omega)   # lazy allocations, code objects
    s.synchronize()

    # Clock priming (untimed, reported as `priming_solves`): after set-up (host-side data
    # generation, PCIe uploads) the GPU sits at idle clocks and needs ~30 ms of load to reach its
    # sustained rate (tools/solve_sequence_probe.py: 2.7, 2.5, 2.4 ... 1.93 ms per solve over the
    # first 15 solves).  A running simulation lives at the sustained rate, so `value` is measured
    # after ~80 ms of the same solves; the same W + K region measured BEFORE them is reported as
    # `value_unprimed`.  The count is agreed across ranks (every rank issues the same exchanges).
    priming, unprimed = 0, None
    if not args.no_priming:
        time.sleep(0.05)                 # let the clocks fall back, as after any host-side pause
        elapsed_cold, _ = timed_region()
        unprimed = cells * iters * args.steps / elapsed_cold
        t_one = rdzv.max([elapsed_cold / args.steps])[0]
        # launch-bound tiny grids do not load the GPU at all: nothing to ramp (and hundreds of queued
        # launches only disturb the host-side launch path that bounds them)
        priming = int(min(400, max(4, 0.08 / t_one))) if t_one >= 0.5e-3 else 0
        for _ in range(priming):
            s.poisson_solve(1.0, iters, omega)
    elapsed, ev_ms = timed_region()
    info = s.last_solve_info()
    schedule = s.get_option(capi.OPT_EXCHANGE_SCHEDULE)
    exchange_us = s.get_option(capi.OPT_MEASURED_WIRE_US)    # -1: nothing to measure (one GPU)

    # ---- parity of the timed configuration: the p the last timed solve left behind is downloaded
    # NOW (with the right-hand side it was solved for); the reference CPU loop runs after all GPU
    # timing is done -- it keeps the host busy for seconds, during which the GPU clocks fall back
    parity, cpu_rec = None, None
    want_parity = not args.no_cpu_baseline and not (world > 1 and args.no_parity)
    got = None
    if want_parity or (world > 1 and not emulate):
        got = s.download(capi.FIELD_PRESSURE)
    if want_parity:
        d_own = s.download(capi.FIELD_DIVERGENCE)
    # every rank's rows of the timed solve's pressure as checksums (multi-GPU lines: two attempts of one run -- the headline and the
    # in-time experiment -- are compared through them without a second reference solve)
    p_sums = rdzv.all_gather(slab_checksums([got])[0] if got is not None else None) if (world > 1 and not emulate) else None

    # ---- full sim step, timed separately (not part of `value`) ---------------------------------
    # A slab reports a back-trace that left its advection halo at synchronize(); every rank still
    # issues the same launches and exchanges, so the failure is recorded, agreed on collectively and
    # never deadlocks a barrier.
    sim_sps, sim_note, sim_sps_calls = None, None, None
    if args.sim_steps > 0:
        failed = []

        def sync_soft():
            try:
                s.synchronize()
            except sfl.SflError as e:
                failed.append(str(e))

        # (slabs run on the automatic advection halo, the default: exact for the velocity advection, guessed and
        # checked after the step for the dye -- no host round trip inside a step, never an SFL_ERR_HALO)
        dtf = np.float32(1 / 30.0)
        # the downloads above left the GPU idle: bring it back to its sustained clocks with untimed steps
        for _ in range(1 if args.no_priming else SIM_PRIMING_STEPS):
            s.step(dtf, 1.0, iters, omega)
        sync_soft()
        rdzv.barrier()
        t1 = time.perf_counter()
        for _ in range(args.sim_steps):
            s.step(dtf, 1.0, iters, omega)
        sync_soft()
        rdzv.barrier()
        sim_t, bad_step = rdzv.max([time.perf_counter() - t1, 1.0 if failed else 0.0])
        if bad_step:
            sim_note = failed[0] if failed else "a peer rank reported an advection-halo overflow"
        else:
            sim_sps_calls = args.sim_steps / sim_t
            sim_sps = sim_sps_calls
            # the sim task's loop calls the step back to back (ino:249-289): the same steps as ONE sfl_step_n call, which on
            # a whole-domain context joins the last kernel of a step and the first of the next (SFL_OPT_STEP_SEAMS); the same
            # results (tests/test_gpu_parity.py::test_step_n_*), timed the same way.  Slab ranks run n x sfl_step inside.
            t1 = time.perf_counter()
            s.step_n(args.sim_steps, dtf, 1.0, iters, omega)
            sync_soft()
            rdzv.barrier()
            sim_tn, bad_step = rdzv.max([time.perf_counter() - t1, 1.0 if failed else 0.0])
            if not bad_step:
                sim_sps = args.sim_steps / sim_tn

    # ---- the sim step's fields across ranks: the slabs' velocity, colour and pressure after all those steps against
    # the same steps on ONE whole-domain context, which rank 0 runs on its own GPU (that path is checked against the reference CPU
    # code by the one-GPU run and the test suite); compared through per-slab checksums.  Collective calls stay outside every
    # try: a rank that fails locally still takes part, and the bench line is never lost to the checker.
    step_parity = None
    if args.sim_steps > 0 and sim_sps is not None and not emulate and not args.no_parity and (world > 1 or args.check_sim_step_parity):
        names = (capi.FIELD_VELOCITY, capi.FIELD_COLOR, capi.FIELD_PRESSURE)
        mine, err = None, None
        try:
            mine = slab_checksums([s.download(f) for f in names])
        except Exception as e:   # noqa: BLE001
            err = repr(e)
        spans = rdzv.all_gather([s.row_begin, s.row_end])
        want = None
        if rank == 0:
            try:
                with sfl.Solver(size, dim_y, device=local_rank) as ref:
                    ref.upload(capi.FIELD_VELOCITY, synthetic_velocity(size, 0, dim_y))
                    ref.upload(capi.FIELD_COLOR, synthetic_color(size, 0, dim_y))
                    for _ in range((1 if args.no_priming else SIM_PRIMING_STEPS) + args.sim_steps):
                        ref.step(dtf, 1.0, iters, omega)
                    ref.step_n(args.sim_steps, dtf, 1.0, iters, omega)
                    ref.synchronize()
                    whole = [ref.download(f) for f in names]
                want = [slab_checksums([f[b:e] for f in whole]) for b, e in spans]
                del whole
            except Exception as e:   # noqa: BLE001
                err = repr(e)
        want = rdzv.all_gather(want)[0]
        same = rdzv.all_gather(None if (want is None or mine is None) else bool(want[rank] == mine))
        errs = [e for e in rdzv.all_gather(err) if e]
        if errs or any(x is None for x in same):
            step_parity = {"error": errs[0] if errs else "no checksums"}
        else:
            step_parity = {"what": f"velocity, colour and pressure after {(1 if args.no_priming else SIM_PRIMING_STEPS) + 2 * args.sim_steps} sim steps: "
                                   "every rank's rows against the same steps on one whole-domain context (rank 0's GPU), by checksums",
                           "bit_exact": all(same), "ranks_differing": [r for r, ok in enumerate(same) if not ok]}

    op_us = None
    if world == 1 and args.sim_steps > 0:
        op_us = gpu_operator_times(s, iters, cells)

    # ---- the opt-in arithmetic beside the default (numerics.value_with_fold): the same W + K region with SFL_OPT_SOR_FOLD flipped,
    # after everything the line's other numbers come from; one GPU only (RCCL ranks would have to agree on the option collectively)
    other_arith = None
    if world == 1 and not emulate and not args.no_fold_leg:
        s.set_option(capi.OPT_SOR_FOLD, 0 if args.sor_fold else 1)
        for _ in range(max(priming, 4)):
            s.poisson_solve(1.0, iters, omega)
        el2, ev2 = timed_region()
        other_arith = {"value": cells * iters * args.steps / el2, "ms_per_solve_hip_events": ev2 / args.steps}
        s.set_option(capi.OPT_SOR_FOLD, 1 if args.sor_fold else 0)

    # ---- parity: the reference CPU loop on the downloaded right-hand side ----------------------
    if want_parity:
        if world == 1:
            want, cpu_rec = cpu_reference_solve(d_own, iters)
            bad = int(np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))
        else:
            # the ranks assemble the right-hand side in shared memory, rank 0 runs the reference on
            # the whole domain (the only way to get exact expectations), every rank checks its slab
            # (files of this run only: created exclusively -- never through a planted link -- under a name no other
            # run uses, and always unlinked, also when a rank fails in between)
            shm = f"/dev/shm/sfl_bench_{os.environ.get('SFL_RDZV_KEY', str(os.getppid()))}_{os.getuid()}"
            d_all = None
            try:
                if rank == 0:
                    for suffix in ("_d.npy", "_p.npy"):
                        os.close(os.open(shm + suffix, os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW | os.O_WRONLY, 0o600))
                    d_all = np.lib.format.open_memmap(shm + "_d.npy", mode="w+", dtype=np.float32,
                                                      shape=(dim_y, size))
                rdzv.barrier()
                if rank != 0:
                    d_all = np.load(shm + "_d.npy", mmap_mode="r+")
                d_all[s.row_begin:s.row_end] = d_own
                d_all.flush()
                rdzv.barrier()
                if rank == 0:
                    want_all, _ = cpu_reference_solve(np.ascontiguousarray(d_all), iters, min_seconds=0.0)
                    with open(shm + "_p.npy", "wb") as f:
                        np.save(f, want_all)
                    del want_all
                rdzv.barrier()
                want = np.ascontiguousarray(np.load(shm + "_p.npy", mmap_mode="r")[s.row_begin:s.row_end])
                bad = int(np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))
                bad = int(sum(rdzv.all_gather(bad)))
            finally:
                del d_all
                if rank == 0:
                    for suffix in ("_d.npy", "_p.npy"):
                        try:
                            os.unlink(shm + suffix)
                        except OSError:
                            pass
        parity = {"config": f"poisson_solve {size}x{dim_y} fp32, {iters} iters, omega 1.96, dx 1, the timed "
                            f"solve's own output vs the reference CPU loop on the same rhs (poisson.cpp:114-125)",
                  "bit_exact": bad == 0, "cells": cells, "mismatching_cells": bad,
                  "tolerance": "1e-5 relative allowed by north_star; asserted 0 ulp"}

    rc = 0
    if emulate:
        name, cus, mem = sfl.device_info(local_rank)
        print(json.dumps({
            "emulated_rank": args.emulate_rank, "of": args.of, "grid": [size, dim_y], "iters": iters,
            "rows_owned": s.row_end - s.row_begin, "ms_per_solve": elapsed / args.steps * 1e3,
            "ms_per_solve_hip_events": ev_ms / args.steps, "ms_per_solve_unprimed":
            (cells * iters / unprimed * 1e3) if unprimed else None,
            "cell_iters_per_sec_of_this_rank": cells * iters * args.steps / elapsed,
            "sor_launches_per_solve": info["launches"], "halo_exchanges_per_solve": info["exchanges"],
            "half_sweeps_fused_per_launch": info["fuse"], "overlap": not args.no_overlap,
            "emulated_wire_us": args.wire_us, "transport": "rccl-to-self" if args.via_rccl else "copy-kernel",
            "exchange_schedule": SCHEDULES.get(schedule, schedule),
            "halo_rows_per_superstep": info["halo"], "measured_exchange_latency_us": exchange_us,
            "sim_step_us": (1e6 / sim_sps) if sim_sps else None, **({"sim_steps_note": sim_note} if sim_note else {}),
            "note": "one rank's program alone on one GPU, halo messages as self-copies of the same size on the "
                    "exchange stream (sfl_comm_emulate); values next to the cuts are meaningless",
            "device": name}), flush=True)
        s.close()
        rdzv.close()
        return 0
    if rank == 0:
        value = cells * iters * args.steps / elapsed
        launches = max(info["launches"], 1)
        # dominant kernel: one launch relaxes every owned cell `fuse`/2 times
        avg_launch_s = (ev_ms / 1e3) / (args.steps * launches)
        bytes_per_launch = SOR_BYTES_PER_CELL_ITER * (cells / world) * iters / launches
        algorithmic_gbs = bytes_per_launch / avg_launch_s / 1e9
        name, cus, mem = sfl.device_info(local_rank)
        pmc_any, pmc_fresh = pmc_record((size, dim_y), info["fuse"], world)
        pmc = pmc_any if pmc_fresh else None    # counters of another kernel version are not quoted
        # The contract names HBM or MFMA as the bound; for this stencil it is HBM (DESIGN.md 4.1 has the
        # finer picture: a launch lasts as long as one wave's chain of iterations, ~0.87 of the access
        # pattern's memory floor).  `achieved` = HBM bytes the launch really moves / its duration, the
        # bytes from the committed rocprofv3 PMC passes when one matches this configuration, else the
        # compulsory 12 B per cell (p in, rhs in, p out).
        traffic = pmc["traffic_bytes_per_launch"] if pmc else None
        compulsory = 12.0 * cells / world
        moved = traffic if traffic else compulsory
        hbm_gbs = moved / avg_launch_s / 1e9
        # VALU side: the reference's 8 individually rounded fp32 operations per relaxation
        # (poisson.cpp:63-112; contraction to FMA would change results) against half of the chip's
        # fp32 vector peak (an FMA counts two): a plain wave64 instruction issues every 2 cycles per
        # SIMD, measured (profiles/r02_ubench_pk_chain.log).
        valu_peak = cus * 4 * VALU_LANES_PER_SIMD_CLK * VALU_CLOCK_GHZ / 1e3
        valu_achieved = value / world * SOR_FLOPS_PER_CELL_ITER / 1e12
        useful_insts = SOR_FLOPS_PER_CELL_ITER * (cells / world) * iters / launches / 64
        roofline = {
            "bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": hbm_gbs / HBM_PEAK_GBS,
            # the same with the COMPULSORY bytes only -- p in, rhs in, p out: 12 B per cell and launch; what `frac` has on top of
            # it are the halo rows and columns neighbouring tiles re-read
            "frac_useful": compulsory / avg_launch_s / 1e9 / HBM_PEAK_GBS,
            "achieved_definition": ("HBM bytes per launch from rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE) / "
                                    "launch duration by HIP events" if traffic else
                                    "compulsory 12 B per cell per launch (no PMC pass committed for this "
                                    "configuration) / launch duration by HIP events"),
            "traffic": traffic,
            # the row of `traffic_source` the bytes can be recomputed from: FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024
            "traffic_row": pmc.get("kernel") if pmc else None,
            "traffic_source": pmc["source"] if pmc else ("stale: " + pmc_any["source"] + " was measured on other kernel "
                                                         "sources; compulsory bytes used" if pmc_any else None),
            "kernel_source_sha16": kernel_source_hash(),
            # the same bytes over the committed profile's OWN steady-state launch time (rocprofv3 kernel trace)
            "frac_from_profile": (traffic / (pmc["avg_launch_us_rocprof"] * 1e-6) / 1e9 / HBM_PEAK_GBS)
            if pmc and pmc.get("avg_launch_us_rocprof") else None,
            "avg_launch_us_profile": pmc.get("avg_launch_us_rocprof") if pmc else None,
            "compulsory_bytes_per_launch": compulsory,
            "kernel": "sor_fused_kernel" if info["fuse"] > 1 else "sor_half_sweep_kernel",
            "avg_launch_us": avg_launch_s * 1e6,
            # SURVEY 8d's figure for an UNFUSED sweep (16 B per cell-iteration); a temporally blocked
            # kernel moves a fraction of it, so this is a ratio, not a fraction of a roofline
            "algorithmic": {"bytes_per_cell_iter": SOR_BYTES_PER_CELL_ITER,
                            "bytes_per_launch": bytes_per_launch, "GBps": algorithmic_gbs,
                            "algorithmic_vs_hbm_peak": algorithmic_gbs / HBM_PEAK_GBS,
                            "note": "ratio of unfused-sweep bytes to the HBM peak, > 1 by temporal "
                                    "blocking; not a roofline fraction"},
            "valu": {"flops_per_cell_iter": SOR_FLOPS_PER_CELL_ITER, "achieved": valu_achieved,
                     "peak": valu_peak, "unit": "TFLOP/s", "frac": valu_achieved / valu_peak,
                     "peak_definition": f"{cus} CUs x 4 SIMDs x {VALU_LANES_PER_SIMD_CLK} fp32 lanes per clock x "
                                        f"{VALU_CLOCK_GHZ} GHz = half of the 157.3 TFLOP/s fp32 vector peak (no "
                                        "FMA: the reference rounds every product and sum)",
                     "issued_over_useful": (pmc["valu_wave_insts_per_launch"] / useful_insts)
                     if pmc and pmc.get("valu_wave_insts_per_launch") else None,
                     "note": "the package runs at its 1400 W power cap under this kernel and clocks 1.6-2.1 GHz depending "
                             "on the box (s_memtime against s_memrealtime in every wave, rocm-smi: "
                             "profiles/r03_clock_probe_ns16_8192.txt, r03_clock_smi_during_kernel.txt); the peak above "
                             "is the nominal 2.4 GHz; 'useful' counts the reference's 8 rounded operations per relaxation "
                             "(poisson.cpp:107-111); the kernel's interior path issues those 8 (7 with --sor-fold: -0.25f "
                             "folded into omega, csrc/sor_stream_core.h relax)"},
        }
        out = {
            "metric": "cell-iters/sec (SOR sweep)", "value": value, "unit": "cell-iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "priming_solves": priming,
            "value_unprimed": unprimed,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"poisson_solve {size}x{dim_y} fp32, {iters} red-black SOR iters/step, "
                                   f"omega 1.96, dx 1, rhs = divergence of a seeded velocity field",
                       "grid": [size, dim_y], "iters": iters,
                       "parallelism": "1 GPU" if world == 1 else f"row-slab x{world}, RCCL halo exchange" +
                                      (f"; ALL {world} RANK PROCESSES ON DEVICE {local_rank} (--share-device: RCCL's socket transport "
                                       "over loopback between them) -- a correctness run of the multi-process path, its timings "
                                       "say nothing about xGMI or scaling" if shared_device is not None and world > 1 else ""),
                       **({"physical_gpus": 1, "ranks_share_device": local_rank} if shared_device is not None and world > 1 else {}),
                       "sor_launches_per_solve": info["launches"],
                       "halo_exchanges_per_solve": info["exchanges"],
                       "exchange_schedule": SCHEDULES.get(schedule, schedule),
                       "halo_rows_per_superstep": info["halo"], "measured_exchange_latency_us": exchange_us,
                       "half_sweeps_fused_per_launch": info["fuse"]},
            # which arithmetic `value` was measured with, and what the other one would give
            "numerics": {
                "sor_fold": 1 if args.sor_fold else 0,
                "interior_relaxation": ("(1 - omega) * p + (-0.25f * omega) * t, ONE product where poisson.cpp:109-111 has two: "
                                        "SFL_OPT_SOR_FOLD = 1, opt-in; the reference's bits only where no operand of t is a nonzero "
                                        "number below 2^-124 (dense fields: yes; the front of a sparsely forced field: no)")
                if args.sor_fold else
                ("(1 - omega) * p + omega * (-0.25f * t) with every product rounded on its own, as poisson.cpp:107-111 "
                 "writes it: the library's default, the reference's bits on every input (dense, sparse, denormal)"),
                "contract": "fp32 fields bit for bit (north_star allows 1e-5 relative), UQ32 / index / interpolation bit-exact; "
                            "tests/test_gpu_parity.py test_quiescent_* hold the sparse-forcing case at 8192^2 x 80",
                **({("value_default_arithmetic" if args.sor_fold else "value_with_fold"): other_arith["value"],
                    ("ms_per_solve_default_arithmetic" if args.sor_fold else "ms_per_solve_with_fold"):
                    other_arith["ms_per_solve_hip_events"],
                    "fold_speedup": (other_arith["value"] / value) if not args.sor_fold else (value / other_arith["value"])}
                   if other_arith else {}),
            },
            "parity": parity,
            **({"pressure_checksums": p_sums} if p_sums is not None else {}),
            **({"sim_step_parity": step_parity} if step_parity is not None else {}),
            "roofline": roofline,
            "sim_steps_per_sec": sim_sps,
            "sim_step_us": (1e6 / sim_sps) if sim_sps else None,
            "sim_steps_api": "sfl_step_n(n): one call for the timed steps",
            "sim_steps_per_sec_as_separate_calls": sim_sps_calls,
            # the kernels of the step outside the solve: bytes from the committed rocprofv3 PMC passes / their own steady-state
            # duration there, quoted only while the entries' source hash matches the kernels this run uses
            "sim_step_kernels": step_kernel_records((size, dim_y)) or
                                ("no PMC entry matches the current kernel sources (profiles/pmc_traffic.json step_entries, "
                                 "sha16 " + step_kernel_source_hash() + "): re-run profiles/run_step_pmc.sh"),
            "sim_step_kernels_note": "inside sfl_step advect_velocity + calculate_divergence run as one kernel and "
                                     "subtract_gradient + advect_color as one, inside sfl_step_n the seam kernel joins the "
                                     "second of one step to the first of the next; the operators below are timed one by one",
            "sim_step_per_operator": op_us,
            **({"sim_steps_note": sim_note} if sim_note else {}),
            "device": name,
        }
        if cpu_rec:
            out["cpu_baseline"] = cpu_rec
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
            out["cpu_baseline"]["host_cpu"] = host_cpu_model()
            # the other half of SURVEY 8(d): the reference's per-operator times for one sim step,
            # in full at the two small BASELINE configs (C1 as 61 x 81, C2)
            small = [cpu_operator_times(61, 81, 20, sfl), cpu_operator_times(2048, 2048, 40, sfl)]
            out["cpu_baseline"]["sim_step_per_operator"] = small
            # parity gate on the small BASELINE configs: the whole sim step, all four fields
            out["parity"]["small_configs"] = [{"grid": r["grid"], "iters": r["iters"], "what": "one whole sim step, "
                                               "velocity / divergence / pressure / dye vs the reference CPU path",
                                               "bit_exact": r["gpu_step_bit_exact"]} for r in small]
            if not all(r["gpu_step_bit_exact"] for r in small):
                out["parity"]["bit_exact"] = False
            # ... and on the input class dense random fields cannot represent: sparse forcing of a quiescent field
            out["parity"]["sparse_forcing"] = sparse_forcing_check(sfl, local_rank, args.sor_fold)
            if not args.sor_fold and not out["parity"]["sparse_forcing"]["bit_exact"]:
                out["parity"]["bit_exact"] = False
        print(json.dumps(out), flush=True)
        if parity and not parity["bit_exact"]:
            print(f"bench.py: PARITY FAILURE: {parity['mismatching_cells']} cells of the timed solve differ from the "
                  f"reference (small configurations: {parity.get('small_configs')}; sparse forcing: {parity.get('sparse_forcing')})",
                  file=sys.stderr)
            rc = 3
        if step_parity and step_parity.get("bit_exact") is False:
            print(f"bench.py: PARITY FAILURE of the sim step's fields on ranks {step_parity['ranks_differing']}", file=sys.stderr)
            rc = 3

    rdzv.barrier()
    s.close()
    rdzv.close()
    return rc


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("SFL_BENCH_WORKER") or (args.gpus <= 1 and world <= 1):
        sys.exit(run_rank(args))            # one rank: a worker of one of the launchers below, or the one-GPU run itself
    if world > 1:
        sys.exit(supervise_rank(args))      # started per rank by torch.distributed.run: supervise one worker per attempt
    sys.exit(launch_with_fallback(args))    # `python bench.py --gpus N` as typed: start and supervise N workers per attempt


if __name__ == "__main__":
    main()
