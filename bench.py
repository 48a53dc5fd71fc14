#!/usr/bin/env python3
"""bench.py -- headline benchmark of the stable-fluids hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--size 8192] [--iters 80]

A "step" is ONE poisson_solve (zero fill + `iters` red-black SOR iterations, poisson.cpp:114-125)
on a `size` x `size` fp32 grid held in HBM -- BASELINE.json's headline metric "cell-iters/sec (SOR
sweep)", config[2] "8192x8192 fp32, 80 SOR iters/step" at N = 1 and config[3] (same grid, row-slab
split with RCCL halo exchange) at N > 1, i.e. STRONG scaling.  After the timed region the full
sim step (advect, divergence, solve, gradient, dye advect: ino:252-287) is timed separately and
reported as `sim_steps_per_sec` (the metric's "+ steps/sec" half).  Before the W warm-up steps the
device is kept busy with ~80 ms of the same solves (untimed, `priming_solves` in the output,
`--no-priming` to skip) so that it runs at its sustained clocks, as in a running simulation.

Launch: N = 1 directly; N > 1 through `python -m torch.distributed.run --nproc-per-node N ...`,
one process per GPU.  torch.distributed (gloo) carries only the bootstrap (RCCL unique id,
barriers, max-over-ranks); every halo byte moves through RCCL send/recv issued by the C++
library on the solver's own HIP stream.

Output: ONE JSON line on rank 0 (contract in the task statement), including
  roofline      algorithmic bytes (16 B per cell-iteration) / avg kernel duration
                measured with HIP events on the solver's stream, against the 8 TB/s HBM peak
  cpu_baseline  the reference's own CPU loop (oracle/_ref, or the oracle port) on this host,
                1 thread, bounded sample; N = 1 only
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
SOR_BYTES_PER_CELL_ITER = 16
SOR_FLOPS_PER_CELL_ITER = 8    # one relaxation of an interior cell: 3 neighbour adds, rhs subtract,
                               # scale by -1/4, two omega products and their sum (never fused);
                               # SURVEY 8a13 counts 10: dx*d is exact at dx = 1, 1-omega is hoisted
VALU_CLOCK_GHZ = 2.4           # MI355X max engine clock (MI355X_MICROARCH.md)


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


def synthetic_color(dim_x, row_begin, row_end, seed=777):
    j = np.arange(row_begin, row_end, dtype=np.uint64)[:, None, None]
    i = np.arange(dim_x, dtype=np.uint64)[None, :, None]
    k = np.arange(3, dtype=np.uint64)[None, None, :]
    h = ((j * np.uint64(dim_x) + i) * np.uint64(3) + k + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    h ^= h >> np.uint64(32)
    return ((h >> np.uint64(20)) & np.uint64(0x7FFFFFFF)).astype(np.uint32)   # raw < 2^31


def cpu_baseline(size, iters, budget_s=12.0):
    """Reference CPU loop (1 thread) on a bounded sample of the same workload: whole
    poisson_solve calls on the same grid, repeated until ~budget_s of CPU time is spent."""
    from oracle import loader  # checker / baseline only
    path = loader.reference() if loader.reference_available() else loader.port()
    d = np.random.default_rng(5).standard_normal((size, size)).astype(np.float32) * np.float32(0.1)
    t0 = time.perf_counter()
    path.poisson_solve(d, 1.0, 2, np.float32(1.96))
    per_iter = (time.perf_counter() - t0) / 2
    run_iters = int(max(2, min(iters, budget_s / max(per_iter, 1e-9))))
    reps = int(max(1, round(budget_s / max(per_iter * run_iters, 1e-9))))
    t0 = time.perf_counter()
    for _ in range(reps):
        path.poisson_solve(d, 1.0, run_iters, np.float32(1.96))
    dt = time.perf_counter() - t0
    return {"value": size * size * run_iters * reps / dt, "unit": "cell-iters/s", "cores": 1,
            "kind": path.kind,
            "sample": f"{reps} x poisson_solve {size}x{size} fp32, {run_iters} iters each, 1 thread, "
                      f"{dt:.1f} s ({'unmodified reference sources' if path.kind == 'reference' else 'oracle C port'}, "
                      f"g++/gcc -O2 -ffp-contract=off)"}


def cpu_operator_times(dim_x, dim_y, iters):
    """Per-operator wall time of the reference CPU path for ONE sim step (1 thread), ms."""
    from oracle import loader
    path = loader.reference() if loader.reference_available() else loader.port()
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
    timed("advect_color", lambda: path.advect_vec3uq32(c, vp, dt, False))
    out["step"] = sum(out.values())
    return {"grid": [dim_x, dim_y], "iters": iters, "kind": path.kind, "ms": out}


def gpu_operator_times(s, capi, iters, reps=3):
    """Per-operator HIP-event time of one sim step on this rank's slab, microseconds."""
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
        out[name] = best * 1e3
    return out


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pmc_traffic(size, fuse, lane_cells, world):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json), or None when no entry matches this exact configuration."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["entries"]
    except Exception:
        return None
    for e in table:
        if (e["grid"] == [size, size] and e["fuse"] == fuse and e["lane_cells"] == lane_cells
                and e["n_gpus"] == world):
            return e
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=80)
    ap.add_argument("--dim-y", type=int, default=0, help="rows (default: size); experiments only")
    ap.add_argument("--fuse", type=int, default=0, help="SOR half-sweeps fused per launch (0 = library default)")
    ap.add_argument("--sor-kernel", type=int, default=0)
    ap.add_argument("--sor-rows", type=int, default=0)
    ap.add_argument("--lane-cells", type=int, default=0, help="cells per lane of the fused kernel (0 auto, 2, 4)")
    ap.add_argument("--sim-steps", type=int, default=3, help="full sim steps timed after the main region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-priming", action="store_true",
                    help="skip the ~80 ms of untimed solves that bring the GPU to its sustained clocks")
    ap.add_argument("--no-fuse-projection", action="store_true",
                    help="sim step: separate subtract_gradient and dye-advection kernels (A/B)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if "SFL_BENCH_DEVICE" in os.environ:   # bring-up aid: several ranks on one device
        local_rank = int(os.environ["SFL_BENCH_DEVICE"])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs the torch.distributed.run launcher (WORLD_SIZE={world})")
        args.gpus = world

    # load the product library (system ROCm runtime) before torch pulls in its own copy
    sfl = importlib.import_module("esp32-fluid-simulation_amd")
    capi = sfl.capi
    if sfl.device_count() < 1:
        sys.exit("bench.py needs a GPU: the product path has no CPU fallback")

    import torch
    import torch.distributed as dist
    if torch.cuda.is_available() and local_rank < torch.cuda.device_count():
        torch.cuda.set_device(local_rank)   # torch.cuda.synchronize() below must mean THIS rank's GPU
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    size, iters = args.size, args.iters
    dim_y = args.dim_y or size
    s = sfl.Solver(size, dim_y, device=local_rank, rank=rank, nranks=world)
    if args.fuse:
        s.set_option(capi.OPT_SOR_FUSE, args.fuse)
    if args.sor_kernel:
        s.set_option(capi.OPT_SOR_KERNEL, args.sor_kernel)
    if args.sor_rows:
        s.set_option(capi.OPT_SOR_ROWS, args.sor_rows)
    if args.lane_cells:
        s.set_option(capi.OPT_SOR_LANE_CELLS, args.lane_cells)
    if args.no_fuse_projection:
        s.set_option(capi.OPT_FUSE_PROJECTION, 0)
    if world > 1:
        uid = [sfl.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        # RCCL prints a version banner on stdout while the communicator comes up; keep stdout
        # clean for the ONE JSON line by pointing fd 1 at stderr for the duration of the call
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            s.comm_attach(uid[0])
        finally:
            os.dup2(saved, 1)
            os.close(saved)

    # synthetic inputs, resident in HBM before anything is timed
    s.upload(capi.FIELD_VELOCITY, synthetic_velocity(size, s.row_begin, s.row_end))
    s.upload(capi.FIELD_COLOR, synthetic_color(size, s.row_begin, s.row_end))
    s.calculate_divergence(1.0)     # right-hand side = divergence of the velocity
    s.synchronize()

    def sync_all():
        s.synchronize()
        torch.cuda.synchronize() if torch.cuda.is_available() else None

    omega = np.float32(1.96)

    # Clock priming (untimed, reported as `priming_solves`): after set-up (host-side data
    # generation, PCIe uploads) the GPU sits at idle clocks and needs ~30 ms of load to reach its
    # sustained rate (tools/solve_sequence_probe.py: 2.7, 2.5, 2.4 ... 1.93 ms per solve over the
    # first 15 solves).  A running simulation lives at the sustained rate, so the bench first keeps
    # the device busy with the same solves for ~80 ms; the W warm-up steps and the K timed steps
    # follow without a gap.  The count is agreed across ranks (every rank must issue the same
    # exchanges).
    priming = 0
    if not args.no_priming:
        s.poisson_solve(1.0, iters, omega)   # lazy allocations, code objects
        sync_all()
        t_one = time.perf_counter()
        s.poisson_solve(1.0, iters, omega)
        sync_all()
        t_one = time.perf_counter() - t_one
        if world > 1:
            tt = torch.tensor([t_one], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_one = float(tt[0])
        # launch-bound tiny grids do not load the GPU at all: nothing to ramp (and hundreds of queued
        # launches only disturb the host-side launch path that bounds them)
        priming = int(min(400, max(4, 0.08 / t_one))) if t_one >= 0.5e-3 else 0
        for _ in range(priming):
            s.poisson_solve(1.0, iters, omega)
        priming += 2
    for _ in range(args.warmup):
        s.poisson_solve(1.0, iters, omega)
    sync_all()
    barrier()
    t0 = time.perf_counter()
    s.timer_start()
    for _ in range(args.steps):
        s.poisson_solve(1.0, iters, omega)
    ev_ms = s.timer_stop()
    sync_all()
    barrier()
    elapsed = time.perf_counter() - t0
    info = s.last_solve_info()

    if world > 1:
        t = torch.tensor([elapsed, ev_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, ev_ms = float(t[0]), float(t[1])

    # full sim step, timed separately (not part of `value`).  A slab reports a back-trace that
    # left its advection halo at synchronize(); every rank still issues the same launches and
    # exchanges, so the failure is recorded, agreed on collectively and never deadlocks a barrier.
    sim_sps, sim_note = None, None
    if args.sim_steps > 0:
        failed = []

        def sync_soft():
            try:
                s.synchronize()
            except sfl.SflError as e:
                failed.append(str(e))

        if world > 1:   # generous advection halo: the projected velocity is not bounded by vamp
            s.set_option(capi.OPT_ADVECT_HALO, 32)
        dtf = np.float32(1 / 30.0)
        s.step(dtf, 1.0, iters, omega)
        sync_soft()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.sim_steps):
            s.step(dtf, 1.0, iters, omega)
        sync_soft()
        barrier()
        sim_t = time.perf_counter() - t1
        bad = 1.0 if failed else 0.0
        if world > 1:
            t = torch.tensor([sim_t, bad], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sim_t, bad = float(t[0]), float(t[1])
        if bad:
            sim_note = failed[0] if failed else "a peer rank reported an advection-halo overflow"
        else:
            sim_sps = args.sim_steps / sim_t

    op_us = None
    if world == 1 and args.sim_steps > 0:
        op_us = gpu_operator_times(s, capi, iters)

    if rank == 0:
        cells = size * dim_y
        value = cells * iters * args.steps / elapsed
        launches = max(info["launches"], 1)
        # dominant kernel: one launch relaxes every owned cell `fuse`/2 times
        avg_launch_s = (ev_ms / 1e3) / (args.steps * launches)
        bytes_per_launch = SOR_BYTES_PER_CELL_ITER * (cells / world) * iters / launches
        achieved = bytes_per_launch / avg_launch_s / 1e9
        name, cus, mem = sfl.device_info(local_rank)
        lane_cells = s.get_option(capi.OPT_SOR_LANE_CELLS) or 2
        pmc = pmc_traffic(size, info["fuse"], lane_cells, world) if dim_y == size else None
        out = {
            "metric": "cell-iters/sec (SOR sweep)", "value": value, "unit": "cell-iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "priming_solves": priming,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"poisson_solve {size}x{dim_y} fp32, {iters} red-black SOR iters/step, "
                                   f"omega 1.96, dx 1, rhs = divergence of a seeded velocity field",
                       "grid": [size, dim_y], "iters": iters,
                       "parallelism": "1 GPU" if world == 1 else f"row-slab x{world}, RCCL halo exchange",
                       "sor_launches_per_solve": info["launches"],
                       "halo_exchanges_per_solve": info["exchanges"],
                       "half_sweeps_fused_per_launch": info["fuse"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc["traffic_bytes_per_launch"] if pmc else None,
                         "traffic_source": pmc["source"] if pmc else None,
                         "kernel": "sor_fused_kernel" if info["fuse"] > 1 else "sor_half_sweep_kernel",
                         "avg_launch_us": avg_launch_s * 1e6,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         # what actually bounds the temporally blocked kernel: the
                         # reference's 8 unfused fp32 operations per relaxation (poisson.cpp:63-112;
                         # contraction to FMA would change results) against one plain fp32 VALU
                         # operation per lane per clock, 64 lanes x CUs x 2.4 GHz
                         "valu": {"flops_per_cell_iter": SOR_FLOPS_PER_CELL_ITER,
                                  "achieved": value / world * SOR_FLOPS_PER_CELL_ITER / 1e12,
                                  "peak": cus * 64 * VALU_CLOCK_GHZ / 1e3, "unit": "TFLOP/s",
                                  "frac": value / world * SOR_FLOPS_PER_CELL_ITER / 1e12
                                          / (cus * 64 * VALU_CLOCK_GHZ / 1e3)}},
            "sim_steps_per_sec": sim_sps,
            "sim_step_per_operator_us": op_us,
            **({"sim_steps_note": sim_note} if sim_note else {}),
            "device": name,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(size, iters)  # always the square headline grid
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
            out["cpu_baseline"]["host_cpu"] = host_cpu_model()
            # the other half of SURVEY 8(d): the reference's per-operator times for one sim step,
            # in full at the two small BASELINE configs (C1 as 61 x 81, C2)
            out["cpu_baseline"]["sim_step_per_operator"] = [cpu_operator_times(61, 81, 20),
                                                            cpu_operator_times(2048, 2048, 40)]
        print(json.dumps(out), flush=True)

    barrier()
    s.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
