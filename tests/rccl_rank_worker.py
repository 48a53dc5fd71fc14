"""Test infrastructure (not a test module): ONE rank process of a real N-rank RCCL communicator whose ranks all sit on device 0 of a
one-GPU box -- every rank carries its own NCCL_HOSTID, so RCCL takes them for ranks on different hosts and moves the halos between
the processes over its socket transport (tests/test_rccl_processes.py starts N of these).  Each scenario drives the
product's C ABI exactly as an application's rank would and checks THIS rank's rows against the oracle (checker only) bit for bit.

    RANK=r WORLD_SIZE=n SFL_RDZV_KEY=k python tests/rccl_rank_worker.py <scenario> <seed> [seconds [big]]

Prints one JSON line: {"rank": r, "ok": bool, ...}.  Scenarios: soak, mismatch, gather, forces, late_peer."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--spawn":
    # python tests/rccl_rank_worker.py --spawn N <scenario> <seed> [seconds]: start the N ranks (this process touches no GPU),
    # print their result lines, exit with the worst status -- for soaks by hand (tools/recipes/soak.sh, leg "ranks")
    import subprocess
    import tempfile
    n = int(sys.argv[2])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NCCL_HOSTID")}
    env.update({"WORLD_SIZE": str(n), "SFL_RDZV_KEY": f"spawn_{os.getpid()}", "SFL_RDZV_DIR": tempfile.mkdtemp(prefix="sfl_rccl_spawn_")})
    kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[3:], env=dict(env, RANK=str(r)),
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for r in range(n)]
    worst = 0
    for k in kids:
        out, _ = k.communicate()
        print("\n".join(l for l in out.splitlines() if l.startswith("{")), flush=True)
        worst = worst or k.returncode
    sys.exit(worst)

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("NCCL_HOSTID", f"sfl-test-rank-{rank}")     # (see the docstring; bench.py --share-device does the same)
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_IB_DISABLE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

sfl = importlib.import_module("esp32-fluid-simulation_amd")
from oracle import loader  # noqa: E402  (this file lives under tests/: the oracle is the checker)

capi, orc = sfl.capi, loader.port()
rdzv = importlib.import_module("esp32-fluid-simulation_amd.rendezvous").Rendezvous(rank, world)
scenario, seed = sys.argv[1], int(sys.argv[2])
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
big = len(sys.argv) > 4 and sys.argv[4] == "big"      # soak: slabs of 200 .. 700 rows x 700 .. 4096 columns per rank
SCHEDULES = {"by-event": {capi.OPT_EXCHANGE_SCHEDULE: 2}, "in-time": {capi.OPT_EXCHANGE_SCHEDULE: 3, capi.OPT_HALO_TIMEOUT_MS: 15000},
             "in-line": {capi.OPT_EXCHANGE_SCHEDULE: 1}}


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def attach(s):
    uid = rdzv.broadcast_bytes(sfl.comm_unique_id() if rank == 0 else None)
    with sfl.stdout_to_stderr():   # (RCCL's banner)
        s.comm_attach(uid)


def slab(s, a):
    return np.ascontiguousarray(a[s.row_begin:s.row_end])


def same(s, field, want):
    return bool(np.array_equal(bits(s.download(field)), bits(slab(s, want))))


def everyone(ok):
    """all ranks' verdicts (a rank must not leave while its peers still expect messages from it)"""
    return rdzv.all_gather(bool(ok))


def scenario_soak():
    """Random slab groups -- shape, iterations, fuse depth, halo depth, schedule, dx, omega, dt, velocity scale drawn from the shared
    seed -- a solve and two or three sim steps each, this rank's rows of all four fields against the oracle."""
    rng = np.random.default_rng(seed)           # the SAME stream on every rank: the ranks agree on every draw
    t0, cases, bad, notes = time.time(), 0, 0, []
    while True:
        go = rdzv.broadcast_bytes((b"1" if time.time() - t0 < budget else b"0") if rank == 0 else None)
        if go != b"1":
            break
        if big:     # many tiles per launch, several launches and exchanges per solve, launches with tiles that wait
            dim_x = int(rng.choice([int(rng.integers(700, 3000)), 1024, 2048, 4096]))
            dim_y = int(rng.integers(world * 200, world * 700))
            iters = int(rng.integers(10, 60))
        else:
            dim_x = int(rng.choice([int(rng.integers(16, 700)), 128, 256, 1024]))
            dim_y = int(rng.integers(world * 34, world * 260))
            iters = int(rng.integers(1, 30))
        dx = float(rng.choice([1.0, 1.0, 0.5]))
        omega = np.float32(rng.choice([1.96, 1.3]))
        dt = np.float32(rng.choice([1 / 30.0, 0.1]))
        vamp = float(rng.choice([0.0, 20.0, 300.0]))
        sched = str(rng.choice(list(SCHEDULES)))
        thinnest = dim_y // world
        fuse = int(rng.choice([0, 0, 4, 8, 10]))
        lowest = fuse if fuse else 16          # (a halo is at least one launch deep)
        halo = int(rng.choice([0, 0, int(rng.integers(lowest, max(lowest + 1, min(thinnest, 160))))]))
        if halo > min(thinnest, 160):
            halo = 0
        opts = {capi.OPT_SOR_FUSE: fuse, capi.OPT_SOR_HALO: halo,
                capi.OPT_ADVECT_KERNEL: int(rng.choice([0, 1, 2])), capi.OPT_FUSE_PROJECTION: int(rng.choice([1, 0]))}
        opts.update(SCHEDULES[sched])
        n_steps = int(rng.choice([2, 3]))
        v = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * vamp).astype(np.float32)
        c = rng.integers(0, 2 ** 31, (dim_y, dim_x, 3), dtype=np.uint32)
        d = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
        texture = str(rng.choice(["dense", "dense", "sparse", "tiny"]))      # (round 6: the textures of tests/fuzz_vs_oracle.py)
        if texture == "sparse":      # a quiescent field with sparse forcing, enough iterations for the front to reach the denormals
            keep = rng.random((dim_y, dim_x)) < 3.0 / (dim_x * dim_y)
            keep[rng.integers(0, dim_y), rng.integers(0, dim_x)] = True
            d = np.where(keep, d * np.float32(rng.choice([1.0, 40.0, 1e-20])), np.float32(0.0)).astype(np.float32)
            v = np.where(keep[..., None], v, np.float32(0.0)).astype(np.float32)
            if not big:
                iters = int(rng.integers(60, 110))
        elif texture == "tiny":
            scale = np.float32(2.0 ** -int(rng.integers(100, 146)))
            d, v = (d * scale).astype(np.float32), (v * scale).astype(np.float32)
        tag = f"{world} ranks {dim_x}x{dim_y} {texture} iters {iters} dx {dx} omega {omega} dt {dt} vamp {vamp} {sched} options {opts}"
        ok = True
        try:
            with sfl.Solver(dim_x, dim_y, device=0, rank=rank, nranks=world) as s:
                for k, val in opts.items():
                    s.set_option(k, val)
                attach(s)
                s.upload(capi.FIELD_DIVERGENCE, slab(s, d))
                s.poisson_solve(dx, iters, omega)
                s.synchronize()
                ok = same(s, capi.FIELD_PRESSURE, orc.poisson_solve(d, dx, iters, omega))
                s.upload(capi.FIELD_VELOCITY, slab(s, v))
                s.upload(capi.FIELD_COLOR, slab(s, c))
                for _ in range(n_steps):
                    s.step(dt, dx, iters, omega)
                s.synchronize()
                want = (v, None, None, c)
                for _ in range(n_steps):
                    want = orc.step(want[0], want[3], dt, dx, iters, omega)
                for f, w in zip((capi.FIELD_VELOCITY, capi.FIELD_DIVERGENCE, capi.FIELD_PRESSURE, capi.FIELD_COLOR), want):
                    ok = ok and same(s, f, w)
                verdicts = everyone(ok)      # (inside the context: nobody closes its communicator while a peer still computes)
        except sfl.SflError as e:
            ok, verdicts = False, [False]
            notes.append(f"ERROR {e} | {tag}")
            print(json.dumps({"rank": rank, "ok": False, "cases": cases, "error": str(e), "config": tag}), flush=True)
            return 1
        cases += 1
        if not all(verdicts):
            bad += 1
            if not ok:
                notes.append(f"MISMATCH on rank {rank} | {tag}")
    print(json.dumps({"rank": rank, "ok": bad == 0, "cases": cases, "bad": bad, "notes": notes[:5]}), flush=True)
    return 0 if bad == 0 else 1


def scenario_mismatch():
    """A rank created with another option must be refused at attach -- by every rank, with the word that differs."""
    with sfl.Solver(256, 64 * world, device=0, rank=rank, nranks=world) as s:
        s.set_option(capi.OPT_SOR_FUSE, 8 if rank == world - 1 else 4)
        try:
            attach(s)
            refused, why = False, ""
        except sfl.SflError as e:
            refused, why = True, str(e)
        verdicts = everyone(refused and "disagree" in why)
    print(json.dumps({"rank": rank, "ok": all(verdicts), "refused": refused, "why": why}), flush=True)
    return 0 if all(verdicts) else 1


def scenario_gather():
    """A velocity too fast for any halo: the advections fall back to gathering the whole field over the ranks (ncclSend / ncclRecv
    between all pairs); and a slow one on the automatic halo, where the dye's guessed halo is checked a call late."""
    dim_x, dim_y, iters = 200, 96 * world, 6
    rng = np.random.default_rng(seed)
    ok = True
    for vamp, dt in ((30000.0, np.float32(0.1)), (40.0, np.float32(1 / 30.0))):
        v = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * vamp).astype(np.float32)
        c = rng.integers(0, 2 ** 31, (dim_y, dim_x, 3), dtype=np.uint32)
        with sfl.Solver(dim_x, dim_y, device=0, rank=rank, nranks=world) as s:
            attach(s)
            s.upload(capi.FIELD_VELOCITY, slab(s, v))
            s.upload(capi.FIELD_COLOR, slab(s, c))
            for _ in range(2):
                s.step(dt, 1.0, iters, np.float32(1.96))
            s.synchronize()
            want = (v, None, None, c)
            for _ in range(2):
                want = orc.step(want[0], want[3], dt, 1.0, iters, np.float32(1.96))
            for f, w in zip((capi.FIELD_VELOCITY, capi.FIELD_DIVERGENCE, capi.FIELD_PRESSURE, capi.FIELD_COLOR), want):
                ok = ok and same(s, f, w)
            verdicts = everyone(ok)
        ok = all(verdicts)
    print(json.dumps({"rank": rank, "ok": ok}), flush=True)
    return 0 if ok else 1


def scenario_forces():
    """Touch forces (ino:264-269) queued on every rank -- cells in every slab, on both sides of a cut, one cell twice -- go in between
    the velocity advection and the divergence of the rank that owns the cell."""
    dim_x, dim_y, iters = 150, 70 * world, 8
    rng = np.random.default_rng(seed)
    v = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * 40.0).astype(np.float32)
    c = rng.integers(0, 2 ** 31, (dim_y, dim_x, 3), dtype=np.uint32)
    cuts = [dim_y * r // world for r in range(1, world)]
    cells = [[10, 5], [dim_x - 2, dim_y - 3], [10, 5]] + [[30 + k, j] for cut in cuts for k, j in enumerate((cut - 1, cut, cut + 1))]
    cells = np.array(cells, np.int32)
    vel = (rng.uniform(-9, 9, (len(cells), 2))).astype(np.float32)
    dt, omega = np.float32(1 / 30.0), np.float32(1.96)
    with sfl.Solver(dim_x, dim_y, device=0, rank=rank, nranks=world) as s:
        attach(s)
        s.upload(capi.FIELD_VELOCITY, slab(s, v))
        s.upload(capi.FIELD_COLOR, slab(s, c))
        s.queue_forces(cells, vel)
        s.step(dt, 1.0, iters, omega)
        s.synchronize()
        va = orc.advect_vec2f(v, v, dt, True)
        for (i, j), u in zip(cells, vel):
            va[j, i] = u
        d = orc.divergence(va, 1.0)
        p = orc.poisson_solve(d, 1.0, iters, omega)
        want_v = orc.subtract_gradient(va, p, 1.0)
        ok = same(s, capi.FIELD_VELOCITY, want_v) and same(s, capi.FIELD_PRESSURE, p) and \
            same(s, capi.FIELD_COLOR, orc.advect_vec3uq32(c, want_v, dt, False))
        verdicts = everyone(ok)
    print(json.dumps({"rank": rank, "ok": all(verdicts), "mine": ok}), flush=True)
    return 0 if all(verdicts) else 1


def scenario_late_peer():
    """Exchanges in time: the last rank comes to every solve 0.4 s late.  Nobody may time out (the wait is on the device, the limit
    15 s), nobody may read a halo before it is there: the results stay the reference's."""
    dim_x, dim_y, iters = 512, 256 * world, 24
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
    want = orc.poisson_solve(d, 1.0, iters, np.float32(1.96))
    ok = True
    with sfl.Solver(dim_x, dim_y, device=0, rank=rank, nranks=world) as s:
        for k, val in SCHEDULES["in-time"].items():
            s.set_option(k, val)
        s.set_option(capi.OPT_SOR_HALO, 16)      # (several exchanges per solve)
        attach(s)
        s.upload(capi.FIELD_DIVERGENCE, slab(s, d))
        for _ in range(4):
            if rank == world - 1:
                time.sleep(0.4)
            s.poisson_solve(1.0, iters, np.float32(1.96))
            s.synchronize()                      # (raises if a wait inside the solve gave up)
            ok = ok and same(s, capi.FIELD_PRESSURE, want)
        info = s.last_solve_info()
        schedule = s.get_option(capi.OPT_EXCHANGE_SCHEDULE)    # 3 = in time, counted on the device
        verdicts = everyone(ok and schedule == 3)
    print(json.dumps({"rank": rank, "ok": all(verdicts), "exchanges": info["exchanges"], "schedule": schedule}), flush=True)
    return 0 if all(verdicts) else 1


def scenario_quiescent():
    """The sketch's own scenario (ino:199, 249-289) between REAL RCCL rank processes: velocity zero, a handful of drag messages queued
    on every rank (sfl_queue_drags), two whole sim steps at 80 SOR iterations -- the pressure front decays through the denormals, the
    projected velocity carries them into the second step -- in every exchange schedule; velocity, divergence, pressure and dye of this
    rank's rows bit for bit against the oracle."""
    dim_x, dim_y, iters, steps = 640, 288 * world, 80, 2
    rng = np.random.default_rng(seed)
    v = np.zeros((dim_y, dim_x, 2), np.float32)
    c = rng.integers(0, 2 ** 31, (dim_y, dim_x, 3), dtype=np.uint32)
    cut = dim_y // world
    drags = [(cut - 1, 300, 35.0, -20.0), (cut, 301, 30.0, -25.0), (dim_y // 2 + 17, 90, -60.0, 12.0), (dim_y - 1, dim_x - 1, 5.0, 5.0)]
    forces = ([(y, x) for x, y, _, _ in drags], [(vy, vx) for _, _, vx, vy in drags])      # cells (i, j), velocities: the swap of ino:264-269
    dt, omega = np.float32(1 / 30.0), np.float32(1.96)
    vo, co = v, c
    for k in range(steps):
        va = orc.advect_vec2f(vo, vo, dt, True)
        if k == 0:
            for (i, j), u in zip(*forces):
                va[j, i] = u
        do = orc.divergence(va, 1.0)
        po = orc.poisson_solve(do, 1.0, iters, omega)
        vo = orc.subtract_gradient(va, po, 1.0)
        co = orc.advect_vec3uq32(co, vo, dt, False)
    front = int(np.count_nonzero((po != 0) & (np.abs(po) < np.float32(2.0 ** -124))))
    ok = front > 50          # (the scenario must reach the denormal range)
    for sched in ("by-event", "in-time", "in-line"):
        with sfl.Solver(dim_x, dim_y, device=0, rank=rank, nranks=world) as s:
            for k, val in SCHEDULES[sched].items():
                s.set_option(k, val)
            attach(s)
            s.upload(capi.FIELD_VELOCITY, slab(s, v))
            s.upload(capi.FIELD_COLOR, slab(s, c))
            s.queue_drags(drags)
            s.step_n(steps, dt, 1.0, iters, omega)
            s.synchronize()
            ok = ok and same(s, capi.FIELD_VELOCITY, vo) and same(s, capi.FIELD_DIVERGENCE, do) and \
                same(s, capi.FIELD_PRESSURE, po) and same(s, capi.FIELD_COLOR, co)
    verdicts = everyone(ok)
    print(json.dumps({"rank": rank, "ok": all(verdicts), "mine": ok, "denormal_front_cells": front}), flush=True)
    return 0 if all(verdicts) else 1


if __name__ == "__main__":
    try:
        code = {"soak": scenario_soak, "mismatch": scenario_mismatch, "gather": scenario_gather, "forces": scenario_forces,
                "late_peer": scenario_late_peer, "quiescent": scenario_quiescent}[scenario]()
    finally:
        rdzv.close()
    sys.exit(code)
