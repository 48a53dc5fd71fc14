#!/usr/bin/env python3
"""Generate the committed golden fixtures from the UNMODIFIED reference.

Run in the authoring container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

Every expected output below is produced by oracle/_ref/libsf_ref.so, i.e. by the reference's
own advect.h / finitediff.cpp / poisson.cpp compiled with g++ -O2 -ffp-contract=off.  The
fixtures are pure data (inputs + expected outputs); no reference source is stored.

Files (np.savez_compressed):
  step_<dimx>x<dimy>_i<iters>.npz   inputs v0, c0 (LCG recipe of SURVEY.md 8c) and, for each of
                                    `nsteps` sim steps, v / div / p / colour after the step
  advc_<dimx>x<dimy>.npz            advect<T, float> for T = float, UQ32, Vector2<UQ32>, Vector3<float>, both no_slip
  ops_<dimx>x<dimy>.npz             per-operator cases on numpy-RNG inputs, incl. both no_slip
                                    values for both advect instantiations, dx != 1, omega != 1.96
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import loader  # noqa: E402

DT = np.float32(1 / 30.0)
OMEGA = np.float32(1.96)

STEP_CASES = [  # dim_x, dim_y, iters, vamp, seed, nsteps
    (61, 81, 10, 100.0, 12345, 3),
    (61, 81, 20, 100.0, 12345, 2),
    (33, 17, 7, 60.0, 7, 2),
    (64, 48, 12, 200.0, 99, 2),
    (2, 2, 5, 3.0, 1, 2),
    (3, 3, 5, 3.0, 2, 2),
    (130, 70, 9, 150.0, 4242, 1),
]

OPS_SHAPES = [(2, 2), (3, 2), (5, 4), (33, 17), (61, 81), (100, 37)]


def main():
    ref = loader.reference()
    port = loader.port()  # only for the LCG input recipe (inputs, not outputs)
    for dim_x, dim_y, iters, vamp, seed, nsteps in STEP_CASES:
        v, c = port.lcg_fields(dim_x, dim_y, seed, vamp)
        out = {"v0": v, "c0": c, "meta": np.array([dim_x, dim_y, iters, nsteps], np.int32),
               "vamp": np.float32(vamp), "seed": np.uint32(seed), "dt": DT, "omega": OMEGA}
        for s in range(1, nsteps + 1):
            v, d, p, c = ref.step(v, c, DT, 1.0, iters, OMEGA)
            out.update({f"v{s}": v, f"div{s}": d, f"p{s}": p, f"c{s}": c})
        path = os.path.join(HERE, f"step_{dim_x}x{dim_y}_i{iters}.npz")
        np.savez_compressed(path, **out)
        print("wrote", os.path.relpath(path), os.path.getsize(path), "bytes")

    for dim_x, dim_y in OPS_SHAPES:
        rng = np.random.default_rng(1000 * dim_x + dim_y)
        v = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * 80).astype(np.float32)
        q = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * 3).astype(np.float32)
        c = rng.integers(0, 2 ** 31, (dim_y, dim_x, 3), dtype=np.uint32)
        s = rng.standard_normal((dim_y, dim_x)).astype(np.float32)
        out = {"v": v, "q": q, "c": c, "s": s, "dt": DT}
        for ns in (0, 1):
            out[f"adv2_self_ns{ns}"] = ref.advect_vec2f(v, v, DT, bool(ns))
            out[f"adv2_other_ns{ns}"] = ref.advect_vec2f(q, v, DT, bool(ns))
            out[f"adv3_ns{ns}"] = ref.advect_vec3uq32(c, v, DT, bool(ns))
        for tag, dx in (("dx1", 1.0), ("dx05", 0.5)):
            out[f"div_{tag}"] = ref.divergence(v, dx)
            out[f"grad_{tag}"] = ref.subtract_gradient(v, s, dx)
        out["pois_i1"] = ref.poisson_solve(s, 1.0, 1, OMEGA)
        out["pois_i8"] = ref.poisson_solve(s, 1.0, 8, OMEGA)
        out["pois_i5_w15_dx05"] = ref.poisson_solve(s, 0.5, 5, np.float32(1.5))
        path = os.path.join(HERE, f"ops_{dim_x}x{dim_y}.npz")
        np.savez_compressed(path, **out)
        print("wrote", os.path.relpath(path), os.path.getsize(path), "bytes")


    # advect<T, float> for the element types other than the sketch's two: float, UQ32, Vector2<UQ32>, Vector3<float>
    # (the reference's own template instantiations, oracle/ref_shim.cpp)
    for dim_x, dim_y in CHANNEL_SHAPES:
        rng = np.random.default_rng(77 * dim_x + dim_y)
        v = (rng.uniform(-1, 1, (dim_y, dim_x, 2)) * 70).astype(np.float32)
        out = {"v": v, "dt": DT}
        for channels, uq in CHANNEL_TYPES:
            shape = (dim_y, dim_x) if channels == 1 else (dim_y, dim_x, channels)
            q = rng.integers(0, 2 ** 31, shape, dtype=np.uint32) if uq else (rng.standard_normal(shape) * 9).astype(np.float32)
            tag = f"c{channels}{'u' if uq else 'f'}"
            out[f"in_{tag}"] = q
            for ns in (0, 1):
                out[f"out_{tag}_ns{ns}"] = ref.advect_channels(q, v, DT, bool(ns))
        path = os.path.join(HERE, f"advc_{dim_x}x{dim_y}.npz")
        np.savez_compressed(path, **out)
        print("wrote", os.path.relpath(path), os.path.getsize(path), "bytes")


CHANNEL_SHAPES = [(2, 2), (5, 4), (33, 17), (61, 81)]
CHANNEL_TYPES = [(1, False), (1, True), (2, True), (3, False)]

if __name__ == "__main__":
    main()
