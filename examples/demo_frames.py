#!/usr/bin/env python3
"""Run the GPU fluid simulation and write the dye as PPM frames.

    python examples/demo_frames.py --size 512 384 --steps 240 --every 8 --out /tmp/frames

Mirrors what the sketch does end to end: an initial dye pattern of three colour sectors (the idea
of setup(), ino:196-241 -- generated here on the host with numpy, it is not part of the hot path),
a circular "finger drag" injected as point forces every step (ino:264-269), sfl_step per frame
(ino:252-287), and the draw task's up-scale + RGB565 pack (ino:116-176) through
sfl_render_rgb565.  Needs a GPU: there is no CPU fallback.
"""
import argparse
import importlib
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def three_sectors(dim_x, dim_y):
    """Three 120-degree dye sectors around the centre, channels in UQ32 raw units."""
    j, i = np.mgrid[0:dim_y, 0:dim_x]
    ang = np.arctan2(-(i - dim_x // 2), j - dim_y // 2)
    full = np.uint32(0xF0000000)
    c = np.zeros((dim_y, dim_x, 3), np.uint32)
    c[..., 0][ang < -math.pi / 3] = full
    c[..., 1][(ang >= -math.pi / 3) & (ang < math.pi / 3)] = full
    c[..., 2][ang >= math.pi / 3] = full
    return c


def rgb565_to_rgb888(img):
    r = ((img >> 11) & 0x1F).astype(np.uint8) << 3
    g = ((img >> 5) & 0x3F).astype(np.uint8) << 2
    b = (img & 0x1F).astype(np.uint8) << 3
    return np.stack([r, g, b], axis=-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=2, default=[321, 241], metavar=("DIM_X", "DIM_Y"))
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--scaling", type=int, default=2)
    ap.add_argument("--out", default="frames")
    ap.add_argument("--sketch-init", action="store_true",
                    help="initial condition from sfl_setup_sketch_fields (ino:196-241) instead of numpy")
    args = ap.parse_args()
    sfl = importlib.import_module("esp32-fluid-simulation_amd")
    dim_x, dim_y = args.size
    os.makedirs(args.out, exist_ok=True)
    with sfl.Solver(dim_x, dim_y) as s:
        if args.sketch_init:   # the sketch's own setup(): ino:196-241, on the GPU
            s.setup_sketch_fields()
        else:
            s.upload(sfl.capi.FIELD_VELOCITY, np.zeros((dim_y, dim_x, 2), np.float32))
            s.upload(sfl.capi.FIELD_COLOR, three_sectors(dim_x, dim_y))
        radius, speed = 0.3 * min(dim_x, dim_y), 0.15 * min(dim_x, dim_y) * 30
        for step in range(args.steps):
            a = 2 * math.pi * step / 90
            ci, cj = dim_x / 2 + radius * math.cos(a), dim_y / 2 + radius * math.sin(a)
            cells = [(int(ci) + di, int(cj) + dj) for di in (-1, 0, 1) for dj in (-1, 0, 1)]
            vel = [(-speed * math.sin(a), speed * math.cos(a))] * len(cells)
            s.queue_forces(cells, vel)
            s.step(np.float32(1 / 30.0), 1.0, args.iters, np.float32(1.96))
            if step % args.every == 0:
                img = rgb565_to_rgb888(s.render_rgb565(args.scaling, byteswap=False))
                path = os.path.join(args.out, f"frame_{step:05d}.ppm")
                with open(path, "wb") as f:
                    f.write(b"P6 %d %d 255\n" % (img.shape[1], img.shape[0]))
                    f.write(img.tobytes())
        s.synchronize()
    print(f"wrote {len(os.listdir(args.out))} frames to {args.out}")


if __name__ == "__main__":
    main()
