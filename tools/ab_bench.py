#!/usr/bin/env python3
"""Interleaved A/B of kernel variants in ONE process (same device, same data).

    python tools/ab_bench.py PANO_COLS_PIPE=0 PANO_COLS_PIPE=1 [--workload cfg3] [--rounds 5]

Each positional argument is one variant: comma-separated NAME=VALUE settings of
the library's A/B environment switches.  Prints per-kernel median / min ms per
step for every variant.
"""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()

    import torch
    from pano360_amd import engine, synth
    cfg = dict(synth.CONFIGS[args.workload])
    rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"],
                                     sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
    shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
    eng = engine.Engine()
    frames = eng.upload_frames([synth.make_frame(i, cfg["width"], cfg["height"], "A")
                                for i in range(cfg["n"])])
    lib = eng.lib

    def run(steps):
        for _ in range(steps):
            plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
            eng.stitch(frames, plan, "multiband", cfg["n_levels"])
        torch.cuda.synchronize()

    names = [lib.pano_kernel_name(k).decode() for k in range(lib.pano_kernel_count())]
    results = {v: {n: [] for n in names} for v in args.variants}
    wall = {v: [] for v in args.variants}
    run(2)
    import time
    for _ in range(args.rounds):
        for variant in args.variants:
            keys = []
            for item in variant.split(","):
                k, val = item.split("=")
                os.environ[k] = val
                keys.append(k)
            run(1)
            lib.pano_timing_enable(1)
            t0 = time.perf_counter()
            run(args.steps)
            wall[variant].append((time.perf_counter() - t0) / args.steps * 1e3)
            for kid, n in enumerate(names):
                ms, cnt = C.c_double(), C.c_int()
                lib.pano_timing_read(kid, C.byref(ms), C.byref(cnt))
                if cnt.value:
                    results[variant][n].append(ms.value / args.steps)
            lib.pano_timing_enable(0)
            for k in keys:
                del os.environ[k]
    for variant in args.variants:
        print(f"== {variant}: step median {statistics.median(wall[variant]):.3f} ms "
              f"min {min(wall[variant]):.3f}")
        for n in names:
            vals = results[variant][n]
            if vals:
                print(f"   {n:28s} median {statistics.median(vals):8.4f}  min {min(vals):8.4f} ms/step")


if __name__ == "__main__":
    main()
