#!/usr/bin/env python3
"""The drop-in entry point as a reference user calls it: stitcher.stitch(regions, blender) with
host images in bundle_adj.Image records, host mosaic out.  python tools/probe_dropin.py [cfg3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bundle_adj, stitcher
from pano360_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cfg = synth.CONFIGS[name]
n, w, h = cfg["n"], cfg["width"], cfg["height"]
rots, intrs = synth.make_cameras(n, w, h, sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
imgs = [synth.make_frame(i % 4, w, h, "A") for i in range(n)]
stitcher.MAX_RESOLUTION = 10 ** 9
def regions():
    return [bundle_adj.Image(im, r.copy(), k.copy()) for im, r, k in zip(imgs, rots, intrs)]
for blender in (stitcher.multiband_blend, stitcher.linear_blend):
    for crop in (False, True):
        stitcher.stitch(regions(), blender, crop=crop)
        regs = regions()
        t0 = time.perf_counter()
        out = stitcher.stitch(regs, blender, crop=crop)
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        a = regs[0].img
        dt2 = time.perf_counter() - t0
        print(f"{name} stitch(regions, {blender.__name__}, crop={crop}): {dt * 1e3:.1f} ms -> {out.shape}; "
              f"first access of regs[0].img ({a.dtype}, {a.shape}): {dt2 * 1e3:.1f} ms")
