#!/usr/bin/env python3
"""Times of the entry points the bench does not cover, at BASELINE sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pano360_amd import blend, engine, synth
eng = engine.engine()
def timeit(f, n=5):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name in ("cfg3", "cfg5"):
    cfg = synth.CONFIGS[name]
    n, w, h = cfg["n"], cfg["width"], cfg["height"]
    rots, intrs = synth.make_cameras(n, w, h, sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
    base = eng.upload_frames([synth.make_frame(i, w, h, "A") for i in range(4)])
    frames = [base[i % 4] for i in range(n)]
    plan = engine.Plan([(h, w)] * n, rots, intrs, False, 10 ** 9)
    for kind in ("linear", "none"):
        print(name, kind, "%.3f ms" % timeit(lambda: eng.stitch(frames, plan, kind)))
    if name == "cfg3":
        print(name, "equalize_gains %.2f ms" % timeit(lambda: eng.equalize_gains(frames, rots, intrs), 3))
a = synth.make_frame(1, 3840, 2160, "B"); b = synth.make_frame(2, 3840, 2160, "B")
print("laplacian_blending 4K, 6 levels: %.2f ms (host arrays in, host array out)" % timeit(lambda: blend.laplacian_blending(a, b), 3))
imgs = [synth.make_frame(i, 3840, 2160, "A") for i in range(4)]
print("shrink_images 4 x 4K by 4: %.2f ms (host arrays in, device frames out)" % timeit(lambda: blend.shrink_images(imgs, 4.0), 3))
