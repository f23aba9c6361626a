#!/usr/bin/env python3
"""Time crop_rect (crop_mosaic's rectangle, stitcher.py:340-369) on full-size valid masks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pano360_amd import engine, synth
eng = engine.Engine()
for name in ("cfg2", "cfg3", "cfg5"):
    cfg = synth.CONFIGS[name]
    rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"], sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
    plan = eng.upload_plan(engine.Plan([(cfg["height"], cfg["width"])] * cfg["n"], rots, intrs, True, 10 ** 9))
    _, valid = eng.ownership_cameras(plan)
    for _ in range(2):
        rect = eng.crop_rect(valid)
    torch.cuda.synchronize()
    eng.timing(True)
    t0 = time.perf_counter()
    for _ in range(5):
        rect = eng.crop_rect(valid)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(name, plan.shape, rect, f"{ms:.3f} ms per crop_rect", {k: round(v[0] / 5, 3) for k, v in eng.kernel_times().items()})
    eng.timing(False)
