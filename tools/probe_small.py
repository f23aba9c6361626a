#!/usr/bin/env python3
"""rocprofv3-free timing of the small kernels of a cfg3 stitch via torch events around Engine calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pano360_amd import engine, synth
eng = engine.Engine()
cfg = synth.CONFIGS["cfg3"]
n, w, h = cfg["n"], cfg["width"], cfg["height"]
rots, intrs = synth.make_cameras(n, w, h, sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
plan = eng.upload_plan(engine.Plan([(h, w)] * n, rots, intrs, True, 10 ** 9))
owner, valid = eng.ownership_cameras(plan)
H, W = plan.shape
def t(f, reps=20):
    f(); torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
print("interior_map (block_owner + interior_tile): %.3f ms" % t(lambda: eng.interior_map(owner, 43, (0, W))))
