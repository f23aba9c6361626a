#!/usr/bin/env python3
"""cProfile of the host side of the fused stitch (config 3), to see what runs
between the owned-regions sync and the first dependent launch."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pano360_amd import engine, synth  # noqa: E402

cfg = dict(synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"])
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"],
                                 sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
eng = engine.Engine()
frames = [eng.upload_frames([synth.make_frame(i, cfg["width"], cfg["height"], "A")])[0]
          for i in range(cfg["n"])]


def step():
    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    eng.stitch(frames, plan, "multiband", cfg["n_levels"])


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print("ms per step:", (time.perf_counter() - t0) / 20 * 1e3)
t0 = time.perf_counter()
for _ in range(20):
    engine.Plan(shapes, rots, intrs, True, 10 ** 9)
print("Plan alone ms:", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
