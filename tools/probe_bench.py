"""Per-step wall time of the first steps of a process (clock ramp / allocator priming)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pano360_amd import engine, synth
cfg = dict(synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"])
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"], sweep_deg=cfg.get("sweep_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
eng = engine.Engine("cuda:0")
frames = [eng.upload_frames([synth.make_frame(i, cfg["width"], cfg["height"], "A")])[0] for i in range(cfg["n"])]
times = []
for k in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan = engine.Plan(shapes, rots, intrs, True, 10**9)
    eng.stitch(frames, plan, "multiband", cfg["n_levels"])
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.1f" % t for t in times))
print("reserved GB", torch.cuda.memory_reserved() / 1e9, "allocated", torch.cuda.memory_allocated() / 1e9)
