"""Which stitches of a long run are slow, and whether the allocator is why."""
import os, sys, time, gc
sys.path.insert(0, os.getcwd())
import torch
from pano360_amd import engine, synth
cfg = dict(synth.CONFIGS["cfg3"])
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"], sweep_deg=cfg.get("sweep_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
eng = engine.Engine("cuda:0")
frames = [eng.upload_frames([synth.make_frame(i, cfg["width"], cfg["height"], "A")])[0] for i in range(cfg["n"])]
if len(sys.argv) > 1 and sys.argv[1] == "nogc":
    gc.disable()
prev = torch.cuda.memory_stats()
for k in range(130):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan = engine.Plan(shapes, rots, intrs, True, 10**9)
    t1 = time.perf_counter()
    eng.stitch(frames, plan, "multiband", cfg["n_levels"])
    t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    st = torch.cuda.memory_stats()
    if (t3 - t0) > 6e-3:
        print(k, "total %.1f ms: plan %.1f, host part of stitch %.1f, drain %.1f; segments +%d, retries +%d, gc gen counts %s" % (
            (t3 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3,
            st["segment.all.allocated"] - prev["segment.all.allocated"],
            st["num_alloc_retries"] - prev["num_alloc_retries"], gc.get_count()))
    prev = st
print("reserved GB", torch.cuda.memory_reserved() / 1e9)
