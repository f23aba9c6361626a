#!/usr/bin/env python3
"""Host-side profile (cProfile) of one rank's strip step at world N: where the time that does
not shrink with N goes.   python tools/strip_host_profile.py [cfg3] [world] [rank]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pano360_amd import dist as pdist  # noqa: E402
from pano360_amd import engine, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else world // 2
cfg = dict(synth.CONFIGS[name])
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"],
                                 sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
eng = engine.Engine()
st = pdist.ShardedStitcher(eng, shapes, rots, intrs, cfg["n_levels"], rank, world, exchange=None)
frames = eng.upload_frames([synth.make_frame(i, cfg["width"], cfg["height"], "A")
                            for i in st.my_frames])
out = torch.zeros(engine.Plan(shapes, rots, intrs, True, 10 ** 9).shape + (3,),
                  dtype=torch.uint8, device=eng.device)


def step():
    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    eng.upload_plan(plan)
    eng.multiband_fused(frames, plan, cfg["n_levels"], frame_ids=st.my_frames,
                        strip=st.strip, mosaic_out=out)


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
host = (time.perf_counter() - t0) / 50 * 1e3
torch.cuda.synchronize()
print(f"world {world} rank {rank}: host {host:.3f} ms per step, "
      f"with the queue drained {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
prof = cProfile.Profile()
prof.enable()
for _ in range(50):
    step()
prof.disable()
torch.cuda.synchronize()
pstats.Stats(prof).sort_stats("tottime").print_stats(40)
