#!/usr/bin/env python3
"""The pixels a rank's strip really processes against an eighth (a world-th) of the whole mosaic's:
records, warped window pixels V, rectangle pixels A, active 32 x 32 tile pixels, interior share.
    python tools/probe_strip_pixels.py [cfg3] [world ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pano360_amd import dist as pdist  # noqa: E402
from pano360_amd import engine, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
worlds = [int(v) for v in sys.argv[2:]] or [1, 2, 4, 8]
cfg = dict(synth.CONFIGS[name])
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"],
                                 sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
eng = engine.Engine()
pool = {}
whole = None
for world in worlds:
    for rank in sorted({0, world // 2, world - 1}):
        st = pdist.ShardedStitcher(eng, shapes, rots, intrs, cfg["n_levels"], rank, world, exchange=None)
        for i in st.my_frames:
            if i % 4 not in pool:
                pool[i % 4] = eng.upload_frames([synth.make_frame(i % 4, cfg["width"], cfg["height"], "A")])[0]
        frames = [pool[i % 4] for i in st.my_frames]
        plan = eng.upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9, table_cols=st.table_cols))
        _, _, _, patches = eng.multiband_fused(frames, plan, cfg["n_levels"], frame_ids=st.my_frames,
                                               strip=st.strip)
        torch.cuda.synchronize()
        v = sum((p.window[1] - p.window[0]) * (p.window[3] - p.window[2]) for p in patches)
        a = sum((p.area[1] - p.area[0]) * (p.area[3] - p.area[2]) for p in patches)
        act = eng.active_tile_pixels()
        cols = st.strip[1] - st.strip[0]
        row = dict(records=len(patches), V=v / 1e6, A=a / 1e6, active=act / 1e6,
                   mosaic=plan.shape[0] * cols / 1e6)
        if world == 1:
            whole = row
        share = {k: row[k] / whole[k] * world for k in row} if whole else {}
        print(f"world {world} rank {rank}: strip {cols} columns, " +
              ", ".join(f"{k} {row[k]:.2f}" + (f" ({share[k]:.2f} x its share)" if share else "")
                        for k in row))
