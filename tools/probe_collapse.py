"""What the collapse has to read (config 3 by default): the non-interior pixels under every record's
rectangle A, as bytes (76 B per pixel and record at L = 5) and as 128-byte lines per plane and row
(what a cache that fetches whole lines has to move at least), against the counted traffic."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pano360_amd import engine, synth

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cfg = synth.CONFIGS[name]
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"], sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
eng = engine.Engine("cuda:0")
k = min(cfg["n"], 6)
pool = [eng.upload_frames([synth.make_frame(i, cfg["width"], cfg["height"], "A")])[0] for i in range(k)]
frames = [pool[i % k] for i in range(cfg["n"])]
plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
mosaic, _, valid, patches = eng.stitch(frames, plan, "multiband", cfg["n_levels"])
torch.cuda.synchronize()
H, W = plan.shape
ws = next(iter(eng._stitch_ws.values()))
ib = eng.interior_block
inter = ws["interior"].cpu().numpy().astype(bool)
non = ~np.repeat(np.repeat(inter, ib, axis=0), ib, axis=1)[:H, :W]
L = cfg["n_levels"]
per_px = 12 + 16 * (L - 1)
planes = 3 + 4 * (L - 1)
table, _ = eng.last_tiles
need_px = 0
lines = 0
runs = 0
for rec in table.host:
    aw, ah = int(rec["aw"]), int(rec["ah"])
    if aw <= 0 or ah <= 0:
        continue
    x0, y0 = int(rec["x0"]) + int(rec["ax0"]), int(rec["y0"]) + int(rec["ay0"])
    sub = non[y0:y0 + ah, x0:x0 + aw]
    need_px += int(sub.sum())
    # rows of a plane are 128-byte aligned: pixel column c of A lies in line c // 32
    cols = np.arange(aw) // 32
    for row in sub:
        if row.any():
            lines += len(np.unique(cols[row]))
            d = np.diff(np.concatenate([[0], row.view(np.int8), [0]]))
            runs += int((d == 1).sum())
M = H * W
print(name, "mosaic", (H, W), "non-interior pixels %.2f MP of %.2f" % (non.sum() / 1e6, M / 1e6))
print("  (pixel, record) pairs the collapse reads: %.2f M -> %.3f GB at %d B" % (need_px / 1e6, need_px * per_px / 1e9, per_px))
print("  whole 128-byte lines per plane and row: %.2f M x %d planes x 128 B = %.3f GB  (runs: %.2f M, %.1f pixels each)"
      % (lines / 1e6, planes, lines * planes * 128 / 1e9, runs / 1e6, need_px / max(runs, 1)))
print("  + maps and mosaic: %.3f GB (owner 2, valid 1, mosaic 3 B per mosaic pixel)" % (6 * M / 1e9))
