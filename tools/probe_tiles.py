"""How the active 32 x 32 blur tiles of cfg3 sit in their rectangles A: whole tiles inside A
(mask-free stores) against edge tiles, and fetch groups that straddle a window edge."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pano360_amd import engine, synth

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"], sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
eng = engine.Engine("cuda:0")
frames = [eng.upload_frames([synth.make_frame(i, cfg["width"], cfg["height"], "A")])[0] for i in range(cfg["n"])]
plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
eng.stitch(frames, plan, "multiband", cfg["n_levels"])
table, flags = eng.last_tiles
on = flags.cpu().numpy()
tot = ins = 0
wgs = live = 0
steps = []
for rec in table.host:
    ax0, ay0, aw, ah = int(rec["ax0"]), int(rec["ay0"]), int(rec["aw"]), int(rec["ah"])
    if aw <= 0 or ah <= 0:
        continue
    gx0 = (ax0 >> 5) << 5
    ntx = ((ax0 + aw - 1) >> 5) - (ax0 >> 5) + 1
    O0, O1 = ay0 >> 5, (ay0 + ah - 1) >> 5
    nty = O1 - O0 + 1
    g = on[int(rec["tiles_off"]):int(rec["tiles_off"]) + ntx * nty].reshape(nty, ntx).astype(bool)
    ty, tx = np.nonzero(g)
    x0 = gx0 + 32 * tx; y0 = 32 * (O0 + ty)
    inside = (y0 >= ay0) & (y0 + 32 <= ay0 + ah) & (x0 >= ax0) & (x0 + 32 <= ax0 + aw)
    tot += len(tx); ins += int(inside.sum())
    colany = g.any(axis=0)
    starts, tx_ = [], 0
    while tx_ < ntx:                      # the greedy pairing of mb_columns_kernel
        if colany[tx_]:
            starts.append(tx_)
            tx_ += 2
        else:
            tx_ += 1
    for tx0 in starts:
        cols = g[:, tx0:tx0 + 2]
        wgs += 1
        live += 1
        want = cols.any(axis=1)
        reach = np.convolve(want.astype(int), np.ones(5, int), "same") > 0   # +-2 bands
        steps.append(int(reach.sum()) + 4)
        both = getattr(np, "_both", [])
    if not hasattr(np, "_stat"):
        np._stat = [0, 0]
    np._stat[0] += int(g.sum())
    np._stat[1] += sum(int(g[:, t:t + 2].any(axis=1).sum()) * 2 for t in starts)
print("records", len(table.host), "active tiles", tot, "inside A", ins, "(%.1f%%)" % (100.0 * ins / tot))
print("workgroups per channel", wgs, "live", live, "steps per live WG: mean %.1f min %d max %d" % (np.mean(steps), min(steps), max(steps)))
print("wanted tiles", np._stat[0], "tile slots in wanted bands of the pairs", np._stat[1])
print("total WG-steps x4 channels", 4 * sum(steps), " per CU (256):", 4 * sum(steps) / 256.0)
