"""Runs of the blur's work items (cfg3 by default): how many steps a workgroup takes, how many of
them have every tile within reach wanted (the branch-free step of the lean kernel applies), how
many column-pass tile slots belong to unwanted tiles (what computing them anyway would waste)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pano360_amd import engine, synth

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"], sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
eng = engine.Engine("cuda:0")
k = min(cfg["n"], 6)
pool = [eng.upload_frames([synth.make_frame(i, cfg["width"], cfg["height"], "A")])[0] for i in range(k)]
frames = [pool[i % k] for i in range(cfg["n"])]
plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
eng.stitch(frames, plan, "multiband", cfg["n_levels"])
table, flags = eng.last_tiles
on = flags.cpu().numpy()
items = 0
steps_all = []
stat = dict(steps=0, full2=0, full1=0, slots2=0, want2=0, slots1=0, want1=0, edge_store=0, stores=0,
            fetch_interior=0, runs=0, regular=0)
run_lens = []
for rec in table.host:
    ax0, ay0, aw, ah = int(rec["ax0"]), int(rec["ay0"]), int(rec["aw"]), int(rec["ah"])
    if aw <= 0 or ah <= 0:
        continue
    h, vy0, vh = int(rec["h"]), int(rec["vy0"]), int(rec["vh"])
    gx0 = (ax0 >> 5) << 5
    ntx = ((ax0 + aw - 1) >> 5) - (ax0 >> 5) + 1
    O0, O1 = ay0 >> 5, (ay0 + ah - 1) >> 5
    nty = O1 - O0 + 1
    g = on[int(rec["tiles_off"]):int(rec["tiles_off"]) + ntx * nty].reshape(nty, ntx).astype(bool)
    colany = g.any(axis=0)
    tx_ = 0
    while tx_ < ntx:
        if not colany[tx_]:
            tx_ += 1
            continue
        tx0 = tx_
        tx_ += 2
        items += 1
        pad = np.zeros((nty + 8, 2), bool)
        pad[4:4 + nty, :g[:, tx0:tx0 + 2].shape[1]] = g[:, tx0:tx0 + 2]
        want_any = pad.any(axis=1)
        reach = np.convolve(want_any.astype(int), np.ones(5, int), "same") > 0      # listed bands
        idx = np.nonzero(reach)[0]
        steps_all.append(len(idx))
        # runs of consecutive listed bands
        brk = np.nonzero(np.diff(idx) > 1)[0]
        stat["runs"] += len(brk) + 1
        lens = np.diff(np.concatenate([[-1], brk, [len(idx) - 1]]))
        run_lens += list(lens)
        for col in range(2):
            w = pad[:, col]
            for i in idx:
                stat["steps"] += 1
                t = O0 + i - 4
                n2 = int(w[i - 2:i + 3].sum())
                n1 = int(w[i - 1:i + 2].sum())
                stat["full2"] += n2 == 5
                stat["full1"] += n1 == 3
                stat["slots2"] += 5
                stat["want2"] += n2
                stat["slots1"] += 3
                stat["want1"] += n1
                stat["fetch_interior"] += (32 * t >= max(0, vy0)) and (32 * t + 32 <= min(h, vy0 + vh))
                o = t - 2
                if 0 <= i - 2 < len(w) and w[i - 2]:
                    stat["stores"] += 1
                    x0 = gx0 + 32 * (tx0 + col)
                    inside = 32 * o >= ay0 and 32 * o + 32 <= ay0 + ah and x0 >= ax0 and x0 + 32 <= ax0 + aw
                    stat["edge_store"] += not inside
print("items", items, "steps per item: mean %.1f min %d max %d" % (np.mean(steps_all), min(steps_all), max(steps_all)))
print("runs", stat["runs"], "run length mean %.1f median %d" % (np.mean(run_lens), int(np.median(run_lens))),
      "hist(<=8, <=16, <=32, <=64, >64):", [int(((np.array(run_lens) > a) & (np.array(run_lens) <= b)).sum())
                                              for a, b in ((0, 8), (8, 16), (16, 32), (32, 64), (64, 10 ** 6))])
s = stat
print("wave-steps (2 tile columns):", s["steps"])
print("  all five tiles wanted (DMAX 2): %.1f %%   all three (DMAX 1): %.1f %%" % (100.0 * s["full2"] / s["steps"], 100.0 * s["full1"] / s["steps"]))
print("  column-pass slots wanted: DMAX 2 %.1f %%   DMAX 1 %.1f %%" % (100.0 * s["want2"] / s["slots2"], 100.0 * s["want1"] / s["slots1"]))
print("  bands whose 32 rows lie inside the patch and inside V: %.1f %%" % (100.0 * s["fetch_interior"] / s["steps"]))
print("  stored tiles %d, on the edge of A: %.1f %%" % (s["stores"], 100.0 * s["edge_store"] / max(1, s["stores"])))
