import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from pano360_amd import engine, synth
for name in ("cfg2", "cfg3", "cfg5"):
    cfg = synth.CONFIGS[name]
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
    cm = 3
    why = dict(left=0, right=0, vx0=0, vx1=0, h=0, regular=0)
    steps = dict(reg=0, irr=0)
    for rec in table.host:
        ax0, ay0, aw, ah = int(rec["ax0"]), int(rec["ay0"]), int(rec["aw"]), int(rec["ah"])
        if aw <= 0 or ah <= 0: continue
        w, h, vx0, vw = int(rec["w"]), int(rec["h"]), int(rec["vx0"]), int(rec["vw"])
        gx0 = (ax0 >> 5) << 5
        ntx = ((ax0 + aw - 1) >> 5) - (ax0 >> 5) + 1
        O0, O1 = ay0 >> 5, (ay0 + ah - 1) >> 5
        nty = O1 - O0 + 1
        g = on[int(rec["tiles_off"]):int(rec["tiles_off"]) + ntx * nty].reshape(nty, ntx).astype(bool)
        colany = g.any(axis=0)
        tx = 0
        while tx < ntx:
            if not colany[tx]:
                tx += 1; continue
            tx0 = tx; tx += 2
            X0 = gx0 + 32 * tx0
            bands = int(np.convolve(g[:, tx0:tx0 + 2].any(axis=1).astype(int), np.ones(5, int), "same").astype(bool).sum()) + 4
            bad = []
            if X0 - 16 * cm < 0: bad.append("left")
            if X0 + 64 + 16 * cm > w: bad.append("right")
            if vx0 & 3: bad.append("vx0")
            if ((vx0 + vw) & 3) and vx0 + vw != w: bad.append("vx1")
            if h < 128: bad.append("h")
            if bad:
                for b in bad: why[b] += 1
                steps["irr"] += bands
            else:
                why["regular"] += 1; steps["reg"] += bands
    print(name, why, steps, "irregular share of steps %.1f %%" % (100.0 * steps["irr"] / (steps["irr"] + steps["reg"])))
    rec = table.host[0]; print("  record 0:", {k: int(rec[k]) for k in ("w", "h", "vx0", "vw", "ax0", "aw")})
