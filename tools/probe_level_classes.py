"""What per-level interior radii would save, counted exactly on a config's real
geometry before any kernel is touched.

Today ONE radius R (the largest level's, 43 / 48) decides where a pixel needs the blend: there all
L - 1 blurred copies are stored by the blur and gathered by the collapse.  A pixel whose (2 r_k + 1)^2
window holds one owner for the levels k < j (class j: the interior test at radius r_{j-1} passes, at
r_j fails) needs, per record that covers it,
    j = 0:  planes (12 B) + all L - 1 copies (16 B each)                      - as today
    j >= 1: G_{j-1} colour (12 B) + copies j .. L-2 (16 B each) (+ planes, 12 B, on the owner's record)
because the levels below j telescope to I - G_{j-1} I of the owner alone.  A 32 x 32 blur tile has to
store level k when some pixel under it has class <= k + 1.

    python tools/probe_level_classes.py [cfg3|cfg2|cfg5]

Prints the collapse's gather bytes and the blur's store bytes now and with classes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pano360_amd import engine, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cfg = synth.CONFIGS[name]
n, w, h, L = cfg["n"], cfg["width"], cfg["height"], cfg["n_levels"]
rots, intrs = synth.make_cameras(n, w, h, sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
shapes = [(h, w)] * n
eng = engine.Engine("cuda:0")
pool = eng.upload_frames([synth.make_frame(i, w, h, "A") for i in range(2)])
frames = [pool[i % 2] for i in range(n)]
plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
_, _, _, patches = eng.stitch(frames, plan, "multiband", L)
table, flags = eng.last_tiles
on = flags.cpu().numpy()
H, W = plan.shape
owner, valid = eng.ownership_cameras(plan)
radii = [engine.gaussian_ksize(s) // 2 for s in engine.level_sigmas(L)]
ib = eng.interior_block
# class of every block: the number of leading levels whose interior test passes
cls = np.zeros(((H + ib - 1) // ib, (W + ib - 1) // ib), np.int32)
for r in radii:
    cls += eng.interior_map(owner, r).cpu().numpy().astype(np.int32)
own_np = owner.cpu().numpy()
# (radii grow with the level, so the tests are nested: a block that passes at r_k passes below)
nb = L - 1
px_cls = np.repeat(np.repeat(cls, ib, 0), ib, 1)[:H, :W]
gather_now = gather_cls = 0.0
store_now = store_cls = 0.0
tiles_now = 0
tiles_lvl = np.zeros(nb, np.int64)
for rec in table.host:
    ax0, ay0, aw, ah = int(rec["ax0"]), int(rec["ay0"]), int(rec["aw"]), int(rec["ah"])
    if aw <= 0 or ah <= 0:
        continue
    my, mx = int(rec["y0"]) + ay0, int(rec["x0"]) + ax0            # A in mosaic coordinates
    y0, y1, x0, x1 = max(my, 0), min(my + ah, H), max(mx, 0), min(mx + aw, W)
    if y1 <= y0 or x1 <= x0:
        continue
    c = px_cls[y0:y1, x0:x1]
    mine = own_np[y0:y1, x0:x1] == int(rec["index"])
    seam = c < nb                                                   # not interior at R: gathered today
    gather_now += float(seam.sum()) * (12 + 16 * nb)
    per = np.where(c == 0, 12 + 16 * nb, 12 + 16 * (nb - np.minimum(c, nb)) + 12 * mine)
    gather_cls += float(per[seam].sum())
    # the record's 32 x 32 tiles (anchored at multiples of 32 in patch coordinates)
    gx0 = (ax0 >> 5) << 5
    ntx = ((ax0 + aw - 1) >> 5) - (ax0 >> 5) + 1
    O0 = ay0 >> 5
    nty = ((ay0 + ah - 1) >> 5) - O0 + 1
    g = on[int(rec["tiles_off"]):int(rec["tiles_off"]) + ntx * nty].reshape(nty, ntx).astype(bool)
    tiles_now += int(g.sum())
    for ty, tx in zip(*np.nonzero(g)):
        ty0, tx0 = 32 * (O0 + ty) - ay0 + my, gx0 + 32 * tx - ax0 + mx      # tile in mosaic coordinates
        sub = px_cls[max(ty0, y0):min(ty0 + 32, y1), max(tx0, x0):min(tx0 + 32, x1)]
        if sub.size == 0:
            continue
        lo = int(sub.min())
        for k in range(nb):
            if lo <= k + 1:
                tiles_lvl[k] += 1
store_now = tiles_now * 1024.0 * 16 * nb
store_cls = float(tiles_lvl.sum()) * 1024.0 * 16
print(f"{name}: L = {L}, radii {radii}, block {ib}; pixels by class "
      + ", ".join(f"{j}: {int((px_cls == j).sum()) / 1e6:.2f} MP" for j in range(nb + 1)))
print(f"collapse gathers: {gather_now / 1e9:.3f} GB now, {gather_cls / 1e9:.3f} GB with classes "
      f"({100 * (1 - gather_cls / gather_now):.1f} % less)")
print(f"blur stores: {tiles_now} active tiles, {store_now / 1e9:.3f} GB now; per level "
      f"{tiles_lvl.tolist()} tiles, {store_cls / 1e9:.3f} GB with classes "
      f"({100 * (1 - store_cls / store_now):.1f} % less)")
