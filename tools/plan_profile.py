#!/usr/bin/env python3
"""Where engine.Plan's host time goes (config 3 cameras), on this host's cores."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pano360_amd import engine, synth
cfg = dict(synth.CONFIGS["cfg3"])
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"], sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
def T(f, n=300):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("plan            %.3f ms" % T(lambda: engine.Plan(shapes, rots, intrs, True, 10**9)))
p = engine.Plan(shapes, rots, intrs, True, 10**9)
print("  inv           %.3f" % T(lambda: np.linalg.inv(np.asarray(intrs, np.float64).reshape(-1,3,3))))
kinv = np.linalg.inv(np.asarray(intrs, np.float64).reshape(-1,3,3))
print("  homs          %.3f" % T(lambda: [np.asarray(r).T.dot(ki) for r, ki in zip(rots, kinv)]))
print("  projs         %.3f" % T(lambda: [np.ascontiguousarray(np.asarray(k).dot(r), np.float64) for r, k in zip(rots, intrs)]))
print("  ranges        %.3f" % T(lambda: engine.ranges_from_border(p.shapes, p.homs)))
ring = engine.border_ring(p.shapes[0])
print("    32 dots     %.3f" % T(lambda: np.stack([h.dot(ring) for h in p.homs])))
pts = np.stack([h.dot(ring) for h in p.homs]); x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
print("    arctan2 th  %.3f" % T(lambda: np.arctan2(x, z)))
print("    arctan2 ph  %.3f" % T(lambda: np.arctan2(y, np.sqrt(x ** 2 + z ** 2))))
print("  shapes tuple  %.3f" % T(lambda: [tuple(int(v) for v in s) for s in shapes]))
print("  resolution    %.3f" % T(lambda: engine.resolution_for(p.ranges, p.shapes[16], p.homs[16], 10**9)))
W, H = p.shape[1], p.shape[0]
theta = np.arange(W, dtype=np.int64) * p.resolution[0] + p.low[0]
phi = np.arange(H, dtype=np.int64) * p.resolution[1] + p.low[1]
print("  theta/phi     %.3f" % T(lambda: (np.arange(W, dtype=np.int64) * p.resolution[0] + p.low[0], np.arange(H, dtype=np.int64) * p.resolution[1] + p.low[1])))
print("  sin+cos full  %.3f" % T(lambda: (np.sin(theta), np.cos(theta))))
print("  sin+cos 1/8   %.3f" % T(lambda: (np.sin(theta[5000:7000]), np.cos(theta[5000:7000]))))
print("  tan           %.3f" % T(lambda: np.tan(phi)))
ok = all(np.array_equal(np.sin(theta)[a:b], np.sin(theta[a:b])) and np.array_equal(np.cos(theta)[a:b], np.cos(theta[a:b]))
         for a, b in [(1, 7), (3, 1000), (5001, 7003), (13001, W), (0, 1), (777, 778), (6, 6 + 1723)])
print("slices of sin / cos equal sin / cos of slices:", ok)
import math
print("np.sin == math.sin on the table:", all(math.sin(v) == s for v, s in zip(theta[:2000], np.sin(theta[:2000]))),
      " arctan2:", all(math.atan2(a, b) == c for a, b, c in zip(x.ravel()[:2000], z.ravel()[:2000], np.arctan2(x, z).ravel()[:2000])))
