#!/usr/bin/env python3
"""Random scenes through the three ownership kernels: two levels of bounds (the default), round 4's
one level, and every camera at every pixel - owner and valid maps must be equal bit for bit, on the
whole mosaic and on a column strip, with the region search's boxes and marks equal as well.
    python tools/fuzz_ownership.py [seeds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pano360_amd import _lib, bundle_adj, engine, synth  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
eng = engine.Engine("cuda:0")
bad = 0
for seed in range(seeds):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(2, 48))
    w, h = int(rng.integers(40, 900)), int(rng.integers(30, 500))
    spread = rng.uniform(0.05, 2.8)
    rots = np.stack([bundle_adj.rotation_to_mat(
        [rng.normal(0, rng.uniform(0.0, 0.3)), rng.uniform(-spread, spread), rng.normal(0, rng.uniform(0.0, 0.4))])
        for _ in range(n)])
    intrs = np.stack([bundle_adj.intrinsics(synth.focal_for(w, rng.uniform(35, 110)),
                                            (rng.normal(0, 5), rng.normal(0, 5)))
                      for _ in range(n)]).astype(np.float64)
    cap = int((10 ** 9, 2500, 900, 300)[seed % 4])
    try:
        plan = eng.upload_plan(engine.Plan([(h, w)] * n, rots, intrs, seed % 2 == 0, cap))
    except ValueError:
        continue                       # (a frame that projects to an empty patch)
    H, W = plan.shape
    if H * W > 60e6:
        continue
    strip = None if seed % 3 else (W // 3 + 1, max(2 * W // 3, W // 3 + 2))
    got = {}
    for opt in (0, 3, 1):
        eng.set_option(_lib.OPT_OWN_PRUNE, opt)
        o, v, reg, marks = eng.ownership_regions(plan, strip=strip, min_gap=5, max_spans=3)
        got[opt] = tuple(t.clone() for t in (o, v, reg, marks))
    eng.set_option(_lib.OPT_OWN_PRUNE, 1)
    torch.cuda.synchronize()
    c0, c1 = strip if strip else (0, W)
    ok = True
    for opt in (1, 3):
        ok &= torch.equal(got[opt][0][:, c0:c1], got[0][0][:, c0:c1])
        ok &= torch.equal(got[opt][1][:, c0:c1], got[0][1][:, c0:c1])
        ok &= torch.equal(got[opt][2][:, :5], got[0][2][:, :5])
        ok &= torch.equal(got[opt][3][:, c0:c1], got[0][3][:, c0:c1])
    owned = int((got[0][0][:, c0:c1] >= 0).sum())
    print(f"seed {seed}: n {n} frame {w}x{h} mosaic {H}x{W} strip {strip} owned {owned} "
          f"{'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
