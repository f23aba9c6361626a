#!/usr/bin/env python3
"""Phase timers of the two-level ownership kernel (library built with EXTRA=-DOW_STAMP, selected
through PANO_LIB): cycles thread 0 of a workgroup spends between the kernel's barriers, averaged
over the workgroups of a config's mosaic; with any library: the kernel's time (HIP events).
    PANO_LIB=build/variants/ow_stamp/libpano360_hip.so python tools/probe_own_stamps.py [cfg3]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pano360_amd import engine, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cfg = synth.CONFIGS[name]
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"], sweep_deg=cfg.get("sweep_deg"),
                                 step_deg=cfg.get("step_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
eng = engine.Engine("cuda:0")
plan = eng.upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9))
lib = eng.lib
has = hasattr(lib, "pano_debug_own_stamps")
for _ in range(3):
    eng.ownership_regions(plan)
torch.cuda.synchronize()
if has:
    buf = (C.c_ulonglong * 16)()
    lib.pano_debug_own_stamps(buf, 1)       # rows cleared: the LAST launch's samples are read below
eng.timing(True)
reps = 10
for _ in range(reps):
    eng.ownership_regions(plan)
torch.cuda.synchronize()
times = eng.kernel_times()
eng.timing(False)
print(name, "mosaic", plan.shape, {k: round(v[0] / v[1], 4) for k, v in times.items()}, "ms per launch")
if has:
    lib.pano_debug_own_stamps(buf, 0)
    v = np.array(buf[:], dtype=np.float64)
    wgs = max(v[14], 1)
    names = ["camera list", "records + ranges -> LDS", "level-1 bounds", "survivors", "level-2 bounds",
             "quarter decisions", "evaluation list", "evaluation", "write-out + runs", "box merge"]
    print("sampled workgroups %d, evaluated quarters per workgroup %.2f" % (wgs, v[15] / wgs))
    print("cycles per workgroup (thread 0): " + " | ".join(f"{nm} {v[k] / wgs:.0f}" for k, nm in enumerate(names))
          + " | sum %.0f" % (v[:10].sum() / wgs))
