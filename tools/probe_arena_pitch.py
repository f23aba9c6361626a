#!/usr/bin/env python3
"""Same pages, other row pitches: the arenas in allocations made once, the records' pitches padded through the
environment (library built with -DLAYOUT_PAD_PROBE): does the blur's time in a 'slow' allocation respond to
the strides of its streams?
    PANO_LIB=build/variants/pad_probe/libpano360_hip.so python tools/probe_arena_pitch.py [cfg3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pano360_amd import engine, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cfg = dict(synth.CONFIGS[name])
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"],
                                 sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
pool = engine.Engine().upload_frames([synth.make_frame(i, cfg["width"], cfg["height"], "A") for i in range(4)])
frames = [pool[i % 4] for i in range(cfg["n"])]
SIZE = {"planes": 3 << 29, "blurred": 3 << 30, "scratch": 1 << 29}          # bytes
BIG, KEEP = {}, []


def placed(self, nm, floats):
    assert floats * 4 <= SIZE[nm], (nm, floats)
    self._arenas[nm] = BIG[nm]
    return BIG[nm]


def measure(label):
    eng = engine.Engine()
    plan = eng.upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9))
    for _ in range(4):
        eng.stitch(frames, plan, "multiband", cfg["n_levels"])
    torch.cuda.synchronize()
    eng.timing(True)
    for _ in range(30):
        eng.stitch(frames, plan, "multiband", cfg["n_levels"])
    torch.cuda.synchronize()
    t = eng.kernel_times()
    eng.timing(False)
    pick = {k.replace("_kernel", ""): round(v[0] / v[1], 4) for k, v in t.items()
            if k in ("blur_lean_kernel", "multiband_compose_kernel", "warp_windows_kernel")}
    print(f"{label}: {pick}", flush=True)
    del eng


engine.Engine.arena = placed
for trial in range(3):
    KEEP.append(dict(BIG))
    for k, nbytes in SIZE.items():
        BIG[k] = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    print(f"-- allocation {trial}: planes {hex(BIG['planes'].data_ptr())} blurred {hex(BIG['blurred'].data_ptr())}")
    for apad, vpad in ((0, 0), (32, 0), (64, 0), (96, 0), (160, 0), (0, 4), (0, 32), (0, 36), (32, 36), (96, 100), (0, 0)):
        os.environ["PANO_APITCH_PAD"], os.environ["PANO_VPITCH_PAD"] = str(apad), str(vpad)
        measure(f"apitch + {apad:3d}, vpitch + {vpad:3d} floats")
