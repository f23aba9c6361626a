#!/usr/bin/env python3
"""Does the blur's time depend on where its arenas start?  One process, config 3, the three arenas as views
at chosen byte offsets into three allocations made ONCE (same pages throughout), one arena moved at a
time; then the same offsets in fresh allocations (other pages): blur / collapse / warp ms per launch.
    python tools/probe_arena_skew.py [cfg3]"""
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
SLACK = 4 << 20
SIZE = {"planes": 1 << 30, "blurred": 5 << 29, "scratch": 1 << 29}          # bytes
BIG = {}
OFF = {"planes": 0, "blurred": 0, "scratch": 0}


def fresh_blocks():
    BIG.clear()
    torch.cuda.empty_cache()
    for k, nbytes in SIZE.items():
        BIG[k] = torch.empty((nbytes + SLACK) // 4, dtype=torch.float32, device="cuda")


def placed(self, nm, floats):
    assert floats * 4 <= SIZE[nm], (nm, floats)
    view = BIG[nm][OFF[nm] // 4:OFF[nm] // 4 + SIZE[nm] // 4]
    self._arenas[nm] = view
    return view


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
            if k in ("blur_lean_kernel", "blur_lean5_kernel", "multiband_compose_kernel", "warp_windows_kernel")}
    ptrs = {k: hex(v.data_ptr()) for k, v in eng._arenas.items() if v is not None and k != "scratch"}
    print(f"{label}: {pick} {ptrs}", flush=True)
    del eng


engine.Engine.arena = placed
OFFSETS = (0, 128, 256, 1024, 4096, 65536, 1 << 20, (1 << 21) + 128)
for trial in range(2):
    fresh_blocks()
    print(f"-- allocation {trial}: planes {hex(BIG['planes'].data_ptr())} blurred {hex(BIG['blurred'].data_ptr())}")
    for which in ("blurred", "planes"):
        for off in OFFSETS:
            OFF.update(planes=0, blurred=0, scratch=0)
            OFF[which] = off
            measure(f"{which} + {off:>8} B")
    OFF.update(planes=128, blurred=128)
    measure("both + 128 B")
    OFF.update(planes=0, blurred=0)
    measure("both + 0 (again)")
