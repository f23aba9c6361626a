#!/usr/bin/env python3
"""The blur's time against WHICH allocations back its arenas: fresh hipMalloc'ed blocks per trial (the
earlier ones kept alive, so every trial gets other pages), with two plain measurements of each block -
a streaming fill and a read of one float every 4 KiB - to see whether a slow block is slow for everything.
    python tools/probe_arena_lottery.py [cfg3] [trials]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pano360_amd import engine, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = dict(synth.CONFIGS[name])
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"],
                                 sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
pool = engine.Engine().upload_frames([synth.make_frame(i, cfg["width"], cfg["height"], "A") for i in range(4)])
frames = [pool[i % 4] for i in range(cfg["n"])]
SIZE = {"planes": 1 << 30, "blurred": 5 << 29, "scratch": 1 << 29}          # bytes
BIG, KEEP = {}, []


def placed(self, nm, floats):
    assert floats * 4 <= SIZE[nm], (nm, floats)
    self._arenas[nm] = BIG[nm]
    return BIG[nm]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def block_speed(t):
    fill = t.numel() * 4 / timed(lambda: t.fill_(1.0)) / 1e6                      # GB/s
    pages = t[:t.numel() // 1024 * 1024].view(-1, 1024)[:, 0]
    sparse = pages.numel() / timed(lambda: pages.sum()) / 1e3                     # M pages / s
    out = f"fill {fill:6.0f} GB/s, one float per 4 KiB {sparse:6.1f} M pages/s"
    # the same float of every page in a RANDOM order of pages (4 KiB, 64 KiB, 2 MiB apart): translations
    for step in (1024, 16384, 524288):
        rows = t[:t.numel() // step * step].view(-1, step)
        order = torch.randperm(rows.shape[0], device=t.device)
        col = rows[:, 0]
        ms = timed(lambda: col[order].sum())
        out += f"; random {step * 4 >> 10} KiB pages {rows.shape[0] / ms / 1e3:7.1f} M/s"
    return out


engine.Engine.arena = placed
for trial in range(trials):
    KEEP.append(dict(BIG))
    for k, nbytes in SIZE.items():
        BIG[k] = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
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
    print(f"trial {trial}: {pick}")
    for k in ("planes", "blurred"):
        print(f"    {k:8s} {hex(BIG[k].data_ptr())}: {block_speed(BIG[k])}", flush=True)
    del eng
