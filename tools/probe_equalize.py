"""Cost of equalize_gains (overlap statistics of all camera pairs) and of a stitch
with per-camera colour tables, against the plain stitch."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pano360_amd import _lib, engine, synth
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cfg = dict(synth.CONFIGS[name])
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"], sweep_deg=cfg.get("sweep_deg"))
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
eng = engine.Engine("cuda:0")
frames = [eng.upload_frames([synth.make_frame(i, cfg["width"], cfg["height"], "B")])[0] for i in range(cfg["n"])]
plan = engine.Plan(shapes, rots, intrs, True, 10**9)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / reps, out
ms, (ov, sz, gains, luts) = timed(lambda: eng.equalize_gains(frames, rots, intrs))
pairs = engine.overlap_pairs(rots, intrs, cfg["width"], cfg["height"])
print(f"{name}: equalize_gains {ms:.2f} ms, {len(pairs)} pairs, {int((sz > 0).sum() // 2)} overlapping, gains {gains.min():.4f}..{gains.max():.4f}")
eng.lib.pano_timing_enable(1)
eng.equalize_gains(frames, rots, intrs); torch.cuda.synchronize()
import ctypes as C
for kid in range(eng.lib.pano_kernel_count()):
    tot, n = C.c_double(0), C.c_int(0)
    eng.lib.pano_timing_read(kid, C.byref(tot), C.byref(n))
    if n.value: print("  ", eng.lib.pano_kernel_name(kid).decode(), f"{tot.value:.3f} ms in {n.value} launches")
eng.lib.pano_timing_enable(0)
a, _ = timed(lambda: eng.stitch(frames, plan, "multiband", cfg["n_levels"]), 10)
b, _ = timed(lambda: eng.stitch(frames, plan, "multiband", cfg["n_levels"], luts=luts), 10)
c, _ = timed(lambda: eng.stitch(frames, plan, "linear"), 10)
d, _ = timed(lambda: eng.stitch(frames, plan, "linear", luts=luts), 10)
print(f"multiband {a:.2f} ms, with tables {b:.2f} ms; linear {c:.2f} ms, with tables {d:.2f} ms")
