"""Phase timers of the matrix-core blur (library built with EXTRA=-DMB_STAMP): cycles per step
of wave 0 (heaviest level) and wave 4 (lightest) of a sample of colour workgroups, cfg3."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pano360_amd import engine, synth
import torch
cfg = synth.CONFIGS["cfg3"]
rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"], sweep_deg=cfg["sweep_deg"])
shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
eng = engine.Engine("cuda:0")
frames = [eng.upload_frames([synth.make_frame(i, cfg["width"], cfg["height"], "A")])[0] for i in range(cfg["n"])]
lib = eng.lib
for it in range(3):
    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    eng.stitch(frames, plan, "multiband", int(os.environ.get("LEVELS", "5")))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 56)()
    lib.pano_debug_stamps(buf, 1)
su = np.array(buf[48:56], dtype=np.float64)
if su[7]:
    print("set-up per regular workgroup (thread 0, cycles): item / geometry %.0f | table copy %.0f | need flags + barrier %.0f | "
          "list %.0f  (%d workgroups)" % (su[0] / su[7], su[1] / su[7], su[2] / su[7], su[3] / su[7], int(su[7])))
if os.environ.get("STAMP_FORM", "stream") == "stream":    # ms_body: waves 0, 2, 4, 6
    names = ["head", "barrier", "prologue", "row pass + stores", "split + column pass", "wait + commit"]
    for w in range(4):
        v = np.array(buf[12 * w:12 * w + 12], dtype=np.float64)
        n = max(v[11], 1)
        print("wave", 2 * w, "steps", int(v[11]), " | ".join(f"{nm} {v[k] / n:.0f}" for k, nm in enumerate(names)),
              "| sum %.0f" % (v[:6].sum() / n))
else:
    names = ["skeleton", "barrier 1", "conversion", "barrier 2", "stores", "fetch", "row pass", "column pass"]
    for w in range(2):
        v = np.array(buf[12 * w:12 * w + 12], dtype=np.float64)
        n = max(v[11], 1)
        print("wave", 4 * w, "steps", int(v[11]), " ".join(f"{nm} {v[k] / n:.0f}" for k, nm in enumerate(names)), "| sum %.0f" % (v[:8].sum() / n))
