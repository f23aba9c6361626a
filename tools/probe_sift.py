"""Time of the SIFT Gaussian / DoG scale space (features.py:192-201 inside OpenCV) per 4K frame."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pano360_amd import engine, features, synth
eng = engine.engine()
w, h = (3840, 2160) if len(sys.argv) < 2 else (int(sys.argv[1]), int(sys.argv[2]))
frames = [eng.upload_frames([synth.make_frame(i, w, h, "A")])[0] for i in range(4)]
for f in frames[:2]:
    features.sift_pyramid_device(f)
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = 8
for k in range(reps):
    g, d = features.sift_pyramid_device(frames[k % 4])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
px = w * h
gauss_px = sum(p.numel() for o in g for p in o); dog_px = sum(p.numel() for o in d for p in o)
print(f"{w}x{h}: {dt*1e3:.2f} ms per frame, {px/dt/1e6:.0f} MP/s of input, octaves {len(g)}, "
      f"{gauss_px/1e6:.0f} MP of Gaussian + {dog_px/1e6:.0f} MP of DoG planes, "
      f"{(px + 8*gauss_px + 4*gauss_px + 4*dog_px + 8*dog_px)/dt/1e9:.0f} GB/s algorithmic")
