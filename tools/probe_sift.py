"""Time of SIFT on a frame: the Gaussian / DoG scale space alone, and detectAndCompute
(features.py:192-201 inside OpenCV)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pano360_amd import engine, features, synth
eng = engine.engine()
w, h = (3840, 2160) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
kind = sys.argv[3] if len(sys.argv) > 3 else "B"
frames = [eng.upload_frames([synth.make_frame(i, w, h, kind)])[0] for i in range(4)]
def timed(fn, reps=8):
    fn(frames[0]); fn(frames[1])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(reps):
        out = fn(frames[k % 4])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out
dt, (g, d) = timed(features.sift_pyramid_device)
px = w * h
gauss_px = sum(p.numel() for p in g); dog_px = sum(p.numel() for p in d)
print(f"{w}x{h} scale space: {dt*1e3:.2f} ms per frame, {px/dt/1e6:.0f} MP/s of input, octaves {len(g)}, "
      f"{gauss_px/1e6:.0f} MP of Gaussian + {dog_px/1e6:.0f} MP of DoG planes, "
      f"{(px + 12*gauss_px + 12*dog_px)/dt/1e9:.0f} GB/s algorithmic")
dt2, (kps, desc) = timed(features.sift_detect_device, 4)
print(f"{w}x{h} detectAndCompute: {dt2*1e3:.2f} ms per frame, {len(kps)} keypoints, "
      f"{len(kps)/dt2/1e6:.2f} M keypoints/s end to end")
