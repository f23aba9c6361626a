#!/usr/bin/env python3
"""Known-byte-count probe for the blur kernels' PMC traffic: one plane per shape,
so rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE can be compared with width*height*4."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pano360_amd import engine  # noqa: E402

eng = engine.Engine()
for h, w in ((4096, 4096), (4096, 4100), (2473, 516), (2473, 2064)):
    plane = torch.rand((h, w), device=eng.device)
    for sigma in (4.0, 10.583):
        k = engine.gaussian_ksize(sigma)
        eng.blur_plane(plane, k, sigma)
        torch.cuda.synchronize()
        print(f"shape {h}x{w} taps {k}: plane bytes {h * ((w + 3) & ~3) * 4}")
