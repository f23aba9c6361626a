#!/usr/bin/env python3
"""Time pano_knn2 on RootSIFT-like descriptors: python tools/probe_knn.py [nq] [nt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pano360_amd import engine, features
eng = engine.Engine()
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
rng = np.random.default_rng(0)
a = rng.random((nq, 128), dtype=np.float32); b = rng.random((nt, 128), dtype=np.float32)
a = np.sqrt(a / a.sum(1, keepdims=True)); b = np.sqrt(b / b.sum(1, keepdims=True))
da, db = torch.from_numpy(a).to(eng.device), torch.from_numpy(b).to(eng.device)
for _ in range(2):
    features.knn2_device(da, db, eng=eng)
torch.cuda.synchronize()
eng.timing(True)
t0 = time.perf_counter()
for _ in range(5):
    idx, dist, res = features.knn2_device(da, db, eng=eng, want_rescans=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
k = eng.kernel_times()
flop = 2.0 * nq * nt * 128
print(f"{nq} x {nt} x 128: {ms:.2f} ms per search, {flop / ms / 1e9:.1f} TFLOP/s of float32-equivalent "
      f"cross terms ({3 * flop / ms / 1e9:.0f} TFLOP/s of f16 products), {res} rescans; kernels "
      f"{ {n: round(v[0] / 5, 3) for n, v in k.items()} }")
