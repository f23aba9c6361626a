"""Synthetic camera sweeps + frames (SURVEY.md §8d recipes).

No reference code: the reference ships no data set generator.  Cameras are
pinholes with hfov 60 deg on a yaw sweep, optionally jittered; frames are
seeded uint8 noise ("A", stress) or a blurred, rescaled version ("B", smooth,
used for the 1e-4 relative-error criterion).
"""
import numpy as np

from .bundle_adj import intrinsics, rotation_to_mat

HFOV_DEG = 60.0


def focal_for(width, hfov_deg=HFOV_DEG):
    return (width / 2.0) / np.tan(np.deg2rad(hfov_deg) / 2.0)


def make_frame(seed, width, height, kind="A"):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (height, width, 3), dtype=np.uint8)
    if kind == "A":
        return img
    if kind != "B":
        raise ValueError(kind)
    # smooth variant: separable box-ish binomial passes (no SciPy needed so the
    # same bytes are produced everywhere), then stretch to the full range
    f = img.astype(np.float64)
    k = np.array([1, 4, 6, 4, 1], dtype=np.float64) / 16.0
    for _ in range(3):
        f = sum(k[i] * np.roll(f, i - 2, axis=0) for i in range(5))
        f = sum(k[i] * np.roll(f, i - 2, axis=1) for i in range(5))
    lo, hi = f.min(), f.max()
    return np.round((f - lo) * (255.0 / (hi - lo))).astype(np.uint8)


def sweep_yaws(n, sweep_deg=None, step_deg=None):
    if step_deg is not None:
        return np.deg2rad(step_deg) * np.arange(n)
    if n == 1:
        return np.zeros(1)
    return np.deg2rad(sweep_deg) * (np.arange(n) / (n - 1) - 0.5)


def make_cameras(n, width, height, sweep_deg=None, step_deg=None,
                 jitter=0.0, seed=0):
    """Returns (rots [n,3,3], intrs [n,3,3]) float64."""
    yaws = sweep_yaws(n, sweep_deg, step_deg)
    rng = np.random.default_rng(seed)
    intr = intrinsics(focal_for(width))
    rots = []
    for yaw in yaws:
        vec = np.array([0.0, yaw, 0.0])
        if jitter:
            vec = vec + rng.normal(0.0, jitter, 3)
        rots.append(rotation_to_mat(vec))
    return np.stack(rots), np.stack([intr.astype(np.float64)] * n)


# named workloads (BASELINE.json configs 2, 3, 5)
CONFIGS = {
    "cfg2": dict(n=8, width=1920, height=1080, sweep_deg=140.0, n_levels=5),
    "cfg3": dict(n=32, width=3840, height=2160, sweep_deg=155.0, n_levels=5),
    "cfg5": dict(n=120, width=7680, height=4320, step_deg=3.0, n_levels=6),
}


def make_scene(n, width, height, sweep_deg=None, step_deg=None, jitter=0.0,
               seed=0, kind="A", n_levels=5):
    rots, intrs = make_cameras(n, width, height, sweep_deg, step_deg,
                               jitter, seed)
    imgs = [make_frame(seed + i, width, height, kind) for i in range(n)]
    return imgs, rots, intrs
