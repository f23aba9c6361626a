"""CPU restatement of ``blend.laplacian_blending`` (reference blend.py:105-140) and of the
8-bit ``cv2.resize`` the CLI applies to every input (stitcher.py:418-420).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this.  The pyramid primitives are the NumPy restatements of OpenCV's pyrDown /
pyrUp in cv2_shim.py (parity unpinned at that boundary, see its header); the function
below is pinned bit for bit by tests/golden/laplacian.npz, which was produced by running
the reference's own blend.py on top of the same primitives (oracle/gen_golden.py).
"""
import numpy as np

import cv2_shim as cv


def gaussian_pyr(img, n_levels):                   # blend.py:117-122
    pyr = [img]
    for _ in range(n_levels):
        img = cv.pyrDown(img)
        pyr.append(img)
    return pyr


def laplacian_pyr(img, n_levels):                  # blend.py:124-130
    pyr = gaussian_pyr(img, n_levels)
    lap = [pyr[-1]]
    for idx in range(n_levels, 0, -1):
        im_ = pyr[idx - 1]
        lap.append(im_ - cv.pyrUp(pyr[idx])[:im_.shape[0], :im_.shape[1]])
    return lap


def default_mask(shape):                           # blend.py:107-111
    hh_, ww_, cc_ = shape
    mask = np.linspace(1, -1, ww_).reshape((1, ww_, 1))
    mask = 1.0 / (1 + np.exp(-100 * mask))
    return np.tile(mask, (hh_, 1, cc_))


def laplacian_blending(img1, img2, mask=None, n_levels=6):
    if mask is None:
        mask = default_mask(img1.shape)
    if mask.shape[2] == 1:                         # blend.py:113-114
        mask = np.repeat(mask, img1.shape[2], axis=2)
    pyr1 = laplacian_pyr(img1.astype("float32"), n_levels)
    pyr2 = laplacian_pyr(img2.astype("float32"), n_levels)
    pyrm = gaussian_pyr(mask, n_levels)[::-1]
    pyrs = [la * gm + lb * (1.0 - gm) for la, lb, gm in zip(pyr1, pyr2, pyrm)]   # :136
    blended = pyrs[0]
    for ls_ in pyrs[1:]:                           # :137-138
        blended = ls_ + cv.pyrUp(blended)[:ls_.shape[0], :ls_.shape[1]]
    return np.clip(blended, 0, 255).astype("uint8")


def shrink(img, factor):
    """``cv2.resize(im, None, fx=1/shrink, fy=1/shrink)`` (stitcher.py:419-420)."""
    return cv.resize(img, None, fx=1 / factor, fy=1 / factor)
