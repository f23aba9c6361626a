"""NumPy restatement of the Gaussian / DoG scale space of OpenCV's SIFT.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference calls
``cv2.xfeatures2d.SIFT_create().detectAndCompute`` (features.py:192-201) and
holds none of this arithmetic; OpenCV is neither pinned by the reference nor
installed here.  What is restated is OpenCV 3.4/4.x's published SIFT front
end with the defaults the reference uses (sigma 1.6, 3 layers per octave,
first octave -1):

  createInitialImage : BGR2GRAY (8-bit, 14-bit fixed point) -> float ->
                       resize x2 INTER_LINEAR -> GaussianBlur(sqrt(1.6^2 - 1))
  buildGaussianPyramid: 6 images per octave, incremental sigmas, next octave =
                       INTER_NEAREST half of layer 3
  buildDoGPyramid    : differences of neighbouring layers

It checks ``pano360_amd.features.sift_pyramid`` (HIP) in tests/.

Third-party cross-checks of THIS file (tests/test_oracle_golden.py, round 6; implementations
not written for this repo): ``gray_u8`` within one level of Pillow's 'L' conversion (and BGR, not
RGB); ``resize_up2`` = scipy.ndimage.zoom(order=1, grid_mode=True, mode='nearest') = torch's
bilinear interpolate(align_corners=False) to 1e-4 of 255; ``decimate2`` = torch's nearest
interpolate, exactly; every Gaussian layer = scipy's correlate1d with float64 taps of the
aperture rule, 'mirror' borders, the sigma schedule recomputed, to 2e-4; layer i = ONE Gaussian
of total sigma 1.6 * 2^(i/3) on the doubled image (the schedule's defining property) to 0.05.
The keypoint / descriptor stages (oracle/sift_oracle.py) have no such counterpart installed.
"""
import numpy as np

import cv2_shim


def gray_u8(bgr):
    b, g, r = (bgr[..., i].astype(np.int64) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.float32)


def _up2_taps(n):
    d = np.arange(2 * n, dtype=np.float32)
    f = (d + np.float32(0.5)) * np.float32(0.5) - np.float32(0.5)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    s[lo], f[lo] = 0, 0
    hi = s >= n - 1
    s[hi], f[hi] = n - 1, 0
    return s, np.minimum(s + 1, n - 1), (np.float32(1) - f).astype(np.float32), f.astype(np.float32)


def resize_up2(img):
    """resize(img, (2w, 2h), INTER_LINEAR): horizontal taps first, then vertical."""
    img = np.asarray(img, np.float32)
    h, w = img.shape
    x0, x1, a0, a1 = _up2_taps(w)
    y0, y1, b0, b1 = _up2_taps(h)
    rows = img[:, x0] * a0 + img[:, x1] * a1
    return rows[y0] * b0[:, None] + rows[y1] * b1[:, None]


def decimate2(img):
    """resize(img, (w//2, h//2), INTER_NEAREST)."""
    h, w = img.shape
    oh, ow = h // 2, w // 2
    sx = np.minimum(np.floor(np.arange(ow) * (w / ow)).astype(np.int64), w - 1)
    sy = np.minimum(np.floor(np.arange(oh) * (h / oh)).astype(np.int64), h - 1)
    return img[sy][:, sx]


def sigmas(sigma=1.6, layers=3):
    k = 2.0 ** (1.0 / layers)
    out = [sigma]
    for i in range(1, layers + 3):
        prev = k ** (i - 1) * sigma
        out.append(float(np.sqrt((prev * k) ** 2 - prev ** 2)))
    return out


def n_octaves(height, width):
    return int(np.rint(np.log(float(min(2 * height, 2 * width))) / np.log(2.0) - 2)) + 1


def sift_pyramid(bgr, octaves=None, sigma=1.6, layers=3, blur=None):
    """``blur(image, sigma)``: the GaussianBlur to use (default: the NumPy shim's; the C
    oracle's, bit-identical and OpenMP-parallel, for 4K frames)."""
    h, w = bgr.shape[:2]
    if octaves is None:
        octaves = n_octaves(h, w)
    sig_diff = float(np.sqrt(max(np.float32(sigma) ** 2 - np.float32(0.5) ** 2 * 4,
                                 np.float32(0.01))))
    if blur is None:
        blur = lambda im, s: cv2_shim.GaussianBlur(im, (0, 0), s, s)    # noqa: E731
    base = blur(resize_up2(gray_u8(bgr)), sig_diff)
    sig = sigmas(sigma, layers)
    gauss, dog = [], []
    for o in range(octaves):
        if o:
            prev = gauss[-1][layers]
            if min(prev.shape) < 2:
                break
            base = decimate2(prev)
        octave = [base]
        for i in range(1, layers + 3):
            octave.append(blur(octave[-1], sig[i]))
        gauss.append(octave)
        dog.append([octave[i + 1] - octave[i] for i in range(layers + 2)])
    return gauss, dog
