"""The Gaussian / pyramid part of the reference ``features.py`` on the GPU.

* ``gaussian_filter(img, sigma=1.0)``  - features.py:20-24
* ``pyr_down`` / ``gaussian_pyramid``   - the ``cv2.pyrDown`` chain of the MSOP
  detector, features.py:138-155
* ``sift_pyramid(img)``                 - the Gaussian and difference-of-Gaussian
  scale space that ``sift_detector`` (features.py:192-201) gets from
  ``cv2.xfeatures2d.SIFT_create().detectAndCompute``.

All filters run in ``libpano360_hip.so`` (``pano_blur_plane``, ``pano_pyr_down``,
``pano_gray_u8``, ``pano_resize_up2``, ``pano_decimate2``, ``pano_subtract``).
The SIFT arithmetic is inside OpenCV, not in the reference repo: its published
algorithm (SIFT defaults: sigma 1.6, 3 layers per octave, first octave -1) is
restated, parity unpinned.  Keypoint detection, description and matching stay
outside this build's scope: ``sift_detector`` is not provided.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import engine as _eng

SIFT_SIGMA = 1.6            # cv2.xfeatures2d.SIFT_create() defaults
SIFT_LAYERS = 3
SIFT_INIT_SIGMA = 0.5


def _to_device(img):
    import torch
    eng = _eng.engine()
    return eng, torch.from_numpy(np.ascontiguousarray(img, np.float32)).to(eng.device)


def gaussian_filter(img, sigma=1.0):
    """Compute the kernel size from sigma and smooth the image
    (features.py:20-24): ksize = max(int((sigma-0.35)/0.15), 1), made odd."""
    ksz = max(int((sigma - 0.35) / 0.15), 1)
    ksz += not ksz % 2
    eng, dev = _to_device(img)
    if dev.ndim == 2:
        return eng.blur_plane(dev, ksz, sigma).cpu().numpy()
    planes = [eng.blur_plane(dev[..., c].contiguous(), ksz, sigma) for c in range(dev.shape[2])]
    import torch
    return torch.stack(planes, dim=-1).cpu().numpy()


def pyr_down(img):
    """``cv2.pyrDown`` of a float32 plane (features.py:155)."""
    eng, dev = _to_device(img)
    return eng.pyr_down(dev).cpu().numpy()


def gaussian_pyramid(img, levels=4):
    """The pyrDown chain the MSOP detector walks (features.py:138-155)."""
    eng, dev = _to_device(img)
    out = [dev]
    for _ in range(levels - 1):
        out.append(eng.pyr_down(out[-1]))
    return [p.cpu().numpy() for p in out]


# ------------------------------------------------------------- SIFT scale space
def sift_sigmas(sigma=SIFT_SIGMA, layers=SIFT_LAYERS):
    """Incremental sigmas of buildGaussianPyramid: sig[0] = sigma,
    sig[i] = sqrt((sigma k^i)^2 - (sigma k^(i-1))^2), k = 2^(1/layers)."""
    k = 2.0 ** (1.0 / layers)
    out = [sigma]
    for i in range(1, layers + 3):
        prev = k ** (i - 1) * sigma
        out.append(float(np.sqrt((prev * k) ** 2 - prev ** 2)))
    return out


def sift_octaves(height, width):
    """nOctaves of SIFT for a base image already doubled in size (first octave
    -1): cvRound(log2(min side of the doubled image) - 2) + 1."""
    return int(np.rint(np.log(float(min(2 * height, 2 * width))) / np.log(2.0) - 2)) + 1


class _Dev:
    """Thin typed wrappers over the pyramid entry points (device tensors)."""

    def __init__(self, eng):
        self.eng, self.lib = eng, eng.lib

    def _new(self, h, w):
        import torch
        return torch.empty((h, w), dtype=torch.float32, device=self.eng.device)

    def gray(self, frame):
        h, w = frame.shape[:2]
        out = self._new(h, w)
        _lib.check(self.lib.pano_gray_u8(_eng._ptr(frame), h, w, _eng._ptr(out),
                                         self.eng.stream()), "pano_gray_u8")
        return out

    def up2(self, plane):
        h, w = plane.shape
        out = self._new(2 * h, 2 * w)
        _lib.check(self.lib.pano_resize_up2(_eng._ptr(plane), h, w, _eng._ptr(out),
                                            self.eng.stream()), "pano_resize_up2")
        return out

    def half(self, plane):
        h, w = plane.shape
        out = self._new(h // 2, w // 2)
        _lib.check(self.lib.pano_decimate2(_eng._ptr(plane), h, w, _eng._ptr(out),
                                           self.eng.stream()), "pano_decimate2")
        return out

    def sub(self, a, b):
        out = self._new(*a.shape)
        _lib.check(self.lib.pano_subtract(_eng._ptr(a), _eng._ptr(b), C.c_size_t(a.numel()),
                                          _eng._ptr(out), self.eng.stream()), "pano_subtract")
        return out

    def blur(self, plane, sigma):
        return self.eng.blur_plane(plane, _eng.gaussian_ksize(sigma), sigma).contiguous()


def sift_pyramid_device(frame, n_octaves=None, sigma=SIFT_SIGMA, layers=SIFT_LAYERS):
    """Gaussian and DoG pyramids of a uint8 BGR frame already on the device.
    Returns (gauss, dog): lists over octaves of lists of float32 planes."""
    dev = _Dev(_eng.engine())
    h, w = frame.shape[:2]
    if n_octaves is None:
        n_octaves = sift_octaves(h, w)
    # createInitialImage: grey -> float -> 2x bilinear -> blur to sigma
    sig_diff = float(np.sqrt(max(np.float32(sigma) ** 2 - np.float32(SIFT_INIT_SIGMA) ** 2 * 4,
                                 np.float32(0.01))))
    base = dev.blur(dev.up2(dev.gray(frame)), sig_diff)
    sig = sift_sigmas(sigma, layers)
    gauss, dog = [], []
    for o in range(n_octaves):
        if o:
            prev = gauss[-1][layers]
            if min(prev.shape) < 2:
                break
            base = dev.half(prev)
        octave = [base]
        for i in range(1, layers + 3):
            octave.append(dev.blur(octave[-1], sig[i]))
        gauss.append(octave)
        dog.append([dev.sub(octave[i + 1], octave[i]) for i in range(layers + 2)])
    return gauss, dog


def sift_pyramid(img, n_octaves=None):
    """Host convenience: uint8 BGR image -> (gauss, dog) as NumPy arrays."""
    eng = _eng.engine()
    frame = eng.upload_frames([img])[0]
    gauss, dog = sift_pyramid_device(frame, n_octaves)
    to_np = lambda pyr: [[p.cpu().numpy() for p in octave] for octave in pyr]   # noqa: E731
    return to_np(gauss), to_np(dog)
