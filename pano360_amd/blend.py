"""``blend.laplacian_blending`` of the reference (blend.py:105-140) on the GPU, and the
8-bit shrink the CLI applies to its inputs (stitcher.py:418-420).

Same call as the reference: ``laplacian_blending(img1, img2, mask=None, n_levels=6)``
with uint8 (or float) ``[H][W][C]`` images and an optional float ``[H][W][1 or C]``
mask; returns uint8 ``[H][W][C]``.  The image pyramids are float32, the mask pyramid
and everything after the per-level mix float64, as NumPy's promotion makes them in the
reference.  All arithmetic runs in ``libpano360_hip.so`` (``pano_pyr_down_image``,
``pano_pyr_up_image``, ``pano_laplacian_mix``, ``pano_clip_u8``, ``pano_resize_u8``);
there is no CPU fallback.  The other experiments of the reference's ``blend.py``
(graph cut, Poisson) are outside the scope.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import engine as _eng


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _Pyr:
    """Pyramid primitives on interleaved device images [h][w][c]."""

    def __init__(self, eng):
        import torch
        self.eng, self.lib, self.torch = eng, eng.lib, torch

    def down(self, img):
        h, w, c = img.shape
        out = self.torch.empty(((h + 1) // 2, (w + 1) // 2, c), dtype=img.dtype,
                               device=img.device)
        _lib.check(self.lib.pano_pyr_down_image(_ptr(img), h, w, c,
                                                int(img.dtype == self.torch.float64),
                                                _ptr(out), self.eng.stream()),
                   "pano_pyr_down_image")
        return out

    def up(self, img, like, mode):
        """pyrUp(img)[:h, :w] with (h, w) = like.shape[:2]; mode 1: like - up, 2: like + up."""
        sh, sw, c = img.shape
        oh, ow = like.shape[:2]
        out = self.torch.empty((oh, ow, c), dtype=img.dtype, device=img.device)
        _lib.check(self.lib.pano_pyr_up_image(_ptr(img), sh, sw, c,
                                              int(img.dtype == self.torch.float64), _ptr(like),
                                              mode, _ptr(out), oh, ow, self.eng.stream()),
                   "pano_pyr_up_image")
        return out

    def gaussian(self, img, n_levels):             # blend.py:117-122
        pyr = [img]
        for _ in range(n_levels):
            pyr.append(self.down(pyr[-1]))
        return pyr

    def laplacian(self, img, n_levels):            # blend.py:124-130
        pyr = self.gaussian(img, n_levels)
        lap = [pyr[-1]]
        for idx in range(n_levels, 0, -1):
            lap.append(self.up(pyr[idx], pyr[idx - 1], 1))
        return lap


def _as_f32(eng, img):
    """``img.astype("float32")`` on the device (blend.py:132-133)."""
    import torch
    img = np.ascontiguousarray(img)
    if img.dtype == np.uint8:
        dev = torch.from_numpy(img).to(eng.device)
        out = torch.empty(img.shape, dtype=torch.float32, device=eng.device)
        _lib.check(eng.lib.pano_u8_to_f32(_ptr(dev), img.size, _ptr(out), eng.stream()),
                   "pano_u8_to_f32")
        return out
    return torch.from_numpy(img.astype(np.float32)).to(eng.device)


def default_mask(shape):
    """The sigmoid ramp of blend.py:107-111 (host: H x W x C float64 of closed form)."""
    hh_, ww_, cc_ = shape
    mask = np.linspace(1, -1, ww_).reshape((1, ww_, 1))
    mask = 1.0 / (1 + np.exp(-100 * mask))
    return np.tile(mask, (hh_, 1, cc_))


def laplacian_blending(img1, img2, mask=None, n_levels=6):
    """Use a Laplacian pyramid on the images for blending (blend.py:105-140)."""
    import torch
    eng = _eng.engine()
    if img1.ndim != 3 or img1.shape != img2.shape or img1.shape[2] > 4:
        raise ValueError("laplacian_blending: two H x W x C images of one shape, C <= 4")
    if mask is None:
        mask = default_mask(img1.shape)
    if mask.shape[2] == 1:                         # blend.py:113-114
        mask = np.repeat(mask, img1.shape[2], axis=2)
    if not np.issubdtype(mask.dtype, np.floating):
        raise NotImplementedError("integer masks take OpenCV's fixed-point pyramids, "
                                  "which this build does not restate")
    smallest = min(img1.shape[:2]) >> (n_levels - 1) if n_levels else 2
    if smallest < 2:
        raise ValueError(f"n_levels={n_levels} leaves a pyramid level narrower than 2 pixels")
    pyr = _Pyr(eng)
    pyr1 = pyr.laplacian(_as_f32(eng, img1), n_levels)
    pyr2 = pyr.laplacian(_as_f32(eng, img2), n_levels)
    gmask = torch.from_numpy(np.ascontiguousarray(mask, dtype=np.float64)).to(eng.device)
    pyrm = pyr.gaussian(gmask, n_levels)[::-1]
    pyrs = []
    for la, lb, gm in zip(pyr1, pyr2, pyrm):       # blend.py:136
        out = torch.empty(la.shape, dtype=torch.float64, device=eng.device)
        _lib.check(eng.lib.pano_laplacian_mix(_ptr(la), _ptr(lb), _ptr(gm), la.numel(),
                                              _ptr(out), eng.stream()), "pano_laplacian_mix")
        pyrs.append(out)
    blended = pyrs[0]
    for ls_ in pyrs[1:]:                           # blend.py:137-138
        blended = pyr.up(blended, ls_, 2)
    out = torch.empty(blended.shape, dtype=torch.uint8, device=eng.device)
    _lib.check(eng.lib.pano_clip_u8(_ptr(blended), blended.numel(), _ptr(out), eng.stream()),
               "pano_clip_u8")
    return out.cpu().numpy()


# ---------------------------------------------------------------- CLI ingest
def _resize_taps(n_out, n_in, scale):
    """(first tap, second tap, 11-bit coefficient of each) per output sample of
    cv2.resize's 8-bit INTER_LINEAR path; float32 coordinates as OpenCV keeps them."""
    f = ((np.arange(n_out) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    f[s < 0] = 0
    s[s < 0] = 0
    f[s >= n_in - 1] = 0
    s[s >= n_in - 1] = n_in - 1
    c0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0))
    c1 = np.rint(f * np.float32(2048.0))
    return np.stack([s, np.minimum(s + 1, n_in - 1), c0, c1], axis=1).astype(np.int32)


def shrink_device(frame, shrink, eng=None):
    """``cv2.resize(im, None, fx=1/shrink, fy=1/shrink)`` (stitcher.py:419-420) of a
    uint8 device image [h][w][c]; returns the shrunk device image."""
    import torch
    eng = eng or _eng.engine()
    h, w, c = frame.shape
    fx = 1 / shrink
    ow, oh = int(np.rint(w * fx)), int(np.rint(h * fx))
    if ow < 1 or oh < 1:
        raise ValueError(f"shrink {shrink} leaves nothing of a {w}x{h} image")
    out = torch.empty((oh, ow, c), dtype=torch.uint8, device=frame.device)
    scale = 1.0 / fx
    if abs(scale - 2.0) < np.finfo(float).eps and w % 2 == 0 and h % 2 == 0:
        xtab = ytab = None                         # exact 2:1: the area path
    else:
        xtab = _eng._to_device(_resize_taps(ow, w, scale), frame.device)
        ytab = _eng._to_device(_resize_taps(oh, h, scale), frame.device)
    _lib.check(eng.lib.pano_resize_u8(_ptr(frame), h, w, c, _ptr(xtab), _ptr(ytab), _ptr(out),
                                      oh, ow, eng.stream()), "pano_resize_u8")
    return out


def shrink_images(imgs, shrink):
    """The resize step of the CLI (stitcher.py:418-420): host uint8 images in, shrunk
    uint8 frames resident on the device out (what ``Engine.stitch`` consumes)."""
    import torch
    eng = _eng.engine()
    frames = [torch.from_numpy(np.ascontiguousarray(im)).to(eng.device) for im in imgs]
    if shrink > 1:
        frames = [shrink_device(f, shrink, eng) for f in frames]
    return frames
