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

    def _wide(self, img):
        return int(img.dtype == self.torch.float64)

    def down(self, img):
        h, w, c = img.shape
        out = self.torch.empty(((h + 1) // 2, (w + 1) // 2, c), dtype=img.dtype,
                               device=img.device)
        _lib.check(self.lib.pano_pyr_down_image(self.eng.ctx(), _ptr(img), h, w, c,
                                                self._wide(img), _ptr(out)),
                   "pano_pyr_down_image")
        return out

    def up(self, img, like, mode):
        """pyrUp(img)[:h, :w] with (h, w) = like.shape[:2]; mode 1: like - up, 2: like + up."""
        sh, sw, c = img.shape
        oh, ow = like.shape[:2]
        out = self.torch.empty((oh, ow, c), dtype=img.dtype, device=img.device)
        _lib.check(self.lib.pano_pyr_up_image(self.eng.ctx(), _ptr(img), sh, sw, c,
                                              self._wide(img), _ptr(like), mode, _ptr(out), oh,
                                              ow), "pano_pyr_up_image")
        return out

    def reduce_chain(self, img, depth):
        """[img, pyrDown(img), pyrDown^2(img), ...]: depth + 1 images (blend.py:117-122)."""
        chain = [img]
        while len(chain) <= depth:
            chain.append(self.down(chain[-1]))
        return chain

    def detail_chain(self, img, depth):
        """Coarsest image first, then each finer image minus the expanded coarser one
        (blend.py:124-130)."""
        chain = self.reduce_chain(img, depth)
        levels = [chain[depth]]
        for k in range(depth, 0, -1):
            levels.append(self.up(chain[k], chain[k - 1], 1))
        return levels


def _as_f32(eng, img):
    """``img.astype("float32")`` on the device (blend.py:132-133)."""
    import torch
    img = np.ascontiguousarray(img)
    if img.dtype == np.uint8:
        dev = torch.from_numpy(img).to(eng.device)
        out = torch.empty(img.shape, dtype=torch.float32, device=eng.device)
        _lib.check(eng.lib.pano_u8_to_f32(eng.ctx(), _ptr(dev), img.size, _ptr(out)),
                   "pano_u8_to_f32")
        return out
    return torch.from_numpy(img.astype(np.float32)).to(eng.device)


def default_mask(shape):
    """The mask the reference builds when none is given (blend.py:107-111): a logistic
    step across the width, 1 / (1 + exp(-100 u)) with u falling linearly from 1 at the left
    edge to -1 at the right, the same on every row and channel; float64."""
    rows, cols, chans = shape
    ramp = np.linspace(1, -1, cols)
    step = 1.0 / (1 + np.exp(-100 * ramp))
    return np.broadcast_to(step[None, :, None], (rows, cols, chans)).copy()


def laplacian_blending(img1, img2, mask=None, n_levels=6):
    """Use a Laplacian pyramid on the images for blending (blend.py:105-140).

    Same call and result type as the reference.  The mask keeps its float type the way
    NumPy's promotion keeps it there: float64 (the default mask) makes the per-level mix
    and the collapse float64, a float32 mask keeps them float32.  Limit: every pyramid
    level must be at least 2 pixels wide and high (OpenCV's pyrUp of a 1-pixel row is not
    restated here); integer masks are not supported."""
    import torch
    eng = _eng.engine()
    if img1.ndim != 3 or img1.shape != img2.shape or img1.shape[2] > 4:
        raise ValueError("laplacian_blending: two H x W x C images of one shape, C <= 4")
    default = mask is None
    if default:
        # the default mask is one row of float64 repeated: only that row crosses the bus (as
        # a 200 MB host array it was 28 of a 4K blend's 32 ms)
        rows, cols, chans = img1.shape
        mask = default_mask((1, cols, 1))
    if not default and mask.shape[2] == 1:         # blend.py:113-114
        mask = np.repeat(mask, img1.shape[2], axis=2)
    if not np.issubdtype(mask.dtype, np.floating):
        raise NotImplementedError("integer masks take OpenCV's fixed-point pyramids, "
                                  "which this build does not restate")
    smallest = min(img1.shape[:2]) >> (n_levels - 1) if n_levels else 2
    if smallest < 2:
        raise ValueError(f"n_levels={n_levels} leaves a pyramid level narrower than 2 pixels")
    wide = mask.dtype != np.float32                # float16 / float64 -> float64 like NumPy's mix
    mdtype, tdtype = (np.float64, torch.float64) if wide else (np.float32, torch.float32)
    pyr = _Pyr(eng)
    details1 = pyr.detail_chain(_as_f32(eng, img1), n_levels)
    details2 = pyr.detail_chain(_as_f32(eng, img2), n_levels)
    dev_mask = torch.from_numpy(np.ascontiguousarray(mask, dtype=mdtype)).to(eng.device)
    if default:
        dev_mask = dev_mask.expand(rows, cols, chans).contiguous()
    weights = pyr.reduce_chain(dev_mask, n_levels)
    blended = None
    # coarsest level first; the weight pyramid is walked from its coarsest end (blend.py:134-138)
    for first, second, weight in zip(details1, details2, reversed(weights)):
        mixed = torch.empty(first.shape, dtype=tdtype, device=eng.device)
        _lib.check(eng.lib.pano_laplacian_mix(eng.ctx(), _ptr(first), _ptr(second), _ptr(weight),
                                              first.numel(), int(wide), _ptr(mixed)),
                   "pano_laplacian_mix")
        blended = mixed if blended is None else pyr.up(blended, mixed, 2)
    out = torch.empty(blended.shape, dtype=torch.uint8, device=eng.device)
    _lib.check(eng.lib.pano_clip_u8(eng.ctx(), _ptr(blended), blended.numel(), int(wide),
                                    _ptr(out)), "pano_clip_u8")
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
        xtab = eng.to_device(_resize_taps(ow, w, scale))
        ytab = eng.to_device(_resize_taps(oh, h, scale))
    _lib.check(eng.lib.pano_resize_u8(eng.ctx(), _ptr(frame), h, w, c, _ptr(xtab), _ptr(ytab),
                                      _ptr(out), oh, ow), "pano_resize_u8")
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
