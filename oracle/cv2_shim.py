"""NumPy stand-in for the handful of `cv2` calls the reference hot path makes.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pano360_amd/`` may import this file.

Why it exists: ``/root/reference/stitcher.py`` does ``import cv2`` (line 12) and
``features.py`` calls ``cv2.xfeatures2d.SIFT_create()`` at import time
(features.py:255,286); OpenCV is not installed in this image and its source is
not under ``/root/reference``.  ``oracle/gen_golden.py`` installs this module as
``sys.modules['cv2']`` so the reference's *own* NumPy code can be executed
unchanged and its intermediate / final arrays dumped as golden fixtures.

PARITY UNPINNED at the OpenCV boundary: the reference (Readme.md:22-24) pins no
OpenCV version and ships no golden image, so ``remap``, ``GaussianBlur``,
``cvtColor`` and ``pyrDown`` below restate OpenCV's *documented / published*
algorithms (OpenCV 3.4/4.x ``imgproc``: ``remap`` with ``INTER_BITS=5``
fixed-point coordinates and the float bilinear table; ``getGaussianKernel`` +
``sepFilter2D`` row/column engines; ``borderInterpolate``).  Every assumed
semantic is spelled out next to the code.  Everything the reference computes in
NumPy itself (stitcher.py:73-157, 160-327, 340-369) is exercised for real.

Written vectorised NumPy, float32 arithmetic with one rounding per operation
(no FMA), so that the independent C restatement in ``pano_oracle.c`` can be
checked against it bit for bit.

Third-party cross-checks of this file (tests/test_oracle_golden.py; implementations not
written for this repo): ``remap`` = scipy.ndimage.map_coordinates(order=1, mode='reflect') =
torch's grid_sample; ``GaussianBlur`` and ``pyrDown`` = scipy's correlate1d ('mirror'); round 6:
``resize`` within one level of scipy.ndimage.zoom(order=1, grid_mode=True) and torch's bilinear
interpolate, its 2 : 1 case = torch's avg_pool2d; ``warpPerspective`` = map_coordinates at
M^-1 (x, y, 1); ``pyrUp`` = zero-stuffing + correlate1d([1 4 6 4 1] / 8) away from the borders.
They pin interpolation conventions, apertures, border rules and separability - not OpenCV's
bit-level rounding, which stays as documented here.
"""
import numpy as np

# constants the reference names (stitcher.py:56-57,259,315-316; features.py)
INTER_NEAREST = 0
INTER_LINEAR = 1
BORDER_CONSTANT = 0
BORDER_REPLICATE = 1
BORDER_REFLECT = 2
BORDER_WRAP = 3
BORDER_REFLECT_101 = 4
BORDER_DEFAULT = 4
BORDER_TRANSPARENT = 5
COLOR_RGB2RGBA = 0
COLOR_BGR2GRAY = 6
CV_32F = 5
RANSAC = 8

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS

_I32_MIN = np.int64(-2 ** 31)


def border_interpolate(p, length, border):
    """OpenCV ``borderInterpolate`` for REFLECT / REFLECT_101 (any distance).

    REFLECT:      ... c b a | a b c ... z | z y x ...   (period 2*len)
    REFLECT_101:  ... c b | a b c ... z | y x ...       (period 2*len-2)
    OpenCV iterates single reflections until the index is in range; that is
    the periodic extension used here.  ``len == 1`` returns 0.
    """
    p = np.asarray(p, dtype=np.int64)
    if length == 1:
        return np.zeros_like(p)
    if border == BORDER_REFLECT:
        per = 2 * length
        m = np.mod(p, per)
        return np.where(m < length, m, per - 1 - m)
    if border == BORDER_REFLECT_101:
        per = 2 * length - 2
        m = np.mod(p, per)
        return np.where(m < length, m, per - m)
    raise NotImplementedError(border)


def cv_round_f32(val):
    """``cvRound(float)`` on x86: cvtss2si, round-half-even, and the "integer
    indefinite" 0x80000000 for NaN / out-of-range inputs."""
    val = np.asarray(val, dtype=np.float32)
    with np.errstate(invalid="ignore"):
        r = np.rint(val.astype(np.float64))
    ok = np.isfinite(r) & (r >= -2.0 ** 31) & (r < 2.0 ** 31)
    out = np.where(ok, r, 0.0).astype(np.int64)
    return np.where(ok, out, _I32_MIN)


def cvtColor(img, code):
    """Only RGB->RGBA on float images is needed (stitcher.py:259): alpha = 1."""
    if code != COLOR_RGB2RGBA:
        raise NotImplementedError(code)
    out = np.empty(img.shape[:2] + (4,), dtype=img.dtype)
    out[..., :3] = img
    out[..., 3] = 1
    return out


def remap(src, map1, map2, interpolation, borderMode=BORDER_CONSTANT):
    """``cv2.remap`` for float32 maps, INTER_LINEAR, BORDER_REFLECT
    (call site stitcher.py:315-316).

    Assumed OpenCV semantics:
      * coordinates are converted to fixed point: ``s = cvRound(v * 32)``
        (float32 multiply), integer part ``s >> 5`` saturated to int16,
        fraction ``s & 31``;
      * weights come from the float bilinear table: products of
        ``(1 - f/32)`` and ``f/32`` rounded to float32 (they are exact);
      * ``dst = v00*w00 + v01*w01 + v10*w10 + v11*w11`` evaluated left to
        right in float32, one rounding per operation;
      * out-of-range taps go through ``borderInterpolate`` per tap.
    """
    if interpolation != INTER_LINEAR or borderMode != BORDER_REFLECT:
        raise NotImplementedError((interpolation, borderMode))
    src = np.ascontiguousarray(src, dtype=np.float32)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[..., None]
    sh, sw = src.shape[:2]
    f32 = np.float32
    sx = cv_round_f32(np.asarray(map1, f32) * f32(INTER_TAB_SIZE))
    sy = cv_round_f32(np.asarray(map2, f32) * f32(INTER_TAB_SIZE))
    fx = (sx & (INTER_TAB_SIZE - 1)).astype(np.int64)
    fy = (sy & (INTER_TAB_SIZE - 1)).astype(np.int64)
    ix = np.clip(sx >> INTER_BITS, -32768, 32767)
    iy = np.clip(sy >> INTER_BITS, -32768, 32767)

    x0 = border_interpolate(ix, sw, borderMode)
    x1 = border_interpolate(ix + 1, sw, borderMode)
    y0 = border_interpolate(iy, sh, borderMode)
    y1 = border_interpolate(iy + 1, sh, borderMode)

    one = f32(1.0)
    ax = fx.astype(f32) * f32(1.0 / INTER_TAB_SIZE)
    ay = fy.astype(f32) * f32(1.0 / INTER_TAB_SIZE)
    w00 = ((one - ay) * (one - ax))[..., None]
    w01 = ((one - ay) * ax)[..., None]
    w10 = (ay * (one - ax))[..., None]
    w11 = (ay * ax)[..., None]

    acc = src[y0, x0] * w00
    acc = acc + src[y0, x1] * w01
    acc = acc + src[y1, x0] * w10
    acc = acc + src[y1, x1] * w11
    return acc[..., 0] if squeeze else acc


def invert3x3(m):
    """``cv::invert`` on a 3x3 CV_64F matrix: adjugate times 1/det, one rounding
    per operation (OpenCV special-cases small matrices instead of running LU)."""
    m = np.asarray(m, dtype=np.float64)
    det = (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1])
           - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0])
           + m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))
    if det == 0.0:
        return np.zeros((3, 3))
    d = 1.0 / det
    return np.array([
        [(m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) * d, (m[0, 2] * m[2, 1] - m[0, 1] * m[2, 2]) * d,
         (m[0, 1] * m[1, 2] - m[0, 2] * m[1, 1]) * d],
        [(m[1, 2] * m[2, 0] - m[1, 0] * m[2, 2]) * d, (m[0, 0] * m[2, 2] - m[0, 2] * m[2, 0]) * d,
         (m[0, 2] * m[1, 0] - m[0, 0] * m[1, 2]) * d],
        [(m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]) * d, (m[0, 1] * m[2, 0] - m[0, 0] * m[2, 1]) * d,
         (m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]) * d]])


def warp_block_width(width, height):
    """Column-block width of ``WarpPerspectiveInvoker`` (BLOCK_SZ = 32): the
    x coordinate enters as block start + offset, which fixes the rounding of
    the double-precision numerators."""
    bh0 = min(16, height)
    return min(1024 // bh0, width)


def warpPerspective(src, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT):
    """``cv2.warpPerspective`` for a float32 multi-channel source, INTER_LINEAR,
    BORDER_TRANSPARENT, no WARP_INVERSE_MAP (call site stitcher.py:56-57).

    Assumed OpenCV semantics:
      * ``M`` (src -> dst) is inverted with ``cv::invert`` (adjugate / det);
      * per destination pixel, in double, with x = block start + x1:
        ``W = 32 / (W0 + M6*x1)`` (0 when the denominator is 0),
        ``fX = (X0 + M0*x1) * W`` clamped to the int range, ``X = cvRound(fX)``;
        integer part ``X >> 5`` saturated to int16, fraction ``X & 31``;
      * ``remap`` with those fixed-point maps: a pixel is written only when all
        four taps are inside the source (``0 <= sx < w-1``, ``0 <= sy < h-1``;
        the BORDER_TRANSPARENT shortcut of ``remapBilinear`` for cn != 3), as
        ``v00*w00 + v01*w01 + v10*w10 + v11*w11`` left to right in float32;
      * pixels not written keep the destination's previous content.  The Python
        binding hands OpenCV a freshly allocated array; the reference reads its
        alpha channel as "0 = not written" (stitcher.py:58), so the
        destination is taken to start as zeros.
    """
    if flags != INTER_LINEAR or borderMode != BORDER_TRANSPARENT:
        raise NotImplementedError((flags, borderMode))
    src = np.ascontiguousarray(src, dtype=np.float32)
    sh, sw = src.shape[:2]
    width, height = dsize
    m = invert3x3(M).ravel()
    bw0 = warp_block_width(width, height)

    col = np.arange(width, dtype=np.int64)
    xb = ((col // bw0) * bw0).astype(np.float64)[None, :]
    x1 = (col % bw0).astype(np.float64)[None, :]
    yy = np.arange(height, dtype=np.float64)[:, None]
    x0_ = m[0] * xb + m[1] * yy + m[2]
    y0_ = m[3] * xb + m[4] * yy + m[5]
    w0_ = m[6] * xb + m[7] * yy + m[8]
    den = w0_ + m[6] * x1
    with np.errstate(divide="ignore", invalid="ignore"):
        wgt = np.where(den != 0.0, float(INTER_TAB_SIZE) / den, 0.0)
    lo, hi = float(-2 ** 31), float(2 ** 31 - 1)
    fx_ = np.maximum(lo, np.minimum(hi, (x0_ + m[0] * x1) * wgt))
    fy_ = np.maximum(lo, np.minimum(hi, (y0_ + m[3] * x1) * wgt))
    big_x = np.rint(fx_).astype(np.int64)
    big_y = np.rint(fy_).astype(np.int64)
    sx = np.clip(big_x >> INTER_BITS, -32768, 32767)
    sy = np.clip(big_y >> INTER_BITS, -32768, 32767)
    fx = big_x & (INTER_TAB_SIZE - 1)
    fy = big_y & (INTER_TAB_SIZE - 1)

    inside = (sx >= 0) & (sx < max(sw - 1, 0)) & (sy >= 0) & (sy < max(sh - 1, 0))
    sx = np.where(inside, sx, 0)
    sy = np.where(inside, sy, 0)
    f32 = np.float32
    one = f32(1.0)
    ax = fx.astype(f32) * f32(1.0 / INTER_TAB_SIZE)
    ay = fy.astype(f32) * f32(1.0 / INTER_TAB_SIZE)
    w00 = ((one - ay) * (one - ax))[..., None]
    w01 = ((one - ay) * ax)[..., None]
    w10 = (ay * (one - ax))[..., None]
    w11 = (ay * ax)[..., None]
    x1i = np.minimum(sx + 1, sw - 1)
    y1i = np.minimum(sy + 1, sh - 1)
    acc = src[sy, sx] * w00
    acc = acc + src[sy, x1i] * w01
    acc = acc + src[y1i, sx] * w10
    acc = acc + src[y1i, x1i] * w11
    return np.where(inside[..., None], acc, f32(0.0)).astype(f32)


def gaussian_ksize(sigma, depth_is_8u=False):
    """Automatic aperture of ``GaussianBlur(ksize=(0,0))``:
    ``cvRound(sigma * (3 if 8-bit else 4) * 2 + 1) | 1``."""
    return int(np.rint(sigma * (3 if depth_is_8u else 4) * 2 + 1)) | 1


def getGaussianKernel(ksize, sigma, ktype=CV_32F):
    """``cv::getGaussianKernel`` for a float32 kernel: taps exp(-x^2/2s^2)
    evaluated in double, stored as float32, summed in double, then each tap
    multiplied by 1/sum in double and stored as float32."""
    if sigma <= 0:
        sigma = ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    scale2x = -0.5 / (sigma * sigma)
    taps = np.exp(scale2x * x * x).astype(np.float32)
    total = 0.0
    for t in taps:          # sequential double sum, as the C loop does
        total += float(t)
    inv = 1.0 / total
    return (taps.astype(np.float64) * inv).astype(np.float32)


def sep_filter_symm(src, taps, border=BORDER_REFLECT_101):
    """Separable symmetric filter, float32, the two engines of sepFilter2D:

    row pass   : s = k[0]*x[0]; s += k[j]*x[j]  (j ascending)       -> buffer
    column pass: s = k[c]*y[c]; s += k[c+j]*(y[c+j] + y[c-j])  (j = 1..r)
    """
    src = np.ascontiguousarray(src, dtype=np.float32)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[..., None]
    h, w = src.shape[:2]
    n = len(taps)
    r = n // 2
    taps = np.asarray(taps, np.float32)

    cols = border_interpolate(np.arange(-r, w + r), w, border)
    padded = src[:, cols]
    acc = padded[:, 0:w] * taps[0]
    for j in range(1, n):
        acc = acc + padded[:, j:j + w] * taps[j]

    rows = border_interpolate(np.arange(-r, h + r), h, border)
    padded = acc[rows]
    out = padded[r:r + h] * taps[r]
    for j in range(1, r + 1):
        out = out + (padded[r + j:r + j + h] + padded[r - j:r - j + h]) * taps[r + j]
    return out[..., 0] if squeeze else out


def GaussianBlur(src, ksize, sigmaX, sigmaY=0, borderType=BORDER_DEFAULT):
    """``cv2.GaussianBlur`` on float32 images (call sites stitcher.py:226,
    features.py:24).  ksize (0,0) -> automatic aperture (4 sigma for float)."""
    kx, ky = ksize
    if sigmaY <= 0:
        sigmaY = sigmaX
    if kx <= 0 and sigmaX > 0:
        kx = gaussian_ksize(sigmaX)
    if ky <= 0 and sigmaY > 0:
        ky = gaussian_ksize(sigmaY)
    if kx != ky or sigmaX != sigmaY:
        raise NotImplementedError("anisotropic blur is never requested")
    return sep_filter_symm(src, getGaussianKernel(kx, sigmaX), borderType)


def _pyr_dtype(src):
    """pyrDown / pyrUp keep CV_32F and CV_64F (working type = the same type)."""
    src = np.asarray(src)
    if src.dtype not in (np.float32, np.float64):
        raise NotImplementedError(f"pyramids of {src.dtype} are never requested by the path")
    return np.ascontiguousarray(src), src.dtype.type


def pyrDown(src):
    """``cv2.pyrDown``: 5x5 separable [1 4 6 4 1]/16 (x1/256 overall),
    REFLECT_101, keep even rows/cols, output ((w+1)//2, (h+1)//2).
    Float path assumed: ``c*6 + (l1+r1)*4 + l2 + r2`` (left to right) along
    rows, the same along columns, scaled by 1/256 at the end.  float32 and
    float64 images keep their type (blend.py:113-117 feeds it a float64 mask)."""
    src, ft = _pyr_dtype(src)
    h, w = src.shape[:2]
    oh, ow = (h + 1) // 2, (w + 1) // 2
    cx = border_interpolate(np.arange(-2, 2 * ow + 2), w, BORDER_REFLECT_101)
    p = src[:, cx]
    c = np.arange(ow) * 2
    row = (p[:, c + 2] * ft(6) + (p[:, c + 1] + p[:, c + 3]) * ft(4)
           + p[:, c] + p[:, c + 4])
    ry = border_interpolate(np.arange(-2, 2 * oh + 2), h, BORDER_REFLECT_101)
    q = row[ry]
    r_ = np.arange(oh) * 2
    out = (q[r_ + 2] * ft(6) + (q[r_ + 1] + q[r_ + 3]) * ft(4)
           + q[r_] + q[r_ + 4])
    return out * ft(1.0 / 256.0)


def pyrUp(src):
    """``cv2.pyrUp`` to the default size (2w, 2h): zero-stuffed source filtered with
    [1 4 6 4 1]/8 per axis (x1/64 overall).  Per axis, from source samples s:
    even outputs ``s[i-1] + s[i]*6 + s[i+1]``, odd outputs ``(s[i] + s[i+1])*4``;
    at the first sample ``s[-1] := s[1]`` (REFLECT_101), past the last one
    ``s[n] := s[n-1]``.  Rows first (OpenCV's pyrUp_ writes the two edge pairs of a row
    in closed form: ``s[0]*6 + s[1]*2``, ``s[n-2] + s[n-1]*7``, ``s[n-1]*8``), then columns
    (three buffered rows chosen by borderInterpolate), scaled by 1/64 at the end."""
    src, ft = _pyr_dtype(src)
    h, w = src.shape[:2]
    if h < 2 or w < 2:
        raise NotImplementedError("pyrUp of a one-pixel-wide image is never requested")

    def up(a, axis, edge_forms):
        a = np.moveaxis(a, axis, 0)
        n = a.shape[0]
        prev = a[np.r_[1, 0:n - 1]]              # s[-1] := s[1]
        nxt = a[np.r_[1:n, n - 1]]               # s[n] := s[n-1]
        even = prev + a * ft(6) + nxt
        odd = (a + nxt) * ft(4)
        if edge_forms:
            # the row pass writes its first and last sample pairs in closed form; the
            # column pass (rows picked by borderInterpolate) keeps the three-term sum
            even[0] = a[0] * ft(6) + a[1] * ft(2)
            even[n - 1] = a[n - 2] + a[n - 1] * ft(7)
            odd[n - 1] = a[n - 1] * ft(8)
        out = np.empty((2 * n,) + a.shape[1:], dtype=a.dtype)
        out[0::2] = even
        out[1::2] = odd
        return np.moveaxis(out, 0, axis)

    return up(up(src, 1, True), 0, False) * ft(1.0 / 64.0)


INTER_AREA = 3


def resize(src, dsize, fx=0.0, fy=0.0, interpolation=INTER_LINEAR):
    """``cv2.resize`` of an 8-bit image by scale factors (stitcher.py:419-420 calls it
    with ``dsize=None, fx=fy=1/shrink``), INTER_LINEAR.  OpenCV's 8-bit path, restated:

    * output size ``cvRound(w * fx) x cvRound(h * fy)``; source coordinate of output d is
      ``float((d + 0.5) / fx - 0.5)``, split into floor and fraction, clamped to the first /
      last sample with fraction 0;
    * coefficients in 11-bit fixed point (``cvRound(c * 2048)`` as int16), horizontal pass
      ``s0*a0 + s1*a1`` in int32, vertical pass
      ``((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2``;
    * an exact 2:1 reduction is taken by the area path instead (``resize`` switches
      INTER_LINEAR to INTER_AREA when both scales are exactly 2): rounded 2 x 2 box means.
    """
    src = np.ascontiguousarray(src)
    if src.dtype != np.uint8 or dsize is not None or interpolation != INTER_LINEAR:
        raise NotImplementedError("only the call of stitcher.py:419 is restated")
    h, w = src.shape[:2]
    ow, oh = int(np.rint(w * fx)), int(np.rint(h * fy))
    sx_, sy_ = 1.0 / fx, 1.0 / fy
    if abs(sx_ - 2.0) < np.finfo(float).eps and abs(sy_ - 2.0) < np.finfo(float).eps \
            and w % 2 == 0 and h % 2 == 0:
        a = src.astype(np.int32)
        box = a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2]
        return ((box + 2) >> 2).astype(np.uint8)

    def taps(n_out, n_in, scale):
        f = ((np.arange(n_out) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        f[s < 0] = 0
        s[s < 0] = 0
        f[s >= n_in - 1] = 0
        s[s >= n_in - 1] = n_in - 1
        c0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int32)
        c1 = np.rint(f * np.float32(2048.0)).astype(np.int32)
        return s, np.minimum(s + 1, n_in - 1), c0, c1

    x0, x1, a0, a1 = taps(ow, w, sx_)
    y0, y1, b0, b1 = taps(oh, h, sy_)
    a = src.astype(np.int32)
    shape = (1, ow) + (1,) * (a.ndim - 2)
    rows = a[:, x0] * a0.reshape(shape) + a[:, x1] * a1.reshape(shape)
    shape = (oh,) + (1,) * (a.ndim - 1)
    out = (((b0.reshape(shape) * (rows[y0] >> 4)) >> 16)
           + ((b1.reshape(shape) * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


class _NoSift:
    """``features.py:255,286`` evaluates ``sift_detector()`` as a default
    argument at import; the object only has to exist."""

    def detectAndCompute(self, img, mask):
        raise NotImplementedError("SIFT lives inside OpenCV; not restated here")


class xfeatures2d:  # noqa: N801  (mirrors the cv2 attribute name)
    @staticmethod
    def SIFT_create(*args, **kwargs):
        return _NoSift()


def install():
    """Register this module as ``cv2`` (used by gen_golden.py only)."""
    import sys
    sys.modules["cv2"] = sys.modules[__name__]
