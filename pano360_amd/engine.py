"""Device-resident warp/blend engine: host geometry + launches over the C ABI.

Everything per-pixel runs in ``libpano360_hip.so``; this module only plans
(float64 scalar geometry, as the reference does on the host) and sequences the
launches.  torch supplies device memory and the stream, nothing else.

Reference lines mirrored by the planning code: stitcher.py:107-157 (ranges,
resolution), :283-302 (mosaic shape, patch rectangles, angle grids), :218
(level sigmas); OpenCV's getGaussianKernel for the taps (host, 33..97 floats).
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import Patch

BORDER_SAMPLES = 100        # stitcher.py:109
MULTIBAND_PAD = 10          # stitcher.py:296-297


# --------------------------------------------------------------- projections
class SphProj:
    """Spherical projection pair (reference stitcher.py:73-87)."""

    @staticmethod
    def hom2proj(pts):
        pts = np.asarray(pts)
        xz = np.sqrt(pts[:, 0] ** 2 + pts[:, 2] ** 2)
        theta = np.arctan2(pts[:, 0], pts[:, 2])
        phi = np.arctan2(pts[:, 1], xz)
        return np.stack([theta, phi], axis=-1)

    @staticmethod
    def proj2hom(pts):
        pts = np.asarray(pts)
        return np.stack([np.sin(pts[:, 0]), np.tan(pts[:, 1]),
                         np.cos(pts[:, 0])], axis=-1)


class CylProj:
    """Cylindrical projection pair (reference stitcher.py:90-104)."""

    @staticmethod
    def hom2proj(pts):
        pts = np.asarray(pts)
        xz = np.sqrt(pts[:, 0] ** 2 + pts[:, 2] ** 2)
        return np.stack([np.arctan2(pts[:, 0], pts[:, 2]), pts[:, 1] / xz],
                        axis=-1)

    @staticmethod
    def proj2hom(pts):
        pts = np.asarray(pts)
        return np.stack([np.sin(pts[:, 0]), pts[:, 1], np.cos(pts[:, 0])],
                        axis=-1)


def hat(size):
    """Triangular weight 0 .. 0.5 .. 1/size (reference stitcher.py:251-254)."""
    centred = np.arange(size) - size / 2
    return 0.5 - np.abs(centred / size)


_RINGS = {}


def border_ring(shape):
    """The 4x100 centred border points of a frame, transposed to [3][400]
    (reference stitcher.py:109-119); depends on the frame size only."""
    if shape not in _RINGS:
        height, width = shape
        ticks_x = np.linspace(0, width, BORDER_SAMPLES)
        ticks_y = np.linspace(0, height, BORDER_SAMPLES)
        ones = np.ones(BORDER_SAMPLES)
        left = np.stack([0 * ones, ticks_y, ones], axis=1)
        right = np.stack([width * ones, ticks_y, ones], axis=1)
        top = np.stack([ticks_x, 0 * ones, ones], axis=1)
        bottom = np.stack([ticks_x, height * ones, ones], axis=1)
        ring = np.concatenate([left, right, top, bottom])
        ring = ring - np.array([width / 2, height / 2, 0])
        _RINGS[shape] = np.ascontiguousarray(ring.T)
    return _RINGS[shape]


def ranges_from_border(shapes, homs):
    """range_from_border for many frames at once: the 3x3 products stay one BLAS
    call per frame (same numbers as the reference's), the arctangents and the
    min / max run over all frames together."""
    low, high = range_arrays_from_border(shapes, homs)
    return [(low[i], high[i]) for i in range(len(shapes))]


def range_arrays_from_border(shapes, homs):
    """The same as two arrays [n][2] = (theta, phi) minima / maxima.  Frames of one size (the
    usual case) take one stacked matmul: bit for bit the per-frame ``hom.dot(ring)`` of the
    reference on every platform tried (tests/test_host_abi.py compares them), at a tenth of
    the call overhead."""
    if len(set(shapes)) == 1:
        pts = np.matmul(np.asarray(homs, np.float64), border_ring(shapes[0]))  # [n][3][400]
    else:
        pts = np.stack([h.dot(border_ring(s)) for s, h in zip(shapes, homs)])
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    theta = np.arctan2(x, z)
    phi = np.arctan2(y, np.sqrt(x ** 2 + z ** 2))
    low = np.stack([theta.min(axis=1), phi.min(axis=1)], axis=1)
    high = np.stack([theta.max(axis=1), phi.max(axis=1)], axis=1)
    return low, high


def range_from_border(shape, hom):
    """Angular bounding box of a frame from 4x100 border points
    (reference stitcher.py:107-122; no wrap-around handling there either)."""
    return ranges_from_border([tuple(shape)], [hom])[0]


def range_from_corners(shape, hom):
    """Angular box from the 4 corners with the +2pi / +pi wrap fixes
    (reference stitcher.py:125-139)."""
    height, width = shape
    half_w, half_h = width / 2, height / 2
    corners = np.array([[-half_w, -half_h, 1], [half_w, -half_h, 1],
                        [-half_w, half_h, 1], [half_w, half_h, 1]])
    ang = SphProj.hom2proj(hom.dot(corners.T).T)
    xmin, xmax = min(ang[0, 0], ang[2, 0]), max(ang[1, 0], ang[3, 0])
    ymin, ymax = min(ang[0, 1], ang[1, 1]), max(ang[2, 1], ang[3, 1])
    if xmin > xmax:
        xmax += 2 * np.pi
    if ymin > ymax:
        ymax += np.pi
    return np.array([xmin, ymin]), np.array([xmax, ymax])


def resolution_for(ranges, mid_shape, mid_hom, max_resolution):
    """rad/px of the central frame, capped so the long mosaic side is at most
    ``max_resolution`` pixels (reference stitcher.py:142-157)."""
    lows, highs = ranges if isinstance(ranges, tuple) else zip(*ranges)   # two [n][2] arrays, or pairs
    low, high = np.min(lows, axis=0), np.max(highs, axis=0)
    c_low, c_high = range_from_corners(mid_shape, mid_hom)
    res = (c_high - c_low) / np.array(mid_shape[::-1])
    longest = np.max((high - low) / res)
    if longest > max_resolution:
        res *= longest / max_resolution
    return res, (low, high)


def gaussian_ksize(sigma):
    """Aperture cv2.GaussianBlur picks for ksize=(0,0) on float images."""
    return int(np.rint(sigma * 4 * 2 + 1)) | 1


def gaussian_taps(ksize, sigma):
    """cv::getGaussianKernel as float32: exp in double, float32 taps, double
    sum of the rounded taps, rescale in double, round again."""
    if sigma <= 0:
        sigma = ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
    offs = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    raw = np.exp(offs * offs * (-0.5 / (sigma * sigma))).astype(np.float32)
    total = 0.0
    for tap in raw:
        total += float(tap)
    return (raw.astype(np.float64) * (1.0 / total)).astype(np.float32)


def padded_taps(taps, extra=0):
    """Tap table layout of include/pano360.h: TAP_LEAD + extra zeros, taps, zeros
    (extra = (R - r) & 3 when several levels share one staged row tile)."""
    out = np.zeros(len(taps) + _lib.TAP_PAD, np.float32)
    lead = _lib.TAP_LEAD + extra
    out[lead:lead + len(taps)] = taps
    return out


def level_sigmas(n_levels):
    """sigma_k = 4*sqrt(2k+1) for the n_levels-1 blurred levels (:218)."""
    return [float(np.sqrt(2 * k + 1.0) * 4) for k in range(n_levels - 1)]


class Plan:
    """Host-side geometry of one stitch: everything stitcher.py:276-302
    derives before a pixel is touched."""

    def __init__(self, shapes, rots, intrs, padded, max_resolution, table_cols=None):
        """``table_cols`` = (a, b): the sin / cos tables are evaluated for mosaic columns [a, b) only
        (one GPU's strip and its halo; the rest is NaN and must not be read) - at eight ranks
        the two full tables are a fifth of a stitch's host time."""
        self.shapes = [tuple(int(v) for v in s) for s in shapes]
        self.n = len(self.shapes)
        # one LAPACK call per matrix either way: the stacked inverse has the same bits as
        # np.linalg.inv per camera (bundle_adj.py:28-29) at a thirtieth of the call overhead
        kinv = np.linalg.inv(np.asarray(intrs, np.float64).reshape(-1, 3, 3))
        self.homs = [np.asarray(r).T.dot(ki) for r, ki in zip(rots, kinv)]
        self.projs = [np.ascontiguousarray(np.asarray(k).dot(r), np.float64)
                      for r, k in zip(rots, intrs)]
        lows, highs = range_arrays_from_border(self.shapes, self.homs)
        self.ranges = [(lows[i], highs[i]) for i in range(self.n)]
        mid = self.n // 2
        self.resolution, (self.low, self.high) = resolution_for(
            (lows, highs), self.shapes[mid], self.homs[mid], max_resolution)
        target = (self.high - self.low) / self.resolution
        self.shape = tuple(int(v) for v in np.round(target))[::-1]      # (H, W)
        limit = target.astype(np.int32)
        first = np.round((lows - self.low) / self.resolution).astype(np.int32)    # [n][x, y]
        last = np.round((highs - self.low) / self.resolution).astype(np.int32)
        if padded:
            first = np.maximum(first - MULTIBAND_PAD, np.int32(0))
            last = np.minimum(last + MULTIBAND_PAD, limit)
        self.rects = [(int(f[1]), int(t[1]), int(f[0]), int(t[0]))       # (y0, y1, x0, x1)
                      for f, t in zip(first, last)]
        if np.any(last <= first):
            raise ValueError("a frame projects to an empty patch")
        cols = max(self.shape[1], max(r[3] for r in self.rects))
        rows = max(self.shape[0], max(r[1] for r in self.rects))
        theta = np.arange(cols, dtype=np.int64) * self.resolution[0] + self.low[0]
        phi = np.arange(rows, dtype=np.int64) * self.resolution[1] + self.low[1]
        if table_cols is None:
            self.sin_t, self.cos_t = np.sin(theta), np.cos(theta)
        else:
            a, b = max(int(table_cols[0]), 0), min(int(table_cols[1]), len(theta))
            self.sin_t = np.full(len(theta), np.nan)
            self.cos_t = np.full(len(theta), np.nan)
            # the same NumPy loops on a slice: element for element the full tables' values
            # (tests/test_host_abi.py checks that on this host)
            self.sin_t[a:b] = np.sin(theta[a:b])
            self.cos_t[a:b] = np.cos(theta[a:b])
        self.tan_p = np.tan(phi)

    @property
    def patch_pixels(self):
        return sum((y1 - y0) * (x1 - x0) for y0, y1, x0, x1 in self.rects)


class PlanMemo:
    """Content-keyed memo of ``Plan`` objects.  A plan is a pure function of the frame shapes,
    the rotations and calibrations (every bit of every float64 entry), the padding flag, the
    resolution cap and the trig-table columns - the key holds exactly those, as bytes, so a
    hit hands back what ``Plan(...)`` would compute again (stitcher.py:276-302 does, per
    stitch) and a camera that moved by one unit in the last place misses.  Host-only: the
    engine stores uploaded plans in one (``Engine.cached_plan``)."""

    def __init__(self, capacity=8):
        self.capacity, self.plans, self.hits, self.misses = int(capacity), {}, 0, 0

    @staticmethod
    def key(shapes, rots, intrs, padded, max_resolution, table_cols=None):
        return (tuple(tuple(int(v) for v in sh) for sh in shapes),
                np.ascontiguousarray(rots, np.float64).tobytes(),
                np.ascontiguousarray(intrs, np.float64).tobytes(),
                bool(padded), float(max_resolution),
                None if table_cols is None else (int(table_cols[0]), int(table_cols[1])))

    def get(self, shapes, rots, intrs, padded, max_resolution, table_cols=None, make=None):
        key = self.key(shapes, rots, intrs, padded, max_resolution, table_cols)
        plan = self.plans.get(key)
        if plan is not None:
            self.hits += 1
            return plan
        self.misses += 1
        if len(self.plans) >= self.capacity:
            self.plans.pop(next(iter(self.plans)))          # the oldest entry
        make = make or Plan
        plan = self.plans[key] = make(shapes, rots, intrs, padded, max_resolution, table_cols)
        return plan

    def __len__(self):
        return len(self.plans)


# ------------------------------------------------------------------ exposure
def find_gains(overlaps, sizes, stdn=0.1, stdg=2):
    """Gains minimising the mean-intensity discrepancies on the overlaps
    (stitcher.py:24-33, eq. (29) of Brown & Lowe): N x N normal equations, host
    float64, ``np.linalg.solve`` like the reference."""
    pair_w = (sizes + sizes.T) / (stdn * stdn)
    prior_w = sizes / (stdg * stdg)
    lhs = np.diag(np.sum(pair_w * overlaps * overlaps + prior_w, axis=1))
    lhs -= pair_w * overlaps * overlaps.T
    return np.linalg.solve(lhs, np.sum(prior_w, axis=1))


def invert3x3(m):
    """``cv::invert`` on 3 x 3 doubles (what cv2.warpPerspective applies to its
    matrix): adjugate times 1/det, one rounding per operation."""
    a, b, c, d, e, f, g, h, i = np.asarray(m, np.float64).ravel()
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    if det == 0.0:
        return np.zeros((3, 3))
    r = 1.0 / det
    return np.array([[(e * i - f * h) * r, (c * h - b * i) * r, (b * f - c * e) * r],
                     [(f * g - d * i) * r, (a * i - c * g) * r, (c * d - a * f) * r],
                     [(d * h - e * g) * r, (b * g - a * h) * r, (a * e - b * d) * r]])


def overlap_pairs(rots, intrs, width, height):
    """The (i, j) pairs equalize_gains samples (stitcher.py:44-55) as a
    ``pano_pair`` table: pixel homography j -> i = T (K_i R_i)(R_j^T K_j^-1) T^-1
    with the reference's association of the products (stitcher.py:48,
    bundle_adj.py:36-38), pairs with a corner of frame j behind frame i
    dropped (:51-52), inverted as OpenCV inverts it."""
    shift = np.array([[1, 0, width / 2], [0, 1, height / 2], [0, 0, 1]])
    unshift = np.array([[1, 0, -width / 2], [0, 1, -height / 2], [0, 0, 1]])
    corners = np.array([[0, 0, 1], [width, 0, 1], [width, height, 1], [0, height, 1]])
    fwd = [intr.dot(rot) for rot, intr in zip(rots, intrs)]
    back = [rot.T.dot(np.linalg.inv(intr)) for rot, intr in zip(rots, intrs)]
    out = []
    for i in range(len(rots)):
        for j in range(i + 1, len(rots)):
            hom = shift.dot(fwd[i].dot(back[j])).dot(unshift)
            if np.any(hom.dot(corners.T).T[:, 2] < 0):
                continue
            out.append((invert3x3(hom).ravel(), i, j))
    table = np.zeros(len(out), dtype=PAIR_DTYPE)
    for k, rec in enumerate(out):
        table[k] = rec
    return table


def gain_tables(gains):
    """Per-camera colour tables of the equalised frames: the reference's
    ``np.clip(gain * img, 0, 1)`` stored back into the float32 image
    (stitcher.py:66) applied to the 256 values float32(u8)/255 can take."""
    base = np.arange(256, dtype=np.float32) / np.float32(255)
    luts = np.empty((len(gains), 256), np.float32)
    for k, gain in enumerate(gains):
        luts[k] = np.clip(gain * base, 0, 1)
    return luts


# ------------------------------------------------------------------- device
def _torch():
    import torch
    return torch


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# numpy mirrors of the C records (include/pano360.h), for building tables in bulk
PATCH_DTYPE = np.dtype([(k, "<u8") for k in ("planes", "mask", "blurred", "scratch")]
                       + [(k, "<i4") for k in ("y0", "x0", "h", "w", "vy0", "vx0", "vh", "vw",
                                               "ay0", "ax0", "ah", "aw", "vpitch", "apitch",
                                               "index", "tiles_off")])
CAMERA_DTYPE = np.dtype([("proj", "<f8", (9,)), ("frame", "<u8"), ("hat_x", "<u8"),
                         ("hat_y", "<u8")]
                        + [(k, "<i4") for k in ("sh", "sw", "y0", "x0", "h", "w")])
PAIR_DTYPE = np.dtype([("minv", "<f8", (9,)), ("i", "<i4"), ("j", "<i4")])
assert PAIR_DTYPE.itemsize == C.sizeof(_lib.Pair) == 80
assert PATCH_DTYPE.itemsize == C.sizeof(Patch) == 96
assert CAMERA_DTYPE.itemsize == C.sizeof(_lib.Camera) == 120


class _PinnedRing:
    """Small host->device uploads (record tables, trig tables) without blocking
    the host: a pageable copy waits for everything already queued on the stream,
    i.e. for the previous stitch, and serialises host and GPU.  Slots are pinned,
    reused round-robin, each guarded by an event recorded after its copy."""

    SLOTS = 8
    SLOT_BYTES = 1 << 20

    def __init__(self, device):
        torch = _torch()
        self.device, self.next = device, 0
        # one pinning call for the whole ring (each costs tens of milliseconds)
        self.store = torch.empty(self.SLOTS * self.SLOT_BYTES, dtype=torch.uint8).pin_memory()
        self.slots = [self.store[k * self.SLOT_BYTES:(k + 1) * self.SLOT_BYTES]
                      for k in range(self.SLOTS)]
        self.events = [torch.cuda.Event() for _ in range(self.SLOTS)]
        self.used = [False] * self.SLOTS

    def upload(self, array):
        torch = _torch()
        raw = np.ascontiguousarray(array).view(np.uint8).reshape(-1)
        if raw.size > self.SLOT_BYTES:               # rare: larger than a slot
            return torch.from_numpy(raw.copy()).to(self.device)
        k = self.next
        self.next = (self.next + 1) % self.SLOTS
        if self.used[k]:
            self.events[k].synchronize()             # its previous copy has landed
        self.slots[k].numpy()[:raw.size] = raw
        dev = self.slots[k][:raw.size].to(self.device, non_blocking=True)
        self.events[k].record(torch.cuda.current_stream(self.device))
        self.used[k] = True
        return dev



def reflect_closed(lo, hi, n):
    """Smallest [a, b) inside [0, n) that holds reflect_101(p, n) for every p in
    [lo, hi) (cv2.BORDER_REFLECT_101: ... c b | a b c ... z | y x ...)."""
    if n == 1:
        return 0, 1
    if lo < -(n - 1) or hi > 2 * n - 1:      # a second reflection could occur
        return 0, n
    a, b = max(lo, 0), min(hi, n)
    if lo < 0:                               # p in [lo, 0) lands on [1, -lo]
        a, b = min(a, 1), max(b, 1 - lo)
    if hi > n:                               # p in [n, hi) lands on [2n-1-hi, n-2]
        a, b = min(a, 2 * n - 1 - hi), max(b, n - 1)
    return max(a, 0), min(b, n)


def windows_for(box, rect, radius, strip=None):
    """Rectangles A and V of include/pano360.h ("Windows") for one patch.

    box: (ymin, ymax, xmin, xmax) of the owned pixels, mosaic coordinates,
    inclusive, empty when ymax < ymin.  ``strip`` = (c0, c1) keeps only the
    part of A inside those mosaic columns (one GPU's share of the mosaic).
    Returns patch-local ((ay0, ay1, ax0, ax1), (vy0, vy1, vx0, vx1)) or None
    when nothing is left."""
    ymin, ymax, xmin, xmax = (int(v) for v in box)
    if ymax < ymin or xmax < xmin:
        return None
    y0, y1, x0, x1 = rect
    h, w = y1 - y0, x1 - x0
    ay0, ay1 = max(ymin - y0 - radius, 0), min(ymax - y0 + 1 + radius, h)
    ax0, ax1 = max(xmin - x0 - radius, 0), min(xmax - x0 + 1 + radius, w)
    if strip is not None:
        ax0, ax1 = max(ax0, strip[0] - x0), min(ax1, strip[1] - x0)
        if ax1 <= ax0:
            return None
    vy0, vy1 = reflect_closed(ay0 - radius, ay1 + radius, h)
    vx0, vx1 = reflect_closed(ax0 - radius, ax1 + radius, w)
    # V's columns end on multiples of 4 (clipped to the patch): include/pano360.h, "Windows"
    vx0, vx1 = min(vx0, ax0) & ~3, min((max(vx1, ax1) + 3) & ~3, w)
    return (ay0, ay1, ax0, ax1), (min(vy0, ay0), max(vy1, ay1), vx0, vx1)


def _reflect_closed_many(lo, hi, n):
    """reflect_closed on int64 arrays."""
    a, b = np.maximum(lo, 0), np.minimum(hi, n)
    neg, over = lo < 0, hi > n
    a = np.where(neg, np.minimum(a, 1), a)
    b = np.where(neg, np.maximum(b, 1 - lo), b)
    a = np.where(over, np.minimum(a, 2 * n - 1 - hi), a)
    b = np.where(over, np.maximum(b, n - 1), b)
    a, b = np.maximum(a, 0), np.minimum(b, n)
    full = (lo < -(n - 1)) | (hi > 2 * n - 1)
    a, b = np.where(full, 0, a), np.where(full, n, b)
    one = n == 1
    return np.where(one, 0, a), np.where(one, 1, b)


def windows_for_many(boxes, rects, radius, strip=None):
    """``windows_for`` for k records at once (the host sits between two GPU stages
    while it lays these out).  boxes, rects: int arrays [k][4].  Returns
    (keep [k] bool, A [k][4], V [k][4]) with A, V patch-local (y0, y1, x0, x1)."""
    boxes, rects = np.asarray(boxes, np.int64).reshape(-1, 4), np.asarray(rects, np.int64).reshape(-1, 4)
    ymin, ymax, xmin, xmax = boxes.T
    y0, y1, x0, x1 = rects.T
    h, w = y1 - y0, x1 - x0
    keep = (ymax >= ymin) & (xmax >= xmin)
    ay0, ay1 = np.maximum(ymin - y0 - radius, 0), np.minimum(ymax - y0 + 1 + radius, h)
    ax0, ax1 = np.maximum(xmin - x0 - radius, 0), np.minimum(xmax - x0 + 1 + radius, w)
    if strip is not None:
        ax0, ax1 = np.maximum(ax0, strip[0] - x0), np.minimum(ax1, strip[1] - x0)
        keep &= ax1 > ax0
    vy0, vy1 = _reflect_closed_many(ay0 - radius, ay1 + radius, h)
    vx0, vx1 = _reflect_closed_many(ax0 - radius, ax1 + radius, w)
    area = np.stack([ay0, ay1, ax0, ax1], axis=1)
    vx0 = np.minimum(vx0, ax0) & ~3
    vx1 = np.minimum((np.maximum(vx1, ax1) + 3) & ~3, w)
    window = np.stack([np.minimum(vy0, ay0), np.maximum(vy1, ay1), vx0, vx1], axis=1)
    return keep, area, window


class DevicePatch:
    """Stage-level patch: four planes and a mask over the whole patch, as the
    blender protocol hands them over (V = A = the patch)."""

    def __init__(self, rect, device, n_blur):
        torch = _torch()
        y0, y1, x0, x1 = rect
        self.rect = rect
        self.h, self.w = y1 - y0, x1 - x0
        self.area = self.window = (0, self.h, 0, self.w)
        self.pitch = (self.w + 3) & ~3
        f32 = dict(dtype=torch.float32, device=device)
        self.planes = torch.empty((4, self.h, self.pitch), **f32)
        self.mask = torch.empty((self.h, self.w), dtype=torch.uint8, device=device)
        self.blurred = torch.empty((n_blur, 4, self.h, self.pitch), **f32) if n_blur else None
        self.scratch = torch.empty((n_blur, 4, self.h, self.pitch), **f32) if n_blur else None

    def record(self, index):
        y0, _, x0, _ = self.rect
        opt = lambda t: t.data_ptr() if t is not None else 0   # noqa: E731
        return (self.planes.data_ptr(), self.mask.data_ptr(), opt(self.blurred),
                opt(self.scratch), y0, x0, self.h, self.w, 0, 0, self.h, self.w,
                0, 0, self.h, self.w, self.pitch, self.pitch, index, 0)


class PatchTable:
    """Device array of ``pano_patch`` records + the extents that size the grids."""

    @classmethod
    def from_layout(cls, records, lay, eng):
        """Records laid out by ``pano_layout_windows`` (tile offsets and extents known)."""
        self = cls.__new__(cls)
        self.host, self.n, self.n_tiles = records, len(records), int(lay.n_tiles)
        self.dev = eng.to_device(records)
        self.max_vw, self.max_vh, self.max_aw, self.max_ah = (int(lay.max_vw), int(lay.max_vh),
                                                              int(lay.max_aw), int(lay.max_ah))
        return self

    def __init__(self, records, eng):
        self.host = np.array(records, dtype=PATCH_DTYPE).reshape(-1)
        self.n = len(self.host)
        aw, ah = self.host["aw"].astype(np.int64), self.host["ah"].astype(np.int64)
        if eng.tile_grid == 32:
            # 32 x 32 tiles anchored at multiples of 32 in patch coordinates
            ax0, ay0 = self.host["ax0"].astype(np.int64), self.host["ay0"].astype(np.int64)
            tiles = (((ax0 + aw - 1) >> 5) - (ax0 >> 5) + 1) * (((ay0 + ah - 1) >> 5) - (ay0 >> 5) + 1)
            tiles = np.where((aw > 0) & (ah > 0), tiles, 0)
        else:
            tiles = ((aw + 63) // 64) * ((ah + 127) // 128)          # 64 x 128 column tiles
        self.host["tiles_off"] = np.concatenate([[0], np.cumsum(tiles)[:-1]]) if self.n else 0
        self.n_tiles = int(tiles.sum())
        self.dev = eng.to_device(self.host)
        mx = lambda k: int(self.host[k].max()) if self.n else 0   # noqa: E731
        self.max_vw, self.max_vh, self.max_aw, self.max_ah = (mx("vw"), mx("vh"), mx("aw"),
                                                              mx("ah"))

    @property
    def ptr(self):
        return _ptr(self.dev)


def patch_table(patches, eng):
    """``pano_patch`` table of stage-level patches."""
    return PatchTable([p.record(i) for i, p in enumerate(patches)], eng)


class WindowInfo:
    """What callers may want to know about one fused patch."""

    def __init__(self, area, window):
        self.area, self.window = area, window


class FusedPatches:
    """Windows of every patch packed into three arenas (colour planes over V,
    blurred copies over A, row-pass scratch) + the patch table pointing into
    them.  ``entries`` = [(camera index, patch rect, ``windows_for`` output)], one
    per owned column span, in camera order."""

    @classmethod
    def from_regions(cls, raw, max_spans, rects, have, radius, strip, n_blur, eng, rec=None):
        """The record table straight from the region search's output: one native call
        lays out rectangles, pitches, arena offsets and tile offsets
        (``pano_layout_windows``); the arenas are (re)used as in ``__init__``.
        ``have``: uint8 [n], 0 = that camera's frame is not resident."""
        lib = _lib.lib()
        n = len(rects)
        if rec is None:
            rec = np.zeros(n * max_spans, dtype=PATCH_DTYPE)
        lay = _lib.Layout()
        raw = np.ascontiguousarray(raw, np.int32)
        rects = np.ascontiguousarray(rects, np.int32)
        have = np.ascontiguousarray(have, np.uint8)
        _lib.check(lib.pano_layout_windows(
            eng.tile_grid, raw.ctypes.data, n, max_spans, rects.ctypes.data, have.ctypes.data,
            radius,
            strip[0], strip[1], n_blur, rec.ctypes.data, len(rec), C.byref(lay)),
            "pano_layout_windows")
        rec = rec[:lay.n_records]
        if lay.missing:
            missing = sorted({int(i) for i in rec["index"] if not have[int(i)]})
            raise _lib.PanoError(f"frames {missing} are needed for columns [{strip[0]}, "
                                 f"{strip[1]}) but are not resident on this device")
        self = cls.__new__(cls)
        self.planes = eng.arena("planes", int(lay.planes_floats))
        self.blurred = eng.arena("blurred", int(lay.blurred_floats))
        self.scratch = eng.arena("scratch", int(lay.scratch_floats))
        _lib.check(lib.pano_layout_place(rec.ctypes.data, len(rec), self.planes.data_ptr(),
                                         self.blurred.data_ptr(), self.scratch.data_ptr()),
                   "pano_layout_place")
        self._area = self._window = None
        self.table = PatchTable.from_layout(rec, lay, eng)
        return self

    @classmethod
    def from_records(cls, rec, lay, table_dev, arenas):
        """Records laid out and uploaded by ``pano_stitch_multiband`` (a host copy of them)."""
        self = cls.__new__(cls)
        self.planes, self.blurred, self.scratch = arenas
        self._area = self._window = None
        table = PatchTable.__new__(PatchTable)
        table.host, table.n, table.n_tiles, table.dev = rec, len(rec), int(lay.n_tiles), table_dev
        table.max_vw, table.max_vh, table.max_aw, table.max_ah = (
            int(lay.max_vw), int(lay.max_vh), int(lay.max_aw), int(lay.max_ah))
        self.table = table
        return self

    def _rectangles(self):
        if self._area is None:
            h = self.table.host
            i64 = lambda k: h[k].astype(np.int64)   # noqa: E731
            self._area = np.stack([i64("ay0"), i64("ay0") + i64("ah"), i64("ax0"),
                                   i64("ax0") + i64("aw")], axis=1)
            self._window = np.stack([i64("vy0"), i64("vy0") + i64("vh"), i64("vx0"),
                                     i64("vx0") + i64("vw")], axis=1)
        return self._area, self._window

    def __init__(self, entries, eng, n_blur):
        if isinstance(entries, tuple):                   # (index [k], rects [k][4], A [k][4], V [k][4])
            index, rects, area, window = (np.asarray(v, np.int64) for v in entries)
        else:
            index = np.array([e[0] for e in entries], np.int64)
            rects = np.array([e[1] for e in entries], np.int64).reshape(-1, 4)
            area = np.array([e[2][0] for e in entries], np.int64).reshape(-1, 4)
            window = np.array([e[2][1] for e in entries], np.int64).reshape(-1, 4)
        n = len(index)
        rec = np.zeros(n, dtype=PATCH_DTYPE)
        self._area, self._window = area, window
        rec["y0"], rec["x0"] = rects[:, 0], rects[:, 2]
        rec["h"], rec["w"] = rects[:, 1] - rects[:, 0], rects[:, 3] - rects[:, 2]
        rec["index"] = index
        rec["vy0"], rec["vx0"] = window[:, 0], window[:, 2]
        rec["vh"], rec["vw"] = window[:, 1] - window[:, 0], window[:, 3] - window[:, 2]
        rec["ay0"], rec["ax0"] = area[:, 0], area[:, 2]
        rec["ah"], rec["aw"] = area[:, 1] - area[:, 0], area[:, 3] - area[:, 2]
        rec["vpitch"] = (rec["vw"] + 3) & ~3
        vh, ah = rec["vh"].astype(np.int64), rec["ah"].astype(np.int64)
        planes_sz = 3 * vh * rec["vpitch"]
        lead = np.zeros(n, np.int64)
        if eng.tile_grid == 32:
            # the matrix-core blur writes 32-column tile rows anchored at multiples of 32 in
            # patch coordinates: 128-byte rows, and the anchor column on a 128-byte boundary,
            # make each such row one cache line instead of two; no row-pass scratch
            rec["apitch"] = (rec["aw"] + 31) & ~31
            lead = rec["ax0"].astype(np.int64) & 31
            blurred_sz = n_blur * 4 * ah * rec["apitch"] + 32
            scratch_sz = np.zeros(n, np.int64)
        else:
            rec["apitch"] = (rec["aw"] + 3) & ~3
            blurred_sz = n_blur * 4 * ah * rec["apitch"]
            scratch_sz = n_blur * 4 * vh * rec["apitch"]
        self.planes = eng.arena("planes", int(planes_sz.sum()))
        self.blurred = eng.arena("blurred", int(blurred_sz.sum()) + 32)
        self.scratch = eng.arena("scratch", int(scratch_sz.sum()))
        for key, sizes, arena in (("planes", planes_sz, self.planes),
                                  ("blurred", blurred_sz, self.blurred),
                                  ("scratch", scratch_sz, self.scratch)):
            offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
            base = arena.data_ptr()
            if key == "blurred":
                base += -base % 128                         # arenas come 512-byte aligned anyway
                offs = offs + lead
            rec[key] = base + 4 * offs
        self.table = PatchTable(rec, eng)

    @property
    def info(self):
        return [WindowInfo(tuple(int(v) for v in a), tuple(int(v) for v in w))
                for a, w in zip(*self._rectangles())]

    def __iter__(self):
        return iter(self.info)

    def __len__(self):
        return self.table.n

    @property
    def warped_pixels(self):
        return int((self.table.host["vh"].astype(np.int64) * self.table.host["vw"]).sum())

    @property
    def blurred_pixels(self):
        return int((self.table.host["ah"].astype(np.int64) * self.table.host["aw"]).sum())


class Engine:
    """Sequences the HIP stages for one device and one stream: owns a ``pano_ctx``
    (include/pano360.h), the workspaces kept from stitch to stitch and the pinned upload
    ring.  Nothing is shared between engines, so several may run side by side (one per
    host thread, each under its own ``torch.cuda.stream``).

    ``blur``: which kernels run the multiband Gaussian levels - "mfma" (default, the fused
    matrix-core kernel) or "valu" (separate float32 row / column passes on the vector ALU).
    ``side_stream``: 0 = one stream (default); 1 = interior collapse and the blur's work
    list on a second stream; 2 = the work list only.  The second stream paid while the host
    kept the GPU waiting between the region search and the warp (2.76 -> 2.67 ms); with the
    native record layout there is no gap left to fill and one stream is as fast or faster
    (cfg3 medians of 5 x 20 stitches: 2.352 / 2.421 / 2.350 ms; cfg2 0.661 / 0.654 /
    0.669), so it stays an option for hosts slower than this pool's."""

    def __init__(self, device=None, blur="mfma", side_stream=0, own_prune=True):
        torch = _torch()
        self.lib = _lib.lib()
        if not torch.cuda.is_available() or self.lib.pano_device_count() < 1:
            raise _lib.PanoError("no MI355X visible: the HIP path has no CPU fallback")
        self.device = torch.device(device if device is not None
                                   else f"cuda:{torch.cuda.current_device()}")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        self._device_index = index
        handle = C.c_void_p()
        self._ctx_stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.pano_ctx_create(index, C.c_void_p(self._ctx_stream), C.byref(handle)),
                   "pano_ctx_create")
        self._ctx = handle
        if blur not in ("mfma", "valu"):
            raise ValueError(f"blur kernel {blur!r}: 'mfma' or 'valu'")
        self.set_option(_lib.OPT_BLUR_KERNEL, _lib.BLUR_VALU if blur == "valu" else _lib.BLUR_MFMA)
        # (PANO_OWN_PRUNE: A/B timing of the ownership kernels - 3 = round 4's one-level kernel)
        self.set_option(_lib.OPT_OWN_PRUNE, int(os.environ.get("PANO_OWN_PRUNE", "1")) if own_prune
                        else 0)
        self.tile_grid = int(self.lib.pano_blur_tile_grid(self._ctx))
        self.interior_block = int(self.lib.pano_interior_block())
        lut = np.arange(256, dtype=np.float32) / np.float32(255)   # stitcher.py:259
        self.lut255 = torch.from_numpy(lut).to(self.device)
        self._hats = {}
        self._taps = {}
        self._region_bufs = {}
        self._arenas = {}           # name -> float32 tensor kept across stitches
        self._ring = None
        # second stream for the part of the collapse that needs no blurred planes
        self.side = torch.cuda.Stream(self.device)
        self.overlap_interior = side_stream == 1
        self.overlap_prepare = side_stream in (1, 2)
        self.warp_need = {"1": True, "0": False}.get(os.environ.get("PANO_WARP_NEED", ""), "auto")
        # ("auto" | True | False: see multiband_fused; the environment switch is for A/B timing)
        self._cam_template = None
        # the whole launch sequence of a fused stitch in one native call (pano_stitch_multiband);
        # False: launch by launch from here (the same entry points; what the side streams use)
        self.native_stitch = side_stream == 0 and os.environ.get("PANO_NATIVE_STITCH", "1") != "0"
        if os.environ.get("PANO_STITCH_STREAMS", "1") == "0":       # (A/B timing)
            self.set_option(_lib.OPT_STITCH_STREAMS, 0)
        if os.environ.get("PANO_STITCH_ASYNC", "0") == "1":         # (A/B timing)
            self.set_option(_lib.OPT_STITCH_ASYNC, 1)
        if os.environ.get("PANO_LEVEL_CLASSES", "0") == "1":        # (A/B: the collapse by level classes)
            self.set_option(_lib.OPT_LEVEL_CLASSES, 1)
        if os.environ.get("PANO_SIFT_GRAPH", "1") == "0":           # (A/B timing: launch by launch)
            self.set_option(_lib.OPT_SIFT_GRAPH, 0)
        if os.environ.get("PANO_BLUR_SEG_T"):                       # (A/B timing of the segments' length)
            self.set_option(_lib.OPT_BLUR_SEG_LEN, int(os.environ["PANO_BLUR_SEG_T"]))
        self._stitch_ws = {}
        self._plans = PlanMemo()
        # Trusted stitches (ShardedStitcher with the plan memo, bench's plan-cached figures):
        # when a stitch repeats the previous one's Plan OBJECT (a memo hit: same cameras, bit for
        # bit), strip, levels and resident frames, its layout is the previous one's - the owner
        # map and the regions are functions of exactly those - so the native call queues
        # everything with that verified layout and returns without its one wait
        # (pano_stitch_args.trust_layout); every kernel still runs.  ``verify_trusted`` compares
        # the layout the device really made.  Off by default: a caller that mutates a Plan in
        # place between stitches must not switch it on.
        self.trust_layout = False
        # ``keep_geometry`` (with trusted stitches): such a repeat also re-uses what the previous
        # stitch left on the device - owner map, valid mask, interior map, record table, tile
        # flags, the blur's work list, all functions of the same inputs - and queues the warp, the
        # blur and the collapse alone (trust_layout = 3).  The owner / valid tensors are then the
        # SAME tensors from stitch to stitch: read-only for the caller.
        self.keep_geometry = False
        self.last_kept_geometry = False
        self._trusted = None            # (signature, Plan, FusedPatches, owner, valid) of the last verified stitch

    def __del__(self):
        ctx, self._ctx = getattr(self, "_ctx", None), None
        if ctx is not None and ctx.value:
            try:
                self.lib.pano_ctx_destroy(ctx)
            except Exception:       # noqa: BLE001 - interpreter shutdown
                pass

    # -- the context ----------------------------------------------------------
    def ctx(self, stream=None):
        """The context handle, targeted at ``stream`` (default: torch's current stream of
        this device - what ``with torch.cuda.stream(s):`` selects in this thread)."""
        raw = (stream.cuda_stream if stream is not None
               else _torch()._C._cuda_getCurrentRawStream(self._device_index))
        if raw != self._ctx_stream:
            _lib.check(self.lib.pano_ctx_set_stream(self._ctx, C.c_void_p(raw)),
                       "pano_ctx_set_stream")
            self._ctx_stream = raw
        return self._ctx

    def set_option(self, option, value):
        _lib.check(self.lib.pano_ctx_set_option(self._ctx, option, int(value)),
                   "pano_ctx_set_option")
        if option == _lib.OPT_BLUR_KERNEL:
            self.tile_grid = int(self.lib.pano_blur_tile_grid(self._ctx))
        # an option may change how a stitch is laid out: the verified stitch a trusted repeat
        # would ride on is not this configuration's (the context voids its side of it too)
        self._trusted = None

    def get_option(self, option):
        value = C.c_int(0)
        _lib.check(self.lib.pano_ctx_get_option(self._ctx, option, C.byref(value)),
                   "pano_ctx_get_option")
        return value.value

    def stitch_counts(self):
        """(stitches that went through on the device-side layout, attempts that fell back to
        the host layout) of this engine's context."""
        ok, back = C.c_int(0), C.c_int(0)
        _lib.check(self.lib.pano_stitch_counts(self._ctx, C.byref(ok), C.byref(back)),
                   "pano_stitch_counts")
        return ok.value, back.value

    def timing(self, on):
        _lib.check(self.lib.pano_timing_enable(self._ctx, 1 if on else 0), "pano_timing_enable")

    def kernel_times(self):
        """{kernel name: (summed ms, launches)} since ``timing(True)``; waits for the events."""
        out = {}
        for kid in range(self.lib.pano_kernel_count()):
            ms, n = C.c_double(), C.c_int()
            _lib.check(self.lib.pano_timing_read(self._ctx, kid, C.byref(ms), C.byref(n)),
                       "pano_timing_read")
            if n.value:
                out[self.lib.pano_kernel_name(kid).decode()] = (ms.value, n.value)
        return out

    # -- workspaces -----------------------------------------------------------
    def arena(self, name, floats):
        """Workspace reused from stitch to stitch (grown by 12 % when too small): a
        fresh multi-gigabyte allocation costs 70-90 ms, more than ten stitches."""
        torch = _torch()
        have = self._arenas.get(name)
        if have is None or have.numel() < floats:
            self._arenas[name] = None                       # let the old block go first
            have = self._arenas[name] = torch.empty(int(floats * 1.125) + 4,
                                                    dtype=torch.float32, device=self.device)
        return have

    def to_device(self, array):
        """Bytes of a NumPy array as a uint8 device tensor (asynchronous copy out of this
        engine's pinned ring)."""
        if self._ring is None:
            self._ring = _PinnedRing(self.device)
        return self._ring.upload(array)

    # -- small cached tables ------------------------------------------------
    def hat_tables(self, shape):
        torch = _torch()
        if shape not in self._hats:
            h, w = shape
            self._hats[shape] = (torch.from_numpy(hat(w)).to(self.device),
                                 torch.from_numpy(hat(h)).to(self.device))
        return self._hats[shape]

    def blur_tables(self, n_levels):
        """(host taps, ntaps C array, n_blur, largest radius) of the n_levels-1 blurs;
        tables back to back, padded as include/pano360.h says.  The context keeps the
        device copies, keyed on the values."""
        if n_levels not in self._taps:
            sig = level_sigmas(n_levels)
            sizes = [gaussian_ksize(s) for s in sig]
            rmax = max([k // 2 for k in sizes], default=0)
            flat = np.concatenate([padded_taps(gaussian_taps(k, s), (rmax - k // 2) & 3)
                                   for k, s in zip(sizes, sig)]) if sig else np.zeros(1, np.float32)
            self._taps[n_levels] = (np.ascontiguousarray(flat, np.float32),
                                    (C.c_int * max(len(sizes), 1))(*sizes), len(sizes), rmax)
        return self._taps[n_levels]

    def upload_frames(self, imgs):
        """uint8 host images -> device tensors; images already on the device pass through."""
        torch = _torch()
        return [im.to(self.device) if isinstance(im, torch.Tensor) else
                torch.from_numpy(np.ascontiguousarray(im, np.uint8)).to(self.device)
                for im in imgs]

    def cached_plan(self, shapes, rots, intrs, padded, max_resolution, table_cols=None):
        """The uploaded ``Plan`` of these cameras, kept from stitch to stitch (``PlanMemo``): a
        rig that does not move stitches every time step with the same geometry, and the float64
        plan is a fifth of a 4K stitch's time on the host.  The reference recomputes all of it
        per stitch (stitcher.py:276-302) - to the same values, the plan being a pure function of
        what the memo is keyed on."""
        return self._plans.get(shapes, rots, intrs, padded, max_resolution, table_cols,
                               make=lambda *a: self.upload_plan(Plan(*a)))

    def upload_plan(self, plan):
        """Trig tables to the device: one asynchronous copy out of a pinned staging
        buffer (three pageable copies cost 1.5 ms of host time per stitch)."""
        torch = _torch()
        nx, ny = len(plan.sin_t), len(plan.tan_p)
        dev = self.to_device(np.concatenate([plan.sin_t, plan.cos_t, plan.tan_p])).view(
            torch.float64)
        plan.dev = (dev[:nx], dev[nx:2 * nx], dev[2 * nx:2 * nx + ny])
        return plan

    # -- stage-level calls (whole patches, the blender protocol) ------------------
    def add_weights(self, frame, lut=None):
        """_add_weights on device: uint8 [H,W,3] -> float32 [H,W,4]; ``lut`` = the
        camera's colour table when the exposures were equalised."""
        torch = _torch()
        h, w = frame.shape[:2]
        hx, hy = self.hat_tables((h, w))
        out = torch.empty((h, w, 4), dtype=torch.float32, device=self.device)
        lut = self.lut255 if lut is None else lut
        _lib.check(self.lib.pano_add_weights(self.ctx(), _ptr(frame), h, w, _ptr(lut), _ptr(hx),
                                             _ptr(hy), _ptr(out)),
                   "pano_add_weights")
        return out

    def warp(self, frame, plan, index, patch, want_maps=False, lut=None):
        torch = _torch()
        lut = self.lut255 if lut is None else lut
        sh, sw = frame.shape[:2]
        hx, hy = self.hat_tables((sh, sw))
        y0, _, x0, _ = plan.rects[index]
        mx = my = None
        if want_maps:
            mx = torch.empty((patch.h, patch.w), dtype=torch.float32, device=self.device)
            my = torch.empty_like(mx)
        proj = plan.projs[index]
        _lib.check(self.lib.pano_warp_spherical(
            self.ctx(), _ptr(frame), sh, sw, proj.ctypes.data_as(C.c_void_p), _ptr(plan.dev[0]),
            _ptr(plan.dev[1]), _ptr(plan.dev[2]), _ptr(lut), _ptr(hx), _ptr(hy),
            x0, y0, patch.w, patch.h, _ptr(patch.planes), _ptr(patch.mask), _ptr(mx),
            _ptr(my)), "pano_warp_spherical")
        return mx, my

    def warp_all(self, frames, plan, n_blur=0, want_maps=False, luts=None):
        patches, maps = [], []
        for i, frame in enumerate(frames):
            patch = DevicePatch(plan.rects[i], self.device, n_blur)
            maps.append(self.warp(frame, plan, i, patch, want_maps,
                                  None if luts is None else luts[i]))
            patches.append(patch)
        return patches, maps

    def ownership(self, table, shape):
        torch = _torch()
        H, W = shape
        owner = torch.empty((H, W), dtype=torch.int16, device=self.device)
        valid = torch.empty((H, W), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.pano_ownership(self.ctx(), table.ptr, table.n, H, W, _ptr(owner),
                                           _ptr(valid)), "pano_ownership")
        return owner, valid

    def multiband(self, patches, shape, n_levels, want_float=False, table=None):
        """Stage-level multiband: ownership from the alpha planes -> blurs ->
        collapse.  Returns (mosaic u8, float mosaic or None, owner, valid)."""
        if table is None:
            table = patch_table(patches, self)
        owner, valid = self.ownership(table, shape)
        mosaic, fl = self.blur_and_compose(table, owner, valid, shape, n_levels, want_float)
        return mosaic, fl, owner, valid

    def interior_map(self, owner, radius, strip=None):
        """uint8 [ceil(H/B)][ceil(W/B)], B = the library's interior block: B x B blocks whose
        pixels all have a single owner within ``radius`` - there the multiband mosaic is the
        owner's colour."""
        torch = _torch()
        H, W = owner.shape
        c0, c1 = strip if strip is not None else (0, W)
        ib = self.interior_block
        shape8 = ((H + ib - 1) // ib, (W + ib - 1) // ib)
        bown = torch.empty((2,) + shape8, dtype=torch.int16, device=self.device)
        interior = torch.empty(shape8, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.pano_interior_map(self.ctx(), _ptr(owner), H, W, c0, c1, radius,
                                              _ptr(bown), _ptr(interior)), "pano_interior_map")
        return interior

    def interior_classes(self, owner, radii, strip=None):
        """(interior map at radii[-1], level classes): uint8 [ceil(H/B)][ceil(W/B)] each.  A block's
        class is the number of leading levels whose Gaussian radius (``radii``, ascending) finds
        one owner around every pixel of the block; class len(radii) = interior.  The collapse
        gathers, on a pixel of class j >= 1, only the blurred copies j - 1 and up."""
        torch = _torch()
        H, W = owner.shape
        c0, c1 = strip if strip is not None else (0, W)
        ib = self.interior_block
        shape8 = ((H + ib - 1) // ib, (W + ib - 1) // ib)
        bown = torch.empty((2,) + shape8, dtype=torch.int16, device=self.device)
        interior = torch.empty(shape8, dtype=torch.uint8, device=self.device)
        classes = torch.empty(shape8, dtype=torch.uint8, device=self.device)
        arr = (C.c_int * len(radii))(*[int(r) for r in radii])
        _lib.check(self.lib.pano_interior_classes(self.ctx(), _ptr(owner), H, W, c0, c1, arr,
                                                  len(radii), _ptr(bown), _ptr(interior),
                                                  _ptr(classes)), "pano_interior_classes")
        return interior, classes

    def level_radii(self, n_levels):
        """Gaussian radii of the n_levels - 1 blurred levels (half their apertures)."""
        return [gaussian_ksize(s) // 2 for s in level_sigmas(n_levels)]

    # Weights of a column's mosaic pixels (ownership, the collapse's interior pixels), valid pixels
    # (the warp's windows) and pixels near a seam (the blur's active tiles, the collapse's gathers) in
    # its cost.  The first two are their kernels' shares of a config-3 stitch (profiles/r05/final);
    # the third, 0.98 by that count, is calibrated on strips instead: a seam also brings records,
    # work items and segment leads with it, and world-4 / world-8 splits of config 3 come out level
    # at 2.5 (profiles/r05/cost_shares_scan_cfg3.txt: slowest rank 0.234 ms against 0.246 at 0.98
    # and 0.252 at 4, where the end strips become the slow ones)
    COLUMN_COST_SHARES = (0.27, 0.25, 2.5)

    def column_costs(self, plan, n_levels, shortcut=True):
        """Relative cost of every mosaic column in a multiband stitch, from the geometry alone (one
        ownership pass and the interior map of the whole mosaic; no frames): what
        ``dist.balanced_strip_bounds`` cuts column strips of equal WORK from.  A sweep's first
        and last cameras own their ends of the mosaic alone - no seam, no blur - so equal-width
        strips leave the middle ranks a third more than their share (config 3, eight ranks)."""
        torch = _torch()
        H, W = plan.shape
        if not hasattr(plan, "dev"):
            self.upload_plan(plan)
        n_blur, radius = self.blur_tables(n_levels)[2:]
        owner, valid = self.ownership_cameras(plan)
        a, b, g = self.COLUMN_COST_SHARES
        if os.environ.get("PANO_COST_SHARES"):                      # (A/B of the model's weights)
            a, b, g = (float(v) for v in os.environ["PANO_COST_SHARES"].split(","))
        cost = np.full(W, a / W)
        per = valid.ne(0).sum(0, dtype=torch.int64).cpu().numpy().astype(np.float64)
        if per.sum() > 0:
            cost += b * per / per.sum()
        if n_blur and shortcut:
            interior = self.interior_map(owner, radius)
            near = interior.eq(0).sum(0, dtype=torch.int64).cpu().numpy().astype(np.float64)
            near = np.repeat(near, self.interior_block)[:W]
            if near.sum() > 0:
                cost += g * near / near.sum()
        elif per.sum() > 0:
            cost += g * per / per.sum()          # every valid pixel goes through the blur
        return cost

    def active_tile_pixels(self):
        """Pixels of the 32 x 32 tiles the last blur really computed - whole tiles, an upper
        bound - or all of every rectangle A when no tile flags were in use.  Synchronises."""
        table, flags = getattr(self, "last_tiles", (None, None))
        if table is None:
            return 0
        host = table.host
        if flags is None:
            return int((host["ah"].astype(np.int64) * host["aw"]).sum())
        if self.tile_grid == 32:
            return int(flags[:table.n_tiles].to(_torch().int64).sum().item()) * 1024
        # the vector-ALU kernels (Engine(blur="valu")) flag 64 x 128 column tiles of A
        on = flags.cpu().numpy()
        total = 0
        for rec in host:
            ntx, nty = (int(rec["aw"]) + 63) // 64, (int(rec["ah"]) + 127) // 128
            if ntx * nty == 0:
                continue
            grid = on[int(rec["tiles_off"]):int(rec["tiles_off"]) + ntx * nty].reshape(nty, ntx)
            wx = np.minimum(64, int(rec["aw"]) - 64 * np.arange(ntx))
            wy = np.minimum(128, int(rec["ah"]) - 128 * np.arange(nty))
            total += int((grid.astype(np.int64) * wy[:, None] * wx[None, :]).sum())
        return total

    def gather_bytes(self, shape, n_levels):
        """Algorithmic bytes the last fused stitch's collapse gathers from the warped planes and the
        blurred copies: over every record's rectangle A, a pixel of class 0 takes the planes (12 B)
        and all n_levels - 1 copies (16 B each), a pixel of class j >= 1 the colour of copy j - 1
        (12 B) and the copies j and up - plus the owner's planes (12 B) once per such pixel - and an
        interior pixel nothing (``multiband_compose_kernel``).  Without level classes (option off,
        no shortcut) every pixel that is not interior counts as class 0.  None before a stitch.
        Synchronises (reporting only)."""
        torch = _torch()
        table, flags = getattr(self, "last_tiles", (None, None))
        classes = getattr(self, "last_classes", None)
        if table is None:
            return None
        H, W = shape
        nb = n_levels - 1
        host = table.host
        interior = getattr(self, "last_interior", None)
        if flags is None or (classes is None and interior is None):
            return float((host["ah"].astype(np.int64) * host["aw"]).sum()) * (12 + 16 * nb)
        if classes is None:                  # no level classes: a pixel is interior (class nb) or class 0
            classes = interior * nb
        ib = self.interior_block
        px = classes.repeat_interleave(ib, 0).repeat_interleave(ib, 1)[:H, :W]
        per_class = torch.tensor([12 + 16 * nb] + [12 + 16 * (nb - j) for j in range(1, nb)] + [0],
                                 dtype=torch.float64, device=self.device)
        total = torch.zeros((), dtype=torch.float64, device=self.device)
        for rec in host:
            aw, ah = int(rec["aw"]), int(rec["ah"])
            if aw <= 0 or ah <= 0:
                continue
            y0, x0 = int(rec["y0"]) + int(rec["ay0"]), int(rec["x0"]) + int(rec["ax0"])
            sub = px[max(y0, 0):min(y0 + ah, H), max(x0, 0):min(x0 + aw, W)]
            if sub.numel():
                counts = torch.bincount(sub.reshape(-1).long(), minlength=nb + 1)[:nb + 1]
                total += (counts.double() * per_class).sum()
        # the owner's planes on the pixels of the classes in between, once each
        mid = ((px > 0) & (px < nb)).sum().double() * 12.0
        return float((total + mid).item())

    def compose_interior_async(self, owner, shape, strip, interior, cams, plan, luts,
                               want_float=False, mosaic_out=None):
        """Part 1 of the collapse - the interior pixels, which need the owner map and
        the frames only - queued on the side stream behind everything queued so far.
        Returns (mosaic, float mosaic, event) for ``blur_and_compose(out=...)``."""
        torch = _torch()
        H, W = shape
        main = torch.cuda.current_stream(self.device)
        mosaic = (mosaic_out if mosaic_out is not None else
                  torch.empty((H, W, 3), dtype=torch.uint8, device=self.device))
        fl = (torch.empty((H, W, 3), dtype=torch.float32, device=self.device)
              if want_float else None)
        ready = torch.cuda.Event()
        ready.record(main)
        self.side.wait_event(ready)
        for t in (mosaic, fl, owner, interior, cams):
            if t is not None:
                t.record_stream(self.side)
        _lib.check(self.lib.pano_multiband_compose(
            self.ctx(self.side), None, 0, H, W, strip[0], strip[1], 1, _ptr(owner), None,
            _ptr(interior), None, _ptr(cams),
            _ptr(plan.dev[0]), _ptr(plan.dev[1]), _ptr(plan.dev[2]), *self._lut_args(luts),
            _ptr(mosaic), _ptr(fl), 1),
            "pano_multiband_compose")
        done = torch.cuda.Event()
        done.record(self.side)
        return mosaic, fl, done

    def prepare_blur_async(self, table, W, interior, flags=None):
        """Tile flags and the sorted work list of the blur (they depend on the records and
        the interior map, not on the warped planes) on the side stream, beside the warp.
        Returns (tile flags, event) for ``blur_and_compose(prepared=...)``."""
        torch = _torch()
        if flags is None and interior is not None:
            flags = torch.empty(max(table.n_tiles, 1), dtype=torch.uint8, device=self.device)
        uploaded = torch.cuda.Event()
        uploaded.record(torch.cuda.current_stream(self.device))      # the record table
        self.side.wait_event(uploaded)
        for t in (flags, interior, table.dev):
            if t is not None:
                t.record_stream(self.side)
        _lib.check(self.lib.pano_multiband_blur_prepare(
            self.ctx(self.side), table.ptr, table.n, table.max_aw, table.max_ah, W, _ptr(interior),
            _ptr(flags)), "pano_multiband_blur_prepare")
        listed = torch.cuda.Event()
        listed.record(self.side)
        return flags, listed

    def blur_and_compose(self, table, owner, valid, shape, n_levels, want_float=False,
                         strip=None, interior=None, cams=None, plan=None, luts=None, out=None,
                         prepared=None, mosaic_out=None, classes=None):
        """All Gaussian levels of all patches (n_levels launches), then the gather
        over the mosaic columns ``strip`` (default: all of them).  With an
        ``interior`` map, blur tiles and gathers are skipped where the result is
        the owner's colour (needs ``cams`` with frame pointers and ``plan``)."""
        torch = _torch()
        H, W = shape
        c0, c1 = strip if strip is not None else (0, W)
        taps, ntaps, n_blur, _ = self.blur_tables(n_levels)
        if n_blur:
            if prepared is not None:        # tile flags and work list queued on the side stream
                flags, listed = prepared
                if listed is not None:
                    torch.cuda.current_stream(self.device).wait_event(listed)
            else:
                flags = (torch.empty(max(table.n_tiles, 1), dtype=torch.uint8,
                                     device=self.device) if interior is not None else None)
            _lib.check(self.lib.pano_multiband_blur(
                self.ctx(), table.ptr, table.n, table.max_aw, table.max_vh, table.max_ah,
                _ptr(owner), W, taps.ctypes.data, ntaps, n_blur, _ptr(interior), _ptr(flags)),
                "pano_multiband_blur")
            self.last_tiles = (table, flags)        # for active_tile_pixels (reporting)
        if out is None:
            mosaic = (mosaic_out if mosaic_out is not None else
                      torch.empty((H, W, 3), dtype=torch.uint8, device=self.device))
            fl = (torch.empty((H, W, 3), dtype=torch.float32, device=self.device)
                  if want_float else None)
            part = 0
        else:                       # the interior pixels are being written on the side stream
            mosaic, fl, done = out
            part = 2
        tabs = plan.dev if interior is not None else (None, None, None)
        _lib.check(self.lib.pano_multiband_compose(
            self.ctx(), table.ptr, table.n, H, W, c0, c1, n_levels, _ptr(owner), _ptr(valid),
            _ptr(interior), _ptr(classes) if interior is not None else None,
            _ptr(cams) if interior is not None else None, _ptr(tabs[0]),
            _ptr(tabs[1]), _ptr(tabs[2]), *self._lut_args(luts), _ptr(mosaic), _ptr(fl),
            part), "pano_multiband_compose")
        if out is not None:
            torch.cuda.current_stream(self.device).wait_event(done)
        return mosaic, fl

    def simple_blend(self, patches, shape, linear, table=None):
        torch = _torch()
        H, W = shape
        if table is None:
            table = patch_table(patches, self)
        mosaic = torch.empty((H, W, 3), dtype=torch.uint8, device=self.device)
        fn = self.lib.pano_linear_blend if linear else self.lib.pano_no_blend
        _lib.check(fn(self.ctx(), table.ptr, table.n, H, W, _ptr(mosaic)),
                   "pano_linear_blend" if linear else "pano_no_blend")
        return mosaic

    def _lut_args(self, luts):
        """(lut, lut_stride) of include/pano360.h: one shared table, or one per camera."""
        if luts is None:
            return _ptr(self.lut255), 0
        return _ptr(luts), 256

    # -- exposure ---------------------------------------------------------------
    def equalize_gains(self, frames, rots, intrs, chunk_bytes=256 << 20):
        """equalize_gains up to the gains (stitcher.py:36-65): overlap sizes and
        mean intensities of every camera pair on device, the N x N solve on the
        host.  ``frames``: uint8 [H,W,3] tensors on this device, one size.
        Returns (overlaps, sizes, gains, luts): the reference's two N x N arrays,
        the gains, and the per-camera colour tables (device float32 [N][256])
        that stand for the equalised images of stitcher.py:66."""
        torch = _torch()
        n = len(frames)
        h, w = (int(v) for v in frames[0].shape[:2])
        if any(tuple(f.shape[:2]) != (h, w) for f in frames):
            raise ValueError("equalize_gains: every overlap is sized by regions[0] "
                             "(stitcher.py:41); frames of different sizes cannot be indexed")
        pairs = overlap_pairs(rots, intrs, w, h)
        overlaps, sizes = np.zeros((n, n)), np.zeros((n, n))
        if len(pairs):
            hx, hy = self.hat_tables((h, w))
            rec = np.zeros(n, dtype=CAMERA_DTYPE)
            for k, frame in enumerate(frames):
                rec[k] = (np.zeros(9), frame.data_ptr(), hx.data_ptr(), hy.data_ptr(), h, w,
                          0, 0, 0, 0)
            cams = self.to_device(rec)
            nblk = int(self.lib.pano_overlap_blocks(h, w))
            bw0 = min(1024 // min(16, h), w)
            step = max(1, chunk_bytes // (nblk * 24))
            stats = torch.empty((len(pairs), 3), dtype=torch.float64, device=self.device)
            partials = torch.empty((min(step, len(pairs)), nblk, 3), dtype=torch.float64,
                                   device=self.device)
            for a in range(0, len(pairs), step):
                part = pairs[a:a + step]
                dev_pairs = torch.from_numpy(part.view(np.uint8).reshape(-1)).to(self.device)
                _lib.check(self.lib.pano_overlap_stats(
                    self.ctx(), _ptr(cams), _ptr(dev_pairs), len(part), h, w, bw0,
                    _ptr(self.lut255),
                    _ptr(partials), _ptr(stats[a:])), "pano_overlap_stats")
            host = stats.cpu().numpy()
            for (_, i, j), (count, sum_i, sum_j) in zip(pairs, host):
                sizes[i, j] = sizes[j, i] = count                       # stitcher.py:59
                if count:                                               # :62-63 (float32 means)
                    overlaps[i, j] = np.float32(sum_i / (3.0 * count))
                    overlaps[j, i] = np.float32(sum_j / (3.0 * count))
        gains = find_gains(overlaps, sizes)
        luts = torch.from_numpy(gain_tables(gains)).to(self.device)
        return overlaps, sizes, gains, luts

    # -- fused path: ownership from the cameras, work only near owned pixels ----
    def camera_table(self, plan, frames=None):
        """Device array of ``pano_camera``; ``frames`` maps camera index -> frame
        tensor (cameras without a frame get a NULL pointer)."""
        # what depends on the frames and their sizes only (pointers, hat tables) is kept from
        # stitch to stitch; the projections and rectangles are filled column-wise
        key = (tuple(plan.shapes),
               tuple(sorted((i, f.data_ptr()) for i, f in frames.items())) if frames else ())
        kept = getattr(plan, "_cams", None)
        if kept is not None and kept[0] == key and kept[1] is self:
            return kept[2]                  # same plan object, same frames: the table is up
        if self._cam_template is None or self._cam_template[0] != key:
            rec = np.zeros(plan.n, dtype=CAMERA_DTYPE)
            for i in range(plan.n):
                sh, sw = plan.shapes[i]
                hx, hy = self.hat_tables((sh, sw))
                frame = frames.get(i) if frames else None
                rec[i] = (np.zeros(9), frame.data_ptr() if frame is not None else 0,
                          hx.data_ptr(), hy.data_ptr(), sh, sw, 0, 0, 0, 0)
            self._cam_template = (key, rec)
        rec = self._cam_template[1].copy()
        rec["proj"] = np.asarray(plan.projs, np.float64).reshape(plan.n, 9)
        rects = np.asarray(plan.rects, np.int32)
        rec["y0"], rec["x0"] = rects[:, 0], rects[:, 2]
        rec["h"], rec["w"] = rects[:, 1] - rects[:, 0], rects[:, 3] - rects[:, 2]
        dev = self.to_device(rec)
        plan._cams = (key, self, dev)
        return dev

    def ownership_cameras(self, plan, strip=None, out=None, cams=None):
        """owner / valid of the mosaic (or of the column strip [xs0, xs1)) from the
        cameras alone (stitcher.py:196-204, 266-271 without any pixel data)."""
        torch = _torch()
        H, W = plan.shape
        if out is None:
            owner = torch.empty((H, W), dtype=torch.int16, device=self.device)
            valid = torch.empty((H, W), dtype=torch.uint8, device=self.device)
        else:
            owner, valid = out
        xs0, xs1 = strip if strip is not None else (0, W)
        if cams is None:
            cams = self.camera_table(plan)
        _lib.check(self.lib.pano_ownership_cameras(
            self.ctx(), _ptr(cams), plan.n, H, W, xs0, xs1, _ptr(plan.dev[0]), _ptr(plan.dev[1]),
            _ptr(plan.dev[2]), _ptr(owner), _ptr(valid)),
            "pano_ownership_cameras")
        return owner, valid

    def ownership_regions(self, plan, strip=None, min_gap=0, max_spans=4, cams=None):
        """ownership_cameras and the region search in one pass over the mosaic
        (pano_ownership_regions): owner, valid, and the device arrays regions
        [n][5 + 2 max_spans] and marks [n][W] of pano_owned_regions."""
        torch = _torch()
        H, W = plan.shape
        owner = torch.empty((H, W), dtype=torch.int16, device=self.device)
        valid = torch.empty((H, W), dtype=torch.uint8, device=self.device)
        marks = torch.empty((plan.n, W), dtype=torch.uint8, device=self.device)
        regions = torch.empty((plan.n, 5 + 2 * max_spans), dtype=torch.int32, device=self.device)
        xs0, xs1 = strip if strip is not None else (0, W)
        if cams is None:
            cams = self.camera_table(plan)
        _lib.check(self.lib.pano_ownership_regions(
            self.ctx(), _ptr(cams), plan.n, H, W, xs0, xs1, _ptr(plan.dev[0]), _ptr(plan.dev[1]),
            _ptr(plan.dev[2]), _ptr(owner), _ptr(valid), min_gap, max_spans, _ptr(marks),
            _ptr(regions)), "pano_ownership_regions")
        return owner, valid, regions, marks

    def owned_regions_async(self, owner, n, strip=None, min_gap=0, max_spans=4):
        """Queues the region search and its device->host copy (pinned), returns a function
        that waits for it: host arrays boxes [n][4] = (ymin, ymax, xmin, xmax) and a list,
        per patch, of the inclusive column spans (xa, xb) (runs closer than ``min_gap``
        merged).  That wait is the only sync of a stitch."""
        torch = _torch()
        H, W = owner.shape
        c0, c1 = strip if strip is not None else (0, W)
        marks = torch.empty((n, W), dtype=torch.uint8, device=self.device)
        regions = torch.empty((n, 5 + 2 * max_spans), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.pano_owned_regions(
            self.ctx(), _ptr(owner), H, W, c0, c1, n, min_gap, max_spans,
                                               _ptr(marks), _ptr(regions)),
                   "pano_owned_regions")
        key = (n, max_spans)
        host_buf = self._region_bufs.get(key)
        if host_buf is None:
            host_buf = self._region_bufs[key] = torch.empty(
                (n, 5 + 2 * max_spans), dtype=torch.int32).pin_memory()
        host_buf.copy_(regions, non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))

        def wait():
            done.synchronize()
            host = host_buf.numpy().copy()
            spans = [host[i, 5:5 + 2 * host[i, 4]].reshape(-1, 2) for i in range(n)]
            return host[:, :4], spans
        wait.raw = lambda: (done.synchronize(), host_buf.numpy())[1]     # valid until the next call
        wait.max_spans = max_spans
        return wait

    def owned_regions(self, owner, n, strip=None, min_gap=0, max_spans=4):
        return self.owned_regions_async(owner, n, strip, min_gap, max_spans)()

    def owned_boxes(self, owner, n, strip=None):
        return self.owned_regions(owner, n, strip)[0]

    def multiband_fused(self, frames, plan, n_levels, want_float=False, frame_ids=None,
                        strip=None, shortcut=True, luts=None, mosaic_out=None):
        """The headline path, for the mosaic columns ``strip`` = (c0, c1) (default:
        the whole mosaic).  ``frames[j]`` is the frame of camera ``frame_ids[j]``
        (default: all cameras in order); every camera whose patch reaches within
        two blur radii of the strip must be among them.

        Column strips are independent: the owner map is evaluated on the strip
        grown by the blur radius R (every owned pixel that can reach the strip is
        in there), each patch's rectangle A is cut to the strip and its window V
        grows from that.  Strips therefore shard a stitch over GPUs with no
        exchange inside the blend, and the union of the strips' columns is the
        single-GPU mosaic bit for bit.  ``mosaic_out``: a uint8 [H][W][3] buffer to
        write the strip's columns into instead of a fresh one (other columns untouched)."""
        H, W = plan.shape
        c0, c1 = strip if strip is not None else (0, W)
        taps, ntaps, n_blur, radius = self.blur_tables(n_levels)
        # the owner map is needed one radius past the strip for the windows, and as
        # far as the block-wise interior test looks, so that every strip classifies
        # its pixels exactly as the whole mosaic would
        ib = self.interior_block
        margin = (max(radius, ib * ((radius + 2 * ib - 2) // ib) + ib - 1) if shortcut and n_blur
                  else radius)
        ext = (max(c0 - margin, 0), min(c1 + margin, W))
        ids = list(range(plan.n)) if frame_ids is None else list(frame_ids)
        have = dict(zip(ids, frames))
        cams = self.camera_table(plan, have)
        if self.native_stitch:
            return self._stitch_native(plan, cams, have, n_levels, want_float, (c0, c1), ext,
                                       shortcut, luts, mosaic_out)
        owner, valid = self.ownership_cameras(plan, strip=ext, cams=cams)
        # one record per (patch, span of columns it owns): spans farther apart than
        # 2R keep disjoint rectangles A, so a pixel still meets a patch at most once.
        # The interior map needs the owner map only: queued first, it keeps the GPU busy
        # while the host waits for the regions and lays out the windows.
        regions = self.owned_regions_async(owner, plan.n, ext, 2 * radius + 2)
        interior = classes = None
        if shortcut and n_blur:
            if self.get_option(_lib.OPT_LEVEL_CLASSES):
                interior, classes = self.interior_classes(owner, self.level_radii(n_levels), ext)
            else:
                interior = self.interior_map(owner, radius, ext)
        # The interior pixels of the mosaic need nothing but the owner map: queued now, on
        # the side stream, they fill the GPU while the host waits for the regions and lays
        # out the windows, and run beside the warp.  (Queued behind the warp instead they
        # share the CUs with the blur and slow it by as much as they take: measured.)
        early = (self.compose_interior_async(owner, plan.shape, (c0, c1), interior, cams, plan,
                                             luts, want_float, mosaic_out)
                 if interior is not None and self.overlap_interior else None)
        # the host is on the critical path from here to the warp: one native call lays out
        # the records (rectangles A and V, arena offsets, tile offsets)
        # (everything that does not need the regions is made ready before the wait)
        resident = np.zeros(plan.n, np.uint8)
        resident[[i for i in have if 0 <= i < plan.n]] = 1
        rects32 = np.ascontiguousarray(plan.rects, np.int32)
        records = np.zeros(plan.n * regions.max_spans, dtype=PATCH_DTYPE)
        patches = FusedPatches.from_regions(regions.raw(), regions.max_spans, rects32,
                                            resident, radius, (c0, c1), n_blur, self, records)
        table = patches.table
        # tile flags first (this stream): they tell the warp which blocks of the windows
        # anything will read; the blur's work list then goes to the side stream
        flags = need = None
        # (worth it when the rectangles are wide against the blur's reach of ~3 tiles either
        # side of a seam: 8 x 1080p -4 %; on 32 x 4K nearly every tile is within reach and the
        # two small kernels in front of the warp cost more than they save)
        wide = table.n and float(np.mean(table.host["aw"])) >= 768.0
        choice = self.warp_need
        if (interior is not None and n_blur and self.tile_grid == 32
                and (choice is True or (choice == "auto" and wide))):
            torch = _torch()
            flags = torch.empty(max(table.n_tiles, 1), dtype=torch.uint8, device=self.device)
            need = torch.empty(max(table.n_tiles, 1), dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.pano_blur_tiles(
                self.ctx(), table.ptr, table.n, table.max_aw, table.max_ah, plan.shape[1], radius,
                _ptr(interior), _ptr(flags), _ptr(need)), "pano_blur_tiles")
        prepared = (self.prepare_blur_async(table, plan.shape[1], interior, flags)
                    if n_blur and self.overlap_prepare else
                    ((flags, None) if flags is not None else None))
        _lib.check(self.lib.pano_warp_windows(
            self.ctx(), _ptr(cams), table.ptr, table.n, table.max_vw, table.max_vh,
            _ptr(plan.dev[0]),
            _ptr(plan.dev[1]), _ptr(plan.dev[2]), *self._lut_args(luts), _ptr(need)), "pano_warp_windows")
        mosaic, fl = self.blur_and_compose(table, owner, valid, plan.shape, n_levels,
                                           want_float, (c0, c1), interior, cams, plan, luts,
                                           out=early, prepared=prepared, mosaic_out=mosaic_out,
                                           classes=classes)
        self.last_classes, self.last_interior = classes, interior
        return mosaic, fl, valid, patches

    def _stitch_workspace(self, H, W, n, max_spans):
        """Buffers of the native stitch that no caller sees, kept per mosaic shape."""
        torch = _torch()
        key = (H, W, n, max_spans)
        ws = self._stitch_ws.get(key)
        if ws is None:
            if len(self._stitch_ws) >= 4:
                self._stitch_ws.clear()
            ib = self.interior_block
            shape8 = ((H + ib - 1) // ib, (W + ib - 1) // ib)
            dev = self.device
            stride = 5 + 2 * max_spans
            cap = n * max_spans
            ws = dict(
                marks=torch.empty((n, W), dtype=torch.uint8, device=dev),
                regions=torch.empty((n, stride), dtype=torch.int32, device=dev),
                regions_host=torch.empty((n, stride), dtype=torch.int32).pin_memory(),
                bown=torch.empty((2,) + shape8, dtype=torch.int16, device=dev),
                # (zeros: blocks outside the strip a rank works on stay class 0 for gather_bytes)
                interior=torch.zeros(shape8, dtype=torch.uint8, device=dev),
                classes=torch.zeros(shape8, dtype=torch.uint8, device=dev),
                records_host=torch.empty(cap * PATCH_DTYPE.itemsize, dtype=torch.uint8).pin_memory(),
                table=torch.empty(cap * PATCH_DTYPE.itemsize, dtype=torch.uint8, device=dev),
                cap=cap, tiles=None, need=None, cap_tiles=0, have=np.zeros(n, np.uint8),
                args=_lib.StitchArgs())
            self._stitch_ws[key] = ws
        return ws

    def _stitch_native(self, plan, cams, have, n_levels, want_float, strip, ext, shortcut, luts,
                       mosaic_out):
        """``multiband_fused`` through ``pano_stitch_multiband``: one native call queues the
        whole stitch (and waits once, for the owned regions)."""
        torch = _torch()
        H, W = plan.shape
        taps, ntaps, n_blur, radius = self.blur_tables(n_levels)
        max_spans = 4
        ws = self._stitch_workspace(H, W, plan.n, max_spans)
        a = ws["args"]
        mosaic = (mosaic_out if mosaic_out is not None else
                  torch.empty((H, W, 3), dtype=torch.uint8, device=self.device))
        fl = (torch.empty((H, W, 3), dtype=torch.float32, device=self.device)
              if want_float else None)
        resident = ws["have"]
        resident[:] = 0
        resident[[i for i in have if 0 <= i < plan.n]] = 1
        rects32 = np.ascontiguousarray(plan.rects, np.int32)
        lut, lut_stride = self._lut_args(luts)
        ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        a.cams, a.rects, a.have = ptr(cams), rects32.ctypes.data, resident.ctypes.data
        a.sin_t, a.cos_t, a.tan_p = (ptr(t) for t in plan.dev)
        a.lut, a.lut_stride = lut, lut_stride
        a.taps, a.ntaps = taps.ctypes.data, C.cast(ntaps, C.c_void_p)
        a.mosaic, a.mosaic_f32 = ptr(mosaic), ptr(fl)
        a.marks, a.regions, a.regions_host = ptr(ws["marks"]), ptr(ws["regions"]), ptr(ws["regions_host"])
        a.block_owner, a.interior = ptr(ws["bown"]), ptr(ws["interior"])
        a.classes = ptr(ws["classes"])
        a.records_host, a.table, a.cap_records = ptr(ws["records_host"]), ptr(ws["table"]), ws["cap"]
        a.n, a.H, a.W = plan.n, H, W
        a.xs0, a.xs1, a.own0, a.own1 = strip[0], strip[1], ext[0], ext[1]
        a.n_levels, a.radius, a.shortcut = n_levels, radius, 1 if shortcut else 0
        a.warp_need = {True: 1, False: 0}.get(self.warp_need, -1)
        a.max_spans, a.min_gap = max_spans, 2 * radius + 2
        # a repeat of the verified stitch (same Plan object and frames set, same strip and form)?
        sig = (id(plan), strip, ext, n_levels, bool(shortcut), fl is not None, luts is None,
               tuple(sorted(have)), tuple(id(self._arenas.get(k)) for k in ("planes", "blurred", "scratch")),
               id(ws["tiles"]))
        kept = self._trusted
        repeat = (self.trust_layout and kept is not None and kept[0] == sig and kept[1] is plan)
        a.trust_layout = (3 if self.keep_geometry else 1) if repeat else 0
        if repeat and self.keep_geometry:
            owner, valid = kept[3], kept[4]         # where the verified stitch left them
        else:
            owner = torch.empty((H, W), dtype=torch.int16, device=self.device)
            valid = torch.empty((H, W), dtype=torch.uint8, device=self.device)
        a.owner, a.valid = ptr(owner), ptr(valid)
        resume = 0
        a.layout.missing = a.layout.n_records = 0   # (an early failure must not read a previous stitch's)
        while True:
            arenas = [self._arenas.get(k) for k in ("planes", "blurred", "scratch")]
            for k, t in zip(("planes", "blurred", "scratch"), arenas):
                setattr(a, k, ptr(t))
                setattr(a, k + "_floats", t.numel() - 64 if t is not None else 0)
            a.tile_flags, a.need, a.cap_tiles = ptr(ws["tiles"]), ptr(ws["need"]), ws["cap_tiles"]
            status = self.lib.pano_stitch_multiband(self.ctx(), C.byref(a), resume)
            if status != _lib.EGROW:
                break
            lay = a.layout                  # grow what is too small, then resume behind the wait
            self.arena("planes", int(lay.planes_floats) + 64)
            self.arena("blurred", int(lay.blurred_floats) + 64)
            self.arena("scratch", int(lay.scratch_floats) + 64)
            if lay.n_tiles > ws["cap_tiles"]:
                ws["cap_tiles"] = int(lay.n_tiles * 1.25) + 64
                ws["tiles"] = torch.empty(ws["cap_tiles"], dtype=torch.uint8, device=self.device)
                ws["need"] = torch.empty(ws["cap_tiles"], dtype=torch.uint8, device=self.device)
            resume = 1
        if status != 0:
            if status == _lib.EINVAL and a.layout.missing:
                rec = ws["records_host"].numpy().view(PATCH_DTYPE)[:a.layout.n_records]
                missing = sorted({int(i) for i in rec["index"] if not resident[int(i)]})
                raise _lib.PanoError(f"frames {missing} are needed for columns [{strip[0]}, "
                                     f"{strip[1]}) but are not resident on this device")
            _lib.check(status, "pano_stitch_multiband")
        lay = a.layout
        if a.trust_layout in (2, 4):
            # queued with the verified layout, nobody waited: the records are the previous stitch's
            # (4: and so are the owner map, the masks and the work list - nothing was recomputed)
            patches = kept[2]
            if self.keep_geometry:
                self._trusted = kept[:3] + (owner, valid)
            self.last_kept_geometry = a.trust_layout == 4
        else:
            rec = ws["records_host"].numpy().view(PATCH_DTYPE)[:lay.n_records].copy()
            patches = FusedPatches.from_records(
                rec, lay, ws["table"][:lay.n_records * PATCH_DTYPE.itemsize],
                tuple(self._arenas.get(k) for k in ("planes", "blurred", "scratch")))
            # (the signature holds the arenas and tile buffers as they are AFTER this stitch: a
            # stitch that grew one of them is not repeated blindly)
            sig = sig[:-2] + (tuple(id(self._arenas.get(k)) for k in ("planes", "blurred", "scratch")),
                              id(ws["tiles"]))
            self._trusted = ((sig, plan, patches) + ((owner, valid) if self.keep_geometry else (None, None))
                             if self.trust_layout else None)
            self.last_kept_geometry = False
        if n_blur:
            self.last_tiles = (patches.table, ws["tiles"] if shortcut else None)
        self.last_classes = (ws["classes"] if shortcut and n_blur
                             and self.get_option(_lib.OPT_LEVEL_CLASSES) else None)
        self.last_interior = ws["interior"] if shortcut and n_blur else None
        return mosaic, fl, valid, patches

    def trust_layouts(self, on=True, keep_geometry=False):
        """Switches trusted stitches on (with the device-side layout they ride on) or off;
        ``keep_geometry``: repeats also re-use the previous stitch's owner map, masks, record
        table and work list (see ``__init__``)."""
        self.trust_layout = bool(on)
        self.keep_geometry = bool(on and keep_geometry)
        self._trusted = None
        self.set_option(_lib.OPT_STITCH_ASYNC, 1 if on else 0)
        return self

    def verify_trusted(self):
        """Compares the layout the device made for the last trusted stitch with the verified one
        it was queued with (``pano_stitch_verify``; waits for that stitch's layout kernel).
        Raises ``PanoError`` if they differ: the cameras were changed under a kept Plan."""
        _lib.check(self.lib.pano_stitch_verify(self._ctx), "pano_stitch_verify")

    def blend_fused(self, frames, plan, linear, frame_ids=None, strip=None, luts=None):
        """linear_blend / no_blend of the mosaic columns ``strip`` straight from
        the frames (no patch buffers).  Returns (mosaic u8, valid u8)."""
        torch = _torch()
        H, W = plan.shape
        c0, c1 = strip if strip is not None else (0, W)
        ids = list(range(plan.n)) if frame_ids is None else list(frame_ids)
        have = dict(zip(ids, frames))
        missing = [i for i, (_, _, x0, x1) in enumerate(plan.rects)
                   if x0 < c1 and x1 > c0 and i not in have]
        if missing:
            raise _lib.PanoError(f"frames {missing} are needed for columns [{c0}, {c1}) "
                                 "but are not resident on this device")
        cams = self.camera_table(plan, have)
        mosaic = torch.empty((H, W, 3), dtype=torch.uint8, device=self.device)
        valid = torch.empty((H, W), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.pano_blend_cameras(
            self.ctx(), _ptr(cams), plan.n, H, W, c0, c1, 1 if linear else 0, _ptr(plan.dev[0]),
            _ptr(plan.dev[1]), _ptr(plan.dev[2]), *self._lut_args(luts), _ptr(mosaic),
            _ptr(valid)), "pano_blend_cameras")
        return mosaic, valid

    # -- crop and filters -------------------------------------------------------------
    def crop_rect(self, valid):
        """Rectangle (y0, x0, h, w) of crop_mosaic, or None when nothing is valid."""
        torch = _torch()
        H, W = valid.shape
        heights = torch.empty((H, W), dtype=torch.int32, device=self.device)
        result = torch.zeros(6, dtype=torch.int64, device=self.device)
        _lib.check(self.lib.pano_crop_rect(
            self.ctx(), _ptr(valid), H, W, _ptr(heights), _ptr(result)), "pano_crop_rect")
        res = result.cpu().numpy()
        if not res[0]:
            return None
        return tuple(int(v) for v in res[1:5])

    def plane_taps(self, ksize, sigma):
        """Padded host tap table of one Gaussian, cached per (ksize, sigma); the context
        keeps the device copy."""
        key = ("plane", int(ksize), float(sigma))
        if key not in self._taps:
            self._taps[key] = np.ascontiguousarray(padded_taps(gaussian_taps(ksize, sigma)))
        return self._taps[key]

    def blur_plane(self, plane, ksize, sigma, out=None):
        """cv2.GaussianBlur on one float32 plane [h][w] already on device.  Rows that
        are a multiple of 4 floats are filtered in place of any padded copy."""
        torch = _torch()
        h, w = plane.shape
        pitch = (w + 3) & ~3
        if pitch == w and plane.is_contiguous():
            src = plane
        else:
            src = torch.zeros((h, pitch), dtype=torch.float32, device=self.device)
            src[:, :w] = plane
        dst = out if out is not None and pitch == w else torch.empty_like(src)
        tmp = torch.empty_like(src)
        _lib.check(self.lib.pano_blur_plane(
            self.ctx(), _ptr(src), _ptr(dst), _ptr(tmp), h, w, pitch,
            self.plane_taps(ksize, sigma).ctypes.data, ksize), "pano_blur_plane")
        return dst[:, :w]

    def pyr_down(self, plane):
        torch = _torch()
        plane = plane.contiguous()
        h, w = plane.shape
        out = torch.empty(((h + 1) // 2, (w + 1) // 2), dtype=torch.float32,
                          device=self.device)
        _lib.check(self.lib.pano_pyr_down(self.ctx(), _ptr(plane), h, w, _ptr(out)),
                   "pano_pyr_down")
        return out

    # -- whole stitch -----------------------------------------------------------
    def stitch(self, frames, plan, blend="multiband", n_levels=5, want_float=False,
               fused=True, shortcut=True, luts=None):
        """uint8 frames on device -> (mosaic u8 on device, float mosaic, valid,
        patches).  ``fused=False`` runs multiband through whole-patch stage
        buffers (what the blender protocol sees); both give the same mosaic."""
        if not hasattr(plan, "dev"):
            self.upload_plan(plan)
        if blend == "multiband" and fused:
            return self.multiband_fused(frames, plan, n_levels, want_float, shortcut=shortcut,
                                        luts=luts)
        if fused:
            mosaic, valid = self.blend_fused(frames, plan, blend == "linear", luts=luts)
            return mosaic, None, valid, []
        n_blur = n_levels - 1 if blend == "multiband" else 0
        patches, _ = self.warp_all(frames, plan, n_blur, luts=luts)
        table = patch_table(patches, self)
        if blend == "multiband":
            mosaic, fl, _, valid = self.multiband(patches, plan.shape, n_levels, want_float,
                                                  table)
            return mosaic, fl, valid, patches
        mosaic = self.simple_blend(patches, plan.shape, blend == "linear", table)
        return mosaic, None, None, patches


_engine = None


def engine():
    """Process-wide engine on the current device."""
    global _engine
    if _engine is None:
        _engine = Engine()
    return _engine
