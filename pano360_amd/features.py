"""The Gaussian / pyramid part of the reference ``features.py`` on the GPU.

* ``gaussian_filter(img, sigma=1.0)``  - features.py:20-24
* ``pyr_down`` / ``gaussian_pyramid``   - the ``cv2.pyrDown`` chain of the MSOP
  detector, features.py:138-155
* ``sift_pyramid(img)``                 - the Gaussian and difference-of-Gaussian
  scale space that ``sift_detector`` (features.py:192-201) gets from
  ``cv2.xfeatures2d.SIFT_create().detectAndCompute``.

All filters run in ``libpano360_hip.so`` (``pano_blur_plane``, ``pano_pyr_down``,
``pano_gray_u8``, ``pano_resize_up2``, ``pano_decimate2``, ``pano_scale_step``).
The SIFT arithmetic is inside OpenCV, not in the reference repo: its published
algorithm (SIFT defaults: sigma 1.6, 3 layers per octave, first octave -1) is
restated, parity unpinned.
* ``sift_detector()``                   - features.py:192-201: keypoints, 128-d
  descriptors (``pano_sift_extrema`` / ``_orient`` / ``_describe``) and the RootSIFT
  normalisation.
* ``flann_matching(des1, des2)``        - features.py:222-232: the 2-nearest-neighbour
  search and Lowe's 0.7 ratio test, exhaustive and exact on the GPU (``pano_knn2``:
  matrix-core cross terms rank the candidates, the winners are re-evaluated in float32)
  where the reference asks FLANN's randomised kd-trees for an approximate answer;
  homography estimation (RANSAC) stays outside.
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
        _lib.check(self.lib.pano_gray_u8(self.eng.ctx(), _eng._ptr(frame), h, w, _eng._ptr(out)),
                   "pano_gray_u8")
        return out

    def up2(self, plane):
        h, w = plane.shape
        out = self._new(2 * h, 2 * w)
        _lib.check(self.lib.pano_resize_up2(self.eng.ctx(), _eng._ptr(plane), h, w,
                                            _eng._ptr(out)), "pano_resize_up2")
        return out

    def half(self, plane):
        h, w = plane.shape
        out = self._new(h // 2, w // 2)
        _lib.check(self.lib.pano_decimate2(self.eng.ctx(), _eng._ptr(plane), h, w,
                                           _eng._ptr(out)), "pano_decimate2")
        return out

    def sub(self, a, b):
        out = self._new(*a.shape)
        _lib.check(self.lib.pano_subtract(self.eng.ctx(), _eng._ptr(a), _eng._ptr(b),
                                          C.c_size_t(a.numel()), _eng._ptr(out)), "pano_subtract")
        return out

    def blur(self, plane, sigma):
        return self.eng.blur_plane(plane, _eng.gaussian_ksize(sigma), sigma).contiguous()


_STEP_TAPS = {}


def _step_taps(sigma):
    """cv::getGaussianKernel(cvRound(8 sigma + 1) | 1, sigma) as a host float32 array."""
    key = float(sigma)
    if key not in _STEP_TAPS:
        _STEP_TAPS[key] = np.ascontiguousarray(_eng.gaussian_taps(_eng.gaussian_ksize(key), key))
    return _STEP_TAPS[key]


def sift_pyramid_device(frame, n_octaves=None, sigma=SIFT_SIGMA, layers=SIFT_LAYERS, eng=None):
    """Gaussian and DoG pyramids of a uint8 BGR frame already on the device.
    Returns (gauss, dog): lists over octaves of contiguous stacks
    [layers+3][h][w] / [layers+2][h][w] (index them like lists of planes).
    One native call (``pano_scale_space``) queues every launch of the frame; one launch per
    layer (``scale_step_kernel``) blurs layer i-1 into layer i - both passes, the row-pass
    image staying in LDS - and writes the DoG layer i-1 from the same tile."""
    import torch
    eng = eng or _eng.engine()
    h, w = (int(v) for v in frame.shape[:2])
    if n_octaves is None:
        n_octaves = sift_octaves(h, w)
    # createInitialImage: grey -> float -> 2x bilinear -> blur to sigma
    sig_diff = float(np.sqrt(max(np.float32(sigma) ** 2 - np.float32(SIFT_INIT_SIGMA) ** 2 * 4,
                                 np.float32(0.01))))
    kernels = [_step_taps(s) for s in [sig_diff] + sift_sigmas(sigma, layers)[1:]]
    dims, rows, cols = [], 2 * h, 2 * w
    for o in range(n_octaves):
        dims.append((rows, cols))
        if min(rows, cols) < 2:             # buildGaussianPyramid would halve it to nothing
            break
        rows, cols = rows // 2, cols // 2
    f32 = dict(dtype=torch.float32, device=eng.device)
    gauss = [torch.empty((layers + 3, r, c), **f32) for r, c in dims]
    dog = [torch.empty((layers + 2, r, c), **f32) for r, c in dims]
    work = torch.empty(5 * h * w, **f32)
    taps = np.ascontiguousarray(np.concatenate(kernels), np.float32)
    ntaps = (C.c_int * len(kernels))(*[len(k) for k in kernels])
    gptr = (C.c_void_p * len(dims))(*[g.data_ptr() for g in gauss])
    dptr = (C.c_void_p * len(dims))(*[d.data_ptr() for d in dog])
    _lib.check(eng.lib.pano_scale_space(eng.ctx(), _eng._ptr(frame), h, w, len(dims), layers,
                                        taps.ctypes.data, ntaps, gptr, dptr, _eng._ptr(work)),
               "pano_scale_space")
    return gauss, dog


# ------------------------------------------------------- SIFT keypoints, descriptors
SIFT_CONTRAST = 0.04        # cv2.xfeatures2d.SIFT_create() defaults
SIFT_EDGE = 10.0
SIFT_FIRST_OCTAVE = -1
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("r", "<i4"), ("c", "<i4")])
assert KP_DTYPE.itemsize == C.sizeof(_lib.SiftKeypoint) == 32


class KeyPoint:
    """The fields of ``cv2.KeyPoint`` the reference reads (features.py:223-232: ``pt``)."""
    __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

    def __init__(self, x, y, size, angle=-1.0, response=0.0, octave=0, class_id=-1):
        self.pt = (float(x), float(y))
        self.size, self.angle, self.response = float(size), float(angle), float(response)
        self.octave, self.class_id = int(octave), int(class_id)

    def __repr__(self):
        return (f"KeyPoint(pt=({self.pt[0]:.2f}, {self.pt[1]:.2f}), size={self.size:.2f}, "
                f"angle={self.angle:.1f}, octave={self.octave})")


def sift_sort_unique(kps):
    """KeyPointsFilter::removeDuplicatedSorted on a KP_DTYPE array: order by x, y,
    size (descending), angle, response (descending), octave (descending); keep the
    first of keypoints that share x, y, size and angle."""
    order = np.lexsort((-kps["octave"].astype(np.int64), -kps["response"], kps["angle"],
                        -kps["size"], kps["y"], kps["x"]))
    kps = kps[order]
    if len(kps) > 1:
        same = ((kps["x"][1:] == kps["x"][:-1]) & (kps["y"][1:] == kps["y"][:-1]) &
                (kps["size"][1:] == kps["size"][:-1]) & (kps["angle"][1:] == kps["angle"][:-1]))
        kps = kps[np.concatenate([[True], ~same])]
    return kps


class _SiftHost:
    """Per-engine host side of the detector: the copy stream, the pinned staging buffer of
    the keypoints and a ring of pinned counter slots.  Engines (one per host thread) share
    nothing; a lock orders the detections of one engine that finish from several threads."""

    SLOTS = 16

    def __init__(self, eng):
        import threading
        import torch
        self.lock = threading.Lock()
        self.side = torch.cuda.Stream(eng.device)
        self.pinned = None
        self.counts = torch.empty((self.SLOTS, 3), dtype=torch.int32).pin_memory()
        self.slot_events = [None] * self.SLOTS
        self.next = 0

    def counter_slot(self):
        """A pinned int32[3] that no queued copy still writes."""
        with self.lock:
            k = self.next
            self.next = (k + 1) % self.SLOTS
            if self.slot_events[k] is not None:
                self.slot_events[k].synchronize()
            return k, self.counts[k]

    def staging(self, nbytes):
        import torch
        if self.pinned is None or self.pinned.numel() < nbytes:
            self.pinned = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8).pin_memory()
        return self.pinned


def _sift_host(eng):
    host = getattr(eng, "_sift_host", None)
    if host is None:
        host = eng._sift_host = _SiftHost(eng)
    return host


class SiftDetection:
    """A ``detectAndCompute`` queued on the device: nothing has been waited for yet.
    ``result()`` waits, checks the counters and returns (keypoints as a KP_DTYPE array in
    OpenCV's order, descriptors float32 [K][128] on the device, values 0..255)."""

    def __init__(self, counts, kpts, desc, max_keypoints, keep, eng):
        import torch
        self.counts, self.kpts, self.desc, self.max_keypoints = counts, kpts, desc, max_keypoints
        self.keep = keep                    # buffers the queued kernels still read
        self._out = None
        self.host = _sift_host(eng)
        # the counters travel to pinned memory (a slot of the engine's ring) behind this frame's
        # kernels; `done` marks that point, so result() waits for THIS frame only, not for
        # whatever was queued after it
        slot, self.host_counts = self.host.counter_slot()
        self.host_counts.copy_(counts, non_blocking=True)
        self.done = torch.cuda.Event()
        self.done.record(torch.cuda.current_stream(counts.device))
        self.host.slot_events[slot] = self.done

    def result(self):
        import torch
        if self._out is None:
            self.done.synchronize()
            n_cand, n_kp, n_out = (int(v) for v in self.host_counts.numpy())
            if max(n_cand, n_kp) > self.max_keypoints:
                raise _lib.PanoError(f"sift: {max(n_cand, n_kp)} keypoints exceed max_keypoints")
            # the keypoints on a stream of their own: a copy on the compute stream would queue
            # behind the next frame's kernels.  One staging buffer per engine: the lock keeps
            # two results of this engine from sharing it
            with self.host.lock:
                pinned = self.host.staging(n_out * 32)
                with torch.cuda.stream(self.host.side):   # pinned: a DMA, no staging kernel
                    pinned[:n_out * 32].copy_(self.kpts[:n_out * 32], non_blocking=True)
                self.host.side.synchronize()
                kps = pinned[:n_out * 32].numpy().view(KP_DTYPE).copy()
            self._out = (kps, self.desc[:n_out])
            self.keep = None
        return self._out


class SiftPipeline:
    """Frames of ONE size through the detector's front end on one engine, a native call per frame
    (``pano_sift_detect``: scale space, extrema, orientations, OpenCV's order, descriptors - about
    110 launches - queued from C++ and, from the second frame of a set of buffers on, replayed as
    ONE HIP graph).  A graph holds addresses, so every buffer of a frame - the pyramid, the
    keypoint lists, the descriptors - lives in one of ``depth`` workspaces used in turn: what
    ``pyramid()`` / ``detect()`` hand back stays valid until ``depth`` more frames have been queued
    on this pipeline (take a detection's ``result()`` before that)."""

    def __init__(self, eng, h, w, depth=3, max_keypoints=1 << 18, n_octaves=None,
                 sigma=SIFT_SIGMA, layers=SIFT_LAYERS):
        import torch
        self.eng, self.h, self.w, self.depth = eng, int(h), int(w), max(int(depth), 1)
        self.max_keypoints, self.layers, self.sigma = int(max_keypoints), layers, sigma
        if n_octaves is None:
            n_octaves = sift_octaves(self.h, self.w)
        sig_diff = float(np.sqrt(max(np.float32(sigma) ** 2 - np.float32(SIFT_INIT_SIGMA) ** 2 * 4,
                                     np.float32(0.01))))
        kernels = [_step_taps(s) for s in [sig_diff] + sift_sigmas(sigma, layers)[1:]]
        self.taps = np.ascontiguousarray(np.concatenate(kernels), np.float32)
        self.ntaps = (C.c_int * len(kernels))(*[len(k) for k in kernels])
        self.dims, rows, cols = [], 2 * self.h, 2 * self.w
        for _ in range(n_octaves):
            self.dims.append((rows, cols))
            if min(rows, cols) < 2:         # buildGaussianPyramid would halve it to nothing
                break
            rows, cols = rows // 2, cols // 2
        self.slots, self.next = [], 0
        self._torch = torch

    def _slot(self):
        """The next workspace in turn (made on first use: 2.3 GB for a 4K frame)."""
        torch, eng, dev = self._torch, self.eng, self.eng.device
        k = self.next % self.depth
        self.next += 1
        if k < len(self.slots):
            return self.slots[k]
        f32 = dict(dtype=torch.float32, device=dev)
        ws = {}
        ws["gauss"] = [torch.empty((self.layers + 3, r, c), **f32) for r, c in self.dims]
        ws["dog"] = [torch.empty((self.layers + 2, r, c), **f32) for r, c in self.dims]
        ws["work"] = torch.empty(5 * self.h * self.w, **f32)
        ws["frame"] = torch.empty((self.h, self.w, 3), dtype=torch.uint8, device=dev)
        n = self.max_keypoints
        ws["cands"] = torch.empty(n * 32, dtype=torch.uint8, device=dev)
        ws["kpts"] = torch.empty(n * 32, dtype=torch.uint8, device=dev)
        ws["counts"] = torch.zeros(3, dtype=torch.int32, device=dev)
        ws["sort"] = torch.empty(int(eng.lib.pano_sift_sort_work_bytes(n)), dtype=torch.uint8,
                                 device=dev)
        ws["desc"] = torch.empty((n, 128), **f32)
        ws["gptr_host"] = (C.c_void_p * len(self.dims))(*[g.data_ptr() for g in ws["gauss"]])
        ws["dptr_host"] = (C.c_void_p * len(self.dims))(*[d.data_ptr() for d in ws["dog"]])
        # 256 entries each - one per value of a keypoint's octave byte, zeros beyond the pyramid:
        # a keypoint record that is not this frame's (octave byte 255 after the first-octave
        # adjustment) then finds an empty plane and samples nothing instead of a wild address
        dims_tab = np.zeros((256, 2), np.int32)
        dims_tab[:len(self.dims)] = self.dims
        gptr_tab = np.zeros(256, np.int64)
        gptr_tab[:len(self.dims)] = [g.data_ptr() for g in ws["gauss"]]
        ws["dims_dev"] = torch.from_numpy(dims_tab.reshape(-1)).to(dev)
        ws["gptr_dev"] = torch.from_numpy(gptr_tab).to(dev)
        a = _lib.SiftArgs()
        a.frame_copy = ws["frame"].data_ptr()
        a.h, a.w, a.n_octaves, a.n_layers = self.h, self.w, len(self.dims), self.layers
        a.taps, a.ntaps = self.taps.ctypes.data, C.cast(self.ntaps, C.c_void_p)
        a.gauss, a.dog = C.cast(ws["gptr_host"], C.c_void_p), C.cast(ws["dptr_host"], C.c_void_p)
        a.work = ws["work"].data_ptr()
        a.contrast_thr, a.edge_thr, a.sigma = SIFT_CONTRAST, SIFT_EDGE, self.sigma
        a.first_octave, a.max_keypoints = SIFT_FIRST_OCTAVE, n
        a.gauss_dev, a.dims_dev = ws["gptr_dev"].data_ptr(), ws["dims_dev"].data_ptr()
        a.cands, a.kpts = ws["cands"].data_ptr(), ws["kpts"].data_ptr()
        a.counts, a.sort_work, a.desc = (ws["counts"].data_ptr(), ws["sort"].data_ptr(),
                                         ws["desc"].data_ptr())
        ws["args"] = a
        self.slots.append(ws)
        return ws

    def _queue(self, frame, detect):
        if tuple(frame.shape) != (self.h, self.w, 3) or not frame.is_contiguous():
            raise ValueError(f"SiftPipeline of {self.h} x {self.w} frames got {tuple(frame.shape)}")
        ws = self._slot()
        a = ws["args"]
        a.frame, a.detect = frame.data_ptr(), 1 if detect else 0
        _lib.check(self.eng.lib.pano_sift_detect(self.eng.ctx(), C.byref(a)), "pano_sift_detect")
        ws["last_frame"] = frame            # (queued kernels read it)
        return ws

    @property
    def replaying(self):
        """True once the frames of the workspace used last go out as one graph launch."""
        return bool(self.eng.lib.pano_sift_detect_replaying(self.eng._ctx))

    def pyramid(self, frame):
        """(gauss, dog) of ``sift_pyramid_device``, in this pipeline's buffers."""
        ws = self._queue(frame, False)
        return ws["gauss"], ws["dog"]

    def detect(self, frame):
        """A queued ``detectAndCompute``: ``SiftDetection`` (``result()`` waits)."""
        ws = self._queue(frame, True)
        det = SiftDetection(ws["counts"], ws["cands"], ws["desc"], self.max_keypoints, (ws,), self.eng)
        det.pyramid = (ws["gauss"], ws["dog"])
        return det


def _pipeline_for(eng, h, w, max_keypoints):
    """The engine's pipeline for frames of this size (a few sizes are kept)."""
    kept = getattr(eng, "_sift_pipelines", None)
    if kept is None:
        kept = eng._sift_pipelines = {}
    key = (int(h), int(w), int(max_keypoints))
    if key not in kept:
        if len(kept) >= 2:
            kept.pop(next(iter(kept)))
        kept[key] = SiftPipeline(eng, h, w, depth=3, max_keypoints=max_keypoints)
    return kept[key]


def sift_detect_async(frame, max_keypoints=1 << 18, pyramid=None, eng=None):
    """detectAndCompute of a uint8 BGR frame on the device, queued without a single wait: the
    candidate and keypoint counts stay on the device (``n_dev`` of ``pano_sift_sort_unique`` /
    ``pano_sift_describe``), so consecutive frames follow each other on the GPU.  (With a wait
    for every counter a 4K frame took 12.4 ms for 7.1 ms of kernels.)  ``pyramid`` =
    (gauss, dog) device stacks replaces the scale space of ``frame``."""
    import torch
    eng = eng or _eng.engine()
    lib = eng.lib
    if pyramid is None:
        # the whole frame in one native call (a HIP graph from the second frame of a workspace on);
        # the result lives in the engine's pipeline of this frame size: valid for three frames
        h, w = (int(v) for v in frame.shape[:2])
        return _pipeline_for(eng, h, w, max_keypoints).detect(frame.contiguous())
    gauss, dog = pyramid
    dev = eng.device
    dims_host = np.array([v for g in gauss for v in g.shape[1:]], np.int32)
    gptr_host = np.array([g.data_ptr() for g in gauss], np.int64)
    dims = eng.to_device(dims_host).view(torch.int32)
    gptr = eng.to_device(gptr_host).view(torch.int64)
    cands = torch.empty(max_keypoints * 32, dtype=torch.uint8, device=dev)
    kpts = torch.empty(max_keypoints * 32, dtype=torch.uint8, device=dev)
    counts = torch.zeros(3, dtype=torch.int32, device=dev)    # candidates, keypoints, kept
    for o, diff in enumerate(dog):
        _, oh, ow = diff.shape
        _lib.check(lib.pano_sift_extrema(eng.ctx(), _eng._ptr(diff), oh, ow, o, SIFT_LAYERS,
                                         SIFT_CONTRAST, SIFT_EDGE, SIFT_SIGMA, _eng._ptr(cands),
                                         _eng._ptr(counts[0:]), max_keypoints),
                   "pano_sift_extrema")
    _lib.check(lib.pano_sift_orient(eng.ctx(), _eng._ptr(gptr), _eng._ptr(dims), SIFT_LAYERS,
                                    _eng._ptr(cands), _eng._ptr(counts[0:]), max_keypoints,
                                    _eng._ptr(kpts), _eng._ptr(counts[1:]), max_keypoints),
               "pano_sift_orient")
    # OpenCV's order and duplicate removal, and the first-octave adjustment (positions and
    # sizes halved, octave byte shifted: sift.cpp, detectAndCompute), on the device: the
    # six-key lexsort took 54 of a 4K frame's 69 ms on the host.  The capacity is sorted; the
    # slots past the device-side count sort to the end.
    work = torch.empty(int(lib.pano_sift_sort_work_bytes(max_keypoints)), dtype=torch.uint8,
                       device=dev)
    _lib.check(lib.pano_sift_sort_unique(eng.ctx(), _eng._ptr(kpts), max_keypoints,
                                         _eng._ptr(counts[1:]), SIFT_FIRST_OCTAVE, _eng._ptr(work),
                                         _eng._ptr(cands), _eng._ptr(counts[2:])),
               "pano_sift_sort_unique")                      # cands: free again, reused as output
    desc = torch.empty((max_keypoints, 128), dtype=torch.float32, device=dev)
    _lib.check(lib.pano_sift_describe(eng.ctx(), _eng._ptr(gptr), _eng._ptr(dims),
                                      SIFT_FIRST_OCTAVE, _eng._ptr(cands), max_keypoints,
                                      _eng._ptr(counts[2:]), _eng._ptr(desc)), "pano_sift_describe")
    return SiftDetection(counts, cands, desc, max_keypoints, (gauss, dog, dims, gptr, kpts, work),
                         eng)


def sift_detect_device(frame, max_keypoints=1 << 18, pyramid=None, eng=None):
    """``sift_detect_async(...).result()``."""
    return sift_detect_async(frame, max_keypoints, pyramid, eng).result()


def sift_detector():
    """Closure, return a SIFT detecting function (features.py:192-201):
    ``_detect(img) -> (keypoints, RootSIFT descriptors)``."""
    def _detect(img):
        eng = _eng.engine()
        frame = eng.upload_frames([img])[0]
        kps, desc = sift_detect_device(frame)
        des = desc.cpu().numpy()
        des = np.sqrt(des / (des.sum(axis=1, keepdims=True) + 1e-7))  # RootSIFT
        kp_ = [KeyPoint(k["x"], k["y"], k["size"], k["angle"], k["response"], k["octave"])
               for k in kps]
        return kp_, des

    return _detect


def sift_pyramid(img, n_octaves=None):
    """Host convenience: uint8 BGR image -> (gauss, dog) as NumPy arrays."""
    eng = _eng.engine()
    frame = eng.upload_frames([img])[0]
    gauss, dog = sift_pyramid_device(frame, n_octaves)
    to_np = lambda pyr: [[p.cpu().numpy() for p in octave] for octave in pyr]   # noqa: E731  (stacks iterate as planes)
    return to_np(gauss), to_np(dog)


# ------------------------------------------------------------------ matching
class DMatch:
    """The fields of ``cv2.DMatch`` the reference reads (features.py:238)."""
    __slots__ = ("queryIdx", "trainIdx", "distance")

    def __init__(self, query, train, distance):
        self.queryIdx, self.trainIdx, self.distance = int(query), int(train), float(distance)


def knn2_device(des1, des2, eng=None, want_rescans=False):
    """Two nearest rows of ``des2`` (Euclidean) for every row of ``des1``; both device
    float32 [K][D], D <= 128.  Returns (indices int64 [K1][2], distances float32 [K1][2]),
    nearest first.  ``pano_knn2``: the cross terms of |a - b|^2 on the matrix cores (split
    float16) rank the rows, the four best per query are re-evaluated exactly in float32 and
    an error bound proves the rest cannot beat them (else that query is rescanned exactly)."""
    import torch
    eng = eng or _eng.engine()
    nq, d = (int(v) for v in des1.shape)
    nt = int(des2.shape[0])
    des1, des2 = des1.contiguous(), des2.contiguous()
    # a power of two that brings the largest magnitude to ~1024: float16 halves stay normal
    peak = float(torch.maximum(des1.abs().amax(), des2.abs().amax()).item()) if nq else 1.0
    scale = float(2.0 ** np.floor(np.log2(1024.0 / peak))) if peak > 0 else 1.0
    work = torch.empty(int(eng.lib.pano_knn2_work_bytes(nq, nt, d)), dtype=torch.uint8,
                       device=eng.device)
    idx = torch.empty((nq, 2), dtype=torch.int32, device=eng.device)
    dist = torch.empty((nq, 2), dtype=torch.float32, device=eng.device)
    rescans = torch.zeros(1, dtype=torch.int32, device=eng.device)
    _lib.check(eng.lib.pano_knn2(eng.ctx(), _eng._ptr(des1), nq, _eng._ptr(des2), nt, d,
                                 C.c_float(scale), _eng._ptr(work), _eng._ptr(idx),
                                 _eng._ptr(dist), _eng._ptr(rescans)), "pano_knn2")
    if want_rescans:
        return idx.long(), dist, int(rescans.item())
    return idx.long(), dist


def flann_matching(des1, des2, ratio=0.7):
    """Given 2 lists of descriptors, match them (features.py:222-232): 2-NN + Lowe's
    ratio test.  Exact search instead of FLANN's approximate one."""
    import torch
    eng = _eng.engine()
    if len(des1) == 0 or len(des2) < 2:
        return []
    d1 = torch.from_numpy(np.ascontiguousarray(des1, np.float32)).to(eng.device)
    d2 = torch.from_numpy(np.ascontiguousarray(des2, np.float32)).to(eng.device)
    idx, dist = knn2_device(d1, d2)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    keep = np.nonzero(dist[:, 0] < ratio * dist[:, 1])[0]
    return [DMatch(q, idx[q, 0], dist[q, 0]) for q in keep]
