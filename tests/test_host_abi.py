"""CPU: the product's host-side geometry against the golden vectors, and the
C-ABI library: it loads and exports every symbol include/pano360.h declares
(no compute calls - there is no GPU here)."""
import ctypes
import os
import pickle
import re

import numpy as np
import pytest

from conftest import ROOT, SCENES, load_golden, n_patches, scene_inputs


def test_library_exports_every_declared_symbol():
    from pano360_amd import _lib
    _lib.build()
    header = open(os.path.join(ROOT, "include", "pano360.h")).read()
    declared = set(re.findall(r"\b(pano_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    handle.pano_version.restype = ctypes.c_char_p
    assert b"gfx950" in handle.pano_version()
    assert handle.pano_pitch(1941) == 1944
    # record sizes against the C compiler's view of the header
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "sz.c")
        with open(src, "w") as fid:
            fid.write('#include <stdio.h>\n#include "pano360.h"\nint main(void){'
                      'printf("%zu %zu %zu %zu %zu %zu", sizeof(pano_patch), sizeof(pano_camera), '
                      'sizeof(pano_pair), sizeof(pano_stitch_args), sizeof(pano_layout), '
                      '__builtin_offsetof(pano_stitch_args, layout));'
                      'printf(" %zu %zu %zu %zu %zu", sizeof(pano_sift_args), '
                      '__builtin_offsetof(pano_sift_args, detect), '
                      '__builtin_offsetof(pano_sift_args, first_octave), '
                      '__builtin_offsetof(pano_sift_args, gauss_dev), '
                      '__builtin_offsetof(pano_sift_args, desc));return 0;}')
        exe = os.path.join(tmp, "sz")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes[:3] == [ctypes.sizeof(_lib.Patch), ctypes.sizeof(_lib.Camera),
                         ctypes.sizeof(_lib.Pair)] == [96, 120, 80]
    # the argument record of pano_stitch_multiband, field for field
    assert sizes[3:6] == [ctypes.sizeof(_lib.StitchArgs), ctypes.sizeof(_lib.Layout),
                          _lib.StitchArgs.layout.offset]
    # ... and of pano_sift_detect
    A = _lib.SiftArgs
    assert sizes[6:] == [ctypes.sizeof(A), A.detect.offset, A.first_octave.offset,
                         A.gauss_dev.offset, A.desc.offset]


def test_header_constants_match_binding():
    from pano360_amd import _lib
    header = open(os.path.join(ROOT, "include", "pano360.h")).read()
    for macro, value in (("PANO_MAX_TAPS", _lib.MAX_TAPS), ("PANO_MAX_LEVELS", _lib.MAX_LEVELS),
                         ("PANO_TAP_LEAD", _lib.TAP_LEAD), ("PANO_TAP_PAD", _lib.TAP_PAD)):
        assert int(re.search(rf"#define {macro} (\d+)", header).group(1)) == value


def test_no_product_import_of_the_oracle():
    """The product path must never route through oracle/ (the checker)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pano360_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not any(k in text for k in ("pano_oracle", "cv2_shim", "laplacian_oracle",
                                                   "sift_oracle")), f


@pytest.mark.parametrize("name", SCENES)
def test_plan_matches_reference(name):
    from pano360_amd import engine
    g = load_golden(name)
    imgs, rots, intrs, mr = scene_inputs(g)
    shapes = [im.shape[:2] for im in imgs]
    plan = engine.Plan(shapes, rots, intrs, True, mr)
    assert plan.shape == tuple(g["mb_shape"])
    assert np.array_equal(plan.resolution, g["resolution"])
    assert np.array_equal(plan.low, g["im_min"]) and np.array_equal(plan.high, g["im_max"])
    for i in range(n_patches(g)):
        assert plan.rects[i] == tuple(g[f"mb_irange_{i}"])
        assert np.array_equal(plan.ranges[i][0], g["range_min"][i])
        assert np.array_equal(plan.ranges[i][1], g["range_max"][i])
    plan = engine.Plan(shapes, rots, intrs, False, mr)
    assert plan.shape == tuple(g["lin_shape"])
    for i in range(n_patches(g, "lin")):
        assert plan.rects[i] == tuple(g[f"lin_irange_{i}"])


def test_stage_functions_match_reference():
    from pano360_amd import bundle_adj, engine
    g = load_golden("pure")
    assert np.array_equal(engine.SphProj.hom2proj(g["sph_pts"]), g["sph_h2p"])
    assert np.array_equal(engine.SphProj.proj2hom(g["sph_h2p"]), g["sph_p2h"])
    assert np.array_equal(engine.CylProj.hom2proj(g["sph_pts"]), g["cyl_h2p"])
    assert np.array_equal(engine.CylProj.proj2hom(g["cyl_h2p"]), g["cyl_p2h"])
    for size in (1, 2, 7, 64, 135):
        assert np.array_equal(engine.hat(size), g[f"hat_{size}"])
    cam = bundle_adj.Image(None, g["cam_rot"], g["cam_intr"])
    assert np.array_equal(cam.hom(), g["cam_hom"]) and np.array_equal(cam.proj(), g["cam_proj"])
    mn, mx = engine.range_from_border((72, 128), cam.hom())
    assert np.array_equal(mn, g["border_min"]) and np.array_equal(mx, g["border_max"])
    mn, mx = engine.range_from_corners((72, 128), cam.hom())
    assert np.array_equal(mn, g["corners_min"]) and np.array_equal(mx, g["corners_max"])
    for vec, mat in zip(g["rot_vecs"], g["rot_mats"]):
        assert np.array_equal(bundle_adj.rotation_to_mat(vec), mat)
    assert np.array_equal(bundle_adj.intrinsics((250.0, 999.0), (3.0, -4.0)), g["intr_pair"])
    assert np.array_equal(bundle_adj.intrinsics(300.0), g["cam_intr"])


def test_camera_inverse_and_projection_round_trip():
    """The reference's own tests that touch the path (pano_tests.py:29-33,59-77)."""
    from pano360_amd import bundle_adj, engine
    rng = np.random.default_rng(42)
    cam = bundle_adj.Image(None, bundle_adj.rotation_to_mat(rng.normal(size=3)),
                           bundle_adj.intrinsics(1e3))
    np.testing.assert_almost_equal(cam.hom().dot(cam.proj()), np.eye(3))
    pts = rng.normal(size=(10, 3))
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    for proj in (engine.SphProj, engine.CylProj):
        back = proj.proj2hom(proj.hom2proj(pts))
        back /= np.linalg.norm(back, axis=1, keepdims=True)
        np.testing.assert_almost_equal(back, pts)


def test_gaussian_taps_match_oracle(oracle):
    from pano360_amd import engine
    assert [engine.gaussian_ksize(s) for s in engine.level_sigmas(6)] == [33, 57, 73, 87, 97]
    for sigma in engine.level_sigmas(6) + [1.0, 2.0]:
        k = engine.gaussian_ksize(sigma)
        assert np.array_equal(engine.gaussian_taps(k, sigma), oracle.gaussian_kernel(k, sigma))
    padded = engine.padded_taps(engine.gaussian_taps(33, 4.0))
    assert len(padded) == 33 + 40 and not padded[:7].any() and not padded[40:].any()


def test_pickled_cameras_use_the_reference_module_path():
    import bundle_adj as top
    cam = top.Image(np.zeros((2, 2, 3), np.uint8), np.eye(3), top.intrinsics(10.0))
    blob = pickle.dumps([cam])
    assert b"bundle_adj" in blob and b"pano360_amd" not in blob.split(b"bundle_adj")[0]
    back = pickle.loads(blob)[0]
    assert isinstance(back, top.Image) and np.array_equal(back.intr, cam.intr)
    # the pickled state is the reference class's plain attribute dictionary (bundle_adj.py:18-25):
    # a cache written here loads into the reference's Image and the other way round
    assert cam.__getstate__().keys() == {"img", "rot", "intr", "range"}
    plain = top.Image.__new__(top.Image)
    plain.__setstate__({"img": cam.img, "rot": cam.rot, "intr": cam.intr, "range": cam.range})
    assert np.array_equal(plain.img, cam.img) and np.array_equal(plain.hom(), cam.hom())
    # pixels left on the device by stitch() arrive when first read, and pickle as arrays
    from pano360_amd.bundle_adj import Deferred
    calls = []
    lazy = top.Image(Deferred(lambda: calls.append(1) or np.ones((2, 2, 4), np.float32)),
                     np.eye(3), top.intrinsics(10.0))
    assert not calls
    assert lazy.img.shape == (2, 2, 4) and lazy.img is lazy.img and calls == [1]
    assert np.array_equal(pickle.loads(pickle.dumps(lazy)).img, np.ones((2, 2, 4), np.float32))


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pano360_amd import _lib, engine
    with pytest.raises(_lib.PanoError):
        engine.Engine()


def _reflect_101(p, n):
    if n == 1:
        return 0
    while not 0 <= p < n:
        p = -p if p < 0 else 2 * (n - 1) - p
    return p


def test_reflect_closed_is_tight_and_safe():
    """The window rule that lets the blur read only part of a patch: it must
    hold every REFLECT_101 image of the requested range, and not be wasteful
    when a single reflection suffices."""
    from pano360_amd.engine import reflect_closed
    for n in (1, 2, 3, 5, 17, 64):
        for lo in range(-3 * n - 2, n + 2):
            for hi in range(lo + 1, 3 * n + 3):
                a, b = reflect_closed(lo, hi, n)
                hit = {_reflect_101(p, n) for p in range(lo, hi)}
                assert 0 <= a < b <= n
                assert min(hit) >= a and max(hit) < b, (n, lo, hi, a, b)
                if n > 1 and lo >= -(n - 1) and hi <= 2 * n - 1 and hi > 0 and lo < n:
                    assert (a, b) == (min(hit), max(hit) + 1), (n, lo, hi, a, b)


def test_windows_for():
    from pano360_amd.engine import windows_for
    rect = (10, 110, 200, 500)                      # 100 x 300 patch
    assert windows_for((5, 4, 0, 0), rect, 43) is None          # owns nothing
    area, win = windows_for((10, 109, 300, 349), rect, 43)
    assert area == (0, 100, 57, 193)
    assert win == (0, 100, 12, 236)                   # 57 - 43 = 14 -> 12, 193 + 43 = 236: multiples of 4
    area, win = windows_for((10, 109, 200, 230), rect, 43)      # touches the left edge
    assert area == (0, 100, 0, 74) and win[2] == 0 and win[3] >= 74 + 43
    # tiny patch: everything reflects several times -> whole patch
    area, win = windows_for((0, 5, 0, 7), (0, 6, 0, 8), 43)
    assert area == (0, 6, 0, 8) and win == (0, 6, 0, 8)


def same64(a, b):
    a, b = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_find_gains_reference_vectors():
    """Host side of equalize_gains: find_gains equals the reference bit for bit on
    the construction of pano_tests.py:79-96 and recovers consistent gains."""
    from pano360_amd import stitcher
    g = load_golden("gains")
    found = stitcher.find_gains(g["fg_overlaps"], g["fg_sizes"])
    assert same64(found, g["fg_gains"])
    assert same64(stitcher.find_gains(g["fg_overlaps"], g["fg_sizes"], stdn=0.5, stdg=1.0),
                  g["fg_gains_wide"])
    ratio = found / g["fg_true"]
    np.testing.assert_almost_equal(ratio, np.full(len(ratio), ratio[0]))
    # the reference's own test, re-drawn
    rng = np.random.default_rng(7)
    gains = 1 + 0.1 * rng.standard_normal(10)
    overlaps = 100 + 10 * rng.standard_normal((10, 10))
    for i in range(10):
        for j in range(i + 1, 10):
            overlaps[i, j] = overlaps[j, i] * gains[j] / gains[i]
    ratio = stitcher.find_gains(overlaps, rng.standard_normal((10, 10)) + 10) / gains
    np.testing.assert_almost_equal(ratio, np.full(10, ratio[0]))


def test_gain_tables_and_pairs_match_reference(oracle):
    """The per-camera colour tables stand for the reference's equalised images
    exactly, and the pair table holds the reference's homographies."""
    from pano360_amd import engine
    g = load_golden("scene_equalize")
    imgs, rots, intrs, _ = scene_inputs(g)
    luts = engine.gain_tables(g["gains"])
    assert np.array_equal(luts[0][imgs[0]].view(np.uint32), g["eq_rgb_0"].view(np.uint32))
    assert np.array_equal(luts[-1][imgs[-1]].view(np.uint32), g["eq_rgb_last"].view(np.uint32))
    h, w = imgs[0].shape[:2]
    pairs = engine.overlap_pairs(rots, intrs, w, h)
    # every pair the reference found an overlap for is in the table
    listed = {(int(p["i"]), int(p["j"])) for p in pairs}
    assert {(i, j) for i, j in zip(*np.nonzero(g["sizes"])) if i < j} <= listed
    for p in pairs:
        hom, behind = oracle.pair_homography(rots[p["i"]], intrs[p["i"]], rots[p["j"]],
                                             intrs[p["j"]], w, h)
        assert not behind and same64(p["minv"].reshape(3, 3), oracle.invert3x3(hom))
    # a camera looking backwards drops out (stitcher.py:51-52)
    back = engine.overlap_pairs([np.eye(3), np.diag([-1.0, 1.0, -1.0])], [intrs[0], intrs[0]], w, h)
    assert len(back) == 0


def test_windows_for_many_equals_windows_for():
    """The vectorised window layout used between the two GPU stages is the scalar one."""
    from pano360_amd.engine import windows_for, windows_for_many
    rng = np.random.default_rng(11)
    boxes, rects = [], []
    for _ in range(400):
        h, w = int(rng.integers(1, 300)), int(rng.integers(1, 500))
        y0, x0 = int(rng.integers(0, 50)), int(rng.integers(0, 900))
        ya, xa = int(rng.integers(y0 - 3, y0 + h + 3)), int(rng.integers(x0 - 3, x0 + w + 3))
        boxes.append((ya, int(rng.integers(ya - 2, y0 + h + 5)), xa, int(rng.integers(xa - 2, x0 + w + 5))))
        rects.append((y0, y0 + h, x0, x0 + w))
    for radius, strip in ((43, None), (5, (300, 700)), (48, (0, 400)), (0, None)):
        keep, area, window = windows_for_many(boxes, rects, radius, strip)
        for i, (box, rect) in enumerate(zip(boxes, rects)):
            want = windows_for(box, rect, radius, strip)
            assert bool(keep[i]) == (want is not None)
            if want is not None:
                assert tuple(area[i]) == want[0] and tuple(window[i]) == want[1], (box, rect, radius)


def test_native_window_layout_equals_windows_for():
    """pano_layout_windows (the host's one call between the region search and the warp)
    against the scalar windows_for, on random rectangles, boxes, spans and strips; arena
    offsets must not overlap and tile offsets must follow the 32 x 32 anchored grid."""
    import ctypes as C
    from pano360_amd import _lib, engine
    lib = _lib.lib()
    rng = np.random.default_rng(11)
    max_spans, radius, n_blur = 4, 43, 4
    for trial in range(30):
        n = int(rng.integers(1, 12))
        rects = np.zeros((n, 4), np.int32)
        raw = np.zeros((n, 5 + 2 * max_spans), np.int32)
        for i in range(n):
            y0, x0 = int(rng.integers(0, 50)), int(rng.integers(0, 3000))
            h, w = int(rng.integers(1, 400)), int(rng.integers(1, 900))
            rects[i] = (y0, y0 + h, x0, x0 + w)
            if rng.random() < 0.2:
                raw[i, :4] = (1, 0, 1, 0)                     # owns nothing
                continue
            ys = np.sort(rng.integers(y0, y0 + h, 2))
            k = int(rng.integers(1, max_spans + 1))
            xs = np.sort(rng.choice(np.arange(x0, x0 + w), size=min(2 * k, w), replace=False))
            k = len(xs) // 2
            raw[i, :5] = (ys[0], ys[1], xs[0], xs[2 * k - 1], k)
            raw[i, 5:5 + 2 * k] = xs[:2 * k]
        strip = (int(rng.integers(0, 1500)), int(rng.integers(1500, 4000))) if trial % 2 else (0, 10 ** 6)
        have = np.ones(n, np.uint8)
        rec = np.zeros(n * max_spans, dtype=engine.PATCH_DTYPE)
        lay = _lib.Layout()
        _lib.check(lib.pano_layout_windows(32, raw.ctypes.data, n, max_spans, rects.ctypes.data,
                                           have.ctypes.data, radius, strip[0], strip[1], n_blur,
                                           rec.ctypes.data, len(rec), C.byref(lay)), "layout")
        want = []
        for i in range(n):
            for s in range(int(raw[i, 4])):
                box = (raw[i, 0], raw[i, 1], raw[i, 5 + 2 * s], raw[i, 6 + 2 * s])
                got = engine.windows_for(box, tuple(int(v) for v in rects[i]), radius, strip)
                if got is not None:
                    want.append((i,) + got)
        assert lay.n_records == len(want)
        tiles = planes = blurred = 0
        for r, (i, a, v) in zip(rec[:lay.n_records], want):
            assert r["index"] == i
            assert (r["ay0"], r["ay0"] + r["ah"], r["ax0"], r["ax0"] + r["aw"]) == a
            assert (r["vy0"], r["vy0"] + r["vh"], r["vx0"], r["vx0"] + r["vw"]) == v
            assert r["vpitch"] == (r["vw"] + 3) & ~3 and r["apitch"] == (r["aw"] + 31) & ~31
            assert r["tiles_off"] == tiles and r["planes"] == planes
            assert r["blurred"] == blurred + (r["ax0"] & 31)
            tiles += ((((a[3] - 1) >> 5) - (a[2] >> 5) + 1) * (((a[1] - 1) >> 5) - (a[0] >> 5) + 1))
            planes += 3 * int(r["vh"]) * int(r["vpitch"])
            blurred += n_blur * 4 * int(r["ah"]) * int(r["apitch"]) + 32
        assert (lay.n_tiles, lay.planes_floats, lay.blurred_floats) == (tiles, planes, blurred + 32)
        have[:] = 0
        _lib.check(lib.pano_layout_windows(32, raw.ctypes.data, n, max_spans, rects.ctypes.data,
                                           have.ctypes.data, radius, strip[0], strip[1], n_blur,
                                           rec.ctypes.data, len(rec), C.byref(lay)), "layout")
        assert lay.missing == lay.n_records


def test_top_level_stitcher_is_the_module_itself():
    """``import stitcher`` must hand out the module whose globals ``stitch`` reads at call
    time: setting ``stitcher.MAX_RESOLUTION`` as a reference caller does (stitcher.py:17,
    153-155) changes the resolution the very next call computes."""
    import bundle_adj
    import stitcher
    import pano360_amd.stitcher as impl
    from pano360_amd import synth
    assert stitcher is impl
    rots, intrs = synth.make_cameras(5, 240, 136, sweep_deg=90.0)
    regions = [bundle_adj.Image(np.zeros((136, 240, 3), np.uint8), r, k) for r, k in zip(rots, intrs)]
    for reg in regions:
        reg.range = stitcher._proj_img_range_border(reg.img.shape[:2], reg.hom())
    saved = stitcher.MAX_RESOLUTION
    try:
        sizes = {}
        for cap in (1400, 300, 10 ** 9):
            stitcher.MAX_RESOLUTION = cap
            res, (lo, hi) = stitcher.estimate_resolution(regions)
            sizes[cap] = np.round((hi - lo) / res).astype(int)
    finally:
        stitcher.MAX_RESOLUTION = saved
    # native size 600 x 156: the default cap of 1400 does not bind, a cap of 300 does
    assert list(sizes[10 ** 9]) == list(sizes[1400]) == [600, 156]
    assert list(sizes[300]) == [300, 78]


def test_plan_tables_on_a_column_range_equal_the_full_tables():
    """Plan(table_cols=...) (one GPU's strip): the sin / cos entries it evaluates are the full
    tables' values bit for bit - NumPy's loops do not depend on where a slice starts - and
    everything else is NaN; ranges, rectangles and tan are untouched."""
    from pano360_amd import engine, synth
    imgs_shape = (270, 480)
    rots, intrs = synth.make_cameras(12, 480, 270, sweep_deg=150.0)
    shapes = [imgs_shape] * 12
    full = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    W = full.shape[1]
    for a, b in [(0, W), (1, 7), (3, W - 5), (W // 3 + 1, W // 2), (-50, 40), (W - 33, W + 100)]:
        part = engine.Plan(shapes, rots, intrs, True, 10 ** 9, table_cols=(a, b))
        lo, hi = max(a, 0), min(b, len(full.sin_t))
        assert np.array_equal(part.sin_t[lo:hi], full.sin_t[lo:hi])
        assert np.array_equal(part.cos_t[lo:hi], full.cos_t[lo:hi])
        assert np.isnan(part.sin_t[:lo]).all() and np.isnan(part.cos_t[hi:]).all()
        assert np.array_equal(part.tan_p, full.tan_p)
        assert part.rects == full.rects and part.shape == full.shape


def test_stacked_border_product_equals_the_per_frame_dot():
    """``range_arrays_from_border`` multiplies all frames' homographies with the border ring in
    one ``np.matmul``; the reference calls ``hom.dot(ring)`` per frame (stitcher.py:119).  Same
    bits - checked here so that a platform whose BLAS disagrees fails loudly."""
    from pano360_amd import engine, synth
    for name in ("cfg2", "cfg3", "cfg5"):
        cfg = synth.CONFIGS[name]
        rots, intrs = synth.make_cameras(cfg["n"], cfg["width"], cfg["height"],
                                         sweep_deg=cfg.get("sweep_deg"),
                                         step_deg=cfg.get("step_deg"))
        shapes = [(cfg["height"], cfg["width"])] * cfg["n"]
        homs = [np.asarray(r).T.dot(np.linalg.inv(k)) for r, k in zip(rots, intrs)]
        ring = engine.border_ring(shapes[0])
        assert np.array_equal(np.matmul(np.asarray(homs), ring),
                              np.stack([h.dot(ring) for h in homs])), name
        low, high = engine.range_arrays_from_border(shapes, homs)
        for i, hom in enumerate(homs[:5]):
            lo, hi = engine.range_from_border(shapes[i], hom)
            assert np.array_equal(lo, low[i]) and np.array_equal(hi, high[i])


def test_plan_memo_is_keyed_on_every_bit_of_the_cameras():
    """engine.PlanMemo (Engine.cached_plan, the default of the strips path): the same shapes,
    rotations, calibrations, padding, cap and table columns hit - and hand back a plan equal to
    a fresh ``Plan(...)``, which is what the reference recomputes per stitch
    (stitcher.py:276-302) - and ONE entry of ONE matrix moved by one unit in the last place
    misses, as do a changed cap, padding flag, shape and table range."""
    import numpy as np
    from pano360_amd import engine, synth
    rots, intrs = synth.make_cameras(6, 160, 90, sweep_deg=100.0, jitter=0.01, seed=3)
    shapes = [(90, 160)] * 6
    memo = engine.PlanMemo()
    made = []

    def make(*a):
        made.append(a)
        return engine.Plan(*a)
    first = memo.get(shapes, rots, intrs, True, 1400, None, make=make)
    again = memo.get([tuple(s) for s in shapes], rots.copy(), intrs.copy(), True, 1400.0, None, make=make)
    assert again is first and len(made) == 1 and (memo.hits, memo.misses) == (1, 1)
    fresh = engine.Plan(shapes, rots, intrs, True, 1400)
    assert fresh.shape == first.shape and fresh.rects == first.rects
    for a, b in ((fresh.sin_t, first.sin_t), (fresh.cos_t, first.cos_t), (fresh.tan_p, first.tan_p),
                 (np.asarray(fresh.projs), np.asarray(first.projs)), (fresh.resolution, first.resolution)):
        assert np.array_equal(a, b)
    # one ulp in one entry of one rotation / one calibration
    for which in ("rot", "intr"):
        r2, k2 = rots.copy(), intrs.copy()
        target = r2 if which == "rot" else k2
        target[4, 1, 1] = np.nextafter(target[4, 1, 1], np.inf)
        assert memo.key(shapes, r2, k2, True, 1400) != memo.key(shapes, rots, intrs, True, 1400)
        moved = memo.get(shapes, r2, k2, True, 1400, None, make=make)
        assert moved is not first
    assert len(made) == 3
    # everything else a plan depends on
    base = memo.key(shapes, rots, intrs, True, 1400)
    assert memo.key(shapes, rots, intrs, False, 1400) != base
    assert memo.key(shapes, rots, intrs, True, 1401) != base
    assert memo.key([(90, 161)] + shapes[1:], rots, intrs, True, 1400) != base
    assert memo.key(shapes, rots, intrs, True, 1400, (0, 64)) != base
    assert memo.key(shapes, rots, intrs, True, 1400, (0, 64)) != memo.key(shapes, rots, intrs, True, 1400, (0, 65))
    # the oldest entry leaves when the memo is full
    small = engine.PlanMemo(capacity=2)
    for cap in (1400, 1401, 1402):
        small.get(shapes, rots, intrs, True, cap)
    assert len(small) == 2 and small.key(shapes, rots, intrs, True, 1400) not in small.plans


def test_bench_reads_the_committed_profiles():
    """bench.py's `scaling_projection` (the committed strip-floor emulation: worlds 1 and 8, factors
    from its own milliseconds) and `pmc_traffic` (the newest round's counter summary of the
    workload, a closing visit's before the earlier ones)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    proj = bench.scaling_projection("cfg3")
    assert proj is not None and proj["emulated_on_one_gpu"] and "excluded" in proj["exchange"]
    ms = proj["ms_per_stitch_and_rank"]
    assert {"1", "8"} <= set(ms) and ms["8"] < ms["1"]
    assert abs(proj["factor_vs_world_1"]["8"] - ms["1"] / ms["8"]) < 1e-9
    assert os.path.exists(os.path.join(ROOT, proj["source"]))
    assert bench.scaling_projection("no_such_workload") is None
    traffic, source = bench.pmc_traffic("blur_lean_kernel", "cfg3")
    assert traffic and traffic > 1e9 and source.startswith("profiles/")
    import glob
    import json
    rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r"))

    def has_cfg3(r):                                # a counter summary OF THIS WORKLOAD in the round
        for path in glob.glob(os.path.join(ROOT, "profiles", r, "**", "pmc_traffic*.json"), recursive=True):
            with open(path) as fid:
                if json.load(fid).get("workload") == "cfg3":
                    return True
        return False
    newest = [r for r in rounds if has_cfg3(r)][-1]
    assert source.split("/")[1] == newest
    assert bench.pmc_traffic("blur_lean_kernel", "no_such_workload") == (None, None)


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_stdout_line_is_compact_and_complete():
    """The ONE stdout line of bench.py (`compact_line`) stays under 4 KB whatever the whole record
    holds - round 5's 28 KB line lost its front in the driver's log - parses, and carries the
    headline, `settings`, `roofline` and `cpu_baseline`; everything else goes to the side file."""
    import json
    bench = _bench_module()
    with open(os.path.join(ROOT, "profiles", "r05", "final", "bench_default.json")) as fid:
        full = json.load(fid)                       # a whole record as rounds 1 - 5 printed it
    assert len(json.dumps(full)) > 20000
    full["settings"] = {"plan": "per_stitch", "lanes": 2, "trusted": False, "keep_geometry": False}
    full["alt_settings"] = {"plan": "memo", "lanes": 3, "trusted": True, "ms_per_step": 1.4}
    full["roofline_by_kernel"] = {f"kernel_{k}": {"bound": "hbm", "frac": 0.4, "ms": 0.1, "x": "y" * 99}
                                  for k in range(14)}
    line = bench.compact_line(full, os.path.join(ROOT, "gpurun_out", "bench_full_cfg3_n1.json"))
    assert len(line) < bench.COMPACT_LIMIT and "\n" not in line
    got = json.loads(line)
    for key in ("metric", "value", "unit", "value_kind", "processed_MPps", "n_gpus", "steps", "warmup",
                "ms_per_step", "dtype", "arithmetic", "higher_is_better", "scaling", "vs_baseline",
                "data", "config", "settings", "roofline", "cpu_baseline", "side_file"):
        assert key in got, key
    assert got["value"] == pytest.approx(full["value"], rel=1e-5)
    assert got["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "blend_frac",
            "weighted_frac"} <= set(got["roofline"])
    assert got["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    assert {"value", "unit", "cores", "kind", "sample"} <= set(got["cpu_baseline"])
    assert len(got["cpu_baseline"]["sample"]) <= 200
    assert "workload" in got["config"] and "model" not in got["config"]
    assert got["settings"]["plan"] == "per_stitch" and got["alt_settings"]["plan"] == "memo"
    assert got["secondary_ms"]["cfg5"] == pytest.approx(full["secondary"]["cfg5"]["ms_per_step"], rel=1e-3)
    # a record bloated far past anything real still comes out under the limit, headline intact
    full["secondary"] = {f"entry_{k}": {"ms_per_step": 1.0} for k in range(400)}
    full["config"]["parallelism"] = "x" * 5000
    line = bench.compact_line(full, None)
    got = json.loads(line)
    assert len(line) < bench.COMPACT_LIMIT
    assert {"metric", "value", "ms_per_step", "roofline", "cpu_baseline", "settings"} <= set(got)
    # the error notes of a failed multi-rank run ride on the line
    full["strips_error"] = "RuntimeError('x')"
    assert json.loads(bench.compact_line(full, None))["strips_error"] == "RuntimeError('x')"
