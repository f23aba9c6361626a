"""CPU: the C oracle against the golden vectors produced by running the
reference (oracle/gen_golden.py).  Everything here is bit-exact."""
import numpy as np
import pytest

from conftest import SCENES, load_golden, n_patches, scene_inputs


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same64(a, b):
    """Bit equality of float64 arrays."""
    a, b = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("name", SCENES)
def test_plan_and_warp(oracle, name):
    g = load_golden(name)
    imgs, rots, intrs, mr = scene_inputs(g)
    plan, patches, maps = oracle.warp_all(imgs, rots, intrs, True, mr)
    assert plan.shape == tuple(g["mb_shape"])
    assert np.array_equal(plan.resolution, g["resolution"])
    assert np.array_equal(plan.lo, g["im_min"]) and np.array_equal(plan.hi, g["im_max"])
    for i, rng in enumerate(plan.ranges):
        assert np.array_equal(rng[0], g["range_min"][i])
        assert np.array_equal(rng[1], g["range_max"][i])
    assert n_patches(g) == len(patches)
    for i, ((warped, mask, _), (mx, my)) in enumerate(zip(patches, maps)):
        assert plan.rects[i] == tuple(g[f"mb_irange_{i}"])
        assert np.array_equal(bits(mx), bits(g[f"mb_mapx_{i}"]))
        assert np.array_equal(bits(my), bits(g[f"mb_mapy_{i}"]))
        assert np.array_equal(mask, g[f"mb_mask_{i}"])
        if f"mb_warped_{i}" in g:
            assert np.array_equal(bits(warped), bits(g[f"mb_warped_{i}"]))
    assert np.array_equal(bits(oracle.add_weights(imgs[0])[..., 3]), bits(g["alpha0"]))


@pytest.mark.parametrize("name", SCENES)
def test_inverse_map_fma_choice_is_immaterial(oracle, name):
    """The 3x3 product in double as FMA chain or as separate mul/add gives the
    same float32 maps on these scenes (inverse map)."""
    g = load_golden(name)
    imgs, rots, intrs, mr = scene_inputs(g)
    plan = oracle.Plan([im.shape[:2] for im in imgs], rots, intrs, True, mr)
    for i in range(len(imgs)):
        a = oracle.inverse_map(plan.projs[i], plan, plan.rects[i], imgs[i].shape[:2], True)
        b = oracle.inverse_map(plan.projs[i], plan, plan.rects[i], imgs[i].shape[:2], False)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("name", SCENES)
def test_blenders_and_crop(oracle, name):
    g = load_golden(name)
    imgs, rots, intrs, mr = scene_inputs(g)
    for lv in (5, 6):
        if f"mb{lv}_mosaic" in g:
            out = oracle.stitch(imgs, rots, intrs, "multiband", lv, max_resolution=mr)
            assert np.array_equal(out, g[f"mb{lv}_mosaic"])
    assert np.array_equal(oracle.stitch(imgs, rots, intrs, "linear", max_resolution=mr),
                          g["linear_mosaic"])
    assert np.array_equal(oracle.stitch(imgs, rots, intrs, "none", max_resolution=mr),
                          g["none_mosaic"])
    assert np.array_equal(
        oracle.stitch(imgs, rots, intrs, "linear", crop=True, max_resolution=mr),
        g["lin_cropped"])
    plan, patches, _ = oracle.warp_all(imgs, rots, intrs, True, mr)
    valid = oracle.valid(patches, plan.shape)
    assert np.array_equal(valid, g["mb_valid"])
    assert oracle.crop_rect(valid) == tuple(g["mb_crop_rect"])
    plan, patches, _ = oracle.warp_all(imgs, rots, intrs, False, mr)
    valid = oracle.valid(patches, plan.shape)
    assert np.array_equal(valid, g["lin_valid"])
    assert oracle.crop_rect(valid) == tuple(g["lin_crop_rect"])


def test_pure_stage_functions(oracle):
    g = load_golden("pure")
    assert np.array_equal(oracle.sph_hom2proj(g["sph_pts"]), g["sph_h2p"])
    assert np.array_equal(oracle.sph_proj2hom(g["sph_h2p"]), g["sph_p2h"])
    assert np.array_equal(oracle.cyl_hom2proj(g["sph_pts"]), g["cyl_h2p"])
    assert np.array_equal(oracle.cyl_proj2hom(g["cyl_h2p"]), g["cyl_p2h"])
    for size in (1, 2, 7, 64, 135):
        assert np.array_equal(oracle.hat(size), g[f"hat_{size}"])
    mn, mx = oracle.range_border((72, 128), g["cam_hom"])
    assert np.array_equal(mn, g["border_min"]) and np.array_equal(mx, g["border_max"])
    mn, mx = oracle.range_corners((72, 128), g["cam_hom"])
    assert np.array_equal(mn, g["corners_min"]) and np.array_equal(mx, g["corners_max"])


def test_spherical_round_trip(oracle):
    """The reference's own test of the path (pano_tests.py:59-67)."""
    pts = np.random.default_rng(42).normal(size=(10, 3))
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    back = oracle.sph_proj2hom(oracle.sph_hom2proj(pts))
    back /= np.linalg.norm(back, axis=1, keepdims=True)
    np.testing.assert_almost_equal(back, pts)
    back = oracle.cyl_proj2hom(oracle.cyl_hom2proj(pts))
    back /= np.linalg.norm(back, axis=1, keepdims=True)
    np.testing.assert_almost_equal(back, pts)


def test_crop_masks(oracle):
    g = load_golden("pure")
    for i in range(int(g["n_crop"])):
        assert oracle.crop_rect(g[f"crop_mask_{i}"]) == tuple(g[f"crop_rect_{i}"]), i
    with pytest.raises(UnboundLocalError):
        oracle.crop_rect(np.zeros((4, 5), bool))


def _bl_patches(g):
    out = []
    for i in range(int(g["bl_n"])):
        y0, y1, x0, x1 = g[f"bl_irange_{i}"]
        out.append((g[f"bl_warped_{i}"].copy(), g[f"bl_mask_{i}"].copy(),
                    np.s_[int(y0):int(y1), int(x0):int(x1)]))
    return out


def test_stage_blenders(oracle):
    g = load_golden("pure")
    shape = tuple(int(v) for v in g["bl_shape"])
    assert np.array_equal(oracle.no_blend(_bl_patches(g), shape), g["bl_none"])
    assert np.array_equal(oracle.linear_blend(_bl_patches(g), shape), g["bl_linear"])
    assert np.array_equal(oracle.multiband_blend(_bl_patches(g), shape), g["bl_mb5"])
    assert np.array_equal(oracle.multiband_blend(_bl_patches(g), shape, 3), g["bl_mb3"])
    assert np.array_equal(oracle.valid(_bl_patches(g), shape), g["bl_valid"])


def test_filters(oracle):
    g = load_golden("pure")
    assert np.array_equal(bits(oracle.gaussian_filter(g["gf_img"])), bits(g["gf_s1"]))
    assert np.array_equal(bits(oracle.gaussian_filter(g["gf_img"], 2.0)), bits(g["gf_s2"]))
    p1 = oracle.pyr_down(g["gf_img"])
    assert np.array_equal(bits(p1), bits(g["pyr_1"]))
    assert np.array_equal(bits(oracle.pyr_down(p1)), bits(g["pyr_2"]))
    assert [oracle.gaussian_ksize(4 * np.sqrt(2 * k + 1.0)) for k in range(5)] == \
        [33, 57, 73, 87, 97]


def test_border_rule(oracle):
    lib = oracle.lib()
    for length in (1, 2, 3, 7):
        for p in range(-3 * length - 2, 3 * length + 3):
            for mode in (2, 4):
                want = p          # OpenCV's iterative rule, stated independently
                if length == 1:
                    want = 0
                else:
                    while not 0 <= want < length:
                        d = 1 if mode == 4 else 0
                        want = -want - 1 + d if want < 0 else length - 1 - (want - length) - d
                assert lib.orc_border(p, length, mode) == want


def test_find_gains_golden(oracle):
    """find_gains against the reference on the construction of its own test
    (pano_tests.py:79-96), and the property that test checks."""
    g = load_golden("gains")
    found = oracle.find_gains(g["fg_overlaps"], g["fg_sizes"])
    assert same64(found, g["fg_gains"])
    assert same64(oracle.find_gains(g["fg_overlaps"], g["fg_sizes"], stdn=0.5, stdg=1.0),
                  g["fg_gains_wide"])
    ratio = found / g["fg_true"]
    np.testing.assert_almost_equal(ratio, np.full(len(ratio), ratio[0]))


def test_equalize_gains_golden(oracle):
    """stitch(equalize=True): overlap sizes / means, gains, equalised frames and
    both mosaics equal the reference's, bit for bit."""
    g = load_golden("scene_equalize")
    imgs, rots, intrs, _ = scene_inputs(g)
    rgbas = [oracle.add_weights(im) for im in imgs]
    overlaps, sizes, gains = oracle.equalize_gains(rgbas, rots, intrs)
    assert np.array_equal(sizes, g["sizes"])
    assert same64(overlaps, g["overlaps"])
    assert same64(gains, g["gains"])
    assert np.array_equal(bits(rgbas[0][..., :3]), bits(g["eq_rgb_0"]))
    assert np.array_equal(bits(rgbas[-1][..., :3]), bits(g["eq_rgb_last"]))
    for blend, key in (("linear", "lin"), ("multiband", "mb5")):
        mosaic = oracle.stitch(imgs, rots, intrs, blend=blend, equalize=True)
        assert np.array_equal(mosaic, g[f"{key}_mosaic"]), blend


def test_warp_perspective_shim_matches_oracle(oracle):
    """The two independent restatements of cv2.warpPerspective(BORDER_TRANSPARENT)
    agree on a pair with a strong perspective component."""
    import cv2_shim
    rng = np.random.default_rng(5)
    h, w = 40, 72
    a = oracle.add_weights(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    b = oracle.add_weights(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    hom = np.array([[0.9, 0.05, 6.0], [-0.04, 1.1, -3.0], [4e-4, -3e-4, 1.0]])
    warped = cv2_shim.warpPerspective(b, hom, (w, h), borderMode=cv2_shim.BORDER_TRANSPARENT)
    mask = warped[..., 3] != 0
    size, mean_i, mean_j = oracle.overlap_stats(a, b, hom)
    assert size == mask.sum() and 0 < size < h * w
    assert np.float32(mean_i) == np.mean(a[mask, :3])
    assert np.float32(mean_j) == np.mean(warped[mask, :3])


def test_sift_oracle_building_blocks():
    """Pieces of the SIFT restatement that can be checked without OpenCV: fastAtan2
    stays within its documented 0.3 degrees of atan2, the float LU solve agrees
    with LAPACK, the keypoint sort is the one removeDuplicatedSorted specifies."""
    import sift_oracle as so
    rng = np.random.default_rng(2)
    y, x = rng.normal(size=4000).astype(np.float32), rng.normal(size=4000).astype(np.float32)
    want = np.degrees(np.arctan2(y.astype(np.float64), x.astype(np.float64))) % 360
    err = np.abs(so.fast_atan2(y, x) - want)
    assert np.minimum(err, 360 - err).max() < 0.3
    assert so.fast_atan2(np.float32(0), np.float32(0)) == 0
    for _ in range(50):
        a = rng.normal(size=(3, 3)).astype(np.float32)
        b = rng.normal(size=3).astype(np.float32)
        np.testing.assert_allclose(so._solve3(a, b), np.linalg.solve(a.astype(np.float64), b),
                                   rtol=2e-3, atol=2e-4)
    assert so._solve3(np.zeros((3, 3)), np.ones(3)) is None
    kp = [dict(x=2.0, y=1.0, size=3.0, angle=10.0, response=0.1, octave=5),
          dict(x=1.0, y=5.0, size=3.0, angle=10.0, response=0.1, octave=5),
          dict(x=2.0, y=1.0, size=4.0, angle=10.0, response=0.1, octave=5),
          dict(x=2.0, y=1.0, size=3.0, angle=10.0, response=0.3, octave=7)]
    out = so.sort_unique(kp)
    assert [(k["x"], k["size"], k["response"]) for k in out] == [(1.0, 3.0, 0.1), (2.0, 4.0, 0.1),
                                                                 (2.0, 3.0, 0.3)]
    assert so.unpack_octave(255 | (2 << 8)) == (-1, 2, 2.0)
    des = so.root_sift(np.array([[4.0, 0.0, 12.0]]))
    np.testing.assert_allclose((des ** 2).sum(), 1.0, atol=1e-6)


# ------------------------------------------------------------------ laplacian / resize
@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_laplacian_oracle_reproduces_the_reference(case):
    """blend.laplacian_blending run from the reference (gen_golden.py) against the
    restatement the GPU tests use as their checker."""
    import laplacian_oracle as lo
    g = load_golden("laplacian")
    mask = g[f"{case}_mask"] if f"{case}_mask" in g else None
    out = lo.laplacian_blending(g[f"{case}_img1"], g[f"{case}_img2"], mask,
                                int(g[f"{case}_levels"]))
    assert out.dtype == np.uint8 and np.array_equal(out, g[f"{case}_blended"])


def test_pyr_up_properties():
    """pyrUp of a constant is that constant (the filter has unit gain, borders
    included), the output is twice the input, float64 stays float64."""
    import cv2_shim as cv
    for dtype in (np.float32, np.float64):
        const = np.full((5, 7, 3), 3.25, dtype)
        up = cv.pyrUp(const)
        assert up.shape == (10, 14, 3) and up.dtype == dtype and np.ptp(up) == 0 and up[0, 0, 0] == 3.25
        assert np.ptp(cv.pyrDown(const)) == 0
    rng = np.random.default_rng(0)
    a = rng.random((6, 9)).astype(np.float32)
    # linear in the image: exact for power-of-two factors
    assert np.array_equal(cv.pyrUp(a * np.float32(4)), cv.pyrUp(a) * np.float32(4))


def test_resize_u8_oracle_properties():
    import cv2_shim as cv
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    half = cv.resize(img, None, fx=0.5, fy=0.5)
    box = img.astype(int).reshape(24, 2, 32, 2, 3).sum(axis=(1, 3))
    assert np.array_equal(half, (box + 2) >> 2)
    # 4:1: source coordinate 4 d + 1.5 -> the mean of samples 4 d + 1 and 4 d + 2 per axis
    quarter = cv.resize(img, None, fx=0.25, fy=0.25)
    assert quarter.shape == (12, 16, 3)
    mid = img.astype(float)
    mid = (mid[:, 1::4] + mid[:, 2::4]) / 2
    mid = (mid[1::4] + mid[2::4]) / 2
    assert np.abs(quarter.astype(float) - mid).max() <= 1.0
    flat = np.full((30, 50, 3), 201, np.uint8)
    for shrink in (2, 3, 4, 2.5):
        assert np.all(cv.resize(flat, None, fx=1 / shrink, fy=1 / shrink) == 201)


def test_multiband_blur_levels_golden(oracle):
    """What the reference's multiband_blend hands to cv2.GaussianBlur and gets back
    (stitcher.py:207-208, 218, 226; recorded by a spy while the reference ran): the input is
    the warped patch with the sharp ownership mask as alpha, the sigmas are 4 sqrt(2k + 1),
    and the oracle's blur reproduces every level bit for bit."""
    g = load_golden("scene_small_noise")
    imgs, rots, intrs, mr = scene_inputs(g)
    idx = int(g["blur_patch"])
    _, patches, _ = oracle.warp_all(imgs, rots, intrs, True, mr)
    own = oracle.ownership(patches, tuple(int(v) for v in g["mb_shape"]))
    rgba = patches[idx][0].copy()
    rgba[..., 3] = own[patches[idx][2]] == idx
    assert np.array_equal(bits(rgba), bits(g["blur_in"]))
    assert np.array_equal(g["blur_sigma"], [np.sqrt(2 * k + 1.0) * 4 for k in range(4)])
    for k, sigma in enumerate(g["blur_sigma"]):
        got = oracle.gaussian_blur(g["blur_in"], oracle.gaussian_ksize(sigma), sigma)
        assert np.array_equal(bits(got), bits(g[f"blur_out_{k}"])), k


# ------------------------------------------------ third-party cross-checks of the cv2 restatement
# oracle/cv2_shim.py restates remap / GaussianBlur / pyrDown from OpenCV's documented algorithm
# (cv2 is not installable here: SURVEY.md 8c); the C oracle and the HIP kernels are compared with
# it.  These tests pin the restatement itself against independent implementations that were
# not written for this repo: scipy.ndimage (C code, float64 arithmetic) and torch's grid_sample.
def _smooth_image(rng, h, w, c=None):
    from scipy import ndimage
    shape = (h, w) if c is None else (h, w, c)
    img = rng.random(shape)
    img = ndimage.gaussian_filter(img, sigma=(1.5, 1.5) + ((0,) if c else ()), mode="nearest")
    return ((img - img.min()) / (img.max() - img.min())).astype(np.float32)


def test_shim_remap_against_scipy_map_coordinates():
    """cv2.remap(INTER_LINEAR, BORDER_REFLECT) (stitcher.py:315-316) = bilinear interpolation
    with half-sample symmetric extension = scipy's map_coordinates(order=1, mode='reflect'),
    once the coordinates lie on remap's 1/32-pixel grid (its fixed-point step is then exact)."""
    import cv2_shim
    from scipy import ndimage
    rng = np.random.default_rng(11)
    for (h, w) in ((37, 53), (8, 120), (64, 9)):
        src = _smooth_image(rng, h, w, 4)
        # coordinates on the 1/32 grid, up to three pixels beyond every edge (one reflection)
        n = 4000
        xs = rng.integers(-3 * 32, (w + 2) * 32, n).astype(np.float32) / np.float32(32)
        ys = rng.integers(-3 * 32, (h + 2) * 32, n).astype(np.float32) / np.float32(32)
        got = cv2_shim.remap(src, xs[None, :], ys[None, :], cv2_shim.INTER_LINEAR,
                             borderMode=cv2_shim.BORDER_REFLECT)[0]
        for ch in range(4):
            want = ndimage.map_coordinates(src[..., ch].astype(np.float64),
                                           [ys.astype(np.float64), xs.astype(np.float64)],
                                           order=1, mode="reflect")
            assert np.abs(got[:, ch] - want).max() <= 1e-6, (h, w, ch)


def test_shim_remap_against_torch_grid_sample():
    """The same inside the image, against torch.nn.functional.grid_sample (bilinear,
    align_corners=True: pixel centres at integer coordinates)."""
    import cv2_shim
    import torch
    rng = np.random.default_rng(12)
    h, w = 41, 67
    src = _smooth_image(rng, h, w, 3)
    n = 5000
    xs = rng.integers(0, (w - 1) * 32 + 1, n).astype(np.float32) / np.float32(32)
    ys = rng.integers(0, (h - 1) * 32 + 1, n).astype(np.float32) / np.float32(32)
    got = cv2_shim.remap(src, xs[None, :], ys[None, :], cv2_shim.INTER_LINEAR,
                         borderMode=cv2_shim.BORDER_REFLECT)[0]
    grid = torch.stack([torch.from_numpy(xs.astype(np.float64)) / (w - 1) * 2 - 1,
                        torch.from_numpy(ys.astype(np.float64)) / (h - 1) * 2 - 1], dim=-1)
    want = torch.nn.functional.grid_sample(
        torch.from_numpy(src.astype(np.float64)).permute(2, 0, 1)[None], grid[None, None],
        mode="bilinear", padding_mode="border", align_corners=True)[0, :, 0].T.numpy()
    assert np.abs(got - want).max() <= 1e-6


@pytest.mark.parametrize("sigma", [4.0, 4 * np.sqrt(3.0), 4 * np.sqrt(5.0), 4 * np.sqrt(7.0), 12.0])
def test_shim_gaussian_blur_against_scipy_correlate1d(sigma):
    """cv2.GaussianBlur(img, (0, 0), sigma) (stitcher.py:218, 226): aperture cvRound(8 sigma + 1) | 1,
    taps exp(-x^2 / 2 sigma^2) normalised, separable, BORDER_REFLECT_101 = scipy's 'mirror'.
    The taps are recomputed here in float64 from the formula, not taken from the shim."""
    import cv2_shim
    from scipy import ndimage
    rng = np.random.default_rng(13)
    img = _smooth_image(rng, 150, 131, 4)
    ksize = int(round(sigma * 8 + 1)) | 1
    assert cv2_shim.gaussian_ksize(sigma) == ksize
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2
    taps = np.exp(-x * x / (2 * sigma * sigma))
    taps /= taps.sum()
    assert np.abs(cv2_shim.getGaussianKernel(ksize, sigma).ravel() - taps).max() <= 1e-8
    got = cv2_shim.GaussianBlur(img, (0, 0), sigma)
    want = ndimage.correlate1d(img.astype(np.float64), taps, axis=1, mode="mirror")
    want = ndimage.correlate1d(want, taps, axis=0, mode="mirror")
    assert np.abs(got - want).max() <= 1e-6
    # ... and the same small image whose every row and column reflects (aperture > size)
    small = _smooth_image(rng, 23, 17)
    got = cv2_shim.GaussianBlur(small, (0, 0), sigma)
    want = ndimage.correlate1d(ndimage.correlate1d(small.astype(np.float64), taps, axis=1, mode="mirror"),
                               taps, axis=0, mode="mirror")
    if ksize // 2 < min(small.shape):          # (scipy's 'mirror' reflects once, like REFLECT_101)
        assert np.abs(got - want).max() <= 1e-6


@pytest.mark.parametrize("shape", [(64, 48), (37, 53), (5, 120), (1, 9)])
def test_shim_pyr_down_against_scipy_correlate1d(shape):
    """cv2.pyrDown (features.py:155, blend.py:119): [1 4 6 4 1] / 16 along both axes with
    REFLECT_101, even rows and columns kept."""
    import cv2_shim
    from scipy import ndimage
    rng = np.random.default_rng(14)
    img = _smooth_image(rng, max(shape[0], 4), max(shape[1], 4))[:shape[0], :shape[1]]
    taps = np.array([1, 4, 6, 4, 1], np.float64) / 16
    want = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), taps, axis=1, mode="mirror"),
                               taps, axis=0, mode="mirror")[::2, ::2]
    got = cv2_shim.pyrDown(img)
    assert got.shape == want.shape
    if min(shape) > 2:                          # (one reflection: what both definitions share)
        assert np.abs(got - want).max() <= 1e-6


# ---- the SIFT scale space's restatement (oracle/sift_pyramid.py) against third parties -------------
# OpenCV's SIFT is not pinned by the reference and not installed: the keypoint and descriptor
# stages stay unpinned.  The SCALE SPACE, though, is made of operations that libraries not written
# for this repo implement too - a grey conversion (Pillow), a 2 x bilinear resize (SciPy, torch), a
# nearest-neighbour decimation (torch), separable Gaussian filters (SciPy) - and of a sigma schedule
# whose defining property can be checked directly (layer i of an octave has the total sigma
# 1.6 * 2^(i/3) of the octave's base resolution).
def test_sift_grey_conversion_against_pillow():
    """cv2.cvtColor(BGR2GRAY) on uint8: 14-bit fixed-point Rec. 601 weights on B, G, R in that
    order.  Pillow's 'L' conversion uses the same weights in 16-bit fixed point: within one level,
    and the channel order is BGR (swapping R and B moves it by tens of levels)."""
    import sift_pyramid as sp
    from PIL import Image
    rng = np.random.default_rng(21)
    bgr = rng.integers(0, 256, (64, 80, 3), dtype=np.uint8)
    got = sp.gray_u8(bgr)
    want = np.asarray(Image.fromarray(bgr[..., ::-1].copy(), "RGB").convert("L"), np.float32)
    assert np.abs(got - want).max() <= 1.0
    assert np.abs(sp.gray_u8(bgr[..., ::-1]) - want).max() > 20.0


@pytest.mark.parametrize("shape", [(31, 45), (64, 64), (9, 120)])
def test_sift_upsampling_against_scipy_zoom_and_torch_interpolate(shape):
    """resize(img, (2w, 2h), INTER_LINEAR): output pixel i samples the input at (i + 0.5) / 2 - 0.5,
    clamped at the borders = scipy.ndimage.zoom(order=1, grid_mode=True, mode='nearest') =
    torch's interpolate(mode='bilinear', align_corners=False)."""
    import sift_pyramid as sp
    import torch
    from scipy import ndimage
    rng = np.random.default_rng(22)
    img = (_smooth_image(rng, *shape) * 255).astype(np.float32)
    got = sp.resize_up2(img)
    assert got.shape == (2 * shape[0], 2 * shape[1])
    want = ndimage.zoom(img.astype(np.float64), 2, order=1, mode="nearest", grid_mode=True)
    assert np.abs(got - want).max() <= 1e-4                     # values up to 255
    want_t = torch.nn.functional.interpolate(torch.from_numpy(img.astype(np.float64))[None, None],
                                             scale_factor=2, mode="bilinear", align_corners=False)[0, 0]
    assert np.abs(got - want_t.numpy()).max() <= 1e-4


@pytest.mark.parametrize("shape", [(64, 48), (37, 53), (5, 121)])
def test_sift_decimation_against_torch_nearest(shape):
    """resize(img, (w // 2, h // 2), INTER_NEAREST): source index floor(dst * src / dst_size) =
    torch's interpolate(mode='nearest')."""
    import sift_pyramid as sp
    import torch
    rng = np.random.default_rng(23)
    img = rng.random(shape).astype(np.float32)
    got = sp.decimate2(img)
    want = torch.nn.functional.interpolate(torch.from_numpy(img)[None, None],
                                           size=(shape[0] // 2, shape[1] // 2), mode="nearest")[0, 0].numpy()
    assert np.array_equal(got, want)


def test_sift_scale_space_layers_against_scipy():
    """Every Gaussian layer of the restated scale space against SciPy: layer i = the previous layer
    filtered with the float64 taps of cv2.GaussianBlur's rule (aperture cvRound(8 sigma + 1) | 1,
    REFLECT_101 = 'mirror'), the sigmas from buildGaussianPyramid's schedule recomputed here; the
    next octave's base = every second pixel of layer 3; DoG = differences.  Then the schedule's
    defining property: layer i has the TOTAL sigma 1.6 * 2^(i/3) - it agrees with ONE wide Gaussian
    of that sigma (scipy.ndimage.gaussian_filter, truncated far out) applied to the doubled grey
    image of nominal sigma 1.0, away from the border and up to what the 4-sigma apertures cut off."""
    import sift_pyramid as sp
    from scipy import ndimage
    rng = np.random.default_rng(24)
    bgr = (np.stack([_smooth_image(rng, 96, 128) for _ in range(3)], axis=-1) * 255).astype(np.uint8)
    gauss, dog = sp.sift_pyramid(bgr)
    assert len(gauss) == sp.n_octaves(96, 128) == 7 and all(len(o) == 6 for o in gauss)

    def blur64(img, sigma):
        ksize = int(round(sigma * 8 + 1)) | 1
        x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2
        taps = np.exp(-x * x / (2 * sigma * sigma))
        taps /= taps.sum()
        out = ndimage.correlate1d(img, taps, axis=1, mode="mirror")
        return ndimage.correlate1d(out, taps, axis=0, mode="mirror")
    k = 2.0 ** (1.0 / 3)
    total = [1.6 * k ** i for i in range(6)]
    steps = [np.sqrt(1.6 ** 2 - 1.0)] + [np.sqrt(total[i] ** 2 - total[i - 1] ** 2) for i in range(1, 6)]
    assert np.allclose(steps[1:], sp.sigmas()[1:], rtol=1e-12)
    base = ndimage.zoom(sp.gray_u8(bgr).astype(np.float64), 2, order=1, mode="nearest", grid_mode=True)
    prev = blur64(base, steps[0])
    for o in range(3):                                   # (the small octaves reflect more than once)
        if o:
            prev = np.asarray(gauss[o - 1][3], np.float64)[::2, ::2]
            assert np.array_equal(gauss[o][0], gauss[o - 1][3][::2, ::2])
        assert np.abs(gauss[o][0] - prev).max() <= 2e-4
        for i in range(1, 6):
            want = blur64(np.asarray(gauss[o][i - 1], np.float64), steps[i])
            assert np.abs(gauss[o][i] - want).max() <= 2e-4, (o, i)
            assert np.array_equal(dog[o][i - 1], gauss[o][i] - gauss[o][i - 1])
    # total sigmas of the first octave: one wide Gaussian on the doubled image (nominal sigma 1.0)
    for i in range(6):
        wide = ndimage.gaussian_filter(base, np.sqrt(total[i] ** 2 - 1.0), mode="mirror", truncate=8.0)
        err = np.abs(np.asarray(gauss[0][i], np.float64) - wide)[24:-24, 24:-24].max()
        assert err <= 0.05, (i, err)                     # of 255: the chained 4-sigma apertures' tails
    # ... and across octaves: layer 3 has twice the base sigma, so the next octave starts at 1.6 again
    assert abs(total[3] - 3.2) < 1e-12


# ---- known answers for the keypoint stages (no third-party SIFT is installed) ---------------------
def _blob_scene(blobs, size=128, ramp=0.0, phi_deg=0.0, base=128.0):
    """uint8 BGR image (three equal channels): Gaussian blobs (cx, cy, std, amplitude) on a linear
    ramp of `ramp` levels per pixel along the direction phi (image coordinates, y down)."""
    yy, xx = np.mgrid[:size, :size].astype(np.float64)
    img = np.full((size, size), base)
    for cx, cy, s, a in blobs:
        img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    p = np.deg2rad(phi_deg)
    img += ramp * ((xx - size / 2) * np.cos(p) + (yy - size / 2) * np.sin(p))
    u8 = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return np.stack([u8] * 3, axis=-1)


SIFT_BLOBS = [(40.3, 38.7, 3.0, +90.0), (88.6, 40.2, 5.0, -90.0), (64.0, 92.5, 8.0, +90.0)]
SIFT_K = 2.0 ** (1.0 / 3.0)


def check_blob_keypoints(kps, blobs=SIFT_BLOBS):
    """A Gaussian blob of standard deviation s is, at scale sigma, a Gaussian of variance s^2 +
    sigma^2 and amplitude ~ s^2 / (s^2 + sigma^2); the difference of the scales k sigma and sigma is
    extremal where d/dv [1 / (s^2 + k^2 v) - 1 / (s^2 + v)] = 0, v = sigma^2, i.e. at
    sigma* = s / sqrt(k) - and SIFT labels the extremum with the LOWER scale of the pair, so the
    keypoint's size / 2 is s / sqrt(k) = 0.891 s (k = 2^(1/3); a little less for small s: the
    image's nominal blur of 0.5 px).  Its position is the blob's centre + 0.25 px on both axes:
    OpenCV doubles the image with pixel-CENTRE alignment and halves the coordinates afterwards."""
    for cx, cy, s, _ in blobs:
        best = min(kps, key=lambda k: (float(k["x"]) - cx - 0.25) ** 2 + (float(k["y"]) - cy - 0.25) ** 2)
        assert abs(float(best["x"]) - cx - 0.25) <= 0.06 and abs(float(best["y"]) - cy - 0.25) <= 0.06, (cx, cy)
        ratio = float(best["size"]) / 2.0 / s
        assert 0.98 / np.sqrt(SIFT_K) <= ratio <= 1.005 / np.sqrt(SIFT_K), (s, ratio)


def check_ramp_orientation(detect):
    """One weak blob (so that there IS an extremum) on a strong linear ramp: the gradient field around
    the keypoint is the ramp's constant gradient plus the blob's radial one, which cancels over the
    orientation window - the keypoint's angle is the ramp's direction (OpenCV's angle runs clockwise
    on the screen, y down: 360 - atan2(up - down, right - left))."""
    for phi in (30.0, 135.0, 250.0, 0.0, 90.0):
        kps = detect(_blob_scene([(64.0, 60.0, 8.0, 30.0)], ramp=1.0, phi_deg=phi))
        near = [k for k in kps if (float(k["x"]) - 64.25) ** 2 + (float(k["y"]) - 60.25) ** 2 < 0.25]
        assert len(near) == 1, (phi, len(near))
        diff = abs((float(near[0]["angle"]) - phi + 180.0) % 360.0 - 180.0)
        assert diff <= 4.0, (phi, float(near[0]["angle"]))


def check_rot90(detect, seed=5, size=96):
    """Turning the image by 90 degrees turns the keypoints with it and leaves the descriptors alone
    (each is computed in its keypoint's own frame).  The detector's + 0.25 px offset does not turn:
    positions agree to 0.8 px."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    base = ndimage.gaussian_filter(rng.random((size, size)), 2.0)
    base = ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)
    k0, d0 = detect(np.stack([base] * 3, -1), True)
    turned = np.rot90(base).copy()                       # new[i, j] = old[j, W - 1 - i]
    k1, d1 = detect(np.stack([turned] * 3, -1), True)
    assert len(k0) > 100 and abs(len(k0) - len(k1)) <= 0.1 * len(k0)
    dist, dang = [], []
    for i, k in enumerate(k0):
        xp, yp = float(k["y"]), size - 1 - float(k["x"])
        js = [j for j, q in enumerate(k1)
              if (float(q["x"]) - xp) ** 2 + (float(q["y"]) - yp) ** 2 < 0.64
              and abs(float(q["size"]) - float(k["size"])) < 0.2 * float(k["size"])]
        if js:
            dist.append(min(float(np.linalg.norm(d0[i] - d1[j])) for j in js))
            dang.append(min(abs((float(k1[j]["angle"]) - (float(k["angle"]) - 90.0) + 180.0) % 360.0 - 180.0)
                            for j in js))
    assert len(dist) >= 0.9 * len(k0)
    assert np.median(dist) <= 10.0 and np.percentile(dist, 90) <= 110.0      # descriptors have norm 512
    assert np.median(dang) <= 0.5


def _oracle_detect(bgr, with_desc=False):
    import sift_oracle as so
    import sift_pyramid as sp
    gauss, dog = sp.sift_pyramid(bgr)
    kps, des = so.detect_and_compute(gauss, dog)
    return (kps, des) if with_desc else kps


def test_sift_oracle_known_answers():
    """The keypoint stages' restatement (oracle/sift_oracle.py) against what can be known without
    OpenCV: blobs' positions and scales in closed form, a ramp's direction, a quarter turn."""
    check_blob_keypoints(_oracle_detect(_blob_scene(SIFT_BLOBS)))
    check_ramp_orientation(_oracle_detect)
    check_rot90(_oracle_detect)


# ---- the remaining restated OpenCV calls against third parties (round 6) ---------------------------
@pytest.mark.parametrize("shrink", [1.5, 2.5, 3.0, 4.0])
def test_shim_resize_against_scipy_zoom_and_torch(shrink):
    """cv2.resize(img, None, fx=1/shrink, fy=1/shrink) on uint8 (stitcher.py:419-420): bilinear at
    source coordinate (d + 0.5) * shrink - 0.5, no antialiasing, 11-bit fixed-point coefficients:
    within one level of the float bilinear of scipy.ndimage.zoom(order=1, grid_mode=True) and of
    torch's interpolate(bilinear, align_corners=False) at the same output size."""
    import cv2_shim as cv
    import torch
    from scipy import ndimage
    rng = np.random.default_rng(31)
    h, w = int(36 * shrink), int(50 * shrink)            # sizes the factor divides: every check runs
    img = (_smooth_image(rng, h, w, 3) * 255).astype(np.uint8)
    got = cv.resize(img, None, fx=1 / shrink, fy=1 / shrink).astype(np.float64)
    assert got.shape[:2] == (36, 50)
    t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
    want_t = torch.nn.functional.interpolate(t, size=(36, 50), mode="bilinear", align_corners=False)
    assert np.abs(got - want_t[0].permute(1, 2, 0).numpy()).max() <= 1.0
    want = np.stack([ndimage.zoom(img[..., c].astype(np.float64), 1 / shrink, order=1, mode="nearest",
                                  grid_mode=True) for c in range(3)], axis=-1)
    assert want.shape == got.shape and np.abs(got - want).max() <= 1.0
    # (and the two third parties agree with each other far below a level)
    assert np.abs(want - want_t[0].permute(1, 2, 0).numpy()).max() <= 1e-6


def test_shim_resize_half_is_the_area_mean_of_torch():
    """An exact 2 : 1 reduction goes through OpenCV's area path: rounded 2 x 2 box means =
    torch's avg_pool2d, rounded half up."""
    import cv2_shim as cv
    import torch
    rng = np.random.default_rng(32)
    img = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    got = cv.resize(img, None, fx=0.5, fy=0.5)
    mean = torch.nn.functional.avg_pool2d(torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None], 2)
    want = np.floor(mean[0].permute(1, 2, 0).numpy() + 0.5)
    assert np.array_equal(got, want.astype(np.uint8))


def test_shim_warp_perspective_against_scipy_map_coordinates():
    """cv2.warpPerspective(INTER_LINEAR, BORDER_TRANSPARENT) (stitcher.py:56-57): destination pixel
    (x, y) samples the source at M^-1 (x, y, 1), bilinear, and is written only when all four taps
    lie inside the source.  With a matrix whose inverse lands on the 1 / 32-pixel grid (a shift by
    multiples of 1/32 and a scale of 1/2) the fixed-point coordinates are exact: equal to SciPy's
    map_coordinates(order=1) to 1e-6 where written, zeros elsewhere; with a general homography the
    coordinates are rounded to 1/32 px: within the image's gradient times that step."""
    import cv2_shim as cv
    from scipy import ndimage
    rng = np.random.default_rng(33)
    h, w = 60, 84
    src = _smooth_image(rng, h, w, 4)
    gy, gx = np.gradient(src.astype(np.float64), axis=(0, 1))
    slope = float(np.sqrt(gx ** 2 + gy ** 2).max())
    for M, exact in ((np.array([[2.0, 0, 7.0 / 16], [0, 2.0, -5.0 / 16], [0, 0, 1.0]]), True),
                     (np.array([[0.95, 0.08, 3.3], [-0.06, 1.04, -2.1], [2e-4, -1e-4, 1.0]]), False)):
        dsize = (100, 72)
        got = cv.warpPerspective(src, M, dsize, flags=cv.INTER_LINEAR, borderMode=cv.BORDER_TRANSPARENT)
        inv = np.linalg.inv(M)
        ys, xs = np.mgrid[:dsize[1], :dsize[0]].astype(np.float64)
        den = inv[2, 0] * xs + inv[2, 1] * ys + inv[2, 2]
        sx = (inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]) / den
        sy = (inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]) / den
        inside = (sx >= 0) & (sx < w - 1) & (sy >= 0) & (sy < h - 1)
        edge = (np.abs(sx) < 0.05) | (np.abs(sx - (w - 1)) < 0.05) | (np.abs(sy) < 0.05) | \
               (np.abs(sy - (h - 1)) < 0.05)           # (a rounded coordinate may fall on the other side)
        for ch in range(4):
            want = ndimage.map_coordinates(src[..., ch].astype(np.float64), [sy, sx], order=1,
                                           mode="constant", cval=0.0)
            tol = 1e-6 if exact else slope * (np.sqrt(2.0) / 64) + 1e-6
            ok = inside & ~edge
            assert np.abs(got[..., ch][ok] - want[ok]).max() <= tol, (exact, ch)
        assert np.all(got[~inside & ~edge] == 0)
        assert inside.mean() > 0.3


def test_shim_pyr_up_against_scipy():
    """cv2.pyrUp (blend.py:127-134): the source zero-stuffed to twice its size and filtered with
    [1 4 6 4 1] / 8 per axis - away from the borders (whose closed forms are OpenCV's own) equal to
    SciPy's correlate1d on the zero-stuffed image."""
    import cv2_shim as cv
    from scipy import ndimage
    rng = np.random.default_rng(34)
    img = _smooth_image(rng, 33, 47)
    got = cv.pyrUp(img)
    stuffed = np.zeros((66, 94))
    stuffed[0::2, 0::2] = img
    taps = np.array([1, 4, 6, 4, 1], np.float64) / 8
    want = ndimage.correlate1d(ndimage.correlate1d(stuffed, taps, axis=1, mode="constant"), taps, axis=0,
                               mode="constant")
    assert np.abs(got[2:-2, 2:-2] - want[2:-2, 2:-2]).max() <= 1e-6
