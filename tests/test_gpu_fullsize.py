"""GPU: parity of the default kernels at plane level and at BASELINE sizes.

* the matrix-core blur (``pano_multiband_blur``) compared PLANE BY PLANE with the
  oracle's ``cv2.GaussianBlur`` (stitcher.py:218, 226), not only through the
  collapsed mosaic; stated bound 1e-6 absolute on [0, 1] data (split-float16
  operands, float32 accumulate), the measured maximum is printed;
* BASELINE config 2 (8 x 1080p) at full size against the oracle: valid mask and
  crop rectangle bit-exact, uint8 mosaic <= 1 LSB, float mosaic <= 1e-4 rel-L2;
* the crop on the real config-3 valid mask and on a config-5-width mask
  (>= 46 079 columns) against the oracle, bit-exact; the config-3 valid mask
  itself against the oracle's inverse maps;
* config 4: the Gaussian / DoG scale space of one 3840 x 2160 frame against the
  oracle (all 11 octaves, 7680 x 4320 base);
* config 5 at full size (120 x 8K, closed 360 degree sweep, L = 6, 4948 x 46 079 mosaic),
  which no CPU oracle finishes: size-independent properties - column strips of a
  world-8 run equal the same columns of the whole mosaic bit for bit, the valid mask
  and crop rectangle equal the oracle's (integer work, bit-exact), the seam-straddling
  frame is split into records.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4
PLANE_TOL = 1e-6        # blurred planes, absolute, data in [0, 1]


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))


def _sharp_rgba(ref_patches, own):
    """The blur input of stitcher.py:207-208: alpha replaced by owner == index."""
    out = []
    for idx, (warped, _, ir) in enumerate(ref_patches):
        rgba = warped.copy()
        rgba[..., 3] = (own[ir] == idx).astype(np.float32)
        out.append(rgba)
    return out


@pytest.mark.parametrize("blur", ["mfma", "valu"])
@pytest.mark.parametrize("kind,levels,size", [("A", 5, (400, 240)), ("B", 5, (330, 200)),
                                              ("A", 8, (300, 170)), ("B", 8, (400, 240))])
def test_multiband_blur_planes_match_oracle(oracle, blur, kind, levels, size):
    """Whole-patch records through ``pano_multiband_blur``: every level's four
    blurred planes (R, G, B and the sharp alpha) against ``oracle.gaussian_blur``.
    L = 5: 33 / 57 / 73 / 87 taps (K-steps C = 1, 2, 3, 3); L = 8 adds 97 / 105 / 115
    taps (C = 4); patch sizes are not multiples of 32, so every tile column and row
    ends in a ragged tile.  Noise ("A") is the stress input, "B" the smooth one."""
    from pano360_amd import engine, synth
    eng = engine.Engine(blur=blur)
    w, h = size
    imgs, rots, intrs = synth.make_scene(4, w, h, sweep_deg=60.0, jitter=0.01, seed=90 + levels,
                                         kind=kind)
    shapes = [im.shape[:2] for im in imgs]
    plan = eng.upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9))
    n_blur = levels - 1
    patches, _ = eng.warp_all(eng.upload_frames(imgs), plan, n_blur)
    table = engine.patch_table(patches, eng)
    owner, valid = eng.ownership(table, plan.shape)
    eng.blur_and_compose(table, owner, valid, plan.shape, levels)
    _, ref_patches, _ = oracle.warp_all(imgs, rots, intrs, True, 10 ** 9)
    own = oracle.ownership(ref_patches, plan.shape)
    assert np.array_equal(owner.cpu().numpy().astype(np.int32), own)
    sig = engine.level_sigmas(levels)
    worst = 0.0
    for dp, rgba in zip(patches, _sharp_rgba(ref_patches, own)):
        got = dp.blurred[:, :, :, :dp.w].cpu().numpy()              # [level][channel][h][w]
        assert np.isfinite(got).all()
        for k, s in enumerate(sig):
            want = oracle.gaussian_blur(rgba, engine.gaussian_ksize(s), s)
            err = np.abs(got[k].transpose(1, 2, 0) - want).max()
            worst = max(worst, float(err))
            assert err <= PLANE_TOL, (blur, kind, levels, k, err)
    print(f"blurred planes ({blur}, {kind}, L={levels}): max abs error {worst:.3e}")


@pytest.mark.parametrize("blur", ["mfma", "valu"])
def test_multiband_blur_planes_match_reference_golden(blur):
    """The same comparison against planes recorded while the REFERENCE ran (a spying
    cv2.GaussianBlur inside its multiband_blend, tests/golden/scene_small_noise.npz): all
    four levels of one patch of the small noise scene."""
    from conftest import load_golden, scene_inputs
    from pano360_amd import engine
    g = load_golden("scene_small_noise")
    imgs, rots, intrs, mr = scene_inputs(g)
    eng = engine.Engine(blur=blur)
    plan = eng.upload_plan(engine.Plan([im.shape[:2] for im in imgs], rots, intrs, True, mr))
    patches, _ = eng.warp_all(eng.upload_frames(imgs), plan, 4)
    table = engine.patch_table(patches, eng)
    owner, valid = eng.ownership(table, plan.shape)
    eng.blur_and_compose(table, owner, valid, plan.shape, 5)
    dp = patches[int(g["blur_patch"])]
    got = dp.blurred[:, :, :, :dp.w].cpu().numpy()
    for k in range(4):
        err = np.abs(got[k].transpose(1, 2, 0) - g[f"blur_out_{k}"]).max()
        assert err <= PLANE_TOL, (blur, k, err)


@pytest.mark.parametrize("blur", ["mfma", "valu"])
def test_windowed_blur_planes_match_oracle(oracle, blur):
    """The fused path's records (windows V, rectangles A cut out of the patches, tiles
    anchored at multiples of 32 in patch coordinates): the blurred copies over A equal the
    oracle's blur of the WHOLE patch cut to A, every level and channel."""
    from pano360_amd import engine, synth
    eng = engine.Engine(blur=blur)
    imgs, rots, intrs = synth.make_scene(6, 640, 360, sweep_deg=50.0, jitter=0.01, seed=31,
                                         kind="A")
    shapes = [im.shape[:2] for im in imgs]
    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    _, _, _, fused = eng.stitch(eng.upload_frames(imgs), plan, "multiband", 5, shortcut=False)
    import torch
    torch.cuda.synchronize()
    _, ref_patches, _ = oracle.warp_all(imgs, rots, intrs, True, 10 ** 9)
    own = oracle.ownership(ref_patches, plan.shape)
    rgbas = _sharp_rgba(ref_patches, own)
    sig = engine.level_sigmas(5)
    want = {}
    arena, base = fused.blurred, fused.blurred.data_ptr()
    worst, seen = 0.0, 0
    for rec in fused.table.host:
        idx, ah, aw, ap = int(rec["index"]), int(rec["ah"]), int(rec["aw"]), int(rec["apitch"])
        ay0, ax0 = int(rec["ay0"]), int(rec["ax0"])
        off = (int(rec["blurred"]) - base) // 4
        got = arena[off:off + 4 * 4 * ah * ap].view(4, 4, ah, ap)[:, :, :, :aw].cpu().numpy()
        if idx not in want:
            want[idx] = [oracle.gaussian_blur(rgbas[idx], engine.gaussian_ksize(s), s)
                         for s in sig]
        for k in range(4):
            ref = want[idx][k][ay0:ay0 + ah, ax0:ax0 + aw]
            err = np.abs(got[k].transpose(1, 2, 0) - ref).max()
            worst = max(worst, float(err))
            assert err <= PLANE_TOL, (blur, idx, k, err)
        seen += 1
    assert seen >= len(imgs)
    assert (int(fused.table.host["aw"].max()) < max(r[3] - r[2] for r in plan.rects))
    print(f"windowed blurred planes ({blur}): max abs error {worst:.3e} over {seen} records")


def test_cfg2_full_size_against_oracle(eng, oracle):
    """BASELINE config 2 as the bench runs it (8 x 1920x1080, 140 degree sweep, native
    resolution, L = 5; smooth pixel set B for the relative-error criterion): valid mask
    and crop rectangle bit-exact, uint8 mosaic within one level, float mosaic within
    1e-4 relative L2 of ``oracle.stitch``."""
    from pano360_amd import engine, synth
    cfg = synth.CONFIGS["cfg2"]
    imgs, rots, intrs = synth.make_scene(cfg["n"], cfg["width"], cfg["height"],
                                         sweep_deg=cfg["sweep_deg"], seed=0, kind="B")
    shapes = [im.shape[:2] for im in imgs]
    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    assert plan.shape == (1237, 6400)
    mosaic, fl, valid, _ = eng.stitch(eng.upload_frames(imgs), plan, "multiband", 5,
                                      want_float=True)
    oplan, ref_patches, _ = oracle.warp_all(imgs, rots, intrs, True, 10 ** 9)
    assert oplan.shape == plan.shape and oplan.rects == plan.rects
    ref_valid = oracle.valid(ref_patches, plan.shape)
    got_valid = valid.cpu().numpy().astype(bool)
    assert np.array_equal(got_valid, ref_valid)
    assert eng.crop_rect(valid) == oracle.crop_rect(ref_valid)
    ref_u8, ref_f = oracle.multiband_blend(ref_patches, plan.shape, 5, return_float=True)
    got = mosaic.cpu().numpy()
    lsb = np.abs(got.astype(int) - ref_u8.astype(int))
    rel = rel_l2(fl.cpu().numpy(), ref_f)
    print(f"cfg2 full size: rel-L2 {rel:.2e}, {int((lsb > 0).sum())} of {lsb.size} uint8 values "
          f"differ by one level")
    assert lsb.max() <= 1 and rel <= REL_TOL
    # the paste and linear blenders on the unpadded plan: exact
    for blend in ("none", "linear"):
        plan_l = engine.Plan(shapes, rots, intrs, False, 10 ** 9)
        got_l, _, _, _ = eng.stitch(eng.upload_frames(imgs), plan_l, blend)
        assert np.array_equal(got_l.cpu().numpy(),
                              oracle.stitch(imgs, rots, intrs, blend, max_resolution=10 ** 9))


def test_cfg3_valid_and_crop_against_oracle(eng, oracle):
    """BASELINE config 3 (32 x 4K): the valid mask of the fused path against the OR of
    the oracle's inverse-map masks (stitcher.py:266-271, 305 MP of float64 maps on the
    host cores), and the crop rectangle of that mask (2474 x 13760) bit-exact."""
    from pano360_amd import engine, synth
    cfg = synth.CONFIGS["cfg3"]
    n, w, h = cfg["n"], cfg["width"], cfg["height"]
    rots, intrs = synth.make_cameras(n, w, h, sweep_deg=cfg["sweep_deg"])
    shapes = [(h, w)] * n
    plan = eng.upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9))
    owner, valid = eng.ownership_cameras(plan)
    oplan = oracle.Plan(shapes, rots, intrs, True, 10 ** 9)
    assert oplan.shape == plan.shape == (2474, 13760) and oplan.rects == plan.rects
    ref_valid = np.zeros(plan.shape, bool)
    for proj, rect in zip(oplan.projs, oplan.rects):
        _, _, mask = oracle.inverse_map(proj, oplan, rect, (h, w))
        ref_valid[rect[0]:rect[1], rect[2]:rect[3]] |= ~mask
    got = valid.cpu().numpy().astype(bool)
    assert np.array_equal(got, ref_valid)
    assert eng.crop_rect(valid) == oracle.crop_rect(ref_valid)
    # with the frames: the stitch reports the same mask
    frames = eng.upload_frames([synth.make_frame(0, w, h, "A")] * 2)
    _, _, valid2, _ = eng.stitch([frames[i % 2] for i in range(n)],
                                 engine.Plan(shapes, rots, intrs, True, 10 ** 9), "multiband", 5)
    assert np.array_equal(valid2.cpu().numpy().astype(bool), ref_valid)


@pytest.mark.parametrize("shape", [(4948, 46079), (150, 65535), (3000, 46080)])
def test_crop_at_cfg5_width(eng, oracle, shape):
    """crop_mosaic's rectangle on masks as wide as config 5's mosaic (46 079 columns; the
    kernel's limit is 65 535): a covered band with wavy upper and lower borders, sparse
    holes and an invalid column run, bit-exact against the oracle."""
    import torch
    H, W = shape
    rng = np.random.default_rng(H + W)
    x = np.arange(W)
    top = (0.08 * H * (1 + np.sin(x / 911.0))).astype(np.int64)
    bot = H - (0.06 * H * (1 + np.cos(x / 1501.0))).astype(np.int64)
    rows = np.arange(H)[:, None]
    mask = (rows >= top[None, :]) & (rows < bot[None, :])
    holes = rng.integers(0, H * W, size=max(H * W // 2_000_000, 3))
    mask.reshape(-1)[holes] = False
    mask[:, W // 3:W // 3 + 7] = False
    dev = torch.from_numpy(mask.view(np.uint8)).to(eng.device)
    assert eng.crop_rect(dev) == oracle.crop_rect(mask)


def test_cfg4_scale_space_of_a_4k_frame(eng, oracle):
    """BASELINE config 4 at full size, one frame: Gaussian and DoG pyramid of a
    3840 x 2160 frame (first octave -1: a 7680 x 4320 base, 11 octaves) against the
    oracle (oracle/sift_pyramid.py with the C oracle's GaussianBlur, which equals the
    NumPy shim bit for bit).  Values are on the 0..255 scale; one FMA per tap against
    multiply + add: <= 2e-4 on the Gaussian layers, <= 4e-4 on the DoG layers."""
    import sift_pyramid as ref
    from pano360_amd import features, synth
    img = synth.make_frame(11, 3840, 2160, "B")
    gauss, dog = features.sift_pyramid_device(eng.upload_frames([img])[0])
    want_g, want_d = ref.sift_pyramid(
        img, blur=lambda im, s: oracle.gaussian_blur(im, oracle.gaussian_ksize(s), s))
    assert len(gauss) == len(want_g) == ref.n_octaves(2160, 3840) == 11
    worst_g = worst_d = 0.0
    for o in range(len(want_g)):
        got_g, got_d = gauss[o].cpu().numpy(), dog[o].cpu().numpy()
        assert got_g.shape == (6,) + want_g[o][0].shape and got_d.shape[0] == 5
        for layer in range(6):
            worst_g = max(worst_g, float(np.abs(got_g[layer] - want_g[o][layer]).max()))
        for layer in range(5):
            worst_d = max(worst_d, float(np.abs(got_d[layer] - want_d[o][layer]).max()))
    print(f"cfg4 4K scale space: max abs error gauss {worst_g:.2e}, dog {worst_d:.2e}")
    assert worst_g <= 2e-4 and worst_d <= 4e-4


def test_two_contexts_on_two_streams_concurrently(eng):
    """A context owns everything the library remembers (work lists, tile flags, tap
    tables, timing): two engines - two ``pano_ctx`` - driven from two host threads on two
    streams of one device at the same time each reproduce the single-stream mosaic bit for
    bit, strip by strip (the column-strip decomposition of the multi-GPU path, 2 and 3
    ranks emulated in turn by either thread)."""
    import threading

    import torch
    from pano360_amd import dist as pdist
    from pano360_amd import engine, synth
    imgs, rots, intrs = synth.make_scene(10, 480, 270, sweep_deg=120.0, jitter=0.01, seed=41,
                                         kind="A")
    shapes = [im.shape[:2] for im in imgs]
    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    whole, _, _, _ = eng.stitch(eng.upload_frames(imgs), plan, "multiband", 5)
    torch.cuda.synchronize()
    results, errors = {}, []

    def worker(tag, world, blur):
        try:
            stream = torch.cuda.Stream(eng.device)
            with torch.cuda.stream(stream):
                mine = engine.Engine(eng.device, blur=blur)
                for _ in range(4):
                    strips, _ = pdist.emulate_on_one_device(mine, imgs, rots, intrs, 5, world)
                stream.synchronize()
            results[tag] = strips
        except Exception as err:      # noqa: BLE001 - reported below, in the main thread
            errors.append((tag, repr(err)))

    threads = [threading.Thread(target=worker, args=args)
               for args in (("a", 2, "mfma"), ("b", 3, "mfma"), ("c", 2, "valu"))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert torch.equal(results["a"], whole) and torch.equal(results["b"], whole)
    # the vector-ALU blur rounds differently: within one level of the matrix-core mosaic
    assert (results["c"].int() - whole.int()).abs().max().item() <= 1


@pytest.mark.parametrize("case", ["random", "unit", "d64", "d100", "one", "ties", "tiny"])
def test_knn2_kernel_is_exact(eng, case):
    """``pano_knn2`` (the search behind flann_matching, features.py:222-232) against a
    float64 brute force: the same two nearest rows for every query, distances to float32
    rounding.  Cases: random 128-d rows, unit-norm RootSIFT-like rows with planted near
    duplicates, 64-d rows (four k-steps), 100-d and 5-d rows (a ragged last k-step), one query against 70 000 rows,
    duplicated train rows (exact ties: the proof of the
    ranking fails and those queries are rescanned), and two / three train rows."""
    import torch
    from pano360_amd import features
    rng = np.random.default_rng({"random": 1, "unit": 2, "d64": 3, "ties": 4, "tiny": 5, "d100": 6,
                                 "one": 7}[case])
    d = {"d64": 64, "d100": 100, "one": 5}.get(case, 128)      # 100, 5: not a multiple of a k-step
    nq, nt = {"tiny": (45, 3), "one": (1, 70000)}.get(case, (700, 1333))
    a = rng.random((nq, d)).astype(np.float32)
    b = rng.random((nt, d)).astype(np.float32)
    if case == "unit":
        a, b = np.sqrt(a / a.sum(1, keepdims=True)), np.sqrt(b / b.sum(1, keepdims=True))
        b[:300] = a[rng.permutation(nq)[:300]] + 1e-3 * rng.random((300, d)).astype(np.float32)
    if case == "ties":
        b[500:700] = b[100:300]                          # every row of 100..299 exists twice
        a[:200] = b[100:300] + 1e-4 * rng.random((200, d)).astype(np.float32)
    for rows in ((a, b), (a, b[:2])) if case == "tiny" else ((a, b),):
        q, t = rows
        idx, dist, rescans = features.knn2_device(torch.from_numpy(q).to(eng.device),
                                                  torch.from_numpy(t).to(eng.device), eng=eng,
                                                  want_rescans=True)
        idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
        ref = np.sqrt(((q[:, None, :].astype(np.float64) - t[None, :, :]) ** 2).sum(-1))
        order = np.argsort(ref, axis=1, kind="stable")[:, :2]
        want = np.take_along_axis(ref, order, axis=1)
        got = np.take_along_axis(ref, idx, axis=1)          # float64 distance of the rows found
        # the rows found are the nearest two (equal distances may swap labels)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
        np.testing.assert_allclose(dist, want, rtol=2e-6, atol=1e-6)
        if case != "ties":
            assert np.array_equal(idx, order)
        else:
            assert rescans > 0                               # the duplicated rows defeated the proof
            assert (idx[:, 0] != idx[:, 1]).all()
    print(f"knn2 {case}: {rescans} of {nq} queries rescanned")


def test_cfg5_full_size_properties(eng, oracle):
    """BASELINE config 5 at full size.  The frames' content cycles through six distinct
    8K images (the properties do not depend on it; 120 distinct ones would be 12 GB)."""
    import torch
    from pano360_amd import dist as pdist
    from pano360_amd import engine, synth
    cfg = synth.CONFIGS["cfg5"]
    n, w, h, levels = cfg["n"], cfg["width"], cfg["height"], cfg["n_levels"]
    rots, intrs = synth.make_cameras(n, w, h, sweep_deg=cfg.get("sweep_deg"),
                                     step_deg=cfg.get("step_deg"))
    shapes = [(h, w)] * n
    base = eng.upload_frames([synth.make_frame(i, w, h, "A") for i in range(6)])
    frames = [base[i % 6] for i in range(n)]
    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    assert plan.shape == (4948, 46079)
    whole, _, valid, patches = eng.stitch(frames, plan, "multiband", levels)
    assert len(patches) > n          # a frame across the +-pi seam owns pixels at both ends
    # integer work against the oracle: valid mask from the oracle's inverse-map masks is too
    # slow at this size, the crop of THIS mask is not
    mask = valid.cpu().numpy()
    assert mask.any(axis=0).all()    # closed sweep: every column is covered somewhere
    assert eng.crop_rect(valid) == oracle.crop_rect(mask)
    # ownership: the pruned kernel against the exhaustive one on the whole 228 MP map (the
    # rectangles of the two frames across the +-pi seam stop 32 columns short of the mosaic's
    # end although the frames reach it: a lower bound of theirs once pruned the only candidate)
    from pano360_amd import _lib
    eng.upload_plan(plan)
    pruned, pruned_valid = (v.clone() for v in eng.ownership_cameras(plan))
    try:
        eng.set_option(_lib.OPT_OWN_PRUNE, 3)              # round 4's kernel: one level of bounds
        one_level, one_level_valid = (v.clone() for v in eng.ownership_cameras(plan))
        eng.set_option(_lib.OPT_OWN_PRUNE, 0)
        exact, exact_valid = eng.ownership_cameras(plan)
    finally:
        eng.set_option(_lib.OPT_OWN_PRUNE, 1)
    assert torch.equal(pruned, exact) and torch.equal(pruned_valid, exact_valid)
    assert torch.equal(one_level, exact) and torch.equal(one_level_valid, exact_valid)
    assert int((exact[200:4700, plan.shape[1] - 32:] == -1).sum()) == 0
    # strips of a world-8 run: first, a middle and the last rank (the last one holds the seam)
    world = 8
    for rank in (0, 3, world - 1):
        st = pdist.ShardedStitcher(eng, shapes, rots, intrs, levels, rank, world, exchange=None)
        assert len(st.my_frames) < n
        # the rank's own plan: sin / cos tables on its columns only, NaN elsewhere
        mine = eng.upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9,
                                           table_cols=st.table_cols))
        strip, _, _, _ = eng.multiband_fused([frames[i] for i in st.my_frames], mine, levels,
                                             frame_ids=st.my_frames, strip=st.strip)
        c0, c1 = st.strip
        assert torch.equal(strip[:, c0:c1], whole[:, c0:c1]), rank


def _oracle_window(oracle, imgs, rgba_cache, rots, intrs, padded, window):
    """The reference's patches cut to a window of the mosaic: per camera whose patch rectangle
    meets it, inverse map + mask + remap of the intersection (stitcher.py:299-319 on a
    sub-rectangle; every stage up to the linear / none blend and the ownership argmax is local
    to a pixel, so the window of the result is the result on the window)."""
    shapes = [im.shape[:2] for im in imgs]
    plan = oracle.Plan(shapes, rots, intrs, padded, 10 ** 9)
    wy0, wy1, wx0, wx1 = window
    patches = []
    for i, (proj, rect) in enumerate(zip(plan.projs, plan.rects)):
        y0, y1, x0, x1 = max(rect[0], wy0), min(rect[1], wy1), max(rect[2], wx0), min(rect[3], wx1)
        if y0 >= y1 or x0 >= x1:
            # keeps the camera's index in the argmax (:196-204): a 2 x 2 patch of masked zeros
            # contributes nothing anywhere (an empty one has no REFLECT_101 border to blur at)
            patches.append((np.zeros((2, 2, 4), np.float32), np.ones((2, 2), bool),
                            np.s_[0:2, 0:2]))
            continue
        key = id(imgs[i])
        if key not in rgba_cache:
            rgba_cache[key] = oracle.add_weights(imgs[i])
        mx, my, mask = oracle.inverse_map(proj, plan, (y0, y1, x0, x1), shapes[i])
        warped = oracle.remap(rgba_cache[key], mx, my)
        warped[..., 3][mask] = 0.0                                     # stitcher.py:317
        patches.append((warped, mask, np.s_[y0 - wy0:y1 - wy0, x0 - wx0:x1 - wx0]))
    return plan, patches, (wy1 - wy0, wx1 - wx0)


def test_cfg5_windows_against_oracle(eng, oracle):
    """BASELINE config 5 at full size against the ORACLE on windows of the mosaic: ownership
    and valid (the multiband path's padded rectangles), fused linear and none blends (unpadded
    rectangles) - all bit-exact.  Windows: both ends of the closed sweep (where the seam
    frames' rectangles stop short of the mosaic), a seam in the middle, the top and the
    bottom-right corner."""
    import torch
    from pano360_amd import engine, synth
    cfg = synth.CONFIGS["cfg5"]
    n, w, h = cfg["n"], cfg["width"], cfg["height"]
    rots, intrs = synth.make_cameras(n, w, h, sweep_deg=cfg.get("sweep_deg"),
                                     step_deg=cfg.get("step_deg"))
    shapes = [(h, w)] * n
    host = [synth.make_frame(i, w, h, "A") for i in range(4)]
    imgs = [host[i % 4] for i in range(n)]
    base = eng.upload_frames(host)
    frames = [base[i % 4] for i in range(n)]
    plan = eng.upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9))
    H, W = plan.shape
    owner, valid = (v.cpu().numpy() for v in eng.ownership_cameras(plan))
    plan_u = engine.Plan(shapes, rots, intrs, False, 10 ** 9)
    assert plan_u.shape == (H, W)
    linear = eng.stitch(frames, plan_u, "linear")[0].cpu().numpy()
    none = eng.stitch(frames, plan_u, "none")[0].cpu().numpy()
    seams = np.nonzero(np.diff(owner[2448]))[0]
    mid = int(seams[np.argmin(np.abs(seams - 23000))])          # a seam near the middle
    windows = [(2000, 2096, W - 128, W), (2000, 2096, 0, 128), (2400, 2496, mid - 64, mid + 64),
               (0, 96, 20000, 20128), (H - 96, H, W - 128, W)]
    cache, seen = {}, []
    for win in windows:
        wy0, wy1, wx0, wx1 = win
        _, patches, shape = _oracle_window(oracle, imgs, cache, rots, intrs, True, win)
        assert np.array_equal(owner[wy0:wy1, wx0:wx1], oracle.ownership(patches, shape)), win
        assert np.array_equal(valid[wy0:wy1, wx0:wx1] != 0, oracle.valid(patches, shape)), win
        _, patches, shape = _oracle_window(oracle, imgs, cache, rots, intrs, False, win)
        assert np.array_equal(linear[wy0:wy1, wx0:wx1], oracle.linear_blend(patches, shape)), win
        assert np.array_equal(none[wy0:wy1, wx0:wx1], oracle.no_blend(patches, shape)), win
        seen.append((len(np.unique(owner[wy0:wy1, wx0:wx1])), int(linear[wy0:wy1, wx0:wx1].max()),
                     sum(not p[1].all() or p[1].size > 4 for p in patches)))
    print("cfg5 windows (owners, max linear value, cameras):", seen)
    assert all(s[1] > 0 and s[2] >= 2 for s in seen) and seen[2][0] >= 2 and any(s[0] >= 3 for s in seen)


def _seam_windows(owner, S, rng, limit=None):
    """One S x S window on every SEAM of the mosaic - a pair (a, b) of owners that are horizontal
    neighbours somewhere - centred on the seam at a seeded random row of its own (drawn among the
    rows where that pair is adjacent, away from the top and bottom by S / 2 where it can be); with
    ``limit`` a seeded choice of that many seams.  Also returns the mosaic's total seam length in
    pixel rows (horizontal owner changes between two valid owners)."""
    H, W = owner.shape
    change = (owner[:, 1:] != owner[:, :-1]) & (owner[:, 1:] >= 0) & (owner[:, :-1] >= 0)
    ys, xs = np.nonzero(change)
    seam_rows = len(ys)
    key = owner[ys, xs].astype(np.int64) * 65536 + owner[ys, xs + 1].astype(np.int64)
    pairs = [int(k) for k in np.unique(key)]
    if limit is not None and len(pairs) > limit:
        pairs = [pairs[i] for i in sorted(rng.choice(len(pairs), size=limit, replace=False))]
    windows = []
    for k in pairs:
        at = np.nonzero(key == k)[0]
        inner = at[(ys[at] >= S // 2) & (ys[at] < H - S // 2)]
        pick = int(rng.choice(inner if len(inner) else at))
        r, x = int(ys[pick]), int(xs[pick])
        y0, x0 = min(max(r - S // 2, 0), H - S), min(max(x - S // 2, 0), W - S)
        windows.append((y0, y0 + S, x0, x0 + S))
    return windows, [(k >> 16, k & 65535) for k in pairs], seam_rows


def _check_multiband_windows(eng, oracle, name, rots, intrs, w, h, levels, plan, windows):
    """The fused multiband mosaic (interior shortcut on) against the oracle on ``windows`` of the
    mosaic, two passes: noise frames (pixel set A) - uint8 within one level, at most 0.4 % of a
    window's values off by that level; smooth frames (pixel set B, SURVEY 8d's set for the float
    criterion) - the FLOAT mosaic before the uint8 truncation within 1e-4 relative L2 on every
    window, uint8 within one level.  Returns the number of mosaic rows compared per window."""
    from pano360_amd import engine, synth
    n = len(rots)
    H, W = plan.shape
    R = max(engine.gaussian_ksize(s) // 2 for s in engine.level_sigmas(levels))
    rows_compared = []
    for kind, distinct in (("A", 4), ("B", 2)):
        host = [synth.make_frame(i, w, h, kind) for i in range(distinct)]
        imgs = [host[i % distinct] for i in range(n)]
        base = eng.upload_frames(host)
        frames = [base[i % distinct] for i in range(n)]
        mosaic, fl, _, _ = eng.stitch(frames, plan, "multiband", levels, want_float=kind == "B")
        mosaic = mosaic.cpu().numpy()
        cache, worst, total, worst_rel = {}, 0, 0, 0.0
        for win in windows:
            wy0, wy1, wx0, wx1 = win
            S_y, S_x = wy1 - wy0, wx1 - wx0
            _, patches, shape = _oracle_window(oracle, imgs, cache, rots, intrs, True, win)
            iy0, iy1 = (R if wy0 > 0 else 0), S_y - (R if wy1 < H else 0)
            ix0, ix1 = (R if wx0 > 0 else 0), S_x - (R if wx1 < W else 0)
            if kind == "B":
                ref, ref_f = oracle.multiband_blend(patches, shape, levels, return_float=True)
                got_f = fl[wy0:wy1, wx0:wx1].cpu().numpy()[iy0:iy1, ix0:ix1]
                want_f = ref_f[iy0:iy1, ix0:ix1]
                rel = float(np.linalg.norm(got_f.astype(np.float64) - want_f)
                            / np.linalg.norm(want_f.astype(np.float64)))
                assert rel <= 1e-4, (name, win, rel)          # float mosaic, pixel set B
                worst_rel = max(worst_rel, rel)
            else:
                ref = oracle.multiband_blend(patches, shape, levels)
                rows_compared.append(iy1 - iy0)
            got = mosaic[wy0:wy1, wx0:wx1][iy0:iy1, ix0:ix1].astype(np.int32)
            diff = np.abs(got - ref[iy0:iy1, ix0:ix1].astype(np.int32))
            assert diff.max() <= 1, (kind, win, int(diff.max()), int((diff > 1).sum()))
            assert ref[iy0:iy1, ix0:ix1].max() > 0
            worst = max(worst, float((diff > 0).mean()))
            total += diff.size
        print(f"{name} full size, multiband on {len(windows)} windows, pixel set {kind}: {total} values "
              f"compared, at most {100 * worst:.3f} % of a window differ by one level"
              + (f", float mosaic rel-L2 at most {worst_rel:.2e}" if kind == "B" else ""))
        assert worst < 0.004                    # five times the measured 0.0008
        del mosaic, fl, frames, base, host, imgs, cache
    return rows_compared


@pytest.mark.parametrize("name", ["cfg3", "cfg5"])
def test_full_size_multiband_windows_against_oracle(eng, oracle, name):
    """The multiband mosaic of BASELINE configs 3 and 5 AT FULL SIZE against the oracle, on
    windows.  Every level blurs the ORIGINAL warped patch (stitcher.py:226), so a pixel of the
    mosaic depends on the patches within the largest Gaussian radius R of it only: the oracle's
    multiband_blend on the patches cut to a 288 x 288 window (its blur reflecting at the cut)
    equals the reference's on the whole mosaic on the window shrunk by R from every cut side.
    Windows: one on EVERY seam of config 3 (31) at a seeded random row, sixteen seams of config 5
    (seeded choice), plus both ends of the sweep (config 5: the seam-straddling frames), the top
    and a corner.  Criteria: ``_check_multiband_windows``."""
    from pano360_amd import engine, synth
    cfg = synth.CONFIGS[name]
    n, w, h, levels = cfg["n"], cfg["width"], cfg["height"], cfg["n_levels"]
    rots, intrs = synth.make_cameras(n, w, h, sweep_deg=cfg.get("sweep_deg"),
                                     step_deg=cfg.get("step_deg"))
    shapes = [(h, w)] * n
    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    H, W = plan.shape
    owner = eng.ownership_cameras(eng.upload_plan(plan))[0].cpu().numpy()
    S = 288
    rng = np.random.default_rng(606 + n)
    windows, pairs, seam_rows = _seam_windows(owner, S, rng, None if name == "cfg3" else 16)
    assert len(windows) == (31 if name == "cfg3" else 16), len(windows)
    n_seam_windows = len(windows)
    mid = min(max(W // 2 - S // 2, 0), W - S)
    windows += [(H // 2, H // 2 + S, W - S, W), (H // 2, H // 2 + S, 0, S),     # both ends
                (0, S, mid, mid + S), (H - S, H, W - S, W)]                     # top, a corner
    del owner
    rows = _check_multiband_windows(eng, oracle, name, rots, intrs, w, h, levels, plan, windows)
    print(f"{name}: {n_seam_windows} seam windows, {sum(rows[:n_seam_windows])} of {seam_rows} seam "
          f"rows compared = {100.0 * sum(rows[:n_seam_windows]) / seam_rows:.2f} % of the seam length")


def test_cfg3_jittered_rig_full_size_against_oracle(eng, oracle):
    """Config 3's cameras with N(0, 0.01 rad) on all three axes (SURVEY 8d's recipe; every other
    full-size scene is a pure-yaw rig): tilted, uneven seams at FULL size.  Bit-exact: the valid
    mask of the whole mosaic against the OR of the oracle's inverse-map masks, the crop rectangle,
    the owner map on a window on every seam (and the corners) against the oracle's argmax.  Then
    the multiband mosaic on twelve of those seam windows: ``_check_multiband_windows``."""
    from pano360_amd import engine, synth
    cfg = synth.CONFIGS["cfg3"]
    n, w, h, levels = cfg["n"], cfg["width"], cfg["height"], cfg["n_levels"]
    rots, intrs = synth.make_cameras(n, w, h, sweep_deg=cfg["sweep_deg"], jitter=0.01, seed=3)
    shapes = [(h, w)] * n
    plan = eng.upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9))
    H, W = plan.shape
    owner_t, valid_t = eng.ownership_cameras(plan)
    owner, valid = owner_t.cpu().numpy(), valid_t.cpu().numpy().astype(bool)
    oplan = oracle.Plan(shapes, rots, intrs, True, 10 ** 9)
    assert oplan.shape == plan.shape and oplan.rects == plan.rects
    # the rig really is tilted: the patches' top rows differ by tens of pixels
    tops = [r[0] for r in plan.rects]
    assert max(tops) - min(tops) >= 20 or max(r[1] for r in plan.rects) - min(r[1] for r in plan.rects) >= 20
    ref_valid = np.zeros(plan.shape, bool)
    for proj, rect in zip(oplan.projs, oplan.rects):
        _, _, mask = oracle.inverse_map(proj, oplan, rect, (h, w))
        ref_valid[rect[0]:rect[1], rect[2]:rect[3]] |= ~mask
    assert np.array_equal(valid, ref_valid)
    assert eng.crop_rect(valid_t) == oracle.crop_rect(ref_valid)
    S = 288
    rng = np.random.default_rng(20260)
    windows, pairs, seam_rows = _seam_windows(owner, S, rng)
    assert len(windows) >= 31, len(windows)
    corners = [(0, S, 0, S), (0, S, W - S, W), (H - S, H, 0, S), (H - S, H, W - S, W)]
    host = [synth.make_frame(i, w, h, "A") for i in range(2)]
    imgs = [host[i % 2] for i in range(n)]
    cache, owners_seen = {}, set()
    for win in windows + corners:
        wy0, wy1, wx0, wx1 = win
        _, patches, shape = _oracle_window(oracle, imgs, cache, rots, intrs, True, win)
        assert np.array_equal(owner[wy0:wy1, wx0:wx1], oracle.ownership(patches, shape)), win
        assert np.array_equal(valid[wy0:wy1, wx0:wx1], oracle.valid(patches, shape)), win
        owners_seen.update(np.unique(owner[wy0:wy1, wx0:wx1]).tolist())
    assert len(owners_seen - {-1}) == n              # every camera owns pixels in some window
    del cache, host, imgs
    pick = [windows[i] for i in sorted(rng.choice(len(windows), size=12, replace=False))]
    # (+ both ends of the sweep at mid height: the corners of a tilted rig may hold no pixel at all)
    ends = [(H // 2 - S // 2, H // 2 + S // 2, 0, S), (H // 2 - S // 2, H // 2 + S // 2, W - S, W)]
    rows = _check_multiband_windows(eng, oracle, "cfg3 jittered", rots, intrs, w, h, levels, plan,
                                    pick + ends)
    print(f"cfg3 jittered: owner map bit-exact on {len(windows)} seam windows + 4 corners; multiband on "
          f"12 seam windows, {sum(rows[:12])} of {seam_rows} seam rows = "
          f"{100.0 * sum(rows[:12]) / seam_rows:.2f} % of the seam length")


@pytest.mark.parametrize("world", [8, 5])
def test_cfg3_full_size_strips_equal_whole(eng, world):
    """BASELINE config 3 at full size cut into column strips (the N-GPU bench path, every
    rank emulated on this GPU): each strip equals the same columns of the whole mosaic bit
    for bit.  At world 8 a strip has 27-39 blur items, which the sort kernel cuts into 3-5
    vertical segments each - the whole mosaic runs unsegmented."""
    import torch
    from pano360_amd import dist as pdist
    from pano360_amd import engine, synth
    cfg = synth.CONFIGS["cfg3"]
    n, w, h, levels = cfg["n"], cfg["width"], cfg["height"], cfg["n_levels"]
    rots, intrs = synth.make_cameras(n, w, h, sweep_deg=cfg.get("sweep_deg"),
                                     step_deg=cfg.get("step_deg"))
    shapes = [(h, w)] * n
    base = eng.upload_frames([synth.make_frame(i, w, h, "A") for i in range(4)])
    frames = [base[i % 4] for i in range(n)]
    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    whole, _, _, _ = eng.stitch(frames, plan, "multiband", levels)
    for rank in range(world):
        st = pdist.ShardedStitcher(eng, shapes, rots, intrs, levels, rank, world, exchange=None)
        mine = eng.upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9,
                                           table_cols=st.table_cols))
        strip, _, _, _ = eng.multiband_fused([frames[i] for i in st.my_frames], mine, levels,
                                             frame_ids=st.my_frames, strip=st.strip)
        c0, c1 = st.strip
        assert torch.equal(strip[:, c0:c1], whole[:, c0:c1]), (world, rank)
        if world == 8 and rank in (0, 4):        # the option: segments off, same bits
            from pano360_amd import _lib
            eng.set_option(_lib.OPT_BLUR_SEGMENTS, 0)
            try:
                plain, _, _, _ = eng.multiband_fused([frames[i] for i in st.my_frames], mine,
                                                     levels, frame_ids=st.my_frames, strip=st.strip)
            finally:
                eng.set_option(_lib.OPT_BLUR_SEGMENTS, 1)
            assert torch.equal(plain[:, c0:c1], whole[:, c0:c1]), (world, rank)
            for seg_len in (12, 48):                 # ... and other cuts than the estimate's
                eng.set_option(_lib.OPT_BLUR_SEG_LEN, seg_len)
                try:
                    forced, _, _, _ = eng.multiband_fused([frames[i] for i in st.my_frames], mine,
                                                          levels, frame_ids=st.my_frames, strip=st.strip)
                finally:
                    eng.set_option(_lib.OPT_BLUR_SEG_LEN, 0)
                assert torch.equal(forced[:, c0:c1], whole[:, c0:c1]), (world, rank, seg_len)
    if world == 8:
        # trusted layouts at full size (two streams inside the stitch, the bench's plan-cached
        # figures and the strips' default): the whole mosaic and a strip, three stitches each
        # through an engine that does not wait, with other pixels in the last one
        fast = engine.Engine(eng.device).trust_layouts(True)
        st = pdist.ShardedStitcher(eng, shapes, rots, intrs, levels, 4, world, exchange=None)
        other = [base[(i + 1) % 4] for i in range(n)]
        for k in range(3):
            plan_t = fast.cached_plan(shapes, rots, intrs, True, 10 ** 9)
            use = other if k == 2 else frames
            got = fast.multiband_fused(use, plan_t, levels)[0]
            want = eng.stitch(use, plan, "multiband", levels)[0] if k == 2 else whole
            assert torch.equal(got, want), k
        fast.verify_trusted()
        c0, c1 = st.strip
        for k in range(3):
            plan_s = fast.cached_plan(shapes, rots, intrs, True, 10 ** 9, st.table_cols)
            got = fast.multiband_fused([frames[i] for i in st.my_frames], plan_s, levels,
                                       frame_ids=st.my_frames, strip=st.strip)[0]
            assert torch.equal(got[:, c0:c1], whole[:, c0:c1]), k
        fast.verify_trusted()
        assert fast.stitch_counts()[0] >= 4 and fast.stitch_counts()[1] == 0
        # ... and with the geometry kept (Engine.keep_geometry): the owner map, masks, records and
        # work list of the first stitch serve the others; new pixels in the second and fourth
        del fast
        keeping = engine.Engine(eng.device).trust_layouts(True, keep_geometry=True)
        want_other = eng.stitch(other, plan, "multiband", levels)[0]
        kept = 0
        for k in range(4):
            plan_t = keeping.cached_plan(shapes, rots, intrs, True, 10 ** 9)
            got = keeping.multiband_fused(other if k & 1 else frames, plan_t, levels)[0]
            kept += keeping.last_kept_geometry
            assert torch.equal(got, want_other if k & 1 else whole), k
        keeping.verify_trusted()
        assert kept == 3
        for k in range(3):
            plan_s = keeping.cached_plan(shapes, rots, intrs, True, 10 ** 9, st.table_cols)
            got = keeping.multiband_fused([(other if k & 1 else frames)[i] for i in st.my_frames], plan_s,
                                          levels, frame_ids=st.my_frames, strip=st.strip)[0]
            kept += keeping.last_kept_geometry
            assert torch.equal(got[:, c0:c1], (want_other if k & 1 else whole)[:, c0:c1]), k
        keeping.verify_trusted()
        assert kept == 5


def test_cfg4_keypoints_of_a_4k_frame_against_a_windowed_oracle(eng):
    """BASELINE config 4 at full size: keypoints and descriptors of a 3840 x 2160 frame.  No
    CPU oracle finishes a 4K frame (135 k keypoints through NumPy loops), but SIFT is local: a
    keypoint of the first octaves whose whole support - the blur chain below its layer, its
    orientation and descriptor windows - stays inside a 192 x 192 crop comes out of the crop
    exactly as out of the frame (the crop starts on a multiple of 64, so the 2x base and the
    nearest-neighbour halvings pick the same samples).  The oracle runs on two crops; inside
    their cores the frame's keypoints must be the crops' (count within 3 %, 97 % matched in
    position, size and angle, descriptors of the matched within the small-size bound)."""
    import sift_oracle
    import sift_pyramid as sp
    from pano360_amd import features, synth
    from test_gpu_parity import _match_keypoints
    tile = synth.make_frame(2, 960, 540, "B")
    img = np.ascontiguousarray(np.tile(tile, (4, 4, 1)))
    assert img.shape == (2160, 3840, 3)
    frame = eng.upload_frames([img])[0]
    got, desc = features.sift_detect_device(frame)
    desc = desc.cpu().numpy()
    assert len(got) > 20000
    size, margin = 192, 48
    total_want = total_got = 0
    for y0, x0 in ((1024, 2048), (448, 3584)):
        crop = img[y0:y0 + size, x0:x0 + size]
        g_or, d_or = sp.sift_pyramid(crop)
        want_k, want_d = sift_oracle.detect_and_compute(g_or, d_or)
        want = np.zeros(len(want_k), features.KP_DTYPE)
        for i, k in enumerate(want_k):
            want[i] = (k["x"], k["y"], k["size"], k["angle"], k["response"], k["octave"], k["r"], k["c"])

        def core(k, ox, oy):
            low = ((k["octave"] & 255) == 255) | ((k["octave"] & 255) == 0)      # octaves -1 and 0
            return (low & (k["x"] - ox >= margin) & (k["x"] - ox < size - margin) &
                    (k["y"] - oy >= margin) & (k["y"] - oy < size - margin))
        keep_w = core(want, 0, 0)
        want, want_d = want[keep_w], want_d[keep_w]
        keep_g = core(got, x0, y0)
        sub, sub_d = got[keep_g].copy(), desc[keep_g]
        sub["x"] -= x0
        sub["y"] -= y0
        assert len(want) > 40
        pairs = _match_keypoints(sub, want)
        assert len(pairs) >= 0.97 * len(want), (len(pairs), len(want), len(sub))
        gi, wi = np.array(pairs).T
        # octave and layer; the third byte is the refined offset inside the layer, which moves
        # with the scale space's float tolerance
        assert np.array_equal(sub["octave"][gi] & 0xffff, want["octave"][wi] & 0xffff)
        assert np.abs(sub_d[gi] - want_d[wi]).mean() < 0.5
        total_want += len(want)
        total_got += len(sub)
    assert abs(total_got - total_want) <= 0.03 * total_want + 2, (total_got, total_want)


@pytest.mark.parametrize("n", [0, 1, 2, 1000, 70001])
def test_sift_sort_unique_on_device_equals_the_host_lexsort(eng, n):
    """``pano_sift_sort_unique`` against ``features.sift_sort_unique`` (np.lexsort + duplicate
    removal, KeyPointsFilter::removeDuplicatedSorted) followed by the first-octave adjustment:
    random keypoints with many ties in every key, exact duplicates, -0.0 and negative values."""
    import ctypes as C
    import torch
    from pano360_amd import _lib, features
    from pano360_amd.engine import _ptr
    rng = np.random.default_rng(n)
    kp = np.zeros(n, features.KP_DTYPE)
    if n:
        kp["x"] = rng.integers(0, 40, n).astype(np.float32) * 0.5 - 3.0
        kp["y"] = rng.integers(0, 30, n).astype(np.float32) * 0.25
        kp["size"] = rng.choice(np.float32([1.5, 2.0, 3.25, 7.0]), n)
        kp["angle"] = rng.choice(np.float32([0.0, -0.0, 10.5, 359.9, 180.0]), n)
        kp["response"] = rng.choice(np.float32([0.01, 0.02, 0.5]), n)
        kp["octave"] = rng.choice(np.int32([0, 1, 255, (2 << 8) | 3, (1 << 16) | (1 << 8) | 2]), n)
        kp["r"], kp["c"] = rng.integers(0, 99, n), rng.integers(0, 99, n)
    want = features.sift_sort_unique(kp.copy())
    want["octave"] = (want["octave"] & ~255) | ((want["octave"] + features.SIFT_FIRST_OCTAVE) & 255)
    for key in ("x", "y", "size"):
        want[key] = want[key] * np.float32(0.5)
    dev = torch.from_numpy(kp.view(np.uint8).reshape(-1).copy()).to(eng.device)
    out = torch.empty(max(n, 1) * 32, dtype=torch.uint8, device=eng.device)
    work = torch.empty(int(eng.lib.pano_sift_sort_work_bytes(n)), dtype=torch.uint8,
                       device=eng.device)
    count = torch.full((1,), -1, dtype=torch.int32, device=eng.device)
    _lib.check(eng.lib.pano_sift_sort_unique(eng.ctx(), _ptr(dev), n, None,
                                             features.SIFT_FIRST_OCTAVE, _ptr(work), _ptr(out),
                                             _ptr(count)), "pano_sift_sort_unique")
    m = int(count.item())
    assert m == len(want)
    got = out[:m * 32].cpu().numpy().view(features.KP_DTYPE)
    # ties in every sort key and in (x, y, size, angle) leave r / c of a kept keypoint to the
    # stable order of equal records; all six sort keys must agree exactly
    for key in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(got[key], want[key]), key
    if n > 2:
        # the count on the device, the capacity on the host: the first 2 n / 3 records only
        part = 2 * n // 3
        n_dev = torch.tensor([part], dtype=torch.int32, device=eng.device)
        _lib.check(eng.lib.pano_sift_sort_unique(eng.ctx(), _ptr(dev), n, _ptr(n_dev),
                                                 features.SIFT_FIRST_OCTAVE, _ptr(work), _ptr(out),
                                                 _ptr(count)), "pano_sift_sort_unique")
        sub = features.sift_sort_unique(kp[:part].copy())
        assert int(count.item()) == len(sub)
        got_sub = out[:len(sub) * 32].cpu().numpy().view(features.KP_DTYPE)
        assert np.array_equal(got_sub["angle"], sub["angle"])
        assert np.array_equal(got_sub["y"], sub["y"] * np.float32(0.5))


def _analytic_texture(theta, phi, ch):
    """A smooth panorama in closed form, one of three channels: values in [0.15, 0.85], features of
    0.1 - 0.3 rad (hundreds of pixels at 4K)."""
    import torch
    a, b, c, d, p = [(41.0, 17.0, 23.0, 31.0, 0.3), (29.0, 23.0, 37.0, 19.0, 1.1),
                     (35.0, 13.0, 27.0, 43.0, 2.0)][ch]
    return 0.5 + 0.2 * torch.sin(a * theta + p) * torch.cos(b * phi) + 0.15 * torch.cos(c * theta - d * phi)


@pytest.mark.parametrize("name,jitter", [("cfg3", 0.0), ("cfg3", 0.01), ("cfg5", 0.0)])
def test_analytic_panorama_is_reproduced_everywhere(eng, name, jitter):
    """A known answer for the WHOLE mosaic of config 3 (all 31 seams from top to bottom; pure-yaw and
    jittered rig) and of config 5 (120 x 8K, the closed sweep: T is 2 pi-periodic): the frames are
    renderings of one analytic panorama T(theta, phi) - pixel (u, v) of camera i looks along
    R_i^T K^-1 (u - w/2, v - h/2, 1), the reference's conventions (bundle_adj.py:28-29,
    stitcher.py:119, 310) - so every patch agrees with every other on their overlaps and the
    multiband mosaic must be T itself at (theta_x, phi_y) = (x, y) * resolution + low
    (stitcher.py:301-302), whatever the ownership, the seams and the levels do: within two uint8
    levels (the frames' own quantisation, one truncation) on every valid pixel farther than the
    largest blur radius from the border of the covered area (there the blend mixes in
    BORDER_REFLECT content by design).  No oracle, no restatement: geometry and blend against
    ground truth, pure-yaw and jittered rig."""
    import torch
    from pano360_amd import engine, synth
    cfg = synth.CONFIGS[name]
    n, w, h, levels = cfg["n"], cfg["width"], cfg["height"], cfg["n_levels"]
    rots, intrs = synth.make_cameras(n, w, h, sweep_deg=cfg.get("sweep_deg"), step_deg=cfg.get("step_deg"),
                                     jitter=jitter, seed=9)
    dev = eng.device
    vv, uu = torch.meshgrid(torch.arange(h, dtype=torch.float64, device=dev),
                            torch.arange(w, dtype=torch.float64, device=dev), indexing="ij")
    pix = torch.stack([uu - w / 2, vv - h / 2, torch.ones_like(uu)], dim=0).reshape(3, -1)
    frames = []
    for i in range(n):
        hom = torch.from_numpy(rots[i].T.dot(np.linalg.inv(intrs[i]))).to(dev)        # pixel -> ray
        ray = hom @ pix
        theta = torch.atan2(ray[0], ray[2])
        phi = torch.atan2(ray[1], torch.sqrt(ray[0] ** 2 + ray[2] ** 2))
        img = torch.stack([_analytic_texture(theta, phi, ch) for ch in range(3)], dim=-1)
        frames.append(torch.round(img * 255.0).clamp(0, 255).to(torch.uint8).reshape(h, w, 3).contiguous())
        del ray, theta, phi, img
    plan = engine.Plan([(h, w)] * n, rots, intrs, True, 10 ** 9)
    mosaic, _, valid, _ = eng.stitch(frames, plan, "multiband", levels)
    H, W = plan.shape
    theta = torch.arange(W, dtype=torch.float64, device=dev) * plan.resolution[0] + plan.low[0]
    phi = torch.arange(H, dtype=torch.float64, device=dev) * plan.resolution[1] + plan.low[1]
    want = [(_analytic_texture(theta[None, :], phi[:, None], ch) * 255.0).float() for ch in range(3)]
    # valid and farther than R (+ the multiband padding) from anything invalid: erosion by max-pooling
    R = max(engine.gaussian_ksize(s) // 2 for s in engine.level_sigmas(levels)) + 12
    bad = (valid == 0).float()[None, None]
    near_bad = torch.nn.functional.max_pool2d(bad, 2 * R + 1, stride=1, padding=R)[0, 0] > 0
    core = ~near_bad
    # (the mosaic's own outer border counts as invalid beyond it)
    core[:R] = core[-R:] = False
    core[:, :R] = core[:, -R:] = False
    share = core.float().mean().item()
    assert share > 0.55, share
    del bad, near_bad

    def worst(img):                                      # (channel by channel: 228 MP at config 5)
        mx, total = 0.0, 0.0
        for ch in range(3):
            d = (img[..., ch].float() - want[ch]).abs()[core]
            mx, total = max(mx, d.max().item()), total + d.sum().item()
        return mx, total / (3.0 * core.sum().item())
    mx, mean = worst(mosaic)
    print(f"analytic panorama, {name}, jitter {jitter}: {int(core.sum().item()) / 1e6:.1f} MP compared "
          f"({100 * share:.0f} % of the mosaic), max |mosaic - T| {mx:.2f} levels, mean {mean:.3f}")
    assert mx <= 2.0 and mean <= 0.8
    # ... and the same through the linear and the paste blenders (unpadded plan)
    plan_u = engine.Plan([(h, w)] * n, rots, intrs, False, 10 ** 9)
    assert plan_u.shape == plan.shape
    for blend in ("linear", "none"):
        assert worst(eng.stitch(frames, plan_u, blend)[0])[0] <= 2.0, blend
