"""GPU: the HIP path (through the C ABI) against the oracle and the committed
golden vectors.  Integer/boolean results bit-exact; float32 planes that use the
reference's operation order bit-exact; the blurred multiband mosaic within the
stated tolerances (1e-4 relative L2 on the float mosaic, 1 LSB on uint8)."""
import numpy as np
import pytest

from conftest import SCENES, load_golden, n_patches, scene_inputs

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4          # BASELINE.json north_star: mosaic within 1e-4 rel-err


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def planes_to_rgba(dp):
    return dp.planes[:, :, :dp.w].permute(1, 2, 0).contiguous().cpu().numpy()


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))


# ------------------------------------------------------------------ warp
@pytest.mark.parametrize("name", SCENES)
def test_warp_maps_masks_planes_bit_exact(eng, name):
    from pano360_amd import engine
    g = load_golden(name)
    imgs, rots, intrs, mr = scene_inputs(g)
    plan = eng.upload_plan(engine.Plan([im.shape[:2] for im in imgs], rots, intrs, True, mr))
    patches, maps = eng.warp_all(eng.upload_frames(imgs), plan, want_maps=True)
    for i, (dp, (mx, my)) in enumerate(zip(patches, maps)):
        assert np.array_equal(bits(mx.cpu().numpy()), bits(g[f"mb_mapx_{i}"])), i
        assert np.array_equal(bits(my.cpu().numpy()), bits(g[f"mb_mapy_{i}"])), i
        assert np.array_equal(dp.mask.cpu().numpy().astype(bool), g[f"mb_mask_{i}"]), i
        if f"mb_warped_{i}" in g:
            assert np.array_equal(bits(planes_to_rgba(dp)), bits(g[f"mb_warped_{i}"])), i


def test_add_weights_bit_exact(eng, oracle):
    g = load_golden("scene_small_noise")
    img = g["imgs"][0]
    got = eng.add_weights(eng.upload_frames([img])[0]).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.add_weights(img)))
    assert np.array_equal(bits(got[..., 3]), bits(g["alpha0"]))


def test_warp_behind_camera_and_far_outside(eng, oracle):
    """A frame warped over a mosaic that wraps past +-90 degrees: z < 0 pixels,
    huge and non-finite coordinates all follow the cvRound / int16-saturation /
    REFLECT rules of the oracle."""
    from pano360_amd import engine, synth
    imgs, rots, intrs = synth.make_scene(3, 64, 48, step_deg=110.0, seed=3, kind="A")
    shapes = [im.shape[:2] for im in imgs]
    plan = engine.Plan(shapes, rots, intrs, True, 400)
    oplan = oracle.Plan(shapes, rots, intrs, True, 400)
    # stretch every patch to the full mosaic so each one sees the back hemisphere
    full = (0, plan.shape[0], 0, plan.shape[1])
    plan.rects = [full] * 3
    eng.upload_plan(plan)
    patches, maps = eng.warp_all(eng.upload_frames(imgs), plan, want_maps=True)
    behind = 0
    for i, (dp, (mx, my)) in enumerate(zip(patches, maps)):
        omx, omy, omask = oracle.inverse_map(oplan.projs[i], oplan, full, shapes[i])
        assert np.array_equal(bits(mx.cpu().numpy()), bits(omx))
        assert np.array_equal(bits(my.cpu().numpy()), bits(omy))
        assert np.array_equal(dp.mask.cpu().numpy().astype(bool), omask)
        want = oracle.remap(oracle.add_weights(imgs[i]), omx, omy)
        want[..., 3] *= ~omask
        assert np.array_equal(bits(planes_to_rgba(dp)), bits(want))
        behind += int((np.abs(omx) > 40000).sum())
    assert behind > 0        # the case really was exercised


# --------------------------------------------------------------- blenders
def _bl_patches(g):
    out = []
    for i in range(int(g["bl_n"])):
        y0, y1, x0, x1 = g[f"bl_irange_{i}"]
        out.append((g[f"bl_warped_{i}"].copy(), g[f"bl_mask_{i}"].copy(),
                    np.s_[int(y0):int(y1), int(x0):int(x1)]))
    return out


def test_stage_api_blenders(eng, oracle):
    """The reference's blender protocol on host patches (stitcher.py:160-248)."""
    from pano360_amd import stitcher
    g = load_golden("pure")
    shape = tuple(int(v) for v in g["bl_shape"])
    assert np.array_equal(stitcher.no_blend(_bl_patches(g), shape), g["bl_none"])
    assert np.array_equal(stitcher.linear_blend(_bl_patches(g), shape), g["bl_linear"])
    assert np.array_equal(stitcher._valid(_bl_patches(g), shape), g["bl_valid"])
    for lv, key in ((5, "bl_mb5"), (3, "bl_mb3")):
        patches = _bl_patches(g)
        got = stitcher.multiband_blend(patches, shape, lv)
        assert np.abs(got.astype(int) - g[key].astype(int)).max() <= 1
        # in-place side effect of the reference: alpha becomes the sharp mask
        own = oracle.ownership(_bl_patches(g), shape)
        for idx, (warped, _, ir) in enumerate(patches):
            assert np.array_equal(warped[..., 3], (own[ir] == idx).astype(np.float32))
    # single level: no blur at all, must be exact
    want1 = oracle.multiband_blend(_bl_patches(g), shape, 1)
    assert np.array_equal(stitcher.multiband_blend(_bl_patches(g), shape, 1), want1)



def test_stitch_exact_mode(eng):
    """``stitcher.EXACT`` (env PANO_EXACT=1): the drop-in ``stitch`` takes the full band sum on
    every pixel and the float32 vector-ALU blur.  Same bound against the reference's mosaic
    (one level; a handful of values sit on an integer boundary either way), and the same
    result as the engine asked for those two things directly."""
    import bundle_adj
    from pano360_amd import engine, stitcher
    g = load_golden("scene_small_noise")
    imgs, rots, intrs, mr = scene_inputs(g)
    saved = stitcher.MAX_RESOLUTION, stitcher.EXACT
    stitcher.MAX_RESOLUTION = mr
    try:
        def run():
            regs = [bundle_adj.Image(im.copy(), r.copy(), k.copy())
                    for im, r, k in zip(imgs, rots, intrs)]
            return stitcher.stitch(regs, stitcher.multiband_blend)
        stitcher.EXACT = False
        fast = run()
        stitcher.EXACT = True
        exact = run()
    finally:
        stitcher.MAX_RESOLUTION, stitcher.EXACT = saved
    ref = g["mb5_mosaic"].astype(int)
    assert np.abs(exact.astype(int) - ref).max() <= 1 and np.abs(fast.astype(int) - ref).max() <= 1
    valu = engine.Engine(blur="valu")
    plan = engine.Plan([im.shape[:2] for im in imgs], rots, intrs, True, mr)
    direct = valu.stitch(valu.upload_frames(imgs), plan, "multiband", 5, shortcut=False)[0]
    assert np.array_equal(exact, direct.cpu().numpy())


@pytest.mark.parametrize("name", SCENES)
def test_stitch_entry_point_against_reference_mosaics(eng, name):
    """stitch(regions, blender, crop) exactly as a reference caller writes it."""
    import bundle_adj
    from pano360_amd import stitcher
    g = load_golden(name)
    imgs, rots, intrs, mr = scene_inputs(g)
    saved = stitcher.MAX_RESOLUTION
    stitcher.MAX_RESOLUTION = mr
    try:
        def regions():
            return [bundle_adj.Image(im.copy(), r.copy(), k.copy())
                    for im, r, k in zip(imgs, rots, intrs)]
        regs = regions()
        assert np.array_equal(stitcher.stitch(regs, stitcher.no_blend), g["none_mosaic"])
        # side effects of the reference entry point (stitcher.py:277-278)
        assert regs[0].img.dtype == np.float32 and regs[0].img.shape[2] == 4
        assert np.array_equal(bits(regs[0].img[..., 3]), bits(g["alpha0"]))
        assert np.array_equal(regs[1].range[0], g["range_min"][1])
        assert np.array_equal(stitcher.stitch(regions(), stitcher.linear_blend),
                              g["linear_mosaic"])
        assert np.array_equal(stitcher.stitch(regions(), stitcher.linear_blend, crop=True),
                              g["lin_cropped"])
        got = stitcher.stitch(regions(), stitcher.multiband_blend)
        assert got.shape == g["mb5_mosaic"].shape
        assert np.abs(got.astype(int) - g["mb5_mosaic"].astype(int)).max() <= 1
        if "mb6_mosaic" in g:
            keep = stitcher.multiband_blend.__defaults__
            stitcher.multiband_blend.__defaults__ = (6,)
            try:
                got = stitcher.stitch(regions(), stitcher.multiband_blend)
            finally:
                stitcher.multiband_blend.__defaults__ = keep
            assert np.abs(got.astype(int) - g["mb6_mosaic"].astype(int)).max() <= 1
        # a user-supplied blender gets host patches in the reference's format
        seen = {}

        def spy(patches, shape):
            seen["n"], seen["shape"] = len(patches), shape
            w, m, ir = patches[0]
            assert w.dtype == np.float32 and w.shape[2] == 4 and m.dtype == bool
            assert w.shape[:2] == m.shape == (ir[0].stop - ir[0].start, ir[1].stop - ir[1].start)
            return np.zeros(shape + (3,), np.uint8)
        stitcher.stitch(regions(), spy)
        assert seen == {"n": len(imgs), "shape": tuple(g["lin_shape"])}
    finally:
        stitcher.MAX_RESOLUTION = saved


def test_equalize_gains_against_reference(eng, oracle):
    """stitch(..., equalize=True) (stitcher.py:36-66, 280-281): overlap sizes bit
    exact, means / gains to the accuracy of the reference's float32 np.mean,
    mosaics within one level of the reference's."""
    import bundle_adj
    from pano360_amd import stitcher
    g = load_golden("scene_equalize")
    imgs, rots, intrs, _ = scene_inputs(g)
    frames = eng.upload_frames(imgs)
    overlaps, sizes, gains, luts = eng.equalize_gains(frames, rots, intrs)
    assert np.array_equal(sizes, g["sizes"])
    np.testing.assert_allclose(overlaps, g["overlaps"], rtol=2e-6, atol=0)
    np.testing.assert_allclose(gains, g["gains"], rtol=2e-5, atol=0)
    assert np.abs(luts.cpu().numpy()[0][imgs[0]] - g["eq_rgb_0"]).max() <= 2e-5
    # deterministic: fixed-order double sums
    again = eng.equalize_gains(frames, rots, intrs)
    assert np.array_equal(again[0], overlaps) and np.array_equal(again[2], gains)

    def regions():
        return [bundle_adj.Image(im.copy(), r.copy(), k.copy())
                for im, r, k in zip(imgs, rots, intrs)]
    for blender, key in ((stitcher.linear_blend, "lin"), (stitcher.multiband_blend, "mb5")):
        regs = regions()
        got = stitcher.stitch(regs, blender, equalize=True)
        assert got.shape == g[f"{key}_mosaic"].shape
        assert np.abs(got.astype(int) - g[f"{key}_mosaic"].astype(int)).max() <= 1, key
        # reg.img is the equalised float32 RGBA image (stitcher.py:66)
        assert np.abs(regs[0].img[..., :3] - g["eq_rgb_0"]).max() <= 2e-5
    # the blender protocol (host patches) sees equalised patches too
    via_stage = stitcher.stitch(regions(), lambda p, s: stitcher.linear_blend(p, s), equalize=True)
    assert np.abs(via_stage.astype(int) - g["lin_mosaic"].astype(int)).max() <= 1
    # equalize_gains(regions) on float32 RGBA regions, as the reference is called
    regs = regions()
    for reg in regs:
        reg.img = stitcher._add_weights(reg.img)
    found = stitcher.equalize_gains(regs)
    np.testing.assert_allclose(found, g["gains"], rtol=2e-5, atol=0)
    assert np.abs(regs[-1].img[..., :3] - g["eq_rgb_last"]).max() <= 2e-5


@pytest.mark.parametrize("case", ["sweep", "tilted", "perspective"])
def test_overlap_stats_match_oracle(eng, oracle, case):
    """pano_overlap_stats against the C oracle on larger frames: counts bit exact
    (tile culling and the fixed-point taps included), means to float32 accuracy."""
    from pano360_amd import synth
    n, w, h = 6, 333, 187
    imgs, rots, intrs = synth.make_scene(n, w, h, sweep_deg=150.0, jitter=0.02, seed=3, kind="B")
    if case == "tilted":
        rots = [synth.rotation_to_mat(np.array([0.3 * (-1) ** i, 0.25 * i - 0.6, 0.2 * i]))
                for i in range(n)]
    elif case == "perspective":
        intrs = [k * np.array([[1 + 0.15 * i, 1, 1], [1, 1 + 0.15 * i, 1], [1, 1, 1]])
                 for i, k in enumerate(intrs)]
    frames = eng.upload_frames(imgs)
    overlaps, sizes, gains, _ = eng.equalize_gains(frames, rots, intrs, chunk_bytes=1 << 16)
    rgbas = [oracle.add_weights(im) for im in imgs]
    want_o, want_s, want_g = oracle.equalize_gains(rgbas, rots, intrs)
    assert np.array_equal(sizes, want_s) and sizes.max() > 0
    if case == "sweep":
        assert (sizes == 0).sum() > n          # pairs without overlap are in the batch too
    np.testing.assert_allclose(overlaps, want_o, rtol=2e-6, atol=0)
    np.testing.assert_allclose(gains, want_g, rtol=2e-5, atol=0)


@pytest.mark.parametrize("name", SCENES)
def test_multiband_float_mosaic_and_valid(eng, oracle, name):
    from pano360_amd import engine
    g = load_golden(name)
    imgs, rots, intrs, mr = scene_inputs(g)
    plan = engine.Plan([im.shape[:2] for im in imgs], rots, intrs, True, mr)
    for lv in (5, 6, 2):
        mosaic, fl, valid, _ = eng.stitch(eng.upload_frames(imgs), plan, "multiband", lv,
                                          want_float=True)
        ref_u8, ref_f = oracle.stitch(imgs, rots, intrs, "multiband", lv, max_resolution=mr,
                                      return_float=True)
        assert rel_l2(fl.cpu().numpy(), ref_f) <= REL_TOL
        assert np.abs(mosaic.cpu().numpy().astype(int) - ref_u8.astype(int)).max() <= 1
        assert np.array_equal(valid.cpu().numpy().astype(bool), g["mb_valid"])


def test_ownership_bit_exact(eng, oracle):
    from pano360_amd import engine
    g = load_golden("scene_small_noise")
    imgs, rots, intrs, mr = scene_inputs(g)
    plan = eng.upload_plan(engine.Plan([im.shape[:2] for im in imgs], rots, intrs, True, mr))
    patches, _ = eng.warp_all(eng.upload_frames(imgs), plan)
    owner, valid = eng.ownership(engine.patch_table(patches, eng), plan.shape)
    _, ref_patches, _ = oracle.warp_all(imgs, rots, intrs, True, mr)
    assert np.array_equal(owner.cpu().numpy().astype(np.int32),
                          oracle.ownership(ref_patches, plan.shape))
    assert np.array_equal(valid.cpu().numpy().astype(bool), oracle.valid(ref_patches, plan.shape))


@pytest.mark.parametrize("name", SCENES)
def test_ownership_from_cameras_equals_ownership_from_planes(eng, name):
    """pano_ownership_cameras (no pixel data) against pano_ownership (warped
    alpha planes): integer maps, bit-exact."""
    import torch
    from pano360_amd import engine
    g = load_golden(name)
    imgs, rots, intrs, mr = scene_inputs(g)
    plan = eng.upload_plan(engine.Plan([im.shape[:2] for im in imgs], rots, intrs, True, mr))
    patches, _ = eng.warp_all(eng.upload_frames(imgs), plan)
    owner_b, valid_b = eng.ownership(engine.patch_table(patches, eng), plan.shape)
    from pano360_amd import _lib
    try:
        for opt in (3, 1):        # one level of bounds (round 4's kernel), two levels (the default)
            eng.set_option(_lib.OPT_OWN_PRUNE, opt)
            owner_a, valid_a = eng.ownership_cameras(plan)
            assert torch.equal(owner_a, owner_b) and torch.equal(valid_a, valid_b), opt
    finally:
        eng.set_option(_lib.OPT_OWN_PRUNE, 1)
    assert np.array_equal(valid_a.cpu().numpy().astype(bool), g["mb_valid"])
    # column strips (the multi-GPU split) tile the same map
    W = plan.shape[1]
    out = (torch.full_like(owner_a, -7), torch.full_like(valid_a, 9))
    for xs in ((0, W // 3), (W // 3, W // 3), (W // 3, W - 5), (W - 5, W)):
        eng.ownership_cameras(plan, strip=xs, out=out)
    assert torch.equal(out[0], owner_a) and torch.equal(out[1], valid_a)
    boxes = eng.owned_boxes(owner_a, plan.n)
    own = owner_a.cpu().numpy()
    for i in range(plan.n):
        ys, xs = np.nonzero(own == i)
        if len(ys):
            assert tuple(boxes[i]) == (ys.min(), ys.max(), xs.min(), xs.max())
        else:
            assert boxes[i][1] < boxes[i][0]


@pytest.mark.parametrize("seed", range(6))
def test_ownership_bounds_against_exhaustive_evaluation(eng, seed):
    """The interval bounds that prune cameras per 64 x 16 tile never change the
    owner / valid maps: random rotations (roll and pitch included), mixed focal
    lengths, native and capped resolutions, a camera looking away."""
    import torch
    from pano360_amd import bundle_adj, engine, synth
    rng = np.random.default_rng(100 + seed)
    n, w, h = 10, 320, 200
    rots = np.stack([bundle_adj.rotation_to_mat(
        [rng.normal(0, 0.2), rng.uniform(-1.1, 1.1), rng.normal(0, 0.35)]) for _ in range(n)])
    if seed == 0:
        rots[3] = bundle_adj.rotation_to_mat([0.0, np.pi * 0.97, 0.0])     # faces backwards
    intrs = np.stack([bundle_adj.intrinsics(synth.focal_for(w, rng.uniform(45, 100)),
                                            (rng.normal(0, 4), rng.normal(0, 4)))
                      for _ in range(n)]).astype(np.float64)
    cap = (10 ** 9, 700, 180)[seed % 3]
    plan = eng.upload_plan(engine.Plan([(h, w)] * n, rots, intrs, seed % 2 == 0, cap))
    from pano360_amd import _lib
    # option bits: 1 = prune by bounds, 2 = the one-level kernel of round 4.  Every camera at
    # every pixel (0, and 2 through the other kernel's evaluation loop) against both pruned forms
    got = {}
    try:
        for opt in (0, 2, 3, 1):
            eng.set_option(_lib.OPT_OWN_PRUNE, opt)
            got[opt] = tuple(v.clone() for v in eng.ownership_cameras(plan))
        torch.cuda.synchronize()
    finally:
        eng.set_option(_lib.OPT_OWN_PRUNE, 1)
    owner_all, valid_all = got[0]
    for opt in (1, 2, 3):
        assert torch.equal(got[opt][0], owner_all) and torch.equal(got[opt][1], valid_all), opt
    assert (owner_all >= 0).any() and (owner_all < 0).any()


@pytest.mark.parametrize("scene", ["cfg2_like", "closed_sweep", "strips"])
def test_stitch_with_the_layout_on_the_device_equals_the_host_layout(eng, scene):
    """pano_stitch_multiband lays the record table out with a kernel and queues the warp, blur
    and collapse behind it at once, sized by the previous stitch's layout (option
    PANO_OPT_STITCH_ASYNC); the host checks the layout's summary afterwards.  Same mosaic,
    valid mask and records as with the host layout: on repeated stitches (the device layout is
    used from the second on), with cameras that move a little between stitches (inside the
    bounds' slack), with bounds the layout exceeds (option value 2: the tail is queued again
    from the host layout) and on column strips."""
    import torch
    from pano360_amd import _lib, engine, synth
    if scene == "closed_sweep":
        n, w, h, kw = 24, 160, 96, dict(step_deg=15.0)
    else:
        n, w, h, kw = 8, 480, 270, dict(sweep_deg=140.0)
    imgs = [synth.make_frame(i, w, h, "A") for i in range(n)]
    frames = eng.upload_frames(imgs)

    def run(rots, intrs, mode, strip=None):
        eng.set_option(_lib.OPT_STITCH_ASYNC, mode)
        try:
            plan = eng.upload_plan(engine.Plan([(h, w)] * n, rots, intrs, True, 10 ** 9))
            mosaic, _, valid, patches = eng.multiband_fused(frames, plan, 5, strip=strip)
            torch.cuda.synchronize()
            return mosaic.clone(), valid.clone(), patches.table.host.copy()
        finally:
            eng.set_option(_lib.OPT_STITCH_ASYNC, 0)

    rots, intrs = synth.make_cameras(n, w, h, jitter=0.004, seed=1, **kw)
    W = engine.Plan([(h, w)] * n, rots, intrs, True, 10 ** 9).shape[1]
    strip = (W // 4 + 3, W // 2 + 1) if scene == "strips" else None
    want = run(rots, intrs, 0, strip)
    before = eng.stitch_counts()
    keys = ("index", "vy0", "vx0", "vh", "vw", "ay0", "ax0", "ah", "aw", "vpitch", "apitch", "tiles_off")
    for mode in (1, 1, 1, 2, 1):
        got = run(rots, intrs, mode, strip)
        c0, c1 = strip if strip else (0, W)
        assert torch.equal(got[0][:, c0:c1], want[0][:, c0:c1]), mode
        assert torch.equal(got[1][:, c0:c1], want[1][:, c0:c1]), mode
        assert len(got[2]) == len(want[2])
        for key in keys:
            assert np.array_equal(got[2][key], want[2][key]), (mode, key)
    after = eng.stitch_counts()
    assert after[0] - before[0] == 4 and after[1] - before[1] == 1, (before, after)
    if scene == "cfg2_like":
        # cameras that move a little: the same mosaic shape is not guaranteed, so only stitches
        # that keep it run through the device layout; every result is checked against mode 0
        rng = np.random.default_rng(0)
        for step in range(4):
            r2 = rots.copy()
            r2[rng.integers(n)] = synth.make_cameras(n, w, h, jitter=0.004, seed=10 + step, **kw)[0][0]
            a, b = run(r2, intrs, 1, strip), run(r2, intrs, 0, strip)
            assert a[0].shape == b[0].shape and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("strip_case", [False, True])
def test_trusted_stitches_equal_waiting_ones_and_a_broken_promise_is_caught(strip_case):
    """Engine.trust_layouts: a stitch that repeats the previous one's Plan object (a PlanMemo hit)
    is queued with that stitch's verified layout and the native call does not wait
    (pano_stitch_args.trust_layout).  Every kernel still runs, so with NEW pixels in the frames
    the mosaics equal those of an engine that waits, stitch for stitch; verify_trusted accepts
    them.  A Plan of other cameras smuggled in under the kept signature is caught by
    verify_trusted, and the engine then stitches correctly again."""
    import torch
    from pano360_amd import _lib, engine, synth
    n, w, h = 8, 480, 270
    rots, intrs = synth.make_cameras(n, w, h, sweep_deg=140.0, jitter=0.004, seed=1)
    shapes = [(h, w)] * n
    waiting, trusting = engine.Engine("cuda:0"), engine.Engine("cuda:0").trust_layouts(True)
    W = engine.Plan(shapes, rots, intrs, True, 10 ** 9).shape[1]
    strip = (W // 4 + 3, W // 2 + 1) if strip_case else None
    c0, c1 = strip if strip else (0, W)
    ids = list(range(n))
    took = []
    for k in range(5):
        frames = waiting.upload_frames([synth.make_frame(100 * k + i, w, h, "A") for i in range(n)])
        plan_w = waiting.cached_plan(shapes, rots, intrs, True, 10 ** 9)
        want, _, want_valid, _ = waiting.multiband_fused(frames, plan_w, 5, frame_ids=ids, strip=strip)
        plan_t = trusting.cached_plan(shapes, rots, intrs, True, 10 ** 9)
        got, _, got_valid, patches = trusting.multiband_fused(frames, plan_t, 5, frame_ids=ids, strip=strip)
        took.append(trusting._stitch_ws[next(iter(trusting._stitch_ws))]["args"].trust_layout)
        torch.cuda.synchronize()
        assert torch.equal(got[:, c0:c1], want[:, c0:c1]), k
        assert torch.equal(got_valid[:, c0:c1], want_valid[:, c0:c1]), k
        assert len(patches) > 0
        trusting.verify_trusted()
    # the first stitch laid out on the host, the second on the device and waited (it verifies the
    # device layout), from the third on nobody waits
    assert took[0] == 0 and took[-1] == 2 and took.count(2) >= 3, took
    # a broken promise: other cameras under the kept signature
    rots2, _ = synth.make_cameras(n, w, h, sweep_deg=100.0, jitter=0.004, seed=2)
    other = trusting.upload_plan(engine.Plan(shapes, rots2, intrs, True, 10 ** 9))
    if other.shape == plan_t.shape:
        sig, _, kept = trusting._trusted[:3]
        trusting._trusted = ((id(other),) + sig[1:], other, kept) + trusting._trusted[3:]
        trusting.multiband_fused(frames, other, 5, frame_ids=ids, strip=strip)
        if strip_case:
            # ... and a GOOD trusted stitch behind the bad one, before anybody verifies: the one
            # summary slot now holds the good stitch's layout, the bad one's lives on in the sticky
            # word its layout kernel set (a void mosaic must not pass because a later one is fine)
            sig, _, kept = trusting._trusted[:3]
            trusting._trusted = ((id(plan_t),) + sig[1:], plan_t, kept) + trusting._trusted[3:]
            trusting.multiband_fused(frames, plan_t, 5, frame_ids=ids, strip=strip)
            assert trusting._stitch_ws[next(iter(trusting._stitch_ws))]["args"].trust_layout == 2
        with pytest.raises(_lib.PanoError, match="verified"):
            trusting.verify_trusted()
    # ... after which the engine lays out from scratch and is right again
    got, _, got_valid, _ = trusting.multiband_fused(frames, plan_t, 5, frame_ids=ids, strip=strip)
    torch.cuda.synchronize()
    assert torch.equal(got[:, c0:c1], want[:, c0:c1]) and torch.equal(got_valid[:, c0:c1], want_valid[:, c0:c1])


@pytest.mark.gpu
@pytest.mark.parametrize("strip_case,equalised", [(False, False), (True, False), (False, True)])
def test_kept_geometry_stitches_equal_waiting_ones(strip_case, equalised):
    """Engine.trust_layouts(keep_geometry=True): a repeat of the verified stitch re-uses the owner
    map, valid mask, interior map, record table, tile flags and work list the previous stitch left
    on the device (pano_stitch_args.trust_layout = 3 -> 4) and queues the warp, the blur and the
    collapse alone.  With NEW pixels in the frames every mosaic equals the waiting engine's, float
    image included; the ownership kernel does not run in a kept stitch; any other call through the
    context voids the kept geometry (the next stitch recomputes it, then keeps again)."""
    import torch
    from pano360_amd import engine, synth
    n, w, h = 8, 480, 270
    rots, intrs = synth.make_cameras(n, w, h, sweep_deg=140.0, jitter=0.004, seed=1)
    shapes = [(h, w)] * n
    waiting = engine.Engine("cuda:0")
    keeping = engine.Engine("cuda:0").trust_layouts(True, keep_geometry=True)
    W = engine.Plan(shapes, rots, intrs, True, 10 ** 9).shape[1]
    strip = (W // 4 + 3, W // 2 + 1) if strip_case else None
    c0, c1 = strip if strip else (0, W)
    ids = list(range(n))
    took, ran_ownership = [], []
    for k in range(7):
        frames = waiting.upload_frames([synth.make_frame(100 * k + i, w, h, "A") for i in range(n)])
        plan_w = waiting.cached_plan(shapes, rots, intrs, True, 10 ** 9)
        luts = None
        if equalised:
            luts = torch.from_numpy(engine.gain_tables([0.7 + 0.08 * ((i + 3 * k) % 6) for i in range(n)])).to("cuda:0")
        want, want_f, want_valid, _ = waiting.multiband_fused(frames, plan_w, 5, frame_ids=ids, strip=strip,
                                                              want_float=True, luts=luts)
        plan_k = keeping.cached_plan(shapes, rots, intrs, True, 10 ** 9)
        if k == 4:      # a foreign call on the context: the kept geometry is void
            keeping.ownership_cameras(plan_k, cams=keeping.camera_table(plan_k, dict(zip(ids, frames))))
        if k == 2 and not strip_case:      # ... the crop only reads the valid mask: it is not
            assert keeping.crop_rect(want_valid) == waiting.crop_rect(want_valid)
        keeping.timing(True)
        got, got_f, got_valid, patches = keeping.multiband_fused(frames, plan_k, 5, frame_ids=ids, strip=strip,
                                                                 want_float=True, luts=luts)
        torch.cuda.synchronize()
        ran_ownership.append("ownership_cameras_kernel" in keeping.kernel_times())
        keeping.timing(False)
        took.append(keeping._stitch_ws[next(iter(keeping._stitch_ws))]["args"].trust_layout)
        assert took[-1] == (4 if keeping.last_kept_geometry else took[-1])
        assert torch.equal(got[:, c0:c1], want[:, c0:c1]), (k, took)
        assert torch.equal(got_f[:, c0:c1], want_f[:, c0:c1]), (k, took)
        assert torch.equal(got_valid[:, c0:c1], want_valid[:, c0:c1]), (k, took)
        assert len(patches) > 0
        keeping.verify_trusted()
    # the first stitch lays out on the host and leaves its geometry; repeats keep it; the foreign
    # call before stitch 4 voids it once
    assert took[0] == 0 and took[1] == 4 and took[3] == 4 and took[4] in (0, 2) and took[6] == 4, took
    assert ran_ownership == [t != 4 for t in took], (ran_ownership, took)


@pytest.mark.parametrize("seed", range(6))
def test_kept_geometry_on_random_rigs(seed):
    """Random rigs (camera count, frame size, sweep, jitter, pyramid levels, whole mosaic or a strip): four
    stitches with other pixels each through an engine that keeps the geometry equal the waiting
    engine's, and every one after the first keeps it."""
    import torch
    from pano360_amd import engine, synth
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(3, 13))
    w, h = int(rng.integers(20, 60)) * 8, int(rng.integers(12, 40)) * 8
    levels = int(rng.choice([2, 5, 6]))
    rots, intrs = synth.make_cameras(n, w, h, sweep_deg=float(rng.uniform(40.0, 200.0)),
                                     jitter=float(rng.uniform(0.0, 0.02)), seed=seed)
    shapes = [(h, w)] * n
    waiting = engine.Engine("cuda:0")
    keeping = engine.Engine("cuda:0").trust_layouts(True, keep_geometry=True)
    W = engine.Plan(shapes, rots, intrs, True, 10 ** 9).shape[1]
    strip = None
    if rng.random() < 0.5 and W > 300:
        c0 = int(rng.integers(0, W // 2))
        strip = (c0, int(rng.integers(c0 + 100, W + 1)))
    c0, c1 = strip if strip else (0, W)
    ids = list(range(n))
    kept = []
    for k in range(4):
        frames = waiting.upload_frames([synth.make_frame(50 * k + i, w, h, "AB"[k & 1]) for i in range(n)])
        want, _, want_valid, _ = waiting.multiband_fused(
            frames, waiting.cached_plan(shapes, rots, intrs, True, 10 ** 9), levels, frame_ids=ids, strip=strip)
        got, _, got_valid, _ = keeping.multiband_fused(
            frames, keeping.cached_plan(shapes, rots, intrs, True, 10 ** 9), levels, frame_ids=ids, strip=strip)
        torch.cuda.synchronize()
        kept.append(keeping.last_kept_geometry)
        assert torch.equal(got[:, c0:c1], want[:, c0:c1]), (seed, k, n, w, h, levels, strip)
        assert torch.equal(got_valid[:, c0:c1], want_valid[:, c0:c1]), (seed, k)
    keeping.verify_trusted()
    assert kept == [False, True, True, True], (seed, kept)


@pytest.mark.parametrize("case", ["sweep", "tilted", "dense", "crowd", "strip"])
def test_ownership_with_regions_equals_the_two_calls(eng, case):
    """pano_ownership_regions (boxes and column marks out of the ownership kernel) against
    pano_ownership_cameras followed by pano_owned_regions: same owner, valid, regions and
    marks - with and without pruning, on a strip, with more candidates per sub-tile than the
    kernel keeps boxes for in LDS (dense), and with more cameras than its list holds (crowd)."""
    import torch
    from pano360_amd import _lib, bundle_adj, engine, synth
    rng = np.random.default_rng(11)
    strip = None
    if case in ("sweep", "strip"):
        n, w, h = 8, 240, 136
        rots, intrs = synth.make_cameras(n, w, h, sweep_deg=150.0, jitter=0.01, seed=3)
    elif case == "tilted":
        n, w, h = 9, 160, 120
        rots = np.stack([bundle_adj.rotation_to_mat(
            [rng.normal(0, 0.15), 0.3 * (i - 4), rng.normal(0, 0.15)]) for i in range(n)])
        intrs = np.stack([bundle_adj.intrinsics(synth.focal_for(w))] * n).astype(np.float64)
    elif case == "dense":
        n, w, h = 40, 96, 64
        rots, intrs = synth.make_cameras(n, w, h, step_deg=2.0, jitter=0.003, seed=1)
    else:
        n, w, h = 300, 48, 32
        rots, intrs = synth.make_cameras(n, w, h, step_deg=0.2, jitter=0.002, seed=4)
    plan = eng.upload_plan(engine.Plan([(h, w)] * n, rots, intrs, True, 10 ** 9))
    W = plan.shape[1]
    if case == "strip":
        strip = (W // 3 + 5, 2 * W // 3 + 1)
    for prune in (1, 0, 3):
        eng.set_option(_lib.OPT_OWN_PRUNE, prune)
        try:
            owner_a, valid_a = eng.ownership_cameras(plan, strip=strip)
            wait = eng.owned_regions_async(owner_a, n, strip=strip, min_gap=7, max_spans=3)
            regions_a = wait.raw().copy()
            owner_b, valid_b, regions_b, marks_b = eng.ownership_regions(
                plan, strip=strip, min_gap=7, max_spans=3)
            torch.cuda.synchronize()
        finally:
            eng.set_option(_lib.OPT_OWN_PRUNE, 1)
        c0, c1 = strip if strip else (0, W)
        assert torch.equal(owner_a[:, c0:c1], owner_b[:, c0:c1])
        assert torch.equal(valid_a[:, c0:c1], valid_b[:, c0:c1])
        regions_b = regions_b.cpu().numpy()
        own = owner_b[:, c0:c1].cpu().numpy()
        for i in range(n):
            cnt = regions_a[i, 4]
            assert regions_b[i, 4] == cnt
            assert np.array_equal(regions_a[i, :5 + 2 * cnt], regions_b[i, :5 + 2 * cnt]), (case, i)
            cols = np.nonzero((own == i).any(axis=0))[0] + c0
            got = np.nonzero(marks_b[i].cpu().numpy())[0]
            assert np.array_equal(cols, got)
        assert (regions_b[:, 4] > 0).sum() >= min(n, 3)


@pytest.mark.parametrize("case", ["dense", "tilted", "wide", "identical"])
def test_ownership_pruning_is_exact(eng, case):
    """The camera-driven kernel samples alpha only where an upper bound says the
    camera can still win; whatever the geometry, the map must equal the
    exhaustive argmax over warped alpha planes (first index on ties)."""
    import torch
    from pano360_amd import bundle_adj, engine, synth
    rng = np.random.default_rng(5)
    if case == "dense":          # 40 cameras 2 deg apart: > 12 cover a pixel (slot overflow)
        n, w, h = 40, 96, 64
        rots, intrs = synth.make_cameras(n, w, h, step_deg=2.0, jitter=0.003, seed=1)
    elif case == "tilted":       # pitch / roll up to ~15 deg
        n, w, h = 9, 160, 120
        rots = np.stack([bundle_adj.rotation_to_mat(
            [rng.normal(0, 0.15), 0.3 * (i - 4), rng.normal(0, 0.15)]) for i in range(n)])
        intrs = np.stack([bundle_adj.intrinsics(synth.focal_for(w))] * n).astype(np.float64)
    elif case == "wide":         # 120 deg lenses, off-centre principal points
        n, w, h = 5, 200, 100
        rots, _ = synth.make_cameras(n, w, h, step_deg=35.0, jitter=0.02, seed=2)
        intrs = np.stack([bundle_adj.intrinsics(synth.focal_for(w, 120.0), (3.0 * i, -2.0))
                          for i in range(n)]).astype(np.float64)
    else:                        # the same camera four times: every pixel is a 4-way tie
        n, w, h = 4, 120, 80
        rots, intrs = synth.make_cameras(1, w, h, step_deg=0.0)
        rots, intrs = np.repeat(rots, n, 0), np.repeat(intrs, n, 0)
    imgs = [synth.make_frame(i, w, h, "A") for i in range(n)]
    plan = eng.upload_plan(engine.Plan([(h, w)] * n, rots, intrs, True, 10 ** 9))
    patches, _ = eng.warp_all(eng.upload_frames(imgs), plan)
    owner_b, valid_b = eng.ownership(engine.patch_table(patches, eng), plan.shape)
    from pano360_amd import _lib
    try:
        for opt in (3, 1):        # one level of bounds (round 4's kernel), two levels (the default)
            eng.set_option(_lib.OPT_OWN_PRUNE, opt)
            owner_a, valid_a = eng.ownership_cameras(plan)
            assert torch.equal(owner_a, owner_b) and torch.equal(valid_a, valid_b), opt
    finally:
        eng.set_option(_lib.OPT_OWN_PRUNE, 1)
    if case == "identical":
        assert int(owner_a.max()) == 0           # ties go to the first index
    if case == "dense":
        cover = sum((torch.zeros(plan.shape, device=eng.device)
                     .index_put_((torch.arange(r[0], r[1], device=eng.device)[:, None],
                                  torch.arange(r[2], r[3], device=eng.device)[None, :]),
                                 torch.ones((), device=eng.device))) for r in plan.rects)
        assert int(cover.max()) > 12             # the slot overflow path really ran


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("levels", [1, 2, 5, 6])
def test_fused_windows_equal_whole_patch_path(eng, name, levels):
    """Restricting warp/blur/gather to the windows near owned pixels changes
    nothing: same uint8 and same float mosaic, bit for bit."""
    import torch
    from pano360_amd import engine
    g = load_golden(name)
    imgs, rots, intrs, mr = scene_inputs(g)
    plan = engine.Plan([im.shape[:2] for im in imgs], rots, intrs, True, mr)
    frames = eng.upload_frames(imgs)
    m1, f1, v1, p1 = eng.stitch(frames, plan, "multiband", levels, want_float=True, fused=True,
                                shortcut=False)
    m2, f2, v2, _ = eng.stitch(frames, plan, "multiband", levels, want_float=True, fused=False)
    assert torch.equal(m1, m2) and torch.equal(v1, v2)
    assert torch.equal(f1.view(torch.int32), f2.view(torch.int32))


def test_warp_need_flags_change_nothing(eng):
    """The warp may skip the blocks of a window nobody reads (pano_blur_tiles' need flags):
    with the flags forced on and off the mosaics are the same bit for bit, also after the
    workspace was filled with other data."""
    import torch
    from pano360_amd import engine, synth
    for seed, (w, h, sweep) in enumerate([(640, 360, 50.0), (960, 200, 30.0)]):
        imgs, rots, intrs = synth.make_scene(4, w, h, sweep_deg=sweep, jitter=0.01, seed=50 + seed,
                                             kind="A")
        shapes = [im.shape[:2] for im in imgs]
        frames = eng.upload_frames(imgs)
        out = {}
        try:
            for mode in (True, False, True):
                eng.warp_need = mode
                eng.arena("planes", 1).fill_(float("nan"))
                m, f, _, _ = eng.stitch(frames, engine.Plan(shapes, rots, intrs, True, 10 ** 9),
                                        "multiband", 5, want_float=True)
                if mode in out:
                    assert torch.equal(out[mode][0], m)
                out[mode] = (m, f)
        finally:
            eng.warp_need = "auto"
        assert torch.equal(out[False][0], out[True][0])
        assert torch.equal(out[False][1].view(torch.int32), out[True][1].view(torch.int32))
        assert not torch.isnan(out[True][1]).any()


def test_fused_windows_on_a_wide_sweep(eng, oracle):
    """Frames much wider than the blur radius, so the windows really cut work
    (and the 64-column / 128-row tile seams fall inside them)."""
    import torch
    from pano360_amd import engine, synth
    imgs, rots, intrs = synth.make_scene(6, 640, 360, sweep_deg=50.0, jitter=0.01, seed=31,
                                         kind="B")
    plan = engine.Plan([im.shape[:2] for im in imgs], rots, intrs, True, 10 ** 9)
    frames = eng.upload_frames(imgs)
    m1, f1, v1, patches = eng.stitch(frames, plan, "multiband", 5, want_float=True,
                                     shortcut=False)
    m2, f2, _, _ = eng.stitch(frames, plan, "multiband", 5, want_float=True, fused=False)
    assert torch.equal(m1, m2) and torch.equal(f1.view(torch.int32), f2.view(torch.int32))
    warped = sum((p.window[1] - p.window[0]) * (p.window[3] - p.window[2]) for p in patches)
    assert warped < 0.8 * plan.patch_pixels          # the windows did skip work
    ref_u8, ref_f = oracle.stitch(imgs, rots, intrs, "multiband", 5, max_resolution=10 ** 9,
                                  return_float=True)
    assert rel_l2(f1.cpu().numpy(), ref_f) <= REL_TOL
    assert np.abs(m1.cpu().numpy().astype(int) - ref_u8.astype(int)).max() <= 1


@pytest.mark.parametrize("levels", [3, 7, 8])
def test_level_counts_against_oracle(eng, oracle, levels):
    """n_levels the goldens do not hold, 8 being the ABI's limit: two groups of levels per
    workgroup and the widest Toeplitz tables (97- and 105-tap levels, C = 4)."""
    from pano360_amd import engine, synth
    imgs, rots, intrs = synth.make_scene(4, 400, 240, sweep_deg=60.0, jitter=0.01, seed=60 + levels,
                                         kind="B")
    plan = engine.Plan([im.shape[:2] for im in imgs], rots, intrs, True, 10 ** 9)
    mosaic, fl, _, _ = eng.stitch(eng.upload_frames(imgs), plan, "multiband", levels,
                                  want_float=True)
    ref_u8, ref_f = oracle.stitch(imgs, rots, intrs, "multiband", levels, max_resolution=10 ** 9,
                                  return_float=True)
    assert rel_l2(fl.cpu().numpy(), ref_f) <= REL_TOL
    assert np.abs(mosaic.cpu().numpy().astype(int) - ref_u8.astype(int)).max() <= 1


@pytest.mark.parametrize("case", ["single", "disjoint", "tiny", "tall"])
def test_degenerate_scenes_against_oracle(eng, oracle, case):
    """Edge cases of the fused path against the oracle: one frame alone; two frames that do
    not overlap (a hole in the coverage); frames smaller than the blur radius (the window
    reflects more than once); a portrait frame.  Valid mask bit-exact, float mosaic within
    1e-4, uint8 within one level, for the multiband, linear and paste blenders."""
    from pano360_amd import engine, synth
    if case == "single":
        imgs, rots, intrs = synth.make_scene(1, 200, 120, sweep_deg=0.0, seed=1, kind="B")
    elif case == "disjoint":
        imgs, rots, intrs = synth.make_scene(2, 160, 100, step_deg=75.0, seed=2, kind="B")
    elif case == "tiny":
        imgs, rots, intrs = synth.make_scene(3, 24, 14, sweep_deg=50.0, jitter=0.01, seed=3,
                                             kind="A")
    else:
        imgs, rots, intrs = synth.make_scene(4, 90, 260, sweep_deg=40.0, jitter=0.01, seed=4,
                                             kind="B")
    shapes = [im.shape[:2] for im in imgs]
    frames = eng.upload_frames(imgs)
    for blend in ("multiband", "linear", "none"):
        plan = engine.Plan(shapes, rots, intrs, blend == "multiband", 10 ** 9)
        mosaic, fl, valid, _ = eng.stitch(frames, plan, blend, 5, want_float=blend == "multiband")
        ref = oracle.stitch(imgs, rots, intrs, blend, 5, max_resolution=10 ** 9,
                            return_float=blend == "multiband")
        ref_u8, ref_f = ref if blend == "multiband" else (ref, None)
        got = mosaic.cpu().numpy()
        assert got.shape == ref_u8.shape, (case, blend)
        assert np.abs(got.astype(int) - ref_u8.astype(int)).max() <= (1 if blend == "multiband" else 0)
        if blend == "multiband":
            assert rel_l2(fl.cpu().numpy(), ref_f) <= REL_TOL, (case, blend)
            _, ref_patches, _ = oracle.warp_all(imgs, rots, intrs, True, max_resolution=10 ** 9)
            assert np.array_equal(valid.cpu().numpy().astype(bool),
                                  oracle.valid(ref_patches, plan.shape)), case
    if case == "disjoint":
        assert not valid.cpu().numpy().all()             # there is a hole between the frames


@pytest.mark.parametrize("kind", ["A", "B"])
def test_interior_shortcut(eng, oracle, kind):
    """Default fused path: farther than the largest Gaussian radius from any seam
    (or border of the covered area) the band-pass stack telescopes to the owner's
    warped colour, so blur and gather are skipped there and the frame is sampled
    directly.  Against the full blend this moves the float mosaic by float32
    rounding only (and, through the uint8 truncation of stitcher.py:241, a few
    values by one level); against the oracle it stays inside the stated bars."""
    import torch
    from pano360_amd import engine, synth
    imgs, rots, intrs = synth.make_scene(6, 640, 360, sweep_deg=50.0, jitter=0.01, seed=31,
                                         kind=kind)
    plan = engine.Plan([im.shape[:2] for im in imgs], rots, intrs, True, 10 ** 9)
    frames = eng.upload_frames(imgs)
    m1, f1, v1, _ = eng.stitch(frames, plan, "multiband", 5, want_float=True)      # shortcut on
    m2, f2, v2, _ = eng.stitch(frames, plan, "multiband", 5, want_float=True, shortcut=False)
    assert torch.equal(v1, v2)
    assert (f1 - f2).abs().max().item() <= 2.5e-7
    assert (m1.int() - m2.int()).abs().max().item() <= 1
    owner, _ = eng.ownership_cameras(eng.upload_plan(plan))
    interior = eng.interior_map(owner, 43).bool()
    frac = interior.float().mean().item()
    assert 0.2 < frac < 0.8                         # both branches are exercised
    ib = eng.interior_block
    big = interior.repeat_interleave(ib, 0).repeat_interleave(ib, 1)[:plan.shape[0], :plan.shape[1]]
    assert torch.equal(m1[~big], m2[~big])          # untouched outside the interior
    # the interior test is conservative: every pixel of an interior block has one
    # owner over the whole (2R+1)^2 window
    own = owner.cpu().numpy().astype(np.int32)
    from scipy.ndimage import maximum_filter, minimum_filter
    same = (minimum_filter(own, size=87, mode="nearest") == own) & \
           (maximum_filter(own, size=87, mode="nearest") == own) & (own >= 0)
    assert same[big.cpu().numpy()].all()
    ref_u8, ref_f = oracle.stitch(imgs, rots, intrs, "multiband", 5, max_resolution=10 ** 9,
                                  return_float=True)
    assert rel_l2(f1.cpu().numpy(), ref_f) <= REL_TOL
    assert np.abs(m1.cpu().numpy().astype(int) - ref_u8.astype(int)).max() <= 1


@pytest.mark.parametrize("levels", [5, 6, 3])
def test_level_classes_of_the_interior_map(eng, levels):
    """``pano_interior_classes``: a block's class = the number of leading levels whose Gaussian
    radius finds one owner around every pixel of the block.  Against the one-radius map run once per
    level (their sum), against a brute-force window test (conservative: class > k implies a
    single-owner (2 r_k + 1)^2 window), nested, and the same on a column strip."""
    from scipy.ndimage import maximum_filter, minimum_filter
    from pano360_amd import engine, synth
    _, rots, intrs = synth.make_scene(7, 640, 360, sweep_deg=60.0, jitter=0.012, seed=91, kind="A")
    plan = eng.upload_plan(engine.Plan([(360, 640)] * 7, rots, intrs, True, 10 ** 9))
    owner, _ = eng.ownership_cameras(plan)
    radii = eng.level_radii(levels)
    assert radii == sorted(radii) and len(radii) == levels - 1
    interior, classes = eng.interior_classes(owner, radii)
    stack = sum(eng.interior_map(owner, r).int() for r in radii)
    assert torch_equal(classes.int(), stack)
    assert torch_equal(interior, eng.interior_map(owner, radii[-1]))
    assert torch_equal(interior.bool(), classes == len(radii))
    hist = [int((classes == j).sum().item()) for j in range(levels)]
    assert all(v > 0 for v in hist), hist          # every class occurs
    own = owner.cpu().numpy().astype(np.int32)
    ib = eng.interior_block
    H, W = own.shape
    big = classes.cpu().numpy().repeat(ib, 0).repeat(ib, 1)[:H, :W]
    for k, r in enumerate(radii):
        size = 2 * r + 1
        same = (minimum_filter(own, size=size, mode="nearest") == own) & \
               (maximum_filter(own, size=size, mode="nearest") == own) & (own >= 0)
        assert same[big > k].all(), (k, r)
    # a strip classifies its blocks as the whole mosaic does
    c0, c1 = 4 * (W // 8), 4 * (W // 8) + 256
    margin = ib * ((radii[-1] + 2 * ib - 2) // ib) + ib - 1
    ext = (max(c0 - margin, 0), min(c1 + margin, W))
    owner_s, _ = eng.ownership_cameras(plan, strip=ext)
    _, cls_s = eng.interior_classes(owner_s, radii, ext)
    b0, b1 = (c0 + ib - 1) // ib, c1 // ib
    assert torch_equal(cls_s[:, b0:b1], classes[:, b0:b1])


def torch_equal(a, b):
    import torch
    return torch.equal(a, b)


@pytest.mark.parametrize("levels,kind", [(5, "A"), (6, "B"), (3, "A")])
def test_collapse_with_level_classes(eng, oracle, levels, kind):
    """The collapse gathers, on a pixel of class j >= 1, the colour of copy j - 1 and the copies
    j and up only (the levels below telescope to the owner's I - G_{j-1} I): against the collapse
    that gathers every copy (option PANO_OPT_LEVEL_CLASSES = 0, rounds 1 - 5) the float mosaic moves
    by float32 rounding, uint8 by at most one level, class-0 pixels not at all; against the oracle
    it stays inside the stated bars.  The launch-by-launch path and the native call agree bit for
    bit, and so do column strips.  (The option is OFF by default: it takes 13 % off the collapse's
    traffic and adds 4 - 7 % to its time, profiles/r06/ab_level_classes.txt.)"""
    import torch
    from pano360_amd import _lib, dist as pdist, engine, synth
    imgs, rots, intrs = synth.make_scene(6, 640, 360, sweep_deg=50.0, jitter=0.01, seed=33, kind=kind)
    shapes = [im.shape[:2] for im in imgs]
    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    frames = eng.upload_frames(imgs)
    on = engine.Engine(eng.device)
    on.set_option(_lib.OPT_LEVEL_CLASSES, 1)
    m1, f1, v1, _ = on.stitch(frames, plan, "multiband", levels, want_float=True)
    classes = on.last_classes.clone()
    plain = eng                                     # the default: no level classes
    assert plain.get_option(_lib.OPT_LEVEL_CLASSES) == 0
    m0, f0, v0, _ = plain.stitch(frames, plan, "multiband", levels, want_float=True)
    assert plain.last_classes is None
    assert torch.equal(v1, v0)
    assert (f1 - f0).abs().max().item() <= 2.5e-7
    assert (m1.int() - m0.int()).abs().max().item() <= 1
    ib = eng.interior_block
    H, W = plan.shape
    px = classes.repeat_interleave(ib, 0).repeat_interleave(ib, 1)[:H, :W]
    assert torch.equal(m1[px == 0], m0[px == 0]) and torch.equal(f1[px == 0], f0[px == 0])
    # the classes in between exist and save the collapse bytes (where a level's tap sum is exactly
    # 1 the two collapses even agree bit for bit: (x 1) / 1 = x)
    mid = (px > 0) & (px < levels - 1)
    assert mid.float().mean().item() > 0.03
    with_classes, without = on.gather_bytes(plan.shape, levels), plain.gather_bytes(plan.shape, levels)
    assert 0.5 * without < with_classes < 0.97 * without, (with_classes, without)
    # launch by launch from Python = the native call
    loose = engine.Engine(eng.device)
    loose.set_option(_lib.OPT_LEVEL_CLASSES, 1)
    loose.native_stitch = False
    m2, f2, _, _ = loose.stitch(frames, plan, "multiband", levels, want_float=True)
    assert torch.equal(m2, m1) and torch.equal(f2.view(torch.int32), f1.view(torch.int32))
    # column strips compose the same mosaic
    strips, _ = pdist.emulate_on_one_device(on, imgs, rots, intrs, levels, 3)
    assert torch.equal(strips, m1)
    ref_u8, ref_f = oracle.stitch(imgs, rots, intrs, "multiband", levels, max_resolution=10 ** 9,
                                  return_float=True)
    assert rel_l2(f1.cpu().numpy(), ref_f) <= REL_TOL
    assert np.abs(m1.cpu().numpy().astype(int) - ref_u8.astype(int)).max() <= 1


def test_closed_360_sweep_with_seam_straddling_frames(eng, oracle):
    """BASELINE config 5 in miniature: a closed 360 degree sweep.  Frames that
    straddle the +-pi seam get full-width patches (the reference has no
    wrap-around handling, stitcher.py:107-122) and own pixels at BOTH ends of the
    mosaic; they are split into one record per column span so that only the
    neighbourhood of what they own is warped and blurred."""
    import torch
    from pano360_amd import engine, synth
    n, w, h = 24, 160, 90
    imgs, rots, intrs = synth.make_scene(n, w, h, step_deg=15.0, jitter=0.004, seed=77, kind="B",
                                         n_levels=6)
    plan = engine.Plan([(h, w)] * n, rots, intrs, True, 10 ** 9)
    widths = [r[3] - r[2] for r in plan.rects]
    assert max(widths) > 0.9 * plan.shape[1] and min(widths) < 0.3 * plan.shape[1]
    frames = eng.upload_frames(imgs)
    m1, f1, v1, patches = eng.stitch(frames, plan, "multiband", 6, want_float=True,
                                     shortcut=False)
    m2, f2, v2, _ = eng.stitch(frames, plan, "multiband", 6, want_float=True, fused=False)
    assert torch.equal(m1, m2) and torch.equal(v1, v2)
    assert torch.equal(f1.view(torch.int32), f2.view(torch.int32))
    assert len(patches) > n                          # some frame was split into spans
    assert patches.warped_pixels < 0.6 * plan.patch_pixels
    ref_u8, ref_f = oracle.stitch(imgs, rots, intrs, "multiband", 6, max_resolution=10 ** 9,
                                  return_float=True)
    assert rel_l2(f1.cpu().numpy(), ref_f) <= REL_TOL
    assert np.abs(m1.cpu().numpy().astype(int) - ref_u8.astype(int)).max() <= 1
    _, ref_patches, _ = oracle.warp_all(imgs, rots, intrs, True, 10 ** 9)
    assert np.array_equal(v1.cpu().numpy().astype(bool), oracle.valid(ref_patches, plan.shape))
    for kind in ("linear", "none"):
        plan_l = engine.Plan([(h, w)] * n, rots, intrs, False, 10 ** 9)
        got, _, _, _ = eng.stitch(frames, plan_l, kind)
        assert np.array_equal(got.cpu().numpy(),
                              oracle.stitch(imgs, rots, intrs, kind, max_resolution=10 ** 9))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_column_strips_compose_the_single_gpu_mosaic(eng, world):
    """The multi-GPU decomposition, every rank emulated in turn on this one GPU:
    each rank holds only the frames near its strip, evaluates ownership only
    around its strip, and the concatenated strips are the single-GPU mosaic bit
    for bit (tests/test_dist_cpu.py covers the gather itself over gloo)."""
    import torch
    from pano360_amd import dist as pdist
    from pano360_amd import engine, synth
    imgs, rots, intrs = synth.make_scene(10, 480, 270, sweep_deg=120.0, jitter=0.01, seed=41,
                                         kind="A")
    plan = engine.Plan([im.shape[:2] for im in imgs], rots, intrs, True, 10 ** 9)
    whole, _, _, _ = eng.stitch(eng.upload_frames(imgs), plan, "multiband", 5)
    strips, bounds = pdist.emulate_on_one_device(eng, imgs, rots, intrs, 5, world)
    assert bounds[-1] == plan.shape[1]
    assert torch.equal(strips, whole)
    # those were strips of equal work (ShardedStitcher's default: cut at the quantiles of
    # Engine.column_costs); strips of equal width compose the same mosaic
    even, even_bounds = pdist.emulate_on_one_device(eng, imgs, rots, intrs, 5, world, balance=False)
    assert even_bounds == pdist.strip_bounds(plan.shape[1], world) and torch.equal(even, whole)
    cost = eng.column_costs(plan, 5)
    assert cost.shape == (plan.shape[1],) and np.isfinite(cost).all() and cost.min() > 0
    assert bounds == pdist.balanced_strip_bounds(cost, world)
    work = lambda b: max(cost[b[r]:b[r + 1]].sum() for r in range(world))      # noqa: E731
    assert work(bounds) <= work(even_bounds) * 1.02
    # the ranks did not all need all frames
    shapes = [im.shape[:2] for im in imgs]
    held = [len(pdist.ShardedStitcher(eng, shapes, rots, intrs, 5, r, world,
                                      exchange=None).my_frames) for r in range(world)]
    assert min(held) < len(imgs)


@pytest.mark.parametrize("levels", [1, 2, 6])
def test_balanced_strips_at_other_level_counts(eng, levels):
    """Strips of equal work with no blur at all (one level: every valid pixel weighs as a seam pixel
    in the cost), one blur, five: the cut is the engine's, the strips compose the whole mosaic."""
    import torch
    from pano360_amd import dist as pdist
    from pano360_amd import engine, synth
    imgs, rots, intrs = synth.make_scene(7, 400, 225, sweep_deg=100.0, jitter=0.01, seed=47, kind="A")
    plan = engine.Plan([im.shape[:2] for im in imgs], rots, intrs, True, 10 ** 9)
    whole, _, _, _ = eng.stitch(eng.upload_frames(imgs), plan, "multiband", levels)
    strips, bounds = pdist.emulate_on_one_device(eng, imgs, rots, intrs, levels, 3)
    assert bounds == pdist.balanced_strip_bounds(eng.column_costs(plan, levels), 3)
    assert bounds[0] == 0 and bounds[-1] == plan.shape[1] and torch.equal(strips, whole)


def test_sharded_stitcher_world_1_equals_stitch(eng):
    import torch
    from pano360_amd import dist as pdist
    from pano360_amd import engine, synth
    imgs, rots, intrs = synth.make_scene(5, 320, 180, sweep_deg=70.0, seed=43, kind="B")
    shapes = [im.shape[:2] for im in imgs]
    st = pdist.ShardedStitcher(eng, shapes, rots, intrs, 5, 0, 1, depth=1)
    assert st.my_frames == list(range(5))
    plan, mosaic, _ = st.step(eng.upload_frames(imgs))
    whole, _, _, _ = eng.stitch(eng.upload_frames(imgs), plan, "multiband", 5)
    assert torch.equal(mosaic, whole)
    # pipelined (depth 2): a step hands back the stitch before it, finish() the last one
    st2 = pdist.ShardedStitcher(eng, shapes, rots, intrs, 5, 0, 1, depth=2, exchange="reduce")
    frames = eng.upload_frames(imgs)
    assert st2.step(frames)[1] is None
    assert torch.equal(st2.step(frames)[1], whole)
    assert torch.equal(st2.finish(), whole)
    # two stitches in flight: consecutive steps alternate between two engines, each on a stream
    # and with exchange buffers of its own (bench.py's strips); every mosaic is the same
    st3 = pdist.ShardedStitcher([eng, engine.Engine(eng.device)], shapes, rots, intrs, 5, 0, 1,
                                depth=2)
    assert len(st3.lanes) == 2 and st3.lanes[1][1] is not None
    got = [st3.step(frames)[1] for _ in range(5)] + [st3.finish()]
    torch.cuda.synchronize()
    assert got[0] is None and all(torch.equal(m, whole) for m in got[1:])
    # a fixed rig (keep_geometry): the lanes' repeat stitches run warp + blur + collapse only - with
    # other pixels every step the mosaics are still the whole engine's
    lanes = [engine.Engine(eng.device), engine.Engine(eng.device)]
    st4 = pdist.ShardedStitcher(lanes, shapes, rots, intrs, 5, 0, 1, depth=2, keep_geometry=True)
    want = []
    for k in range(7):
        fk = eng.upload_frames([np.roll(im, 7 * k, axis=1) for im in imgs])
        want.append(eng.stitch(fk, plan, "multiband", 5)[0])
        m = st4.step(fk)[1]
        torch.cuda.synchronize()
        assert (m is None) == (k == 0) and (m is None or torch.equal(m, want[k - 1])), k
    assert torch.equal(st4.finish(), want[-1])
    assert all(e.keep_geometry and e.last_kept_geometry for e in lanes)
    with pytest.raises(Exception):       # a needed frame that is not resident
        eng.multiband_fused(eng.upload_frames(imgs[:2]), eng.upload_plan(plan), 5,
                            frame_ids=[0, 1])


def test_cli_plumbing_config_1(eng, oracle, tmp_path, monkeypatch):
    """BASELINE config 1 (`stitcher.py DIR --shrink 4 --blend linear`): the CMU2
    images are not redistributable, so the plumbing runs on a synthetic directory
    plus the `ba_<name>_s<shrink>.pkl` camera cache the reference CLI writes
    (stitcher.py:414, 430-439).  The written image must equal the oracle's
    linear mosaic."""
    import pickle
    from PIL import Image as PilImage
    import bundle_adj
    import stitcher as top
    from pano360_amd import synth
    imgs, rots, intrs = synth.make_scene(5, 200, 120, sweep_deg=80.0, jitter=0.01, seed=9,
                                         kind="B")
    data = tmp_path / "CMU2"
    data.mkdir()
    for i, im in enumerate(imgs):
        PilImage.fromarray(im).save(data / f"frame{i}.png")
    regions = [bundle_adj.Image(im, r, k) for im, r, k in zip(imgs, rots, intrs)]
    with open(tmp_path / "ba_CMU2_s4.0.pkl", "wb") as fid:
        pickle.dump(regions, fid, protocol=pickle.HIGHEST_PROTOCOL)
    monkeypatch.chdir(tmp_path)
    out = tmp_path / "mosaic.png"
    got = top.main([str(data), "--shrink", "4", "--blend", "linear", "-c", "-o", str(out)])
    want = oracle.stitch(imgs, rots, intrs, "linear", crop=True)
    assert np.array_equal(got, want)
    # cv2.imwrite stores BGR; the file holds the same pixels, channels reversed
    assert np.array_equal(np.asarray(PilImage.open(out))[..., ::-1], want)
    with pytest.raises(SystemExit):      # no camera cache: registration is out of scope
        top.main([str(tmp_path / "nowhere"), "-b", "linear"])


def test_cli_ingest_reads_and_shrinks_on_the_device(eng, oracle, tmp_path, monkeypatch):
    """The head of the reference's main (stitcher.py:415-421): list, read, shrink.  With a
    cameras-only cache (``img=None`` records) the CLI reads the directory in ``os.listdir``
    order, shrinks the images on the device (``pano_resize_u8``) and stitches them: the mosaic
    equals the oracle's on the images shrunk by the restated ``cv2.resize``."""
    import os
    import pickle
    from PIL import Image as PilImage
    import bundle_adj
    import laplacian_oracle as lo
    import stitcher as top
    from pano360_amd import synth
    imgs, rots, intrs = synth.make_scene(4, 400, 240, sweep_deg=70.0, jitter=0.01, seed=21,
                                         kind="B")
    small_intr = synth.make_cameras(4, 200, 120, sweep_deg=70.0)[1]
    data = tmp_path / "RIG"
    data.mkdir()
    for i, im in enumerate(imgs):
        PilImage.fromarray(np.ascontiguousarray(im[..., ::-1])).save(data / f"f{i}.png")   # imread: BGR
    (data / "notes.txt").write_text("not an image")
    order = [int(f[1]) for f in os.listdir(data) if f.endswith(".png")]
    regions = [bundle_adj.Image(None, rots[i], small_intr[i]) for i in order]
    with open(tmp_path / "ba_RIG_s2.0.pkl", "wb") as fid:
        pickle.dump(regions, fid, protocol=pickle.HIGHEST_PROTOCOL)
    monkeypatch.chdir(tmp_path)
    got = top.main([str(data), "--shrink", "2", "--blend", "linear"])
    shrunk = [lo.shrink(imgs[i], 2) for i in order]
    want = oracle.stitch(shrunk, [rots[i] for i in order], [small_intr[i] for i in order], "linear")
    assert np.array_equal(got, want)
    os.remove(tmp_path / "ba_RIG_s2.0.pkl")
    # a JPEG whose EXIF orientation says "rotated": cv2.imread applies the tag (its default),
    # so does the ingest - the frame arrives upright with its shape swapped
    from pano360_amd import stitcher as product
    rot = tmp_path / "ROT"
    rot.mkdir()
    exif = PilImage.Exif()
    exif[0x0112] = 6                          # rotate 90 degrees clockwise to display
    PilImage.fromarray(np.ascontiguousarray(imgs[0][..., ::-1])).save(rot / "a.png", exif=exif)
    frame = product.ingest(str(rot), 1)[0]
    assert tuple(frame.shape) == (400, 240, 3)
    assert np.array_equal(frame.cpu().numpy(), np.rot90(imgs[0], -1))
    # without the camera cache nothing is decoded or uploaded: the exit comes first
    monkeypatch.setattr(product, "ingest", lambda *a: pytest.fail("ingest before the cache check"))
    with pytest.raises(SystemExit, match="not found"):
        top.main([str(data), "--shrink", "2"])


# ------------------------------------------------------------------- crop
def test_crop_rectangles_bit_exact(eng, oracle):
    import torch
    from pano360_amd import stitcher
    g = load_golden("pure")
    for i in range(int(g["n_crop"])):
        mask = g[f"crop_mask_{i}"]
        dev = torch.from_numpy(mask.astype(np.uint8)).to(eng.device)
        assert eng.crop_rect(dev) == tuple(g[f"crop_rect_{i}"]), i
    rng = np.random.default_rng(7)
    for shape, p in (((97, 300), 0.9), ((33, 1000), 0.995), ((300, 129), 0.8), ((5, 5), 1.0),
                     ((64, 64), 0.5), ((211, 777), 0.999)):
        mask = rng.random(shape) < p
        dev = torch.from_numpy(mask.astype(np.uint8)).to(eng.device)
        assert eng.crop_rect(dev) == oracle.crop_rect(mask), (shape, p)
    with pytest.raises(UnboundLocalError):
        stitcher.crop_mosaic(np.zeros((4, 5, 3), np.uint8), np.zeros((4, 5), bool))
    mosaic = rng.integers(0, 255, (23, 37, 3), dtype=np.uint8)
    view = stitcher.crop_mosaic(mosaic, g["crop_mask_6"])
    y0, x0, h, w = g["crop_rect_6"]
    assert np.shares_memory(view, mosaic) and view.shape == (h, w, 3)
    assert np.array_equal(view, mosaic[y0:y0 + h, x0:x0 + w])


# ------------------------------------------------------------------ filters
@pytest.mark.parametrize("shape", [(37, 53), (1, 9), (200, 3), (130, 700), (5, 1030)])
@pytest.mark.parametrize("sigma", [1.0, 4.0, 4 * np.sqrt(7.0), 12.0])
def test_blur_plane_matches_oracle(eng, oracle, shape, sigma):
    """REFLECT_101 with multiple reflections (tiny planes), tile seams (wide /
    tall planes) and every multiband aperture; float tolerance 1e-6 absolute on
    unit-range data (FMA vs mul+add rounding only)."""
    import torch
    from pano360_amd import engine
    img = np.random.default_rng(11).random(shape).astype(np.float32)
    k = engine.gaussian_ksize(sigma)
    got = eng.blur_plane(torch.from_numpy(img).to(eng.device), k, sigma).cpu().numpy()
    want = oracle.gaussian_blur(img, k, sigma)
    assert np.abs(got - want).max() <= 1e-6


def test_gaussian_filter_and_pyr_down(eng, oracle):
    from pano360_amd import features
    g = load_golden("pure")
    for sigma, key in ((1.0, "gf_s1"), (2.0, "gf_s2")):
        got = features.gaussian_filter(g["gf_img"], sigma)
        assert got.dtype == np.float32 and np.abs(got - g[key]).max() <= 1e-6
    p1 = features.pyr_down(g["gf_img"])
    assert np.array_equal(bits(p1), bits(g["pyr_1"]))
    assert np.array_equal(bits(features.pyr_down(p1)), bits(g["pyr_2"]))


@pytest.mark.parametrize("shape", [(48, 64), (61, 35)])
def test_sift_scale_space_matches_oracle(eng, shape):
    """Gaussian / DoG pyramid of the SIFT front end (features.py:192-201 via
    OpenCV; restated, parity unpinned): grey conversion, 2x bilinear base and
    nearest halving are exact, the blurred layers agree to float tolerance
    (values are on the 0..255 scale, one FMA per tap vs mul+add)."""
    import sift_pyramid as ref
    from pano360_amd import features
    img = np.random.default_rng(3).integers(0, 256, shape + (3,), dtype=np.uint8)
    gauss, dog = features.sift_pyramid(img)
    want_g, want_d = ref.sift_pyramid(img)
    assert len(gauss) == len(want_g) == features.sift_octaves(*shape) == ref.n_octaves(*shape)
    for o, (go, wo) in enumerate(zip(gauss, want_g)):
        assert len(go) == len(wo) == 6 and len(dog[o]) == 5
        for layer, (a, b) in enumerate(zip(go, wo)):
            assert a.shape == b.shape, (o, layer)
            assert np.abs(a - b).max() <= 2e-4, (o, layer)
        for layer, (a, b) in enumerate(zip(dog[o], want_d[o])):
            assert np.abs(a - b).max() <= 4e-4, (o, layer)
    # the exact pieces, bit for bit
    dev = features._Dev(eng)
    frame = eng.upload_frames([img])[0]
    gray = dev.gray(frame)
    assert np.array_equal(gray.cpu().numpy(), ref.gray_u8(img))
    assert np.array_equal(bits(dev.up2(gray).cpu().numpy()), bits(ref.resize_up2(ref.gray_u8(img))))
    assert np.array_equal(dev.half(gray).cpu().numpy(), ref.decimate2(ref.gray_u8(img)))
    assert features.sift_sigmas() == ref.sigmas()


# ------------------------------------------- size-independent properties
def _match_keypoints(got, want):
    """Greedy one-to-one matching on (x, y, size, angle); returns index pairs."""
    pairs, used = [], set()
    for i, k in enumerate(want):
        d = (np.abs(got["x"] - k["x"]) + np.abs(got["y"] - k["y"]) + np.abs(got["size"] - k["size"])
             + np.minimum(np.abs(got["angle"] - k["angle"]), 360 - np.abs(got["angle"] - k["angle"])) / 90)
        for j in np.argsort(d)[:3]:
            if d[j] < 0.05 and int(j) not in used:
                used.add(int(j))
                pairs.append((int(j), i))
                break
    return pairs


@pytest.mark.parametrize("kind,seed", [("B", 3), ("blobs", 0)])
def test_sift_keypoints_and_descriptors_match_oracle(eng, kind, seed):
    """features.sift_detector's keypoints and descriptors (OpenCV SIFT restated, parity
    unpinned) against the NumPy restatement: (1) on the oracle's own scale space the
    kernels reproduce every keypoint and descriptor (LDS-atomic summation order aside);
    (2) end to end, the few extrema that sit on a threshold may come or go."""
    import torch
    import sift_oracle
    import sift_pyramid as sp
    from pano360_amd import features, synth
    w, h = 144, 104
    if kind == "blobs":
        rng = np.random.default_rng(seed)
        yy, xx = np.mgrid[:h, :w]
        img = np.zeros((h, w), np.float64)
        for _ in range(25):
            cx, cy, s = rng.uniform(8, w - 8), rng.uniform(8, h - 8), rng.uniform(1.5, 6)
            img += rng.uniform(-1, 1) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
        img = np.clip(128 + 100 * img, 0, 255).astype(np.uint8)
        img = np.stack([img, np.roll(img, 1, 0), np.roll(img, 1, 1)], axis=-1)
    else:
        img = synth.make_frame(seed, w, h, kind)
    g_or, d_or = sp.sift_pyramid(img)
    want_k, want_d = sift_oracle.detect_and_compute(g_or, d_or)
    want = np.zeros(len(want_k), features.KP_DTYPE)
    for i, k in enumerate(want_k):
        want[i] = (k["x"], k["y"], k["size"], k["angle"], k["response"], k["octave"], k["r"], k["c"])
    assert len(want) > 15
    frame = eng.upload_frames([img])[0]
    # (1) same scale space
    stacks = ([torch.from_numpy(np.stack(o)).to(eng.device) for o in g_or],
              [torch.from_numpy(np.stack(o)).to(eng.device) for o in d_or])
    got, desc = features.sift_detect_device(frame, pyramid=stacks)
    desc = desc.cpu().numpy()
    assert len(got) == len(want)
    assert np.array_equal(got["octave"], want["octave"])
    for key, tol in (("x", 2e-3), ("y", 2e-3), ("size", 2e-3), ("response", 1e-5)):
        assert np.abs(got[key] - want[key]).max() <= tol, key
    dang = np.abs(got["angle"] - want["angle"])
    assert np.minimum(dang, 360 - dang).max() <= 0.05
    diff = np.abs(desc - want_d)
    assert diff.max() <= 2 and (diff > 0).mean() < 0.02
    # (2) end to end
    got2, desc2 = features.sift_detect_device(frame)
    pairs = _match_keypoints(got2, want)
    assert len(pairs) >= 0.97 * len(want) and abs(len(got2) - len(want)) <= 0.03 * len(want) + 2
    gi, wi = np.array(pairs).T
    assert np.abs(desc2.cpu().numpy()[gi] - want_d[wi]).mean() < 0.5
    # the reference-facing closure: keypoint objects + RootSIFT rows of unit L2 norm
    kp, des = features.sift_detector()(img)
    assert len(kp) == len(got2) and des.shape == (len(kp), 128)
    np.testing.assert_allclose((des ** 2).sum(axis=1), 1.0, atol=1e-5)
    assert abs(kp[0].pt[0] - got2["x"][0]) < 1e-6 and kp[0].octave == got2["octave"][0]


def test_sift_pipeline_graph_replay_equals_launch_by_launch(eng):
    """``features.SiftPipeline`` - one native call per frame (``pano_sift_detect``), captured into
    a HIP graph the second time a workspace is used and replayed afterwards - against the entry
    points called one by one from Python (``sift_pyramid_device`` + ``sift_detect_async`` on that
    pyramid): the scale space bit for bit, the keypoints bit for bit in position, size, response and
    octave (angles and descriptors to the summation order of their LDS atomics, as between any two
    runs).  Three frames go round two workspaces four times: every workspace sees a launch-by-launch
    frame, a captured one and replays; then once more with graphs switched off."""
    import torch
    from pano360_amd import _lib, engine, features, synth
    w, h = 320, 200
    imgs = [synth.make_frame(70 + i, w, h, "B") for i in range(3)]
    frames = eng.upload_frames(imgs)
    want = []
    for frame in frames:
        pyr = features.sift_pyramid_device(frame, eng=eng)
        kps, desc = features.sift_detect_async(frame, pyramid=pyr, eng=eng).result()
        want.append((pyr, kps, desc.cpu().numpy()))
    assert all(len(wk) > 50 for _, wk, _ in want)

    def same(det, ref):
        (g_ref, d_ref), kps_ref, desc_ref = ref
        kps, desc = det.result()
        gauss, dog = det.pyramid
        assert all(torch.equal(a, b) for a, b in zip(gauss, g_ref))
        assert all(torch.equal(a, b) for a, b in zip(dog, d_ref))
        assert len(kps) == len(kps_ref)
        for key in ("x", "y", "size", "response", "octave", "r", "c"):
            assert np.array_equal(kps[key], kps_ref[key]), key
        dang = np.abs(kps["angle"] - kps_ref["angle"])
        assert np.minimum(dang, 360 - dang).max() <= 0.01
        diff = np.abs(desc.cpu().numpy() - desc_ref)
        assert diff.max() <= 2 and (diff > 0).mean() < 0.02

    use = engine.Engine(eng.device)
    stream = torch.cuda.Stream(eng.device)
    with torch.cuda.stream(stream):
        pipe = features.SiftPipeline(use, h, w, depth=2, max_keypoints=1 << 14)
        for k in range(12):
            same(pipe.detect(frames[k % 3]), want[k % 3])
            if k == 1:
                assert not pipe.replaying              # both workspaces have run launch by launch only
        assert pipe.replaying
        # the scale space alone goes through the same call (its own graph: `detect` is in the key)
        for k in range(6):
            gauss, dog = pipe.pyramid(frames[k % 3])
            assert all(torch.equal(a, b) for a, b in zip(gauss, want[k % 3][0][0]))
            assert all(torch.equal(a, b) for a, b in zip(dog, want[k % 3][0][1]))
        # graphs off: launch by launch every time, same results
        use.set_option(_lib.OPT_SIFT_GRAPH, 0)
        plain = features.SiftPipeline(use, h, w, depth=2, max_keypoints=1 << 14)
        for k in range(5):
            same(plain.detect(frames[k % 3]), want[k % 3])
        # on the legacy default stream too (captured on a stream of the library's own)
    use.set_option(_lib.OPT_SIFT_GRAPH, 1)
    torch.cuda.synchronize()
    pipe0 = features.SiftPipeline(use, h, w, depth=1, max_keypoints=1 << 14)
    for k in range(4):
        same(pipe0.detect(frames[k % 3]), want[k % 3])
    assert pipe0.replaying
    # the reference-facing entry point rides on the engine's pipeline
    kps, desc = features.sift_detect_device(frames[0], eng=use)
    assert len(kps) == len(want[0][1]) and np.array_equal(kps["x"], want[0][1]["x"])


@pytest.mark.parametrize("h,w", [(33, 47), (16, 16), (9, 200), (64, 65), (12, 11)])
def test_sift_native_call_on_small_and_odd_frames(eng, h, w):
    """``pano_sift_detect`` on frames whose pyramid ends early, whose octaves are odd-sized or
    narrower than the extrema search's border: the same keypoints as the entry points called one by
    one (none at all on the smallest), through the eager frame, the captured one and a replay."""
    from pano360_amd import features, synth
    frame = eng.upload_frames([synth.make_frame(h * 1000 + w, w, h, "B")])[0]
    pyr = features.sift_pyramid_device(frame, eng=eng)
    want, _ = features.sift_detect_async(frame, pyramid=pyr, eng=eng).result()
    pipe = features.SiftPipeline(eng, h, w, depth=1, max_keypoints=1 << 12)
    for _ in range(3):
        det = pipe.detect(frame)
        got, desc = det.result()
        assert len(got) == len(want) and desc.shape == (len(want), 128)
        assert np.array_equal(got["x"], want["x"]) and np.array_equal(got["octave"], want["octave"])
        assert all(a.shape == b.shape for a, b in zip(det.pyramid[0], pyr[0]))


def test_sift_kernels_known_answers(eng):
    """The HIP detector (``features.sift_detect_device``: scale space, extrema, orientations,
    descriptors) against what is known without OpenCV or any restatement of it - the checks of
    tests/test_oracle_golden.py, where the oracle passes them too: Gaussian blobs are found at their
    centres (+ 0.25 px: the doubled first octave) with size / 2 = s / sqrt(k), the closed-form
    maximum of the difference of Gaussians in scale; a weak blob on a strong linear ramp gets the
    ramp's direction as its angle; a quarter turn of the image turns the keypoints and leaves their
    descriptors alone."""
    from test_oracle_golden import (SIFT_BLOBS, _blob_scene, check_blob_keypoints, check_ramp_orientation,
                                    check_rot90)
    from pano360_amd import features

    def detect(bgr, with_desc=False):
        kps, desc = features.sift_detect_device(eng.upload_frames([np.ascontiguousarray(bgr)])[0], eng=eng)
        return (kps, desc.cpu().numpy()) if with_desc else kps
    check_blob_keypoints(detect(_blob_scene(SIFT_BLOBS)))
    check_ramp_orientation(detect)
    check_rot90(detect)


def test_sift_graphs_are_evicted_and_recaptured(eng):
    """A context keeps at most eight captured launch sequences (least recently used leaves): eleven
    pipelines of different frame sizes on one engine, each run into its replay, then the first one
    again - its graph is gone, it runs launch by launch, is captured again and replays - with the
    same keypoints every time."""
    import torch
    from pano360_amd import engine, features, synth
    use = engine.Engine(eng.device)
    stream = torch.cuda.Stream(eng.device)
    sizes = [(96 + 8 * k, 128 + 16 * k) for k in range(11)]
    with torch.cuda.stream(stream):
        pipes, first = [], []
        for h, w in sizes:
            frame = use.upload_frames([synth.make_frame(h + w, w, h, "B")])[0]
            pipe = features.SiftPipeline(use, h, w, depth=1, max_keypoints=1 << 12)
            got = [pipe.detect(frame).result()[0] for _ in range(3)]
            assert pipe.replaying and len(got[0]) > 0
            assert all(np.array_equal(g["x"], got[0]["x"]) and np.array_equal(g["octave"], got[0]["octave"])
                       for g in got[1:])
            pipes.append((pipe, frame))
            first.append(got[0])
        pipe, frame = pipes[0]                      # evicted by the ninth pipeline's graph
        again = [pipe.detect(frame).result()[0] for _ in range(3)]
        assert pipe.replaying
        assert all(np.array_equal(g["x"], first[0]["x"]) for g in again)
    torch.cuda.synchronize()


def test_exhaustive_two_nearest_neighbour_matching(eng):
    """flann_matching (features.py:222-232) as an exact search: every match the ratio
    test keeps equals the brute-force answer; the same frame matched against a shifted
    copy of itself finds the shift."""
    from pano360_amd import features, synth
    rng = np.random.default_rng(9)
    a = rng.random((700, 128)).astype(np.float32)
    b = np.concatenate([a[rng.permutation(700)[:400]] + 0.01 * rng.random((400, 128)).astype(np.float32),
                        rng.random((900, 128)).astype(np.float32)])
    got = features.flann_matching(a, b)
    d = np.sqrt(((a[:, None, :].astype(np.float64) - b[None, :, :]) ** 2).sum(-1))
    order = np.argsort(d, axis=1)[:, :2]
    best, second = d[np.arange(700), order[:, 0]], d[np.arange(700), order[:, 1]]
    want = {int(q): int(order[q, 0]) for q in np.nonzero(best < 0.7 * second)[0]}
    assert {m.queryIdx: m.trainIdx for m in got} == want and len(want) >= 390
    assert max(abs(m.distance - best[m.queryIdx]) for m in got) < 1e-4
    img = synth.make_frame(5, 320, 200, "B")
    det = features.sift_detector()
    (k1, d1), (k2, d2) = det(img), det(np.roll(img, 7, axis=1))
    good = features.flann_matching(d1, d2)
    shift = np.array([k2[m.trainIdx].pt[0] - k1[m.queryIdx].pt[0] for m in good])
    assert len(good) > 50 and np.median(np.abs(shift - 7)) < 0.05


def test_full_size_properties_1080p(eng):
    """BASELINE config 2 at full size (8 x 1080p, native resolution), checked
    through properties that need no oracle: a constant-colour scene must come
    back constant wherever it is valid (the band-pass stack telescopes and the
    weights normalise), blending is idempotent, and the no-blend / linear /
    multiband mosaics agree on which pixels are covered."""
    from pano360_amd import engine, synth
    rots, intrs = synth.make_cameras(8, 1920, 1080, sweep_deg=140.0)
    imgs = [np.full((1080, 1920, 3), (40, 120, 200), np.uint8) for _ in range(8)]
    plan = engine.Plan([im.shape[:2] for im in imgs], rots, intrs, True, 10 ** 9)
    frames = eng.upload_frames(imgs)
    mosaic, fl, valid, _ = eng.stitch(frames, plan, "multiband", 5, want_float=True)
    again, _, _, _ = eng.stitch(frames, plan, "multiband", 5)
    import torch
    assert torch.equal(mosaic, again)
    v = valid.bool()
    want = torch.tensor([40, 120, 200], device=eng.device, dtype=torch.float32) / 255
    err = (fl[v] - want).abs().max().item()
    assert err <= 2e-6, err
    assert (mosaic[~v] == 0).all()
    inside = mosaic[v].int()
    target = torch.tensor([40, 120, 200], device=eng.device, dtype=torch.int32)
    assert ((inside - target).abs() <= 1).all()
    # the paste blender covers exactly the valid area of the unpadded plan
    plan_l = eng.upload_plan(engine.Plan([im.shape[:2] for im in imgs], rots, intrs, False,
                                         10 ** 9))
    non, _, valid_f, _ = eng.stitch(frames, plan_l, "none")
    non_staged, _, _, patches = eng.stitch(frames, plan_l, "none", fused=False)
    _, valid_l = eng.ownership(engine.patch_table(patches, eng), plan_l.shape)
    assert torch.equal(non, non_staged) and torch.equal(valid_f, valid_l)
    assert torch.equal(non.any(-1), valid_l.bool())
    lin, _, _, _ = eng.stitch(frames, plan_l, "linear")
    lin_staged, _, _, _ = eng.stitch(frames, plan_l, "linear", fused=False)
    assert torch.equal(lin, lin_staged)


def test_full_size_properties_32x4k(eng):
    """BASELINE config 3 at full size (32 x 4K, native resolution: the bench workload),
    through properties that need no oracle: the eight column strips of a multi-GPU run
    compose the single-GPU mosaic bit for bit; the optional two-stream schedule (interior
    pixels and the blur's work list on a side stream) equals the single-stream one bit for
    bit; the interior shortcut moves no pixel by more than one level; a constant scene
    comes back constant."""
    import torch
    from pano360_amd import dist as pdist
    from pano360_amd import engine, synth
    cfg = synth.CONFIGS["cfg3"]
    n, w, h = cfg["n"], cfg["width"], cfg["height"]
    rots, intrs = synth.make_cameras(n, w, h, sweep_deg=cfg["sweep_deg"])
    rng = np.random.default_rng(3)
    tile = rng.integers(0, 256, (240, 256, 3), dtype=np.uint8)
    imgs = [np.roll(np.tile(tile, (h // 240, w // 256, 1)), 17 * i, axis=1) for i in range(n)]
    shapes = [(h, w)] * n
    frames = eng.upload_frames(imgs)
    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    assert plan.shape == (2474, 13760)
    whole, fl, valid, _ = eng.stitch(frames, plan, "multiband", 5, want_float=True)
    # two streams: interior collapse and the blur's work list on the side stream
    eng.overlap_interior = eng.overlap_prepare = True
    try:
        two, _, _, _ = eng.stitch(frames, engine.Plan(shapes, rots, intrs, True, 10 ** 9),
                                  "multiband", 5)
    finally:
        eng.overlap_interior = eng.overlap_prepare = False
    assert torch.equal(whole, two)
    # column strips of 8 ranks, run one after the other on this GPU
    strips, bounds = pdist.emulate_on_one_device(eng, imgs, rots, intrs, 5, 8)
    assert bounds[-1] == plan.shape[1] and torch.equal(strips, whole)
    # the interior shortcut against the full blend
    full, fl_full, _, _ = eng.stitch(frames, engine.Plan(shapes, rots, intrs, True, 10 ** 9),
                                     "multiband", 5, want_float=True, shortcut=False)
    assert (whole.int() - full.int()).abs().max().item() <= 1
    assert (fl - fl_full).abs().max().item() <= 1e-6
    del strips, full, fl_full, two
    # a constant scene telescopes back to the constant
    const = eng.upload_frames([np.full((h, w, 3), (200, 90, 30), np.uint8)] * 2)
    flat, fl2, valid2, _ = eng.stitch(
        [const[i % 2] for i in range(n)], engine.Plan(shapes, rots, intrs, True, 10 ** 9),
        "multiband", 5, want_float=True)
    want = torch.tensor([200, 90, 30], device=eng.device, dtype=torch.float32) / 255
    v = valid2.bool()
    assert (fl2[v] - want).abs().max().item() <= 2e-6 and (flat[~v] == 0).all()
    assert torch.equal(valid2, valid)


# ------------------------------------------------------------------ laplacian / ingest
@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_laplacian_blending_matches_reference_golden(eng, case):
    """blend.laplacian_blending on the GPU against the mosaic the reference produced."""
    from pano360_amd import blend
    g = load_golden("laplacian")
    mask = g[f"{case}_mask"] if f"{case}_mask" in g else None
    out = blend.laplacian_blending(g[f"{case}_img1"], g[f"{case}_img2"], mask,
                                   int(g[f"{case}_levels"]))
    assert out.dtype == np.uint8 and np.array_equal(out, g[f"{case}_blended"])


def test_laplacian_pyramid_levels_bit_exact(eng):
    """Every pyramid primitive against the oracle, float32 and float64, odd sizes,
    1 to 4 channels; then a 1080p blend end to end."""
    import torch
    import cv2_shim as cv
    import laplacian_oracle as lo
    from pano360_amd import blend, synth
    pyr = blend._Pyr(eng)
    rng = np.random.default_rng(5)
    for (h, w, c), dtype in [((37, 53, 3), np.float32), ((64, 64, 1), np.float64),
                             ((4, 3, 4), np.float32), ((9, 130, 2), np.float64)]:
        a = (rng.random((h, w, c)) * 255).astype(dtype)
        dev = torch.from_numpy(a).to(eng.device)
        down = pyr.down(dev)
        ref_down = cv.pyrDown(a)
        assert np.array_equal(down.cpu().numpy(), ref_down), (h, w, c, dtype)
        up = pyr.up(down, dev, 0).cpu().numpy()
        assert np.array_equal(up, cv.pyrUp(ref_down)[:h, :w]), (h, w, c, dtype)
        lap = pyr.up(down, dev, 1).cpu().numpy()
        assert np.array_equal(lap, a - cv.pyrUp(ref_down)[:h, :w])
    img1 = synth.make_frame(1, 1920, 1080, "A")
    img2 = synth.make_frame(2, 1920, 1080, "A")
    assert np.array_equal(blend.laplacian_blending(img1, img2),
                          lo.laplacian_blending(img1, img2))


def test_laplacian_blending_keeps_a_float32_mask_float32(eng):
    """NumPy keeps the per-level mix and the collapse in float32 when the mask is float32
    (blend.py:134-140: float32 * float32, 1.0 - float32): bit-exact against the oracle, and
    different from what the same mask gives as float64."""
    import laplacian_oracle as lo
    from pano360_amd import blend, synth
    img1, img2 = synth.make_frame(7, 203, 130, "B"), synth.make_frame(8, 203, 130, "A")
    yy, xx = np.mgrid[0:130, 0:203]
    mask64 = (0.5 + 0.5 * np.sin(xx / 17.0) * np.cos(yy / 11.0))[..., None]
    mask32 = mask64.astype(np.float32)
    got32 = blend.laplacian_blending(img1, img2, mask32, 5)
    assert np.array_equal(got32, lo.laplacian_blending(img1, img2, mask32, 5))
    got64 = blend.laplacian_blending(img1, img2, mask32.astype(np.float64), 5)
    assert np.array_equal(got64, lo.laplacian_blending(img1, img2, mask32.astype(np.float64), 5))


def test_laplacian_blending_argument_errors(eng):
    from pano360_amd import blend
    img = np.zeros((8, 8, 3), np.uint8)
    with pytest.raises(ValueError):
        blend.laplacian_blending(img, np.zeros((8, 9, 3), np.uint8))
    with pytest.raises(ValueError):
        blend.laplacian_blending(img, img, n_levels=6)            # 8 px cannot take 6 levels
    with pytest.raises(NotImplementedError):
        blend.laplacian_blending(img, img, np.ones((8, 8, 1), np.uint8), n_levels=1)


@pytest.mark.parametrize("shrink", [2, 4, 3, 2.5, 1.5])
def test_shrink_matches_resize_oracle(eng, shrink):
    """The CLI's cv2.resize(im, None, fx=1/shrink, fy=1/shrink) (stitcher.py:419-420):
    bit-exact against the restated 8-bit path, even and odd sizes."""
    import laplacian_oracle as lo
    from pano360_amd import blend
    rng = np.random.default_rng(int(shrink * 10))
    for h, w in ((270, 480), (135, 241), (64, 96)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = blend.shrink_images([img], shrink)[0].cpu().numpy()
        assert np.array_equal(got, lo.shrink(img, shrink)), (h, w, shrink)
    assert blend.shrink_images([img], 1)[0].shape == img.shape


@pytest.mark.parametrize("scene", ["sweep", "jitter", "closed", "tall", "low", "tiny"])
def test_lean_blur_kernel_equals_the_general_kernel(scene):
    """``blur_lean_kernel`` (option PANO_OPT_BLUR_LEAN, default on) takes the work items whose
    bands need no special case and runs the same products in the same order as the general
    kernel: every blurred plane it writes, the float mosaic and the uint8 mosaic are the
    general kernel's bit for bit - interior shortcut on and off, L = 5 and 3, strips (vertical
    segments of the items) included; L = 6 puts five Gaussian levels into ONE launch with the two
    lightest on one wave pair (against the general kernel's 2 + 2 + 1 split launches), L = 2 leaves
    three wave pairs without a level.  "low" and "tiny": patches under 128 rows and under
    64 columns, whose band rows and columns reflect more than once (the lean kernel's
    element-wise form takes them too, as it does every item on a patch's edge)."""
    import torch
    from pano360_amd import _lib, engine, synth
    if scene == "low":
        imgs, rots, intrs = synth.make_scene(5, 400, 90, sweep_deg=80.0, jitter=0.01, seed=5, kind="A")
    elif scene == "tiny":
        imgs, rots, intrs = synth.make_scene(4, 50, 38, sweep_deg=40.0, seed=6, kind="B")
    elif scene == "sweep":
        imgs, rots, intrs = synth.make_scene(6, 640, 360, sweep_deg=100.0, seed=1, kind="A")
    elif scene == "jitter":
        imgs, rots, intrs = synth.make_scene(7, 480, 270, sweep_deg=120.0, jitter=0.02, seed=2, kind="B")
    elif scene == "closed":
        imgs, rots, intrs = synth.make_scene(12, 320, 240, step_deg=30.0, seed=3, kind="A")
    else:
        imgs, rots, intrs = synth.make_scene(4, 300, 700, sweep_deg=50.0, jitter=0.01, seed=4, kind="B")
    shapes = [im.shape[:2] for im in imgs]
    eng = engine.Engine()
    frames = eng.upload_frames(imgs)
    for levels, shortcut, strip in ((5, True, None), (5, False, None), (3, True, None), (5, True, 0.4),
                                    (6, True, None), (6, False, 0.3), (2, True, None)):
        got = []
        for lean in (1, 1, 0):              # (the first run only makes the arenas exist)
            eng.set_option(_lib.OPT_BLUR_LEAN, lean)
            plan = eng.upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9))
            W = plan.shape[1]
            cols = None if strip is None else (int(W * strip), int(W * (strip + 0.3)))
            for name in ("blurred", "planes"):
                if name in eng._arenas and eng._arenas[name] is not None:
                    eng._arenas[name].zero_()
            mosaic, fl, valid, patches = eng.multiband_fused(frames, plan, levels, want_float=True,
                                                            shortcut=shortcut, strip=cols)
            c0, c1 = cols if cols else (0, W)
            got.append((mosaic[:, c0:c1].cpu(), fl[:, c0:c1].cpu(), eng._arenas["blurred"].clone().cpu()))
        got = got[1:]
        assert torch.equal(got[0][0], got[1][0]), (scene, levels, shortcut, strip)
        assert torch.equal(got[0][1], got[1][1]), (scene, levels, shortcut, strip)
        assert got[0][2].numel() == got[1][2].numel()
        assert torch.equal(got[0][2].view(torch.int32), got[1][2].view(torch.int32))
    eng.set_option(_lib.OPT_BLUR_LEAN, 1)


def test_native_stitch_equals_the_launch_by_launch_path(oracle):
    """``pano_stitch_multiband`` (one native call per stitch: the default) against the same
    entry points called one by one from Python: same mosaic, float mosaic, valid map and
    records, bit for bit - whole mosaic, column strips (frames resident where needed only),
    a first call whose arenas must grow (the EGROW / resume round trip) and a missing frame."""
    import torch
    from pano360_amd import _lib, engine, synth
    imgs, rots, intrs = synth.make_scene(6, 320, 180, sweep_deg=100.0, jitter=0.01, seed=3, kind="B")
    shapes = [im.shape[:2] for im in imgs]
    native, plain = engine.Engine(), engine.Engine()
    plain.native_stitch = False
    assert native.native_stitch
    for levels in (5, 3, 1):
        for shortcut in (True, False):
            out = []
            for e in (native, plain):
                plan = e.upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9))
                mosaic, fl, valid, patches = e.multiband_fused(e.upload_frames(imgs), plan, levels,
                                                               want_float=True, shortcut=shortcut)
                out.append((mosaic.cpu(), fl.cpu(), valid.cpu(), patches.table.host.copy()))
            assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
            assert torch.equal(out[0][2], out[1][2])
            for key in ("y0", "x0", "vy0", "vx0", "vh", "vw", "ay0", "ax0", "ah", "aw", "index",
                        "tiles_off", "vpitch", "apitch"):
                assert np.array_equal(out[0][3][key], out[1][3][key]), key
    whole = out[0][0]
    plan = native.upload_plan(engine.Plan(shapes, rots, intrs, True, 10 ** 9))
    W = plan.shape[1]
    target = torch.zeros(plan.shape + (3,), dtype=torch.uint8, device=native.device)
    from pano360_amd import dist as pdist
    for rank in range(3):
        st = pdist.ShardedStitcher(native, shapes, rots, intrs, 1, rank, 3, exchange=None)
        frames = native.upload_frames([imgs[i] for i in st.my_frames])
        native.multiband_fused(frames, plan, 1, frame_ids=st.my_frames, strip=st.strip,
                               mosaic_out=target)
    assert torch.equal(target.cpu(), whole)
    with pytest.raises(_lib.PanoError, match="not resident"):
        native.multiband_fused(native.upload_frames(imgs[:2]), plan, 5, frame_ids=[0, 1])


@pytest.mark.parametrize("mode", [1, 2])
def test_side_stream_modes_give_the_same_mosaic(eng, mode):
    """``Engine(side_stream=...)``: 2 = the blur's tile flags and work list are made on a second
    stream beside the warp, 1 = also the interior pixels of the collapse.  Same mosaic bit for
    bit as on one stream, stitch after stitch (the streams are ordered by events only)."""
    import torch
    from pano360_amd import engine, synth
    imgs, rots, intrs = synth.make_scene(10, 640, 360, sweep_deg=120.0, jitter=0.01, seed=77, kind="A")
    shapes = [im.shape[:2] for im in imgs]
    frames = eng.upload_frames(imgs)
    want = eng.stitch(frames, engine.Plan(shapes, rots, intrs, True, 10 ** 9), "multiband", 5)[0]
    other = engine.Engine(side_stream=mode)
    for _ in range(12):
        got = other.stitch(frames, engine.Plan(shapes, rots, intrs, True, 10 ** 9), "multiband", 5)[0]
        assert torch.equal(got, want)


@pytest.mark.parametrize("step_deg", [0.9, 0.02])
def test_three_hundred_cameras(eng, oracle, step_deg):
    """More records than the wave-wide record test of the collapse holds in its masks (256) and
    more cameras than fit one 64-record ballot: 300 small frames, 0.9 degrees apart (every pixel
    under ~60 of them) or 0.02 degrees apart (every pixel under nearly all 300: more than the
    256 a tile's camera list holds, so every tile walks all cameras), against the oracle -
    multiband within one level, valid and the fused linear / none blends bit-exact."""
    from pano360_amd import engine, synth
    n, w, h = 300, 96, 64
    imgs, rots, intrs = synth.make_scene(n, w, h, step_deg=step_deg, jitter=0.002, seed=300,
                                         kind="B")
    shapes = [(h, w)] * n
    frames = eng.upload_frames(imgs)
    plan = engine.Plan(shapes, rots, intrs, True, 10 ** 9)
    mosaic, fl, valid, patches = eng.stitch(frames, plan, "multiband", 3, want_float=True)
    assert len(patches) > 256 or step_deg < 0.1
    ref_u8, ref_f = oracle.stitch(imgs, rots, intrs, "multiband", 3, max_resolution=10 ** 9,
                                  return_float=True)
    assert rel_l2(fl.cpu().numpy(), ref_f) <= REL_TOL
    assert np.abs(mosaic.cpu().numpy().astype(int) - ref_u8.astype(int)).max() <= 1
    _, ref_patches, _ = oracle.warp_all(imgs, rots, intrs, True, 10 ** 9)
    assert np.array_equal(valid.cpu().numpy().astype(bool), oracle.valid(ref_patches, plan.shape))
    plan_l = engine.Plan(shapes, rots, intrs, False, 10 ** 9)
    for kind in ("linear", "none"):
        got = eng.stitch(frames, plan_l, kind)[0]
        assert np.array_equal(got.cpu().numpy(),
                              oracle.stitch(imgs, rots, intrs, kind, max_resolution=10 ** 9)), kind


def _edge_scene(case):
    from pano360_amd import synth
    if case in ("one camera", "two cameras"):
        imgs, rots, intrs = synth.make_scene(3, 160, 90, sweep_deg=40.0, seed=1, kind="B")
        k = 1 if case == "one camera" else 2
        return imgs[:k], rots[:k], intrs[:k], 5, 10 ** 9
    if case == "gaps between the frames":
        return synth.make_scene(3, 120, 80, sweep_deg=170.0, seed=2, kind="B") + (5, 10 ** 9)
    if case == "frames of 16 x 12":
        return synth.make_scene(4, 16, 12, sweep_deg=30.0, seed=3, kind="A") + (2, 10 ** 9)
    imgs, rots, intrs = synth.make_scene(4, 200, 120, sweep_deg=60.0, seed=4, kind="B")
    if case == "mixed frame sizes":
        imgs = [imgs[0], imgs[1][:100, :150].copy(), imgs[2], imgs[3][:90, :180].copy()]
        return imgs, rots, intrs, 5, 10 ** 9
    if case == "mosaic capped at 37 pixels":
        return imgs, rots, intrs, 2, 37
    assert case == "strong roll and pitch"
    return synth.make_scene(5, 200, 120, sweep_deg=60.0, jitter=0.08, seed=5, kind="B") + (5, 10 ** 9)


@pytest.mark.parametrize("case", ["one camera", "two cameras", "gaps between the frames",
                                  "frames of 16 x 12", "mixed frame sizes",
                                  "mosaic capped at 37 pixels", "strong roll and pitch"])
def test_edge_case_scenes(eng, oracle, case):
    """Degenerate and awkward inputs through the fused paths against the oracle: multiband within
    one level, linear / none / valid / crop rectangle bit-exact."""
    from pano360_amd import engine
    imgs, rots, intrs, levels, mr = _edge_scene(case)
    shapes = [im.shape[:2] for im in imgs]
    frames = eng.upload_frames(imgs)
    plan = engine.Plan(shapes, rots, intrs, True, mr)
    mosaic, _, valid, _ = eng.stitch(frames, plan, "multiband", levels)
    ref = oracle.stitch(imgs, rots, intrs, "multiband", levels, max_resolution=mr)
    assert np.abs(mosaic.cpu().numpy().astype(int) - ref.astype(int)).max() <= 1
    _, ref_patches, _ = oracle.warp_all(imgs, rots, intrs, True, mr)
    ref_valid = oracle.valid(ref_patches, plan.shape)
    assert np.array_equal(valid.cpu().numpy().astype(bool), ref_valid)
    assert eng.crop_rect(valid) == oracle.crop_rect(ref_valid)
    plan_l = engine.Plan(shapes, rots, intrs, False, mr)
    for kind in ("linear", "none"):
        got = eng.stitch(frames, plan_l, kind)[0].cpu().numpy()
        assert np.array_equal(got, oracle.stitch(imgs, rots, intrs, kind, max_resolution=mr)), kind


def test_owned_regions_on_a_mosaic_wider_than_one_span_chunk(eng):
    """pano_owned_regions on a 70 000-column owner map: owned_spans_kernel ballots the column
    marks 65 536 columns at a time (its LDS words), so a wider mosaic takes a second chunk with
    the run bookkeeping carried across - and init_regions_kernel clears 4 x 70 000 marks."""
    import torch
    H, W, n = 6, 70000, 4
    own = np.full((H, W), -1, np.int16)
    own[:, 10:300] = 0
    own[1:4, 65000:65530] = 1
    own[2:5, 65530:66000] = 2          # camera 2's run crosses column 65 536
    own[0:2, 69990:70000] = 3
    own[3, 40000] = 3                  # a second, far span of camera 3
    owner = torch.from_numpy(own).to(eng.device)
    boxes, spans = eng.owned_regions(owner, n, min_gap=5, max_spans=4)
    for i in range(n):
        ys, xs = np.nonzero(own == i)
        assert tuple(boxes[i]) == (ys.min(), ys.max(), xs.min(), xs.max())
    assert [tuple(s) for s in spans[0]] == [(10, 299)]
    assert [tuple(s) for s in spans[1]] == [(65000, 65529)]
    assert [tuple(s) for s in spans[2]] == [(65530, 65999)]
    assert [tuple(s) for s in spans[3]] == [(40000, 40000), (69990, 69999)]
