"""Drop-in surface of the reference ``stitcher.py`` for the warp/blend/crop path.

Same names, argument order, defaults, return types and side effects as the
reference: ``stitch``, ``no_blend`` / ``linear_blend`` /
``multiband_blend`` (the ``blender(patches, shape)`` protocol), ``SphProj``,
``CylProj``, ``_proj_img_range_border``, ``_proj_img_range_corners``,
``estimate_resolution``, ``_hat``, ``_add_weights``, ``_valid``,
``crop_mosaic``, ``find_gains``, ``equalize_gains``, ``BLENDERS``,
``MAX_RESOLUTION`` and the CLI ``main``.
Per-pixel work goes to hand-written HIP kernels through ``_lib`` (ctypes over
``libpano360_hip.so``); there is no CPU fallback for it.

Out of scope here: feature matching and bundle adjustment - the
CLI therefore needs the ``ba_<name>.pkl`` camera cache the reference CLI writes
(stitcher.py:430-439).
"""
import argparse
import logging
import os
import pickle
import time

import numpy as np

from . import bundle_adj as _ba
from . import engine as _eng
from .engine import CylProj, SphProj  # noqa: F401  (re-exported API)

MAX_RESOLUTION = 1400       # read at call time, like the reference (stitcher.py:17,154)
# Multiband accuracy contract of ``stitch`` (read at call time; env PANO_EXACT=1 sets the
# default).  False: the fast path - interior pixels take the owner's colour directly (the
# band-pass stack telescopes to it in real arithmetic) and the Gaussian levels run on the
# matrix cores in split float16 - uint8 mosaic within ONE level of the reference wherever
# 255 v sits on an integer boundary (about 1 value in 1000), float mosaic within 1e-4
# relative L2.  True: every pixel through the full band sum and the float32 vector-ALU blur
# (one FMA per tap): the same bounds, but the only deviations left are float32 roundings
# of the blur (<= 1e-6 per plane).  Integer results (valid mask, crop) are exact either way.
EXACT = os.environ.get("PANO_EXACT", "0") not in ("", "0")
_exact_engine = None


def _engine_for_stitch():
    """The process-wide engine, or its exact-mode sibling (vector-ALU blur)."""
    global _exact_engine
    if not EXACT:
        return _eng.engine()
    if _exact_engine is None:
        _exact_engine = _eng.Engine(blur="valu")
    return _exact_engine


# ------------------------------------------------------------------ exposure
def find_gains(overlaps, sizes, stdn=0.1, stdg=2):
    """Find the gains minimizing discrepancies between mean intensities
    (stitcher.py:24-33)."""
    return _eng.find_gains(overlaps, sizes, stdn, stdg)


def _frames_of(regions):
    """uint8 frames behind ``reg.img``: either still uint8 (before
    ``_add_weights``) or the float32 RGBA image ``_add_weights`` made of one, whose
    colours are float32(u8)/255 and convert back exactly."""
    base = np.arange(256, dtype=np.float32) / np.float32(255)
    frames = []
    for reg in regions:
        img = reg.img
        if img.dtype != np.uint8:
            back = np.clip(np.rint(img[..., :3] * np.float32(255)), 0, 255).astype(np.uint8)
            if not np.array_equal(base[back], img[..., :3]):
                raise ValueError("equalize_gains works on frames that came from uint8 images "
                                 "(reg.img as _add_weights leaves it, stitcher.py:257-263)")
            img = back
        frames.append(np.ascontiguousarray(img[..., :3]))
    return frames


def equalize_gains(regions):
    """Equalize the exposures by minimizing differences on overlaps
    (stitcher.py:36-66).  Like the reference it rescales ``reg.img[..., :3]`` of
    float32 RGBA regions in place; the pair statistics run on the GPU
    (``pano_overlap_stats``).  Returns the gains (the reference returns None)."""
    eng = _eng.engine()
    frames = eng.upload_frames(_frames_of(regions))
    _, _, gains, _ = eng.equalize_gains(frames, [r.rot for r in regions],
                                        [r.intr for r in regions])
    for reg, gain in zip(regions, gains):
        if reg.img.dtype != np.uint8:
            reg.img[..., :3] = np.clip(gain * reg.img[..., :3], 0, 1)      # stitcher.py:66
    return gains


# ------------------------------------------------------------ host geometry
def _proj_img_range_border(shape, hom):
    """Extent of a projected frame from its border (stitcher.py:107-122)."""
    return _eng.range_from_border(shape, hom)


def _proj_img_range_corners(shape, hom):
    """Extent from the corners, with wrap-around check (stitcher.py:125-139)."""
    return _eng.range_from_corners(shape, hom)


def estimate_resolution(regions):
    """Resolution of the final image (stitcher.py:142-157)."""
    mid = regions[len(regions) // 2]
    return _eng.resolution_for([reg.range for reg in regions], mid.img.shape[:2],
                               mid.hom(), MAX_RESOLUTION)


def _hat(size):
    """Triangular function of a given size (stitcher.py:251-254)."""
    return _eng.hat(size)


def _add_weights(img):
    """uint8 RGB -> float32 RGBA/255 with alpha = hat(y)*hat(x)
    (stitcher.py:257-263); computed by ``pano_add_weights`` on the GPU."""
    eng = _eng.engine()
    frame = eng.upload_frames([img])[0]
    return eng.add_weights(frame).cpu().numpy()


# ------------------------------------------------------------ patch plumbing
def _upload_patches(eng, patches, n_blur):
    """Host patches of the blender protocol -> device patches (planar planes)."""
    import torch
    out = []
    for warped, mask, irange in patches:
        ys, xs = irange
        dp = _eng.DevicePatch((ys.start, ys.stop, xs.start, xs.stop), eng.device, n_blur)
        src = torch.from_numpy(np.ascontiguousarray(warped, np.float32)).to(eng.device)
        dp.planes[:, :, :dp.w] = src.permute(2, 0, 1)
        if dp.pitch != dp.w:
            dp.planes[:, :, dp.w:] = 0
        dp.mask.copy_(torch.from_numpy(np.ascontiguousarray(mask).astype(np.uint8)))
        out.append(dp)
    return out


def no_blend(patches, shape):
    """Paste the patches without blending (stitcher.py:160-168)."""
    eng = _eng.engine()
    dev = _upload_patches(eng, patches, 0)
    return eng.simple_blend(dev, tuple(shape), linear=False).cpu().numpy()


def linear_blend(patches, shape):
    """Linearly blend patches (stitcher.py:171-183)."""
    eng = _eng.engine()
    dev = _upload_patches(eng, patches, 0)
    return eng.simple_blend(dev, tuple(shape), linear=True).cpu().numpy()


def multiband_blend(patches, shape, n_levels=5):
    """Multi-band blending (stitcher.py:186-241).  As in the reference, each
    patch's alpha channel is overwritten in place with its sharp ownership
    mask (stitcher.py:207-208)."""
    eng = _eng.engine()
    dev = _upload_patches(eng, patches, n_levels - 1)
    mosaic, _, owner, _ = eng.multiband(dev, tuple(shape), n_levels)
    owner = owner.cpu().numpy()
    for idx, (warped, _, irange) in enumerate(patches):
        warped[..., 3] = owner[irange] == idx
    return mosaic.cpu().numpy()


BLENDERS = {
    "none": no_blend,
    "linear": linear_blend,
    "multiband": multiband_blend,
}
_FUSED = {no_blend: "none", linear_blend: "linear", multiband_blend: "multiband"}


def _valid(patches, shape):
    """Area of validity, OR of ~mask (stitcher.py:266-271)."""
    eng = _eng.engine()
    dev = _upload_patches(eng, patches, 0)
    table = _eng.patch_table(dev, eng)
    _, valid = eng.ownership(table, tuple(shape))
    return valid.cpu().numpy().astype(bool)


def _crop_rect(valid):
    import torch
    eng = _eng.engine()
    dev = torch.from_numpy(np.ascontiguousarray(valid).astype(np.uint8)).to(eng.device)
    rect = eng.crop_rect(dev)
    if rect is None:
        # the reference falls off the end of its scan with `last` unbound
        raise UnboundLocalError("local variable 'last' referenced before assignment")
    return rect


def crop_mosaic(mosaic, valid):
    """Remove the black borders: largest all-valid rectangle with the
    reference's scan-order tie-break; returns a view (stitcher.py:340-369)."""
    y0, x0, h, w = _crop_rect(valid)
    return mosaic[y0:y0 + h, x0:x0 + w, :]


# ------------------------------------------------------------------- stitch
def _download_patches(dev_patches):
    out = []
    for dp in dev_patches:
        y0, y1, x0, x1 = dp.rect
        warped = dp.planes[:, :, :dp.w].permute(1, 2, 0).contiguous().cpu().numpy()
        out.append((warped, dp.mask.cpu().numpy().astype(bool), np.s_[y0:y1, x0:x1]))
    return out


def stitch(regions, blender=no_blend, equalize=False, crop=False):
    """Stitch the images together (stitcher.py:274-327).

    ``regions``: list of ``bundle_adj.Image``.  Side effects kept from the
    reference: ``reg.range`` is filled in and ``reg.img`` is replaced by the
    float32 RGBA weighted image (stitcher.py:277-278).  A blender that is not
    one of this module's three is called with host patches, exactly as the
    reference would call it.

    Accuracy: none / linear and every integer result (valid mask, crop rectangle) equal
    the reference's bit for bit; multiband is within one uint8 level and 1e-4 relative L2
    (see ``EXACT`` above for the two modes).
    """
    eng = _engine_for_stitch()
    frames_host = [reg.img for reg in regions]
    padded = blender == multiband_blend                     # stitcher.py:295
    plan = _eng.Plan([im.shape[:2] for im in frames_host], [r.rot for r in regions],
                     [r.intr for r in regions], padded, MAX_RESOLUTION)
    frames = eng.upload_frames(frames_host)
    luts = None
    if equalize:                                            # stitcher.py:280-281
        logging.debug("Equalizing gain...")
        luts = eng.equalize_gains(frames, [r.rot for r in regions],
                                  [r.intr for r in regions])[3]
    for i, (reg, rng, frame) in enumerate(zip(regions, plan.ranges, frames)):
        reg.range = rng
        # stitcher.py:277-278 leaves _add_weights' float32 RGBA image in reg.img; our record
        # type fetches it from the device when it is first read (32 x 133 MB for config 3)
        rgba = (lambda f=frame, l=None if luts is None else luts[i]:
                eng.add_weights(f, l).cpu().numpy())
        reg.img = _ba.Deferred(rgba) if isinstance(reg, _ba.Image) else rgba()
    eng.upload_plan(plan)

    kind = _FUSED.get(blender)
    if kind is not None:
        n_levels = multiband_blend.__defaults__[0]
        mosaic, _, valid, patches = eng.stitch(frames, plan, kind, n_levels, luts=luts,
                                               shortcut=not EXACT)
    else:
        patches, _ = eng.warp_all(frames, plan, luts=luts)
        valid = None
        mosaic = blender(_download_patches(patches), plan.shape)
    if hasattr(mosaic, "cpu"):
        mosaic = mosaic.cpu().numpy()
    if crop:
        logging.debug("Cropping...")
        if valid is None:
            table = _eng.patch_table(patches, eng)
            _, valid = eng.ownership(table, plan.shape)
        rect = eng.crop_rect(valid)
        if rect is None:
            raise UnboundLocalError("local variable 'last' referenced before assignment")
        y0, x0, h, w = rect
        mosaic = mosaic[y0:y0 + h, x0:x0 + w, :]
    return mosaic


# ---------------------------------------------------------------------- CLI
IMAGE_EXTENSIONS = (".jpg", ".png", ".bmp", ".JPG", ".PNG", ".BMP")     # stitcher.py:411-412


def ingest(path, shrink):
    """The head of the reference's ``main`` (stitcher.py:415-421): every image of the
    directory, in ``os.listdir`` order, as ``cv2.imread`` would return it (uint8 BGR), shrunk by
    ``cv2.resize(im, None, fx=1/shrink, fy=1/shrink)`` when shrink > 1 - on the device.
    Returns uint8 [h][w][3] device tensors."""
    from PIL import Image as PilImage
    from PIL import ImageOps
    from . import blend as _blend
    files = [f for f in os.listdir(path) if any(f.endswith(ext) for ext in IMAGE_EXTENSIONS)]

    def read(name):
        # cv2.imread's defaults: the EXIF orientation applied (a phone's rotated JPEG arrives
        # upright, with its shape swapped), 8 bits, three channels: 16-bit images are scaled
        # down to 8 bits and an alpha channel is dropped, as IMREAD_COLOR does
        im = ImageOps.exif_transpose(PilImage.open(os.path.join(path, name)))
        if im.mode in ("I;16", "I;16B", "I;16L", "I"):
            im = PilImage.fromarray((np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8))
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])
    return _blend.shrink_images([read(f) for f in files], shrink)


def main(argv=None):
    """Same command line as the reference (stitcher.py:390-451)."""
    parser = argparse.ArgumentParser(description="Stitch images.")
    parser.add_argument("path", type=str, help="directory with the images to process.")
    parser.add_argument("-s", "--shrink", type=float, default=2,
                        help="downsample the images by this amount.")
    parser.add_argument("--ba", default="incr", choices=["none", "incr", "last"],
                        help="bundle adjustment type.")
    parser.add_argument("--equalize", "-e", action="store_true",
                        help="equalize image gain before stitching.")
    parser.add_argument("--crop", "-c", action="store_true", help="remove the black borders.")
    parser.add_argument("--blend", "-b", default="multiband", choices=list(BLENDERS.keys()),
                        help="blending algorithm.")
    parser.add_argument("-o", "--out", type=str, help="save result to this file")
    args = parser.parse_args(argv)

    name = f"{os.path.basename(os.path.normpath(args.path))}_s{args.shrink}"
    cache = f"ba_{name}.pkl"
    try:
        with open(cache, "rb") as fid:
            regions = pickle.load(fid)
    except IOError:
        regions = None
    # stitcher.py:415-421: list the directory (os.listdir order, the reference's extensions),
    # read, shrink.  The reference does this before it looks at its caches and, with a
    # ``ba_*.pkl`` present, never uses the result (the pickle carries the shrunk images); here
    # the images are read only when something consumes them: a cache without pixels
    # (``img=None`` records: cameras only, a few hundred bytes per frame) or no cache at all.
    # Decoding is Pillow's, on the host; the resize runs on the device (``pano_resize_u8``)
    # and the shrunk frames stay there for ``stitch``.
    if regions is None:
        # (before anything is decoded or uploaded)
        raise SystemExit(
            f"{cache} not found: feature matching and bundle adjustment are outside this "
            "build's scope; produce the camera cache with the reference (it is the pickle "
            "written at stitcher.py:438-439) and re-run")
    if any(reg.img is None for reg in regions):
        frames = ingest(args.path, args.shrink) if os.path.isdir(args.path) else []
        if len(frames) != len(regions):
            raise SystemExit(f"{cache} holds {len(regions)} cameras, {args.path} "
                             f"{len(frames)} images")
        for reg, frame in zip(regions, frames):
            if reg.img is None:
                reg.img = frame

    start = time.time()
    mosaic = stitch(regions, blender=BLENDERS[args.blend], equalize=args.equalize,
                    crop=args.crop)
    logging.info(f"Built mosaic, time: {time.time() - start}")
    if args.out:
        from PIL import Image as PilImage
        PilImage.fromarray(np.ascontiguousarray(mosaic[..., ::-1])).save(args.out)
    return mosaic


if __name__ == "__main__":
    logging.basicConfig(level=logging.DEBUG)
    main()
