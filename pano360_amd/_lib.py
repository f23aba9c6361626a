"""ctypes binding of ``libpano360_hip.so`` (declared in ``include/pano360.h``).

There is no CPU fallback: if the library is missing or a call fails the caller
gets an exception.  ``torch`` is imported first so that the HIP runtime torch
ships is the one both sides use (the library is linked against the SONAME
``libamdhip64.so.7``, which the loader then resolves to the already loaded
copy); torch itself is only used for device memory, streams and
``torch.distributed``.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# PANO_LIB: another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("PANO_LIB") or os.path.join(_HERE, "libpano360_hip.so")

MAX_TAPS = 129
MAX_LEVELS = 8
TAP_LEAD = 7
TAP_PAD = 40
# pano_ctx options (include/pano360.h)
OPT_BLUR_KERNEL, OPT_OWN_PRUNE, OPT_BLUR_SEGMENTS, OPT_BLUR_LEAN, OPT_STITCH_STREAMS = 0, 1, 2, 3, 4
OPT_STITCH_ASYNC = 5
OPT_BLUR_SEG_LEN = 6
OPT_SIFT_GRAPH = 7
OPT_LEVEL_CLASSES = 8
BLUR_MFMA, BLUR_VALU = 0, 1


class PanoError(RuntimeError):
    """A C-ABI call returned a negative status."""


class Patch(C.Structure):
    """``pano_patch`` of include/pano360.h (96 bytes)."""
    _fields_ = [("planes", C.c_void_p), ("mask", C.c_void_p),
                ("blurred", C.c_void_p), ("scratch", C.c_void_p),
                ("y0", C.c_int32), ("x0", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("vy0", C.c_int32), ("vx0", C.c_int32), ("vh", C.c_int32), ("vw", C.c_int32),
                ("ay0", C.c_int32), ("ax0", C.c_int32), ("ah", C.c_int32), ("aw", C.c_int32),
                ("vpitch", C.c_int32), ("apitch", C.c_int32),
                ("index", C.c_int32), ("tiles_off", C.c_int32)]


class Layout(C.Structure):
    """``pano_layout`` of include/pano360.h."""
    _fields_ = [("planes_floats", C.c_int64), ("blurred_floats", C.c_int64),
                ("scratch_floats", C.c_int64), ("n_records", C.c_int32), ("n_tiles", C.c_int32),
                ("max_vw", C.c_int32), ("max_vh", C.c_int32), ("max_aw", C.c_int32),
                ("max_ah", C.c_int32), ("missing", C.c_int32)]


class SiftArgs(C.Structure):
    """``pano_sift_args`` of include/pano360.h."""
    _fields_ = [("frame", C.c_void_p), ("frame_copy", C.c_void_p),
                ("h", C.c_int32), ("w", C.c_int32), ("n_octaves", C.c_int32), ("n_layers", C.c_int32),
                ("taps", C.c_void_p), ("ntaps", C.c_void_p), ("gauss", C.c_void_p),
                ("dog", C.c_void_p), ("work", C.c_void_p), ("detect", C.c_int32),
                ("contrast_thr", C.c_float), ("edge_thr", C.c_float), ("sigma", C.c_float),
                ("first_octave", C.c_int32), ("max_keypoints", C.c_int32),
                ("gauss_dev", C.c_void_p), ("dims_dev", C.c_void_p), ("cands", C.c_void_p),
                ("kpts", C.c_void_p), ("counts", C.c_void_p), ("sort_work", C.c_void_p),
                ("desc", C.c_void_p)]


class StitchArgs(C.Structure):
    """``pano_stitch_args`` of include/pano360.h."""
    _fields_ = ([(k, C.c_void_p) for k in (
        "cams", "rects", "have", "sin_t", "cos_t", "tan_p", "lut", "taps", "ntaps", "owner",
        "valid", "marks", "regions", "regions_host", "block_owner", "interior", "records_host",
        "table", "planes", "blurred", "scratch", "tile_flags", "need", "mosaic", "mosaic_f32")]
        + [(k, C.c_int64) for k in ("planes_floats", "blurred_floats", "scratch_floats")]
        + [(k, C.c_int32) for k in (
            "n", "H", "W", "xs0", "xs1", "own0", "own1", "lut_stride", "n_levels", "radius",
            "shortcut", "warp_need", "max_spans", "min_gap", "cap_records", "cap_tiles",
            "used_need", "trust_layout")]
        + [("classes", C.c_void_p), ("layout", Layout)])


EINVAL = -1     # PANO_EINVAL
EGROW = 1       # pano_stitch_multiband: an arena is too small, args.layout says what is needed


class Pair(C.Structure):
    """``pano_pair`` of include/pano360.h (80 bytes)."""
    _fields_ = [("minv", C.c_double * 9), ("i", C.c_int32), ("j", C.c_int32)]


class SiftKeypoint(C.Structure):
    """``pano_sift_keypoint`` of include/pano360.h (32 bytes)."""
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("r", C.c_int32), ("c", C.c_int32)]


class Camera(C.Structure):
    """``pano_camera`` of include/pano360.h (120 bytes)."""
    _fields_ = [("proj", C.c_double * 9), ("frame", C.c_void_p),
                ("hat_x", C.c_void_p), ("hat_y", C.c_void_p),
                ("sh", C.c_int32), ("sw", C.c_int32),
                ("y0", C.c_int32), ("x0", C.c_int32), ("h", C.c_int32), ("w", C.c_int32)]


_vp, _i = C.c_void_p, C.c_int
_SIGNATURES = {
    "pano_version": (C.c_char_p, []),
    "pano_last_error": (C.c_char_p, []),
    "pano_device_count": (_i, []),
    "pano_pitch": (_i, [_i]),
    "pano_interior_block": (_i, []),
    "pano_ctx_create": (_i, [_i, _vp, C.POINTER(C.c_void_p)]),
    "pano_ctx_destroy": (_i, [_vp]),
    "pano_ctx_set_stream": (_i, [_vp, _vp]),
    "pano_ctx_set_option": (_i, [_vp, _i, _i]),
    "pano_ctx_get_option": (_i, [_vp, _i, C.POINTER(C.c_int)]),
    "pano_timing_enable": (_i, [_vp, _i]),
    "pano_kernel_count": (_i, []),
    "pano_kernel_name": (C.c_char_p, [_i]),
    "pano_timing_read": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "pano_add_weights": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "pano_warp_spherical": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i,
                                 _i, _vp, _vp, _vp, _vp]),
    "pano_warp_windows": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "pano_blur_tiles": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pano_ownership": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pano_ownership_cameras": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "pano_owned_regions": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "pano_ownership_regions": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                    _i, _i, _vp, _vp]),
    "pano_multiband_blur": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, C.POINTER(C.c_int), _i,
                                 _vp, _vp]),
    "pano_multiband_blur_prepare": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "pano_layout_windows": (_i, [_i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "pano_layout_place": (_i, [_vp, _i, _vp, _vp, _vp]),
    "pano_interior_map": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "pano_interior_classes": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "pano_multiband_compose": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _i, _vp, _vp, _i]),
    "pano_blend_cameras": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp,
                                _vp]),
    "pano_blur_tile_grid": (_i, [_vp]),
    "pano_overlap_blocks": (_i, [_i, _i]),
    "pano_overlap_stats": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pano_linear_blend": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "pano_no_blend": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "pano_crop_rect": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "pano_blur_plane": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i]),
    "pano_pyr_down": (_i, [_vp, _vp, _i, _i, _vp]),
    "pano_gray_u8": (_i, [_vp, _vp, _i, _i, _vp]),
    "pano_scale_step": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "pano_scale_space": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "pano_resize_up2": (_i, [_vp, _vp, _i, _i, _vp]),
    "pano_decimate2": (_i, [_vp, _vp, _i, _i, _vp]),
    "pano_subtract": (_i, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "pano_pyr_down_image": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "pano_pyr_up_image": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _i]),
    "pano_u8_to_f32": (_i, [_vp, _vp, C.c_size_t, _vp]),
    "pano_laplacian_mix": (_i, [_vp, _vp, _vp, _vp, C.c_size_t, _i, _vp]),
    "pano_clip_u8": (_i, [_vp, _vp, C.c_size_t, _i, _vp]),
    "pano_resize_u8": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i]),
    "pano_sift_extrema": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_float, C.c_float, C.c_float, _vp,
                               _vp, _i]),
    "pano_sift_orient": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i]),
    "pano_sift_describe": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "pano_sift_sort_work_bytes": (C.c_size_t, [_i]),
    "pano_sift_sort_unique": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "pano_knn2_work_bytes": (C.c_size_t, [_i, _i, _i]),
    "pano_knn2": (_i, [_vp, _vp, _i, _vp, _i, _i, C.c_float, _vp, _vp, _vp, _vp]),
    "pano_sift_detect": (_i, [_vp, _vp]),
    "pano_sift_detect_replaying": (_i, [_vp]),
    "pano_stitch_multiband": (_i, [_vp, _vp, _i]),
    "pano_stitch_counts": (_i, [_vp, _vp, _vp]),
    "pano_stitch_verify": (_i, [_vp]),
}
EXPORTS = tuple(_SIGNATURES)


def sources_digest():
    """sha256 over everything the library is compiled from (csrc/*.hip, *.h, *.inc, the Makefile,
    include/pano360.h), names included.  A package installed without ../include hashes the rest."""
    import hashlib
    src = os.path.join(_HERE, "csrc")
    files = sorted(os.path.join(src, f) for f in os.listdir(src)
                   if f.endswith((".hip", ".h", ".inc")) or f == "Makefile")
    files.append(os.path.join(os.path.dirname(_HERE), "include", "pano360.h"))
    h = hashlib.sha256()
    for path in files:
        if not os.path.exists(path):
            continue
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as fid:
            h.update(fid.read())
    return h.hexdigest()


BUILD_RECORD = os.path.join(_HERE, "csrc", ".build_stamp")


def build(force=False):
    """Compile the HIP sources in-tree (hipcc cross-compiles without a GPU).  `make` alone trusts
    file times, which a checkout or a copied tree does not keep: the objects are therefore
    stamped with the digest of the sources they were built from, and a tree whose stamp is missing
    or differs is rebuilt from scratch (`make -B`).  The stamp says what happened
    (`csrc/.build_stamp`: digest, mode "full" | "incremental" | "up to date")."""
    import json
    src = os.path.join(_HERE, "csrc")
    digest = sources_digest()
    have = None
    try:
        with open(BUILD_RECORD) as fid:
            have = json.load(fid).get("sources_sha256")
    except (OSError, ValueError):
        pass
    full = force or have != digest or not os.path.exists(LIB_PATH)
    before = os.path.getmtime(LIB_PATH) if os.path.exists(LIB_PATH) else None
    cmd = ["make", "-C", src, "-s", "-j4"] + (["-B"] if full else [])
    subprocess.check_call(cmd)
    after = os.path.getmtime(LIB_PATH)
    mode = "full" if full else ("incremental" if after != before else "up to date")
    with open(BUILD_RECORD, "w") as fid:
        json.dump({"sources_sha256": digest, "mode": mode}, fid)
    return LIB_PATH


_lib = None


def lib():
    """Load (once) and return the bound library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PanoError(
                f"{LIB_PATH} is missing: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        import torch  # noqa: F401  (pins the HIP runtime, see module docstring)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().pano_last_error().decode("utf-8", "replace")
        raise PanoError(f"{what} failed ({status}): {msg}")
