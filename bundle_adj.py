"""Module path ``bundle_adj`` for the camera record, so ``ba_<name>.pkl`` caches
written by the reference CLI unpickle here (reference bundle_adj.py:18-33)."""
from pano360_amd.bundle_adj import Image, intrinsics, rotation_to_mat  # noqa: F401
