"""Camera record consumed by the warp/blend hot path.

Only the *input type* of the path is mirrored here (reference
``bundle_adj.py:18-33`` ``Image``, ``:82-87`` ``intrinsics``, ``:96-101``
``rotation_to_mat``).  Feature matching and Levenberg-Marquardt bundle
adjustment are registration, outside the accelerated path, and
are not rebuilt.

The class is importable as ``bundle_adj.Image`` through the top-level
``bundle_adj.py`` re-export, so ``ba_<name>.pkl`` caches written by the
reference CLI (stitcher.py:430-439) unpickle into it unchanged.
"""
from dataclasses import dataclass, field

import numpy as np


def _zero_range():
    return (np.zeros(2), np.zeros(2))


class Deferred:
    """A pixel array that exists on the device and is copied to the host when first read
    (``stitch`` leaves the float32 RGBA image of ``_add_weights`` in ``reg.img``,
    stitcher.py:277-278: 133 MB per 4K frame that most callers never look at)."""

    def __init__(self, make):
        self.make = make


@dataclass(repr=False)
class Image:
    """One registered frame: pixels, rotation R, calibration K, angular range - a dataclass
    with the reference's four fields (``dataclasses.fields`` / ``asdict`` / ``replace`` and the
    keyword constructor work as they do on the reference's class).

    ``img``   uint8 [H, W, 3] on entry to ``stitch`` (channel order opaque); float32 RGBA
              afterwards, as in the reference.  ``stitch`` stores a ``Deferred`` there: the
              RGBA image is made on the device and downloaded when ``img`` is first READ.
              Until then the record keeps the device frame, the colour table and the engine
              alive; reading it (or assigning to it) releases them.  The read runs
              ``_add_weights`` on the engine's current stream at that moment.
    ``rot``   float64 3x3 world->camera rotation.
    ``intr``  float64 3x3 ``[[f,0,cx],[0,f,cy],[0,0,1]]``.
    ``range`` (min, max) spherical angles, filled in by ``stitch``.
    """

    img: np.ndarray
    rot: np.ndarray
    intr: np.ndarray
    range: tuple = field(default_factory=_zero_range)        # noqa: A003 - the reference's name

    # the reference's class pickles its plain attributes: keep that layout in both directions
    def __getstate__(self):
        return {"img": self.img, "rot": self.rot, "intr": self.intr, "range": self.range}

    def __setstate__(self, state):
        for key, value in state.items():
            setattr(self, key, value)

    def __repr__(self):
        return (f"Image(img={type(self.__dict__.get('_img')).__name__}, rot={self.rot!r}, "
                f"intr={self.intr!r}, range={self.range!r})")

    def hom(self):
        """Pixel -> ray: ``R^T K^-1`` (reference bundle_adj.py:27-29)."""
        return self.rot.T.dot(np.linalg.inv(self.intr))

    def proj(self):
        """Ray -> pixel: ``K R`` (reference bundle_adj.py:31-33)."""
        return self.intr.dot(self.rot)


def _img_get(self):
    value = self.__dict__.get("_img")
    if isinstance(value, Deferred):
        value = self.__dict__["_img"] = value.make()
    return value


def _img_set(self, value):
    self.__dict__["_img"] = value


# the field `img` is served by a property (attached after @dataclass has read the annotations):
# the generated __init__ / __eq__ / replace go through it like any other attribute access
Image.img = property(_img_get, _img_set)

# pickles name the class by module path: keep the reference's, so camera caches
# move between the reference CLI and this build in both directions
Image.__module__ = "bundle_adj"


def intrinsics(focal, center=(0, 0)):
    """Calibration matrix; like the reference (bundle_adj.py:82-87) a pair of
    focals is accepted but only the first one is used for both axes."""
    if not isinstance(focal, (list, tuple)):
        focal = (focal, focal)
    f = focal[0]
    return np.array([[f, 0, center[0]],
                     [0, f, center[1]],
                     [0, 0, 1]])


def rotation_to_mat(rad):
    """Rodrigues formula, exponential map -> matrix (bundle_adj.py:96-101).
    Unlike the reference there is no random default argument."""
    rad = np.asarray(rad, dtype=np.float64)
    ang = np.linalg.norm(rad)
    axis = rad / ang if ang else rad
    k = np.array([[0, -axis[2], axis[1]],
                  [axis[2], 0, -axis[0]],
                  [-axis[1], axis[0], 0]])
    return np.eye(3) + k * np.sin(ang) + (1 - np.cos(ang)) * k.dot(k)
