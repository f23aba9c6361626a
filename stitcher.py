"""``python stitcher.py PATH [-s] [--ba] [-e] [-c] [-b] [-o]`` - the reference's
command line (stitcher.py:390-457), served by the MI355X build.

``import stitcher`` gives the very module object of ``pano360_amd.stitcher``, not a copy of
its names: module attributes the reference reads at call time (``MAX_RESOLUTION``,
stitcher.py:17,154; ``BLENDERS``) can be set on it exactly as on the reference's module."""
import logging
import sys

import pano360_amd.stitcher as _impl

if __name__ == "__main__":
    logging.basicConfig(level=logging.DEBUG)
    _impl.main()
else:
    sys.modules[__name__] = _impl
