"""``python stitcher.py PATH [-s] [--ba] [-e] [-c] [-b] [-o]`` - the reference's
command line (stitcher.py:390-457), served by the MI355X build."""
import logging

from pano360_amd.stitcher import *  # noqa: F401,F403
from pano360_amd.stitcher import (_add_weights, _hat, _proj_img_range_border,  # noqa: F401
                                  _proj_img_range_corners, _valid, main)

if __name__ == "__main__":
    logging.basicConfig(level=logging.DEBUG)
    main()
