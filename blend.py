"""``blend.laplacian_blending`` of the reference (blend.py:105-140), served by the
MI355X build; the other experiments of the reference's blend.py are out of scope."""
from pano360_amd.blend import laplacian_blending  # noqa: F401
