"""The Gaussian / pyramid part of the reference ``features.py`` on the GPU.

Mirrors ``gaussian_filter(img, sigma=1.0)`` (features.py:20-24) and the
``cv2.pyrDown`` pyramid step of the MSOP detector (features.py:155); the
filters run in ``libpano360_hip.so`` (``pano_blur_plane``, ``pano_pyr_down``).
Keypoint detection, description and matching stay outside this build's scope
(SURVEY.md §2).
"""
import numpy as np

from . import engine as _eng


def _to_device(img):
    import torch
    eng = _eng.engine()
    return eng, torch.from_numpy(np.ascontiguousarray(img, np.float32)).to(eng.device)


def gaussian_filter(img, sigma=1.0):
    """Compute the kernel size from sigma and smooth the image
    (features.py:20-24): ksize = max(int((sigma-0.35)/0.15), 1), made odd."""
    ksz = max(int((sigma - 0.35) / 0.15), 1)
    ksz += not ksz % 2
    eng, dev = _to_device(img)
    if dev.ndim == 2:
        return eng.blur_plane(dev, ksz, sigma).cpu().numpy()
    planes = [eng.blur_plane(dev[..., c].contiguous(), ksz, sigma) for c in range(dev.shape[2])]
    import torch
    return torch.stack(planes, dim=-1).cpu().numpy()


def pyr_down(img):
    """``cv2.pyrDown`` of a float32 plane (features.py:155)."""
    eng, dev = _to_device(img)
    return eng.pyr_down(dev).cpu().numpy()


def gaussian_pyramid(img, levels=4):
    """The pyrDown chain the MSOP detector walks (features.py:138-155)."""
    eng, dev = _to_device(img)
    out = [dev]
    for _ in range(levels - 1):
        out.append(eng.pyr_down(out[-1]))
    return [p.cpu().numpy() for p in out]
