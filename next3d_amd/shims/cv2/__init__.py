"""`cv2` as far as the Next3D generator forward needs it (next3d_amd/shims/__init__.py):
  floodFill  volumetric_rendering/renderer.py:593 — `cv2.floodFill(img32f, mask, (0, 0), (255,)*3, (0,)*3, (254,)*3, FLOODFILL_FIXED_RANGE)`
             on libn3d.so (n3d_flood_fill).  The reference hands over a NumPy image (`image[0].cpu().numpy() * 255`, :588-589): it is copied
             to the device, filled and copied back in place, so the reference's own fill_mouth runs unchanged; a float32 HIP tensor is
             filled in place without any copy.
  imread     triplane_next3d.py:91 (the uv mask) — BGR uint8 through PIL.
Anything else raises: this is not OpenCV."""
import numpy as np
import torch

from ... import _lib

FLOODFILL_FIXED_RANGE = 1 << 16
FLOODFILL_MASK_ONLY = 1 << 17
IMREAD_COLOR = 1


def _device():
    return torch.device('cuda')


def _scalar(v):
    return float(v[0]) if isinstance(v, (tuple, list, np.ndarray)) else float(v)


def floodFill(image, mask, seedPoint, newVal, loDiff=0, upDiff=0, flags=4):
    if tuple(seedPoint) != (0, 0) or not (flags & FLOODFILL_FIXED_RANGE) or (flags & FLOODFILL_MASK_ONLY) or (flags & 0xff) not in (0, 4):
        raise RuntimeError('cv2 shim: floodFill is implemented for seed (0, 0), FLOODFILL_FIXED_RANGE, 4-connectivity (the generator-forward call)')
    is_np = isinstance(image, np.ndarray)
    if is_np:
        if image.dtype != np.float32 or image.ndim != 2:
            raise RuntimeError('cv2 shim: floodFill takes a 2-D float32 image')
        t = torch.from_numpy(np.ascontiguousarray(image)).to(_device())
    else:
        t = image
        if t.dtype != torch.float32 or t.ndim != 2 or not t.is_contiguous():
            raise RuntimeError('cv2 shim: floodFill takes a contiguous 2-D float32 image')
        _lib.require_device(t)
    h, w = t.shape
    _lib.check(_lib.lib().n3d_flood_fill(_lib.ptr(t), 1, h, w, _scalar(newVal), _scalar(loDiff), _scalar(upDiff), _lib.stream()))
    if is_np:
        image[...] = t.cpu().numpy()
    return 0, image, mask, (0, 0, w, h)


def imread(filename, flags=IMREAD_COLOR):
    import os
    if not os.path.exists(filename):
        return None                                   # OpenCV's behaviour (the reference then fails on `.astype`, as with the real cv2)
    from PIL import Image
    rgb = np.asarray(Image.open(filename).convert('RGB'), dtype=np.uint8)
    return np.ascontiguousarray(rgb[:, :, ::-1])      # BGR


def __getattr__(name):
    if name.startswith('__'):
        raise AttributeError(name)

    def _missing(*a, **k):
        raise RuntimeError(f'cv2 shim: cv2.{name} is outside the generator-forward path (next3d_amd/shims/cv2 is not OpenCV)')
    return _missing
