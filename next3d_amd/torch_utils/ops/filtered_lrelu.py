"""filtered_lrelu — the Python signature of the reference's torch_utils/ops/filtered_lrelu.py:58-118 (StyleGAN3's alias-free
non-linearity: bias -> up-sampling FIR -> leaky ReLU, gain, clamp -> down-sampling FIR), forward only, ONE launch of
n3d_filtered_lrelu (csrc/filtered_lrelu.hip: the up-sampled intermediate never leaves LDS).

The Next3D generator never executes this op (its only caller is training/networks_stylegan3.py:357, merely imported by
superresolution.py:22); it is part of the operator boundary.  Separable (1-D) filters are expanded to their 2-D outer
product on the host — the same taps the reference's two passes apply, up to float rounding of the products.
"""
import numpy as np
import torch

from ... import _lib
from .upfirdn2d import _get_filter_size, _parse_padding


def _taps2d(f, device):
    """None | [taps] | [fh, fw] float32 -> (contiguous [fh, fw] tensor or None, fh, fw)."""
    if f is None:
        return None, 1, 1
    if f.dtype != torch.float32 or f.ndim not in (1, 2):
        raise RuntimeError('filtered_lrelu: filters must be 1-D or 2-D float32 tensors (upfirdn2d.setup_filter)')
    f2 = (f[:, None] * f[None, :]) if f.ndim == 1 else f
    return f2.to(device).contiguous(), int(f2.shape[0]), int(f2.shape[1])


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None,
                   flip_filter=False, impl='cuda'):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert impl in ['ref', 'cuda']
    if impl == 'ref':
        raise RuntimeError("impl='ref' is not part of the product: the CPU restatement is oracle/ops.py (tests only)")
    _lib.require_device(x, fu, fd, b)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    assert gain == float(gain) and gain > 0 and slope == float(slope) and slope >= 0
    assert clamp is None or (clamp == float(clamp) and clamp >= 0)
    if x.dtype not in (torch.float32, torch.float16):
        raise RuntimeError(f'filtered_lrelu: float32 or float16 input expected, got {x.dtype}')
    if b is not None and (b.dtype != x.dtype or b.ndim != 1 or b.shape[0] != x.shape[1]):
        raise RuntimeError('filtered_lrelu: b must be a 1-D tensor with one entry per channel and the dtype of x')
    px0, px1, py0, py1 = _parse_padding(padding)
    fu_w, fu_h = _get_filter_size(fu)
    fd_w, fd_h = _get_filter_size(fd)
    n, c, in_h, in_w = x.shape
    out_w = (in_w * up + (px0 + px1) - (fu_w - 1) - (fd_w - 1) + (down - 1)) // down
    out_h = (in_h * up + (py0 + py1) - (fu_h - 1) - (fd_h - 1) + (down - 1)) // down
    if out_w < 1 or out_h < 1:
        raise RuntimeError('filtered_lrelu: output would be empty')
    out_dtype = x.dtype
    x32 = _lib.cast(x, torch.float32).contiguous()                       # fp16: fp32 arithmetic, fp16 storage
    b32 = None if b is None else _lib.cast(b, torch.float32).contiguous()
    fu2, fuh, fuw = _taps2d(fu, x.device)
    fd2, fdh, fdw = _taps2d(fd, x.device)
    y = torch.empty([n, c, out_h, out_w], dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().n3d_filtered_lrelu(_lib.ptr(x32), _lib.ptr(fu2), _lib.ptr(fd2), _lib.ptr(b32), _lib.ptr(y), n, c, in_h, in_w,
                                             fuh, fuw, fdh, fdw, up, down, px0, px1, py0, py1, float(gain), float(slope),
                                             float(-1 if clamp is None else clamp), 1 if flip_filter else 0, _lib.stream()))
    return _lib.cast(y, out_dtype)
