"""filtered_lrelu — same signature as reference torch_utils/ops/filtered_lrelu.py:58-118 (StyleGAN3's alias-free
non-linearity).  The Next3D generator never executes it (its only caller is training/networks_stylegan3.py:357, merely
imported by superresolution.py:22), so it is provided at the operator boundary as the reference's own decomposition
(`_filtered_lrelu_ref`, :123-155) on libn3d.so kernels, with the gain / leaky-ReLU / clamp fused into the up-sampling
FIR's epilogue:   bias_act(x, b)  ->  upfirdn2d(fu, up, pad, gain=up^2) [+ lrelu*gain, clamp]  ->  upfirdn2d(fd, down).
"""
import numpy as np
import torch

from ... import _lib
from . import bias_act, upfirdn2d
from .upfirdn2d import _get_filter_size, _parse_padding


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
    if b is not None:
        if b.dtype != x.dtype or b.ndim != 1 or b.shape[0] != x.shape[1]:
            raise RuntimeError('filtered_lrelu: b must be a 1-D tensor with one entry per channel and the dtype of x')
    px0, px1, py0, py1 = _parse_padding(padding)
    fu_w, fu_h = _get_filter_size(fu)
    fd_w, fd_h = _get_filter_size(fd)
    n, c, in_h, in_w = x.shape
    out_w = (in_w * up + (px0 + px1) - (fu_w - 1) - (fd_w - 1) + (down - 1)) // down
    out_h = (in_h * up + (py0 + py1) - (fu_h - 1) - (fd_h - 1) + (down - 1)) // down
    if b is not None:
        x = bias_act.bias_act(x=x, b=b)
    act = _lib.make_epilogue(act='lrelu', alpha=slope, gain=gain, clamp=clamp)
    if fu is None or fu.ndim == 2:      # the activation rides in the (single) up-FIR launch
        x = upfirdn2d.upfirdn2d(x=x, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter, _epilogue=act)
    else:                               # separable filter: two passes, activation after the second
        x = upfirdn2d.upfirdn2d(x=x, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter, _epilogue=act)
    x = upfirdn2d.upfirdn2d(x=x, f=fd, down=down, flip_filter=flip_filter)
    assert tuple(x.shape) == (n, c, out_h, out_w)
    return x
