"""bias_act — same Python signature as reference torch_utils/ops/bias_act.py:54, forward only, HIP only."""
import numpy as np
import torch

from ... import _lib
from ...dnnlib import EasyDict

# name -> defaults; cuda_idx doubles as the C-ABI activation id (reference bias_act.py:23-33)
activation_funcs = {
    'linear':   EasyDict(def_alpha=0,   def_gain=1,          cuda_idx=1),
    'relu':     EasyDict(def_alpha=0,   def_gain=np.sqrt(2), cuda_idx=2),
    'lrelu':    EasyDict(def_alpha=0.2, def_gain=np.sqrt(2), cuda_idx=3),
    'tanh':     EasyDict(def_alpha=0,   def_gain=1,          cuda_idx=4),
    'sigmoid':  EasyDict(def_alpha=0,   def_gain=1,          cuda_idx=5),
    'elu':      EasyDict(def_alpha=0,   def_gain=1,          cuda_idx=6),
    'selu':     EasyDict(def_alpha=0,   def_gain=1,          cuda_idx=7),
    'softplus': EasyDict(def_alpha=0,   def_gain=1,          cuda_idx=8),
    'swish':    EasyDict(def_alpha=0,   def_gain=np.sqrt(2), cuda_idx=9),
}


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """y = clamp(act(x + b) * gain); x any shape, b 1-D along `dim`.  float32 / float16."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'ref':
        raise RuntimeError("impl='ref' is not part of the product: the CPU restatement is oracle/ops.py (tests only)")
    assert clamp is None or clamp >= 0
    _lib.require_device(x, b)
    spec = activation_funcs[act]
    alpha = float(alpha if alpha is not None else spec.def_alpha)
    gain = float(gain if gain is not None else spec.def_gain)
    clamp = float(clamp if clamp is not None else -1)
    if x.dtype not in (torch.float32, torch.float16):
        raise RuntimeError('bias_act: x must be float32 or float16')          # bias_act.cpp:40-42 analogue
    if b is not None:
        assert b.ndim == 1 and 0 <= dim < x.ndim
        if b.shape[0] != x.shape[dim]:
            raise RuntimeError('bias_act: b has wrong number of elements')     # bias_act.cpp:52
        if b.dtype != x.dtype:
            raise RuntimeError('bias_act: b must have the same dtype as x')
        b = b.contiguous()
    x = x.contiguous()
    y = torch.empty_like(x)
    step_b = 1
    for d in range(dim + 1, x.ndim):
        step_b *= x.shape[d]
    _lib.check(_lib.lib().n3d_bias_act(_lib.ptr(x), _lib.ptr(b), _lib.ptr(y), x.numel(),
                                       b.shape[0] if b is not None else 1, step_b,
                                       0 if x.dtype == torch.float32 else 1, spec.cuda_idx, alpha, gain, clamp, _lib.stream()))
    return y
