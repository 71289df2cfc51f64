"""fma(a, b, c) = a * b + c (reference torch_utils/ops/fma.py:17-28).  Only the non-fused modulated-conv branch
calls it, which inference never takes; kept for import compatibility and executed as a bias_act-style HIP pass
is not needed: the generator fuses this into the conv epilogue (n3d_epilogue.row_scale + noise)."""
import torch


def fma(a, b, c):
    if a.device.type != 'cuda':
        raise RuntimeError('n3d ops run on a HIP device only')
    return torch.addcmul(c, a, b)
