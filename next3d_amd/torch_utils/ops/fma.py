"""fma(a, b, c) = a * b + c (reference torch_utils/ops/fma.py:17-28): only the non-fused modulated-conv branch calls it
(`x = fma(x, dcoefs[N,C,1,1], noise)`), which inference never takes — the generator fuses this into the conv epilogue.
Kept at the operator boundary on a HIP kernel (n3d_fma) with the broadcasting that call site needs.  float32, or float16
operands (converted on the device, float32 arithmetic, float16 result as the reference's `a * b + c` on half tensors)."""
import torch

from ... import _lib


def fma(a, b, c):
    _lib.require_device(a, b, c)
    if not (a.dtype == b.dtype == c.dtype) or a.dtype not in (torch.float32, torch.float16):
        raise RuntimeError(f'fma: three float32 or three float16 operands expected, got {a.dtype}, {b.dtype}, {c.dtype}')
    out_dtype = a.dtype
    shape = torch.broadcast_shapes(a.shape, b.shape, c.shape)
    if len(shape) != 4:
        raise RuntimeError('fma: 4-D (NCHW) operands expected on the generator path')
    a = _lib.cast(a.expand(shape).contiguous(), torch.float32)
    n, ch, h, w = shape

    def strides(t):
        t = _lib.cast(t.contiguous(), torch.float32)
        while t.ndim < 4:
            t = t.unsqueeze(0)
        if tuple(t.shape) == tuple(shape):
            return t.contiguous(), h * w, 1
        if t.shape[2] == 1 and t.shape[3] == 1:                      # per-(n,c) factor
            return t.expand(n, ch, 1, 1).contiguous(), 1, 0
        if t.shape[0] == 1 and t.shape[1] == 1:                      # per-pixel term shared by all (n,c)
            return t.expand(1, 1, h, w).contiguous(), 0, 1
        return t.expand(shape).contiguous(), h * w, 1

    b2, b_nc, b_p = strides(b)
    c2, c_nc, c_p = strides(c)
    y = torch.empty(shape, dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().n3d_fma(_lib.ptr(a), _lib.ptr(b2), _lib.ptr(c2), _lib.ptr(y), n * ch, h * w, b_nc, b_p, c_nc, c_p, _lib.stream()))
    return _lib.cast(y, out_dtype)
