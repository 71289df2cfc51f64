"""oracle/ops.py — CPU fp32 restatement of the reference's operator layer (TEST INFRASTRUCTURE).

Each function follows the `_ref` implementation the reference itself falls back to off-GPU.
All paths below are relative to /root/reference.
"""
import numpy as np
import torch
import torch.nn.functional as F

# name -> (function, default alpha, default gain); torch_utils/ops/bias_act.py:23-33
_SQRT2 = float(np.sqrt(2))
ACTIVATIONS = {
    'linear':   (lambda x, a: x,                         0.0, 1.0),
    'relu':     (lambda x, a: F.relu(x),                 0.0, _SQRT2),
    'lrelu':    (lambda x, a: F.leaky_relu(x, a),        0.2, _SQRT2),
    'tanh':     (lambda x, a: torch.tanh(x),             0.0, 1.0),
    'sigmoid':  (lambda x, a: torch.sigmoid(x),          0.0, 1.0),
    'elu':      (lambda x, a: F.elu(x),                  0.0, 1.0),
    'selu':     (lambda x, a: F.selu(x),                 0.0, 1.0),
    'softplus': (lambda x, a: F.softplus(x),             0.0, 1.0),
    'swish':    (lambda x, a: torch.sigmoid(x) * x,      0.0, _SQRT2),
}


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """torch_utils/ops/bias_act.py:93-122 (_bias_act_ref): y = clamp(act(x + b) * gain)."""
    fn, def_alpha, def_gain = ACTIVATIONS[act]
    alpha = float(def_alpha if alpha is None else alpha)
    gain = float(def_gain if gain is None else gain)
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    x = fn(x, alpha)
    if gain != 1:
        x = x * gain
    if clamp is not None and clamp >= 0:
        x = x.clamp(-float(clamp), float(clamp))
    return x


def setup_filter(taps=(1, 3, 3, 1), normalize=True, gain=1.0, separable=None):
    """torch_utils/ops/upfirdn2d.py:72-116: 1-D taps => outer product when fewer than 8, else kept separable (1-D);
    normalised to sum 1."""
    f = torch.as_tensor(taps, dtype=torch.float32)
    if f.ndim == 0:
        f = f[None]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    if normalize:
        f = f / f.sum()
    return f * (gain ** (f.ndim / 2))


def _pad4(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    return [int(p) for p in padding]


def _xy(s):
    return (int(s), int(s)) if isinstance(s, int) else (int(s[0]), int(s[1]))


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
    """torch_utils/ops/upfirdn2d.py:169-213 (_upfirdn2d_ref): zero-insert, pad/crop, FIR, decimate; `up` / `down` are ints or
    (x, y) pairs, `f` is [fh,fw], a separable [taps] or None (identity tap)."""
    n, c, h, w = x.shape
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    upx, upy = _xy(up)
    downx, downy = _xy(down)
    px0, px1, py0, py1 = _pad4(padding)
    assert w * upx + px0 + px1 >= f.shape[-1] and h * upy + py0 + py1 >= f.shape[0]
    x = x.reshape(n, c, h, 1, w, 1)
    x = F.pad(x, [0, upx - 1, 0, 0, 0, upy - 1])                          # zeros AFTER each sample (:188-190)
    x = x.reshape(n, c, h * upy, w * upx)
    x = F.pad(x, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    x = x[:, :, max(-py0, 0): x.shape[2] - max(-py1, 0), max(-px0, 0): x.shape[3] - max(-px1, 0)]
    f = f * (gain ** (f.ndim / 2))
    f = f.to(x.dtype)
    if not flip_filter:
        f = f.flip(list(range(f.ndim)))
    if f.ndim == 2:
        x = F.conv2d(x, f[None, None].repeat(c, 1, 1, 1), groups=c)
    else:
        x = F.conv2d(x, f[None, None, None, :].repeat(c, 1, 1, 1), groups=c)
        x = F.conv2d(x, f[None, None, :, None].repeat(c, 1, 1, 1), groups=c)
    return x[:, :, ::downy, ::downx]


def upsample2d(x, f, up=2, gain=1.0):
    """torch_utils/ops/upfirdn2d.py:315-350."""
    fw = fh = f.shape[-1]
    p = [(fw + up - 1) // 2, (fw - up) // 2, (fh + up - 1) // 2, (fh - up) // 2]
    return upfirdn2d(x, f, up=up, padding=p, gain=gain * up * up)


def downsample2d(x, f, down=2, gain=1.0):
    """torch_utils/ops/upfirdn2d.py:354-389."""
    fw = fh = f.shape[-1]
    p = [(fw - down + 1) // 2, (fw - down) // 2, (fh - down + 1) // 2, (fh - down) // 2]
    return upfirdn2d(x, f, down=down, padding=p, gain=gain)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, quant=None):
    """torch_utils/ops/conv2d_resample.py:48-143 — the branches the generator reaches.
    `quant` (optional callable) is applied to the result of every internal convolution / FIR pass: with
    `lambda t: t.half().float()` this emulates the storage rounding of the reference's fp16 path on fp32 arithmetic.

    flip_weight=True means correlation (= F.conv2d).  For up>1 the reference transposes the
    weight and calls conv_transpose2d with flip_weight inverted (:114-131), i.e. for the
    generator's up-layers (flip_weight=False) no explicit flip happens anywhere.
    """
    kh, kw = w.shape[2], w.shape[3]
    fw = fh = 1 if f is None else f.shape[-1]
    px0, px1, py0, py1 = _pad4(padding)
    if up > 1:
        px0 += (fw + up - 1) // 2; px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2; py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2; px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2; py1 += (fh - down) // 2

    q = quant if quant is not None else (lambda t: t)
    _fir = globals()['upfirdn2d']

    def upfirdn2d(*a, **k):                                    # every FIR pass of this function is quantised
        return q(_fir(*a, **k))

    def conv(x, w, stride=1, pad=(0, 0), transpose=False, flip=True):
        if not flip and (w.shape[2] > 1 or w.shape[3] > 1):
            w = w.flip([2, 3])
        if transpose:
            return q(F.conv_transpose2d(x, w, stride=stride, padding=pad, groups=groups))
        return q(F.conv2d(x, w, stride=stride, padding=pad, groups=groups))

    if kw == 1 and kh == 1 and down > 1 and up == 1:          # :96
        x = upfirdn2d(x, f, down=down, padding=[px0, px1, py0, py1])
        return conv(x, w, flip=flip_weight)
    if kw == 1 and kh == 1 and up > 1 and down == 1:          # :102
        x = conv(x, w, flip=flip_weight)
        return upfirdn2d(x, f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2)
    if down > 1 and up == 1:                                   # :108
        x = upfirdn2d(x, f, padding=[px0, px1, py0, py1])
        return conv(x, w, stride=down, flip=flip_weight)
    if up > 1:                                                 # :114
        oc, icg = w.shape[0], w.shape[1]
        if groups == 1:
            w = w.transpose(0, 1)
        else:
            w = w.reshape(groups, oc // groups, icg, kh, kw).transpose(1, 2)
            w = w.reshape(groups * icg, oc // groups, kh, kw)
        px0 -= kw - 1; px1 -= kw - up; py0 -= kh - 1; py1 -= kh - up
        pxt = max(min(-px0, -px1), 0); pyt = max(min(-py0, -py1), 0)
        x = conv(x, w, stride=up, pad=(pyt, pxt), transpose=True, flip=(not flip_weight))
        x = upfirdn2d(x, f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2)
        if down > 1:
            x = upfirdn2d(x, f, down=down)
        return x
    if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:   # :134
        return conv(x, w, pad=(py0, px0), flip=flip_weight)
    x = upfirdn2d(x, (f if up > 1 else None), up=up, padding=[px0, px1, py0, py1], gain=up ** 2)
    x = conv(x, w, flip=flip_weight)
    if down > 1:
        x = upfirdn2d(x, f, down=down)
    return x


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None,
                     demodulate=True, flip_weight=True):
    """training_avatar_texture/networks_stylegan2.py:34-91, fused (grouped-conv) branch :81-91,
    which is what `fused_modconv_default='inference_only'` selects in eval mode."""
    n = x.shape[0]
    oc, ic, kh, kw = weight.shape
    w = weight.unsqueeze(0) * styles.reshape(n, 1, -1, 1, 1)
    if demodulate:
        d = (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
        w = w * d.reshape(n, -1, 1, 1, 1)
    x = x.reshape(1, -1, *x.shape[2:])
    w = w.reshape(-1, ic, kh, kw)
    x = conv2d_resample(x, w, f=resample_filter, up=up, down=down, padding=padding, groups=n, flip_weight=flip_weight)
    x = x.reshape(n, -1, *x.shape[2:])
    if noise is not None:
        x = x + noise
    return x


def fully_connected(x, weight, bias, activation='linear', lr_multiplier=1.0):
    """training_avatar_texture/networks_stylegan2.py:114-127 (FullyConnectedLayer.forward)."""
    w = weight * (lr_multiplier / np.sqrt(weight.shape[1]))
    b = bias
    if b is not None and lr_multiplier != 1:
        b = b * lr_multiplier
    if activation == 'linear' and b is not None:
        return torch.addmm(b.unsqueeze(0), x, w.t())
    x = x.matmul(w.t())
    return bias_act(x, b, act=activation)


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=_SQRT2, slope=0.2,
                   clamp=None, flip_filter=False):
    """torch_utils/ops/filtered_lrelu.py:123-155 (_filtered_lrelu_ref)."""
    px0, px1, py0, py1 = _pad4(padding)
    x = bias_act(x, b)
    x = upfirdn2d(x, fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    x = bias_act(x, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
    return upfirdn2d(x, fd, down=down, flip_filter=flip_filter)
