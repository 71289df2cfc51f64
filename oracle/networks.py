"""oracle/networks.py — CPU fp32 restatement of the StyleGAN2 / StyleUNet / SR networks
(TEST INFRASTRUCTURE).  Functional style: every function takes the flat state dict `P`
(reference parameter names) and a dotted `prefix`.  Paths relative to /root/reference;
`tat/` = training_avatar_texture/.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops

FIR = ops.setup_filter((1, 3, 3, 1))
_LRELU_GAIN = float(np.sqrt(2))


def channels_dict(img_resolution, channel_base=32768, channel_max=512):
    """tat/networks_stylegan2.py:613-614."""
    log2 = int(np.log2(img_resolution))
    return {2 ** i: min(channel_base // 2 ** i, channel_max) for i in range(2, log2 + 1)}


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    """tat/networks_stylegan2.py:27-29."""
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


def mapping_network(P, prefix, z, c, num_ws, num_layers=2, truncation_psi=1.0, truncation_cutoff=None,
                    lr_multiplier=0.01):
    """tat/networks_stylegan2.py:233-268 (MappingNetwork.forward)."""
    x = normalize_2nd_moment(z.to(torch.float32))
    y = ops.fully_connected(c.to(torch.float32), P[f'{prefix}.embed.weight'], P[f'{prefix}.embed.bias'])
    x = torch.cat([x, normalize_2nd_moment(y)], dim=1)
    for i in range(num_layers):
        x = ops.fully_connected(x, P[f'{prefix}.fc{i}.weight'], P[f'{prefix}.fc{i}.bias'],
                                activation='lrelu', lr_multiplier=lr_multiplier)
    x = x.unsqueeze(1).repeat(1, num_ws, 1)
    if truncation_psi != 1:
        w_avg = P[f'{prefix}.w_avg']
        if truncation_cutoff is None:
            x = w_avg.lerp(x, truncation_psi)
        else:
            x[:, :truncation_cutoff] = w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
    return x


def conv2d_layer(P, prefix, x, kernel_size, activation='linear', up=1, down=1, conv_clamp=None, gain=1.0):
    """tat/networks_stylegan2.py:173-183 (Conv2dLayer.forward)."""
    weight = P[f'{prefix}.weight']
    bias = P.get(f'{prefix}.bias')
    w = weight * (1.0 / np.sqrt(weight.shape[1] * kernel_size ** 2))
    x = ops.conv2d_resample(x, w, f=FIR, up=up, down=down, padding=kernel_size // 2, flip_weight=(up == 1))
    act_gain = ops.ACTIVATIONS[activation][2] * gain
    act_clamp = conv_clamp * gain if conv_clamp is not None else None
    return ops.bias_act(x, bias, act=activation, gain=act_gain, clamp=act_clamp)


def synthesis_layer(P, prefix, x, w, up=1, noise_mode='const', conv_clamp=None, gain=1.0):
    """tat/networks_stylegan2.py:311-330 (SynthesisLayer.forward)."""
    styles = ops.fully_connected(w, P[f'{prefix}.affine.weight'], P[f'{prefix}.affine.bias'])
    noise = None
    if noise_mode == 'const':
        noise = P[f'{prefix}.noise_const'] * P[f'{prefix}.noise_strength']
    x = ops.modulated_conv2d(x, P[f'{prefix}.weight'], styles, noise=noise, up=up, padding=1,
                             resample_filter=FIR, flip_weight=(up == 1))
    act_clamp = conv_clamp * gain if conv_clamp is not None else None
    return ops.bias_act(x, P[f'{prefix}.bias'], act='lrelu', gain=_LRELU_GAIN * gain, clamp=act_clamp)


def torgb_layer(P, prefix, x, w, conv_clamp=None):
    """tat/networks_stylegan2.py:353-357 (ToRGBLayer.forward)."""
    weight = P[f'{prefix}.weight']
    styles = ops.fully_connected(w, P[f'{prefix}.affine.weight'], P[f'{prefix}.affine.bias'])
    styles = styles * (1.0 / np.sqrt(weight.shape[1] * weight.shape[2] ** 2))
    x = ops.modulated_conv2d(x, weight, styles, demodulate=False)
    return ops.bias_act(x, P[f'{prefix}.bias'], clamp=conv_clamp)


def _q(t):
    return t.half().float()


def _q16(t):
    """float16 storage rounding on float32 values."""
    return t.half().float()


def _modconv_fp16(P, prefix, x, styles, up, demodulate, noise=None):
    """modulated_conv2d's float16 branch (tat/networks_stylegan2.py:56-91, fused): weight / style pre-normalisation against
    overflow (:57-59), per-sample weights `w.to(float16)`, grouped convolution.  Emulated on float32 arithmetic: every tensor
    the reference holds in float16 is rounded to float16 here (products of two float16 values are exact in float32, the
    accumulation is float32 as in cuDNN's half kernels)."""
    weight = P[f'{prefix}.weight']
    n = x.shape[0]
    o, i, kh, kw = weight.shape
    if demodulate:
        weight = weight * (1 / np.sqrt(i * kh * kw) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)
    w = weight.unsqueeze(0) * styles.reshape(n, 1, -1, 1, 1)
    if demodulate:
        w = w * (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt().reshape(n, -1, 1, 1, 1)
    y = ops.conv2d_resample(x.reshape(1, -1, *x.shape[2:]), _q16(w.reshape(-1, i, kh, kw)), f=FIR, up=up, padding=kh // 2, groups=n,
                            flip_weight=(up == 1), quant=_q16)
    y = y.reshape(n, -1, *y.shape[2:])
    if noise is not None:                                  # `x.add_(noise)` on the float16 result (:89-90): float32 sum, float16 storage
        y = _q16(y + noise)
    return y


def _bias_act_half_cpu(x, b, act='linear', gain=1.0, clamp=None):
    """_bias_act_ref applied to float16 tensors (torch_utils/ops/bias_act.py:93-122 — what the reference's float16 blocks run
    off-GPU): x + b, the activation and x * gain are separate float16 tensor ops, each rounding; emulated on float32 values."""
    x = _q(x + _q(b).reshape(1, -1, 1, 1))
    if act == 'lrelu':
        x = _q(torch.where(x > 0, x, x * 0.2))
    if gain != 1:
        x = _q(x * gain)
    return x.clamp(-clamp, clamp) if clamp is not None else x


def synthesis_block_fp16(P, prefix, x, img, ws, conv_clamp=256, cpu_rounding=False, noise_mode='none'):
    """SynthesisBlock.forward with use_fp16 and not force_fp32 (training/networks_stylegan2.py:417-452; noise_mode 'none': the
    super-resolution blocks, 'const': float16 blocks of the backbones, num_fp16_res > 0): x is cast to float16 at entry, every layer
    returns float16, the skip image is float32.
    bias_act: `cpu_rounding=False` models bias_act.cu (bias_act.cu:19-50: float32 inside, ONE float16 rounding — the reference on a
    GPU); `cpu_rounding=True` models _bias_act_ref on half tensors (the reference off-GPU).  The latter reproduces the reference's
    own CPU run of this branch (oracle/pin_against_reference.py --fp16, tests/golden/*_fp16sr.npz) up to the accumulation order
    inside ATen's half convolutions: the two settings differ in nothing else."""
    w0, w1, w2 = ws.unbind(dim=1)
    aff = lambda k, w: ops.fully_connected(w, P[f'{prefix}.{k}.affine.weight'], P[f'{prefix}.{k}.affine.bias'])
    ba = _bias_act_half_cpu if cpu_rounding else (lambda x, b, **kw: _q(ops.bias_act(x, _q(b), **kw)))
    x = _q(x)
    for k, w, up in (('conv0', w0, 2), ('conv1', w1, 1)):
        nz = P[f'{prefix}.{k}.noise_const'] * P[f'{prefix}.{k}.noise_strength'] if noise_mode == 'const' else None
        x = _modconv_fp16(P, f'{prefix}.{k}', x, aff(k, w), up, True, noise=nz)
        x = ba(x, P[f'{prefix}.{k}.bias'], act='lrelu', gain=_LRELU_GAIN, clamp=conv_clamp)
    wt = P[f'{prefix}.torgb.weight']
    y = _modconv_fp16(P, f'{prefix}.torgb', x, aff('torgb', w2) * (1.0 / np.sqrt(wt.shape[1] * wt.shape[2] ** 2)), 1, False)
    y = ba(y, P[f'{prefix}.torgb.bias'], clamp=conv_clamp)
    img = ops.upsample2d(img, FIR) + y if img is not None else y
    return x, img


def synthesis_block(P, prefix, x, img, ws, in_channels, noise_mode='const', conv_clamp=None):
    """tat/networks_stylegan2.py:544-588 (SynthesisBlock.forward), 'skip' architecture, fp32."""
    w_iter = iter(ws.unbind(dim=1))
    if in_channels == 0:
        x = P[f'{prefix}.const'].unsqueeze(0).repeat(ws.shape[0], 1, 1, 1)
        x = synthesis_layer(P, f'{prefix}.conv1', x, next(w_iter), noise_mode=noise_mode, conv_clamp=conv_clamp)
    else:
        x = synthesis_layer(P, f'{prefix}.conv0', x, next(w_iter), up=2, noise_mode=noise_mode, conv_clamp=conv_clamp)
        x = synthesis_layer(P, f'{prefix}.conv1', x, next(w_iter), noise_mode=noise_mode, conv_clamp=conv_clamp)
    if img is not None:
        img = ops.upsample2d(img, FIR)
    y = torgb_layer(P, f'{prefix}.torgb', x, next(w_iter), conv_clamp=conv_clamp)
    img = img + y if img is not None else y
    return x, img


def _split_ws(ws, block_resolutions):
    """tat/networks_stylegan2.py:632-640: first block has 1 conv, the others 2; +1 torgb each."""
    out, idx = [], 0
    for res in block_resolutions:
        nconv = 1 if res == 4 else 2
        out.append(ws.narrow(1, idx, nconv + 1))
        idx += nconv
    return out


def synthesis_network(P, prefix, ws, img_resolution=256, noise_mode='const', return_features=False, fp16_resolution=None, conv_clamp=None,
                      cpu_rounding=False):
    """tat/networks_stylegan2.py:630-645 (SynthesisNetwork.forward).  fp16_resolution (num_fp16_res > 0, :615-621): the blocks of that
    resolution and up are float16 blocks (emulated, synthesis_block_fp16); conv_clamp applies to every block."""
    cd = {r: int(P[f'{prefix}.b{r}.conv1.weight'].shape[0]) for r in channels_dict(img_resolution)}     # (any channel_base / channel_max: read off the weights)
    block_res = sorted(cd.keys())
    x = img = None
    for res, cur_ws in zip(block_res, _split_ws(ws.to(torch.float32), block_res)):
        in_ch = cd[res // 2] if res > 4 else 0
        if fp16_resolution is not None and res >= fp16_resolution:
            x, img = synthesis_block_fp16(P, f'{prefix}.b{res}', x, img, cur_ws, conv_clamp=conv_clamp, cpu_rounding=cpu_rounding, noise_mode=noise_mode)
        else:
            x, img = synthesis_block(P, f'{prefix}.b{res}', x, img, cur_ws, in_ch, noise_mode=noise_mode, conv_clamp=conv_clamp)
    return (img, x) if return_features else img


def encoder_res_block(P, prefix, inp, skip, downsample):
    """tat/networks_stylegan2_styleunet.py:107-115 (EncoderResBlock.forward)."""
    if downsample:
        inp = ops.downsample2d(inp, FIR)
    out = conv2d_layer(P, f'{prefix}.fromrgb', inp, 1, activation='linear')
    if skip is not None:
        out = out + skip
    out = conv2d_layer(P, f'{prefix}.conv1', out, 3, activation='lrelu')
    out = conv2d_layer(P, f'{prefix}.conv2', out, 3, activation='lrelu', down=2)
    return inp, out


def styleunet_synthesis(P, prefix, x_in, ws, img_resolution=256, in_size=64, final_size=4, num_cond_res=64,
                        noise_mode='const', fp16_resolution=None, conv_clamp=None, cpu_rounding=False):
    """tat/networks_stylegan2_styleunet.py:554-588 (conditional SynthesisNetwork.forward)."""
    cd = {r: int(P[f'{prefix}.b{r}.conv1.weight'].shape[0]) for r in channels_dict(img_resolution)}     # (any channel_base / channel_max: read off the weights)
    block_res = sorted(cd.keys())
    block_ws = _split_ws(ws.to(torch.float32), block_res)
    enc_res = [2 ** i for i in range(int(np.log2(in_size)), int(np.log2(final_size)) - 1, -1)]

    cond_list, cond_out = [], None
    for i, res in enumerate(enc_res[:-1]):          # same count as the reversed iteration at :569
        x_in, cond_out = encoder_res_block(P, f'{prefix}.encoder.{i}', x_in, cond_out,
                                           downsample=(enc_res[:-1][i] < in_size))
        cond_list.append(cond_out)
    cond_list = cond_list[::-1]

    x = img = None
    start = int(np.log2(final_size)) - 1
    for index, (res, cur_ws) in enumerate(zip(block_res[start:], block_ws[start:])):
        if 2 ** (index + int(np.log2(final_size))) < num_cond_res:
            if index == 0:
                x = conv2d_layer(P, f'{prefix}.fusion.{index}', cond_list[index], 3, activation='linear')
            else:
                x = conv2d_layer(P, f'{prefix}.fusion.{index}', torch.cat([x, cond_list[index]], dim=1), 3,
                                 activation='linear')
        in_ch = cd[res // 2] if res > 4 else 0
        if fp16_resolution is not None and res >= fp16_resolution:     # (torch.cat above promoted a float16 x to float32: same values)
            x, img = synthesis_block_fp16(P, f'{prefix}.b{res}', x, img, cur_ws, conv_clamp=conv_clamp, cpu_rounding=cpu_rounding, noise_mode=noise_mode)
        else:
            x, img = synthesis_block(P, f'{prefix}.b{res}', x, img, cur_ws, in_ch, noise_mode=noise_mode, conv_clamp=conv_clamp)
    return img


def synthesis_block_noup(P, prefix, x, img, ws, noise_mode='const', conv_clamp=None):
    """tat/superresolution.py:210-254 (SynthesisBlockNoUp.forward, 'skip' architecture, fp32): conv0 without up-sampling, conv1, toRGB; the skip
    image is added WITHOUT upsample2d (the two lines are commented out in the reference, :244-246)."""
    w0, w1, w2 = ws.unbind(dim=1)
    x = synthesis_layer(P, f'{prefix}.conv0', x, w0, noise_mode=noise_mode, conv_clamp=conv_clamp)
    x = synthesis_layer(P, f'{prefix}.conv1', x, w1, noise_mode=noise_mode, conv_clamp=conv_clamp)
    y = torgb_layer(P, f'{prefix}.torgb', x, w2, conv_clamp=conv_clamp)
    return x, (img + y if img is not None else y)


# class name -> (input_resolution, resize rule, first block is a SynthesisBlockNoUp) — tat/superresolution.py:29-124, :264-290
_SR = {'SuperresolutionHybrid8XDC': (128, 'ne', False), 'SuperresolutionHybrid8X': (128, 'ne', False),
       'SuperresolutionHybrid4X': (128, 'lt', True), 'SuperresolutionHybrid2X': (64, 'ne', True)}


def superresolution(P, prefix, rgb, x, ws, force_fp32=True, cpu_rounding=False, sr_class='SuperresolutionHybrid8XDC', aliased_raw=None):
    """tat/superresolution.py:279-290 (SuperresolutionHybrid8XDC.forward; :46-58 8X, :79-91 4X, :113-124 2X), noise_mode='none'; fp32 path (what the
    reference runs off-GPU and what the goldens pin) or, with force_fp32=False (8XDC / 8X), its fp16 blocks emulated (synthesis_block_fp16).
    conv_clamp=256 because the module is built with use_fp16 = sr_num_fp16_res > 0 (:269-275).  Channel counts come from the parameters' shapes."""
    ws = ws[:, -1:, :].repeat(1, 3, 1)
    res_in, rule, noup = _SR[sr_class]
    if (x.shape[-1] != res_in) if rule == 'ne' else (x.shape[-1] < res_in):
        x = F.interpolate(x, size=(res_in, res_in), mode='bilinear', align_corners=False, antialias=True)
        rgb = F.interpolate(rgb, size=(res_in, res_in), mode='bilinear', align_corners=False, antialias=True)
        if aliased_raw is not None:
            aliased_raw['resized'] = True
    if not force_fp32:      # the reference's default on a GPU (use_fp16 = sr_num_fp16_res > 0): emulated float16 storage
        assert not noup, 'float16 emulation of SynthesisBlockNoUp is not restated'
        x, rgb = synthesis_block_fp16(P, f'{prefix}.block0', x, rgb, ws, cpu_rounding=cpu_rounding)
        x, rgb = synthesis_block_fp16(P, f'{prefix}.block1', x, rgb, ws, cpu_rounding=cpu_rounding)
        return rgb
    if noup:
        resized = aliased_raw is not None and aliased_raw.get('resized', False)
        x, rgb = synthesis_block_noup(P, f'{prefix}.block0', x, rgb, ws, noise_mode='none', conv_clamp=256)
        # SynthesisBlockNoUp adds toRGB's output IN PLACE (`img = img.add_(y)`, tat/superresolution.py:250): when the module did not resize, `img` is the caller's
        # `rgb_image = feature_image[:, :3]` view (triplane_next3d.py:185), so the 'image_raw' the reference RETURNS is rgb + toRGB(block0) — reported to the caller here
        if aliased_raw is not None and not resized:
            aliased_raw['image_raw'] = rgb
    else:
        x, rgb = synthesis_block(P, f'{prefix}.block0', x, rgb, ws, P[f'{prefix}.block0.conv0.weight'].shape[1], noise_mode='none', conv_clamp=256)
    x, rgb = synthesis_block(P, f'{prefix}.block1', x, rgb, ws, P[f'{prefix}.block1.conv0.weight'].shape[1], noise_mode='none', conv_clamp=256)
    return rgb
