"""The generator's convolutional networks assembled from the fused HIP layers (next3d_amd/layers.py).

Each class prepares its per-layer constants once (`PreparedConv`) from the flat parameter dict and then runs
forward passes with no further parameter processing.  Structure follows the reference modules:
  SynthesisNet     training_avatar_texture/networks_stylegan2.py:596-645 (+ SynthesisBlock :492-588, 'skip' arch)
  StyleUNet        training_avatar_texture/networks_stylegan2_styleunet.py:493-588 (+ EncoderResBlock :97-115)
  SuperRes8XDC     training_avatar_texture/superresolution.py:264-290
  MappingNet       training_avatar_texture/networks_stylegan2.py:193-268
"""
import numpy as np
import torch

from . import layers as L
from . import spec as S
from .torch_utils.ops import upfirdn2d as uf


class _Block:
    def __init__(self, P, prefix, in_channels, conv_clamp=None):
        self.in_channels = in_channels
        self.conv_clamp = conv_clamp
        self.const = P.get(f'{prefix}.const')
        self.conv0 = L.PreparedConv(P, f'{prefix}.conv0', modulated=True) if in_channels != 0 else None
        self.conv1 = L.PreparedConv(P, f'{prefix}.conv1', modulated=True)
        self.torgb = L.PreparedConv(P, f'{prefix}.torgb', modulated=True, demodulate=False)

    def entries(self, first_slot):
        """(layer, ws slot, kind) triples for a StyleBank; `first_slot` = ws index of this block's first conv."""
        out, k = [], first_slot
        if self.conv0 is not None:
            out.append((self.conv0, k, 'conv')); k += 1
        out.append((self.conv1, k, 'conv')); k += 1
        out.append((self.torgb, k, 'torgb'))
        return out

    def _ps_base(self, xshape, fir, noise_mode):
        return (self.conv0 is not None and noise_mode != 'random' and fir.ndim == 2 and tuple(fir.shape) == (4, 4) and self.conv0.out_channels % 64 == 0 and
                self.conv0.wt16 is not None and L.PRECISION == 'bf16x3' and L.cg.bf16x3_eligible(xshape[1], xshape[2], xshape[3], 3, 2))

    def _presplit(self, n, xshape, fir, noise_mode):
        """conv0 on the pre-split transposed kernel (c8 result) and its FIR handing the output to conv1 pre-split, with conv1's styles
        multiplied in (layers.presplit_ok).  Layers the register-staged kernel would run with split-K take this route from 64 x 64
        up (measured at batch 1, 64 x 64: 84 us against 46 + a 12 us conversion, tools/ps_splitk_bench.py)."""
        return (self._ps_base(xshape, fir, noise_mode) and
                (L.cg.pick_ksplit_bf16x3(n, xshape[1], self.conv0.out_channels, xshape[2], xshape[3], 2) == 1 or xshape[2] * xshape[3] >= 4096) and
                L.presplit_ok(n, self.conv1, 2 * xshape[2], 2 * xshape[3]))

    def _ps_nchw(self, n, xshape, fir, noise_mode):
        """conv0 on the pre-split transposed kernel writing float32 NCHW: the few-position layers (conv1 not a pre-split layer) with at
        least 1024 positions in the batch — 16 x 16 at batch 4: 43 us against 58 for the register-staged kernel + split-K reduction."""
        return bool(L.UP_PRESPLIT and L.UP_PS_NCHW and xshape[1] % 16 == 0 and self._ps_base(xshape, fir, noise_mode) and
                    not self._presplit(n, xshape, fir, noise_mode) and n * xshape[2] * xshape[3] >= 1024 and
                    not L.cg.up_sk_eligible(n, xshape[1], self.conv0.out_channels, xshape[2], xshape[3]))

    def takes_split8(self, n, xshape, fir, noise_mode):
        """Does conv0 read its [n, I, h, w] input in the split8 layout (layers.synthesis_layer, the transposed pre-split kernel)?"""
        return bool(L.UP_PRESPLIT and xshape[1] % 16 == 0 and self._presplit(n, xshape, fir, noise_mode)) or self._ps_nchw(n, xshape, fir, noise_mode)

    def __call__(self, x, img, bank, n, fir, noise_mode, x_out=None, x_split8=None, next_block=None, last=False):
        """SynthesisBlock.forward (float32 block); `bank` = StyleBank.compute(ws) result.  -> (x, img, xs): `next_block` = the block
        whose conv0 reads THIS block's x unchanged; when it takes split8 input, toRGB writes it on the side (layers.torgb_layer) and
        `xs` is to be passed to that block as `x_split8` (None otherwise).  `last`: nobody reads this block's x but its own toRGB — where the
        kernels allow (layers.fused_torgb_ok) conv1 evaluates toRGB in its epilogue and x is never written: -> (None, img, None)."""
        sl = lambda layer: dict(zip(('styles', 'dcoef'), bank[layer.prefix]))
        if self.in_channels == 0:
            x = self.const.unsqueeze(0).expand(n, -1, -1, -1)
            x = L.synthesis_layer(self.conv1, x, None, fir, noise_mode=noise_mode, conv_clamp=self.conv_clamp, **sl(self.conv1))
        else:
            if x is None:                  # the previous block never wrote its x in float32 (fused toRGB + split8 side output): only x_split8 exists
                x = x_split8
            pre = self._presplit(n, x.shape, fir, noise_mode)
            # ... or, where conv0 runs on the register-staged transposed kernel (few positions, split-K), at least its FIR writes split8
            pre_nchw = (not pre and L.NCHW_FIR_SPLIT8 and noise_mode != 'random' and fir.ndim == 2 and tuple(fir.shape) == (4, 4) and
                        self.conv0.out_channels % 8 == 0 and L.PRECISION == 'bf16x3' and self.conv0.wt16 is not None and
                        L.cg.bf16x3_eligible(x.shape[1], x.shape[2], x.shape[3], 3, 2) and L.presplit_ok(n, self.conv1, 2 * x.shape[2], 2 * x.shape[3]))
            psn = self._ps_nchw(n, x.shape, fir, noise_mode)
            x = L.synthesis_layer(self.conv0, x, None, fir, up=2, noise_mode=noise_mode, conv_clamp=self.conv_clamp,
                                  split_for=bank[self.conv1.prefix][0] if pre else None, x_split8=x_split8 if (pre or psn) else None,
                                  split_for_nchw=bank[self.conv1.prefix][0] if pre_nchw else None, ps_nchw=psn, **sl(self.conv0))
            # x's only readers are this block's toRGB (<= 4 colours) and, unless `last`, the next block's transposed convolution on split8 input:
            # conv1 evaluates the toRGB in its epilogue and writes that operand image itself — x is never written in float32 (layers.fused_torgb_ok)
            nb_split8 = next_block is not None and self.conv1.out_channels % 8 == 0 and next_block.takes_split8(n, (n, self.conv1.out_channels, x.shape[2], x.shape[3]), fir, noise_mode)
            if (last or (nb_split8 and (L.FUSED_TORGB_MID or self.torgb.out_channels <= 4))) and x_out is None and L.fused_torgb_ok(self.conv1, self.torgb, x, noise_mode):
                t = self.torgb
                part = L.synthesis_layer(self.conv1, x, None, fir, noise_mode=noise_mode, conv_clamp=self.conv_clamp,
                                         rgb=(t.weight.reshape(t.out_channels, t.in_channels), bank[t.prefix][0]),
                                         rgb_side_style=None if last else bank[next_block.conv0.prefix][0], **sl(self.conv1))
                part, xs = (part, None) if last else part
                up = fir if (img is not None and fir.ndim == 2 and tuple(fir.shape) == (4, 4)) else None
                if img is not None and up is None:
                    img = uf.upsample2d(img, fir)
                return None, L.torgb_combine(t, part, conv_clamp=self.conv_clamp, residual=img, residual_up_filter=up), xs
            x = L.synthesis_layer(self.conv1, x, None, fir, noise_mode=noise_mode, conv_clamp=self.conv_clamp, out=x_out, **sl(self.conv1))
        # skip-image update img = upsample2d(img) + toRGB(x): upsample2d is evaluated inside the toRGB epilogue (4 taps of the
        # half-resolution image per pixel)
        up = fir if (img is not None and fir.ndim == 2 and tuple(fir.shape) == (4, 4)) else None
        if img is not None and up is None:
            img = uf.upsample2d(img, fir)
        side = (bank[next_block.conv0.prefix][0] if (next_block is not None and L.torgb_side_ok(self.torgb, x) and
                                                      next_block.takes_split8(n, x.shape, fir, noise_mode)) else None)
        img = L.torgb_layer(self.torgb, x, None, conv_clamp=self.conv_clamp, residual=img, styles=bank[self.torgb.prefix][0],
                            residual_up_filter=up, side_style=side)
        if side is not None:
            img, xs = img
            return x, img, xs
        return x, img, None


def _f16_block(blk, xh, img, fir, noise_mode, w16):
    """SynthesisBlock.forward with use_fp16 and not force_fp32 (tat/networks_stylegan2.py:544-588, dtype float16) on the f16 kernels:
    xh `_lib.H8` [N,I,h,w] -> (`_lib.H8` [N,O,2h,2w], float32 skip image).  w16 = the block's per-sample weights (conv0, conv1, torgb)."""
    w0, w1, wt = w16
    xh = L.synthesis_layer_f16(blk.conv0, xh, None, fir, up=2, noise_mode=noise_mode, conv_clamp=blk.conv_clamp, w16=w0)
    xh = L.synthesis_layer_f16(blk.conv1, xh, None, fir, up=1, noise_mode=noise_mode, conv_clamp=blk.conv_clamp, w16=w1)
    img = L.torgb_layer_f16(blk.torgb, xh, None, fir, conv_clamp=blk.conv_clamp, img_lo=img, w16=wt)
    return xh, img


def _f16_block_ok(r, b):
    return (b.conv0 is not None and L.f16_layer_ok(b.conv0, r // 2, r // 2, 2) and L.f16_layer_ok(b.conv1, r, r, 1) and
            b.torgb.in_channels <= 512 and b.torgb.in_channels % 8 == 0)


def _f16_eligible(blocks, fir, noise_mode, owner, what):
    """Which of the reference's float16 blocks (res -> _Block, each fed an [*, I, res/2, res/2] input) run on the f16 kernels: the
    highest-resolution run of eligible blocks (per block, not all-or-nothing).  The blocks below it that the kernels do not
    take — 8 x 8 / 16 x 16 blocks with num_fp16_res >= 5: the stride-1 f16 kernel needs W >= 32 — run in float32 (a superset in accuracy of
    the reference's float16 arithmetic, NOT its rounding: documented in INTEGRATION.md), with a one-time warning naming them.  A non-4x4 filter
    rules the f16 kernels out altogether."""
    if not blocks:
        return {}
    ok = {}
    if tuple(fir.shape) == (4, 4):                                  # (random noise: the noisy layers run sample by sample, layers.synthesis_layer_f16)
        for r in sorted(blocks, reverse=True):
            if not _f16_block_ok(r, blocks[r]):
                break
            ok[r] = blocks[r]
    if len(ok) < len(blocks):
        _warn_f16_once(owner, what, sorted(set(blocks) - set(ok)))
    return ok


def _f16_weights(blocks, bank, n):
    """Per-sample float16 weights of every layer of `blocks` (res -> _Block): n3d_modulate_weights_f16_multi, 8 layers per launch
    -> {res: (w_conv0, w_conv1, w_torgb)}."""
    st = lambda layer: bank[layer.prefix][0]
    jobs = [(l, st(l), k == 'conv') for r in sorted(blocks) for (l, _, k) in blocks[r].entries(0)]
    base = jobs[0][1]
    base = base._base if base._base is not None else base            # StyleBank's packed [N, total] styles buffer
    flat = []
    for a in range(0, len(jobs), 8):
        flat += L.modulate_weights_f16_multi(jobs[a:a + 8], base, n)
    return {r: tuple(flat[3 * k:3 * k + 3]) for k, r in enumerate(sorted(blocks))}


def _warn_f16_once(obj, what, which=None):
    import warnings
    if not getattr(obj, '_warned32', False):
        obj._warned32 = True
        blocks = 'the float16 blocks' if which is None else 'float16 blocks ' + ', '.join(f'b{r}' for r in which)
        warnings.warn(f'{blocks} of {what} are not available on the f16 kernels for this configuration (filter / shapes): running '
                      'them in float32 (the force_fp32=True arithmetic); the other float16 blocks stay float16')


def _first_slots(block_resolutions):
    """ws index of each block's first conv: b4 has one conv, the others two (networks_stylegan2.py:632-640)."""
    out, idx = {}, 0
    for res in block_resolutions:
        out[res] = idx
        idx += 1 if res == 4 else 2
    return out


def _ws3(ws):
    ws = ws.to(torch.float32)
    return ws if (ws.stride(2) == 1 and ws.stride(1) == ws.shape[2]) else ws.contiguous()


def _channels_of(P, prefix, img_resolution):
    """resolution -> channels of the network's blocks, read off the weights (the reference's channels_dict, tat/networks_stylegan2.py:614:
    min(channel_base // res, channel_max) — 32768 / 512 in the ffhq-512 pickle, any other `--cbase` / `--cmax` of train_next3d.py:199-200 arrives here as shapes)."""
    return {r: int(P[f'{prefix}.b{r}.conv1.weight'].shape[0]) for r in sorted(S.channels_dict(img_resolution))}


class SynthesisNet:
    def __init__(self, P, prefix, img_resolution=256, fp16_resolution=None, conv_clamp=None):
        """fp16_resolution: blocks of that resolution and up are the reference's float16 blocks (`num_fp16_res` > 0: fp16_resolution =
        max(2 ** (log2(img_resolution) + 1 - num_fp16_res), 8), tat/networks_stylegan2.py:615-621; legacy.load_network_pkl(force_fp16=True)
        sets num_fp16_res = 4, conv_clamp = 256, legacy.py:49-59); conv_clamp applies to every block."""
        self.cd = _channels_of(P, prefix, img_resolution)
        self.block_res = sorted(self.cd)
        self.prefix, self.fp16_resolution = prefix, fp16_resolution
        self.blocks = {r: _Block(P, f'{prefix}.b{r}', self.cd[r // 2] if r > 4 else 0, conv_clamp=conv_clamp) for r in self.block_res}
        self.fir = P[f'{prefix}.b4.resample_filter']
        self.num_ws = 2 * len(self.block_res)
        slots = _first_slots(self.block_res)
        self.bank = L.StyleBank(sum((self.blocks[r].entries(slots[r]) for r in self.block_res), []), self.fir.device)
        uf.fir_factor(self.fir)            # the one host read of the filter happens here, at model preparation

    def __call__(self, ws, noise_mode='const', bank=None, force_fp32=False):
        """`bank`: this network's styles / demodulation coefficients when the caller computed them already (generator: one
        StyleBank over all five networks, two launches per forward instead of ten).  force_fp32: SynthesisBlock.forward's flag
        (tat/networks_stylegan2.py:544-550) — the float16 blocks, if any, run in float32."""
        from . import _lib
        ws = _ws3(ws)
        if bank is None:
            bank = self.bank.compute(ws)
        f16 = _f16_eligible({r: self.blocks[r] for r in self.block_res if self.fp16_resolution and r >= self.fp16_resolution and not force_fp32},
                            self.fir, noise_mode, self, self.prefix)
        w16 = _f16_weights(f16, bank, ws.shape[0]) if f16 else {}
        x = img = xs = xh = None
        for k, res in enumerate(self.block_res):
            if res in f16:
                if xh is None:
                    xh = _lib.H8.from_nchw(x)                 # x.to(float16) at the first float16 block's entry (:560-562)
                xh, img = _f16_block(self.blocks[res], xh, img, self.fir, noise_mode, w16[res])
                continue
            nxt = self.blocks[self.block_res[k + 1]] if (k + 1 < len(self.block_res) and self.block_res[k + 1] not in f16) else None
            x, img, xs = self.blocks[res](x, img, bank, ws.shape[0], self.fir, noise_mode, x_split8=xs, next_block=nxt, last=(k + 1 == len(self.block_res)))
        return img


_CAT_COPY = False          # True: concatenate by copy (torch.cat) instead of writing both halves in place — A/B and tests


class _EncoderBlock:
    def __init__(self, P, prefix, downsample):
        self.fromrgb = L.PreparedConv(P, f'{prefix}.fromrgb', modulated=False)
        self.conv1 = L.PreparedConv(P, f'{prefix}.conv1', modulated=False)
        self.conv2 = L.PreparedConv(P, f'{prefix}.conv2', modulated=False)
        self.downsample = downsample

    def __call__(self, inp, skip, fir, out_buf=None):
        if self.downsample:
            inp = uf.downsample2d(inp, fir)
        out = L.conv2d_layer(self.fromrgb, inp, fir, activation='linear', residual=skip, sole_consumer=self.conv1)
        out = L.conv2d_layer(self.conv1, out, fir, activation='lrelu')
        out = L.conv2d_layer(self.conv2, out, fir, activation='lrelu', down=2, out=out_buf)
        return inp, out


class StyleUNet:
    def __init__(self, P, prefix, img_resolution=256, in_size=64, final_size=4, num_cond_res=64, fp16_resolution=None, conv_clamp=None):
        self.prefix, self.fp16_resolution = prefix, fp16_resolution
        self.cd = _channels_of(P, prefix, img_resolution)
        self.block_res = sorted(self.cd)
        self.final_log2 = int(np.log2(final_size))
        self.num_cond_res = num_cond_res
        self.start = self.final_log2 - 1
        enc_res = [2 ** i for i in range(int(np.log2(in_size)), self.final_log2 - 1, -1)]
        self.encoder = [_EncoderBlock(P, f'{prefix}.encoder.{i}', downsample=(r < in_size)) for i, r in enumerate(enc_res[:-1])]
        self.used_res = self.block_res[self.start:]
        self.blocks = {r: _Block(P, f'{prefix}.b{r}', self.cd[r // 2], conv_clamp=conv_clamp) for r in self.used_res}
        n_fusion = sum(1 for i in range(len(self.used_res)) if 2 ** (i + self.final_log2) < num_cond_res)
        self.fusion = [L.PreparedConv(P, f'{prefix}.fusion.{i}', modulated=False) for i in range(n_fusion)]
        self.fir = P[f'{prefix}.b4.resample_filter']
        slots = _first_slots(self.block_res)
        self.bank = L.StyleBank(sum((self.blocks[r].entries(slots[r]) for r in self.used_res), []), self.fir.device)
        uf.fir_factor(self.fir)

    def __call__(self, x_in, ws, noise_mode='const', bank=None, force_fp32=False):
        from . import _lib
        ws = _ws3(ws)
        if bank is None:
            bank = self.bank.compute(ws)
        f16 = _f16_eligible({r: self.blocks[r] for r in self.used_res if self.fp16_resolution and r >= self.fp16_resolution and not force_fp32},
                            self.fir, noise_mode, self, self.prefix)
        w16 = _f16_weights(f16, bank, ws.shape[0]) if f16 else {}
        cat_copy = _CAT_COPY or bool(f16)          # float16 blocks hand their feature map over as h8: the concatenation is a copy then
        # The decoder concatenates its feature map with the encoder's condition before every fusion conv (reference
        # networks_stylegan2_styleunet.py:565-567: torch.cat([x, conds[idx]], 1)).  Both halves are WRITTEN IN PLACE into one
        # buffer by the convolutions that produce them (channel-slice views: the kernels take a batch stride), so no
        # concatenation pass over 2 x (x + cond) bytes runs.
        n, n_enc = x_in.shape[0], len(self.encoder)
        cat = {}                                                   # fusion index -> [N, Cx + Cc, H, W]
        conds, cond = [None] * n_enc, None
        h = x_in.shape[2]
        for i, enc in enumerate(self.encoder):
            idx = n_enc - 1 - i                                    # position of this encoder block's output in conds[::-1]
            hin = h // 2 if enc.downsample else h
            out_buf = None
            if 1 <= idx < len(self.fusion) and not cat_copy:
                cx, cc = self.cd[self.used_res[idx - 1]], enc.conv2.out_channels
                cat[idx] = torch.empty(n, cx + cc, hin // 2, hin // 2, dtype=torch.float32, device=x_in.device)
                out_buf = cat[idx][:, cx:]
            x_in, cond = enc(x_in, cond, self.fir, out_buf)
            conds[idx] = cond
            h = hin
        x = img = xs = xh = None
        for idx, res in enumerate(self.used_res):
            if idx < len(self.fusion):
                xs = None                                             # this block reads the fusion layer's output, not the previous x
                if idx == 0:
                    x = L.conv2d_layer(self.fusion[0], conds[0], self.fir, activation='linear')
                else:
                    if xh is not None:         # torch.cat([x (float16), cond (float32)]) promotes to float32 (styleunet.py:565-567)
                        x, xh = xh.to_nchw(), None
                    x = L.conv2d_layer(self.fusion[idx], cat[idx] if idx in cat else torch.cat([x, conds[idx]], dim=1), self.fir, activation='linear')
            if res in f16:
                if xh is None:
                    xh = _lib.H8.from_nchw(x)
                xh, img = _f16_block(self.blocks[res], xh, img, self.fir, noise_mode, w16[res])
                continue
            nxt = cat.get(idx + 1)
            x_out = nxt[:, :self.cd[res]] if (nxt is not None and nxt.shape[2] == res) else None
            # the next block reads this x directly unless a fusion layer (or the concatenation buffer's channel-slice view) is in between
            nb = self.blocks[self.used_res[idx + 1]] if (idx + 1 < len(self.used_res) and idx + 1 >= len(self.fusion) and x_out is None and
                                                         self.used_res[idx + 1] not in f16) else None
            x, img, xs = self.blocks[res](x, img, bank, ws.shape[0], self.fir, noise_mode, x_out=x_out, x_split8=xs, next_block=nb,
                                          last=(idx + 1 == len(self.used_res)))
            if nxt is not None and x_out is None:                 # shapes did not line up: fall back to a copy
                nxt[:, :self.cd[res]].copy_(x)
        return img


class _BlockNoUp(_Block):
    """SynthesisBlockNoUp (tat/superresolution.py:158-254: the first block of SuperresolutionHybrid4X / 2X): conv0 WITHOUT up-sampling, conv1, toRGB; the skip
    image is added as it is (the upsample2d of SynthesisBlock.forward is commented out there, :244-246).  Plain layer calls: these two modules are outside the
    benchmarked configuration, what matters is that the reference's architectures run and match."""

    def takes_split8(self, n, xshape, fir, noise_mode):
        return False

    def __call__(self, x, img, bank, n, fir, noise_mode, x_out=None, x_split8=None, next_block=None, last=False):
        sl = lambda layer: dict(zip(('styles', 'dcoef'), bank[layer.prefix]))
        x = L.synthesis_layer(self.conv0, x, None, fir, up=1, noise_mode=noise_mode, conv_clamp=self.conv_clamp, **sl(self.conv0))
        x = L.synthesis_layer(self.conv1, x, None, fir, up=1, noise_mode=noise_mode, conv_clamp=self.conv_clamp, out=x_out, **sl(self.conv1))
        img = L.torgb_layer(self.torgb, x, None, conv_clamp=self.conv_clamp, residual=img, styles=bank[self.torgb.prefix][0])
        return x, img, None


class SuperRes8XDC:
    """The reference's super-resolution modules (tat/superresolution.py; spec.SR_MODULES): SuperresolutionHybrid8XDC — the ffhq-512 configuration, the class
    name this one keeps — and, round 5, SuperresolutionHybrid8X (other channel counts), 4X and 2X (a SynthesisBlockNoUp first, 256 x 256 / 128 x 128 output)."""

    def __init__(self, P, prefix, conv_clamp=256, sr_class=S.DEFAULT_SR):
        """conv_clamp: 256 when the model was built with sr_num_fp16_res > 0, else None (superresolution.py:273-278)."""
        _, self.input_resolution, self.resize_rule, blocks, _ = S.SR_MODULES[sr_class]
        mk = lambda bi: (_Block if blocks[bi][0] == 'up' else _BlockNoUp)(P, f'{prefix}.block{bi}', blocks[bi][1], conv_clamp=conv_clamp)
        self.block0, self.block1 = mk(0), mk(1)
        self.all_up = all(b[0] == 'up' for b in blocks)
        self.fir = P[f'{prefix}.block0.resample_filter']
        self._banks = {}
        uf.fir_factor(self.fir)

    def _f16_ok(self, x, noise_mode):
        """Can the reference's float16 blocks run on the f16 kernels for this input (conv2d_f16.hip)?  When not (a
        non-4x4 resampling filter, odd shapes) they run in float32 — the force_fp32=True arithmetic, a superset in accuracy — with
        a one-time warning, never an error: the reference's default `synthesis(ws, c, v)` call must run."""
        import warnings
        h = x.shape[2]
        ok = (self.all_up and noise_mode in ('random', 'const', 'none') and tuple(self.fir.shape) == (4, 4) and      # (a SynthesisBlockNoUp has no float16 form here: 4X / 2X run in float32)
              all(L.f16_layer_ok(b.conv0, hh, hh, 2) and L.f16_layer_ok(b.conv1, 2 * hh, 2 * hh, 1) and b.torgb.in_channels <= 512
                  for b, hh in ((self.block0, h), (self.block1, 2 * h))))
        if not ok and not getattr(self, '_warned32', False):
            self._warned32 = True
            why = ('this super-resolution module has a block without up-sampling (SynthesisBlockNoUp: SuperresolutionHybrid4X / 2X), which has no float16 form here'
                   if not self.all_up else 'resampling filter / layer shapes outside the f16 kernels')
            warnings.warn(f'float16 super-resolution blocks are not available for this configuration ({why}): '
                          'running the whole module in float32 (the force_fp32=True arithmetic)')
        return ok

    def _forward_f16(self, x, rgb, bank, noise_mode):
        """Both blocks on the f16 kernels (layers.synthesis_layer_f16 / torgb_layer_f16): h8 activations, float32 skip image."""
        from . import _lib
        xh = _lib.H8.from_nchw(x)                                   # x.to(float16) at the block entry (networks_stylegan2.py:437)
        st = lambda layer: bank[layer.prefix][0]
        layers6 = [(l, st(l), k == 'conv') for blk in (self.block0, self.block1) for (l, _, k) in blk.entries(0)]
        base = st(self.block0.conv0)
        base = base._base if base._base is not None else base       # StyleBank's packed [N, total] styles buffer
        w16 = L.modulate_weights_f16_multi(layers6, base, x.shape[0])           # all six layers' per-sample weights in one launch
        for bi, blk in enumerate((self.block0, self.block1)):
            w0, w1, wt = w16[3 * bi:3 * bi + 3]
            xh = L.synthesis_layer_f16(blk.conv0, xh, None, self.fir, up=2, noise_mode=noise_mode, conv_clamp=blk.conv_clamp, w16=w0)
            if bi == 1 and L.FUSED_TORGB and blk.torgb.out_channels <= 4:
                # the last layer's only reader is its toRGB: evaluated in conv1's epilogue, the 512 x 512 x 128-channel map is never written
                part = L.synthesis_layer_f16(blk.conv1, xh, None, self.fir, up=1, noise_mode=noise_mode, conv_clamp=blk.conv_clamp, w16=w1,
                                             rgb=(wt, blk.torgb.out_channels))
                return L.torgb_combine_f16(blk.torgb, part, self.fir, conv_clamp=blk.conv_clamp, img_lo=rgb)
            xh = L.synthesis_layer_f16(blk.conv1, xh, None, self.fir, up=1, noise_mode=noise_mode, conv_clamp=blk.conv_clamp, w16=w1)
            rgb = L.torgb_layer_f16(blk.torgb, xh, None, self.fir, conv_clamp=blk.conv_clamp, img_lo=rgb, w16=wt)
        return rgb

    def bank_entries(self, last):
        """StyleBank entries of both blocks, all reading latent slot `last`."""
        return [(l, last, k) for blk in (self.block0, self.block1) for (l, _, k) in blk.entries(0)]

    def __call__(self, rgb, x, ws, resize_fn, noise_mode='none', fp16=False, bank=None):
        """`resize_fn(t, size)` = antialiased bilinear resize (superresolution.py:282-286).  Every layer is driven by the
        LAST latent of `ws` (`ws[:, -1:].repeat(1, 3, 1)`, :280): all StyleBank jobs read that one slot."""
        ws = _ws3(ws)
        last = ws.shape[1] - 1
        if bank is None:
            if last not in self._banks:
                self._banks[last] = L.StyleBank(self.bank_entries(last), self.fir.device)
            bank = self._banks[last].compute(ws)
        self.aliased_raw = None
        resized = (x.shape[-1] != self.input_resolution) if self.resize_rule == 'ne' else (x.shape[-1] < self.input_resolution)     # (4X resizes only a SMALLER render, :82)
        if resized:
            x = resize_fn(x, self.input_resolution)
            rgb = resize_fn(rgb, self.input_resolution)
        if not rgb.is_contiguous():                                  # (no resize: the caller's channel-slice view of the feature image)
            rgb = rgb.contiguous()
        if fp16 and self._f16_ok(x, noise_mode):
            return self._forward_f16(x, rgb, bank, noise_mode)
        x0, rgb, xs = self.block0(x, rgb, bank, ws.shape[0], self.fir, noise_mode, next_block=self.block1)
        if isinstance(self.block0, _BlockNoUp) and not resized:
            # SynthesisBlockNoUp adds toRGB's output IN PLACE (`img = img.add_(y)`, tat/superresolution.py:250) and without a resize `img` IS the caller's
            # `rgb_image = feature_image[:, :3]` view (triplane_next3d.py:185): the 'image_raw' the reference returns is rgb + toRGB(block0) — handed to synthesis()
            # (pinned: oracle/pin_against_reference.py --sr-noresize, 0.0 against the reference)
            self.aliased_raw = rgb
        x1, rgb, _ = self.block1(x0, rgb, bank, ws.shape[0], self.fir, noise_mode, x_split8=xs, last=True)
        return rgb
