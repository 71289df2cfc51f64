#!/usr/bin/env python3
"""Per-kernel timing of the float16 super-resolution route at the benchmark's shapes (batch 4): every kernel of
networks.SuperRes8XDC._forward_f16, timed with events on the launch stream; the NBUF / ablation lines need the tuning build (tools/build_tuning.sh, N3D_LIB=)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib, layers as L          # noqa: E402


class Lay:
    def __init__(self, o, i, k, dev):
        self.weight = torch.randn(o, i, k, k, device=dev)
        self.out_channels, self.in_channels, self.ksize = o, i, k
        self.bias = torch.randn(o, device=dev) * 0.1
        self.noise_const = self.noise_strength = None


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    dev = torch.device('cuda:0')
    N = int(os.environ.get('N', 4))
    fir = torch.tensor([1., 3., 3., 1.]); fir = (fir[:, None] * fir[None, :] / 64).to(dev)
    for (i, o, h) in ((32, 256, 128), (256, 128, 256)):
        conv0, conv1, torgb = Lay(o, i, 3, dev), Lay(o, o, 3, dev), Lay(3, o, 1, dev)
        s0, s1, s2 = (torch.randn(N, c, device=dev) + 1 for c in (i, o, o))
        x = _lib.H8.from_nchw(torch.randn(N, i, h, h, device=dev))
        w0, w1 = L.modulate_weights_f16(conv0, s0), L.modulate_weights_f16(conv1, s1)
        print(f'--- block {i}->{o}, {h}^2 -> {2 * h}^2, batch {N}')
        print(f'modulate conv0 {timeit(lambda: L.modulate_weights_f16(conv0, s0)):8.1f} us   conv1 {timeit(lambda: L.modulate_weights_f16(conv1, s1)):8.1f} us'
              f'   torgb {timeit(lambda: L.modulate_weights_f16(torgb, s2, demodulate=False)):8.1f} us')
        gf = 2 * N * o * i * 9 * h * h / 1e9
        t = timeit(lambda: L.conv2d_f16(x, w0, o, 2))
        print(f'transposed: {t:8.1f} us  {gf / t * 1e3:7.1f} TFLOP/s')
        z = L.conv2d_f16(x, w0, o, 2)
        epi = _lib.make_epilogue(bias=conv0.bias, act='lrelu', gain=2 ** 0.5, clamp=256.0)
        by = 2 * N * o * ((2 * h + 1) ** 2 + (2 * h) ** 2)
        for sep in ('0', '1'):
            L.uf.FIR_SEP = sep == '1'
            t = timeit(lambda: L.fir4_h8(z, fir, epi))
            print(f'fir4_h8 separable={sep}: {t:8.1f} us  {by / t / 1e6:6.2f} TB/s')
        L.uf.FIR_SEP = True
        y = L.fir4_h8(z, fir, epi)
        gf = 2 * N * o * o * 9 * 4 * h * h / 1e9
        # tile shapes of the stride-1 kernel (conv2d_f16.hip: F16Tile; tuning build): 0 = 64 ch x 16 x 32 px, 2 = 64 ch x 32 x 32 px
        for wide, what in ((0, '<2,2> 64ch x 16x32'), (2, '<2,4> 64ch x 32x32')):
            os.environ['N3D_F16_WIDE'] = str(wide)
            t = timeit(lambda: L.conv2d_f16(y, w1, o, 0, epi))
            print(f'stride-1 tile {what}: {t:8.1f} us  {gf / t * 1e3:7.1f} TFLOP/s')
            for dbg, abl in ((1, 'no stores'), (4, 'no DMA after chunk 0'), (5, 'neither')):
                os.environ['N3D_CONV_DBG'] = str(dbg)
                t = timeit(lambda: L.conv2d_f16(y, w1, o, 0, epi))
                print(f'   ablation {abl}: {t:8.1f} us  {gf / t * 1e3:7.1f} TFLOP/s')
            del os.environ['N3D_CONV_DBG']
        del os.environ['N3D_F16_WIDE']
        t = timeit(lambda: L.conv2d_f16(y, w1, o, 0, epi))
        print(f'stride-1 library default: {t:8.1f} us  {gf / t * 1e3:7.1f} TFLOP/s')
        b = L.conv2d_f16(y, w1, o, 0, epi)
        wt16 = L.modulate_weights_f16(torgb, s2, demodulate=False)
        t = timeit(lambda: L.conv2d_f16(y, w1, o, 0, epi, rgb=(wt16, 3)))
        print(f'stride-1 with the toRGB fused into its epilogue (no feature-map store): {t:8.1f} us  {gf / t * 1e3:7.1f} TFLOP/s')
        img_lo = torch.randn(N, 3, h, h, device=dev)
        t = timeit(lambda: L.torgb_layer_f16(torgb, b, s2, fir, conv_clamp=256, img_lo=img_lo))
        print(f'torgb_h8 (incl. its weight modulation): {t:8.1f} us  {2 * N * o * 4 * h * h / t / 1e6:6.2f} TB/s')


if __name__ == '__main__':
    main()
