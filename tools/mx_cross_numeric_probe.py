#!/usr/bin/env python3
"""CPU numeric probe of a 2-pass-cheaper operand split for the 3x3 convolutions (not built; DESIGN.md section 7 'next').
Shipped: x = hi + lo with bf16 hi and lo, three bf16 MFMAs per MAC (hi.hi + hi.lo + lo.hi), error 2^-17 class.
Probe:   hi in FLOAT16 (11 bits; same MFMA rate as bf16), so the cross terms only need ~5 bits: both cross operands in OCP fp8 e4m3 on the
         block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 rate) with FIXED power-of-two scales -> 16 instead of 24 MFMA passes per 16 K.
Prints the error of both schemes against float64 on one K = 4608 contraction for activations of a given magnitude."""
import torch
torch.manual_seed(0)


def fp8(t, scale):          # value of fp8_e4m3fn(t * scale) / scale, saturating as the hardware conversion does
    return (t * scale).clamp(-448, 448).to(torch.float8_e4m3fn).float() / scale


def schemes(x, w):
    ref = x.double() @ w.double().T
    xh, wh = x.bfloat16().float(), w.bfloat16().float()
    xl, wl = (x - xh).bfloat16().float(), (w - wh).bfloat16().float()
    a = (xh @ wh.T + xh @ wl.T + xl @ wh.T).double()
    # float16 hi of x * 2^-8 (range to 1.6e7), weights as they are (|w| < 16 here); lo = exact float32 remainder
    xh16 = (x * 2.0 ** -8).half().float() * 2.0 ** 8
    wh16 = w.half().float()
    xl8, wl8 = fp8(x - xh16, 2.0 ** 12), fp8(w - wh16, 2.0 ** 14)
    xh8, wh8 = fp8(x, 1.0), fp8(w, 2.0 ** 4)
    b = (xh16 @ wh16.T + xh8 @ wl8.T + xl8 @ wh8.T).double()
    c = (xh16 @ wh16.T).double()
    den = ref.abs().mean()
    return [float(((v - ref).abs().max()) / den) for v in (a, b, c)]


P, O, K = 4096, 256, 4608
w = torch.randn(O, K) / K ** 0.5 * 3
for name, x in [('N(0,1) lrelu', torch.nn.functional.leaky_relu(torch.randn(P, K), 0.2) * 2 ** 0.5),
                ('x 30', torch.nn.functional.leaky_relu(torch.randn(P, K), 0.2) * 30),
                ('x 0.02', torch.nn.functional.leaky_relu(torch.randn(P, K), 0.2) * 0.02),
                ('heavy tail (1 % of the elements x 1000)', torch.randn(P, K) * torch.where(torch.rand(P, K) < 0.01, 1000.0, 1.0))]:
    a, b, c = schemes(x, w)
    print(f'{name:42s} max error / mean |y|:  split-bf16 x3 {a:.2e}   f16 + 2 fp8 crosses {b:.2e}   f16 alone {c:.2e}')
