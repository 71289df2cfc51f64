// Shared host/device helpers for libn3d.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/n3d.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libn3d.so kernels are written for gfx950 only (160 KB of LDS per workgroup, v_mfma_f32_32x32x16_bf16 / _f16): build with --offload-arch=gfx950"
#endif

// ---- host-side error + profiling plumbing (runtime.hip)
int n3d_set_error(const char* fmt, ...);
struct N3dProfScope {          // brackets the launches of one entry point with events on the launch stream
    N3dProfScope(int family, hipStream_t s, double flops, double bytes);
    ~N3dProfScope();
    int slot;
    hipStream_t stream;
};

#define N3D_CHECK(cond, ...)                         \
    do {                                             \
        if (!(cond)) return n3d_set_error(__VA_ARGS__); \
    } while (0)

#define N3D_LAUNCH_CHECK()                                                              \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) return n3d_set_error("HIP launch failed: %s", hipGetErrorString(e_)); \
    } while (0)

// Tuning switches (ablation bits, kernel-variant selectors) exist in tuning builds only (tools/build_variant.sh <tag> <file> -DN3D_TUNING):
// the shipped library reads no environment variable on its launch paths.
#ifdef N3D_TUNING
#include <stdlib.h>
static inline int n3d_tune(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline int n3d_tune(const char*, int dflt) { return dflt; }
#endif

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device-side: activation + the fused epilogue (n3d_epilogue in n3d.h)
__device__ __forceinline__ float n3d_act(float x, int act, float alpha) {
    // forward formulas of bias_act.cu:27-150 / bias_act.py:23-33
    switch (act) {
        case N3D_ACT_RELU: return x > 0.f ? x : 0.f;
        case N3D_ACT_LRELU: return x > 0.f ? x : x * alpha;
        case N3D_ACT_TANH: return tanhf(x);
        case N3D_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        case N3D_ACT_ELU: return x > 0.f ? x : expf(x) - 1.f;
        case N3D_ACT_SELU: {
            const float s = 1.0507009873554804934193349852946f, a = 1.6732632423543772848170429916717f;
            return x > 0.f ? s * x : s * a * (expf(x) - 1.f);
        }
        case N3D_ACT_SOFTPLUS: return x > 20.f ? x : log1pf(expf(x));
        case N3D_ACT_SWISH: return x / (1.f + expf(-x));
        default: return x;
    }
}

// x2 upsampling of a low-resolution residual on the fly (n3d_epilogue.residual_up_filter): upfirdn2d with up=2, padding
// (2,1,2,1), 4x4 taps, gain 4 (upfirdn2d.py:upsample2d).  Output o takes the inputs iA = floor((o-1)/2) and iA+1 with
// taps kA = 1 - 2 iA + o and kA - 2 (index arithmetic of upfirdn2d.cu:47-67 with up=2, pad0=2, down=1).
struct n3d_up2_taps {
    int off[4];      // offsets into the low-res plane (0 where the tap is outside the image)
    float w[4];      // tap weights, gain included (0 outside)
};
__device__ __forceinline__ n3d_up2_taps n3d_up2_setup(const float* __restrict__ f, int oy, int ox, int LH, int LW) {
    n3d_up2_taps t;
    const int iyA = (oy - 1) >> 1, ixA = (ox - 1) >> 1;
    const int kyA = 1 - 2 * iyA + oy, kxA = 1 - 2 * ixA + ox;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int iy = iyA + a, ix = ixA + b;
            const bool ok = iy >= 0 && iy < LH && ix >= 0 && ix < LW;
            t.off[a * 2 + b] = ok ? iy * LW + ix : 0;
            t.w[a * 2 + b] = ok ? 4.f * f[(kyA - 2 * a) * 4 + (kxA - 2 * b)] : 0.f;
        }
    return t;
}
__device__ __forceinline__ float n3d_up2_apply(const n3d_up2_taps& t, const float* __restrict__ plane) {
    return ((plane[t.off[0]] * t.w[0] + plane[t.off[1]] * t.w[1]) + plane[t.off[2]] * t.w[2]) + plane[t.off[3]] * t.w[3];
}

__device__ __forceinline__ float n3d_round16(float v, int on) { return on ? (float)(_Float16)v : v; }

__device__ __forceinline__ float n3d_apply_epilogue(float v, const n3d_epilogue& e, int n, int o, int O, int oy, int ox,
                                                    int OH, int OW) {
    float sc = e.const_scale;
    if (e.row_scale) sc *= e.row_scale[(int64_t)n * (e.row_scale_stride ? e.row_scale_stride : O) + o];
    v *= sc;
    if (e.noise) v += e.noise[(int64_t)oy * OW + ox] * e.noise_strength[0];
    if (e.bias) v += e.bias[o];
    v = n3d_act(v, e.act, e.alpha) * e.gain;
    if (e.clamp >= 0.f) v = fminf(fmaxf(v, -e.clamp), e.clamp);
    if (e.round_f16) v = (float)(_Float16)v;
    if (e.residual) {
        if (e.residual_up_filter) {
            const int LH = OH >> 1, LW = OW >> 1;
            const n3d_up2_taps t = n3d_up2_setup(e.residual_up_filter, oy, ox, LH, LW);
            v += n3d_up2_apply(t, e.residual + (int64_t)n * e.residual_batch_stride + (int64_t)o * LH * LW);
        } else {
            v += e.residual[(int64_t)n * e.residual_batch_stride + ((int64_t)o * OH + oy) * OW + ox];
        }
    }
    return v;
}
