// The non-convolution kernels of the reference's FLOAT16 blocks (super-resolution default route; conv2d_f16.hip has the 3x3
// convolutions): per-sample weight modulation, the FIR behind the transposed convolution, toRGB + skip-image accumulation, and
// the float32 -> h8 conversion at the block entry.  Every tensor the reference holds in float16 is held in float16 here, every
// operator computes in float32 on the float16 inputs and rounds ONCE on output — what the reference's CUDA plugins do
// (upfirdn2d.cu / bias_act.cu: `typedef typename InternalType<T>::scalar_t scalar_t` = float for half tensors).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------------------------------------
// Per-sample modulated (and demodulated) float16 weights = modulated_conv2d's fused branch (training/networks_stylegan2.py:53-66,
// :88 `w.to(x.dtype)`), operation for operation in float32:
//   demodulate:  wn = weight * ((1 / max|weight[o]|) * c),  c = float(1 / sqrt(I k k))          (:55, scalar / tensor = reciprocal * scalar)
//                sn = styles[n] / max|styles[n]|                                                  (:56)
//                wm = wn * sn[i];  d = 1 / sqrt(sum_{i,k} wm^2 + 1e-8);  w16 = half(wm * d)       (:61-66)
//   otherwise :  w16 = half(weight * styles[n, i])                                                (toRGB: demodulate = False)
// Output layouts: ksize 3 -> [N][tap][I/16][2][O][8] (what conv2d_f16.hip streams), ksize 1 -> [N][O][I].
// One workgroup per (o, n); the I k k <= 4608 products of a row stay in registers between the reduction and the store.
struct ModwParams {
    const float* w; const float* styles; _Float16* out;
    int N, O, I, K2;             // K2 = ksize^2
    int64_t sstride;
    int demod;
    float c;
};

template <int EPT>
__device__ __forceinline__ void modulate_row(const ModwParams& p, int o, int n) {
    __shared__ float s_red[8];
    const int tid = threadIdx.x;
    const int L = p.I * p.K2;
    const float* wrow = p.w + (int64_t)o * L;
    const float* srow = p.styles + (int64_t)n * p.sstride;
    auto block_reduce = [&](float v, bool is_max) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float u = __shfl_xor(v, off);
            v = is_max ? fmaxf(v, u) : v + u;
        }
        __syncthreads();
        if ((tid & 63) == 0) s_red[tid >> 6] = v;
        __syncthreads();
        float r = s_red[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) r = is_max ? fmaxf(r, s_red[k]) : r + s_red[k];
        return r;
    };
    float wv[EPT], sv[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int e = tid + 256 * j;
        wv[j] = e < L ? wrow[e] : 0.f;
        sv[j] = e < L ? srow[e / p.K2] : 0.f;
    }
    float dco = 1.f;
    if (p.demod) {
        float wmax = 0.f, smax = 0.f;
#pragma unroll
        for (int j = 0; j < EPT; ++j) wmax = fmaxf(wmax, fabsf(wv[j]));
        wmax = block_reduce(wmax, true);
        for (int i = tid; i < p.I; i += 256) smax = fmaxf(smax, fabsf(srow[i]));
        smax = block_reduce(smax, true);
        const float wscale = (1.f / wmax) * p.c;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            wv[j] = (wv[j] * wscale) * (sv[j] / smax);
            ss += wv[j] * wv[j];
        }
        ss = block_reduce(ss, false);
        dco = 1.f / sqrtf(ss + 1e-8f);
    } else {
#pragma unroll
        for (int j = 0; j < EPT; ++j) wv[j] = wv[j] * sv[j];
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int e = tid + 256 * j;
        if (e >= L) continue;
        const _Float16 h = (_Float16)(p.demod ? wv[j] * dco : wv[j]);
        const int i = e / p.K2, tap = e % p.K2;
        if (p.K2 == 9) {
            const int64_t unit = (((int64_t)(n * 9 + tap) * (p.I / 16) + (i >> 4)) * 2 + ((i >> 3) & 1)) * p.O + o;
            p.out[unit * 8 + (i & 7)] = h;
        } else {
            p.out[((int64_t)n * p.O + o) * p.I + i] = h;
        }
    }
}
template <int EPT>
__global__ __launch_bounds__(256) void modulate_weights_f16_kernel(ModwParams p) { modulate_row<EPT>(p, blockIdx.x, blockIdx.y); }

extern "C" int n3d_modulate_weights_f16(const float* w, const float* styles, int64_t styles_stride, void* w16, int N, int O, int I, int ksize,
                                        int demodulate, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK((ksize == 3 && I % 16 == 0) || ksize == 1, "modulate_weights_f16: 3x3 (I %% 16 == 0) or 1x1 weights");
    N3D_CHECK(N >= 0 && O > 0 && I > 0 && I * ksize * ksize <= 256 * 18, "modulate_weights_f16: at most 4608 weights per output channel");
    N3D_CHECK(O <= 65535 * 32 && N <= 65535, "modulate_weights_f16: grid too large");
    if (N == 0) return 0;
    N3D_CHECK(w && styles && w16, "modulate_weights_f16: null tensor");
    ModwParams p;
    p.w = w; p.styles = styles; p.out = (_Float16*)w16; p.N = N; p.O = O; p.I = I; p.K2 = ksize * ksize;
    p.sstride = styles_stride ? styles_stride : I; p.demod = demodulate;
    p.c = (float)(1.0 / sqrt((double)I * ksize * ksize));
    N3dProfScope prof(N3D_K_MISC, stream, 3.0 * N * O * (double)I * p.K2, 4.0 * O * (double)I * p.K2 + 2.0 * N * O * (double)I * p.K2);
    const int L = I * p.K2;
    const dim3 grid(O, N);
    if (L <= 256 * 2) hipLaunchKernelGGL(modulate_weights_f16_kernel<2>, grid, dim3(256), 0, stream, p);
    else if (L <= 256 * 5) hipLaunchKernelGGL(modulate_weights_f16_kernel<5>, grid, dim3(256), 0, stream, p);
    else if (L <= 256 * 9) hipLaunchKernelGGL(modulate_weights_f16_kernel<9>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(modulate_weights_f16_kernel<18>, grid, dim3(256), 0, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}

// All layers of a network in ONE launch (the six layers of the super-resolution module: 6 x ~11 us of launch-bound kernels
// otherwise): the jobs travel by value in the kernel arguments, a workgroup finds its (job, output channel) from the running sum
// of the jobs' channel counts.
constexpr int MODW_MAX_JOBS = 8;
struct ModwMultiParams {
    ModwParams job[MODW_MAX_JOBS];
    int row_end[MODW_MAX_JOBS];
    int njobs;
};
template <int EPT>
__global__ __launch_bounds__(256) void modulate_weights_f16_multi_kernel(ModwMultiParams mp) {
    int j = 0;
    while (j + 1 < mp.njobs && (int)blockIdx.x >= mp.row_end[j]) ++j;
    const int o = blockIdx.x - (j ? mp.row_end[j - 1] : 0);
    modulate_row<EPT>(mp.job[j], o, blockIdx.y);
}

extern "C" int n3d_modulate_weights_f16_multi(const n3d_modw_job* jobs, int njobs, const float* styles_base, int64_t styles_stride, int N,
                                              n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(jobs && njobs >= 1 && njobs <= MODW_MAX_JOBS, "modulate_weights_f16_multi: 1..8 jobs");
    N3D_CHECK(N >= 0 && N <= 65535 && styles_stride > 0, "modulate_weights_f16_multi: bad batch / styles stride");
    if (N == 0) return 0;
    N3D_CHECK(styles_base, "modulate_weights_f16_multi: null styles");
    ModwMultiParams mp;
    mp.njobs = njobs;
    int rows = 0, maxL = 0;
    double flops = 0, bytes = 0;
    for (int j = 0; j < njobs; ++j) {
        const n3d_modw_job& J = jobs[j];
        N3D_CHECK((J.ksize == 3 && J.I % 16 == 0) || J.ksize == 1, "modulate_weights_f16_multi: 3x3 (I %% 16 == 0) or 1x1 weights");
        N3D_CHECK(J.O > 0 && J.I > 0 && J.I * J.ksize * J.ksize <= 256 * 18 && J.w && J.w16, "modulate_weights_f16_multi: bad job");
        ModwParams& p = mp.job[j];
        p.w = J.w; p.styles = styles_base + J.styles_offset; p.out = (_Float16*)J.w16; p.N = N; p.O = J.O; p.I = J.I; p.K2 = J.ksize * J.ksize;
        p.sstride = styles_stride; p.demod = J.demodulate; p.c = (float)(1.0 / sqrt((double)J.I * p.K2));
        rows += J.O; mp.row_end[j] = rows;
        maxL = J.I * p.K2 > maxL ? J.I * p.K2 : maxL;
        flops += 3.0 * N * J.O * (double)J.I * p.K2; bytes += (4.0 + 2.0 * N) * J.O * (double)J.I * p.K2;
    }
    N3dProfScope prof(N3D_K_MISC, stream, flops, bytes);
    const dim3 grid(rows, N);
    if (maxL <= 256 * 9) hipLaunchKernelGGL(modulate_weights_f16_multi_kernel<9>, grid, dim3(256), 0, stream, mp);
    else hipLaunchKernelGGL(modulate_weights_f16_multi_kernel<18>, grid, dim3(256), 0, stream, mp);
    N3D_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// float32 NCHW -> h8 (the block entry `x.to(torch.float16)`, training/networks_stylegan2.py:437) and back (tests, callers that want
// the feature map).  One work item = one 16-byte unit.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_h8_kernel(const T* __restrict__ x, f16x8* __restrict__ y, int C8, int64_t HW, int64_t xbs) {
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c8 = blockIdx.y, n = blockIdx.z;
    if (pix >= HW) return;
    const T* xp = x + (int64_t)n * xbs + (int64_t)c8 * 8 * HW + pix;
    f16x8 v;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (_Float16)xp[k * HW];
    y[((int64_t)n * C8 + c8) * HW + pix] = v;
}
template <typename T>
__global__ __launch_bounds__(256) void h8_to_nchw_kernel(const f16x8* __restrict__ x, T* __restrict__ y, int C8, int64_t HW) {
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c8 = blockIdx.y, n = blockIdx.z;
    if (pix >= HW) return;
    const f16x8 v = x[((int64_t)n * C8 + c8) * HW + pix];
    T* yp = y + ((int64_t)n * C8 + c8) * 8 * HW + pix;
#pragma unroll
    for (int k = 0; k < 8; ++k) yp[k * HW] = (T)v[k];
}

extern "C" int n3d_cast_h8_ex(const void* x, void* y, int N, int C, int64_t HW, int64_t x_batch_stride, int to_h8, int nchw_dtype, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && C > 0 && C % 8 == 0 && HW > 0, "cast_h8: C %% 8 == 0");
    N3D_CHECK(nchw_dtype == N3D_F32 || nchw_dtype == N3D_F16, "cast_h8: the NCHW side is float32 or float16");
    if (N == 0) return 0;
    N3D_CHECK(x && y && C / 8 <= 65535 && N <= 65535, "cast_h8: null tensor or grid too large");
    const bool f16 = nchw_dtype == N3D_F16;
    N3dProfScope prof(N3D_K_MISC, stream, 0.0, (f16 ? 4.0 : 6.0) * N * C * (double)HW);
    const dim3 grid((unsigned)cdiv64(HW, 256), C / 8, N);
    if (to_h8) {
        N3D_CHECK(((uintptr_t)y & 15) == 0, "cast_h8: misaligned h8 tensor");
        const int64_t xbs = x_batch_stride ? x_batch_stride : (int64_t)C * HW;
        if (f16) hipLaunchKernelGGL(nchw_to_h8_kernel<_Float16>, grid, dim3(256), 0, stream, (const _Float16*)x, (f16x8*)y, C / 8, HW, xbs);
        else hipLaunchKernelGGL(nchw_to_h8_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, (f16x8*)y, C / 8, HW, xbs);
    } else {
        N3D_CHECK(((uintptr_t)x & 15) == 0 && x_batch_stride == 0, "cast_h8: misaligned / strided h8 tensor");
        if (f16) hipLaunchKernelGGL(h8_to_nchw_kernel<_Float16>, grid, dim3(256), 0, stream, (const f16x8*)x, (_Float16*)y, C / 8, HW);
        else hipLaunchKernelGGL(h8_to_nchw_kernel<float>, grid, dim3(256), 0, stream, (const f16x8*)x, (float*)y, C / 8, HW);
    }
    N3D_LAUNCH_CHECK();
    return 0;
}
extern "C" int n3d_cast_h8(const void* x, void* y, int N, int C, int64_t HW, int64_t x_batch_stride, int to_h8, n3d_stream_t stream_) {
    return n3d_cast_h8_ex(x, y, N, C, HW, x_batch_stride, to_h8, N3D_F32, stream_);
}

// ------------------------------------------------------------------------------------------------------------------------------
// The 4x4 FIR behind the transposed convolution of a float16 block + the layer's bias_act (conv2d_resample.py:128-129
// `upfirdn2d(x, f, padding = [1,1,1,1], gain = up^2)`, then SynthesisLayer's noise / bias_act, training/networks_stylegan2.py:
// 320-329), h8 in -> h8 out [N][C/8][H-1][W-1][8].  upfirdn2d.cu accumulates the taps in float32 and stores float16; bias_act.cu
// reads that float16 value, computes in float32 and stores float16: two roundings, reproduced here.
// Workgroup = 64 x 16 outputs x 8 channels (one h8 unit plane), 512 work items, each one column x 2 rows x 8 channels.
struct FirH8Params {
    const f16x8* x; const float* f; const float* f1d; f16x8* y;
    int N, C, H, W, OH, OW, flip, tiles_x;
    float gain;
    const float* bias; const float* noise; const float* noise_strength;
    float alpha, act_gain, clamp;
    int multi_round;              // != 0: bias_act as a chain of float16 tensor ops (_bias_act_ref on half tensors; conv2d_f16.hip f16_layer_epilogue)
};

__global__ __launch_bounds__(512, 2) void fir4_h8_kernel(FirH8Params p) {
    constexpr int TW = 64, NT = 512, TH = 16, RPT = 2, FW = TW + 3, FH = TH + 3;
    constexpr int UNITS = FH * FW, LPT = (UNITS + NT - 1) / NT;           // 1273 units -> 3 per work item
    __shared__ f16x8 s_in[UNITS];
    const int tile = blockIdx.x;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int c8 = blockIdx.y, n = blockIdx.z;
    const int ox0 = tx * TW, oy0 = ty * TH;
    float f[4][4];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) f[ky][kx] = p.flip ? p.f[ky * 4 + kx] : p.f[(3 - ky) * 4 + (3 - kx)];
    const int64_t plane_in = (int64_t)p.H * p.W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + ((int64_t)n * (p.C / 8) + c8) * plane_in), 0,
                                                                          (int)(plane_in * 16), 0x00020000);
    {
        f16x8 stage[LPT];
        int slot[LPT];
#pragma unroll
        for (int j = 0; j < LPT; ++j) {                                   // all loads in flight before the first LDS write
            const int e = threadIdx.x + NT * j;
            const int r = e / FW, q = e % FW;
            const int iy = oy0 - 1 + r, ix = ox0 - 1 + q;                 // padding 1: footprint origin = output origin - 1
            const bool ok = e < UNITS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const int off = ok ? (iy * p.W + ix) * 16 : (int)0x80000000;
            stage[j] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
            slot[j] = e < UNITS ? e : -1;
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j)
            if (slot[j] >= 0) s_in[slot[j]] = stage[j];
    }
    __syncthreads();
    const int lx = threadIdx.x % TW, ry = threadIdx.x / TW;
    const int ox = ox0 + lx, oyb = oy0 + ry * RPT;
    if (ox >= p.OW || oyb >= p.OH) return;
    float acc[RPT][8];
#pragma unroll
    for (int j = 0; j < RPT; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[j][c] = 0.f;
#pragma unroll
    for (int rr = 0; rr < RPT + 3; ++rr) {
        f16x8 in[4];
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) in[kx] = s_in[(ry * RPT + rr) * FW + lx + kx];
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int ky = rr - j;
            if (ky < 0 || ky > 3) continue;
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) acc[j][c] += (float)in[kx][c] * f[ky][kx];
        }
    }
    const float nstr = p.noise ? p.noise_strength[0] : 0.f;
    float b16[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) b16[c] = p.bias ? (float)(_Float16)p.bias[c8 * 8 + c] : 0.f;
    f16x8* yp = p.y + ((int64_t)n * (p.C / 8) + c8) * (int64_t)p.OH * p.OW;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int oy = oyb + j;
        if (oy >= p.OH) break;
        const float nz = p.noise ? p.noise[(int64_t)oy * p.OW + ox] * nstr : 0.f;
        f16x8 out;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = (float)(_Float16)(acc[j][c] * p.gain);              // upfirdn2d's float16 output
            if (p.noise) v = (float)(_Float16)(v + nz);                    // x.add_(noise)
            float t = v + b16[c];
            if (p.multi_round) {
                t = (float)(_Float16)t;
                t = (float)(_Float16)fmaxf(t, t * p.alpha);
                if (p.act_gain != 1.f) t = (float)(_Float16)(t * p.act_gain);
            } else {
                t = fmaxf(t, t * p.alpha) * p.act_gain;
            }
            out[c] = (_Float16)__builtin_amdgcn_fmed3f(t, -p.clamp, p.clamp);
        }
        yp[(int64_t)oy * p.OW + ox] = out;
    }
}

// Separable form for filters f = outer(a, a) (the model's [1,3,3,1] filter): a vertical pass per input column, shared through LDS, then
// a horizontal pass — 8 instead of 16 multiply-adds per output and channel, no input staging.  The result is the same float32
// sum evaluated in another order (the f16 inputs times these taps are exact in float32; so are the partial sums unless the terms'
// exponents differ by more than ~9 bits).  Tile: 61 x 16 outputs = 64 input columns, 256 work items = 64 columns x 4 groups of 4 rows.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void fir4_h8_sep_kernel(FirH8Params p) {
    constexpr int SW = 61, COLS = 64, TH = 16, RPT = 4;
    __shared__ f32x4 s_v[2 * TH * COLS];                                  // [channel half][row][column] vertical sums
    const int lx = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
    const int c8 = blockIdx.y, n = blockIdx.z;
    const int ox0 = tx * SW, oy0 = ty * TH;
    float a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = p.flip ? p.f1d[k] : p.f1d[3 - k];
    const int64_t plane_in = (int64_t)p.H * p.W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + ((int64_t)n * (p.C / 8) + c8) * plane_in), 0,
                                                                          (int)(plane_in * 16), 0x00020000);
    {   // vertical pass: input column ox0 - 1 + lx, input rows oy0 - 1 + 4 g + (0..6) -> sums for the output rows oy0 + 4 g + (0..3)
        const int ix = ox0 - 1 + lx;
        f16x8 in[RPT + 3];
#pragma unroll
        for (int r = 0; r < RPT + 3; ++r) {
            const int iy = oy0 - 1 + RPT * g + r;
            const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            in[r] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (iy * p.W + ix) * 16 : (int)0x80000000, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (float)in[j][c] * a[0];
#pragma unroll
            for (int ky = 1; ky < 4; ++ky)
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] += (float)in[j + ky][c] * a[ky];
            s_v[(RPT * g + j) * COLS + lx] = f32x4{v[0], v[1], v[2], v[3]};
            s_v[(TH + RPT * g + j) * COLS + lx] = f32x4{v[4], v[5], v[6], v[7]};
        }
    }
    __syncthreads();
    const int ox = ox0 + lx;
    if (lx >= SW || ox >= p.OW) return;
    const float nstr = p.noise ? p.noise_strength[0] : 0.f;
    float b16[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) b16[c] = p.bias ? (float)(_Float16)p.bias[c8 * 8 + c] : 0.f;
    f16x8* yp = p.y + ((int64_t)n * (p.C / 8) + c8) * (int64_t)p.OH * p.OW;
    const float ga = p.gain;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int oy = oy0 + RPT * g + j;
        if (oy >= p.OH) break;
        const f32x4* r0 = s_v + (RPT * g + j) * COLS + lx, *r1 = r0 + TH * COLS;
        f32x4 s0 = r0[0] * a[0], s1 = r1[0] * a[0];
#pragma unroll
        for (int kx = 1; kx < 4; ++kx) { s0 += r0[kx] * a[kx]; s1 += r1[kx] * a[kx]; }
        const float nz = p.noise ? p.noise[(int64_t)oy * p.OW + ox] * nstr : 0.f;
        f16x8 out;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = (float)(_Float16)((c < 4 ? s0[c] : s1[c - 4]) * ga);     // upfirdn2d's float16 output
            if (p.noise) v = (float)(_Float16)(v + nz);                        // x.add_(noise)
            float t = v + b16[c];
            if (p.multi_round) {
                t = (float)(_Float16)t;
                t = (float)(_Float16)fmaxf(t, t * p.alpha);
                if (p.act_gain != 1.f) t = (float)(_Float16)(t * p.act_gain);
            } else {
                t = fmaxf(t, t * p.alpha) * p.act_gain;
            }
            out[c] = (_Float16)__builtin_amdgcn_fmed3f(t, -p.clamp, p.clamp);
        }
        yp[(int64_t)oy * p.OW + ox] = out;
    }
}

extern "C" int n3d_fir4_h8(const void* x, const float* f, const float* f1d, void* y, int N, int C, int H, int W, int flip, float gain,
                           const n3d_epilogue* epi, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && C > 0 && C % 8 == 0 && H > 3 && W > 3, "fir4_h8: bad shape (C %% 8 == 0)");
    N3D_CHECK(!epi || (!epi->residual && !epi->residual_up_filter && !epi->row_scale && epi->const_scale == 1.f), "fir4_h8: no residual / row scale");
    N3D_CHECK(!epi || epi->act == N3D_ACT_LINEAR || (epi->act == N3D_ACT_LRELU && epi->alpha >= 0.f && epi->alpha <= 1.f), "fir4_h8: linear or leaky-ReLU epilogue only");
    N3D_CHECK(!epi || !epi->noise || epi->noise_strength, "fir4_h8: noise without noise_strength");
    if (N == 0) return 0;
    N3D_CHECK(x && f && y && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "fir4_h8: null or misaligned tensor");
    N3D_CHECK(C / 8 <= 65535 && N <= 65535 && (int64_t)H * W * 16 < (1ll << 31), "fir4_h8: tensor too large");
    FirH8Params p;
    p.x = (const f16x8*)x; p.f = f; p.f1d = f1d; p.y = (f16x8*)y; p.N = N; p.C = C; p.H = H; p.W = W; p.OH = H - 1; p.OW = W - 1; p.flip = flip; p.gain = gain;
    p.tiles_x = cdiv(p.OW, 64);
    p.bias = epi ? epi->bias : nullptr; p.noise = epi ? epi->noise : nullptr; p.noise_strength = epi ? epi->noise_strength : nullptr;
    p.alpha = (epi && epi->act == N3D_ACT_LRELU) ? epi->alpha : 1.f; p.act_gain = epi ? epi->gain : 1.f;
    p.clamp = (epi && epi->clamp >= 0.f) ? epi->clamp : INFINITY;
    p.multi_round = epi && epi->round_f16 == 2;
    N3dProfScope prof(N3D_K_UPFIRDN2D, stream, 2.0 * N * C * (double)p.OH * p.OW * 16, 2.0 * N * C * ((double)H * W + (double)p.OH * p.OW));
    if (f1d) {                                                            // caller's promise: f == outer(f1d, f1d)
        p.tiles_x = cdiv(p.OW, 61);
        hipLaunchKernelGGL(fir4_h8_sep_kernel, dim3(p.tiles_x * cdiv(p.OH, 16), C / 8, N), dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL(fir4_h8_kernel, dim3(p.tiles_x * cdiv(p.OH, 16), C / 8, N), dim3(512), 0, stream, p);
    }
    N3D_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// ToRGBLayer of a float16 block + the skip-image update (training/networks_stylegan2.py:353-357, :446-451):
//   y16 = conv1x1(x16, w16[n])  (float16 result)  ->  bias_act(y16, bias.half(), clamp)  (float16)  ->  img = upsample2d(img) + y.float()
// x h8 [N][C/8][H][W][8], w16 [N][O][C] from n3d_modulate_weights_f16 (demodulate = 0, styles already times weight_gain), O <= 4;
// img_lo [N][O][H/2][W/2] float32 or NULL (first block), img [N][O][H][W] float32.  HBM-bound: one read of x.
struct ToRgbH8Params {
    const f16x8* x; const _Float16* w16; const float* bias; const float* img_lo; const float* upf; float* img;
    int N, C, O, H, W;
    float clamp;
};

__global__ __launch_bounds__(256) void torgb_h8_kernel(ToRgbH8Params p) {
    __shared__ f16x8 s_w[4 * 64];                                         // [o][C/8] units, C <= 512
    const int n = blockIdx.y, C8 = p.C / 8;
    for (int e = threadIdx.x; e < p.O * C8; e += 256) s_w[e] = reinterpret_cast<const f16x8*>(p.w16 + (int64_t)n * p.O * p.C)[e];
    __syncthreads();
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const f16x8* xp = p.x + (int64_t)n * C8 * HW + pix;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int u = 0; u < C8; ++u) {
        const f16x8 v = xp[(int64_t)u * HW];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o >= p.O) break;
            const f16x8 wv = s_w[o * C8 + u];
#pragma unroll
            for (int k = 0; k < 8; k += 2)
                acc[o] = __builtin_amdgcn_fdot2(f16x2{v[k], v[k + 1]}, f16x2{wv[k], wv[k + 1]}, acc[o], false);
        }
    }
    const int oy = (int)(pix / p.W), ox = (int)(pix % p.W);
    n3d_up2_taps taps;
    if (p.img_lo) taps = n3d_up2_setup(p.upf, oy, ox, p.H >> 1, p.W >> 1);
    for (int o = 0; o < p.O; ++o) {
        float t = (float)(_Float16)acc[o] + (p.bias ? (float)(_Float16)p.bias[o] : 0.f);
        float v = (float)(_Float16)fminf(fmaxf(t, -p.clamp), p.clamp);
        if (p.img_lo) v += n3d_up2_apply(taps, p.img_lo + ((int64_t)n * p.O + o) * (HW >> 2));
        p.img[((int64_t)n * p.O + o) * HW + pix] = v;
    }
}

// The same layer with MORE than 4 output channels — the toRGB layers of fp16 blocks inside the StyleGAN2 backbones (32 neural-texture
// channels, 96 tri-plane channels: num_fp16_res > 0 / legacy.load_network_pkl(force_fp16=True)).  A workgroup takes 16 output channels
// (blockIdx.z) of 256 pixels; x is re-read O/16 times (from L2 / Infinity Cache: 2-6 passes over a tensor the 3x3 layer before just wrote).
__global__ __launch_bounds__(256) void torgb_h8_wide_kernel(ToRgbH8Params p) {
    __shared__ f16x8 s_w[16 * 64];                                        // [o][C/8] units, C <= 512
    const int n = blockIdx.y, C8 = p.C / 8, o0 = blockIdx.z * 16, no = min(16, p.O - o0);
    for (int e = threadIdx.x; e < no * C8; e += 256) s_w[e] = reinterpret_cast<const f16x8*>(p.w16 + ((int64_t)n * p.O + o0) * p.C)[e];
    __syncthreads();
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const f16x8* xp = p.x + (int64_t)n * C8 * HW + pix;
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = 0.f;
    for (int u = 0; u < C8; ++u) {
        const f16x8 v = xp[(int64_t)u * HW];
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            const f16x8 wv = s_w[(o < no ? o : 0) * C8 + u];
#pragma unroll
            for (int k = 0; k < 8; k += 2)
                acc[o] = __builtin_amdgcn_fdot2(f16x2{v[k], v[k + 1]}, f16x2{wv[k], wv[k + 1]}, acc[o], false);
        }
    }
    const int oy = (int)(pix / p.W), ox = (int)(pix % p.W);
    n3d_up2_taps taps;
    if (p.img_lo) taps = n3d_up2_setup(p.upf, oy, ox, p.H >> 1, p.W >> 1);
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        if (o < no) {
            const float t = (float)(_Float16)acc[o] + (p.bias ? (float)(_Float16)p.bias[o0 + o] : 0.f);
            float v = (float)(_Float16)fminf(fmaxf(t, -p.clamp), p.clamp);
            if (p.img_lo) v += n3d_up2_apply(taps, p.img_lo + ((int64_t)n * p.O + o0 + o) * (HW >> 2));
            p.img[((int64_t)n * p.O + o0 + o) * HW + pix] = v;
        }
    }
}

extern "C" int n3d_torgb_h8(const void* x, const void* w16, const float* bias, const float* img_lo, const float* up_filter, float* img, int N, int C,
                            int O, int H, int W, float clamp, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && C >= 8 && C % 8 == 0 && C <= 512 && O >= 1 && O <= 1024 && H > 0 && W > 0, "torgb_h8: C %% 8 == 0, C <= 512, O <= 1024");
    N3D_CHECK(!img_lo || (up_filter && H % 2 == 0 && W % 2 == 0), "torgb_h8: the low-resolution image needs the 4x4 filter and even H, W");
    if (N == 0) return 0;
    N3D_CHECK(x && w16 && img && (((uintptr_t)x | (uintptr_t)w16) & 15) == 0 && N <= 65535, "torgb_h8: null or misaligned tensor");
    ToRgbH8Params p;
    p.x = (const f16x8*)x; p.w16 = (const _Float16*)w16; p.bias = bias; p.img_lo = img_lo; p.upf = up_filter; p.img = img;
    p.N = N; p.C = C; p.O = O; p.H = H; p.W = W; p.clamp = clamp >= 0.f ? clamp : INFINITY;
    const double HW = (double)H * W;
    N3dProfScope prof(N3D_K_CONV1X1_BF16X3, stream, 2.0 * N * O * C * HW, N * HW * (2.0 * C + 4.0 * O + (img_lo ? 1.0 * O : 0.0)));
    if (O <= 4) hipLaunchKernelGGL(torgb_h8_kernel, dim3((unsigned)cdiv64((int64_t)H * W, 256), N), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(torgb_h8_wide_kernel, dim3((unsigned)cdiv64((int64_t)H * W, 256), N, cdiv(O, 16)), dim3(256), 0, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}
