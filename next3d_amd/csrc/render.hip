// Tri-plane volume renderer for gfx950: ONE wavefront renders ONE ray end to end and only the composited
// [N,32,R,R] feature image and [N,1,R,R] depth ever reach HBM (the reference materialises ~20 tensors of
// [N, R^2 * S, ...], volumetric_rendering/renderer.py:95-147).
//
//   lane s  : stratified depth t_s -> point -> 3 bilinear tri-plane gathers (channels-last planes: one texel's 32
//             channels are 128 contiguous bytes = 8 x 16-byte loads) -> mean -> 32->64 softplus ->64->33 MLP
//             (weights are wave-uniform: scalar loads, no LDS) -> (rgb[32], sigma) parked in LDS
//   wave    : mid-point ray march (exclusive cumprod), weight smoothing, inverse-CDF importance depths,
//             second decode pass, rank-merge of the two sample sets, final march + composite.
// Bound: gather/L2 (604 MB of texel traffic per frame, 25 MB compulsory HBM) + 3.3 GFLOP of fp32 VALU.
//
// Replaces RaySampler.forward (reference volumetric_rendering/ray_sampler.py:24-63), ImportanceRenderer.forward
// (renderer.py:95-147), sample_from_planes (:62-72, grid_sampler_2d bilinear/zeros/align_corners=False),
// OSGDecoder.forward (training_avatar_texture/triplane_next3d.py:359-371), MipRayMarcher2 (ray_marcher.py:27-66),
// sample_importance / sample_pdf / unify_samples (renderer.py:164-268).
#include <math.h>

#include "common.h"

#define RN_MAX_S 256   // max coarse + fine samples per ray
#define RN_C 32        // plane channels
#define RN_HID 64

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct RenderParams {
    const float* planes;      // [N,3,PH,PW,32] channels-last
    const float* cam2world;   // [N,16]
    const float* intrinsics;  // [N,9]
    const float* tlin;        // [Sc] torch.linspace(ray_start, ray_end, Sc)
    const float* jitter;      // [N,R*R,Sc]
    const float* u;           // [N*R*R,Sf]
    const float* w1;          // [64,32] pre-scaled by 1/sqrt(32)
    const float* b1;          // [64]
    const float* w2;          // [64,34] = (W2 / sqrt(64))^T, rows padded with one zero: hidden unit j's 33 outgoing weights are contiguous
    const float* b2;          // [33]
    const float* bounds;      // [2] global min / max of the coarse depths (ray_marcher.py:54)
    float* feat;              // [N,32,R,R]
    float* depth;             // [N,1,R,R]
    float* wsum;              // [N,R*R] or NULL
    int N, R, Sc, Sf, PH, PW;
    float depth_delta, coord_scale;
};

// softplus / sigmoid on the hardware transcendental units (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each): the libm
// log1pf(expf(x)) pair is ~100 VALU instructions and was 40 % of this VALU-bound kernel's instruction stream.  For very
// negative x, log(1 + e^x) loses the e^x tail below 2^-24 — an absolute error < 6e-8 on a quantity of order 1.
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : __logf(1.f + __expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return __frcp_rn(1.f + __expf(-x)); }

// bilinear, zeros padding, align_corners=False: the 4 taps of one plane as (texel pointer, weight); taps outside the plane
// get weight 0 and point at texel (0,0) (always loadable), so the gather below is branch-free
__device__ __forceinline__ void plane_taps(const float* __restrict__ plane, int PH, int PW, float gx, float gy,
                                           const float4* (&tp)[4], float (&tw)[4]) {
    const float ix = ((gx + 1.f) * (float)PW - 1.f) * 0.5f;
    const float iy = ((gy + 1.f) * (float)PH - 1.f) * 0.5f;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    // guard against NaN/inf/huge coordinates before the int conversion
    const bool sane = fx0 > -2.f && fx0 < (float)PW + 1.f && fy0 > -2.f && fy0 < (float)PH + 1.f;
    const int x0 = sane ? (int)fx0 : -4, y0 = sane ? (int)fy0 : -4;
    const float wx1 = ix - fx0, wy1 = iy - fy0;
    const float wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
    const float wgt[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};   // nw, ne, sw, se
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
        const bool ok = xx >= 0 && xx < PW && yy >= 0 && yy < PH;
        tp[k] = reinterpret_cast<const float4*>(plane + (ok ? ((int64_t)yy * PW + xx) * RN_C : 0));
        tw[k] = ok ? wgt[k] : 0.f;
    }
}

// f += sum_k tw[k] * texel_k[0..31]: all 32 16-byte loads of the plane's four taps are issued before the first use
__device__ __forceinline__ void gather4(const float4* const (&tp)[4], const float (&tw)[4], f32x2 (&f)[RN_C / 2]) {
    float4 v[4][RN_C / 4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < RN_C / 4; ++q) v[k][q] = tp[k][q];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x2 w2 = f32x2{tw[k], tw[k]};
#pragma unroll
        for (int q = 0; q < RN_C / 4; ++q) {
            f[2 * q] = __builtin_elementwise_fma(w2, f32x2{v[k][q].x, v[k][q].y}, f[2 * q]);
            f[2 * q + 1] = __builtin_elementwise_fma(w2, f32x2{v[k][q].z, v[k][q].w}, f[2 * q + 1]);
        }
    }
}

// decode one sample: features at `pt` -> (rgb[32] in out[1..32], sigma in out[0])
__device__ __forceinline__ void decode_point(const RenderParams& p, int n, float px, float py, float pz, float (&out)[RN_C + 1]) {
    const float cx = p.coord_scale * px, cy = p.coord_scale * py, cz = p.coord_scale * pz;
    const float* base = p.planes + (int64_t)n * 3 * p.PH * p.PW * RN_C;
    const int64_t ps = (int64_t)p.PH * p.PW * RN_C;
    f32x2 f2[RN_C / 2];
#pragma unroll
    for (int c = 0; c < RN_C / 2; ++c) f2[c] = f32x2{0.f, 0.f};
    const float4* tp[3][4];
    float tw[3][4];
    plane_taps(base, p.PH, p.PW, cx, cy, tp[0], tw[0]);            // plane 0: (x, y)
    plane_taps(base + ps, p.PH, p.PW, cx, cz, tp[1], tw[1]);       // plane 1: (x, z)
    plane_taps(base + 2 * ps, p.PH, p.PW, cz, cy, tp[2], tw[2]);   // plane 2: (z, y)   (renderer.py:42-44)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) gather4(tp[pl], tw[pl], f2);
#pragma unroll
    for (int c = 0; c < RN_C / 2; ++c) f2[c] = f2[c] * (1.f / 3.f);   // mean over the three planes (triplane_next3d.py:361)
    // 32 -> 64 softplus -> 33 MLP on packed fp32 FMAs (v_pk_fma_f32: two lanes of math per VALU slot; the vector unit's fp32
    // peak assumes them).  Weights are wave-uniform -> scalar loads; the 33 outputs are 17 (even, odd) pairs.
    f32x2 o2[17];
#pragma unroll
    for (int k = 0; k < 16; ++k) o2[k] = f32x2{p.b2[2 * k], p.b2[2 * k + 1]};
    o2[16] = f32x2{p.b2[32], 0.f};
    for (int j = 0; j < RN_HID; ++j) {      // hidden unit j: uniform weight addresses -> scalar loads
        const f32x2* w1r = reinterpret_cast<const f32x2*>(p.w1 + j * RN_C);
        f32x2 h2 = f32x2{p.b1[j], 0.f};
#pragma unroll
        for (int c = 0; c < RN_C / 2; ++c) h2 = __builtin_elementwise_fma(w1r[c], f2[c], h2);
        const float h = softplus_f(h2.x + h2.y);
        const f32x2 hh = f32x2{h, h};
        const f32x2* w2r = reinterpret_cast<const f32x2*>(p.w2 + j * 34);
#pragma unroll
        for (int k = 0; k < 17; ++k) o2[k] = __builtin_elementwise_fma(w2r[k], hh, o2[k]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) { out[2 * k] = o2[k].x; out[2 * k + 1] = o2[k].y; }
    out[32] = o2[16].x;
#pragma unroll
    for (int k = 1; k <= RN_C; ++k) out[k] = sigmoid_f(out[k]) * (1.f + 2.f * 0.001f) - 0.001f;
}

// LDS layout per ray (one wave per block): all arrays sized for M = Sc + Sf samples
struct RayLds {
    float* col;     // [M][33]  (col[s][0..31] rgb, [32] pad)
    float* sig;     // [M]
    float* dep;     // [M]
    float* wgt;     // [M]   march weights
    float* fac;     // [M]   1 - alpha + 1e-10
    float* trn;     // [M]   exclusive cumprod
    float* cdf;     // [M]
    float* bins;    // [M]
    int* order;     // [M]
};

// mid-point ray march over `count` samples addressed through idx(i) (ray_marcher.py:28-46); fills wgt[0..count-2]
template <typename IdxFn>
__device__ __forceinline__ void march_weights(const RayLds& L, int count, int lane, IdxFn idx) {
    for (int i = lane; i < count - 1; i += 64) {
        const int a = idx(i), b = idx(i + 1);
        const float delta = L.dep[b] - L.dep[a];
        const float dm = softplus_f((L.sig[a] + L.sig[b]) / 2.f - 1.f);
        const float alpha = 1.f - expf(-(dm * delta));
        L.wgt[i] = alpha;
        L.fac[i] = 1.f - alpha + 1e-10f;
    }
    __syncthreads();
    if (lane == 0) {       // sequential exclusive cumprod, same order as torch.cumprod
        float T = 1.f;
        for (int i = 0; i < count - 1; ++i) { L.trn[i] = T; T *= L.fac[i]; }
    }
    __syncthreads();
    for (int i = lane; i < count - 1; i += 64) L.wgt[i] = L.wgt[i] * L.trn[i];
    __syncthreads();
}

__global__ __launch_bounds__(64) void render_rays_kernel(RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int ray = blockIdx.x, n = blockIdx.y;
    const int R = p.R, Sc = p.Sc, Sf = p.Sf, M = Sc + Sf;
    RayLds L;
    L.col = smem;
    L.sig = L.col + M * 33;
    L.dep = L.sig + M; L.wgt = L.dep + M; L.fac = L.wgt + M; L.trn = L.fac + M; L.cdf = L.trn + M; L.bins = L.cdf + M;
    L.order = reinterpret_cast<int*>(L.bins + M);

    // ---- ray (ray_sampler.py:33-61); ray index m = i*R + j, x from j, y from i
    const float* c2w = p.cam2world + n * 16;
    const float* K = p.intrinsics + n * 9;
    const float fx = K[0], fy = K[4], cxk = K[2], cyk = K[5], sk = K[1];
    const int ri = ray / R, rj = ray % R;
    const float inv = (float)(1.0 / (double)R), half = (float)(0.5 / (double)R);
    const float x_cam = __fadd_rn(__fmul_rn((float)rj, inv), half);
    const float y_cam = __fadd_rn(__fmul_rn((float)ri, inv), half);
    const float x_lift = (x_cam - cxk + cyk * sk / fy - sk * y_cam / fy) / fx;
    const float y_lift = (y_cam - cyk) / fy;
    const float ox = c2w[3], oy = c2w[7], oz = c2w[11];
    float dx = c2w[0] * x_lift + c2w[1] * y_lift + c2w[2] + c2w[3] - ox;
    float dy = c2w[4] * x_lift + c2w[5] * y_lift + c2w[6] + c2w[7] - oy;
    float dz = c2w[8] * x_lift + c2w[9] * y_lift + c2w[10] + c2w[11] - oz;
    const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
    dx /= nrm; dy /= nrm; dz /= nrm;

    // ---- coarse pass: sample s -> slot s
    const float* jit = p.jitter + ((int64_t)n * R * R + ray) * Sc;
    for (int s = lane; s < Sc; s += 64) {
        const float t = __fadd_rn(p.tlin[s], __fmul_rn(jit[s], p.depth_delta));   // renderer.py:203-205
        float out[RN_C + 1];
        decode_point(p, n, __fadd_rn(ox, __fmul_rn(t, dx)), __fadd_rn(oy, __fmul_rn(t, dy)), __fadd_rn(oz, __fmul_rn(t, dz)), out);
        L.dep[s] = t; L.sig[s] = out[0];
#pragma unroll
        for (int c = 0; c < RN_C; ++c) L.col[s * 33 + c] = out[1 + c];
    }
    __syncthreads();

    int count = Sc;
    if (Sf > 0) {
        march_weights(L, Sc, lane, [](int i) { return i; });
        // ---- importance depths (renderer.py:209-268)
        const int Lw = Sc - 1;       // number of march weights
        const int Np = Sc - 3;       // pdf bins (weights[:, 1:-1])
        for (int i = lane; i < Lw; i += 64) {
            const float wl = i > 0 ? L.wgt[i - 1] : -INFINITY, wc = L.wgt[i], wr = i + 1 < Lw ? L.wgt[i + 1] : -INFINITY;
            const float m0 = fmaxf(wl, wc), m1 = fmaxf(wc, wr);       // max_pool1d(2,1,pad=1)
            L.fac[i] = (m0 + m1) / 2.f + 0.01f;                        // avg_pool1d(2,1) + 0.01
            L.bins[i] = 0.5f * (L.dep[i] + L.dep[i + 1]);
        }
        __syncthreads();
        float part = 0.f;
        for (int k = lane; k < Np; k += 64) part += L.fac[k + 1] + 1e-5f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) {
            float c = 0.f;
            L.cdf[0] = 0.f;
            for (int k = 0; k < Np; ++k) { c += (L.fac[k + 1] + 1e-5f) / part; L.cdf[k + 1] = c; }
        }
        __syncthreads();
        const float* uu = p.u + ((int64_t)n * R * R + ray) * Sf;
        for (int j = lane; j < Sf; j += 64) {
            const float u = uu[j];
            int ind = 0;                                             // searchsorted(cdf, u, right=True)
            for (int k = 0; k <= Np; ++k) ind += (L.cdf[k] <= u) ? 1 : 0;
            const int below = max(ind - 1, 0), above = min(ind, Np);
            const float c0 = L.cdf[below], c1 = L.cdf[above], b0 = L.bins[below], b1 = L.bins[above];
            float denom = c1 - c0;
            if (denom < 1e-5f) denom = 1.f;
            const float t = __fadd_rn(b0, __fmul_rn((u - c0) / denom, (b1 - b0)));
            float out[RN_C + 1];
            decode_point(p, n, __fadd_rn(ox, __fmul_rn(t, dx)), __fadd_rn(oy, __fmul_rn(t, dy)), __fadd_rn(oz, __fmul_rn(t, dz)), out);
            const int s = Sc + j;
            L.dep[s] = t; L.sig[s] = out[0];
#pragma unroll
            for (int c = 0; c < RN_C; ++c) L.col[s * 33 + c] = out[1 + c];
        }
        __syncthreads();
        // ---- unify_samples: stable rank of every sample in the merged order (renderer.py:164-182)
        for (int k = lane; k < M; k += 64) {
            const float d = L.dep[k];
            int rank = 0;
            for (int q = 0; q < M; ++q) {
                const float dq = L.dep[q];
                rank += (dq < d || (dq == d && q < k)) ? 1 : 0;
            }
            L.order[rank] = k;
        }
        __syncthreads();
        count = M;
        const int* ord = L.order;
        march_weights(L, M, lane, [ord](int i) { return ord[i]; });
    } else {
        for (int k = lane; k < Sc; k += 64) L.order[k] = k;
        __syncthreads();
        march_weights(L, Sc, lane, [](int i) { return i; });
    }

    // ---- composite (ray_marcher.py:48-59): lanes 0..31 one channel each, lane 32 depth, lane 33 weight total
    float acc = 0.f;
    if (lane < RN_C) {
        for (int i = 0; i < count - 1; ++i) {
            const int a = L.order[i], b = L.order[i + 1];
            acc += L.wgt[i] * ((L.col[a * 33 + lane] + L.col[b * 33 + lane]) / 2.f);
        }
        p.feat[((int64_t)n * RN_C + lane) * R * R + ray] = acc * 2.f - 1.f;
    } else if (lane == 32) {
        float wt = 0.f;
        for (int i = 0; i < count - 1; ++i) {
            const int a = L.order[i], b = L.order[i + 1];
            acc += L.wgt[i] * ((L.dep[a] + L.dep[b]) / 2.f);
            wt += L.wgt[i];
        }
        float d = acc / wt;
        if (isnan(d)) d = INFINITY;                       // nan_to_num(nan=inf)
        d = fminf(fmaxf(d, p.bounds[0]), p.bounds[1]);
        p.depth[(int64_t)n * R * R + ray] = d;
        if (p.wsum) p.wsum[(int64_t)n * R * R + ray] = wt;
    }
}

// global min / max of the coarse depths over the whole batch (ray_marcher.py:54 clamps against them).  Multi-block: every
// block reduces a slice and merges with integer atomics on the float bit patterns (the depths are positive, so the bit
// patterns order like the values); bounds_ws is initialised by a one-thread launch in front.
__global__ void render_depth_bounds_init_kernel(float* __restrict__ bounds) {
    bounds[0] = INFINITY; bounds[1] = 0.f;
}
__global__ __launch_bounds__(256) void render_depth_bounds_kernel(const float* __restrict__ tlin, const float* __restrict__ jitter,
                                                                  int64_t rays, int Sc, float delta, float* __restrict__ bounds) {
    __shared__ float smin[4], smax[4];
    float lo = INFINITY, hi = 0.f;
    const float t0 = tlin[0], t1 = tlin[Sc - 1];
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rays; r += (int64_t)gridDim.x * blockDim.x) {
        lo = fminf(lo, __fadd_rn(t0, __fmul_rn(jitter[r * Sc], delta)));
        hi = fmaxf(hi, __fadd_rn(t1, __fmul_rn(jitter[r * Sc + Sc - 1], delta)));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off, 64)); hi = fmaxf(hi, __shfl_xor(hi, off, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = fminf(lo, smin[w]); hi = fmaxf(hi, smax[w]); }
        atomicMin(reinterpret_cast<unsigned int*>(bounds), __float_as_uint(lo));
        atomicMax(reinterpret_cast<unsigned int*>(bounds) + 1, __float_as_uint(hi));
    }
}

// planes_cl[n][p][y][x][c] = dyn_p[n][c][y][x] * a + static[n][p*32+c][y][x] * (1 - a),  a = alpha[n][p][y][x]
// (training_avatar_texture/triplane_next3d.py:171-174), written channels-last for the renderer's gathers.
__global__ __launch_bounds__(256) void blend_planes_kernel(const float* __restrict__ front, const float* __restrict__ side,
                                                           const float* __restrict__ top, const float* __restrict__ stat,
                                                           const float* __restrict__ alpha, float* __restrict__ out, int N,
                                                           int H, int W) {
    __shared__ float tile[RN_C][65];
    const int x0 = blockIdx.x * 64, y = blockIdx.y, np = blockIdx.z, n = np / 3, pl = np % 3;
    const float* dyn = (pl == 0 ? front : (pl == 1 ? side : top)) + (int64_t)n * RN_C * H * W;
    const float* st = stat + ((int64_t)n * 3 + pl) * RN_C * H * W;
    const float* al = alpha + ((int64_t)n * 3 + pl) * H * W + (int64_t)y * W;
    for (int e = threadIdx.x; e < RN_C * 64; e += 256) {
        const int c = e / 64, xx = e % 64, x = x0 + xx;
        float v = 0.f;
        if (x < W) {
            const float a = al[x];
            const int64_t o = ((int64_t)c * H + y) * W + x;
            v = dyn[o] * a + st[o] * (1.f - a);
        }
        tile[c][xx] = v;
    }
    __syncthreads();
    float* dst = out + ((((int64_t)n * 3 + pl) * H + y) * W + x0) * RN_C;
    for (int e = threadIdx.x; e < RN_C * 64; e += 256) {
        const int xx = e / RN_C, c = e % RN_C;
        if (x0 + xx < W) dst[(int64_t)xx * RN_C + c] = tile[c][xx];
    }
}

// NCHW planes -> channels-last (for callers that hold blended planes in the reference layout)
__global__ __launch_bounds__(256) void planes_to_cl_kernel(const float* __restrict__ src, float* __restrict__ out, int H, int W) {
    __shared__ float tile[RN_C][65];
    const int x0 = blockIdx.x * 64, y = blockIdx.y, np = blockIdx.z;
    const float* s = src + (int64_t)np * RN_C * H * W;
    for (int e = threadIdx.x; e < RN_C * 64; e += 256) {
        const int c = e / 64, xx = e % 64, x = x0 + xx;
        tile[c][xx] = x < W ? s[((int64_t)c * H + y) * W + x] : 0.f;
    }
    __syncthreads();
    float* dst = out + (((int64_t)np * H + y) * W + x0) * RN_C;
    for (int e = threadIdx.x; e < RN_C * 64; e += 256) {
        const int xx = e / RN_C, c = e % RN_C;
        if (x0 + xx < W) dst[(int64_t)xx * RN_C + c] = tile[c][xx];
    }
}

extern "C" int n3d_blend_planes(const float* front, const float* side, const float* top, const float* stat, const float* alpha,
                                float* planes_cl, int N, int H, int W, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && H > 0 && W > 0 && H <= 65535, "blend_planes: bad shape");
    if (N == 0) return 0;
    N3D_CHECK(front && side && top && stat && alpha && planes_cl, "blend_planes: null tensor");
    N3dProfScope prof(N3D_K_RENDER, stream, 3.0 * N * 3 * RN_C * (double)H * W, 4.0 * N * 3 * (double)H * W * (3 * RN_C + 1));
    hipLaunchKernelGGL(blend_planes_kernel, dim3(cdiv(W, 64), H, N * 3), dim3(256), 0, stream, front, side, top, stat, alpha,
                       planes_cl, N, H, W);
    N3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int n3d_planes_to_channels_last(const float* planes, float* planes_cl, int N, int H, int W, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && H > 0 && W > 0 && H <= 65535, "planes_to_channels_last: bad shape");
    if (N == 0) return 0;
    N3D_CHECK(planes && planes_cl, "planes_to_channels_last: null tensor");
    N3dProfScope prof(N3D_K_RENDER, stream, 0.0, 8.0 * N * 3 * RN_C * (double)H * W);
    hipLaunchKernelGGL(planes_to_cl_kernel, dim3(cdiv(W, 64), H, N * 3), dim3(256), 0, stream, planes, planes_cl, H, W);
    N3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int n3d_render_rays(const float* planes_cl, const float* cam2world, const float* intrinsics, const float* tlin,
                               const float* jitter, const float* u, const float* w1, const float* b1, const float* w2,
                               const float* b2, float* feat, float* depth, float* wsum, float* bounds_ws, int N, int R, int Sc,
                               int Sf, int PH, int PW, float depth_delta, float coord_scale, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && R > 0 && R * R <= 1 << 24, "render_rays: bad resolution");
    N3D_CHECK(Sc >= 4 && Sf >= 0 && Sc + Sf <= RN_MAX_S, "render_rays: need 4 <= Sc and Sc + Sf <= %d", RN_MAX_S);
    N3D_CHECK(N <= 65535, "render_rays: batch too large");
    if (N == 0) return 0;
    N3D_CHECK(planes_cl && cam2world && intrinsics && tlin && jitter && (u || Sf == 0) && w1 && b1 && w2 && b2 && feat && depth && bounds_ws,
              "render_rays: null tensor");
    N3D_CHECK(((uintptr_t)planes_cl & 15) == 0, "render_rays: planes must be 16-byte aligned");
    RenderParams p;
    p.planes = planes_cl; p.cam2world = cam2world; p.intrinsics = intrinsics; p.tlin = tlin; p.jitter = jitter; p.u = u;
    p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.bounds = bounds_ws; p.feat = feat; p.depth = depth; p.wsum = wsum;
    p.N = N; p.R = R; p.Sc = Sc; p.Sf = Sf; p.PH = PH; p.PW = PW; p.depth_delta = depth_delta; p.coord_scale = coord_scale;
    const int M = Sc + Sf;
    const size_t lds = (size_t)M * (33 + 7) * sizeof(float) + (size_t)M * sizeof(int);
    const double pts = (double)N * R * R * M;
    N3dProfScope prof(N3D_K_RENDER, stream, pts * 2.0 * (RN_C * RN_HID + RN_HID * (RN_C + 1)),
                      pts * 12.0 * RN_C * 4.0 + 4.0 * N * R * R * (RN_C + 1));
    hipLaunchKernelGGL(render_depth_bounds_init_kernel, dim3(1), dim3(1), 0, stream, bounds_ws);
    N3D_LAUNCH_CHECK();
    const int64_t nrays = (int64_t)N * R * R;
    hipLaunchKernelGGL(render_depth_bounds_kernel, dim3((unsigned)(cdiv64(nrays, 256) > 256 ? 256 : cdiv64(nrays, 256))), dim3(256), 0, stream,
                       tlin, jitter, nrays, Sc, depth_delta, bounds_ws);
    N3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(render_rays_kernel, dim3(R * R, N), dim3(64), lds, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}

// ---- point queries: tri-plane features + decoder at arbitrary 3-D points (shape extraction).  One lane per point, the
// same gather + MLP code as the renderer.  Replaces ImportanceRenderer.run_model (reference vr/renderer.py:149-155) as
// called by TriPlaneGenerator.sample / sample_mixed (tat/triplane_next3d.py:232-322).
__global__ __launch_bounds__(256) void sample_points_kernel(RenderParams p, const float* __restrict__ coords, float* __restrict__ rgb,
                                                            float* __restrict__ sigma, int64_t M) {
    const int n = blockIdx.y;
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float* c = coords + ((int64_t)n * M + m) * 3;
    float out[RN_C + 1];
    decode_point(p, n, c[0], c[1], c[2], out);
    sigma[(int64_t)n * M + m] = out[0];
    float4* dst = reinterpret_cast<float4*>(rgb + ((int64_t)n * M + m) * RN_C);
#pragma unroll
    for (int q = 0; q < RN_C / 4; ++q) dst[q] = make_float4(out[1 + 4 * q], out[2 + 4 * q], out[3 + 4 * q], out[4 + 4 * q]);
}

extern "C" int n3d_sample_points(const float* planes_cl, const float* coords, const float* w1, const float* b1, const float* w2t,
                                 const float* b2, float* rgb, float* sigma, int N, int64_t M, int PH, int PW, float coord_scale,
                                 n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && M >= 0 && N <= 65535 && cdiv64(M, 256) < (1ll << 31), "sample_points: bad sizes");
    if (N == 0 || M == 0) return 0;
    N3D_CHECK(planes_cl && coords && w1 && b1 && w2t && b2 && rgb && sigma, "sample_points: null tensor");
    N3D_CHECK((((uintptr_t)planes_cl | (uintptr_t)rgb) & 15) == 0, "sample_points: planes / rgb must be 16-byte aligned");
    RenderParams p = {};
    p.planes = planes_cl; p.w1 = w1; p.b1 = b1; p.w2 = w2t; p.b2 = b2; p.N = N; p.PH = PH; p.PW = PW; p.coord_scale = coord_scale;
    const double pts = (double)N * (double)M;
    N3dProfScope prof(N3D_K_RENDER, stream, pts * 2.0 * (RN_C * RN_HID + RN_HID * (RN_C + 1)), pts * (12.0 * RN_C * 4.0 + 4.0 * (RN_C + 4)));
    hipLaunchKernelGGL(sample_points_kernel, dim3((unsigned)cdiv64(M, 256), N), dim3(256), 0, stream, p, coords, rgb, sigma, M);
    N3D_LAUNCH_CHECK();
    return 0;
}
