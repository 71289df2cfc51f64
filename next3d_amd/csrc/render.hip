// Tri-plane volume renderer for gfx950: one wavefront renders TWO rays end to end and only the composited [N,32,R,R] feature
// image and [N,1,R,R] depth ever reach HBM (the reference materialises ~20 tensors of [N, R^2 * S, ...],
// volumetric_rendering/renderer.py:95-147).
//
//   decode  : passes of 32 samples over the two rays' sample lists; the lane pair (s, s + 32) owns a sample: stratified /
//             importance depth -> point -> 3 bilinear tri-plane gathers (channels-last planes: a texel's 32 channels are 128
//             contiguous bytes; each lane half takes 64 of them = 4 x 16-byte loads per tap) -> mean -> 32 -> 64 softplus -> 33 MLP
//             on the fp32 matrix pipe (see render_rays_kernel) -> (rgb[32], sigma) parked in LDS
//   per ray : one ray per lane half: mid-point ray march (exclusive cumprod), weight smoothing, inverse-CDF importance depths,
//             rank-merge of the two sample sets, final march + composite.
// Bound: gather/L1 (604 MB of texel traffic per frame through the texture path, 25 MB compulsory HBM) + 3.3 GFLOP of fp32 MFMA.
// (sample_points_kernel at the end keeps the per-lane VALU decoder: decode_point.)
//
// Replaces RaySampler.forward (reference volumetric_rendering/ray_sampler.py:24-63), ImportanceRenderer.forward
// (renderer.py:95-147), sample_from_planes (:62-72, grid_sampler_2d bilinear/zeros/align_corners=False),
// OSGDecoder.forward (training_avatar_texture/triplane_next3d.py:359-371), MipRayMarcher2 (ray_marcher.py:27-66),
// sample_importance / sample_pdf / unify_samples (renderer.py:164-268).
#include <math.h>
#include <stdlib.h>

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
    // rendering options outside the ffhq configuration (n3d_render_opts)
    int depth_mode;           // 0: tlin[i] + jitter * depth_delta (fixed ray_start / ray_end, renderer.py:197-201);
                              // 1: disparity-space sampling (:186-193: tlin = linspace(0, 1), depth = 1 / (1/start * (1 - d) + 1/end * d));
                              // 2: per-ray (start, end) from ray_bounds (ray_start = ray_end = 'auto', :99-106 + :194-196)
    float inv_start, inv_end; // depth_mode 1
    const float* ray_bounds;  // depth_mode 2: [N*R*R][2]
    int white_back;           // ray_marcher.py:56-57
    const float* noise_c;     // density noise draws [N,R*R,Sc] / [N,R*R,Sf] (renderer.py:152-153: sigma += randn_like * density_noise) or NULL
    const float* noise_f;
    float noise_scale;
    float* fine_out;          // n3d_render_opts.fine_depths_out / fine_depths_in (verification seam) or NULL
    const float* fine_in;
};

// total order on floats as unsigned keys (negative depths are possible with 'auto' bounds when the camera sits inside the box)
__device__ __forceinline__ unsigned int f2key(float f) { const unsigned int b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float key2f(unsigned int k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }

// coarse (stratified) depth of sample i of one ray, operation for operation as sample_stratified (renderer.py:184-207)
__device__ __forceinline__ float coarse_depth(const RenderParams& p, int64_t gray /* n * R*R + ray */, int i, float jit) {
    if (p.depth_mode == 0) return __fadd_rn(p.tlin[i], __fmul_rn(jit, p.depth_delta));
    if (p.depth_mode == 1) {
        const float d = __fadd_rn(p.tlin[i], __fmul_rn(jit, p.depth_delta));
        return __fdiv_rn(1.f, __fadd_rn(__fmul_rn(p.inv_start, __fsub_rn(1.f, d)), __fmul_rn(p.inv_end, d)));
    }
    const float s = p.ray_bounds[2 * gray], e = p.ray_bounds[2 * gray + 1];
    const float step = __fdiv_rn((float)i, (float)(p.Sc - 1));                    // math_utils.linspace: arange(num) / (num - 1)
    const float span = __fsub_rn(e, s);
    return __fadd_rn(__fadd_rn(s, __fmul_rn(step, span)), __fmul_rn(jit, __fdiv_rn(span, (float)(p.Sc - 1))));
}

// ray of pixel (ri, rj) of an R x R image (ray_sampler.py:33-61): direction (unit), origin = cam2world translation
__device__ __forceinline__ void ray_direction(const float* c2w, const float* K, int R, int ray, float& dx, float& dy, float& dz) {
    const float fx = K[0], fy = K[4], cxk = K[2], cyk = K[5], sk = K[1];
    const int ri = ray / R, rj = ray % R;
    const float inv = (float)(1.0 / (double)R), half = (float)(0.5 / (double)R);
    const float x_cam = __fadd_rn(__fmul_rn((float)rj, inv), half);
    const float y_cam = __fadd_rn(__fmul_rn((float)ri, inv), half);
    const float x_lift = (x_cam - cxk + cyk * sk / fy - sk * y_cam / fy) / fx;
    const float y_lift = (y_cam - cyk) / fy;
    const float ox = c2w[3], oy = c2w[7], oz = c2w[11];
    dx = c2w[0] * x_lift + c2w[1] * y_lift + c2w[2] + c2w[3] - ox;
    dy = c2w[4] * x_lift + c2w[5] * y_lift + c2w[6] + c2w[7] - oy;
    dz = c2w[8] * x_lift + c2w[9] * y_lift + c2w[10] + c2w[11] - oz;
    const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
    dx /= nrm; dy /= nrm; dz /= nrm;
}

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
#if defined(RN_ABL) && RN_ABL == 1      // tuning build: every tap reads texel (0, 0) (cache hits, same instruction stream)
        tp[k] = reinterpret_cast<const float4*>(plane);
#else
        tp[k] = reinterpret_cast<const float4*>(plane + (ok ? ((int64_t)yy * PW + xx) * RN_C : 0));
#endif
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

// ------------------------------------------------------------------------------------------------------------------------------
// render_rays_kernel: ONE wavefront renders TWO rays.
//
// Decoding (the 32 -> 64 softplus -> 33 MLP at every sample) runs on the matrix pipe in fp32 (v_mfma_f32_32x32x2_f32: the same
// flop rate as packed VALU FMAs, but on a pipe the kernel otherwise leaves idle, and with the weights resident in registers as
// matrix operands instead of ~130 scalar loads per hidden unit and sample batch).  A decode pass takes 32 samples; a sample is
// owned by the lane PAIR (s, s + 32): lane half hb = lane >> 5 gathers channels 16 hb .. 16 hb + 15 of the sample's 12 texels, so
// that the gathered features ARE the B operand (k = hb) of layer 1 with no cross-lane traffic:
//   layer 1: H[j][s] = sum_c W1[j][c] F[c][s]   M = j (2 blocks of 32), N = s, K = 32 channels as 16 steps (c = 16 hb + kk), + 1 bias step
//   layer 2: O[o][s] = sum_j W2[o][j] softplus(H[j][s])   M = o (32 colour channels), N = s, K = 64: the accumulator register r of
//            block jb IS the B operand of K step (jb, r) (rows j = 32 jb + 8 (r / 4) + 4 hb + r % 4), the weights are arranged to match;
//   sigma  : one more output row, 32 VALU FMAs per lane + one cross-half add.
// Two rays x 48 samples = 3 passes without idle lanes (one ray per wave left 16 of 64 lanes idle in each of its two passes).
// The per-ray stages (ray march, importance sampling, merge, composite) run one ray per lane half.
// ------------------------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RN_CP 36       // colour row pitch in LDS (floats): 16-byte aligned rows, 36 s mod 64 distinct for 16 consecutive s

// The decoder's matrix operands as an LDS image shared by the workgroup's waves: RN_WROWS rows of 64 floats, row = one MFMA's A
// operand (lane-indexed), the last row = the sigma weights.  (Resident registers were tried first: 67 operands + the 192 registers of
// the texels in flight made hipcc spill, and a spill reload waits on vmcnt(0), i.e. on the very loads it was meant to overlap.)
//   rows  0..33 : layer 1, [hidden block jb][K step kk]: kk < 16 -> W1[32 jb + l31][16 hb + kk], kk = 16 -> bias (half 0) / 0 (half 1)
//   rows 34..65 : layer 2 colour rows, [jb][r] -> W2[colour l31][hidden 32 jb + 8 (r / 4) + 4 hb + r % 4]
//   row  66     : layer 2 bias step;   row 67 : sigma row, [hb][jb][r] at lane 32 hb + 16 jb + r
#define RN_WROWS 68
__device__ __forceinline__ void stage_decoder(const RenderParams& p, float* wimg, int tid, int nthreads) {
    for (int e = tid; e < RN_WROWS * 64; e += nthreads) {
        const int row = e >> 6, l = e & 63, l31 = l & 31, hb = l >> 5;
        float v;
        if (row < 34) {
            const int jb = row / 17, kk = row % 17;
            v = kk < 16 ? p.w1[(32 * jb + l31) * RN_C + 16 * hb + kk] : (hb == 0 ? p.b1[32 * jb + l31] : 0.f);
        } else if (row < 66) {
            const int jb = (row - 34) >> 4, r = (row - 34) & 15;
            v = p.w2[(32 * jb + 8 * (r >> 2) + 4 * hb + (r & 3)) * 34 + 1 + l31];     // output unit 1 + c = colour channel c (unit 0 = sigma, triplane_next3d.py:368-370)
        } else if (row == 66) {
            v = hb == 0 ? p.b2[1 + l31] : 0.f;
        } else {
            v = p.w2[(32 * ((l >> 4) & 1) + 8 * ((l & 15) >> 2) + 4 * hb + (l & 3)) * 34];
        }
        wimg[e] = v;
    }
}

// The same decoder for the SPLIT-bf16 route (n3d_render_opts.decoder_split_bf16; the convolutions' arithmetic): every fp32 operand as
// hi = bf16(x), lo = bf16(x - hi) and a . b = a_hi b_lo + a_lo b_hi + a_hi b_hi on v_mfma_f32_32x32x16_bf16 — 24 instructions of 8 passes per 32 samples
// instead of 67 of 16 passes (the f32-input MFMA runs at 1/16 of the bf16 rate; three split instructions are 5.3x cheaper per MAC).  Same M / N / K
// assignment as above; a K step is 16 values, lane half hb supplying slots 8 hb .. 8 hb + 7:
//   layer 1, step s: slot (hb, j) = channel 16 hb + 8 s + j — the 16 features a lane holds are its B operands of the two steps as they are;
//   layer 2, step t: slot (hb, q) = hidden unit 32 (t / 2) + 16 (t % 2) + 8 (q / 4) + 4 hb + q % 4 — the accumulator registers 8 (t % 2) .. + 7 of block t / 2
//                    ARE the B operand (after softplus and the split), no cross-lane traffic; the weights are arranged to match.
// Biases enter as the accumulators' initial values (exact).  The base-2 constants of softplus / sigmoid are folded into the weights: layer 1 computes
// h' = log2(e) h, the hidden activation is a = log2(1 + 2^h') = softplus(h) / ln 2, the colour rows of layer 2 are NEGATED and their bias scaled by -log2(e)
// (ln 2 log2(e) = 1: o' = -log2(e) o exactly as the sigmoid's exponent wants it), the sigma row is scaled by ln 2: two multiplies less per hidden unit, one per colour
// (with the MFMAs cheap the kernel is VALU-bound: 341 -> 321 us).  Image: 16 rows of 64 x 16 bytes (bf16x8 per lane), then 160 floats:
//   rows 0..7 : layer 1 [jb][step][hi|lo];  rows 8..15 : layer 2 [t][hi|lo];  floats: b1 [hb][jb][16], b2 colours [hb][16], sigma weights [hb][jb][16]
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define RN_SPLIT_ROWS 16
#define RN_LOG2E 1.44269504f
#define RN_LN2 0.693147181f
static_assert(RN_SPLIT_ROWS * 64 * 4 + 160 <= RN_WROWS * 64, "the split image fits the fp32 image's LDS");
__device__ __forceinline__ void split8_rn(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v[i];
        hi[i] = h;
        lo[i] = (__bf16)(v[i] - (float)h);
    }
}
__device__ __forceinline__ void stage_decoder_split(const RenderParams& p, float* wimg, int tid, int nthreads) {
    bf16x8* img = reinterpret_cast<bf16x8*>(wimg);
    for (int e = tid; e < RN_SPLIT_ROWS * 64; e += nthreads) {
        const int row = e >> 6, l = e & 63, l31 = l & 31, hb = l >> 5;
        float v[8];
        if (row < 8) {
            const int jb = row >> 2, step = (row >> 1) & 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = RN_LOG2E * p.w1[(32 * jb + l31) * RN_C + 16 * hb + 8 * step + j];
        } else {
            const int t = (row - 8) >> 1;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = -p.w2[(32 * (t >> 1) + 16 * (t & 1) + 8 * (q >> 2) + 4 * hb + (q & 3)) * 34 + 1 + l31];
        }
        bf16x8 hi, lo;
        split8_rn(v, hi, lo);
        img[e] = (row & 1) ? lo : hi;
    }
    float* fb = wimg + RN_SPLIT_ROWS * 64 * 4;
    for (int e = tid; e < 160; e += nthreads) {
        float v;
        if (e < 64) { const int hb = e >> 5, jb = (e >> 4) & 1, r = e & 15; v = RN_LOG2E * p.b1[32 * jb + 8 * (r >> 2) + 4 * hb + (r & 3)]; }
        else if (e < 96) { const int k = e - 64, hb = k >> 4, r = k & 15; v = -RN_LOG2E * p.b2[1 + 8 * (r >> 2) + 4 * hb + (r & 3)]; }
        else { const int k = e - 96, hb = k >> 5, jb = (k >> 4) & 1, r = k & 15; v = RN_LN2 * p.w2[(32 * jb + 8 * (r >> 2) + 4 * hb + (r & 3)) * 34]; }
        fb[e] = v;
    }
}

// The waves of a workgroup share nothing but the decoder image; inside a wave LDS operations execute in issue order, so the
// stages of a ray need a compiler-level fence only, not an s_barrier across the workgroup.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// softplus / sigmoid for the matrix-pipe decoder: raw v_exp_f32 / v_log_f32 / v_rcp_f32 (1 + e^x >= 1: no denormal handling needed)
__device__ __forceinline__ float softplus_raw(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 1.44269504f);
    return x > 20.f ? x : __builtin_amdgcn_logf(1.f + e) * 0.693147181f;
}
__device__ __forceinline__ float sigmoid_raw(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.44269504f)); }

// One decode pass in two halves, so that the NEXT pass's texel gathers are in flight while this pass runs the decoder (the wave has
// its SIMD to itself: nothing else hides an L2 round trip).
struct PassFetch {     // a pass between its two halves: 48 texel quarters in flight + what the second half needs
    float4 v[3][4][4];
    const float4* tp[3][4];   // the 12 taps' texels (this lane's half: + 4 hb float4)
    float tw[3][4];
    int slot;          // sample slot in the ray's LDS arrays, -1 = idle lane pair
    int q;             // ray of the wave (0 / 1)
};

// first half: the sample's position -> 12 tap addresses / weights; the 48 16-byte loads of this lane's 16 channels are issued in
// four parts of 12 (pass_load) so that the caller can spread them under the previous pass's decoder: four waves share a CU's
// texture path, a burst of 48 x 64 scattered 16-byte accesses per wave keeps it busy for ~6,000 cycles and a wave that issues
// them back to back stalls on the full queue instead of multiplying.
__device__ __forceinline__ void pass_taps(const RenderParams& p, int n, int hb, float px, float py, float pz, PassFetch& F) {
    const float cx = p.coord_scale * px, cy = p.coord_scale * py, cz = p.coord_scale * pz;
    const float* base = p.planes + (int64_t)n * 3 * p.PH * p.PW * RN_C + 16 * hb;
    const int64_t ps = (int64_t)p.PH * p.PW * RN_C;
    plane_taps(base, p.PH, p.PW, cx, cy, F.tp[0], F.tw[0]);            // plane 0: (x, y)
    plane_taps(base + ps, p.PH, p.PW, cx, cz, F.tp[1], F.tw[1]);       // plane 1: (x, z)
    plane_taps(base + 2 * ps, p.PH, p.PW, cz, cy, F.tp[2], F.tw[2]);   // plane 2: (z, y)   (renderer.py:42-44)
}
__device__ __forceinline__ void pass_load(PassFetch& F, int part) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int pl = (3 * part + c) >> 2, k = (3 * part + c) & 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) F.v[pl][k][q] = F.tp[pl][k][q];
    }
}

// second half, part 1: wait for the texels, blend them (the 12 taps' weights, mean over the planes) -> this lane's 16 features
__device__ __forceinline__ void pass_blend(const PassFetch& F, float (&f)[16]) {
    f32x2 f2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) f2[c] = f32x2{0.f, 0.f};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x2 w2 = f32x2{F.tw[pl][k], F.tw[pl][k]};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f2[2 * q] = __builtin_elementwise_fma(w2, f32x2{F.v[pl][k][q].x, F.v[pl][k][q].y}, f2[2 * q]);
                f2[2 * q + 1] = __builtin_elementwise_fma(w2, f32x2{F.v[pl][k][q].z, F.v[pl][k][q].w}, f2[2 * q + 1]);
            }
        }
#pragma unroll
    for (int c = 0; c < 8; ++c) { f[2 * c] = f2[c].x * (1.f / 3.f); f[2 * c + 1] = f2[c].y * (1.f / 3.f); }   // mean over the planes (triplane_next3d.py:361)
}

// ---- Round 6: COALESCED gathers.  The pass above lets every lane fetch 64 bytes of ITS sample's texels as four 16-byte loads: a wave load
// instruction then touches 64 different cache lines (one 16-byte piece each), and the texture path — address processing and tag look-ups at
// ~0.5 line per clock and CU — is what the kernel waits for: 48 loads x 64 lines x 4 waves = 12,288 line accesses per round of passes against
// 4,288 cycles of decoder MFMAs (profiles/r05_render_pmc.txt: 58.7 line accesses per wave load, 0.48 per clock, matrix pipe 0.17, one wave per
// SIMD).  Here EIGHT ADJACENT LANES fetch one texel (128 contiguous bytes): lane l = 8 a + b loads piece b of the texels of the samples
// s' = 8 j + a, j = 0..3 — the same 48 instructions and bytes per lane, 8 texels (8-16 lines) per instruction instead of 64.  The 12 tap
// offsets / weights of a sample are computed ONCE by its owner pair (the old pass's arithmetic) and handed to the eight loading lanes through
// the sample's own colour row in LDS (free until the sample's pass blends; a first version let every lane derive the taps of its four
// samples itself: +51 % VALU instructions, profiles/r06_render_gather_pmc.txt); the loading lanes blend in the loaded
// layout — 4 channels of 4 samples per lane, the same fma chain per channel as pass_blend: bit-identical features — and hands the features to
// the sample's owner pair (s, s + 32) through the sample's OWN colour row in LDS, which the pass only fills at its end: no extra LDS.
struct PassFetch2 {
    float4 v[4][12];          // [sample group j][tap = 4 plane + k]: piece b of the texel
    float tw[4][12];
    int row[4];               // LDS float index (from the wave's region) of the colour row of sample 8 j + a, -1 = no such sample
};
// taps of one plane as byte offsets into the sample's three planes (taps outside get weight 0 and offset 0: always loadable)
__device__ __forceinline__ void plane_taps_off(int plane_texel0, int PH, int PW, float gx, float gy, int (&off)[4], float (&tw)[4]) {
    const float ix = ((gx + 1.f) * (float)PW - 1.f) * 0.5f;
    const float iy = ((gy + 1.f) * (float)PH - 1.f) * 0.5f;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const bool sane = fx0 > -2.f && fx0 < (float)PW + 1.f && fy0 > -2.f && fy0 < (float)PH + 1.f;
    const int x0 = sane ? (int)fx0 : -4, y0 = sane ? (int)fy0 : -4;
    const float wx1 = ix - fx0, wy1 = iy - fy0;
    const float wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
    const float wgt[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};   // nw, ne, sw, se
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
        const bool ok = xx >= 0 && xx < PW && yy >= 0 && yy < PH;
#if defined(RN_ABL) && RN_ABL == 1
        off[k] = 0;
#else
        off[k] = ok ? (plane_texel0 + yy * PW + xx) * (RN_C * 4) : 0;
#endif
        tw[k] = ok ? wgt[k] : 0.f;
    }
}
// blend of one sample group in the loaded layout: 4 channels (piece b) of sample 8 j + a; pass_blend's arithmetic per channel
__device__ __forceinline__ f32x4 pass_blend2(const PassFetch2& F, int j) {
    f32x2 lo = f32x2{0.f, 0.f}, hi = f32x2{0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        const f32x2 w2 = f32x2{F.tw[j][t], F.tw[j][t]};
        lo = __builtin_elementwise_fma(w2, f32x2{F.v[j][t].x, F.v[j][t].y}, lo);
        hi = __builtin_elementwise_fma(w2, f32x2{F.v[j][t].z, F.v[j][t].w}, hi);
    }
    return f32x4{lo.x * (1.f / 3.f), lo.y * (1.f / 3.f), hi.x * (1.f / 3.f), hi.y * (1.f / 3.f)};
}

#ifdef RN_TRACE   // tuning builds only (tools/build_variant.sh trace render.hip -DRN_TRACE): stage time stamps of every wave
__device__ long long rn_trace_buf[16384 * 16];
#define RN_STAMP2(k) do { if (g0 == 32 && slot0 == 0) RN_STAMP(k); } while (0)
#define RN_STAMP3(k) do { if (rn_tr) RN_STAMP(k); } while (0)
#define RN_STAMP(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.y == 0 && blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) < 16384) rn_trace_buf[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
extern "C" int n3d_render_trace_dump(double* avg, int nwg) {
    static long long host[16384 * 16];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(rn_trace_buf), sizeof(host)) != hipSuccess) return 1;
    for (int k = 0; k < 16; ++k) {
        double sum = 0;
        for (int w = 0; w < nwg; ++w) sum += (double)(host[w * 16 + k] - host[w * 16]);
        avg[k] = sum / nwg;
    }
    return 0;
}
#else
#define RN_STAMP(k)
#define RN_STAMP2(k)
#define RN_STAMP3(k)
#endif

// second half, part 2: 32 -> 64 softplus -> 33 on the matrix pipe -> rgb[r] = colour channel 8 (r / 4) + 4 hb + r % 4, sigma.
// Straight-line code, four quarters of 17 / 17 / 16 / 17 dependent MFMAs with the softplus batches between them.  (A pinned
// sched_group_barrier order that alternates one MFMA with one hidden unit's softplus was measured: no difference — the pass is
// not bound by the decoder's instruction order.)
// LOADS: the next pass's 48 texel loads are issued from here (`Fn`), four parts spread over the quarters.
template <bool LOADS, typename LoadFn>
__device__ __forceinline__ void pass_mlp(const float* wl /* decoder image + lane */, float bsig, int lane, const float (&f)[16], float (&rgb)[16], float& sigma,
                                         LoadFn pass_load_part, bool rn_tr = false) {
    const int hb = lane >> 5;
    const float one0 = hb == 0 ? 1.f : 0.f;
    const float* wsig = wl - lane + 67 * 64 + hb * 32;
    f32x16 h0, h1, o;
#pragma unroll
    for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; o[r] = 0.f; }
    if (LOADS) { pass_load_part(0); pass_load_part(1); }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[kk * 64], f[kk], h0, 0, 0, 0);
    h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[16 * 64], one0, h0, 0, 0, 0);
    RN_STAMP3(12);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(17 + kk) * 64], f[kk], h1, 0, 0, 0);
    h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(17 + 16) * 64], one0, h1, 0, 0, 0);
    RN_STAMP3(13);
    float hs0[16], hs1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) hs0[r] = softplus_raw(h0[r]);
    if (LOADS) pass_load_part(2);
    float sp = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(34 + r) * 64], hs0[r], o, 0, 0, 0);
        sp = fmaf(wsig[r], hs0[r], sp);
    }
    RN_STAMP3(14);
#pragma unroll
    for (int r = 0; r < 16; ++r) hs1[r] = softplus_raw(h1[r]);
    if (LOADS) pass_load_part(3);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(34 + 16 + r) * 64], hs1[r], o, 0, 0, 0);
        sp = fmaf(wsig[16 + r], hs1[r], sp);
    }
    o = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[66 * 64], one0, o, 0, 0, 0);
    RN_STAMP3(15);
    sigma = sp + __shfl_xor(sp, 32, 64) + bsig;
#pragma unroll
    for (int r = 0; r < 16; ++r) rgb[r] = sigmoid_raw(o[r]) * (1.f + 2.f * 0.001f) - 0.001f;
}

// the split-bf16 decoder (stage_decoder_split's image): same interface, same interleaving of the next pass's loads
template <bool LOADS, typename LoadFn>
__device__ __forceinline__ void pass_mlp_split(const float* wimg, float bsig, int lane, const float (&f)[16], float (&rgb)[16], float& sigma,
                                               LoadFn pass_load_part, bool rn_tr = false) {
    const int hb = lane >> 5;
    const bf16x8* A = reinterpret_cast<const bf16x8*>(wimg) + lane;        // + row * 64
    const float* fb = wimg + RN_SPLIT_ROWS * 64 * 4;
    const f32x4* b1p = reinterpret_cast<const f32x4*>(fb + hb * 32);
    const f32x4* b2p = reinterpret_cast<const f32x4*>(fb + 64 + hb * 16);
    const float* wsig = fb + 96 + hb * 32;
    f32x16 h0, h1, o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t0 = b1p[q], t1 = b1p[4 + q], t2 = b2p[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) { h0[4 * q + k] = t0[k]; h1[4 * q + k] = t1[k]; o[4 * q + k] = t2[k]; }
    }
    if (LOADS) { pass_load_part(0); pass_load_part(1); }
    bf16x8 xh[2], xl[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f[8 * st + j];
        split8_rn(v, xh[st], xl[st]);
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const bf16x8 a0h = A[(st * 2) * 64], a0l = A[(st * 2 + 1) * 64], a1h = A[(4 + st * 2) * 64], a1l = A[(4 + st * 2 + 1) * 64];
        h0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, xh[st], h0, 0, 0, 0);
        h1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, xh[st], h1, 0, 0, 0);
        h0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, xl[st], h0, 0, 0, 0);
        h1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, xl[st], h1, 0, 0, 0);
        h0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, xh[st], h0, 0, 0, 0);
        h1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, xh[st], h1, 0, 0, 0);
    }
    RN_STAMP3(13);
    float sp = 0.f;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        float hs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {                                     // log2(1 + 2^h'), = h' beyond softplus' threshold 20 (torch.nn.Softplus' default)
            const float x = jb ? h1[r] : h0[r];
            hs[r] = x > 20.f * RN_LOG2E ? x : __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(x));
        }
        if (LOADS) pass_load_part(2 + jb);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = hs[8 * tt + q];
            bf16x8 bh, bl;
            split8_rn(v, bh, bl);
            const int t = 2 * jb + tt;
            const bf16x8 ah = A[(8 + 2 * t) * 64], al = A[(8 + 2 * t + 1) * 64];
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, o, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sp = fmaf(wsig[16 * jb + r], hs[r], sp);
        RN_STAMP3(14 + jb);
    }
    sigma = sp + __shfl_xor(sp, 32, 64) + bsig;
#pragma unroll
    for (int r = 0; r < 16; ++r) rgb[r] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(o[r])) * (1.f + 2.f * 0.001f) - 0.001f;      // o = -log2(e) x (folded): sigmoid(x)
}

// LDS of one ray: arrays sized for M = Sc + Sf samples
struct RayLds {
    float* col;     // [M][RN_CP]  colour rows
    float* sig;     // [M]
    float* dep;     // [M]
    float* wgt;     // [M]   march weights
    float* fac;     // [M]   1 - alpha + 1e-10
    float* trn;     // [M]   exclusive cumprod
    float* cdf;     // [M]
    float* bins;    // [M]
    int* order;     // [M]
};
__host__ __device__ constexpr int ray_lds_floats(int M) { return M * (RN_CP + 8); }

// Exclusive prefix product / sum of src[0..n) -> dst[0..n) on the 32 lanes of one ray (n <= 256): every lane folds a contiguous
// chunk, the chunk totals are scanned across the lanes with five shuffles.  (The order of the floating-point operations differs
// from a sequential torch.cumprod / cumsum on the CPU exactly as the reference's own parallel scan on the GPU does.)
template <bool PROD>
__device__ __forceinline__ void scan_half(const float* src, float* dst, int n, int l31) {
    const float id = PROD ? 1.f : 0.f;
    const int C = (n + 31) >> 5, i0 = l31 * C;
    float v[8], total = id;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        v[c] = c < C && i0 + c < n ? src[i0 + c] : id;
        total = PROD ? total * v[c] : total + v[c];
    }
    float inc = total;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const float t = __shfl_up(inc, off, 32);
        if (l31 >= off) inc = PROD ? t * inc : t + inc;
    }
    float run = __shfl_up(inc, 1, 32);
    if (l31 == 0) run = id;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c < C && i0 + c < n) { dst[i0 + c] = run; run = PROD ? run * v[c] : run + v[c]; }
}

// unify_samples (renderer.py:164-182): stable rank of each of the ray's M samples (<= 32 KP) in the merged depth order ->
// order[rank] = sample.  Lane l31 ranks samples l31 + 32 m; one broadcast LDS read per compared sample serves all of them.
template <int KP>
__device__ __forceinline__ void rank_half(const RayLds& L, int M, int l31) {
    // Round 6: ONE strict count per sample (no second "<=" count: a tie shows up as an order slot nobody wrote), FOUR partial counters per sample so that the
    // sixteen compare-and-add steps of a batch are four dependent chains of four instead of one of sixteen (the wave has its SIMD to itself: the chain length is
    // the stage's time — 13.3 k of a wave's 133 k cycles before, profiles/r06_render_gather_pmc.txt), and the rank of every sample kept (rank[]: the composite's coefficients)
    float d[KP];
    int c0[KP], c1[KP], c2[KP], c3[KP];
    int* rank = reinterpret_cast<int*>(L.bins);                            // (bins is dead behind the importance depths)
#pragma unroll
    for (int m = 0; m < KP; ++m) {
        d[m] = l31 + 32 * m < M ? L.dep[l31 + 32 * m] : INFINITY; c0[m] = 0; c1[m] = 0; c2[m] = 0; c3[m] = 0;
        if (l31 + 32 * m < M) L.order[l31 + 32 * m] = -1;
    }
    for (int q0 = 0; q0 < M; q0 += 16) {                                  // sixteen compared samples per LDS round trip
        float dq[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) dq[k] = q0 + k < M ? L.dep[q0 + k] : INFINITY;
#pragma unroll
        for (int k = 0; k < 16; k += 4)
#pragma unroll
            for (int m = 0; m < KP; ++m) { c0[m] += dq[k] < d[m] ? 1 : 0; c1[m] += dq[k + 1] < d[m] ? 1 : 0; c2[m] += dq[k + 2] < d[m] ? 1 : 0; c3[m] += dq[k + 3] < d[m] ? 1 : 0; }
    }
    int lt[KP];
#pragma unroll
    for (int m = 0; m < KP; ++m) lt[m] = (c0[m] + c1[m]) + (c2[m] + c3[m]);
    wave_sync();                                                          // (order[] = -1 everywhere)
#pragma unroll
    for (int m = 0; m < KP; ++m)
        if (l31 + 32 * m < M) L.order[lt[m]] = l31 + 32 * m;
    wave_sync();
    // no two samples of the ray at the same depth (the normal case): every slot is written.  Equal depths are ordered by sample index (torch.sort is stable there):
    // counted exactly in a second pass only when some slot stayed empty
    bool tie = false;
#pragma unroll
    for (int m = 0; m < KP; ++m) tie = tie || (l31 + 32 * m < M && L.order[l31 + 32 * m] < 0);
    if (__any(tie)) {
#pragma unroll
        for (int m = 0; m < KP; ++m) lt[m] = 0;
        for (int q = 0; q < M; ++q) {
            const float dq = L.dep[q];
#pragma unroll
            for (int m = 0; m < KP; ++m) lt[m] += (dq < d[m] || (dq == d[m] && q < l31 + 32 * m)) ? 1 : 0;
        }
        wave_sync();
#pragma unroll
        for (int m = 0; m < KP; ++m)
            if (l31 + 32 * m < M) L.order[lt[m]] = l31 + 32 * m;
    }
#pragma unroll
    for (int m = 0; m < KP; ++m)
        if (l31 + 32 * m < M) rank[l31 + 32 * m] = lt[m];
}

// mid-point ray march over `count` samples addressed through idx(i) (ray_marcher.py:28-46); fills wgt[0..count-2].  Every ray
// of the wave runs this on its own 32 lanes (l31 = lane within the half); the barriers are the wave's.
template <typename IdxFn>
__device__ __forceinline__ void march_weights(const RayLds& L, int count, int l31, IdxFn idx) {
    for (int i = l31; i < count - 1; i += 32) {
        const int a = idx(i), b = idx(i + 1);
        const float delta = L.dep[b] - L.dep[a];
        const float dm = softplus_f((L.sig[a] + L.sig[b]) / 2.f - 1.f);
        const float alpha = 1.f - expf(-(dm * delta));
        L.wgt[i] = alpha;
        L.fac[i] = 1.f - alpha + 1e-10f;
    }
    wave_sync();
    scan_half<true>(L.fac, L.trn, count - 1, l31);        // exclusive cumprod (ray_marcher.py:43)
    wave_sync();
    for (int i = l31; i < count - 1; i += 32) L.wgt[i] = L.wgt[i] * L.trn[i];
    wave_sync();
}


// Two rays per wave: no idle lanes in the decode passes; the colours of 2 x (Sc + Sf) samples live in LDS, which leaves room for
// one wave per SIMD (variants with two waves per SIMD — one ray per wave, or the colours parked in a global workspace — were
// measured equal or slower in round 2 and are gone: DESIGN.md 3.2).
constexpr int RPW = 2;
template <bool COALESCED, bool SPLIT>
__device__ __forceinline__ void render_rays_body(const RenderParams& p, float* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hb = lane >> 5;
    const int n = blockIdx.y;
    const int R = p.R, RR = R * R, Sc = p.Sc, Sf = p.Sf, M = Sc + Sf;
    if (SPLIT) stage_decoder_split(p, smem, threadIdx.x, blockDim.x);
    else stage_decoder(p, smem, threadIdx.x, blockDim.x);
    __syncthreads();                                                      // the only workgroup-wide barrier
    const int ray0 = (blockIdx.x * (blockDim.x >> 6) + wave) * RPW;       // this wave's rays: ray0 (, ray0 + 1: may not exist)
    if (ray0 >= RR) return;
    const int nrays = min(RPW, RR - ray0);
    RN_STAMP(0);
    const float* wl = smem + lane;
    const float bsig = p.b2[0];
    const int cp = RN_CP;                                                // colour row pitch
    const int rlf = ray_lds_floats(M);                                    // LDS floats per ray
    float* wsm = smem + RN_WROWS * 64 + wave * RPW * rlf;                 // this wave's rays
    RayLds Ls[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        RayLds& L = Ls[q];
        float* base = wsm + (q < RPW ? q : 0) * rlf;
        L.col = base;
        L.sig = base + M * RN_CP;
        L.dep = L.sig + M; L.wgt = L.dep + M; L.fac = L.wgt + M; L.trn = L.fac + M; L.cdf = L.trn + M; L.bins = L.cdf + M;
        L.order = reinterpret_cast<int*>(L.bins + M);
    }

    // ---- the two rays (ray_sampler.py:33-61); ray index m = i*R + j, x from j, y from i.  Every lane holds both.
    const float* c2w = p.cam2world + n * 16;
    const float* K = p.intrinsics + n * 9;
    const float ox = c2w[3], oy = c2w[7], oz = c2w[11];
    float rdx[2], rdy[2], rdz[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) ray_direction(c2w, K, R, min(ray0 + q, RR - 1), rdx[q], rdy[q], rdz[q]);

    // decode `cnt` samples per ray, slots slot0 .. slot0 + cnt - 1, whose depths are in the rays' dep[] arrays.  Passes of 32 samples
    // over the concatenated sample lists of the wave's rays; pass k + 1's gathers are issued before pass k's decoder runs.
    auto taps = [&](int g0, int cnt, int slot0, PassFetch& F) {
        const int g = min(g0 + l31, nrays * cnt - 1);                     // (lanes beyond the last sample REPEAT it: see decode_all2)
        F.q = g >= cnt ? 1 : 0;
        const int i = g - F.q * cnt;
        F.slot = slot0 + i;
        const float t = (F.q ? Ls[1] : Ls[0]).dep[slot0 + i];
        const float dx = F.q ? rdx[1] : rdx[0], dy = F.q ? rdy[1] : rdy[0], dz = F.q ? rdz[1] : rdz[0];
        pass_taps(p, n, hb, __fadd_rn(ox, __fmul_rn(t, dx)), __fadd_rn(oy, __fmul_rn(t, dy)), __fadd_rn(oz, __fmul_rn(t, dz)), F);
    };
    auto store_pass = [&](int slot, int q, const float (&rgb)[16], float sigma) {
        {                                                                 // (every lane owns a sample: the lanes beyond the last one repeat it, and store the same words again)
            const RayLds& L = q ? Ls[1] : Ls[0];
            if (hb == 0) {
                if (p.noise_c) {                                          // density noise (renderer.py:152-153): one draw per decoded sample
                    const int64_t gr = (int64_t)n * RR + ray0 + q;
                    sigma += (slot < Sc ? p.noise_c[gr * Sc + slot] : p.noise_f[gr * Sf + (slot - Sc)]) * p.noise_scale;
                }
                L.sig[slot] = sigma;
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
                *reinterpret_cast<f32x4*>(L.col + slot * cp + 8 * a + 4 * hb) = f32x4{rgb[4 * a], rgb[4 * a + 1], rgb[4 * a + 2], rgb[4 * a + 3]};
        }
    };
    auto decode_all = [&](int cnt, int slot0) {
        const int total = nrays * cnt;
        PassFetch F;
        taps(0, cnt, slot0, F);
#pragma unroll
        for (int part = 0; part < 4; ++part) pass_load(F, part);
        for (int g0 = 0; g0 < total; g0 += 32) {
            __builtin_amdgcn_sched_barrier(0);
            float f[16];
            pass_blend(F, f);                                             // waits for this pass's texels ...
            const int slot = F.slot, q = F.q;
            __builtin_amdgcn_sched_barrier(0);                            // (... all of them, before ...)
            RN_STAMP2(10);
            const bool more = g0 + 32 < total;
            if (more) taps(g0 + 32, cnt, slot0, F);                       // ... their registers take the next pass's, loaded under the decoder
            RN_STAMP2(11);
            float rgb[16], sigma;
            auto part = [&](int k) { pass_load(F, k); };
            if (SPLIT) {
                if (more) pass_mlp_split<true>(smem, bsig, lane, f, rgb, sigma, part, g0 == 32 && slot0 == 0);
                else pass_mlp_split<false>(smem, bsig, lane, f, rgb, sigma, part, g0 == 32 && slot0 == 0);
            } else {
                if (more) pass_mlp<true>(wl, bsig, lane, f, rgb, sigma, part, g0 == 32 && slot0 == 0);
                else pass_mlp<false>(wl, bsig, lane, f, rgb, sigma, part, g0 == 32 && slot0 == 0);
            }
            store_pass(slot, q, rgb, sigma);
            RN_STAMP2(9);
        }
    };
    // the same with coalesced gathers (PassFetch2): sample group j of the NEXT pass = its 12 tap offsets / weights + its 12 loads, issued as part j under this pass's decoder
    const __amdgpu_buffer_rsrc_t r_planes = __builtin_amdgcn_make_buffer_rsrc((void*)(p.planes + (int64_t)n * 3 * p.PH * p.PW * RN_C), 0, 3 * p.PH * p.PW * RN_C * 4, 0x00020000);
    const int ga = lane >> 3, gb16 = (lane & 7) * 16;
    // owner pair (s, s + 32) of sample gF + s: the 12 tap offsets / weights ONCE per sample (round 5's arithmetic), parked in the sample's OWN colour row
    // (unwritten until its pass blends): offsets in floats 0..11 (written by the lane of half 0), weights in 12..23 (half 1)
    auto stage_taps = [&](int gF, int cnt, int slot0) {
        const int g = min(gF + l31, nrays * cnt - 1);
        const int q = g >= cnt ? 1 : 0;
        const int i = g - q * cnt;
        const float t = (q ? Ls[1] : Ls[0]).dep[slot0 + i];
        const float dx = q ? rdx[1] : rdx[0], dy = q ? rdy[1] : rdy[0], dz = q ? rdz[1] : rdz[0];
        const float cx = p.coord_scale * __fadd_rn(ox, __fmul_rn(t, dx)), cy = p.coord_scale * __fadd_rn(oy, __fmul_rn(t, dy)), cz = p.coord_scale * __fadd_rn(oz, __fmul_rn(t, dz));
        int off[3][4];
        float tw[3][4];
        plane_taps_off(0, p.PH, p.PW, cx, cy, off[0], tw[0]);                       // plane 0: (x, y)
        plane_taps_off(p.PH * p.PW, p.PH, p.PW, cx, cz, off[1], tw[1]);             // plane 1: (x, z)
        plane_taps_off(2 * p.PH * p.PW, p.PH, p.PW, cz, cy, off[2], tw[2]);         // plane 2: (z, y)   (renderer.py:42-44)
        {
            float* row = wsm + q * rlf + (slot0 + i) * cp + 12 * hb;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *reinterpret_cast<f32x4*>(row + 4 * pl) = hb ? f32x4{tw[pl][0], tw[pl][1], tw[pl][2], tw[pl][3]}
                                                             : f32x4{__int_as_float(off[pl][0]), __int_as_float(off[pl][1]), __int_as_float(off[pl][2]), __int_as_float(off[pl][3])};
        }
        wave_sync();
    };
    // reader lane 8 a + b: sample group j = the sample gF + 8 j + a; its taps come out of the sample's row, its 12 loads fetch piece b of the 12 texels
    auto fetch_group = [&](int g0, int cnt, int slot0, int j, PassFetch2& F) {
        const int g = min(g0 + 8 * j + ga, nrays * cnt - 1);
        const int q = g >= cnt ? 1 : 0;
        const int i = g - q * cnt;
        const int row = q * rlf + (slot0 + i) * cp;
        F.row[j] = row;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const f32x4 o4 = *reinterpret_cast<const f32x4*>(wsm + row + 4 * pl), w4 = *reinterpret_cast<const f32x4*>(wsm + row + 12 + 4 * pl);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                F.tw[j][4 * pl + k] = w4[k];
                F.v[j][4 * pl + k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_planes, __float_as_int(o4[k]) + gb16, 0, 0));
            }
        }
    };
    // A partly filled last pass has NO idle lanes: the lanes beyond the last sample repeat it (g = min(g, total - 1) everywhere — taps, loads, blend, decoder, store: the same words
    // written twice).  The decode loop then has no lane-dependent branch at all: an MFMA takes its A rows from all 64 lanes and honours EXEC, and builds of this kernel whose
    // partly filled passes ran under lane-dependent control flow returned wrong colours on 24 + 24 / 64 + 17 / 8 + 8 samples (profiles/r06_render_split_ab.txt).
    auto decode_all2 = [&](int cnt, int slot0) {
        const int total = nrays * cnt;
        PassFetch2 F;
        stage_taps(0, cnt, slot0);
#pragma unroll
        for (int j = 0; j < 4; ++j) fetch_group(0, cnt, slot0, j, F);
        if (32 < total) stage_taps(32, cnt, slot0);                       // taps run ONE pass ahead of the loads that use them: the LDS round trip is off the decoder's path
        for (int g0 = 0; g0 < total; g0 += 32) {
            __builtin_amdgcn_sched_barrier(0);
            // this pass's owner pair: sample g0 + l31 -> (ray q, slot)
            const int g = min(g0 + l31, total - 1);
            const int q = g >= cnt ? 1 : 0;
            const int slot = slot0 + g - q * cnt;
            // blend in the loaded layout (waits for the texels), park the features in the samples' own colour rows, read the owner's 16 back
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 f4 = pass_blend2(F, j);
                *reinterpret_cast<f32x4*>(wsm + F.row[j] + (lane & 7) * 4) = f4;
            }
            wave_sync();
            float f[16];
            {
                const float* own = wsm + q * rlf + slot * cp + 16 * hb;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(own + 4 * a);
                    f[4 * a] = t4[0]; f[4 * a + 1] = t4[1]; f[4 * a + 2] = t4[2]; f[4 * a + 3] = t4[3];
                }
            }
            wave_sync();                                                  // (the rows are rewritten by store_pass below: reads first)
            __builtin_amdgcn_sched_barrier(0);
            RN_STAMP2(10);
            const bool more = g0 + 32 < total;
            float rgb[16], sigma;
            auto part = [&](int k) { fetch_group(g0 + 32, cnt, slot0, k, F); if (k == 3 && g0 + 64 < total) stage_taps(g0 + 64, cnt, slot0); };
            RN_STAMP2(11);
            if (SPLIT) {
                if (more) pass_mlp_split<true>(smem, bsig, lane, f, rgb, sigma, part, g0 == 32 && slot0 == 0);
                else pass_mlp_split<false>(smem, bsig, lane, f, rgb, sigma, part, g0 == 32 && slot0 == 0);
            } else {
                if (more) pass_mlp<true>(wl, bsig, lane, f, rgb, sigma, part, g0 == 32 && slot0 == 0);
                else pass_mlp<false>(wl, bsig, lane, f, rgb, sigma, part, g0 == 32 && slot0 == 0);
            }
            store_pass(slot, q, rgb, sigma);
            RN_STAMP2(9);
        }
    };

    // ---- coarse pass: stratified depths (renderer.py:203-205) -> dep[0 .. Sc), then decode into slots 0 .. Sc
    RN_STAMP(1);
    {
        const float* jit0 = p.jitter + ((int64_t)n * RR + ray0) * Sc;
        for (int g = lane; g < nrays * Sc; g += 64) {
            const int q = g >= Sc ? 1 : 0, i = g - q * Sc;
            (q ? Ls[1] : Ls[0]).dep[i] = coarse_depth(p, (int64_t)n * RR + ray0 + q, i, jit0[g]);
        }
    }
    wave_sync();
    if (COALESCED) decode_all2(Sc, 0); else decode_all(Sc, 0);
    wave_sync();
    RN_STAMP(2);

    // ---- per-ray stages: lane half hb works on ray hb (a missing second ray repeats the first; its results are not stored)
    const int myq = hb < nrays ? hb : 0;
    const RayLds& L = myq ? Ls[1] : Ls[0];
    const bool store = hb < nrays;
    const int ray = ray0 + myq;
    int count = Sc;
    if (Sf > 0) {
        march_weights(L, Sc, l31, [](int i) { return i; });
        RN_STAMP(3);
        // ---- importance depths (renderer.py:209-268)
        const int Lw = Sc - 1;       // number of march weights
        const int Np = Sc - 3;       // pdf bins (weights[:, 1:-1])
        for (int i = l31; i < Lw; i += 32) {
            const float wl = i > 0 ? L.wgt[i - 1] : -INFINITY, wc = L.wgt[i], wr = i + 1 < Lw ? L.wgt[i + 1] : -INFINITY;
            const float m0 = fmaxf(wl, wc), m1 = fmaxf(wc, wr);       // max_pool1d(2,1,pad=1)
            L.fac[i] = (m0 + m1) / 2.f + 0.01f;                        // avg_pool1d(2,1) + 0.01
            L.bins[i] = 0.5f * (L.dep[i] + L.dep[i + 1]);
        }
        wave_sync();
        float part = 0.f;
        for (int k = l31; k < Np; k += 32) part += L.fac[k + 1] + 1e-5f;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        for (int k = l31; k <= Np; k += 32) L.trn[k] = k < Np ? (L.fac[k + 1] + 1e-5f) / part : 0.f;      // pdf (renderer.py:243)
        wave_sync();
        scan_half<false>(L.trn, L.cdf, Np + 1, l31);                    // cdf[0] = 0, cdf[k + 1] = pdf[0] + .. + pdf[k] (:244-245)
        wave_sync();
        const float* uu = p.u + ((int64_t)n * RR + ray) * Sf;
        for (int jb = 0; jb < Sf; jb += 64) {                         // two draws per lane at once: one walk over the cdf serves both (round 6)
          const int ja = jb + l31, jc = jb + 32 + l31;
          const float ua = ja < Sf ? uu[ja] : 0.f, uc = jc < Sf ? uu[jc] : 0.f;
          int a0 = 0, a1 = 0, c0 = 0, c1 = 0;                         // searchsorted(cdf, u, right=True): two partial counts per draw
          int k = 0;
          for (; k + 1 <= Np; k += 2) {
              const float e0 = L.cdf[k], e1 = L.cdf[k + 1];
              a0 += (e0 <= ua) ? 1 : 0; a1 += (e1 <= ua) ? 1 : 0; c0 += (e0 <= uc) ? 1 : 0; c1 += (e1 <= uc) ? 1 : 0;
          }
          for (; k <= Np; ++k) { const float e0 = L.cdf[k]; a0 += (e0 <= ua) ? 1 : 0; c0 += (e0 <= uc) ? 1 : 0; }
          for (int half2 = 0; half2 < 2; ++half2) {
            const int j = half2 ? jc : ja;
            if (j >= Sf) continue;
            const float u = half2 ? uc : ua;
            const int ind = half2 ? c0 + c1 : a0 + a1;
            const int below = max(ind - 1, 0), above = min(ind, Np);
            const float cc0 = L.cdf[below], cc1 = L.cdf[above], b0 = L.bins[below], b1 = L.bins[above];
            float denom = cc1 - cc0;
            if (denom < 1e-5f) denom = 1.f;
            float dj = __fadd_rn(b0, __fmul_rn((u - cc0) / denom, (b1 - b0)));
            if (p.fine_out && store) p.fine_out[((int64_t)n * RR + ray) * Sf + j] = dj;
            if (p.fine_in) dj = p.fine_in[((int64_t)n * RR + ray) * Sf + j];
            L.dep[Sc + j] = dj;
          }
        }
        wave_sync();
        RN_STAMP(4);
        // ---- second decode pass at the importance depths -> slots Sc + j
        if (COALESCED) decode_all2(Sf, Sc); else decode_all(Sf, Sc);
        wave_sync();
        RN_STAMP(5);
        // ---- unify_samples: stable rank of every sample in the merged order (renderer.py:164-182)
        if (M <= 96) rank_half<3>(L, M, l31); else rank_half<8>(L, M, l31);
        wave_sync();
        RN_STAMP(6);
        count = M;
        const int* ord = L.order;
        march_weights(L, M, l31, [ord](int i) { return ord[i]; });
        RN_STAMP(7);
    } else {
        for (int k = l31; k < Sc; k += 32) { L.order[k] = k; reinterpret_cast<int*>(L.bins)[k] = k; }
        wave_sync();
        march_weights(L, Sc, l31, [](int i) { return i; });
    }

    // ---- composite (ray_marcher.py:48-59): the ray's 32 lanes take one colour channel each; depth and weight total as
    // lane-strided partial sums
    float acc = 0.f, dacc = 0.f, wt = 0.f;
    {
        // Round 6: sum_i w_i (c_ord(i) + c_ord(i+1)) / 2 regrouped PER SAMPLE: the sample of rank r carries (w_{r-1} + w_r) / 2 — the coefficients are computed once by the
        // ray's lanes (three samples each), and the channel sum walks the samples in storage order with independent LDS reads instead of chasing order[] -> colour row
        // sixteen dependent round trips at a time (8.8 k cycles per wave before).  Same products; another summation order (differences of float32 rounding).
        const int* rank = reinterpret_cast<const int*>(L.bins);
        float* coef = L.cdf;                                               // (dead behind the importance depths)
        for (int sI = l31; sI < count; sI += 32) {
            const int r = rank[sI];
            coef[sI] = 0.5f * ((r > 0 ? L.wgt[r - 1] : 0.f) + (r < count - 1 ? L.wgt[r] : 0.f));
        }
        wave_sync();
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int sI = 0;
        for (; sI + 16 <= count; sI += 16) {
            float w[16], cn[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { w[k] = coef[sI + k]; cn[k] = L.col[(sI + k) * cp + l31]; }
#pragma unroll
            for (int k = 0; k < 16; k += 4) { a0 = fmaf(w[k], cn[k], a0); a1 = fmaf(w[k + 1], cn[k + 1], a1); a2 = fmaf(w[k + 2], cn[k + 2], a2); a3 = fmaf(w[k + 3], cn[k + 3], a3); }
        }
        for (; sI < count; ++sI) a0 = fmaf(coef[sI], L.col[sI * cp + l31], a0);
        acc = (a0 + a1) + (a2 + a3);
    }
    for (int i = l31; i < count - 1; i += 32) {
        const int a = L.order[i], b = L.order[i + 1];
        dacc += L.wgt[i] * ((L.dep[a] + L.dep[b]) / 2.f);
        wt += L.wgt[i];
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) { dacc += __shfl_xor(dacc, off, 64); wt += __shfl_xor(wt, off, 64); }
    RN_STAMP(8);
    if (store) {
        if (p.white_back) acc = acc + 1.f - wt;               // ray_marcher.py:56-57
        p.feat[((int64_t)n * RN_C + l31) * RR + ray] = acc * 2.f - 1.f;
        if (l31 == 0) {
            float d = dacc / wt;
            if (isnan(d)) d = INFINITY;                       // nan_to_num(nan=inf)
            d = fminf(fmaxf(d, key2f(__float_as_uint(p.bounds[0]))), key2f(__float_as_uint(p.bounds[1])));
            p.depth[(int64_t)n * RR + ray] = d;
            if (p.wsum) p.wsum[(int64_t)n * RR + ray] = wt;
        }
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void render_rays_kernel(RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    render_rays_body<false, false>(p, smem);
}
// the coalesced gather (eight lanes per texel, PassFetch2): the launcher picks it where four waves share a CU (n3d_render_rays_ex)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void render_rays_c8_kernel(RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    render_rays_body<true, false>(p, smem);
}
// the same two with the decoder on split-bf16 MFMAs (n3d_render_opts.decoder_split_bf16)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void render_rays_s_kernel(RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    render_rays_body<false, true>(p, smem);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void render_rays_c8s_kernel(RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    render_rays_body<true, true>(p, smem);
}

// global min / max of the coarse depths over the whole batch (ray_marcher.py:54 clamps against them).  Multi-block: every
// block reduces a slice and merges with integer atomics on order-preserving keys (f2key); bounds_ws is initialised by a
// one-thread launch in front and read back through key2f by the renderer.  The depths of a ray increase with the sample index in
// every depth mode, so its first / last coarse samples are its extremes.
__global__ void render_depth_bounds_init_kernel(unsigned int* __restrict__ bounds) {
    bounds[0] = 0xffffffffu; bounds[1] = 0u;
}
__global__ __launch_bounds__(256) void render_depth_bounds_kernel(RenderParams p, int64_t rays, unsigned int* __restrict__ bounds) {
    __shared__ unsigned int smin[4], smax[4];
    unsigned int lo = 0xffffffffu, hi = 0u;
    const int Sc = p.Sc;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rays; r += (int64_t)gridDim.x * blockDim.x) {
        lo = min(lo, f2key(coarse_depth(p, r, 0, p.jitter[r * Sc])));
        hi = max(hi, f2key(coarse_depth(p, r, Sc - 1, p.jitter[r * Sc + Sc - 1])));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, (unsigned int)__shfl_xor((int)lo, off, 64)); hi = max(hi, (unsigned int)__shfl_xor((int)hi, off, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = min(lo, smin[w]); hi = max(hi, smax[w]); }
        atomicMin(bounds, lo);
        atomicMax(bounds + 1, hi);
    }
}

// ray_start = ray_end = 'auto' (renderer.py:99-106): every ray's entry / exit distance of the box [-side/2, side/2]^3
// (math_utils.get_ray_limits_box, the slab test with its operation order), then the reference's repair of the rays that miss the box:
// their start becomes the smallest, their END the largest START among the rays that hit it (sic, :103-104).
__global__ __launch_bounds__(256) void ray_limits_box_kernel(RenderParams p, float* __restrict__ rb, float half_side, unsigned int* __restrict__ keys) {
    const int RR = p.R * p.R;
    const int64_t rays = (int64_t)p.N * RR;
    __shared__ unsigned int smin[4], smax[4];
    unsigned int lo = 0xffffffffu, hi = 0u;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rays; r += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(r / RR), ray = (int)(r % RR);
        const float* c2w = p.cam2world + n * 16;
        float d[3];
        ray_direction(c2w, p.intrinsics + n * 9, p.R, ray, d[0], d[1], d[2]);
        const float o[3] = {c2w[3], c2w[7], c2w[11]};
        float t0[3], t1[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float inv = __fdiv_rn(1.f, d[a]);
            const bool neg = inv < 0.f;
            t0[a] = __fmul_rn(__fsub_rn(neg ? half_side : -half_side, o[a]), inv);
            t1[a] = __fmul_rn(__fsub_rn(neg ? -half_side : half_side, o[a]), inv);
        }
        bool valid = !(t0[0] > t1[1] || t0[1] > t1[0]);
        float tmin = fmaxf(t0[0], t0[1]), tmax = fminf(t1[0], t1[1]);
        if (tmin > t1[2] || t0[2] > tmax) valid = false;
        tmin = fmaxf(tmin, t0[2]); tmax = fminf(tmax, t1[2]);
        if (!valid) { tmin = -1.f; tmax = -2.f; }
        rb[2 * r] = tmin; rb[2 * r + 1] = tmax;
        if (tmax > tmin) { lo = min(lo, f2key(tmin)); hi = max(hi, f2key(tmin)); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, (unsigned int)__shfl_xor((int)lo, off, 64)); hi = max(hi, (unsigned int)__shfl_xor((int)hi, off, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = min(lo, smin[w]); hi = max(hi, smax[w]); }
        atomicMin(keys, lo);
        atomicMax(keys + 1, hi);
    }
}
__global__ __launch_bounds__(256) void ray_limits_fixup_kernel(float* __restrict__ rb, int64_t rays, const unsigned int* __restrict__ keys) {
    if (keys[0] == 0xffffffffu) return;                                   // no ray hits the box: left as they are (torch.any(...) false)
    const float lo = key2f(keys[0]), hi = key2f(keys[1]);
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rays; r += (int64_t)gridDim.x * blockDim.x)
        if (!(rb[2 * r + 1] > rb[2 * r])) { rb[2 * r] = lo; rb[2 * r + 1] = hi; }
}

// planes_cl[n][p][y][x][c] = dyn_p[n][c][y][x] * a + static[n][p*32+c][y][x] * (1 - a),  a = alpha[n][p][y][x]
// (training_avatar_texture/triplane_next3d.py:171-174), written channels-last for the renderer's gathers.
__global__ __launch_bounds__(256) void blend_planes_kernel(const float* __restrict__ front, const float* __restrict__ side,
                                                           const float* __restrict__ top, const float* __restrict__ stat,
                                                           const float* __restrict__ alpha, float* __restrict__ out, int N,
                                                           int H, int W, int views, int v0, int v1, int v2) {
    __shared__ float tile[RN_C][65];
    const int x0 = blockIdx.x * 64, y = blockIdx.y, np = blockIdx.z, n = np / 3, pl = np % 3;
    const float* dyn = (pl == 0 ? front : (pl == 1 ? side : top)) + (int64_t)n * RN_C * H * W;
    const float* st = stat + ((int64_t)n * 3 + pl) * RN_C * H * W;
    // alpha [N][views][H][W]: plane pl blends with the alpha image of view v_pl (views = 3, (0, 1, 2): the dense [N,3,H,W] form)
    const float* al = alpha + ((int64_t)n * views + (pl == 0 ? v0 : (pl == 1 ? v1 : v2))) * H * W + (int64_t)y * W;
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
                       planes_cl, N, H, W, 3, 0, 1, 2);
    N3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int n3d_blend_planes_views(const float* front, const float* side, const float* top, const float* stat, const float* alpha_views,
                                      float* planes_cl, int N, int H, int W, int views, int v_front, int v_side, int v_top, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && H > 0 && W > 0 && H <= 65535, "blend_planes: bad shape");
    N3D_CHECK(views >= 1 && v_front >= 0 && v_front < views && v_side >= 0 && v_side < views && v_top >= 0 && v_top < views, "blend_planes_views: view index out of range");
    if (N == 0) return 0;
    N3D_CHECK(front && side && top && stat && alpha_views && planes_cl, "blend_planes: null tensor");
    N3dProfScope prof(N3D_K_RENDER, stream, 3.0 * N * 3 * RN_C * (double)H * W, 4.0 * N * 3 * (double)H * W * (3 * RN_C + 1));
    hipLaunchKernelGGL(blend_planes_kernel, dim3(cdiv(W, 64), H, N * 3), dim3(256), 0, stream, front, side, top, stat, alpha_views,
                       planes_cl, N, H, W, views, v_front, v_side, v_top);
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

extern "C" int n3d_render_rays_ex(const float* planes_cl, const float* cam2world, const float* intrinsics, const float* tlin,
                                  const float* jitter, const float* u, const float* w1, const float* b1, const float* w2,
                                  const float* b2, float* feat, float* depth, float* wsum, float* bounds_ws, int N, int R, int Sc,
                                  int Sf, int PH, int PW, float depth_delta, float coord_scale, const n3d_render_opts* opts, n3d_stream_t stream_) {
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
    p.depth_mode = 0; p.inv_start = p.inv_end = 0.f; p.ray_bounds = nullptr; p.white_back = 0; p.noise_c = p.noise_f = nullptr; p.noise_scale = 0.f;
    p.fine_out = nullptr; p.fine_in = nullptr;
    const int64_t nrays = (int64_t)N * R * R;
    const unsigned bgrid = (unsigned)(cdiv64(nrays, 256) > 256 ? 256 : cdiv64(nrays, 256));
    unsigned int* keys = reinterpret_cast<unsigned int*>(bounds_ws);
    if (opts) {
        p.white_back = opts->white_back != 0;
        p.fine_out = opts->fine_depths_out; p.fine_in = opts->fine_depths_in;
        if (opts->density_noise > 0.f) {
            N3D_CHECK(opts->density_noise_coarse && (opts->density_noise_fine || Sf == 0), "render_rays: density_noise > 0 needs the normal draws (density_noise_coarse / _fine)");
            p.noise_c = opts->density_noise_coarse; p.noise_f = opts->density_noise_fine; p.noise_scale = opts->density_noise;
        }
        N3D_CHECK(!(opts->auto_bounds && opts->disparity_space_sampling), "render_rays: 'auto' ray bounds with disparity-space sampling is not a combination the reference can run");
        if (opts->disparity_space_sampling) {                             // tlin = linspace(0, 1, Sc), depth_delta = 1 / (Sc - 1) (the caller's, as for mode 0)
            N3D_CHECK(opts->ray_start > 0.f && opts->ray_end > 0.f, "render_rays: disparity-space sampling needs positive ray_start / ray_end");
            p.depth_mode = 1;
            p.inv_start = (float)(1.0 / (double)opts->ray_start); p.inv_end = (float)(1.0 / (double)opts->ray_end);
        } else if (opts->auto_bounds) {
            N3D_CHECK(opts->ray_bounds_ws && opts->box_side > 0.f, "render_rays: 'auto' ray bounds need the [N*R*R*2] scratch and the box side length");
            p.depth_mode = 2; p.ray_bounds = opts->ray_bounds_ws;
            hipLaunchKernelGGL(render_depth_bounds_init_kernel, dim3(1), dim3(1), 0, stream, keys);
            hipLaunchKernelGGL(ray_limits_box_kernel, dim3(bgrid), dim3(256), 0, stream, p, opts->ray_bounds_ws, opts->box_side * 0.5f, keys);
            hipLaunchKernelGGL(ray_limits_fixup_kernel, dim3(bgrid), dim3(256), 0, stream, opts->ray_bounds_ws, nrays, (const unsigned int*)keys);
            N3D_LAUNCH_CHECK();
        }
    }
    const int M = Sc + Sf;
    // waves per workgroup: as many as the LDS holds beside the shared decoder image (at most one per SIMD)
    const size_t per_wave = (size_t)RPW * ray_lds_floats(M) * sizeof(float), image = (size_t)RN_WROWS * 64 * sizeof(float);
    int wpb = (int)((160 * 1024 - image) / per_wave);
    wpb = wpb > 4 ? 4 : wpb;
    N3D_CHECK(wpb >= 1, "render_rays: %d samples per ray do not fit the LDS", M);
    const size_t lds = image + wpb * per_wave;
    // the gather: eight lanes per texel (render_rays_c8_kernel) where four waves share a CU's texture path — there the address / tag rate of 64 scattered 16-byte
    // accesses per load instruction is what a pass waits for (510 -> 453 us on the benchmark shape); with two or three waves per CU (more than 96 samples per ray:
    // the colour rows fill the LDS) round 5's 64-bytes-per-lane gather is the faster one (1.76 against 1.85 ms at 96 + 96): profiles/r06_render_gather_pmc.txt
    const bool coalesced = n3d_tune("N3D_RENDER_GATHER", wpb >= 4 ? 1 : 0) != 0 && (int64_t)3 * PH * PW * RN_C * 4 < (1ll << 31);      // (32-bit texel offsets per sample)
    const bool split = n3d_tune("N3D_RENDER_SPLIT", opts && opts->decoder_split_bf16 ? 1 : 0) != 0;
    const void* kfn = split ? (coalesced ? (const void*)render_rays_c8s_kernel : (const void*)render_rays_s_kernel)
                            : (coalesced ? (const void*)render_rays_c8_kernel : (const void*)render_rays_kernel);
    const double pts = (double)N * R * R * M;
    N3dProfScope prof(N3D_K_RENDER_RAYS, stream, pts * 2.0 * (RN_C * RN_HID + RN_HID * (RN_C + 1)),
                      pts * 12.0 * RN_C * 4.0 + 4.0 * N * R * R * (RN_C + 1));          // bytes: 12 texels x 128 B gathered per point + the outputs
    if (lds > 48 * 1024)
        N3D_CHECK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess, "render_rays: %zu bytes of LDS refused", lds);
    hipLaunchKernelGGL(render_depth_bounds_init_kernel, dim3(1), dim3(1), 0, stream, keys);
    N3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(render_depth_bounds_kernel, dim3(bgrid), dim3(256), 0, stream, p, nrays, keys);
    N3D_LAUNCH_CHECK();
    const dim3 rgrid(cdiv((R * R + 1) / 2, wpb), N), rblock(64 * wpb);
    if (split) {
        if (coalesced) hipLaunchKernelGGL(render_rays_c8s_kernel, rgrid, rblock, lds, stream, p);
        else hipLaunchKernelGGL(render_rays_s_kernel, rgrid, rblock, lds, stream, p);
    } else {
        if (coalesced) hipLaunchKernelGGL(render_rays_c8_kernel, rgrid, rblock, lds, stream, p);
        else hipLaunchKernelGGL(render_rays_kernel, rgrid, rblock, lds, stream, p);
    }
    N3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int n3d_render_rays(const float* planes_cl, const float* cam2world, const float* intrinsics, const float* tlin,
                               const float* jitter, const float* u, const float* w1, const float* b1, const float* w2,
                               const float* b2, float* feat, float* depth, float* wsum, float* bounds_ws, int N, int R, int Sc,
                               int Sf, int PH, int PW, float depth_delta, float coord_scale, n3d_stream_t stream_) {
    return n3d_render_rays_ex(planes_cl, cam2world, intrinsics, tlin, jitter, u, w1, b1, w2, b2, feat, depth, wsum, bounds_ws, N, R, Sc, Sf, PH, PW,
                              depth_delta, coord_scale, nullptr, stream_);
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
