// 1x1 convolution (toRGB, fromRGB, fusion layers) on the bf16 matrix cores with split-bf16 operands, for gfx950.
//
// A 1x1 conv is the GEMM  y[o, px] = sum_i w[o,i] * s[n,i] * x[n,i,px]  with a tiny M (3..512 output channels) and
// N = H*W pixels: it is HBM-bound (read x once, write y once) as long as the contraction itself keeps up.  The fp32 MFMA
// (v_mfma_f32_32x32x2_f32, 157 TFLOP/s) does NOT keep up for O >= 64 — 128 output channels cost 2048 matrix cycles per
// 2 KB of activations = 4 B/clk/CU, a third of what HBM delivers — so the contraction runs as three bf16 MFMAs on hi/lo
// operand halves (see conv2d_bf16x3.hip for the arithmetic): 384 cycles per 2 KB, 21 B/clk/CU.
//
// No halo, so the activations never touch LDS: every wave owns 32 pixels and gathers its own MFMA B fragments straight
// from HBM (lane = pixel, 8 consecutive channels per lane -> 8 dword loads, each a coalesced 128-byte row segment per
// half-wave), multiplies by the style, splits to bf16 hi/lo in registers and feeds the matrix pipe.  Only the weights go
// through LDS: the pre-split K-major tiles of n3d_conv2d_prep_weight_bf16x3 (ksize 1) are copied 32 channels at a time,
// double-buffered, and read back as conflict-free ds_read_b128 A fragments shared by the 8 waves of the workgroup.
// Workgroup = 512 threads = 8 waves = 256 pixels x (32 MT) output channels; the loads of K-step s+1 are issued before
// the MFMAs of step s and consumed after them.
// The epilogue is the common fused one (row scale, noise, bias, activation, clamp, residual) and can upsample a
// half-resolution residual on the fly (n3d_epilogue.residual_up_filter): the skip-image update
// img = upsample2d(img) + toRGB(x) of SynthesisBlock.forward (tat/networks_stylegan2.py:580-584) in one pass.
// Replaces the same reference call sites as n3d_conv2d with ksize 1 (conv2d_resample.py:96-136 inside ToRGBLayer,
// tat/networks_stylegan2.py:353-357, and Conv2dLayer :173-183).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct C1Params {
    const float* x; const bf16x8* wt16; const float* style; float* y;
    int N, I, O, OP64, H, W, HW;
    int tiles_p, tiles_m;
    bf16x8* side; const float* side_style; int64_t side_style_stride;    // n3d_conv2d_desc.side_split8 (NULL = none)
    int y_split8;                // y is the split8 layout (bf16 [N][2][O/8][HW][8]) of the following 3x3 layer instead of float32 NCHW
    int64_t xbs, ybs, style_stride, yrs;
    int64_t wbs;                 // 16-byte units between consecutive samples' weight tiles (0 = shared by the batch)
    n3d_epilogue epi;
};

__device__ __noinline__ float conv1_act_generic(float v, int act, float alpha) { return n3d_act(v, act, alpha); }

// RES (MT = 1, I <= 256: the toRGB layers of the large resolutions): the weights of ALL K steps are staged once (<= 32 KB) and the K
// loop runs without barriers — with a handful of output channels the per-step barrier, not the arithmetic, paced the activation
// loads (3.6 TB/s read-only on the 512 x 512 x 128-channel toRGB).
template <int MT, bool RES, bool SIDE = false>
__global__ __launch_bounds__(512, 4) void conv1x1_bf16x3_kernel(C1Params p) {
    constexpr int BM = 32 * MT, NTHR = 512;
    constexpr int KC = MT <= 2 ? 2 : 1;                                   // 16-channel chunks per K step (register budget: 128 VGPRs)
    constexpr int A_ITEMS = KC * 4 * BM;                                  // 16-byte slots per K step: [kc][hl 2][half 2][row BM]
    constexpr int A_PER_T = (A_ITEMS + NTHR - 1) / NTHR;
    __shared__ bf16x8 A_s[(RES ? 8 : 2) * A_ITEMS];
    __shared__ float s_style[1024];
    __shared__ float s_side[SIDE ? 1024 : 1];                            // the side output's styles (the next block's conv0)
    __shared__ float s_rs[BM], s_bs[BM];                                  // per-channel epilogue factors (no dependent global loads in the store loop)

    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    int lb;                                                               // XCD-aware logical block id, M-tile fastest (conv2d_bf16x3.hip)
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int mt_i = lb % p.tiles_m; lb /= p.tiles_m;
    const int tp = lb % p.tiles_p, n = lb / p.tiles_p;
    const int m0 = mt_i * BM;
    const int px = tp * 256 + wn * 32 + l31;
    const bool px_ok = px < p.HW;
    const int nsteps = p.I / (16 * KC);

    for (int i = tid; i < p.I; i += NTHR) s_style[i] = p.style ? p.style[(int64_t)n * p.style_stride + i] : 1.f;
    if (SIDE)
        for (int i = tid; i < p.I; i += NTHR) s_side[i] = p.side_style[(int64_t)n * p.side_style_stride + i];
    if (tid < BM) {
        const n3d_epilogue& E = p.epi;
        const int o = min(m0 + tid, p.O - 1);
        s_rs[tid] = E.const_scale * (E.row_scale ? E.row_scale[(int64_t)n * (E.row_scale_stride ? E.row_scale_stride : p.O) + o] : 1.f);
        s_bs[tid] = E.bias ? E.bias[o] : 0.f;
    }

    // weights: thread's j-th slot e = tid + 512 j -> (kc, hl, half, row)
    const bf16x8* a_src[A_PER_T];
    bool a_ok[A_PER_T];
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
        const int e = tid + j * NTHR;
        const int row = e % BM, hf = (e / BM) & 1, hl = (e / (2 * BM)) & 1, kc = e / (4 * BM);
        a_ok[j] = e < A_ITEMS && m0 + row < p.OP64;
        a_src[j] = p.wt16 + (int64_t)n * p.wbs + (a_ok[j] ? ((int64_t)(kc * 2 + hl) * 2 + hf) * p.OP64 + m0 + row : 0);
    }
    const int64_t a_step = (int64_t)KC * 4 * p.OP64;                      // KC chunks x 4 slabs of OP64 slots per K step
    // buffer loads: one descriptor per sample (SGPRs), 32-bit per-lane byte offset, per-channel offset in an SGPR — the 16
    // loads of a step need no per-lane 64-bit addresses (flat loads cost 2 VGPRs each and pushed the kernel into scratch)
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.xbs), 0, p.I * p.HW * 4, 0x00020000);
    const int x_voff = ((px_ok ? px : 0) + half * 8 * p.HW) * 4;

    bf16x8 ra[A_PER_T];
    float raw[KC][8];
    auto load_step = [&](int s) {
        if (!RES) {
#pragma unroll
            for (int j = 0; j < A_PER_T; ++j)
                if (a_ok[j]) ra[j] = a_src[j][s * a_step];
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                raw[kc][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, x_voff, ((s * KC + kc) * 16 + c) * p.HW * 4, 0));
    };
    auto store_a = [&](int s) {
        bf16x8* dst = A_s + (s & 1) * A_ITEMS;
#pragma unroll
        for (int j = 0; j < A_PER_T; ++j) {
            const int e = tid + j * NTHR;
            if (e < A_ITEMS) {
                bf16x8 z;
#pragma unroll
                for (int c = 0; c < 8; ++c) z[c] = (__bf16)0.f;
                dst[e] = a_ok[j] ? ra[j] : z;
            }
        }
    };

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    if (RES) {                                                            // every step's weight slab, once
        for (int st = 0; st < nsteps; ++st) {
#pragma unroll
            for (int j = 0; j < A_PER_T; ++j) {
                const int e = tid + j * NTHR;
                if (e < A_ITEMS) {
                    bf16x8 z;
#pragma unroll
                    for (int c = 0; c < 8; ++c) z[c] = (__bf16)0.f;
                    A_s[st * A_ITEMS + e] = a_ok[j] ? a_src[j][st * a_step] : z;
                }
            }
        }
    }
    load_step(0);
    if (!RES) store_a(0);
    __syncthreads();                                                      // s_style + A(0) (RES: every A) visible

    for (int s = 0; s < nsteps; ++s) {
        bf16x8 bh[KC], bl[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const float* st = s_style + s * (16 * KC) + kc * 16 + half * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float v = raw[kc][c] * st[c];
                const __bf16 h = (__bf16)v;
                bh[kc][c] = h;
                bl[kc][c] = (__bf16)(v - (float)h);
            }
        }
        if (SIDE && px_ok) {
            // second reader of x (the next block's transposed convolution): x * its styles, hi / lo split, one 16-byte unit
            // per lane — channels (chunk, half) x 8 of this lane's pixel, the unit n3d_split8_from_nchw would write
            const int64_t C8HW = (int64_t)(p.I >> 3) * p.HW;
            bf16x8* so = p.side + (int64_t)n * 2 * C8HW + px;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const float* st2 = s_side + s * (16 * KC) + kc * 16 + half * 8;
                bf16x8 sh, sl;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float v = raw[kc][c] * st2[c];
                    const __bf16 h = (__bf16)v;
                    sh[c] = h;
                    sl[c] = (__bf16)(v - (float)h);
                }
                const int64_t u = (int64_t)((s * KC + kc) * 2 + half) * p.HW;
                so[u] = sh;
                so[C8HW + u] = sl;
            }
        }
        if (s + 1 < nsteps) load_step(s + 1);                             // issue only: consumed after the MFMA block
        const bf16x8* A = A_s + (RES ? s : (s & 1)) * A_ITEMS + half * BM + l31;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bf16x8 ah = A[(kc * 4 + 0) * BM + mt * 32], al = A[(kc * 4 + 2) * BM + mt * 32];
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[kc], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[kc], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[kc], acc[mt], 0, 0, 0);
            }
        if (!RES) {
            if (s + 1 < nsteps) store_a(s + 1);
            __syncthreads();
        }
    }

    // epilogue (C/D layout: col = lane&31 = pixel, row = (r&3) + 8*(r>>2) + 4*(lane>>5) = channel)
    if (!px_ok) return;
    const n3d_epilogue& E = p.epi;
    const int oy = px / p.W, ox = px % p.W;
    const float nz = E.noise ? E.noise[px] * E.noise_strength[0] : 0.f;
    const bool lrelu = E.act == N3D_ACT_LRELU, linear = E.act == N3D_ACT_LINEAR;
    const int64_t yplane = (int64_t)p.H * p.yrs;
    float* dst = p.y + (int64_t)n * p.ybs + (int64_t)oy * p.yrs + ox;
    const bool res_up = E.residual && E.residual_up_filter;
    const float* res = E.residual ? E.residual + (int64_t)n * E.residual_batch_stride + (res_up ? 0 : px) : nullptr;
    const int64_t lplane = (int64_t)(p.H >> 1) * (p.W >> 1);
    n3d_up2_taps up2;
    if (res_up) up2 = n3d_up2_setup(E.residual_up_filter, oy, ox, p.H >> 1, p.W >> 1);
    if (p.y_split8) {
        // single-consumer 1x1 layers in front of a pre-split 3x3 layer (the encoders' fromrgb, networks_stylegan2_styleunet.py:
        // fromrgb -> + skip -> conv1): the hi / lo bf16 pair goes out directly — lanes 0-31 / 32-63 hold channels 8g..8g+3 /
        // 8g+4..8g+7 of a pixel, i.e. the two 8-byte halves of one 16-byte split8 unit (a wave store = 512 contiguous bytes).
        // Same operation order as the generic epilogue below followed by n3d_split8_from_nchw (scale 1) => identical bits.
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        const int64_t O8 = p.O >> 3;
        __bf16* yb = (__bf16*)p.y + (int64_t)n * 2 * O8 * p.HW * 8;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 hi, lo;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ol = mt * 32 + 8 * g + 4 * half + k;
                    float v = acc[mt][4 * g + k] * s_rs[ol] + nz + s_bs[ol];
                    if (lrelu) v = v > 0.f ? v : v * E.alpha;
                    else if (!linear) v = conv1_act_generic(v, E.act, E.alpha);
                    v *= E.gain;
                    if (E.clamp >= 0.f) v = fminf(fmaxf(v, -E.clamp), E.clamp);
                    v = n3d_round16(v, E.round_f16);
                    if (res_up) v += n3d_up2_apply(up2, res + (int64_t)(m0 + ol) * lplane);
                    else if (res) v += res[(int64_t)(m0 + ol) * p.HW];
                    const __bf16 h = (__bf16)v;
                    hi[k] = h;
                    lo[k] = (__bf16)(v - (float)h);
                }
                __bf16* u = yb + (((int64_t)((m0 >> 3) + mt * 4 + g)) * p.HW + px) * 8 + 4 * half;
                *(bf16x4*)u = hi;
                *(bf16x4*)(u + O8 * p.HW * 8) = lo;
            }
        return;
    }
    if ((linear || (lrelu && E.alpha >= 0.f && E.alpha <= 1.f)) && !E.residual && m0 + BM <= p.O) {
        // the common epilogue as straight-line code: leaky ReLU = max(v, alpha v) (linear: alpha 1), no clamp = clamp at +inf
        const float alpha_eff = lrelu ? E.alpha : 1.f, clamp_eff = E.clamp >= 0.f ? E.clamp : INFINITY;
        float* d0 = dst + (int64_t)(m0 + 4 * half) * yplane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ol = mt * 32 + (r & 3) + 8 * (r >> 2);
                float v = acc[mt][r] * s_rs[ol + 4 * half] + nz + s_bs[ol + 4 * half];
                v = fmaxf(v, v * alpha_eff) * E.gain;
                d0[(int64_t)ol * yplane] = n3d_round16(fminf(fmaxf(v, -clamp_eff), clamp_eff), E.round_f16);
            }
        return;
    }
    if (res_up && linear && m0 + BM <= p.O) {
        // toRGB with the fused skip-image upsample (every toRGB but the first of a network), straight-line: bias, clamp at +inf
        // when absent, 4 taps of the half-resolution image
        const float clamp_eff = E.clamp >= 0.f ? E.clamp : INFINITY;
        float* d0 = dst + (int64_t)(m0 + 4 * half) * yplane;
        const float* r0 = res + (int64_t)(m0 + 4 * half) * lplane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ol = mt * 32 + (r & 3) + 8 * (r >> 2);
                float v = (acc[mt][r] * s_rs[ol + 4 * half] + nz + s_bs[ol + 4 * half]) * E.gain;
                v = n3d_round16(fminf(fmaxf(v, -clamp_eff), clamp_eff), E.round_f16);
                d0[(int64_t)ol * yplane] = v + n3d_up2_apply(up2, r0 + (int64_t)ol * lplane);
            }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (o >= p.O) continue;
            float v = acc[mt][r] * s_rs[o - m0] + nz + s_bs[o - m0];
            if (lrelu) v = v > 0.f ? v : v * E.alpha;
            else if (!linear) v = conv1_act_generic(v, E.act, E.alpha);
            v *= E.gain;
            if (E.clamp >= 0.f) v = fminf(fmaxf(v, -E.clamp), E.clamp);
            v = n3d_round16(v, E.round_f16);
            if (res_up) v += n3d_up2_apply(up2, res + (int64_t)o * lplane);
            else if (res) v += res[(int64_t)o * p.HW];
            dst[(int64_t)o * yplane] = v;
        }
}

// ---- few-pixel layers (<= 64 x 64): the K loop above is a serial chain of I/32 load -> MFMA steps, ~2 us each, on a
// handful of workgroups.  Here the 8 waves of a workgroup split the INPUT CHANNELS of one 32-pixel tile instead: every wave
// issues all loads of its I/8 slice at once (activations by buffer loads, its weight fragments straight from the L2-resident
// prepared tiles — no LDS staging, no barrier in the loop), and the eight partial accumulators are summed through LDS in a
// fixed order; wave w then applies the epilogue to rows r = w (mod 8) of the 32 MT x 32 tile.
template <int MT, bool SIDE = false>
__global__ __launch_bounds__(512, 2) void conv1x1_bf16x3_ksplit_kernel(C1Params p) {
    constexpr int BM = 32 * MT;
    constexpr int GS = MT <= 2 ? 4 : 2;                                   // 16-channel chunks in flight per wave
    __shared__ float red[8 * MT * 16 * 64];

    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    int lb = blockIdx.x;
    const int mt_i = lb % p.tiles_m; lb /= p.tiles_m;
    const int tp = lb % p.tiles_p, n = lb / p.tiles_p;
    const int m0 = mt_i * BM;
    const int px = tp * 32 + l31;
    const bool px_ok = px < p.HW;
    const int nchunk = p.I / 128;                                         // chunks per wave (I % 128 == 0)
    const int c_begin = wn * nchunk;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.xbs), 0, p.I * p.HW * 4, 0x00020000);
    const int x_voff = ((px_ok ? px : 0) + half * 8 * p.HW) * 4;
    const float* sty = p.style ? p.style + (int64_t)n * p.style_stride : nullptr;
    const float* sty2 = SIDE ? p.side_style + (int64_t)n * p.side_style_stride : nullptr;     // side output: see conv1x1_bf16x3_kernel
    const int64_t C8HW = (int64_t)(p.I >> 3) * p.HW;
    bf16x8* so = SIDE ? p.side + (int64_t)n * 2 * C8HW + px : nullptr;

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    for (int g0 = 0; g0 < nchunk; g0 += GS) {
        float raw[GS][8];
        bf16x8 ah[GS][MT], al[GS][MT];
#pragma unroll
        for (int g = 0; g < GS; ++g) {
            if (g0 + g >= nchunk) continue;
            const int c = c_begin + g0 + g;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch)
                raw[g][ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, x_voff, (c * 16 + ch) * p.HW * 4, 0));
            const bf16x8* a = p.wt16 + (int64_t)n * p.wbs + ((int64_t)c * 4 + half) * p.OP64 + m0 + l31;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool ok = m0 + mt * 32 + l31 < p.OP64;
                ah[g][mt] = a[ok ? mt * 32 : 0];
                al[g][mt] = a[(ok ? mt * 32 : 0) + 2 * p.OP64];
            }
        }
#pragma unroll
        for (int g = 0; g < GS; ++g) {
            if (g0 + g >= nchunk) continue;
            const int c = c_begin + g0 + g;
            bf16x8 bh, bl;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                const float v = raw[g][ch] * (sty ? sty[c * 16 + half * 8 + ch] : 1.f);
                const __bf16 h = (__bf16)v;
                bh[ch] = h;
                bl[ch] = (__bf16)(v - (float)h);
            }
            if (SIDE && px_ok) {
                bf16x8 sh, sl;
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) {
                    const float v = raw[g][ch] * sty2[c * 16 + half * 8 + ch];
                    const __bf16 h = (__bf16)v;
                    sh[ch] = h;
                    sl[ch] = (__bf16)(v - (float)h);
                }
                const int64_t u = (int64_t)(c * 2 + half) * p.HW;
                so[u] = sh;
                so[C8HW + u] = sl;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[g][mt], bh, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g][mt], bl, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g][mt], bh, acc[mt], 0, 0, 0);
            }
        }
    }
    // rows of weights beyond OP64 were read from row 0: they only feed output channels >= O, which are never stored

#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wn * MT + mt) * 16 + r) * 64 + lane] = acc[mt][r];
    __syncthreads();
    if (!px_ok) return;

    const n3d_epilogue& E = p.epi;
    const int oy = px / p.W, ox = px % p.W;
    const float nz = E.noise ? E.noise[px] * E.noise_strength[0] : 0.f;
    const int64_t yplane = (int64_t)p.H * p.yrs;
    float* dst = p.y + (int64_t)n * p.ybs + (int64_t)oy * p.yrs + ox;
    const bool res_up = E.residual && E.residual_up_filter;
    const float* res = E.residual ? E.residual + (int64_t)n * E.residual_batch_stride + (res_up ? 0 : px) : nullptr;
    const int64_t lplane = (int64_t)(p.H >> 1) * (p.W >> 1);
    n3d_up2_taps up2;
    if (res_up) up2 = n3d_up2_setup(E.residual_up_filter, oy, ox, p.H >> 1, p.W >> 1);
    const float* rsp = E.row_scale ? E.row_scale + (int64_t)n * (E.row_scale_stride ? E.row_scale_stride : p.O) : nullptr;
#pragma unroll
    for (int q = 0; q < MT * 2; ++q) {                                    // this wave's rows: item = q*8 + wn -> (mt, r)
        const int item = q * 8 + wn, mt = item >> 4, r = item & 15;
        const int o = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (o >= p.O) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[((w * MT + mt) * 16 + r) * 64 + lane];
        v = v * (E.const_scale * (rsp ? rsp[o] : 1.f)) + nz + (E.bias ? E.bias[o] : 0.f);
        v = conv1_act_generic(v, E.act, E.alpha) * E.gain;
        if (E.clamp >= 0.f) v = fminf(fmaxf(v, -E.clamp), E.clamp);
        v = n3d_round16(v, E.round_f16);
        if (res_up) v += n3d_up2_apply(up2, res + (int64_t)o * lplane);
        else if (res) v += res[(int64_t)o * p.HW];
        dst[(int64_t)o * yplane] = v;
    }
}

// called by n3d_conv2d_bf16x3 for ksize == 1 (descriptor already validated for the common fields)
int conv1x1_bf16x3_launch(const n3d_conv2d_desc* d, hipStream_t stream) {
    N3D_CHECK(d->mode == 0, "conv2d_bf16x3: 1x1 supports stride 1 only");
    N3D_CHECK(d->I % 32 == 0 && d->I <= 1024, "conv2d_bf16x3: 1x1 needs I %% 32 == 0 and I <= 1024 (got %d)", d->I);
    N3D_CHECK((int64_t)d->I * d->H * d->W * 4 < (1ll << 31), "conv2d_bf16x3: 1x1 sample larger than 2 GiB");
    C1Params p;
    p.x = d->x; p.wt16 = (const bf16x8*)d->wt; p.style = d->style; p.y = d->y;
    p.N = d->N; p.I = d->I; p.O = d->O; p.OP64 = (d->O + 63) / 64 * 64; p.H = d->H; p.W = d->W; p.HW = d->H * d->W;
    p.xbs = d->x_batch_stride; p.ybs = d->y_batch_stride; p.epi = d->epi;
    p.style_stride = d->style_stride ? d->style_stride : d->I;
    N3D_CHECK((d->wt_batch_stride & 15) == 0, "conv2d_bf16x3: wt_batch_stride must be a multiple of 16 bytes");
    p.wbs = d->wt_batch_stride / 16;
    p.yrs = d->y_row_stride ? d->y_row_stride : d->W;
    N3D_CHECK(p.yrs >= d->W, "conv2d_bf16x3: y_row_stride smaller than the output width");
    p.side = (bf16x8*)d->side_split8; p.side_style = d->side_style;
    p.side_style_stride = d->side_style_stride ? d->side_style_stride : d->I;
    N3D_CHECK(!d->side_split8 || (d->side_style && d->O <= 128 && d->y_layout == N3D_LAYOUT_NCHW_F32 && ((uintptr_t)d->side_split8 & 15) == 0),
              "conv2d_bf16x3: side_split8 needs side_style, O <= 128 (one channel tile), a float32 NCHW y and a 16-byte aligned buffer");
    p.y_split8 = d->y_layout == N3D_LAYOUT_SPLIT8;
    N3D_CHECK(!p.y_split8 || (d->O % 32 == 0 && ((uintptr_t)d->y & 15) == 0), "conv2d_bf16x3: a split8 output of the 1x1 kernel needs O %% 32 == 0 and a 16-byte aligned y");
    N3D_CHECK(!d->epi.residual_up_filter || (d->epi.residual && d->H % 2 == 0 && d->W % 2 == 0),
              "conv2d_bf16x3: residual_up_filter needs a residual and an even output size");
    const double flops = 2.0 * d->N * (double)d->O * d->I * p.HW;
    const double bytes = 4.0 * ((double)d->N * d->I * p.HW + (double)d->N * d->O * p.HW + (double)d->O * d->I);
    if (p.HW <= 4096 && d->I % 128 == 0 && !p.y_split8) {          // few pixels, long K: waves split the input channels (see above)
        int mt = cdiv(d->O, 32);
        if (mt > 4) mt = 4;
        p.tiles_p = cdiv(p.HW, 32);
        p.tiles_m = cdiv(d->O, 32 * mt);
        N3dProfScope prof(N3D_K_CONV1X1_BF16X3, stream, flops, bytes);
        const dim3 grid((unsigned)(p.tiles_p * p.tiles_m * d->N));
        if (p.side) switch (mt) {                                          // O <= 128 -> tiles_m == 1: x is read once
            case 1: hipLaunchKernelGGL((conv1x1_bf16x3_ksplit_kernel<1, true>), grid, dim3(512), 0, stream, p); break;
            case 2: hipLaunchKernelGGL((conv1x1_bf16x3_ksplit_kernel<2, true>), grid, dim3(512), 0, stream, p); break;
            case 3: hipLaunchKernelGGL((conv1x1_bf16x3_ksplit_kernel<3, true>), grid, dim3(512), 0, stream, p); break;
            default: hipLaunchKernelGGL((conv1x1_bf16x3_ksplit_kernel<4, true>), grid, dim3(512), 0, stream, p); break;
        } else switch (mt) {
            case 1: hipLaunchKernelGGL(conv1x1_bf16x3_ksplit_kernel<1>, grid, dim3(512), 0, stream, p); break;
            case 2: hipLaunchKernelGGL(conv1x1_bf16x3_ksplit_kernel<2>, grid, dim3(512), 0, stream, p); break;
            case 3: hipLaunchKernelGGL(conv1x1_bf16x3_ksplit_kernel<3>, grid, dim3(512), 0, stream, p); break;
            default: hipLaunchKernelGGL(conv1x1_bf16x3_ksplit_kernel<4>, grid, dim3(512), 0, stream, p); break;
        }
        N3D_LAUNCH_CHECK();
        return 0;
    }
    p.tiles_p = cdiv(p.HW, 256);
    // output-channel tile: everything in one workgroup up to 128 channels (x is then read exactly once); few-pixel layers
    // take 32-channel tiles instead so that more than a handful of CUs work on them (x re-reads hit L2)
    int mt = cdiv(d->O, 32);
    if (mt > 4) mt = 4;
    if ((int64_t)p.tiles_p * d->N * cdiv(d->O, 32 * mt) < 128 && !p.side) mt = 1;          // (side output: ONE channel tile, x is read once)
    if (p.y_split8 && d->O % (32 * mt) != 0) mt = 1;                       // the split8 epilogue writes whole tiles
    p.tiles_m = cdiv(d->O, 32 * mt);
    const int64_t nblk = (int64_t)p.tiles_p * p.tiles_m * d->N;
    N3D_CHECK(nblk < (1ll << 31), "conv2d_bf16x3: grid too large");
    N3dProfScope prof(N3D_K_CONV1X1_BF16X3, stream, flops, bytes);
    const dim3 grid((unsigned)nblk);
    switch (mt) {
        case 1:
            if (p.side) {                                                    // one channel tile (O <= 128): every workgroup sees all channels of its pixels once
                if (d->I <= 256 && d->I % 32 == 0) hipLaunchKernelGGL((conv1x1_bf16x3_kernel<1, true, true>), grid, dim3(512), 0, stream, p);
                else hipLaunchKernelGGL((conv1x1_bf16x3_kernel<1, false, true>), grid, dim3(512), 0, stream, p);
            } else if (d->I <= 256 && d->I % 32 == 0) hipLaunchKernelGGL((conv1x1_bf16x3_kernel<1, true>), grid, dim3(512), 0, stream, p);
            else hipLaunchKernelGGL((conv1x1_bf16x3_kernel<1, false>), grid, dim3(512), 0, stream, p);
            break;
        case 2:
            if (p.side) hipLaunchKernelGGL((conv1x1_bf16x3_kernel<2, false, true>), grid, dim3(512), 0, stream, p);
            else hipLaunchKernelGGL((conv1x1_bf16x3_kernel<2, false>), grid, dim3(512), 0, stream, p);
            break;
        case 3:
            // (all 8 K steps' weights resident in LDS, the MT = 1 kernels' barrier-free loop, was measured for the 96-channel tri-plane
            // toRGB: 65.2 against 63.6 us — profiles/r03_c1_bench.txt — so the staged loop stays)
            if (p.side) hipLaunchKernelGGL((conv1x1_bf16x3_kernel<3, false, true>), grid, dim3(512), 0, stream, p);
            else hipLaunchKernelGGL((conv1x1_bf16x3_kernel<3, false>), grid, dim3(512), 0, stream, p);
            break;
        default:
            if (p.side) hipLaunchKernelGGL((conv1x1_bf16x3_kernel<4, false, true>), grid, dim3(512), 0, stream, p);
            else hipLaunchKernelGGL((conv1x1_bf16x3_kernel<4, false>), grid, dim3(512), 0, stream, p);
            break;
    }
    N3D_LAUNCH_CHECK();
    return 0;
}
