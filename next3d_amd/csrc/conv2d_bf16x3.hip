// conv2d on the bf16 matrix cores with fp32-class accuracy ("bf16x3" operand splitting) for gfx950.
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate, 1/16 of bf16 MFMA.  Every fp32 operand is therefore
// split into two bf16 halves, x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits together), and
//     a*b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (the dropped a_lo*b_lo term is <= 2^-16 relative)
// is evaluated with THREE v_mfma_f32_32x32x16_bf16 per K=16 chunk, accumulating in fp32: 96 MFMA cycles per
// 32x32x16 block instead of 512 with the fp32 instruction — a 5.3x higher ceiling (833 "fp32-equivalent" TFLOP/s).
// Each bf16 x bf16 product is exact in fp32, so the only error is the operand truncation: measured 8e-5 max-abs on the
// final 512x512 RGB when EVERY convolution of the generator is emulated this way on the CPU (tolerance: 1e-3).
//
// Structure: same implicit GEMM as conv2d.hip (K-major weights, input patch with halo staged once per stage and reused
// by all 9 taps, style modulation applied while staging, next stage prefetched into registers during the MFMA block),
// re-laid-out for K=16 fragments:
//   weights are split and tiled ONCE at model load (n3d_conv2d_prep_weight_bf16x3):
//       wt16[tap][I/16][hl][half][OP64][8]   hl: 0 = hi, 1 = lo;  half: channels 0-7 / 8-15 of the chunk
//     so a workgroup's (tap, hl, half) slab is 64 rows x 16 bytes, copied verbatim into LDS and read back as
//     conflict-free ds_read_b128 fragments (lane = output channel row, 8 consecutive k per lane);
//   activations stay fp32 NCHW in HBM; while staging, each work item gathers 8 channels of one patch pixel, applies the
//     style, splits and packs them into two 16-byte LDS slots  B_{hi,lo}[half][pixel][8]  (lane = pixel, 8 consecutive k).
// Workgroup = 256 threads = 4 waves; tile = 64 output channels x (8 x 32) pixels; each wave 64 x 64 (2x2 accumulators).
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Conv16Params {
    const float* x; const bf16x8* wt16; const float* style; float* y; float* partial;
    int N, I, O, OP64, H, W, OH, OW;
    int tiles_x, tiles_y, tiles_m, ksplit, ic_per_split;
    int tw, th;   // transposed kernel: tile = th rows x tw columns of input-grid positions, flattened over the waves' lanes
    int y_c8;     // transposed kernel only: write y channel-interleaved, [N][O/8][OH][yrs pixels][8] float32 (N3D_LAYOUT_C8_F32)
    int dbg;      // N3D_CONV_DBG ablation bits (tuning only): 1 skip stores, 2 skip MFMA, 4 skip stage loads, 8 skip LDS fragment reads
    int64_t xbs, ybs, style_stride, yrs;      // yrs: output row pitch in floats
    int64_t wbs;                              // 16-byte units between consecutive samples' weight tiles (0 = shared by the batch)
    n3d_epilogue epi;
};

int conv1x1_bf16x3_launch(const n3d_conv2d_desc* d, hipStream_t stream);      // conv1x1_bf16x3.hip
int conv2d_s2_bf16x3_launch(const n3d_conv2d_desc* d, hipStream_t stream);    // conv2d_s2_bf16x3.hip
int conv2d_ps_bf16x3_launch(const n3d_conv2d_desc* d, hipStream_t stream);     // conv2d_ps_bf16x3.hip (split8 input)
int conv2d_sk_bf16x3_try_launch(const n3d_conv2d_desc* d, hipStream_t stream);  // conv2d_sk_bf16x3.hip (few-pixel layers; 1 = not its layer)
extern "C" int n3d_conv2d_sk_eligible(int N, int I, int O, int H, int W);
int conv2d_up_sk_bf16x3_try_launch(const n3d_conv2d_desc* d, hipStream_t stream);   // conv2d_sk_bf16x3.hip (few-position transposed layers; 1 = not its layer)
int conv2d_up_ps_bf16x3_launch(const n3d_conv2d_desc* d, hipStream_t stream);  // conv2d_ps_bf16x3.hip (split8 input, transposed, c8 output)
int conv2d_s2_ps_bf16x3_launch(const n3d_conv2d_desc* d, hipStream_t stream);  // conv2d_ps_bf16x3.hip (split8 input, stride 2)
int conv2d_p_bf16x3_try_launch(const n3d_conv2d_desc* d, int tiles_x, int tiles_y, hipStream_t stream, int* launched);   // conv2d_p_bf16x3.hip

__device__ __noinline__ float conv16_act_generic(float v, int act, float alpha) { return n3d_act(v, act, alpha); }

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v[i];
        hi[i] = h;
        lo[i] = (__bf16)(v[i] - (float)h);
    }
}

// 3x3, stride 1, padding 1.  NW waves per workgroup, each owning 2 pixel rows: tile = 64 channels x (2 NW x 32) pixels.
// NW = 8 (512 threads, one workgroup per CU) halves the weight-slab traffic and the per-thread staging work per MFMA.
template <int NW, bool FLAT>
__device__ __forceinline__ void conv2d_bf16x3_body(const Conv16Params& p, const int bid, const int nwg) {
    constexpr int NT_ = 64 * NW;                                          // threads
    constexpr int BM = 64, TH = 2 * NW, TW = 32, ICB = 16, TAPS = 9;
    constexpr int PH = TH + 2, PW_C = TW + 2, PPIX = PH * PW_C;           // NW=4: 10 x 34 = 340 patch pixels
    // FLAT (images narrower than 32): the tile is th x tw = p.th x p.tw pixels flattened row-major over the 64 NW MFMA
    // columns (position q -> row q / tw, col q % tw), so a 16x16 or 8x8 image still fills the 32-wide N tiles; the patch
    // is (th+2) x (tw+2) <= PPIX (host plan) with a run-time row pitch.
    const int PW = FLAT ? p.tw + 2 : PW_C;
    constexpr int A_ITEMS = TAPS * 2 * 2 * BM;                            // 16-byte slots: [tap][hl][half][row]
    constexpr int B_ITEMS = 2 * PPIX;                                     // (half, pixel) work items
    constexpr int TAP_STEP = NT_ / 256;                                   // thread's j-th A item = tap (tid/256 + TAP_STEP*j)
    constexpr int A_PER_T = (TAPS + TAP_STEP - 1) / TAP_STEP;             // 9 (NW=4) / 5 (NW=8)
    constexpr int B_PER_T = (B_ITEMS + NT_ - 1) / NT_;

    // NW == 8: LDS is double-buffered (2 x 76 KB) and the eight waves run a ping-pong schedule — waves 0-3 convert + store
    // stage s+1 and issue the loads of stage s+2 BEFORE their MFMA block, waves 4-7 AFTER it — so on every SIMD (waves w and
    // w+4 share one) the bf16 matrix pipe is fed by one wave while the other does the VALU/LDS staging work; one barrier per
    // stage.  NW == 4 (small images): single buffer, two barriers per stage, two workgroups per CU.
    constexpr int NBUF = (NW == 8) ? 2 : 1;
    constexpr int A_SZ = TAPS * 2 * BM, B_SZ = 2 * PPIX;
    __shared__ bf16x8 A_hi[NBUF * A_SZ], A_lo[NBUF * A_SZ];               // [buf][tap][half][row]
    __shared__ bf16x8 B_hi[NBUF * B_SZ], B_lo[NBUF * B_SZ];               // [buf][half][pixel]
    __shared__ float s_style[1024];

    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    // 1-D grid, XCD-aware: hardware block b runs on XCD b % 8 — remap so that each XCD owns a contiguous range of LOGICAL
    // ids, ordered M-tile fastest: the O/64 workgroups that read the same input patch run back-to-back on ONE XCD and share
    // it through that XCD's L2 instead of fetching it O/64 times from HBM / Infinity Cache.
    int lb;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt_i = lb % p.tiles_m; lb /= p.tiles_m;
    const int tile_i = lb % (p.tiles_x * p.tiles_y); lb /= (p.tiles_x * p.tiles_y);
    const int tx = tile_i % p.tiles_x, ty = tile_i / p.tiles_x;
    const int m0 = mt_i * BM;
    const int ks = lb % p.ksplit, n = lb / p.ksplit;
    const int y0 = FLAT ? ty * p.th : ty * TH, x0 = FLAT ? 0 : tx * TW;
    const int prows = FLAT ? p.th + 2 : PH;
    const int ic_begin = ks * p.ic_per_split;
    const int ic_end = min(p.I, ic_begin + p.ic_per_split);
    const int nstage = (ic_end - ic_begin) / ICB;
    const int KC = p.I / ICB;
    const int HW = p.H * p.W;

    for (int i = tid; i < ic_end - ic_begin; i += NT_) s_style[i] = p.style ? p.style[(int64_t)n * p.style_stride + ic_begin + i] : 1.f;

    // A staging: thread owns (row, half, hl) for all 9 taps
    const int a_row = tid & 63, a_q = (tid >> 6) & 3, a_half = a_q & 1, a_hl = a_q >> 1, a_t0 = tid >> 8;
    const int64_t a_tap_stride = (int64_t)KC * 4 * p.OP64, a_stage_stride = (int64_t)4 * p.OP64;
    const bf16x8* a_src = p.wt16 + (int64_t)n * p.wbs + ((int64_t)(ic_begin / ICB) * 4 + a_hl * 2 + a_half) * p.OP64 + m0 + a_row + a_t0 * a_tap_stride;
    bf16x8* a_dst = (a_hl ? A_lo : A_hi) + a_half * BM + a_row + a_t0 * 2 * BM;
    // B staging: work item e = tid + 256 j -> (half, patch pixel)
    int b_goff[B_PER_T];
    bool b_ok[B_PER_T];
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j) {
        const int e = tid + j * NT_;
        const int hf = e / PPIX, pp = e % PPIX;
        const int pr = pp / PW;
        const int iy = y0 - 1 + pr, ix = x0 - 1 + pp % PW;
        b_ok[j] = e < B_ITEMS && pr < prows && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        // byte offset inside the sample; pixels outside the image get an offset beyond the buffer: the load returns 0
        b_goff[j] = b_ok[j] ? (hf * 8 * HW + iy * p.W + ix) * 4 : (int)0x80000000;
    }
    // buffer loads: one descriptor per sample (SGPRs) + a 32-bit lane offset + the channel offset in an SGPR — no 64-bit
    // per-lane address arithmetic, and the zero padding of the halo comes from the hardware range check
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.xbs), 0, p.I * HW * 4, 0x00020000);

    bf16x8 ra[A_PER_T];
    float rb[B_PER_T][8];
    auto load_stage = [&](int st) {                       // issue only; consumed in store_stage (after the MFMA block)
        const bf16x8* as = a_src + st * a_stage_stride;
#pragma unroll
        for (int j = 0; j < A_PER_T; ++j)
            if (a_t0 + TAP_STEP * j < TAPS) ra[j] = as[j * TAP_STEP * a_tap_stride];
#pragma unroll
        for (int j = 0; j < B_PER_T; ++j)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                rb[j][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, b_goff[j], (ic_begin + st * ICB + c) * HW * 4, 0));
    };
    auto store_stage = [&](int st) {
        const int bo_a = (NBUF == 2 && (st & 1)) ? A_SZ : 0, bo_b = (NBUF == 2 && (st & 1)) ? B_SZ : 0;
#pragma unroll
        for (int j = 0; j < A_PER_T; ++j)
            if (a_t0 + TAP_STEP * j < TAPS) a_dst[bo_a + j * TAP_STEP * 2 * BM] = ra[j];
#pragma unroll
        for (int j = 0; j < B_PER_T; ++j) {
            const int e = tid + j * NT_;
            if (e >= B_ITEMS) continue;
            const int hf = e / PPIX;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = rb[j][c] * s_style[st * ICB + hf * 8 + c];
            bf16x8 hi, lo;
            split8(v, hi, lo);
            B_hi[bo_b + e] = hi;
            B_lo[bo_b + e] = lo;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    __syncthreads();
    if (nstage > 0) { load_stage(0); store_stage(0); }
    __syncthreads();
    const int a_frag = half * BM + l31;                                   // + tap*2*BM + mt*32
    int q_row[2], q_col[2];                                               // this lane's pixel in each of its two N tiles
    bool q_act[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        if (FLAT) {
            const int q = (wn * 2 + nt) * 32 + l31;
            q_act[nt] = q < p.th * p.tw;
            q_row[nt] = q_act[nt] ? q / p.tw : 0; q_col[nt] = q_act[nt] ? q % p.tw : 0;
        } else { q_act[nt] = true; q_row[nt] = wn * 2 + nt; q_col[nt] = l31; }
    }
    const int b_frag0 = half * PPIX + q_row[0] * PW + q_col[0];           // + ky*PW + kx
    const int b_frag1 = half * PPIX + q_row[1] * PW + q_col[1];
    auto mfma_block = [&](int st) {
        const int bo_a = (NBUF == 2 && (st & 1)) ? A_SZ : 0, bo_b = (NBUF == 2 && (st & 1)) ? B_SZ : 0;
        if (!(p.dbg & 16)) __builtin_amdgcn_s_setprio(1);      // role-split schedule: favour the wave that is feeding the matrix pipe
        // fragments of tap t+1 are fetched (second register set) before the MFMAs of tap t are issued: the LDS latency
        // hides behind 12 MFMAs instead of draining the matrix pipe at every tap
        bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
        auto fetch = [&](int t, int s) {
            const int boff = (t / 3) * PW + (t % 3);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) { ah[s][mt] = A_hi[bo_a + t * 2 * BM + a_frag + mt * 32]; al[s][mt] = A_lo[bo_a + t * 2 * BM + a_frag + mt * 32]; }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) { bh[s][nt] = B_hi[bo_b + (nt ? b_frag1 : b_frag0) + boff]; bl[s][nt] = B_lo[bo_b + (nt ? b_frag1 : b_frag0) + boff]; }
        };
        fetch(0, 0);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int s = t & 1;
            if (t + 1 < TAPS) fetch(t + 1, s ^ 1);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s][mt], bh[s][nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][mt], bl[s][nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][mt], bh[s][nt], acc[mt][nt], 0, 0, 0);
                }
        }
        if (!(p.dbg & 16)) __builtin_amdgcn_s_setprio(0);
    };
    if (NBUF == 2) {
        // registers hold the raw data of stage st+1 at the top of iteration st
        if (nstage > 1) load_stage(1);
        const bool stage_first = wn < NW / 2;
        for (int st = 0; st < nstage; ++st) {
            if (stage_first) {
                if (st + 1 < nstage && !(p.dbg & 8)) store_stage(st + 1);
                if (st + 2 < nstage && !(p.dbg & 4)) load_stage(st + 2);
                if (!(p.dbg & 2)) mfma_block(st);
            } else {
                if (!(p.dbg & 2)) mfma_block(st);
                if (st + 1 < nstage && !(p.dbg & 8)) store_stage(st + 1);
                if (st + 2 < nstage && !(p.dbg & 4)) load_stage(st + 2);
            }
            __syncthreads();
        }
    } else {
        for (int st = 0; st < nstage; ++st) {
            if (st + 1 < nstage && !(p.dbg & 4)) load_stage(st + 1);
            if (!(p.dbg & 2)) mfma_block(st);
            __syncthreads();
            if (st + 1 < nstage) { store_stage(st + 1); __syncthreads(); }
        }
    }

    // epilogue (C/D layout: col = lane&31 = pixel, row = (r&3) + 8*(r>>2) + 4*(lane>>5) = channel)
    if (p.dbg & 1) { if (acc[0][0][0] == 123.456f) p.y[0] = 1.f; return; }
    const n3d_epilogue& E = p.epi;
    const int64_t plane = (int64_t)p.OH * p.OW;
    float rs[2][16], bs[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int oc = o < p.O ? o : p.O - 1;
            rs[mt][r] = E.const_scale * (E.row_scale ? E.row_scale[(int64_t)n * (E.row_scale_stride ? E.row_scale_stride : p.O) + oc] : 1.f);
            bs[mt][r] = E.bias ? E.bias[oc] : 0.f;
        }
    const float nstr = E.noise ? E.noise_strength[0] : 0.f;
    const bool lrelu = E.act == N3D_ACT_LRELU, linear = E.act == N3D_ACT_LINEAR;
    const bool simple = !p.partial && (linear || (lrelu && E.alpha >= 0.f && E.alpha <= 1.f)) && !E.residual && m0 + BM <= p.O;
    const float alpha_eff = lrelu ? E.alpha : 1.f, clamp_eff = E.clamp >= 0.f ? E.clamp : INFINITY;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int oy = y0 + q_row[nt], ox = x0 + q_col[nt];
        if (!q_act[nt] || oy >= p.OH || ox >= p.OW) continue;
        const int64_t po = (int64_t)oy * p.OW + ox;
        if (p.partial) {
            float* dst = p.partial + ((int64_t)ks * p.N + n) * p.O * plane + po;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (o < p.O) dst[(int64_t)o * plane] = acc[mt][nt][r];
                }
            continue;
        }
        const float nz = E.noise ? E.noise[po] * nstr : 0.f;
        const int64_t yplane = (int64_t)p.OH * p.yrs;
        float* dst = p.y + (int64_t)n * p.ybs + (int64_t)oy * p.yrs + ox;
        if (simple) {       // the common layer epilogue as straight-line code: leaky ReLU = max(v, alpha v) (linear: alpha 1),
            float* d0 = dst + (int64_t)(m0 + 4 * half) * yplane;          // no clamp = clamp at +inf, full channel tile, no residual
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[mt][nt][r] * rs[mt][r] + nz + bs[mt][r];
                    v = fmaxf(v, v * alpha_eff) * E.gain;
                    d0[(int64_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * yplane] = fminf(fmaxf(v, -clamp_eff), clamp_eff);
                }
            continue;
        }
        const float* res = E.residual ? E.residual + (int64_t)n * E.residual_batch_stride + po : nullptr;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (o >= p.O) continue;
                float v = acc[mt][nt][r] * rs[mt][r] + nz + bs[mt][r];
                if (lrelu) v = v > 0.f ? v : v * E.alpha;
                else if (!linear) v = conv16_act_generic(v, E.act, E.alpha);
                v *= E.gain;
                if (E.clamp >= 0.f) v = fminf(fmaxf(v, -E.clamp), E.clamp);
                if (res) v += res[(int64_t)o * plane];
                dst[(int64_t)o * yplane] = v;
            }
    }
}


template <int NW, bool FLAT>
__global__ __launch_bounds__(64 * NW, 2) void conv2d_bf16x3_kernel(Conv16Params p) { conv2d_bf16x3_body<NW, FLAT>(p, blockIdx.x, gridDim.x); }

// Transposed 3x3 stride-2 (the up-sampling layers) on the split-bf16 path: all four output phases from one staged patch,
// exactly as conv2d_up_mfma_kernel in conv2d.hip (tap (ky,kx) feeds phase (ky==1, kx==1) from patch offset
// (ky==2 ? 0 : 1, kx==2 ? 0 : 1)).  Workgroup = 64 output channels x (th x tw <= 32*NW) input-grid positions x 4 phases.  The
// MFMA N dimension is 32 independent positions, so the tile's positions are FLATTENED row-major over the waves' lanes
// (position q = wave*32 + lane -> row q / tw, col q % tw): the grid is (H+1) x (W+1), and a 33-wide tile covers W = 32 in
// one piece (a fixed 32-wide tile would need a second, 97 %-empty tile for the last column).  acc[2 row-tiles][4 phases].  Per K=16 chunk: 8 B-fragment reads (4 offsets x hi/lo) are
// shared by all 9 taps, 36 A-fragment reads, 54 MFMAs.
template <int NW>
__device__ __forceinline__ void conv2d_up_bf16x3_body(const Conv16Params& p, const int bid, const int nwg) {
    constexpr int NT_ = 64 * NW;
    constexpr int BM = 64, ICB = 16, TAPS = 9;
    constexpr int PPIX = (NW + 1) * 33;                                   // patch capacity: (th+1) x (tw+1) <= PPIX (host plan)
    constexpr int TAP_STEP = NT_ / 256;
    constexpr int A_PER_T = (TAPS + TAP_STEP - 1) / TAP_STEP;
    constexpr int B_ITEMS = 2 * PPIX;
    constexpr int B_PER_T = (B_ITEMS + NT_ - 1) / NT_;

    constexpr int NBUF = (NW == 8) ? 2 : 1;                               // NW == 8: double buffer + ping-pong (see above)
    constexpr int A_SZ = TAPS * 2 * BM, B_SZ = 2 * PPIX;
    __shared__ bf16x8 A_hi[NBUF * A_SZ], A_lo[NBUF * A_SZ];
    __shared__ bf16x8 B_hi[NBUF * B_SZ], B_lo[NBUF * B_SZ];
    __shared__ float s_style[1024];

    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    // 1-D grid, XCD-aware: hardware block b runs on XCD b % 8 — remap so that each XCD owns a contiguous range of LOGICAL
    // ids, ordered M-tile fastest: the O/64 workgroups that read the same input patch run back-to-back on ONE XCD and share
    // it through that XCD's L2 instead of fetching it O/64 times from HBM / Infinity Cache.
    int lb;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt_i = lb % p.tiles_m; lb /= p.tiles_m;
    const int tile_i = lb % (p.tiles_x * p.tiles_y); lb /= (p.tiles_x * p.tiles_y);
    const int tx = tile_i % p.tiles_x, ty = tile_i / p.tiles_x;
    const int m0 = mt_i * BM;
    const int ks = lb % p.ksplit, n = lb / p.ksplit;
    const int y0 = ty * p.th, x0 = tx * p.tw;
    const int PW = p.tw + 1, prows = p.th + 1;
    const int ic_begin = ks * p.ic_per_split;
    const int ic_end = min(p.I, ic_begin + p.ic_per_split);
    const int nstage = (ic_end - ic_begin) / ICB;
    const int KC = p.I / ICB;
    const int HW = p.H * p.W;
    const int GH = p.H + 1, GW = p.W + 1;

    for (int i = tid; i < ic_end - ic_begin; i += NT_) s_style[i] = p.style ? p.style[(int64_t)n * p.style_stride + ic_begin + i] : 1.f;

    const int a_row = tid & 63, a_q = (tid >> 6) & 3, a_half = a_q & 1, a_hl = a_q >> 1, a_t0 = tid >> 8;
    const int64_t a_tap_stride = (int64_t)KC * 4 * p.OP64, a_stage_stride = (int64_t)4 * p.OP64;
    const bf16x8* a_src = p.wt16 + (int64_t)n * p.wbs + ((int64_t)(ic_begin / ICB) * 4 + a_hl * 2 + a_half) * p.OP64 + m0 + a_row + a_t0 * a_tap_stride;
    bf16x8* a_dst = (a_hl ? A_lo : A_hi) + a_half * BM + a_row + a_t0 * 2 * BM;
    int b_goff[B_PER_T];
    bool b_ok[B_PER_T];
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j) {
        const int e = tid + j * NT_;
        const int hf = e / PPIX, pp = e % PPIX;
        const int pr = pp / PW;
        const int iy = y0 - 1 + pr, ix = x0 - 1 + pp % PW;
        b_ok[j] = e < B_ITEMS && pr < prows && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        b_goff[j] = b_ok[j] ? (hf * 8 * HW + iy * p.W + ix) * 4 : (int)0x80000000;     // byte offset; outside the image: beyond the buffer -> 0
    }
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.xbs), 0, p.I * HW * 4, 0x00020000);

    bf16x8 ra[A_PER_T];
    float rb[B_PER_T][8];
    auto load_stage = [&](int st) {
        const bf16x8* as = a_src + st * a_stage_stride;
#pragma unroll
        for (int j = 0; j < A_PER_T; ++j)
            if (a_t0 + TAP_STEP * j < TAPS) ra[j] = as[j * TAP_STEP * a_tap_stride];
#pragma unroll
        for (int j = 0; j < B_PER_T; ++j)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                rb[j][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, b_goff[j], (ic_begin + st * ICB + c) * HW * 4, 0));
    };
    auto store_stage = [&](int st) {
        const int bo_a = (NBUF == 2 && (st & 1)) ? A_SZ : 0, bo_b = (NBUF == 2 && (st & 1)) ? B_SZ : 0;
#pragma unroll
        for (int j = 0; j < A_PER_T; ++j)
            if (a_t0 + TAP_STEP * j < TAPS) a_dst[bo_a + j * TAP_STEP * 2 * BM] = ra[j];
#pragma unroll
        for (int j = 0; j < B_PER_T; ++j) {
            const int e = tid + j * NT_;
            if (e >= B_ITEMS) continue;
            const int hf = e / PPIX;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = rb[j][c] * s_style[st * ICB + hf * 8 + c];
            bf16x8 hi, lo;
            split8(v, hi, lo);
            B_hi[bo_b + e] = hi;
            B_lo[bo_b + e] = lo;
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][ph][r] = 0.f;

    __syncthreads();
    if (nstage > 0) { load_stage(0); store_stage(0); }
    __syncthreads();
    const int a_frag = half * BM + l31;
    const int q_pos = wn * 32 + l31;                                      // flattened tile position of this lane
    const bool q_act = q_pos < p.th * p.tw;
    const int q_row = q_act ? q_pos / p.tw : 0, q_col = q_act ? q_pos % p.tw : 0;
    const int b_frag = half * PPIX + q_row * PW + q_col;                  // + dy*PW + dx
    auto mfma_block = [&](int st) {
        const int bo_a = (NBUF == 2 && (st & 1)) ? A_SZ : 0, bo_b = (NBUF == 2 && (st & 1)) ? B_SZ : 0;
        if (!(p.dbg & 16)) __builtin_amdgcn_s_setprio(1);
        bf16x8 bh[4], bl[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) { bh[d] = B_hi[bo_b + b_frag + (d >> 1) * PW + (d & 1)]; bl[d] = B_lo[bo_b + b_frag + (d >> 1) * PW + (d & 1)]; }
        // A fragments of tap t+1 are fetched into a second register set before the MFMAs of tap t are issued
        bf16x8 ah[2][2], al[2][2];
        auto fetch_a = [&](int t, int s) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) { ah[s][mt] = A_hi[bo_a + t * 2 * BM + a_frag + mt * 32]; al[s][mt] = A_lo[bo_a + t * 2 * BM + a_frag + mt * 32]; }
        };
        fetch_a(0, 0);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int ky = t / 3, kx = t % 3, s = t & 1;
            const int ph = (ky == 1 ? 2 : 0) + (kx == 1 ? 1 : 0);
            const int d = (ky == 2 ? 0 : 2) + (kx == 2 ? 0 : 1);
            if (t + 1 < TAPS) fetch_a(t + 1, s ^ 1);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[mt][ph] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s][mt], bh[d], acc[mt][ph], 0, 0, 0);
                acc[mt][ph] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][mt], bl[d], acc[mt][ph], 0, 0, 0);
                acc[mt][ph] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][mt], bh[d], acc[mt][ph], 0, 0, 0);
            }
        }
        if (!(p.dbg & 16)) __builtin_amdgcn_s_setprio(0);
    };
    if (NBUF == 2) {
        if (nstage > 1) load_stage(1);
        const bool stage_first = wn < NW / 2;
        for (int st = 0; st < nstage; ++st) {
            if (stage_first) {
                if (st + 1 < nstage && !(p.dbg & 8)) store_stage(st + 1);
                if (st + 2 < nstage && !(p.dbg & 4)) load_stage(st + 2);
                if (!(p.dbg & 2)) mfma_block(st);
            } else {
                if (!(p.dbg & 2)) mfma_block(st);
                if (st + 1 < nstage && !(p.dbg & 8)) store_stage(st + 1);
                if (st + 2 < nstage && !(p.dbg & 4)) load_stage(st + 2);
            }
            __syncthreads();
        }
    } else {
        for (int st = 0; st < nstage; ++st) {
            if (st + 1 < nstage) load_stage(st + 1);
            mfma_block(st);
            __syncthreads();
            if (st + 1 < nstage) { store_stage(st + 1); __syncthreads(); }
        }
    }

    // epilogue: phases (a,0),(a,1) of one position are adjacent output pixels -> one 8-byte store per lane
    if (p.dbg & 1) { if (acc[0][0][0] == 123.456f) p.y[0] = 1.f; return; }
    const n3d_epilogue& E = p.epi;
    const int64_t plane = (int64_t)p.OH * p.OW;
    const float nstr = E.noise ? E.noise_strength[0] : 0.f;
    const bool lrelu = E.act == N3D_ACT_LRELU, linear = E.act == N3D_ACT_LINEAR;
    struct __attribute__((packed, aligned(4))) pair_t { float v0, v1; };
    // per-channel factors through LDS (the K loop is over: s_style is free): with 128 accumulators live there are no
    // registers to hoist their global loads out of the store loop, which made it a chain of dependent load -> store steps
    float* s_rs = s_style, *s_bs = s_style + BM;
    if (tid < BM) {
        const int o = min(m0 + tid, p.O - 1);
        s_rs[tid] = E.const_scale * (E.row_scale ? E.row_scale[(int64_t)n * (E.row_scale_stride ? E.row_scale_stride : p.O) + o] : 1.f);
        s_bs[tid] = E.bias ? E.bias[o] : 0.f;
    }
    __syncthreads();
    const int gy = y0 + q_row, gx = x0 + q_col;
    if (!q_act || gy >= GH || gx >= GW) return;
    if (p.y_c8) {
        // channel-interleaved output for the FIR that follows (fir4_c8_split8_kernel): one 32-byte unit = 8 consecutive channels
        // of one pixel.  This lane holds, per 32-row tile mt and group g, the 4 consecutive channels 8g + 4*half + (0..3) — half a
        // unit — for each of its 4 output pixels: 32 16-byte stores per lane (the NCHW form needs 64 8-byte stores).
        // (demodulation only, like `simple` below: the launch guarantees it)
        float* yb = p.y + (int64_t)n * p.ybs;
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
            const int oy = 2 * gy + pa;
            if (oy >= p.OH) continue;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int ox = 2 * gx + pb;
                if (ox >= p.OW) continue;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c8 = (m0 >> 3) + mt * 4 + g, ol = mt * 32 + 8 * g + 4 * half;
                        const f32x16& a = acc[mt][pa * 2 + pb];
                        const int rh = E.round_f16;
                        const f32x4 v = {n3d_round16(a[4 * g + 0] * s_rs[ol + 0], rh), n3d_round16(a[4 * g + 1] * s_rs[ol + 1], rh),
                                         n3d_round16(a[4 * g + 2] * s_rs[ol + 2], rh), n3d_round16(a[4 * g + 3] * s_rs[ol + 3], rh)};
                        *reinterpret_cast<f32x4*>(yb + (((int64_t)c8 * p.OH + oy) * p.yrs + ox) * 8 + 4 * half) = v;
                    }
            }
        }
        return;
    }
    const bool simple = !p.partial && linear && !E.noise && !E.bias && !E.residual && E.clamp < 0.f && E.gain == 1.f && m0 + BM <= p.O;
#pragma unroll
    for (int pa = 0; pa < 2; ++pa) {
        const int oy = 2 * gy + pa, ox = 2 * gx;
        if (oy >= p.OH) continue;
        const bool two = ox + 1 < p.OW;
        const int64_t po = (int64_t)oy * p.OW + ox;
        if (p.partial) {
            float* dst = p.partial + ((int64_t)ks * p.N + n) * p.O * plane + po;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (o >= p.O) continue;
                    if (two) *reinterpret_cast<pair_t*>(dst + (int64_t)o * plane) = pair_t{acc[mt][pa * 2][r], acc[mt][pa * 2 + 1][r]};
                    else dst[(int64_t)o * plane] = acc[mt][pa * 2][r];
                }
            continue;
        }
        const int64_t yplane = (int64_t)p.OH * p.yrs;
        float* dst = p.y + (int64_t)n * p.ybs + (int64_t)oy * p.yrs + ox;
        if (simple) {               // the up-sampling layers: y = acc * demodulation, everything else happens in the FIR pass
            float* d0 = dst + (int64_t)(m0 + 4 * half) * yplane;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ol = mt * 32 + (r & 3) + 8 * (r >> 2);
                    const float rs = s_rs[ol + 4 * half];
                    if (two) *reinterpret_cast<pair_t*>(d0 + (int64_t)ol * yplane) = pair_t{acc[mt][pa * 2][r] * rs, acc[mt][pa * 2 + 1][r] * rs};
                    else d0[(int64_t)ol * yplane] = acc[mt][pa * 2][r] * rs;
                }
            continue;
        }
        const float nz0 = E.noise ? E.noise[po] * nstr : 0.f;
        const float nz1 = (E.noise && two) ? E.noise[po + 1] * nstr : 0.f;
        const float* res = E.residual ? E.residual + (int64_t)n * E.residual_batch_stride + po : nullptr;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (o >= p.O) continue;
                const float rs = s_rs[o - m0], bs = s_bs[o - m0];
                float v[2] = {acc[mt][pa * 2][r] * rs + nz0 + bs, acc[mt][pa * 2 + 1][r] * rs + nz1 + bs};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (lrelu) v[q] = v[q] > 0.f ? v[q] : v[q] * E.alpha;
                    else if (!linear) v[q] = conv16_act_generic(v[q], E.act, E.alpha);
                    v[q] *= E.gain;
                    if (E.clamp >= 0.f) v[q] = fminf(fmaxf(v[q], -E.clamp), E.clamp);
                    if (res && (q == 0 || two)) v[q] += res[(int64_t)o * plane + q];
                }
                if (two) *reinterpret_cast<pair_t*>(dst + (int64_t)o * yplane) = pair_t{v[0], v[1]};
                else dst[(int64_t)o * yplane] = v[0];
            }
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void conv2d_up_bf16x3_kernel(Conv16Params p) { conv2d_up_bf16x3_body<NW>(p, blockIdx.x, gridDim.x); }

// split-K second pass: sum the partial tiles and apply the epilogue.  VEC: 4 consecutive pixels of one row per thread
// (OW % 4 == 0, aligned pointers): 16-byte loads/stores and one index decomposition per 4 outputs.
struct SplitkArgs { const float* partial; float* y; int ksplit, N, O, OH, OW; int64_t ybs, yrs; n3d_epilogue epi; };
template <bool VEC>
__device__ __forceinline__ void conv16_splitk_epilogue_body(const SplitkArgs& a, const int bid, const int nwg) {
    const float* __restrict__ partial = a.partial; float* __restrict__ y = a.y;
    const int ksplit = a.ksplit, N = a.N, O = a.O, OH = a.OH, OW = a.OW;
    const int64_t ybs = a.ybs, yrs = a.yrs;
    const n3d_epilogue& epi = a.epi;
    const int64_t plane = (int64_t)OH * OW, total = (int64_t)N * O * plane;
    constexpr int V = VEC ? 4 : 1;
    for (int64_t i = ((int64_t)bid * blockDim.x + threadIdx.x) * V; i < total; i += (int64_t)nwg * blockDim.x * V) {
        float v[V];
#pragma unroll
        for (int q = 0; q < V; ++q) v[q] = 0.f;
        for (int k = 0; k < ksplit; ++k) {
            if (VEC) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(partial + (int64_t)k * total + i);
#pragma unroll
                for (int q = 0; q < V; ++q) v[q] += t[q];
            } else {
                v[0] += partial[(int64_t)k * total + i];
            }
        }
        const int64_t pl = i / plane;
        const int pix = (int)(i % plane), n = (int)(pl / O), o = (int)(pl % O);
        const int oy = pix / OW, ox = pix % OW;
#pragma unroll
        for (int q = 0; q < V; ++q) v[q] = n3d_apply_epilogue(v[q], epi, n, o, O, oy, ox + q, OH, OW);
        float* dst = y + (int64_t)n * ybs + ((int64_t)o * OH + oy) * yrs + ox;
        if (VEC) *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
        else dst[0] = v[0];
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void conv16_splitk_epilogue_kernel(SplitkArgs a) { conv16_splitk_epilogue_body<VEC>(a, blockIdx.x, gridDim.x); }

// w [O,I,k,k] fp32 -> wt16[tap][I/16][hl][half][OP64][8] bf16 (hi / lo split, zero padded rows)
__global__ __launch_bounds__(256) void conv16_prep_weight_kernel(const float* __restrict__ w, __bf16* __restrict__ wt16, int O, int I, int KK,
                                                                 int OP64) {
    const int64_t total = (int64_t)KK * I * OP64;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(e % OP64);
        const int i = (int)((e / OP64) % I);
        const int t = (int)(e / ((int64_t)OP64 * I));
        const float v = o < O ? w[((int64_t)o * I + i) * KK + t] : 0.f;
        const __bf16 hi = (__bf16)v;
        const __bf16 lo = (__bf16)(v - (float)hi);
        const int kc = i / 16, hf = (i % 16) / 8, c = i % 8;
        const int64_t base = ((((int64_t)t * (I / 16) + kc) * 2) * 2 + hf) * OP64 + o;       // hl = 0
        wt16[base * 8 + c] = hi;
        wt16[(base + (int64_t)2 * OP64) * 8 + c] = lo;                                        // hl = 1
    }
}

extern "C" int n3d_conv2d_prep_weight_bf16x3(const float* w, void* wt16, int O, int I, int ksize, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(w && wt16 && O > 0 && I > 0 && (ksize == 3 || ksize == 1), "conv2d_prep_weight_bf16x3: bad arguments (3x3 / 1x1 only)");
    N3D_CHECK(I % 16 == 0, "conv2d_prep_weight_bf16x3: input channels must be a multiple of 16");
    const int OP64 = (O + 63) / 64 * 64;
    const int64_t total = (int64_t)ksize * ksize * I * OP64;
    const int grid = (int)(cdiv64(total, 256) > 8192 ? 8192 : cdiv64(total, 256));
    N3dProfScope prof(N3D_K_MISC, stream, 0.0, 8.0 * total);
    hipLaunchKernelGGL(conv16_prep_weight_kernel, dim3(grid), dim3(256), 0, stream, w, (__bf16*)wt16, O, I, ksize * ksize, OP64);
    N3D_LAUNCH_CHECK();
    return 0;
}

// Tile plan shared by the launch and by n3d_conv2d_bf16x3_blocks (the host's split-K heuristic).  Stride-1: 8-wave
// workgroups when the grid still covers the chip with them, else 4-wave.  Transposed mode: balanced (th x tw) tiles of at most 32*NW
// flattened positions, tw <= 33, th <= 32 (patch (th+1) x (tw+1) <= (NW+1)*33 entries of LDS).
void conv16_up_tiles(int gh, int gw, int nw, int* tiles_x, int* tiles_y, int* tw, int* th) {
    if (gw > 66) {                     // wide images: 32-aligned tile columns (the 33rd-column remainder is a few percent)
        *tw = 32; *th = nw;
        *tiles_x = cdiv(gw, 32); *tiles_y = cdiv(gh, nw);
        return;
    }
    *tiles_x = cdiv(gw, 33);
    *tw = cdiv(gw, *tiles_x);
    int th_max = (32 * nw) / *tw;
    if (th_max > 32) th_max = 32;
    while ((th_max + 1) * (*tw + 1) > (nw + 1) * 33) --th_max;
    *tiles_y = cdiv(gh, th_max);
    *th = cdiv(gh, *tiles_y);
}

static void conv16_plan(int N, int O, int H, int W, int mode, bool* big, int* tiles_x, int* tiles_y, int* tw, int* th) {
    const int tiles_m = cdiv(O, 64);
    if (mode == 2) {                   // always the 8-wave kernel (measured 1.5x the 4-wave one even at 0.6 workgroups per CU)
        conv16_up_tiles(H + 1, W + 1, 8, tiles_x, tiles_y, tw, th);
        *big = true;
        return;
    }
    if (W < 32) {                      // narrow image: flattened 4-wave tiles (256 positions), th rows of the full width
        int th_max = 256 / W;
        if (th_max > 32) th_max = 32;
        while ((th_max + 2) * (W + 2) > 340) --th_max;
        *big = false; *tw = W; *tiles_x = 1;
        *tiles_y = cdiv(H, th_max); *th = cdiv(H, *tiles_y);
        return;
    }
    const int64_t blocks8 = (int64_t)cdiv(W, 32) * cdiv(H, 16) * tiles_m * N;
    *big = H >= 16 && blocks8 >= 256;
    *tw = 32; *th = *big ? 16 : 8;
    *tiles_x = cdiv(W, 32); *tiles_y = cdiv(H, *th);
}

static bool splitk_vec(const float* partial, const float* y, int OW, int64_t ybs, int64_t yrs) {
    return (OW & 3) == 0 && ((ybs | yrs) & 3) == 0 && (((uintptr_t)partial | (uintptr_t)y) & 15) == 0;
}
static int splitk_grid(int64_t total, bool vec) {
    const int64_t items = vec ? total / 4 : total;
    return (int)(cdiv64(items, 256) > 4096 ? 4096 : cdiv64(items, 256));
}
int conv16_splitk_epilogue_launch(const float* partial, float* y, int ksplit, int N, int O, int OH, int OW, int64_t ybs, int64_t yrs,
                                  const n3d_epilogue& epi, hipStream_t stream) {
    const bool vec = splitk_vec(partial, y, OW, ybs, yrs);
    const int grid = splitk_grid((int64_t)N * O * OH * OW, vec);
    const SplitkArgs a = {partial, y, ksplit, N, O, OH, OW, ybs, yrs, epi};
    if (vec) hipLaunchKernelGGL(conv16_splitk_epilogue_kernel<true>, dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(conv16_splitk_epilogue_kernel<false>, dim3(grid), dim3(256), 0, stream, a);
    N3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int n3d_conv2d_bf16x3_blocks(int N, int O, int H, int W, int mode) {
    if (mode == 1) {
        const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
        const int64_t b = (int64_t)cdiv(OW, 32) * cdiv(OH, 16) * cdiv(O, 64) * N;
        return b > 0x7fffffff ? 0x7fffffff : (int)b;
    }
    bool big; int tx, ty, tw, th;
    conv16_plan(N, O, H, W, mode, &big, &tx, &ty, &tw, &th);
    const int64_t b = (int64_t)tx * ty * cdiv(O, 64) * N;
    return b > 0x7fffffff ? 0x7fffffff : (int)b;
}

// Parameters, tile plan and kernel kind of the register-staged 3x3 kernels (stride 1 / transposed) for one descriptor.
enum { C16_UP8 = 0, C16_BIG8 = 1, C16_FLAT4 = 2, C16_ROW4 = 3 };
static int conv16_setup(const n3d_conv2d_desc* d, Conv16Params& p, int& kind, int64_t& nblk) {
    const bool up = d->mode == 2;
    p.x = d->x; p.wt16 = (const bf16x8*)d->wt; p.style = d->style; p.y = d->y; p.partial = d->workspace;
    p.dbg = n3d_tune("N3D_CONV_DBG", 0);
    p.y_c8 = d->y_layout == N3D_LAYOUT_C8_F32;
    p.N = d->N; p.I = d->I; p.O = d->O; p.OP64 = (d->O + 63) / 64 * 64; p.H = d->H; p.W = d->W;
    p.OH = up ? 2 * d->H + 1 : d->H; p.OW = up ? 2 * d->W + 1 : d->W;
    p.xbs = d->x_batch_stride; p.ybs = d->y_batch_stride; p.epi = d->epi;
    N3D_CHECK((d->wt_batch_stride & 15) == 0, "conv2d_bf16x3: wt_batch_stride must be a multiple of 16 bytes");
    p.wbs = d->wt_batch_stride / 16;
    p.style_stride = d->style_stride ? d->style_stride : d->I;
    p.yrs = d->y_row_stride ? d->y_row_stride : p.OW;
    N3D_CHECK(p.yrs >= p.OW, "conv2d_bf16x3: y_row_stride smaller than the output width");
    N3D_CHECK(!d->epi.residual_up_filter, "conv2d_bf16x3: residual_up_filter is only supported by the 1x1 kernel");
    N3D_CHECK((int64_t)d->I * d->H * d->W * 4 < (1ll << 31), "conv2d_bf16x3: one sample's input exceeds 2 GiB (32-bit buffer offsets)");
    bool big;
    conv16_plan(d->N, d->O, d->H, d->W, d->mode, &big, &p.tiles_x, &p.tiles_y, &p.tw, &p.th);
    const int max_split = d->I / 16;
    p.ksplit = d->ksplit < 1 ? 1 : (d->ksplit > max_split ? max_split : d->ksplit);
    p.ic_per_split = cdiv(cdiv(d->I, p.ksplit), 16) * 16;
    p.ksplit = cdiv(d->I, p.ic_per_split);
    N3D_CHECK(p.ksplit == 1 || d->workspace != nullptr, "conv2d_bf16x3: ksplit > 1 needs a workspace");
    if (p.y_c8) {
        const n3d_epilogue& E = d->epi;
        N3D_CHECK(up && p.ksplit == 1 && d->O % 64 == 0, "conv2d_bf16x3: the channel-interleaved output is written by the un-split transposed kernel with O %% 64 == 0");
        N3D_CHECK(E.act == N3D_ACT_LINEAR && !E.noise && !E.bias && !E.residual && E.clamp < 0.f && E.gain == 1.f,
                  "conv2d_bf16x3: the channel-interleaved output takes the demodulation-only epilogue (the layer epilogue runs behind the FIR)");
        N3D_CHECK(((uintptr_t)d->y & 15) == 0 && (d->y_batch_stride & 3) == 0, "conv2d_bf16x3: channel-interleaved output must be 16-byte aligned");
    }
    N3D_CHECK(p.ic_per_split <= 1024, "conv2d_bf16x3: more than 1024 input channels per K-split");
    if (p.ksplit == 1) p.partial = nullptr;
    p.tiles_m = cdiv(p.O, 64);
    nblk = (int64_t)p.tiles_x * p.tiles_y * p.tiles_m * p.N * p.ksplit;
    N3D_CHECK(nblk < (1ll << 31), "conv2d_bf16x3: grid too large");
    kind = up ? C16_UP8 : (big ? C16_BIG8 : (d->W < 32 ? C16_FLAT4 : C16_ROW4));
    return 0;
}

extern "C" int n3d_conv2d_bf16x3(const n3d_conv2d_desc* d, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(d != nullptr, "conv2d_bf16x3: null descriptor");
    N3D_CHECK((d->ksize == 3 && d->mode >= 0 && d->mode <= 2) || (d->ksize == 1 && d->mode == 0),
              "conv2d_bf16x3: 3x3 (stride 1, stride 2, transposed stride 2) or 1x1 stride-1 only");
    N3D_CHECK(d->N >= 0 && d->I > 0 && d->O > 0 && d->H > 0 && d->W > 0 && d->I % 16 == 0, "conv2d_bf16x3: bad shape (I %% 16 == 0)");
    N3D_CHECK(d->epi.act >= N3D_ACT_LINEAR && d->epi.act <= N3D_ACT_SWISH, "conv2d_bf16x3: unknown activation %d", d->epi.act);
    N3D_CHECK(d->epi.noise == nullptr || d->epi.noise_strength != nullptr, "conv2d_bf16x3: noise without noise_strength");
    if (d->N == 0) return 0;
    N3D_CHECK(!d->rgb_partial || (d->x_layout == N3D_LAYOUT_SPLIT8 && d->ksize == 3 && d->mode == 0), "conv2d_bf16x3: the fused toRGB (rgb_partial) is an option of the 3x3 stride-1 kernel on split8 input");
    N3D_CHECK(d->x && d->wt && (d->y || d->rgb_partial), "conv2d_bf16x3: null tensor");
    N3D_CHECK(((uintptr_t)d->wt & 15) == 0, "conv2d_bf16x3: wt must be 16-byte aligned");
    N3D_CHECK(d->x_row_stride == 0 || d->x_row_stride == d->W || (d->ksize == 3 && d->mode == 1), "conv2d_bf16x3: only the stride-2 kernel takes a pitched input");
    N3D_CHECK(d->x_layout == N3D_LAYOUT_NCHW_F32 || d->x_layout == N3D_LAYOUT_SPLIT8, "conv2d_bf16x3: unknown x_layout %d", d->x_layout);
    N3D_CHECK(d->y_layout == N3D_LAYOUT_NCHW_F32 || (d->y_layout == N3D_LAYOUT_C8_F32 && d->ksize == 3 && d->mode == 2) ||
              (d->y_layout == N3D_LAYOUT_SPLIT8 && d->ksize == 1 && d->x_layout == N3D_LAYOUT_NCHW_F32),
              "conv2d_bf16x3: y_layout %d is not available for this kernel", d->y_layout);
    N3D_CHECK(d->x_layout != N3D_LAYOUT_SPLIT8 || d->ksize == 3, "conv2d_bf16x3: split8 input goes to the 3x3 kernels");
    N3D_CHECK(!d->side_split8 || (d->ksize == 1 && d->x_layout == N3D_LAYOUT_NCHW_F32) || d->rgb_partial, "conv2d_bf16x3: side_split8 is written by the 1x1 kernel, or by the 3x3 kernel with the fused toRGB");
    N3D_CHECK(d->wt_batch_stride == 0 || d->x_layout != N3D_LAYOUT_SPLIT8 || d->mode == 0, "conv2d_bf16x3: per-sample weights with a split8 input: stride-1 kernel only");
    if (d->x_layout == N3D_LAYOUT_SPLIT8)
        return d->mode == 2 ? conv2d_up_ps_bf16x3_launch(d, stream) : (d->mode == 1 ? conv2d_s2_ps_bf16x3_launch(d, stream) : conv2d_ps_bf16x3_launch(d, stream));
    if (d->ksize == 1) return conv1x1_bf16x3_launch(d, stream);
    N3D_CHECK(!d->epi.round_f16 || d->y_layout == N3D_LAYOUT_C8_F32, "conv2d_bf16x3: round_f16 is supported by the pre-split path (split8 / c8 layouts) and the 1x1 kernel only");
    N3D_CHECK(d->mode != 1 || d->wt_batch_stride == 0, "conv2d_bf16x3: the stride-2 kernels take one weight tensor for the whole batch (wt_batch_stride 0)");
    if (d->mode == 1) return conv2d_s2_bf16x3_launch(d, stream);
    if (d->mode == 0 && d->wt_batch_stride == 0) {                         // few-pixel layers: K split inside the workgroup, one launch (samples share its weight fragments)
        const int r = conv2d_sk_bf16x3_try_launch(d, stream);
        if (r <= 0) return r;
    }
    if (d->mode == 2 && d->wt_batch_stride == 0 && d->y_layout == N3D_LAYOUT_NCHW_F32) {     // few-position transposed layers: the same, no split-K reduce launch
        const int r = conv2d_up_sk_bf16x3_try_launch(d, stream);
        if (r <= 0) return r;
    }
    Conv16Params p;
    int kind; int64_t nblk;
    if (conv16_setup(d, p, kind, nblk) != 0) return -1;
    const bool up = d->mode == 2;
    const bool big = kind == C16_BIG8;
    const double flops = 2.0 * d->N * (double)d->O * d->I * 9 * (up ? (double)d->H * d->W : (double)p.OH * p.OW);
    const double bytes = 4.0 * ((double)d->N * d->I * d->H * d->W + (double)d->N * d->O * p.OH * p.OW + (double)d->O * d->I * 9);
    N3dProfScope prof(N3D_K_CONV2D_BF16X3, stream, flops, bytes);
    const dim3 grid((unsigned)nblk);
    if (up) hipLaunchKernelGGL(conv2d_up_bf16x3_kernel<8>, grid, dim3(512), 0, stream, p);
    else if (big) {
        int launched = 0;                  // several tiles per CU: the persistent kernel (K loop pipelined across tiles)
        if (p.ksplit == 1 && p.dbg == 0 && p.wbs == 0 && conv2d_p_bf16x3_try_launch(d, p.tiles_x, p.tiles_y, stream, &launched) != 0) return -1;
        if (!launched) hipLaunchKernelGGL((conv2d_bf16x3_kernel<8, false>), grid, dim3(512), 0, stream, p);
    }
    else if (d->W < 32) hipLaunchKernelGGL((conv2d_bf16x3_kernel<4, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv2d_bf16x3_kernel<4, false>), grid, dim3(256), 0, stream, p);
    N3D_LAUNCH_CHECK();
    if (p.ksplit > 1) return conv16_splitk_epilogue_launch(p.partial, p.y, p.ksplit, p.N, p.O, p.OH, p.OW, p.ybs, p.yrs, p.epi, stream);
    return 0;
}
