// Persistent variant of the 8-wave stride-1 split-bf16 convolution (conv2d_bf16x3_kernel<8, false> in conv2d_bf16x3.hip:
// same arithmetic, same LDS layout, same ping-pong wave roles) for the layers with several tiles per CU.
//
// The plain kernel runs one 64-channel x (16 x 32)-pixel tile per workgroup and, at 152 KB of LDS, one workgroup per CU: when
// a tile ends, the matrix pipe idles through the epilogue (per-channel factors fetched, 64 stores per wave drained), the
// dispatch of the next workgroup and its prologue (first patch fetched from HBM, converted, stored, barrier) — measured at
// ~18 % of the kernel (docs/history/DESIGN_rounds1-4.md 3.1b).  Here one workgroup per CU walks its own sequence of tiles and the K loop simply
// continues across tile boundaries: (tile, 16-channel chunk) pairs form ONE software pipeline — while the last chunks of
// tile k are multiplied, the first chunks of tile k+1 are already being fetched, converted and stored into the other LDS
// buffer.  The epilogue of a finished tile runs under the other wave role's MFMA block: waves 4-7 (MFMA first) store their
// tile right after their last MFMA block while waves 0-3 multiply; waves 0-3 (staging first) keep their accumulators and
// store at the top of the next iteration while waves 4-7 multiply.  Per-tile style vectors and epilogue factors are staged
// through small parity-double-buffered LDS arrays one tile ahead.
//
// Restrictions (the launch checks them, everything else takes the plain kernel): no split-K, O % 64 == 0, 32 <= I <= 512,
// linear / leaky-ReLU epilogue without residual.
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ConvPParams {
    const float* x; const bf16x8* wt16; const float* style; float* y;
    int N, I, O, OP64, H, W;
    int tiles_x, tiles_y, tiles_m, total_tiles;
    int64_t xbs, ybs, style_stride, yrs;
    n3d_epilogue epi;
};

struct ConvPTile { int n, m0, y0, x0; };

__device__ __forceinline__ void p_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v[i];
        hi[i] = h;
        lo[i] = (__bf16)(v[i] - (float)h);
    }
}

__global__ __launch_bounds__(512, 2) void conv2d_p_bf16x3_kernel(ConvPParams p) {
    constexpr int NT_ = 512, NW = 8, BM = 64, TH = 16, TW = 32, ICB = 16, TAPS = 9;
    constexpr int PH = TH + 2, PW = TW + 2, PPIX = PH * PW;               // 18 x 34 = 612 patch pixels
    constexpr int B_ITEMS = 2 * PPIX;                                     // (half, pixel) work items
    constexpr int TAP_STEP = NT_ / 256, A_PER_T = (TAPS + TAP_STEP - 1) / TAP_STEP, B_PER_T = (B_ITEMS + NT_ - 1) / NT_;
    constexpr int A_SZ = TAPS * 2 * BM, B_SZ = 2 * PPIX;
    constexpr int SMAX = 512;                                             // max input channels (one style value per thread)
    __shared__ bf16x8 A_hi[2 * A_SZ], A_lo[2 * A_SZ];                     // [buf][tap][half][row]
    __shared__ bf16x8 B_hi[2 * B_SZ], B_lo[2 * B_SZ];                     // [buf][half][pixel]
    __shared__ float s_style[2][SMAX];                                    // [tile parity][channel]
    __shared__ __attribute__((aligned(16))) float s_rs[2][BM];            // [tile parity][channel of the M tile]: demod * gain factors
    __shared__ __attribute__((aligned(16))) float s_bs[2][BM];            //                                       bias
    __shared__ float s_nz[2][TH * TW];                                    // [tile parity][pixel of the tile]: noise * strength

    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;

    // This workgroup's tile sequence.  Hardware block b runs on XCD b % 8: each XCD owns a contiguous range of logical tile
    // ids (M tile fastest), and its gw workgroups take ids  lo + j, lo + j + gw, ...  — at any moment the workgroups of one XCD
    // work on neighbouring ids, i.e. on the same input patches (shared through that XCD's L2), like the plain kernel's grid.
    const int gw = gridDim.x >> 3, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int tq = p.total_tiles >> 3, tr = p.total_tiles & 7;
    const int lo = xcd * tq + min(xcd, tr), cnt = tq + (xcd < tr ? 1 : 0);
    const int ntile = jx < cnt ? (cnt - jx + gw - 1) / gw : 0;
    if (ntile == 0) return;
    const int nstage = p.I / ICB, HW = p.H * p.W, tiles_xy = p.tiles_x * p.tiles_y;
    auto decode = [&](int k) {
        int L = lo + jx + k * gw;
        ConvPTile t;
        t.m0 = (L % p.tiles_m) * BM; L /= p.tiles_m;
        const int ti = L % tiles_xy;
        t.n = L / tiles_xy;
        t.y0 = (ti / p.tiles_x) * TH; t.x0 = (ti % p.tiles_x) * TW;
        return t;
    };

    const n3d_epilogue& E = p.epi;
    const float nstr = E.noise ? E.noise_strength[0] : 0.f;
    const float alpha_eff = E.act == N3D_ACT_LRELU ? E.alpha : 1.f, clamp_eff = E.clamp >= 0.f ? E.clamp : INFINITY;

    // A staging: thread owns (row, half, hl) for taps a_t0, a_t0 + 2, ...
    const int a_row = tid & 63, a_q = (tid >> 6) & 3, a_half = a_q & 1, a_hl = a_q >> 1, a_t0 = tid >> 8;
    const int64_t a_tap_stride = (int64_t)nstage * 4 * p.OP64, a_stage_stride = (int64_t)4 * p.OP64;
    const bf16x8* a_base = p.wt16 + (int64_t)(a_hl * 2 + a_half) * p.OP64 + a_row + a_t0 * a_tap_stride;
    bf16x8* a_dst = (a_hl ? A_lo : A_hi) + a_half * BM + a_row + a_t0 * 2 * BM;

    // ---- load cursor: (tile ld_k, chunk ld_st) of the NEXT load_stage call
    int ld_k = 0, ld_st = 0;
    const bf16x8* ld_a;
    const float* ld_x;
    int b_goff[B_PER_T];
    auto set_load_tile = [&](int k) {
        const ConvPTile t = decode(k);
        ld_a = a_base + t.m0;
        ld_x = p.x + (int64_t)t.n * p.xbs;
#pragma unroll
        for (int j = 0; j < B_PER_T; ++j) {
            const int e = tid + j * NT_;
            const int hf = e / PPIX, pp = e % PPIX;
            const int iy = t.y0 - 1 + pp / PW, ix = t.x0 - 1 + pp % PW;
            const bool ok = e < B_ITEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            b_goff[j] = ok ? (hf * 8 * HW + iy * p.W + ix) * 4 : (int)0x80000000;      // outside the image: beyond the buffer -> 0
        }
    };
    bf16x8 ra[A_PER_T];
    float rb[B_PER_T][8];
    auto load_stage = [&]() {                              // issue only; consumed by the next store_stage
        // the sample base is wave-uniform; say so explicitly, or the descriptor lands in VGPRs and every load becomes a waterfall loop
        const uint64_t xa = (uint64_t)ld_x;
        const uint64_t xu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(xa >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)xa);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xu, 0, p.I * HW * 4, 0x00020000);
        const int st_u = __builtin_amdgcn_readfirstlane(ld_st);
        const bf16x8* as = ld_a + st_u * a_stage_stride;
#pragma unroll
        for (int j = 0; j < A_PER_T; ++j)
            if (a_t0 + TAP_STEP * j < TAPS) ra[j] = as[j * TAP_STEP * a_tap_stride];
#pragma unroll
        for (int j = 0; j < B_PER_T; ++j)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                rb[j][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, b_goff[j], (st_u * ICB + c) * HW * 4, 0));
    };
    auto advance_load = [&]() {                            // called outside the (lane-masked) role branches: the cursors stay scalar
        if (++ld_st == nstage) {
            ld_st = 0;
            if (++ld_k < ntile) set_load_tile(ld_k);
        }
        ld_st = __builtin_amdgcn_readfirstlane(ld_st); ld_k = __builtin_amdgcn_readfirstlane(ld_k);      // scalar cursors (buffer soffset)
    };
    // ---- store cursor: (tile s_k, chunk s_st) of the data held in ra / rb
    int s_k = 0, s_st = 0;
    auto store_stage = [&](int gs) {                       // gs: global stage index -> LDS buffer gs & 1
        const int bo_a = (gs & 1) ? A_SZ : 0, bo_b = (gs & 1) ? B_SZ : 0;
        const float* sty = s_style[s_k & 1] + s_st * ICB;
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
            for (int c = 0; c < 8; ++c) v[c] = rb[j][c] * sty[hf * 8 + c];
            bf16x8 hi, lo;
            p_split8(v, hi, lo);
            B_hi[bo_b + e] = hi;
            B_lo[bo_b + e] = lo;
        }
    };
    auto advance_store = [&]() {
        if (++s_st == nstage) { s_st = 0; ++s_k; }
        s_st = __builtin_amdgcn_readfirstlane(s_st); s_k = __builtin_amdgcn_readfirstlane(s_k);
    };

    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    };
    const int a_frag = half * BM + l31;                                   // + tap*2*BM + mt*32
    const int b_frag0 = half * PPIX + (wn * 2) * PW + l31;                // + ky*PW + kx
    const int b_frag1 = b_frag0 + PW;
    auto mfma_block = [&](int g) {
        const int bo_a = (g & 1) ? A_SZ : 0, bo_b = (g & 1) ? B_SZ : 0;
        __builtin_amdgcn_s_setprio(1);
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
        __builtin_amdgcn_s_setprio(0);
    };
    // epilogue of one finished tile (C/D layout: col = lane&31 = pixel, row = (r&3) + 8*(r>>2) + 4*(lane>>5) = channel)
    const int64_t yplane = (int64_t)p.H * p.yrs;
    auto epilogue = [&](const ConvPTile& t, int par) {
        float* d0[2];
        float nzs[2];
        bool ok[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int oy = t.y0 + wn * 2 + nt, ox = t.x0 + l31;
            ok[nt] = oy < p.H && ox < p.W;
            nzs[nt] = s_nz[par][(wn * 2 + nt) * TW + l31];
            d0[nt] = p.y + (int64_t)t.n * p.ybs + (int64_t)oy * p.yrs + ox + (int64_t)(t.m0 + 4 * half) * yplane;
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                 // 4 consecutive channels: one 16-byte read of each factor array
                const f32x4 rs4 = *reinterpret_cast<const f32x4*>(&s_rs[par][mt * 32 + 8 * q + 4 * half]);
                const f32x4 bs4 = *reinterpret_cast<const f32x4*>(&s_bs[par][mt * 32 + 8 * q + 4 * half]);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (!ok[nt]) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[mt][nt][q * 4 + i] * rs4[i] + nzs[nt] + bs4[i];
                        v = fmaxf(v, v * alpha_eff) * E.gain;
                        d0[nt][(int64_t)(mt * 32 + i + 8 * q) * yplane] = fminf(fmaxf(v, -clamp_eff), clamp_eff);
                    }
                }
            }
    };

    // ---- prologue: style of tile 0, first chunk into LDS buffer 0, second chunk into registers
    ConvPTile ct = decode(0);                                             // the tile being multiplied
    if (tid < p.I) s_style[0][tid] = p.style ? p.style[(int64_t)ct.n * p.style_stride + tid] : 1.f;
    set_load_tile(0);
    zero_acc();
    __syncthreads();
    const int total = ntile * nstage;                                     // nstage >= 2 (launch check)
    load_stage(); advance_load();
    store_stage(0); advance_store();
    __syncthreads();
    load_stage(); advance_load();

    const bool stage_first = wn < NW / 2;
    int c_k = 0, c_st = 0;
    bool epi_pending = false;
    ConvPTile et = ct;
    int epar = 0;
    const int bt = tid - NT_ / 2;                                         // index among the MFMA-first waves' 256 threads
    for (int g = 0; g < total; ++g) {
        const bool first = c_st == 0, last = c_st == nstage - 1;
        // Order inside both roles: the next chunk's store + load come BEFORE a finished tile's epilogue stores, so no wait for a
        // staged load ever has 64 younger stores queued behind it (vmcnt counts in order and saturates at 63).
        if (stage_first) {
            if (g + 1 < total) store_stage(g + 1);
            if (g + 2 < total) load_stage();
            if (epi_pending) { epilogue(et, epar); zero_acc(); epi_pending = false; }          // previous tile, under waves 4-7's MFMA block
            mfma_block(g);
            if (last) { epi_pending = true; et = ct; epar = c_k & 1; }
        } else {
            mfma_block(g);
            if (g + 1 < total) store_stage(g + 1);
            if (g + 2 < total) load_stage();
            if (first) {
                // Per-tile side data through LDS, by these waves' 256 threads while they would otherwise sit at the barrier (waves
                // 0-3 are still multiplying): the style of tile k+1 (first read while the LAST chunk of tile k is multiplied), the
                // epilogue factors and the noise tile of tile k (read by its epilogue, >= 1 barrier later: nstage >= 2).  Load and
                // LDS write are kept together on purpose: no load result stays pending across the loop's back edge.
                float f_s0 = 1.f, f_s1 = 1.f, f_rs = 1.f, f_bs = 0.f, f_n0 = 0.f, f_n1 = 0.f;
                const bool fill_style = c_k + 1 < ntile;
                if (fill_style && p.style) {
                    const float* sp = p.style + (int64_t)decode(c_k + 1).n * p.style_stride;
                    if (bt < p.I) f_s0 = sp[bt];
                    if (bt + 256 < p.I) f_s1 = sp[bt + 256];
                }
                if (bt < BM) {
                    const int o = ct.m0 + bt;
                    if (E.row_scale) f_rs = E.row_scale[(int64_t)ct.n * (E.row_scale_stride ? E.row_scale_stride : p.O) + o];
                    if (E.bias) f_bs = E.bias[o];
                }
                if (E.noise) {
                    const int oy = ct.y0 + (bt >> 5), ox = ct.x0 + (bt & 31);                 // pixels bt and bt + 256 (8 rows further down)
                    if (oy < p.H && ox < p.W) f_n0 = E.noise[(int64_t)oy * p.W + ox];
                    if (oy + 8 < p.H && ox < p.W) f_n1 = E.noise[(int64_t)(oy + 8) * p.W + ox];
                }
                if (fill_style) { s_style[(c_k + 1) & 1][bt] = f_s0; s_style[(c_k + 1) & 1][bt + 256] = f_s1; }
                if (bt < BM) { s_rs[c_k & 1][bt] = f_rs * E.const_scale; s_bs[c_k & 1][bt] = f_bs; }
                s_nz[c_k & 1][bt] = f_n0 * nstr; s_nz[c_k & 1][bt + 256] = f_n1 * nstr;
            }
            if (last) { epilogue(ct, c_k & 1); zero_acc(); }
        }
        if (g + 1 < total) advance_store();
        if (g + 2 < total) advance_load();
        if (last) {
            c_st = 0;
            if (++c_k < ntile) ct = decode(c_k);
        } else {
            ++c_st;
        }
        c_st = __builtin_amdgcn_readfirstlane(c_st); c_k = __builtin_amdgcn_readfirstlane(c_k);
        __syncthreads();
    }
    if (epi_pending) epilogue(et, epar);
}

// Number of workgroups the device keeps resident at one per CU (the kernel's LDS footprint allows no more).
static int conv_p_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 256;
        n = v;
    }
    return n;
}

// 1 = the persistent kernel takes this layer; 0 = the caller launches the plain kernel.  N3D_CONV_PERSIST=0 switches it off.
int conv2d_p_bf16x3_try_launch(const n3d_conv2d_desc* d, int tiles_x, int tiles_y, hipStream_t stream, int* launched) {
    *launched = 0;
    static int enabled = -1;
    if (enabled < 0) enabled = n3d_tune("N3D_CONV_PERSIST", 1);
    const n3d_epilogue& E = d->epi;
    const bool act_ok = E.act == N3D_ACT_LINEAR || (E.act == N3D_ACT_LRELU && E.alpha >= 0.f && E.alpha <= 1.f);
    const int tiles_m = d->O / 64;
    const int64_t total = (int64_t)tiles_x * tiles_y * tiles_m * d->N;
    const int ncu = conv_p_num_cus();
    if (!enabled || d->O % 64 != 0 || d->I < 32 || d->I > 512 || !act_ok || E.residual || total < 2 * (int64_t)ncu || total >= (1ll << 31)) return 0;
    ConvPParams p;
    p.x = d->x; p.wt16 = (const bf16x8*)d->wt; p.style = d->style; p.y = d->y;
    p.N = d->N; p.I = d->I; p.O = d->O; p.OP64 = d->O; p.H = d->H; p.W = d->W;
    p.tiles_x = tiles_x; p.tiles_y = tiles_y; p.tiles_m = tiles_m; p.total_tiles = (int)total;
    p.xbs = d->x_batch_stride; p.ybs = d->y_batch_stride; p.epi = d->epi;
    p.style_stride = d->style_stride ? d->style_stride : d->I;
    p.yrs = d->y_row_stride ? d->y_row_stride : d->W;
    hipLaunchKernelGGL(conv2d_p_bf16x3_kernel, dim3((unsigned)(ncu & ~7)), dim3(512), 0, stream, p);
    N3D_LAUNCH_CHECK();
    *launched = 1;
    return 0;
}
