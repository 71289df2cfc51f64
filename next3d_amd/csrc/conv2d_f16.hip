// 3x3 convolutions of the reference's FLOAT16 blocks (the super-resolution module's default route, sr_num_fp16_res = 4:
// training/networks_stylegan2.py:417-424 `dtype = torch.float16 if self.use_fp16 and not force_fp32`) on the f16 matrix cores:
// ONE v_mfma_f32_32x32x16_f16 per 16-channel chunk and tap — float16 operands, float32 accumulation, exactly the arithmetic of
// the reference's half-precision convolution (every f16 x f16 product is exact in float32) — instead of the three split-bf16
// instructions the float32 route spends per MAC.
//
// The reference runs these layers FUSED (modulated_conv2d, training/networks_stylegan2.py:53-91 with fused_modconv = True at
// inference): the per-sample weights w[n] = weight * styles[n] * dcoefs[n] are formed in float32 and cast to float16
// (`w.to(x.dtype)`, :88) and the activations enter unmodulated.  So here:
//   x : "h8" layout (include/n3d.h)  f16 [N][I/8][H][W][8] — one 16-byte unit = 8 consecutive channels of one pixel, UNMODULATED;
//   w : per-sample f16 weights from n3d_modulate_weights_f16:  [N][tap][I/16][2 (k half)][O][8]   (O % 64 == 0)
//   y : h8 [N][O/8][OH][OW][8]
// One tensor serves every consumer (toRGB and the next block's up-sampling layer read the same h8 image).
// Staging is the LDS-DMA scheme of conv2d_ps_bf16x3.hip (`buffer_load_dwordx4 ... lds`, per-lane source address = patch pixel,
// halo from the descriptor's range check) with half the bytes per chunk; the MFMAs take the weights as matrix rows, so a lane's
// accumulator registers are 4 x 4 consecutive channels of ONE pixel = the h8 unit halves (8-byte stores, a wave instruction
// covers 512 contiguous bytes).
#include <stdlib.h>

#include "common.h"
#include "up_tiles.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int F_TW = 32, F_TAPS = 9;
// Tile shapes of the stride-1 kernel: a workgroup (8 waves) owns MT 32-channel groups x (8 NT rows x 32 pixels); every wave multiplies ALL
// MT channel groups with its own NT rows, i.e. MT + NT fragment reads per MT * NT MFMAs and tap:
//   <2, 2>  64 channels x 16 x 32 pixels   1.00 ds_read_b128 per MFMA — at the matrix pipe's data-sheet rate that is the whole LDS bandwidth
//                                           of a CU (8 waves x 4 KB per 4 MFMAs of 32 cycles on each SIMD = 128 B/clk): the small-grid form;
//   <2, 4>  64 channels x 32 x 32 pixels   0.75 — 0.5 with the row reuse of f16_mfma_chunk (round 4: long-K layers on grids that fill the chip)
//   (<4, 2>, 128 channels x 16 x 32 pixels, 0.75: measured, not faster, not instantiated)
template <int MT, int NT> struct F16Tile {
    static constexpr int BM = 32 * MT, TH = 8 * NT;
    static constexpr int PH = TH + 2, PW = F_TW + 2, PPIX = PH * PW;                     // patch with halo
    static constexpr int BCH = (PPIX + 63) / 64, BPAD = BCH * 64;                         // DMA pieces (64 slots) per k half
    static constexpr int A_SZ = F_TAPS * 2 * BM;                                          // 16-byte slots: [tap][k half][row]
    static constexpr int B_SZ = 2 * BPAD;                                                 //                [k half][patch pixel]
    static constexpr int A_PIECES = F_TAPS * 2 * (BM / 64), B_PIECES = 2 * BCH;           // pieces of 1 KB per chunk
    static constexpr int BUF = A_SZ + B_SZ;                                               // <2,2>: 38,912 B; <4,2> and <2,4>: 57,344 B per buffer
};

struct ConvF16Params {
    const f16x8* x; const f16x8* w; f16x8* y;
    int N, I, O, H, W;
    int tiles_x, tiles_y, tiles_m;
    int64_t xbs, wbs, ybs;       // 16-byte units between consecutive samples
    const float* bias; const float* noise; const float* noise_strength;
    float alpha, gain, clamp;    // alpha = 1: linear; clamp = INFINITY: none
    int multi_round;             // f16_layer_epilogue's `multi`
    int dbg;
    // fused toRGB of a LAST block (n3d_conv2d_desc.rgb_*): per-sample float16 toRGB weights [N][RC][O] (n3d_modulate_weights_f16, demodulate = 0);
    // the partial colours of this workgroup's 64 channels go to rgb_partial [N][O/64][RC][H][W]; y may be NULL
    const _Float16* rgb_w16; float* rgb_partial; int rgb_channels;
};
constexpr int F_RGB_MAX = 4;

// The layer epilogue in the reference's float16 order (training/networks_stylegan2.py:91 + :327-329): the convolution returns
// float16; `x.add_(noise)` rounds again; bias_act reads float16 x and b (`self.bias.to(x.dtype)`) and returns float16.
//   multi == 0: bias_act.cu:19-50 — `InternalType<c10::Half>::scalar_t` = float: (x + b) -> lrelu -> * gain -> clamp in float32, ONE
//               rounding.  What the reference does on a GPU; the default.
//   multi != 0: _bias_act_ref on half tensors (bias_act.py:93-122, what the reference runs off-GPU): every step is a float16
//               tensor op — x + b, leaky_relu, x * gain (skipped when gain == 1) each round to float16.  n3d_epilogue.round_f16 = 2;
//               it exists so that the reference's own CPU run of its float16 branch (tests/golden/*_fp16sr.npz) can be matched.
__device__ __forceinline__ _Float16 f16_layer_epilogue(float acc, float nz, bool has_noise, float b16, float alpha, float gain, float clamp, int multi) {
    float v = (float)(_Float16)acc;
    if (has_noise) v = (float)(_Float16)(v + nz);
    float t = v + b16;
    if (multi) {
        t = (float)(_Float16)t;
        t = (float)(_Float16)fmaxf(t, t * alpha);
        if (gain != 1.f) t = (float)(_Float16)(t * gain);
        return (_Float16)__builtin_amdgcn_fmed3f(t, -clamp, clamp);
    }
    t = fmaxf(t, t * alpha) * gain;
    return (_Float16)__builtin_amdgcn_fmed3f(t, -clamp, clamp);
}

// One 16-channel chunk of a wave's MT x NT accumulator tiles from the staged weights A [tap][k half][row] and patch B [k half][patch pixel].
// Every accumulator sums its taps in the order kx-major, ky-minor in BOTH loop forms, so the tile shapes are bit-identical.
//   plain (MT > 2): per tap MT + NT fragment reads for MT * NT MFMAs;
//   row reuse (MT <= 2): the B fragment of patch row pr and column offset kx serves the taps ky = 0..2 of the rows nt = pr - ky — it is read
//   ONCE: per kx 3 MT + NT + 2 reads for 3 MT NT MFMAs (<2,4>: 0.5 ds_read_b128 per MFMA, <2,2>: 0.83).
template <int MT, int NT>
__device__ __forceinline__ void f16_mfma_chunk(const f16x8* A, const f16x8* B, int a_frag, int b_frag0, f32x16 (&acc)[MT][NT]) {
    constexpr int BM = F16Tile<MT, NT>::BM, PW = F16Tile<MT, NT>::PW;
    if constexpr (MT <= 2) {
        f16x8 a[2][3][MT], b[2];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[0][ky][mt] = A[(ky * 3) * 2 * BM + a_frag + mt * 32];
        b[0] = B[b_frag0];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            if (kx + 1 < 3) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a[(kx + 1) & 1][ky][mt] = A[(ky * 3 + kx + 1) * 2 * BM + a_frag + mt * 32];
            }
#pragma unroll
            for (int pr = 0; pr < NT + 2; ++pr) {
                const int c = kx * (NT + 2) + pr;                         // running fragment counter (compile time after unrolling)
                if (pr + 1 < NT + 2) b[(c + 1) & 1] = B[b_frag0 + (pr + 1) * PW + kx];
                else if (kx + 1 < 3) b[(c + 1) & 1] = B[b_frag0 + kx + 1];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int nt = pr - ky;
                    if (nt >= 0 && nt < NT) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)                    // weights = matrix rows: D[channel row][pixel column]
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kx & 1][ky][mt], b[c & 1], acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
    } else {
        f16x8 a[2][MT], b[2][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[0][mt] = A[a_frag + mt * 32];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[0][nt] = B[b_frag0 + nt * PW];
#pragma unroll
        for (int tt = 0; tt < F_TAPS; ++tt) {                             // tt = kx * 3 + ky
            const int s = tt & 1;
            if (tt + 1 < F_TAPS) {
                const int ky = (tt + 1) % 3, kx = (tt + 1) / 3, t = ky * 3 + kx;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[s ^ 1][mt] = A[t * 2 * BM + a_frag + mt * 32];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[s ^ 1][nt] = B[b_frag0 + (nt + ky) * PW + kx];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][mt], b[s][nt], acc[mt][nt], 0, 0, 0);
        }
    }
}

// NBUF = 2: one workgroup per CU, the next chunk's DMA under this chunk's MFMAs; NBUF = 1: two workgroups per CU with one buffer
// each (the other workgroup's MFMAs fill this one's DMA waits and its tile stores) — see conv2d_ps_bf16x3_body.
template <int MT, int NT, int NBUF, bool RGB>
__device__ __forceinline__ void conv2d_h8_f16_body(const ConvF16Params& p, f16x8* smem) {
    using T = F16Tile<MT, NT>;
    constexpr int F_BM = T::BM, F_TH = T::TH, F_PW = T::PW, F_PPIX = T::PPIX, F_BCH = T::BCH, F_BPAD = T::BPAD, F_A_SZ = T::A_SZ, F_A_PIECES = T::A_PIECES,
                  F_B_PIECES = T::B_PIECES, F_BUF = T::BUF;
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int lb;
    {   // XCD-aware 1-D grid, M tile fastest: the O/64 workgroups reading one input patch share it in one L2
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int m0 = (lb % p.tiles_m) * F_BM; lb /= p.tiles_m;
    const int tile_i = lb % (p.tiles_x * p.tiles_y), n = lb / (p.tiles_x * p.tiles_y);
    const int y0 = (tile_i / p.tiles_x) * F_TH, x0 = (tile_i % p.tiles_x) * F_TW;
    const int KC = p.I / 16, HW = p.H * p.W;

    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (int64_t)n * p.wbs), 0, F_TAPS * KC * 2 * p.O * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.xbs), 0, (p.I / 8) * HW * 16, 0x00020000);

    // this wave's copy pieces: weights pa = wn + 8 j = (tap, k half, 64-row group) slabs; patch q = wn + 8 j = (k half, 64-pixel run)
    constexpr int NA = (F_A_PIECES + 7) / 8, NB = (F_B_PIECES + 7) / 8, RG = F_BM / 64;
    int ldsA[NA], sofA[NA], voffA[NA], ldsB[NB], sofB[NB], voffB[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int pa = min(wn + 8 * j, F_A_PIECES - 1), rg = pa % RG, t = (pa / RG) >> 1, hf = (pa / RG) & 1;
        ldsA[j] = (t * 2 + hf) * F_BM + rg * 64;
        sofA[j] = ((t * KC) * 2 + hf) * p.O * 16;
        voffA[j] = (m0 + rg * 64 + lane) * 16;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int q = min(wn + 8 * j, F_B_PIECES - 1), hf = q / F_BCH, c = q % F_BCH;
        ldsB[j] = F_A_SZ + hf * F_BPAD + c * 64;
        sofB[j] = hf * HW * 16;
        const int pp = c * 64 + lane;
        const int iy = y0 - 1 + pp / F_PW, ix = x0 - 1 + pp % F_PW;
        const bool ok = pp < F_PPIX && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        voffB[j] = ok ? (iy * p.W + ix) * 16 : (int)0x80000000;
    }
    const int strideA = 2 * p.O * 16, strideB = 2 * HW * 16;              // scalar-offset step per 16-channel chunk
    auto copy_chunk = [&](int kc, int buf) {
        f16x8* base = smem + buf * F_BUF;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (wn + 8 * j < F_A_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_void*)(base + ldsA[j]), 16, voffA[j], sofA[j] + kc * strideA, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (wn + 8 * j < F_B_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x, (lds_void*)(base + ldsB[j]), 16, voffB[j], sofB[j] + kc * strideB, 0, 0);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const int a_frag = half * F_BM + l31;                                 // + tap*2*BM + mt*32
    const int b_frag0 = half * F_BPAD + (wn * NT) * F_PW + l31;           // + ky*PW + kx (+ nt*PW for the wave's further rows)
    auto mfma_block = [&](int buf) {
        const f16x8* A = smem + buf * F_BUF;
        __builtin_amdgcn_s_setprio(1);
        f16_mfma_chunk<MT, NT>(A, A + F_A_SZ, a_frag, b_frag0, acc);
        __builtin_amdgcn_s_setprio(0);
    };

    float* s_bs = reinterpret_cast<float*>(smem + NBUF * F_BUF);
    if (tid < F_BM) s_bs[tid] = p.bias ? (float)(_Float16)p.bias[m0 + tid] : 0.f;          // self.bias.to(float16)
    float* s_cw = s_bs + F_BM;                                            // RGB: [colour][channel of this tile] toRGB weights of this sample
    if (RGB && tid < F_RGB_MAX * F_BM) {
        const int j = tid / F_BM, c = tid % F_BM;
        s_cw[tid] = j < p.rgb_channels ? (float)p.rgb_w16[((int64_t)n * p.rgb_channels + j) * p.O + m0 + c] : 0.f;
    }

    if (NBUF == 1) {
        __builtin_amdgcn_s_barrier();
        for (int kc = 0; kc < KC; ++kc) {
            if (!(p.dbg & 4) || kc == 0) copy_chunk(kc, 0);
            __builtin_amdgcn_s_waitcnt(0x0f70);                           // vmcnt(0): this wave's pieces are in LDS ...
            __builtin_amdgcn_s_barrier();                                 // ... and after the barrier everybody's are
            if (!(p.dbg & 2)) mfma_block(0);
            __builtin_amdgcn_s_barrier();                                 // every wave has read its fragments: the buffer may be refilled
        }
    } else {
        copy_chunk(0, 0);
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_s_barrier();
        for (int kc = 0; kc < KC; ++kc) {
            if (kc + 1 < KC && !(p.dbg & 4)) copy_chunk(kc + 1, (kc + 1) & 1);
            if (!(p.dbg & 2)) mfma_block(kc & 1);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (p.dbg & 1) { if (acc[0][0][0] == 123.456f) p.y[0][0] = (_Float16)1.f; return; }

    // epilogue.  D layout: column = lane & 31 = pixel x0 + l31 of the wave's row nt; register r = channel (r & 3) + 8 (r >> 2) + 4 half of
    // group mt -> registers 4 gg .. 4 gg + 3 are the channels 8 gg + 4 half + (0..3): one half of the h8 unit (m0 / 8 + 4 mt + gg).
    const float nstr = p.noise ? p.noise_strength[0] : 0.f;
    const int ox = x0 + l31;
    if constexpr (RGB) {
        // Fused toRGB (ToRGBLayer on a float16 block: float16 x and weights, float32 accumulation — n3d_torgb_h8's sum).  A lane holds 32 of the
        // tile's 64 channels of its pixel (the other 32 sit in lane ^ 32): it sums its share over the ROUNDED layer outputs, the halves are
        // added across the lane pair, and the 64-channel partial colours go to rgb_partial; n3d_rgb_combine adds the tiles' partial images
        // in index order and applies toRGB's float16 epilogue.
        float col[NT][F_RGB_MAX], nz[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int oy = min(y0 + wn * NT + nt, p.H - 1);
            nz[nt] = p.noise ? p.noise[(int64_t)oy * p.W + min(ox, p.W - 1)] * nstr : 0.f;
#pragma unroll
            for (int j = 0; j < F_RGB_MAX; ++j) col[nt][j] = 0.f;
        }
        // pass 1: the layer outputs, rounded (the accumulators die here: 64 registers become 32 packed float16 pairs)
        f16x4 o16[MT][4][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const f32x4 bs = *reinterpret_cast<const f32x4*>(s_bs + mt * 32 + 8 * gg + 4 * half);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        o16[mt][gg][nt][k] = f16_layer_epilogue(acc[mt][nt][4 * gg + k], nz[nt], p.noise != nullptr, bs[k], p.alpha, p.gain, p.clamp, p.multi_round);
            }
        if (p.y != nullptr && ox < p.W) {
            _Float16* yb16 = reinterpret_cast<_Float16*>(p.y + (int64_t)n * p.ybs);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int oy = y0 + wn * NT + nt;
                if (oy >= p.H) continue;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg)
                        *reinterpret_cast<f16x4*>(yb16 + (((int64_t)((m0 >> 3) + mt * 4 + gg) * p.H + oy) * p.W + ox) * 8 + 4 * half) = o16[mt][gg][nt];
            }
        }
        // pass 2: colours
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int ol = mt * 32 + 8 * gg + 4 * half;
#pragma unroll
                for (int j = 0; j < F_RGB_MAX; ++j) {
                    const f32x4 cw = *reinterpret_cast<const f32x4*>(s_cw + j * F_BM + ol);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int k = 0; k < 4; ++k) col[nt][j] = fmaf((float)o16[mt][gg][nt][k], cw[k], col[nt][j]);
                }
            }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < F_RGB_MAX; ++j) col[nt][j] += __shfl_xor(col[nt][j], 32);
        if (half == 0 && ox < p.W) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int oy = y0 + wn * NT + nt;
                if (oy >= p.H) continue;
                float* dst = p.rgb_partial + (((int64_t)n * p.tiles_m + m0 / F_BM) * p.rgb_channels) * (int64_t)p.H * p.W + (int64_t)oy * p.W + ox;
#pragma unroll
                for (int j = 0; j < F_RGB_MAX; ++j)
                    if (j < p.rgb_channels) dst[(int64_t)j * p.H * p.W] = col[nt][j];
            }
        }
    }
    const bool store_y = !RGB && ox < p.W;                               // (RGB: stored above; no early `return` in front of stores: see conv2d_ps_bf16x3.hip)
    _Float16* yb = reinterpret_cast<_Float16*>(p.y + (int64_t)n * p.ybs);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int oy = y0 + wn * NT + nt;
        if (oy >= p.H || !store_y) continue;
        const float nz = p.noise ? p.noise[(int64_t)oy * p.W + ox] * nstr : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int ol = mt * 32 + 8 * gg + 4 * half;
                const f32x4 bs = *reinterpret_cast<const f32x4*>(s_bs + ol);
                f16x4 out;
#pragma unroll
                for (int k = 0; k < 4; ++k) out[k] = f16_layer_epilogue(acc[mt][nt][4 * gg + k], nz, p.noise != nullptr, bs[k], p.alpha, p.gain, p.clamp, p.multi_round);
                const int64_t unit = ((int64_t)((m0 >> 3) + mt * 4 + gg) * p.H + oy) * p.W + ox;
                *reinterpret_cast<f16x4*>(yb + unit * 8 + 4 * half) = out;
            }
    }
}

// (concrete kernels around the body template: hipcc's host pass does not emit the launch stub of a __global__ TEMPLATE with this body)
template <int MT, int NT, int NBUF, bool RGB = false> constexpr int f16_smem_slots() { return NBUF * F16Tile<MT, NT>::BUF + (1 + (RGB ? F_RGB_MAX : 0)) * F16Tile<MT, NT>::BM * 4 / 16; }
__global__ __launch_bounds__(512, 2) void conv2d_h8_f16_kernel(ConvF16Params p) {             // <2,2>, two buffers
    __shared__ f16x8 smem[f16_smem_slots<2, 2, 2>()];
    conv2d_h8_f16_body<2, 2, 2, false>(p, smem);
}
__global__ __launch_bounds__(512, 2) void conv2d_h8_f16_rgb_kernel(ConvF16Params p) {         // <2,2> + fused toRGB (a network's last layer)
    __shared__ f16x8 smem[f16_smem_slots<2, 2, 2, true>()];
    conv2d_h8_f16_body<2, 2, 2, true>(p, smem);
}
__global__ __launch_bounds__(512, 4) void conv2d_h8_f16_nbuf1_kernel(ConvF16Params p) {       // <2,2>, one buffer, two workgroups per CU (tuning builds)
    __shared__ f16x8 smem[f16_smem_slots<2, 2, 1>()];
    conv2d_h8_f16_body<2, 2, 1, false>(p, smem);
}
__global__ __launch_bounds__(512, 2) void conv2d_h8_f16_r32_kernel(ConvF16Params p) {         // <2,4>: 64 channels x 32 x 32 pixels
    __shared__ f16x8 smem[f16_smem_slots<2, 4, 2>()];
    conv2d_h8_f16_body<2, 4, 2, false>(p, smem);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Transposed 3x3 stride-2 convolution of a float16 block (conv2d_resample.py:114-127: conv_transpose2d(stride = 2) of the
// up-sampling layer; the FIR behind it is n3d_fir4_h8): all four output phases from one staged patch exactly as
// conv2d_up_ps_body (tap (ky, kx) feeds phase (ky == 1, kx == 1) from patch offset (ky == 2 ? 0 : 1, kx == 2 ? 0 : 1); flattened
// th x tw <= 256 positions of the (H+1) x (W+1) position grid per tile).  The result is the float16 tensor the reference's
// conv_transpose2d returns: the only epilogue is the rounding.

constexpr int FU_PPIX = 9 * 33;                                           // patch capacity: (th+1) x (tw+1) <= 297
constexpr int FU_BCH = (FU_PPIX + 63) / 64, FU_BPAD = FU_BCH * 64;        // 5 pieces = 320 slots per k half
constexpr int FU_B_SZ = 2 * FU_BPAD, FU_B_PIECES = 2 * FU_BCH;            // 640 slots, 10 pieces

struct ConvUpF16Params {
    const f16x8* x; const f16x8* w; f16x8* y;
    int N, I, O, H, W, OH, OW;
    UpTilePlan plan; int tiles_m;
    int64_t xbs, wbs, ybs;
    int thin_last;
    int dbg;
};

constexpr int fu_buf_slots(int nmt) { return F_TAPS * 2 * 32 * nmt + FU_B_SZ; }      // per buffer, 16-byte slots

// NMT = 32-channel groups per workgroup, NW waves, PG 32-position groups per wave (NW * PG = 8).
template <int NMT, int NW, int PG>
__device__ __forceinline__ void conv2d_up_f16_body(const ConvUpF16Params& p, f16x8* smem) {
    static_assert(NW * PG == 8, "256 positions per tile");
    constexpr int BM = 32 * NMT, A_SZ = F_TAPS * 2 * BM, BUF = fu_buf_slots(NMT);
    constexpr int A_PIECES = F_TAPS * NMT;                                // 64-slot pieces: NMT = 2 (tap, k half) x 64 rows; NMT = 1 (tap): 2 halves x 32 rows
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int lb;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    // Launch order: every FULL tile of the layer first, the thin edge tiles (position row gy = H, position column gx = W) of all samples
    // last.  With two workgroups per CU (512 slots) a 64 x 64 layer at batch 4 is 512 full + 64 thin workgroups, a 128 x 128 one
    // 1024 + 32: the tail of the launch then consists of thin tiles only, whose idle waves skip their multiplies (wave_on below),
    // instead of a last round of full tiles on an eighth of the chip.
    int m0, tile_i, n;
    {
        const int main_per = p.plan.tiles_x * p.plan.tiles_y, thin_per = p.plan.total - main_per;
        const int main_total = p.thin_last ? main_per * p.tiles_m * p.N : 0;
        if (lb < main_total) { m0 = (lb % p.tiles_m) * BM; lb /= p.tiles_m; tile_i = lb % main_per; n = lb / main_per; }
        else if (p.thin_last) { lb -= main_total; m0 = (lb % p.tiles_m) * BM; lb /= p.tiles_m; tile_i = main_per + lb % thin_per; n = lb / thin_per; }
        else { m0 = (lb % p.tiles_m) * BM; lb /= p.tiles_m; tile_i = lb % p.plan.total; n = lb / p.plan.total; }       // (tuning builds: the old order, A/B)
    }
    int y0, x0, th, tw, end_y, end_x;
    up_tile_decode(p.plan, tile_i, y0, x0, th, tw, end_y, end_x);
    const bool wave_on = (wn * PG) * 32 < th * tw;                         // wave-uniform: does this wave own any position of the tile?
    const int PW = tw + 1, prows = th + 1;
    const int KC = p.I / 16, HW = p.H * p.W, GH = p.H + 1, GW = p.W + 1;

    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (int64_t)n * p.wbs), 0, F_TAPS * KC * 2 * p.O * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.xbs), 0, (p.I / 8) * HW * 16, 0x00020000);
    constexpr int NA = (A_PIECES + NW - 1) / NW, NB = (FU_B_PIECES + NW - 1) / NW;
    int ldsA[NA], sofA[NA], ldsB[NB], sofB[NB], voffB[NB];
    const int voffA = NMT == 2 ? (m0 + lane) * 16 : ((lane >> 5) * p.O + m0 + (lane & 31)) * 16;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int pa = min(wn + NW * j, A_PIECES - 1);
        const int t = NMT == 2 ? pa >> 1 : pa, hf = NMT == 2 ? pa & 1 : 0;
        ldsA[j] = (t * 2 + hf) * BM;
        sofA[j] = ((t * KC) * 2 + hf) * p.O * 16;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int q = min(wn + NW * j, FU_B_PIECES - 1), hf = q / FU_BCH, c = q % FU_BCH;
        ldsB[j] = A_SZ + hf * FU_BPAD + c * 64;
        sofB[j] = hf * HW * 16;
        const int pp = c * 64 + lane;                                     // patch pixel of this lane (row-major, run-time pitch PW)
        const int pr = pp / PW, iy = y0 - 1 + pr, ix = x0 - 1 + pp % PW;
        const bool ok = pr < prows && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        voffB[j] = ok ? (iy * p.W + ix) * 16 : (int)0x80000000;
    }
    const int strideA = 2 * p.O * 16, strideB = 2 * HW * 16;
    f32x16 acc[NMT][PG][4];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int g = 0; g < PG; ++g)
#pragma unroll
            for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][g][ph][r] = 0.f;
    const int a_frag = half * BM + l31;
    bool q_act[PG];
    int q_row[PG], q_col[PG], b_frag[PG];
#pragma unroll
    for (int g = 0; g < PG; ++g) {
        const int q_pos = (wn * PG + g) * 32 + l31;                       // flattened tile position of this lane
        q_act[g] = q_pos < th * tw;
        q_row[g] = q_act[g] ? q_pos / tw : 0; q_col[g] = q_act[g] ? q_pos % tw : 0;
        b_frag[g] = half * FU_BPAD + q_row[g] * PW + q_col[g];            // + dy*PW + dx
    }
    for (int kc = -1; kc < KC; ++kc) {
        if (kc + 1 < KC && (!(p.dbg & 4) || kc < 0)) {
            f16x8* base = smem + ((kc + 1) & 1) * BUF;
#pragma unroll
            for (int j = 0; j < NA; ++j)
                if (wn + NW * j < A_PIECES)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_void*)(base + ldsA[j]), 16, voffA, sofA[j] + (kc + 1) * strideA, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (wn + NW * j < FU_B_PIECES)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x, (lds_void*)(base + ldsB[j]), 16, voffB[j], sofB[j] + (kc + 1) * strideB, 0, 0);
        }
        if (kc >= 0 && wave_on) {
            const f16x8* A = smem + (kc & 1) * BUF, *B = A + A_SZ;
            __builtin_amdgcn_s_setprio(1);
            f16x8 b[PG][4];
#pragma unroll
            for (int g = 0; g < PG; ++g)
#pragma unroll
                for (int d = 0; d < 4; ++d) b[g][d] = B[b_frag[g] + (d >> 1) * PW + (d & 1)];
            f16x8 a[2][NMT];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) a[0][mt] = A[a_frag + mt * 32];
#pragma unroll
            for (int t = 0; t < F_TAPS; ++t) {
                const int ky = t / 3, kx = t % 3, s = t & 1;
                const int ph = (ky == 1 ? 2 : 0) + (kx == 1 ? 1 : 0);
                const int d = (ky == 2 ? 0 : 2) + (kx == 2 ? 0 : 1);
                if (t + 1 < F_TAPS) {
#pragma unroll
                    for (int mt = 0; mt < NMT; ++mt) a[s ^ 1][mt] = A[(t + 1) * 2 * BM + a_frag + mt * 32];
                }
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                    for (int g = 0; g < PG; ++g)
                        acc[mt][g][ph] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][mt], b[g][d], acc[mt][g][ph], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_s_barrier();
    }

    if (p.dbg & 1) { if (acc[0][0][0][0] == 123.456f) p.y[0][0] = (_Float16)1.f; return; }
    _Float16* yb = reinterpret_cast<_Float16*>(p.y + (int64_t)n * p.ybs);
#pragma unroll
    for (int g = 0; g < PG; ++g) {
        const int gy = y0 + q_row[g], gx = x0 + q_col[g];
        if (!q_act[g] || gy >= end_y || gx >= end_x) continue;
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
            const int oy = 2 * gy + pa;
            if (oy >= p.OH) continue;
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb) {                       // the two horizontally adjacent phases back to back (adjacent 16-byte units)
                        const int ox = 2 * gx + pb;
                        if (ox >= p.OW) continue;
                        const f32x16& a = acc[mt][g][pa * 2 + pb];
                        const f16x4 v = {(_Float16)a[4 * gg + 0], (_Float16)a[4 * gg + 1], (_Float16)a[4 * gg + 2], (_Float16)a[4 * gg + 3]};
                        const int64_t unit = ((int64_t)((m0 >> 3) + mt * 4 + gg) * p.OH + oy) * p.OW + ox;
                        *reinterpret_cast<f16x4*>(yb + unit * 8 + 4 * half) = v;
                    }
        }
    }
}

__global__ __launch_bounds__(256, 2) void conv2d_up_h8_f16_w64_kernel(ConvUpF16Params p) {       // 4 waves x 64 positions x 32 channels
    __shared__ f16x8 smem[2 * fu_buf_slots(1)];
    conv2d_up_f16_body<1, 4, 2>(p, smem);
}
// (the 8 waves x 32 positions x 32 / 64 channel forms of the body measured 16-19 % slower in round 3 and are not instantiated)

// ------------------------------------------------------------------------------------------------------------------------------
extern "C" int n3d_conv2d_f16(const n3d_conv2d_desc* d, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(d && d->ksize == 3 && (d->mode == 0 || d->mode == 2), "conv2d_f16: 3x3, stride 1 (mode 0) or transposed stride 2 (mode 2)");
    N3D_CHECK(d->x_layout == N3D_LAYOUT_H8_F16 && d->y_layout == N3D_LAYOUT_H8_F16, "conv2d_f16: input and output in the h8 layout (N3D_LAYOUT_H8_F16)");
    N3D_CHECK(!d->side_split8, "conv2d_f16: no side output");
    const bool rgb = d->rgb_partial != nullptr;
    N3D_CHECK(!rgb || (d->mode == 0 && d->rgb_weight && !d->rgb_style && d->rgb_channels >= 1 && d->rgb_channels <= F_RGB_MAX),
              "conv2d_f16: fused toRGB on the stride-1 kernel: rgb_weight = per-sample float16 weights [N][C <= 4][O], rgb_style NULL");
    N3D_CHECK(d->style == nullptr && d->ksplit <= 1, "conv2d_f16: the modulation is part of the per-sample weights (n3d_modulate_weights_f16); no split-K");
    N3D_CHECK(d->N >= 0 && d->I >= 16 && d->I % 16 == 0 && d->O >= 64 && d->O % 64 == 0, "conv2d_f16: I %% 16 == 0, O %% 64 == 0");
    if (d->N == 0) return 0;
    N3D_CHECK(d->x && d->wt && (d->y || rgb) && (((uintptr_t)d->x | (uintptr_t)d->wt | (uintptr_t)d->y) & 15) == 0, "conv2d_f16: null or misaligned tensor");
    N3D_CHECK((int64_t)(d->I / 8) * d->H * d->W * 16 < (1ll << 31), "conv2d_f16: one sample's input exceeds 2 GiB (32-bit buffer offsets)");
    N3D_CHECK(d->x_batch_stride == 0 && d->y_batch_stride == 0 && d->x_row_stride == 0 && d->y_row_stride == 0, "conv2d_f16: dense h8 tensors only (strides 0)");
    const n3d_epilogue& E = d->epi;
    N3D_CHECK(!E.row_scale && E.const_scale == 1.f && !E.residual && !E.residual_up_filter, "conv2d_f16: no row scale / residual (the demodulation is in the weights)");
    const int dbg = n3d_tune("N3D_CONV_DBG", 0);
    const int KC = d->I / 16;
    const int64_t wbs = (int64_t)F_TAPS * KC * 2 * d->O;
    const double flops = 2.0 * d->N * (double)d->O * d->I * 9 * (double)d->H * d->W;
    if (d->mode == 0) {
        N3D_CHECK(d->H >= 16 && d->W >= 32, "conv2d_f16: images of at least 16 x 32");
        N3D_CHECK(E.act == N3D_ACT_LINEAR || (E.act == N3D_ACT_LRELU && E.alpha >= 0.f && E.alpha <= 1.f), "conv2d_f16: linear or leaky-ReLU epilogue only");
        N3D_CHECK(!E.noise || E.noise_strength, "conv2d_f16: noise without noise_strength");
        ConvF16Params p;
        p.x = (const f16x8*)d->x; p.w = (const f16x8*)d->wt; p.y = (f16x8*)d->y;
        p.N = d->N; p.I = d->I; p.O = d->O; p.H = d->H; p.W = d->W;
        // tile shape (F16Tile): the 0.75-reads-per-MFMA forms when their grid still covers the chip
        const int nbuf = n3d_tune("N3D_F16_NBUF", 2);     // measured (tools/f16_bench.py, 1024 / 2048 workgroups): two buffers 263 / 285 us, one buffer + two workgroups per CU 285 / 317 us
        // <2,4> (one workgroup per CU, 115 KB of LDS) where its grid still covers the chip AND the K loop is long enough to amortise a tile's
        // stores, which nothing overlaps then; measured (tools/f16_bench.py, batch 4): 256 -> 256 at 256 x 256: 264 us against 281 for <2,2>;
        // 128 -> 128 at 512 x 512 (8 chunks per tile): 319 against 286 — <2,2>'s 78 KB let two workgroups share a CU.  (A 128-channel
        // <4,2> form measured 274 / 324 us and is not built.)
        const int wide_sel = n3d_tune("N3D_F16_WIDE", -1);                // tuning builds: 0 = <2,2> always, 2 = <2,4> wherever the image has 32 rows
        int shape = 0;
        if (nbuf == 2 && d->H >= 32 && !rgb) {
            const bool ok24 = d->I >= 256 && (int64_t)cdiv(d->W, F_TW) * cdiv(d->H, 32) * d->N * (d->O / 64) >= 256;
            shape = wide_sel < 0 ? (ok24 ? 2 : 0) : (wide_sel == 2 ? 2 : 0);
        }
        const int th = shape == 2 ? 32 : 16;
        p.tiles_x = cdiv(d->W, F_TW); p.tiles_y = cdiv(d->H, th); p.tiles_m = d->O / 64;
        p.xbs = (int64_t)(d->I / 8) * d->H * d->W; p.wbs = wbs; p.ybs = (int64_t)(d->O / 8) * d->H * d->W;
        p.bias = E.bias; p.noise = E.noise; p.noise_strength = E.noise_strength;
        p.alpha = E.act == N3D_ACT_LRELU ? E.alpha : 1.f; p.gain = E.gain; p.clamp = E.clamp >= 0.f ? E.clamp : INFINITY;
        p.multi_round = E.round_f16 == 2; p.dbg = dbg;
        p.rgb_w16 = (const _Float16*)d->rgb_weight; p.rgb_partial = d->rgb_partial; p.rgb_channels = d->rgb_channels;
        const int64_t nblk = (int64_t)p.tiles_x * p.tiles_y * p.tiles_m * p.N;
        N3D_CHECK(nblk < (1ll << 31), "conv2d_f16: grid too large");
        const double bytes = 2.0 * ((double)d->N * d->I * d->H * d->W + (double)d->N * d->O * d->H * d->W + (double)d->N * d->O * d->I * 9);
        N3dProfScope prof(N3D_K_CONV2D_F16, stream, flops, bytes);
        if (rgb) hipLaunchKernelGGL(conv2d_h8_f16_rgb_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
        else if (shape == 2) hipLaunchKernelGGL(conv2d_h8_f16_r32_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
        else if (nbuf == 2) hipLaunchKernelGGL(conv2d_h8_f16_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
        else hipLaunchKernelGGL(conv2d_h8_f16_nbuf1_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
        N3D_LAUNCH_CHECK();
        return 0;
    }
    N3D_CHECK(d->H >= 4 && d->W >= 4, "conv2d_f16 (transposed): input of at least 4 x 4");
    N3D_CHECK(E.act == N3D_ACT_LINEAR && !E.noise && !E.bias && E.clamp < 0.f && E.gain == 1.f,
              "conv2d_f16 (transposed): no epilogue (the layer epilogue runs behind the FIR, n3d_fir4_h8)");
    ConvUpF16Params p;
    p.x = (const f16x8*)d->x; p.w = (const f16x8*)d->wt; p.y = (f16x8*)d->y;
    p.N = d->N; p.I = d->I; p.O = d->O; p.H = d->H; p.W = d->W; p.OH = 2 * d->H + 1; p.OW = 2 * d->W + 1;
    p.plan = up_tile_plan(d->H, d->W, true);
    p.tiles_m = d->O / 32;
    p.xbs = (int64_t)(d->I / 8) * d->H * d->W; p.wbs = wbs; p.ybs = (int64_t)(d->O / 8) * p.OH * p.OW;
    p.dbg = dbg;
    p.thin_last = n3d_tune("N3D_UP_THIN_LAST", 1);
    const int64_t nblk = (int64_t)p.plan.total * p.tiles_m * p.N;
    N3D_CHECK(nblk < (1ll << 31) && nblk > 0, "conv2d_f16: grid too large");
    const double bytes = 2.0 * ((double)d->N * d->I * d->H * d->W + (double)d->N * d->O * p.OH * p.OW + (double)d->N * d->O * d->I * 9);
    N3dProfScope prof(N3D_K_CONV2D_F16, stream, flops, bytes);
    hipLaunchKernelGGL(conv2d_up_h8_f16_w64_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}
