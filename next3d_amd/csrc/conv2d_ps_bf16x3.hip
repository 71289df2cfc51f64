// 3x3 stride-1 convolution on the bf16 matrix cores whose ACTIVATIONS ARRIVE ALREADY SPLIT ("split8" layout, include/n3d.h):
// the operand split (hi = bf16(x), lo = bf16(x - hi)) and the style modulation were done ONCE by the producer's epilogue
// (csrc/upfirdn2d.hip: fir4_c8_split8_kernel, or elementwise.hip: split8_from_nchw_kernel), so this kernel's K loop contains no staging arithmetic at all.
//
//   x  : [N][2 (hi, lo)][I/8][H][W][8] bf16 — one 16-byte unit = 8 consecutive channels of one pixel, already multiplied by
//        this layer's style (modulation, tat/networks_stylegan2.py:70) by whoever wrote it;
//   wt : the same split K-major weight tiles as conv2d_bf16x3.hip, shared by the whole batch.
//
// Staging is a pure copy done by LDS-DMA (`buffer_load_dwordx4 ... lds`): per 16-channel chunk a workgroup issues 36 weight
// pieces + 40 patch pieces of 1 KB (64 lanes x 16 B, lane-linear in LDS, per-lane source address = patch pixel; the halo
// comes from the buffer descriptor's range check) — 9.5 instructions per wave instead of 24 four-byte gathers, ~100 VALU
// (style multiply, two conversions, a subtraction, packing per value) and 11 ds_write_b128 in conv2d_bf16x3_kernel.  With
// nothing to convert there are no wave roles: all eight waves issue the next chunk's DMA, multiply the current chunk, wait
// for their own pieces (s_waitcnt vmcnt(0)) and meet at ONE raw s_barrier per chunk; two LDS buffers (2 x 76 KB), or one buffer and two workgroups per CU (NBUF).
// Tile: 64 output channels x (16 x 32) pixels, wave tile 64 x 64 (2 x 2 accumulators of v_mfma_f32_32x32x16_bf16, computed
// TRANSPOSED — pixels as matrix rows — so that the epilogue writes 16-byte runs of pixels); the same products and the same
// accumulation order per output as conv2d_bf16x3_kernel => bit-identical results for identical operands.
// Replaces the same reference call sites as conv2d_bf16x3.hip (F.conv2d inside modulated_conv2d, networks_stylegan2.py:34-91).
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "up_tiles.h"

int conv16_splitk_epilogue_launch(const float* partial, float* y, int ksplit, int N, int O, int OH, int OW, int64_t ybs, int64_t yrs,
                                  const n3d_epilogue& epi, hipStream_t stream);      // conv2d_bf16x3.hip

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

struct ConvPsParams {
    const bf16x8* x; const bf16x8* wt16; float* y;
    float* partial;              // ksplit > 1: raw partial sums [ksplit][N][O][H][W] (reduced + epilogue by conv16_splitk_epilogue_launch)
    int N, I, O, OP64, H, W;
    int tiles_x, tiles_y, tiles_m;
    int ksplit, kc_per_split;    // split-K over extra workgroups: 16-channel chunks per split (layers whose tiles alone leave CUs idle)
    int64_t xbs;                 // 16-byte units between consecutive samples of x (= 2 * I/8 * H * W for a dense tensor)
    int64_t wbs;                 // 16-byte units between consecutive samples' weight tiles (0 = shared by the batch)
    int64_t ybs, yrs;            // floats
    int dbg;                     // N3D_CONV_DBG ablation bits (tuning only): 1 skip stores, 2 skip MFMA, 4 skip the DMA of chunks > 0,
                                 // 8 no per-chunk barrier / vmcnt wait (wrong results; timing only)
    n3d_epilogue epi;
    // fused toRGB of a LAST block (n3d_conv2d_desc.rgb_*): the activated tile is multiplied with the RC x O toRGB weights (times the sample's
    // toRGB styles) in the epilogue and only the partial colours of this workgroup's 64 channels leave the chip; y may then be NULL
    const float* rgb_weight; const float* rgb_style; float* rgb_partial; int rgb_channels; int64_t rgb_style_stride;
    // ... and, when the layer has a second reader (the next block's transposed convolution), its operand image: the activated tile times
    // side_style [N][O] in the dense split8 layout — exactly n3d_split8_from_nchw(y, side_style) — written from the same LDS stage
    bf16x8* side; const float* side_style; int64_t side_style_stride;
};

constexpr int PS_RGB32_MAX = 32;                                                       // ... on the matrix cores (RGB == 2)
constexpr int PS_RGB_MAX = 4, PS_RGB_PITCH = 16 * 32 + 4;                               // fused toRGB: colours, floats per staged channel row
constexpr int PS_BM = 64, PS_TH = 16, PS_TW = 32, PS_TAPS = 9;
constexpr int PS_PH = PS_TH + 2, PS_PW = PS_TW + 2, PS_PPIX = PS_PH * PS_PW;          // 18 x 34 = 612 patch pixels
constexpr int PS_BCH = (PS_PPIX + 63) / 64, PS_BPAD = PS_BCH * 64;                     // 10 DMA pieces = 640 slots per (hi|lo, half)
constexpr int PS_A_SZ = PS_TAPS * 2 * PS_BM;                                           // 16-byte slots per (buffer, hi|lo): [tap][half][row]
constexpr int PS_B_SZ = 2 * PS_BPAD;                                                   //                                     [half][pixel]
constexpr int PS_A_PIECES = PS_TAPS * 2 * 2, PS_B_PIECES = 2 * 2 * PS_BCH, PS_PIECES = PS_A_PIECES + PS_B_PIECES;   // 36 + 40
// ONE LDS array (a second __shared__ object makes hipcc drain vmcnt before every fragment read of an LDS-DMA pipeline):
//   [buf 0: A_hi | A_lo | B_hi | B_lo][buf 1: ...]   then 2 x 64 floats of epilogue factors
constexpr int PS_BUF = 2 * PS_A_SZ + 2 * PS_B_SZ;                                      // 4864 slots = 77,824 B per buffer

// NBUF = 2: one workgroup per CU, the next chunk's DMA runs under this chunk's MFMAs (two 76 KB buffers).
// NBUF = 1: TWO workgroups per CU, one 76 KB buffer each — a workgroup loads, waits and multiplies in turn and the CU's other
//           workgroup fills the gaps: its MFMAs run while this one waits for its DMA or writes its tile.  With one workgroup per
//           CU every chunk's DMA wait, every barrier skew and the whole epilogue (a 128 KB tile written while every other CU
//           writes its own: ~7 us at the chip's ~4.7 TB/s of store bandwidth) leave the matrix pipe idle.
// RGB: 0 = plain layer; 1 = fused toRGB of at most PS_RGB_MAX colours on the VALU (the super-resolution's 3-colour layers); 2 = fused toRGB of up to
// 32 colours on the matrix cores (the backbones' 32-channel toRGB layers).  SIDE (with RGB): the split8 side output for the layer's second reader.
template <int NBUF, int RGB, bool SIDE = false>
__device__ __forceinline__ void conv2d_ps_bf16x3_body(const ConvPsParams& p, bf16x8* smem, int lb_in = -1) {
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // XCD-aware 1-D grid, M tile fastest (conv2d_bf16x3.hip): the O/64 workgroups reading one input patch share it in one L2
    int lb = lb_in;                                                       // (>= 0: the persistent form hands over the logical workgroup index)
    if (lb_in < 0) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int m0 = (lb % p.tiles_m) * PS_BM; lb /= p.tiles_m;
    const int ks = lb % p.ksplit; lb /= p.ksplit;
    const int tile_i = lb % (p.tiles_x * p.tiles_y), n = lb / (p.tiles_x * p.tiles_y);
    const int y0 = (tile_i / p.tiles_x) * PS_TH, x0 = (tile_i % p.tiles_x) * PS_TW;
    const int KC = p.I / 16, HW = p.H * p.W;
    const int kc0 = ks * p.kc_per_split, kc1 = min(KC, kc0 + p.kc_per_split);        // this workgroup's chunks

    // descriptors (range-checked: a lane offset beyond the buffer reads as zero -> the halo of the patch).  hi and lo planes of
    // one sample are contiguous, so ONE descriptor covers both and the plane is selected through the scalar offset.
    const int plane_bytes = (p.I / 8) * HW * 16;
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.wt16 + (int64_t)n * p.wbs), 0, PS_TAPS * KC * 4 * p.OP64 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.xbs), 0, 2 * plane_bytes, 0x00020000);

    // This wave's copy pieces, all constants hoisted out of the K loop (the chunk only adds a stride to the scalar offsets):
    //   weights: pieces pa = wn + 8 j (j < 5, pa < 36) = (tap, hi|lo, half) slabs of 64 rows x 16 B
    //   patch  : pieces q  = wn + 8 j (j < 5)          = (hi|lo, half, 64-pixel run c) of the 18 x 34 patch
    constexpr int NA = (PS_A_PIECES + 7) / 8, NB = PS_B_PIECES / 8;       // 5, 5
    static_assert(PS_B_PIECES % 8 == 0, "patch pieces must divide evenly over the 8 waves");
    int ldsA[NA], sofA[NA], ldsB[NB], sofB[NB], voffB[NB];
    const int voffA = (m0 + lane) * 16;                                   // 64 consecutive weight rows (OP64 is padded to 64)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int pa = wn + 8 * j, t = pa >> 2, hl = (pa >> 1) & 1, hf = pa & 1;
        ldsA[j] = hl * PS_A_SZ + (t * 2 + hf) * PS_BM;
        sofA[j] = ((t * KC) * 4 + hl * 2 + hf) * p.OP64 * 16;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int q = wn + 8 * j, hl = q / (2 * PS_BCH), hf = (q / PS_BCH) & 1, c = q % PS_BCH;
        ldsB[j] = 2 * PS_A_SZ + hl * PS_B_SZ + hf * PS_BPAD + c * 64;
        sofB[j] = hl * plane_bytes + hf * HW * 16;
        const int pp = c * 64 + lane;                                     // patch pixel of this lane
        const int iy = y0 - 1 + pp / PS_PW, ix = x0 - 1 + pp % PS_PW;
        const bool ok = pp < PS_PPIX && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        voffB[j] = ok ? (iy * p.W + ix) * 16 : (int)0x80000000;
    }
    const int strideA = 4 * p.OP64 * 16, strideB = 2 * HW * 16;           // scalar-offset step per 16-channel chunk
    auto copy_chunk = [&](int kc, int buf) {
        bf16x8* base = smem + buf * PS_BUF;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (j < NA - 1 || wn + 8 * j < PS_A_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_void*)(base + ldsA[j]), 16, voffA, sofA[j] + kc * strideA, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x, (lds_void*)(base + ldsB[j]), 16, voffB[j], sofB[j] + kc * strideB, 0, 0);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const int a_frag = half * PS_BM + l31;                                // + tap*2*BM + mt*32
    const int b_frag0 = half * PS_BPAD + (wn * 2) * PS_PW + l31;          // + ky*PW + kx (+ PW for the wave's second row)
    auto mfma_block = [&](int buf) {
        const bf16x8* A_hi = smem + buf * PS_BUF, *A_lo = A_hi + PS_A_SZ, *B_hi = A_hi + 2 * PS_A_SZ, *B_lo = B_hi + PS_B_SZ;
        __builtin_amdgcn_s_setprio(1);
        bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
        auto fetch = [&](int t, int s) {
            const int boff = (t / 3) * PS_PW + (t % 3);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) { ah[s][mt] = A_hi[t * 2 * PS_BM + a_frag + mt * 32]; al[s][mt] = A_lo[t * 2 * PS_BM + a_frag + mt * 32]; }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) { bh[s][nt] = B_hi[b_frag0 + nt * PS_PW + boff]; bl[s][nt] = B_lo[b_frag0 + nt * PS_PW + boff]; }
        };
        fetch(0, 0);
#pragma unroll
        for (int t = 0; t < PS_TAPS; ++t) {
            const int s = t & 1;
            if (t + 1 < PS_TAPS) fetch(t + 1, s ^ 1);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    // pixels are the matrix ROWS here (first operand), channels the columns: the accumulator then holds, per lane, ONE
                    // channel and 16 pixels in 4 runs of 4 consecutive x -> the epilogue stores 16 bytes at a time (see below)
                    if constexpr (RGB == 2) {   // weights as rows: a lane then holds 16 CHANNELS of one pixel — the epilogue contraction's operand (below)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s][mt], bh[s][nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][mt], bl[s][nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][mt], bh[s][nt], acc[mt][nt], 0, 0, 0);
                    } else {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[s][nt], al[s][mt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[s][nt], ah[s][mt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[s][nt], ah[s][mt], acc[mt][nt], 0, 0, 0);
                    }
                }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // per-channel epilogue factors -> LDS (read after the K loop; plain stores, no DMA involved)
    float* s_rs = reinterpret_cast<float*>(smem + NBUF * PS_BUF), *s_bs = s_rs + PS_BM;
    const n3d_epilogue& E = p.epi;
    if (tid < PS_BM) {
        const int o = min(m0 + tid, p.O - 1);
        s_rs[tid] = E.const_scale * (E.row_scale ? E.row_scale[(int64_t)n * (E.row_scale_stride ? E.row_scale_stride : p.O) + o] : 1.f);
        s_bs[tid] = E.bias ? E.bias[o] : 0.f;
    }
    float* s_cw = s_bs + PS_BM;                                           // RGB: [colour][64 channels] toRGB weight x style of this sample
    if (RGB == 1 && tid < PS_RGB_MAX * PS_BM) {
        const int j = tid / PS_BM, o = m0 + tid % PS_BM;
        s_cw[tid] = (j < p.rgb_channels && o < p.O) ? p.rgb_weight[(int64_t)j * p.O + o] * p.rgb_style[(int64_t)n * p.rgb_style_stride + o] : 0.f;
    }
    float* s_sd = s_cw + PS_RGB_MAX * PS_BM;                              // RGB + side output: the next layer's styles of this tile's channels
    if (RGB && SIDE && tid < PS_BM) s_sd[tid] = (p.side && m0 + tid < p.O) ? p.side_style[(int64_t)n * p.side_style_stride + m0 + tid] : 0.f;

    if (NBUF == 1) {
        __builtin_amdgcn_s_barrier();                                     // (the epilogue factors above are plain LDS stores)
        for (int kc = kc0; kc < kc1; ++kc) {
            if (!(p.dbg & 4) || kc == kc0) copy_chunk(kc, 0);
            __builtin_amdgcn_s_waitcnt(0x0f70);                           // vmcnt(0): this wave's pieces are in LDS ...
            __builtin_amdgcn_s_barrier();                                 // ... and after the barrier everybody's are
            if (!(p.dbg & 2)) mfma_block(0);
            __builtin_amdgcn_s_barrier();                                 // every wave has read its fragments: the buffer may be refilled
        }
        if (p.dbg & 1) { if (acc[0][0][0] == 123.456f) p.y[0] = 1.f; return; }
    } else {
    copy_chunk(kc0, kc0 & 1);
    __builtin_amdgcn_s_waitcnt(0x0f70);                                   // vmcnt(0): this wave's pieces of the first chunk are in LDS ...
    __builtin_amdgcn_s_barrier();                                         // ... and after the barrier everybody's are
    if (p.dbg == 0) {
        for (int kc = kc0; kc < kc1; ++kc) {
            if (kc + 1 < kc1) copy_chunk(kc + 1, (kc + 1) & 1);           // the other buffer's last readers passed the previous barrier
            mfma_block(kc & 1);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __builtin_amdgcn_s_barrier();
        }
    } else {                                                              // ablations (N3D_CONV_DBG): which part of the loop costs what
        for (int kc = kc0; kc < kc1; ++kc) {
            if (kc + 1 < kc1 && !(p.dbg & 4)) copy_chunk(kc + 1, (kc + 1) & 1);
            if (!(p.dbg & 2)) mfma_block(kc & 1);
            if (!(p.dbg & 8)) { __builtin_amdgcn_s_waitcnt(0x0f70); __builtin_amdgcn_s_barrier(); }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        if (p.dbg & 1) { if (acc[0][0][0] == 123.456f) p.y[0] = 1.f; return; }
    }
    }

    // epilogue.  C/D layout with the operands as above: col = lane&31 = CHANNEL of the 32-channel group mt, row = (r&3) + 8*(r>>2)
    // + 4*(lane>>5) = PIXEL of the wave's row nt.  A lane's 16 values per accumulator are 4 runs (g = r>>2) of 4 consecutive
    // pixels x0 + 8g + 4*half + (0..3): one 16-byte store each — 16 stores per lane instead of 64 four-byte ones (the epilogue of
    // the channel-per-register form was 10-17 % of the kernel, store-issue bound: tools/conv_ps_abl.py), and the per-channel
    // factors are two registers per lane instead of 32.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    if (RGB != 2 && p.ksplit > 1) {                                       // split-K: raw partial sums, dense [ks][n][o][H][W]; the reduce pass applies the epilogue
        float* part = p.partial + (((int64_t)ks * p.N + n) * p.O) * (int64_t)p.H * p.W;
        const bool vec4 = (p.W & 3) == 0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int o = m0 + mt * 32 + l31;
            if (o >= p.O) continue;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int oy = y0 + wn * 2 + nt;
                if (oy >= p.H) continue;
                float* drow = part + ((int64_t)o * p.H + oy) * p.W;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ox = x0 + 8 * g + 4 * half;
                    if (ox >= p.W) continue;
                    if (vec4) *reinterpret_cast<f32x4*>(drow + ox) = f32x4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                    else
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (ox + k < p.W) drow[ox + k] = acc[mt][nt][4 * g + k];
                }
            }
        }
        return;
    }
    const float nstr = E.noise ? E.noise_strength[0] : 0.f;
    const bool lrelu = E.act == N3D_ACT_LRELU;
    const float alpha_eff = lrelu ? E.alpha : 1.f, clamp_eff = E.clamp >= 0.f ? E.clamp : INFINITY;
    if constexpr (RGB == 2) {
        // Fused toRGB with up to 32 colours (round 5: the 32-channel toRGB layers of the texture / mouth / blending networks,
        // tat/networks_stylegan2.py:575-584) as an EPILOGUE CONTRACTION on the matrix cores, straight from the accumulators.  This variant multiplies
        // with the weights as matrix ROWS (mfma_block), so a lane's 16 values per accumulator are 4 runs of 4 consecutive CHANNELS of ONE pixel:
        //     channel = 32 mt + (r & 3) + 8 (r >> 2) + 4 half,    pixel = (tile row 2 wn + nt, column lane & 31).
        // Eight of them (r = 8 s .. 8 s + 7) are, after the layer epilogue and the hi / lo split, a first-operand fragment of
        // v_mfma_f32_32x32x16_bf16 (pixels as rows) — the K index of that instruction is just a NAME for "which channel": the toRGB weights times the
        // sample's toRGB styles are laid out in LDS in the same permuted order, so no data moves between lanes and nothing is staged:
        //     colour[p][j] += x_hi w_lo + x_lo w_hi + x_hi w_hi   (float32 accumulation: the arithmetic of the separate 1x1 split-bf16 kernel),
        // 24 MFMAs per wave — a quarter of ONE 16-channel chunk of the K loop.  The result has lane = colour and 4 runs of 4 consecutive pixels:
        // 16-byte stores of the partial colour image of this workgroup's 64 channels; n3d_rgb_combine adds the O/64 partial images in index order.
        // SIDE: a run of 4 channels times the next layer's styles is half a split8 unit (n3d_split8_from_nchw's arithmetic): 8-byte stores, the
        // lanes l and l + 32 complete each 16-byte unit.  The feature map itself is never written (y must be NULL).
        bf16x8* s_bw = smem;                                              // [k step 4 = (mt, s)][hi|lo][half][32 colours] = 8 KB of the free chunk buffers
        int lane_e;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int tid_e = wn * 64 + lane_e, l31_e = lane_e & 31, half_e = lane_e >> 5;
        __syncthreads();                                                  // every wave is past its last fragment read: the chunk buffers are free
        {
            const int colour = tid_e & 31, grp = tid_e >> 5;              // grp = (mt, s, half, element quad): 4 consecutive channels
            const int gmt = grp >> 3, gs = (grp >> 2) & 1, gh = (grp >> 1) & 1, gq = grp & 1;
            const int c0 = gmt * 32 + 8 * (2 * gs + gq) + 4 * gh;         // fragment elements 4 gq .. 4 gq + 3 of (k step (gmt, gs), half gh)
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 wh, wl;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = m0 + c0 + k;
                const float wv = (colour < p.rgb_channels && o < p.O) ? p.rgb_weight[(int64_t)colour * p.O + o] * p.rgb_style[(int64_t)n * p.rgb_style_stride + o] : 0.f;
                const __bf16 h = (__bf16)wv;
                wh[k] = h;
                wl[k] = (__bf16)(wv - (float)h);
            }
            const int slot = (gmt * 2 + gs) * 128 + gh * 32 + colour;     // + 64 for lo
            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(s_bw + slot) + gq * 4) = wh;
            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(s_bw + slot + 64) + gq * 4) = wl;
        }
        __syncthreads();
        const __amdgpu_buffer_rsrc_t r_side = __builtin_amdgcn_make_buffer_rsrc((void*)(p.side + (int64_t)n * 2 * (p.O / 8) * HW), 0, (SIDE && p.side) ? 2 * (p.O / 8) * HW * 16 : 0, 0x00020000);
        const int ox_e = x0 + l31_e;
        f32x16 racc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) racc[nt][r] = 0.f;
            const int oy = y0 + wn * 2 + nt;
            const float nz = E.noise ? E.noise[(int64_t)min(oy, p.H - 1) * p.W + min(ox_e, p.W - 1)] * nstr : 0.f;
            const bool px_ok = oy < p.H && ox_e < p.W;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float v[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 rs4 = *reinterpret_cast<const f32x4*>(s_rs + mt * 32 + 8 * g + 4 * half_e), bs4 = *reinterpret_cast<const f32x4*>(s_bs + mt * 32 + 8 * g + 4 * half_e);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float t = acc[mt][nt][4 * g + q] * rs4[q] + nz + bs4[q];
                        t = fmaxf(t, t * alpha_eff) * E.gain;
                        v[4 * g + q] = fminf(fmaxf(t, -clamp_eff), clamp_eff);
                    }
                    if constexpr (SIDE) {
                        const f32x4 sd4 = *reinterpret_cast<const f32x4*>(s_sd + mt * 32 + 8 * g + 4 * half_e);
                        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                        bf16x4 sh, sl;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float t = v[4 * g + q] * sd4[q];             // n3d_split8_from_nchw's arithmetic
                            const __bf16 th = (__bf16)t;
                            sh[q] = th;
                            sl[q] = (__bf16)(t - (float)th);
                        }
                        if (p.side != nullptr) {
                            typedef int i32x2 __attribute__((ext_vector_type(2)));
                            const int unit = (m0 + mt * 32) / 8 + g;
                            // (out-of-image pixels and channels beyond O get an offset beyond the descriptor's range: dropped by the hardware)
                            const int voff = (px_ok && unit * 8 < p.O) ? (unit * HW + oy * p.W + ox_e) * 16 + 8 * half_e : (int)0x80000000;
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, sh), r_side, voff, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, sl), r_side, voff, (p.O / 8) * HW * 16, 0);
                        }
                    }
                }
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    bf16x8 xh, xl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const __bf16 h = (__bf16)v[8 * st + e];
                        xh[e] = h;
                        xl[e] = (__bf16)(v[8 * st + e] - (float)h);
                    }
                    const bf16x8 bw_hi = s_bw[(mt * 2 + st) * 128 + half_e * 32 + l31_e], bw_lo = s_bw[(mt * 2 + st) * 128 + 64 + half_e * 32 + l31_e];
                    racc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bw_lo, racc[nt], 0, 0, 0);
                    racc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bw_hi, racc[nt], 0, 0, 0);
                    racc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bw_hi, racc[nt], 0, 0, 0);
                }
            }
        }
        // partial colours: lane = colour, 4 runs of 4 consecutive pixels per tile row
        if (l31_e < p.rgb_channels) {
            float* dst = p.rgb_partial + (((int64_t)n * p.tiles_m + m0 / PS_BM) * p.rgb_channels + l31_e) * (int64_t)p.H * p.W;
            const bool vec4 = (p.W & 3) == 0;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int oy = y0 + wn * 2 + nt;
                if (oy >= p.H) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ox = x0 + 8 * g + 4 * half_e;
                    if (ox >= p.W) continue;
                    if (vec4) *reinterpret_cast<f32x4*>(dst + (int64_t)oy * p.W + ox) = f32x4{racc[nt][4 * g], racc[nt][4 * g + 1], racc[nt][4 * g + 2], racc[nt][4 * g + 3]};
                    else
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (ox + k < p.W) dst[(int64_t)oy * p.W + ox + k] = racc[nt][4 * g + k];
                }
            }
        }
    } else if constexpr (RGB == 1) {
        // Fused toRGB.  The layer epilogue runs on the accumulators exactly as below; instead of going to HBM the activated tile is staged in
        // LDS 32 channels at a time ([channel][16 x 32 pixels], the chunk buffers are free now) and every thread sums ITS pixel over the
        // channels in float32: colour j += x[c] * (w[j][c] * style[n][c]).  The two 64-channel workgroups of a 128-channel layer write separate
        // partial images; n3d_rgb_combine adds them in a fixed order (bitwise reproducible) and applies toRGB's own epilogue.
        float* stage = reinterpret_cast<float*>(smem);
        float col[PS_RGB_MAX] = {0.f, 0.f, 0.f, 0.f};
        // (lane / thread index re-derived here so that nothing extra stays live across the K loop: the loop sits at the 128-VGPR limit)
        int lane_e;                                                       // (volatile asm: not hoisted above the loop as a loop invariant)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int tid_e = wn * 64 + lane_e, l31_e = lane_e & 31, half_e = lane_e >> 5;
        __syncthreads();                                                  // every wave is past its last fragment read; s_cw is visible
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float rs = s_rs[mt * 32 + l31_e], bs = s_bs[mt * 32 + l31_e];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int oy = min(y0 + wn * 2 + nt, p.H - 1);
                const float* nrow = E.noise ? E.noise + (int64_t)oy * p.W : nullptr;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ox = x0 + 8 * g + 4 * half_e;
                    f32x4 out;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float t = acc[mt][nt][4 * g + k] * rs + (nrow ? nrow[min(ox + k, p.W - 1)] * nstr : 0.f) + bs;
                        t = fmaxf(t, t * alpha_eff) * E.gain;
                        out[k] = fminf(fmaxf(t, -clamp_eff), clamp_eff);
                    }
                    *reinterpret_cast<f32x4*>(stage + l31_e * PS_RGB_PITCH + (wn * 2 + nt) * 32 + 8 * g + 4 * half_e) = out;
                }
            }
            __syncthreads();
            const int oy_s = y0 + (tid_e >> 5), ox_s = x0 + (tid_e & 31);
            const bool side_px = SIDE && p.side != nullptr && oy_s < p.H && ox_s < p.W;
            // (buffer stores: one VGPR of address per lane — the pixel — and the (plane, unit) part in the scalar offset; pixels outside the image
            // get an offset beyond the descriptor's range: dropped by the hardware)
            const __amdgpu_buffer_rsrc_t r_side = __builtin_amdgcn_make_buffer_rsrc((void*)(p.side + (int64_t)n * 2 * (p.O / 8) * HW), 0, (SIDE && p.side) ? 2 * (p.O / 8) * HW * 16 : 0, 0x00020000);
            const int voff_side = side_px ? (oy_s * p.W + ox_s) * 16 : (int)0x80000000;
#pragma unroll 1
            for (int u = 0; u < 4; ++u) {                                 // 8-channel units of this 32-channel group (not unrolled: register budget)
                bf16x8 hi, lo;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    const int c = u * 8 + cc;
                    const float xv = stage[c * PS_RGB_PITCH + tid_e];
#pragma unroll
                    for (int j = 0; j < PS_RGB_MAX; ++j) col[j] = fmaf(xv, s_cw[j * PS_BM + mt * 32 + c], col[j]);
                    if constexpr (SIDE) {
                        const float t = xv * s_sd[mt * 32 + c];            // n3d_split8_from_nchw's arithmetic
                        const __bf16 h = (__bf16)t;
                        hi[cc] = h;
                        lo[cc] = (__bf16)(t - (float)h);
                    }
                }
                const int unit = (m0 + mt * 32) / 8 + u;
                if (SIDE && p.side != nullptr && unit * 8 < p.O) {
                    typedef int i32x4 __attribute__((ext_vector_type(4)));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, hi), r_side, voff_side, unit * HW * 16, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, lo), r_side, voff_side, (p.O / 8 + unit) * HW * 16, 0);
                }
            }
            __syncthreads();
        }
        const int oy = y0 + (tid_e >> 5), ox = x0 + (tid_e & 31);
        if (oy < p.H && ox < p.W) {
            float* dst = p.rgb_partial + (((int64_t)n * p.tiles_m + m0 / PS_BM) * p.rgb_channels) * (int64_t)p.H * p.W + (int64_t)oy * p.W + ox;
#pragma unroll
            for (int j = 0; j < PS_RGB_MAX; ++j)
                if (j < p.rgb_channels) dst[(int64_t)j * p.H * p.W] = col[j];
        }
    }
    const bool store_y = RGB != 2 && (!RGB || p.y != nullptr);      // (RGB == 2: other accumulator layout, y is NULL by contract)          // (an early `return` here makes hipcc wrap every LDS-DMA copy of the K loop into a waterfall loop)
    const int64_t plane = (int64_t)p.H * p.W, yplane = (int64_t)p.H * p.yrs;
    const bool vec = ((p.W | p.yrs | p.ybs) & 3) == 0 && ((uintptr_t)p.y & 15) == 0 &&
                     (!E.residual || ((E.residual_batch_stride & 3) == 0 && ((uintptr_t)E.residual & 15) == 0)) && (!E.noise || ((uintptr_t)E.noise & 15) == 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int o = m0 + mt * 32 + l31;
        if (o >= p.O || !store_y) continue;
        const float rs = s_rs[mt * 32 + l31], bs = s_bs[mt * 32 + l31];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int oy = y0 + wn * 2 + nt;
            if (oy >= p.H) continue;
            float* drow = p.y + (int64_t)n * p.ybs + (int64_t)o * yplane + (int64_t)oy * p.yrs;
            const float* rrow = E.residual ? E.residual + (int64_t)n * E.residual_batch_stride + (int64_t)o * plane + (int64_t)oy * p.W : nullptr;
            const float* nrow = E.noise ? E.noise + (int64_t)oy * p.W : nullptr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ox = x0 + 8 * g + 4 * half;
                if (ox >= p.W) continue;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = acc[mt][nt][4 * g + k];
                if (vec) {                                                 // W % 4 == 0: the whole run is inside the image
                    f32x4 nz = {0.f, 0.f, 0.f, 0.f};
                    if (nrow) nz = *reinterpret_cast<const f32x4*>(nrow + ox) * nstr;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float t = v[k] * rs + nz[k] + bs;
                        t = fmaxf(t, t * alpha_eff) * E.gain;               // leaky ReLU (0 <= alpha <= 1) or linear
                        v[k] = n3d_round16(fminf(fmaxf(t, -clamp_eff), clamp_eff), E.round_f16);
                    }
                    f32x4 out = {v[0], v[1], v[2], v[3]};
                    if (rrow) out += *reinterpret_cast<const f32x4*>(rrow + ox);
                    *reinterpret_cast<f32x4*>(drow + ox) = out;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (ox + k >= p.W) break;
                        float t = v[k] * rs + (nrow ? nrow[ox + k] * nstr : 0.f) + bs;
                        t = fmaxf(t, t * alpha_eff) * E.gain;
                        t = n3d_round16(fminf(fmaxf(t, -clamp_eff), clamp_eff), E.round_f16);
                        if (rrow) t += rrow[ox + k];
                        drow[ox + k] = t;
                    }
                }
            }
        }
    }
}

// (concrete kernels around the body template: hipcc's host pass can drop the launch stub of a __global__ template, see conv2d_f16.hip)
constexpr int ps_smem_slots(int nbuf, bool rgb) { return nbuf * PS_BUF + (2 + (rgb ? PS_RGB_MAX + 1 : 0)) * PS_BM * 4 / 16; }
__global__ __launch_bounds__(512, 4) void conv2d_ps1_bf16x3_kernel(ConvPsParams p) {         // one buffer, two workgroups per CU
    __shared__ bf16x8 smem[ps_smem_slots(1, false)];
    conv2d_ps_bf16x3_body<1, 0>(p, smem);
}
// Persistent form (tuning builds: N3D_PS_PERSIST=1): 2 workgroups per CU walk the logical workgroups of their XCD's share in steps of the XCD's
// resident workgroups — a tile's stores drain under the next tile's first DMA instead of holding the CU's slot until they are acknowledged.
__device__ __forceinline__ void ps_persistent_range(int total, int& first, int& count, int& step) {
    const int q = total >> 3, r = total & 7, xcd = blockIdx.x & 7;
    first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    count = q + (xcd < r ? 1 : 0) - (blockIdx.x >> 3);                    // logical indices left in this XCD's share from `first` on
    step = gridDim.x >> 3;                                                // (the grid is a multiple of 8)
}
__global__ __launch_bounds__(512, 4) void conv2d_ps1p_bf16x3_kernel(ConvPsParams p, int total) {
    __shared__ bf16x8 smem[ps_smem_slots(1, false)];
    int first, count, step;
    ps_persistent_range(total, first, count, step);
    for (int k = 0; k < count; k += step) {
        conv2d_ps_bf16x3_body<1, 0>(p, smem, first + k);
        __syncthreads();                                                  // the epilogue's LDS factors are rewritten by the next tile
    }
}
// DYNAMIC persistent form (round 6; tuning builds: N3D_PS_PERSIST=2): the resident workgroups PULL tiles from a queue instead of
// walking a static share — blockIdx & 7 (the XCD a workgroup is observed to run on: a speed assumption only) selects one of eight queues, each covering the
// contiguous eighth of the logical tile order the launch-per-tile form would have given that XCD (same L2 locality), one relaxed agent-scope fetch_add per tile
// (MI355X_MICROARCH.md "dequeue": 0.3-1.3 us, shared per XCD).  A workgroup that finishes early takes the next tile at once: no lockstep, no static imbalance;
// its stores drain under the next tile's first DMA.  q[0..7] = the queues' heads, q[8] = workgroups that have left: the last one re-arms all nine words
// (the caller's zeroed per-stream pool, n3d_conv2d_desc.tickets).
__device__ __forceinline__ int ps_dequeue(unsigned* q, int xcd, int* s_slot) {
    if (threadIdx.x == 0) *s_slot = (int)__hip_atomic_fetch_add(q + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int t = __builtin_amdgcn_readfirstlane(*s_slot);
    __syncthreads();                                                      // (the slot is rewritten by the next dequeue)
    return t;
}
__device__ __forceinline__ void ps_queue_leave(unsigned* q) {
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(q + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) __hip_atomic_store(q + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void ps_queue_share(int total, int& first, int& count) {
    const int q8 = total >> 3, r = total & 7, xcd = blockIdx.x & 7;
    first = xcd < r ? xcd * (q8 + 1) : r * (q8 + 1) + (xcd - r) * q8;
    count = q8 + (xcd < r ? 1 : 0);
}
__global__ __launch_bounds__(512, 4) void conv2d_ps1d_bf16x3_kernel(ConvPsParams p, int total, unsigned* q) {
    __shared__ bf16x8 smem[ps_smem_slots(1, false) + 1];
    int* s_slot = reinterpret_cast<int*>(smem + ps_smem_slots(1, false));
    int first, count;
    ps_queue_share(total, first, count);
    for (;;) {
        const int t = ps_dequeue(q, blockIdx.x & 7, s_slot);
        if (t >= count) break;
        conv2d_ps_bf16x3_body<1, 0>(p, smem, first + t);
    }
    ps_queue_leave(q);
}
__global__ __launch_bounds__(512, 2) void conv2d_ps2_bf16x3_kernel(ConvPsParams p) {         // two buffers, one workgroup per CU
    __shared__ bf16x8 smem[ps_smem_slots(2, false)];
    conv2d_ps_bf16x3_body<2, 0>(p, smem);
}
__global__ __launch_bounds__(512, 4) void conv2d_ps1_rgb_bf16x3_kernel(ConvPsParams p) {     // + fused toRGB (a network's last layer)
    __shared__ bf16x8 smem[ps_smem_slots(1, true)];
    conv2d_ps_bf16x3_body<1, 1>(p, smem);
}
__global__ __launch_bounds__(512, 2) void conv2d_ps2_rgb_bf16x3_kernel(ConvPsParams p) {
    __shared__ bf16x8 smem[ps_smem_slots(2, true)];
    conv2d_ps_bf16x3_body<2, 1>(p, smem);
}
// + fused toRGB of up to 32 colours on the matrix cores (the backbones' toRGB layers), without / with the split8 side output.  One workgroup per CU
// (the epilogue holds the 64 main and 32 colour accumulators: beyond the 128-register cap of the two-per-CU form)
__global__ __launch_bounds__(512, 2) void conv2d_ps2_rgb32_bf16x3_kernel(ConvPsParams p) {
    __shared__ bf16x8 smem[ps_smem_slots(2, true)];
    conv2d_ps_bf16x3_body<2, 2>(p, smem);
}
__global__ __launch_bounds__(512, 2) void conv2d_ps2_rgb32s_bf16x3_kernel(ConvPsParams p) {
    __shared__ bf16x8 smem[ps_smem_slots(2, true)];
    conv2d_ps_bf16x3_body<2, 2, true>(p, smem);
}
// + the split8 side output for the layer's second reader.  Two buffers / one workgroup per CU only: the epilogue needs ~160 VGPRs, and under the
// 128-register cap of the two-workgroups-per-CU form hipcc spills ACCUMULATORS inside the K loop
__global__ __launch_bounds__(512, 2) void conv2d_ps2_rgbs_bf16x3_kernel(ConvPsParams p) {
    __shared__ bf16x8 smem[ps_smem_slots(2, true)];
    conv2d_ps_bf16x3_body<2, 1, true>(p, smem);
}

// Layers this kernel takes (the host asks before it lets a producer write split8): 3x3 stride 1, I % 16 == 0, images of at
// least 16 x 32 whose 8-wave tiles — times a split-K factor of at most I / 64 (four 16-channel chunks per workgroup at least) —
// cover the chip, linear / leaky-ReLU (0 <= alpha <= 1) epilogue.  n3d_conv2d_split8_ksplit: the factor the launch will use (0 =
// not this kernel's layer, 1 = no split; > 1: the caller provides the partial-sum workspace of ksplit * N * O * H * W floats).
extern "C" int n3d_conv2d_split8_ksplit(int N, int I, int O, int H, int W) {
    if (I % 16 != 0 || I < 16 || H < 16 || W < 32 || N < 1 || O < 1) return 0;
    const int64_t blocks = (int64_t)cdiv(W, PS_TW) * cdiv(H, PS_TH) * cdiv(O, PS_BM) * N;
    if ((int64_t)(I / 8) * H * W * 16 >= (1ll << 31)) return 0;           // 32-bit buffer offsets per plane
    { const int force = n3d_tune("N3D_PS_KS", 0); if (force > 0 && (I / 16) % force == 0) return force; }      // (tuning builds: tools/ps_b1_sweep.py)
    if (blocks >= 256) return 1;
    static const int on = n3d_tune("N3D_PS_SPLITK", 1);
    int ks = 1;
    while (blocks * ks < 256 && I / (ks * 2) >= 64 && ks < 16) ks *= 2;
    return (on && ks > 1 && blocks * ks >= 192) ? ks : 0;
}
extern "C" int n3d_conv2d_split8_eligible(int N, int I, int O, int H, int W) { return n3d_conv2d_split8_ksplit(N, I, O, H, W) > 0 ? 1 : 0; }

int conv2d_ps_bf16x3_launch(const n3d_conv2d_desc* d, hipStream_t stream) {
    N3D_CHECK(d->ksize == 3 && d->mode == 0, "conv2d_bf16x3: a split8 input is taken by the 3x3 stride-1 kernel only");
    N3D_CHECK(d->style == nullptr, "conv2d_bf16x3: a split8 input carries its modulation already (style must be NULL)");
    const int ksplit = n3d_conv2d_split8_ksplit(d->N, d->I, d->O, d->H, d->W);
    N3D_CHECK(ksplit > 0, "conv2d_bf16x3: shape not eligible for the split8 kernel (n3d_conv2d_split8_eligible)");
    N3D_CHECK(ksplit == 1 || d->workspace != nullptr, "conv2d_bf16x3: this split8 layer runs with split-K %d (n3d_conv2d_split8_ksplit): workspace required", ksplit);
    const n3d_epilogue& E = d->epi;
    N3D_CHECK(E.act == N3D_ACT_LINEAR || (E.act == N3D_ACT_LRELU && E.alpha >= 0.f && E.alpha <= 1.f), "conv2d_bf16x3 (split8): linear or leaky-ReLU epilogue only");
    N3D_CHECK(!E.residual_up_filter, "conv2d_bf16x3: residual_up_filter is only supported by the 1x1 kernel");
    N3D_CHECK(((uintptr_t)d->x & 15) == 0, "conv2d_bf16x3: split8 input must be 16-byte aligned");
    ConvPsParams p;
    p.x = (const bf16x8*)d->x; p.wt16 = (const bf16x8*)d->wt; p.y = d->y;
    p.N = d->N; p.I = d->I; p.O = d->O; p.OP64 = (d->O + 63) / 64 * 64; p.H = d->H; p.W = d->W;
    p.tiles_x = cdiv(d->W, PS_TW); p.tiles_y = cdiv(d->H, PS_TH); p.tiles_m = cdiv(d->O, PS_BM);
    p.kc_per_split = cdiv(d->I / 16, ksplit); p.ksplit = cdiv(d->I / 16, p.kc_per_split);
    p.partial = p.ksplit > 1 ? d->workspace : nullptr;
    p.xbs = d->x_batch_stride ? d->x_batch_stride / 4 : (int64_t)2 * (d->I / 8) * d->H * d->W;      // x_batch_stride counts fp32-sized elements
    N3D_CHECK((d->wt_batch_stride & 15) == 0, "conv2d_bf16x3: wt_batch_stride must be a multiple of 16 bytes");
    p.wbs = d->wt_batch_stride / 16;
    p.ybs = d->y_batch_stride; p.yrs = d->y_row_stride ? d->y_row_stride : d->W;
    N3D_CHECK(d->x_batch_stride % 4 == 0, "conv2d_bf16x3: split8 batch stride must be a multiple of 16 bytes");
    N3D_CHECK(p.yrs >= d->W, "conv2d_bf16x3: y_row_stride smaller than the output width");
    p.epi = d->epi;
    const bool rgb = d->rgb_partial != nullptr;
    if (rgb) {
        N3D_CHECK(d->rgb_weight && d->rgb_style && d->rgb_channels >= 1 && d->rgb_channels <= PS_RGB32_MAX, "conv2d_bf16x3: fused toRGB needs rgb_weight, rgb_style and 1..32 colours");
        N3D_CHECK(p.ksplit == 1 && !E.residual && !E.round_f16, "conv2d_bf16x3: fused toRGB on a layer without split-K, residual or float16 rounding");
    }
    N3D_CHECK(d->y != nullptr || rgb, "conv2d_bf16x3: y is NULL");
    N3D_CHECK(!(rgb && d->rgb_channels > PS_RGB_MAX) || d->y == nullptr, "conv2d_bf16x3: the fused toRGB with more than 4 colours does not write the feature map (y must be NULL)");
    N3D_CHECK(!d->side_split8 || (rgb && d->side_style && d->O % 8 == 0 && ((uintptr_t)d->side_split8 & 15) == 0 && (int64_t)2 * (d->O / 8) * d->H * d->W * 16 < (1ll << 31)),
              "conv2d_bf16x3 (split8 input): the split8 side output is an option of the fused toRGB (rgb_partial, side_style, O %% 8 == 0)");
    p.side = (bf16x8*)d->side_split8; p.side_style = d->side_style; p.side_style_stride = d->side_style_stride ? d->side_style_stride : d->O;
    p.rgb_weight = d->rgb_weight; p.rgb_style = d->rgb_style; p.rgb_partial = d->rgb_partial; p.rgb_channels = d->rgb_channels;
    p.rgb_style_stride = d->rgb_style_stride ? d->rgb_style_stride : d->O;
    p.dbg = n3d_tune("N3D_CONV_DBG", 0);                                  // tuning builds re-read it per launch: tools/conv_ps_abl.py flips it in-process
    const int64_t nblk = (int64_t)p.tiles_x * p.tiles_y * p.tiles_m * p.N * p.ksplit;
    N3D_CHECK(nblk < (1ll << 31), "conv2d_bf16x3: grid too large");
    const double flops = 2.0 * d->N * (double)d->O * d->I * 9 * (double)d->H * d->W;
    const double bytes = 4.0 * ((double)d->N * d->I * d->H * d->W + (d->y ? (double)d->N * d->O * d->H * d->W : 0.0) + (double)d->O * d->I * 9 +
                                (rgb ? (double)d->N * p.tiles_m * d->rgb_channels * d->H * d->W : 0.0) + (d->side_split8 ? (double)d->N * d->O * d->H * d->W : 0.0));
    N3dProfScope prof(N3D_K_CONV2D_BF16X3, stream, flops, bytes);
    // two workgroups per CU (single LDS buffer each) once the grid holds at least three per CU: with fewer, the in-workgroup
    // double buffering wins (measured, tools/conv_ps_abl.py: 64x64 x 512 channels = 256 workgroups: 156 us vs 201 us; 512
    // workgroups: equal; 1024+: 174 vs 182 us, 697 vs 735 us).  Starting every other batch of workgroups late to de-phase the
    // CUs' store bursts was measured too: no effect.
    { const int nbuf = n3d_tune("N3D_PS_NBUF", nblk >= 768 ? 1 : 2);
      if (rgb && d->rgb_channels > PS_RGB_MAX && d->side_split8) hipLaunchKernelGGL(conv2d_ps2_rgb32s_bf16x3_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
      else if (rgb && d->rgb_channels > PS_RGB_MAX) hipLaunchKernelGGL(conv2d_ps2_rgb32_bf16x3_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
      else if (rgb && d->side_split8) hipLaunchKernelGGL(conv2d_ps2_rgbs_bf16x3_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
      else if (rgb && nbuf == 2) hipLaunchKernelGGL(conv2d_ps2_rgb_bf16x3_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
      else if (rgb) hipLaunchKernelGGL(conv2d_ps1_rgb_bf16x3_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
      else if (nbuf == 2) hipLaunchKernelGGL(conv2d_ps2_bf16x3_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
      else if (n3d_tune("N3D_PS_PERSIST", 0) == 2 && p.ksplit == 1 && nblk > 512 && d->tickets && d->ticket_count >= 9) hipLaunchKernelGGL(conv2d_ps1d_bf16x3_kernel, dim3(512), dim3(512), 0, stream, p, (int)nblk, d->tickets);
      else if (n3d_tune("N3D_PS_PERSIST", 0) == 1 && p.ksplit == 1 && nblk > 512) hipLaunchKernelGGL(conv2d_ps1p_bf16x3_kernel, dim3(512), dim3(512), 0, stream, p, (int)nblk);
      else hipLaunchKernelGGL(conv2d_ps1_bf16x3_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p); }
    N3D_LAUNCH_CHECK();
    if (p.ksplit > 1) return conv16_splitk_epilogue_launch(p.partial, p.y, p.ksplit, p.N, p.O, p.H, p.W, p.ybs, p.yrs, p.epi, stream);
    return 0;
}

// The second half of the fused toRGB: sum the workgroups' partial colour images in index order and apply the toRGB layer's epilogue.
// blockIdx.y = (sample, colour) plane, blockIdx.x walks the plane's VEC-pixel vectors (VEC = 4: 16-byte loads of the M partial images and a 16-byte
// store; VEC = 1 for widths that are not a multiple of 4): no 64-bit division per element, bias / clamp are per-plane scalars.  With 32 colours
// (round 5: the backbones' toRGB layers) this kernel moves 100 MB per 256 x 256 block at batch 4 — it has to stream, not crawl.
template <int VEC>
__global__ __launch_bounds__(256) void rgb_combine_kernel(const float* __restrict__ partial, float* __restrict__ y, int N, int M, int C, int H, int W, n3d_epilogue E) {
    const int nc = blockIdx.y, n = nc / C, c = nc % C;
    const int plane = H * W, nvec = plane / VEC, WV = W / VEC;
    const float* p0 = partial + ((int64_t)n * M * C + c) * plane;
    const int64_t mstride = (int64_t)C * plane;
    float* yp = y + (int64_t)nc * plane;
    const float* rplane = E.residual ? E.residual + (int64_t)n * E.residual_batch_stride + (int64_t)c * (E.residual_up_filter ? (plane >> 2) : plane) : nullptr;
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
        vec_t s = *reinterpret_cast<const vec_t*>(p0 + (int64_t)i * VEC);
        for (int m = 1; m < M; ++m) s += *reinterpret_cast<const vec_t*>(p0 + m * mstride + (int64_t)i * VEC);
        const int oy = i / WV, ox = (i - oy * WV) * VEC;
        vec_t out;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float v = s[k];
            if (E.round_f16) {    // ToRGBLayer of a float16 block (n3d_torgb_h8's order): the float32 sum rounds to float16, bias_act on float16, float32 skip image
                const float t = (float)(_Float16)v + (E.bias ? (float)(_Float16)E.bias[c] : 0.f);
                const float cl = E.clamp >= 0.f ? E.clamp : INFINITY;
                v = (float)(_Float16)fminf(fmaxf(t, -cl), cl);
                if (rplane) {
                    if (E.residual_up_filter) v += n3d_up2_apply(n3d_up2_setup(E.residual_up_filter, oy, ox + k, H >> 1, W >> 1), rplane);
                    else v += rplane[(int64_t)i * VEC + k];
                }
            } else {
                v = n3d_apply_epilogue(v, E, n, c, C, oy, ox + k, H, W);
            }
            out[k] = v;
        }
        *reinterpret_cast<vec_t*>(yp + (int64_t)i * VEC) = out;
    }
}
extern "C" int n3d_rgb_combine(const float* partial, float* y, int N, int M, int C, int H, int W, const n3d_epilogue* epi, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && M >= 1 && C >= 1 && H >= 1 && W >= 1 && epi, "rgb_combine: bad arguments");
    if (N == 0) return 0;
    N3D_CHECK(partial && y, "rgb_combine: null tensor");
    N3D_CHECK(!epi->row_scale && !epi->noise, "rgb_combine: toRGB epilogue only (const scale, bias, activation, clamp, residual)");
    N3D_CHECK(!epi->round_f16 || (epi->act == N3D_ACT_LINEAR && epi->const_scale == 1.f && epi->gain == 1.f), "rgb_combine: a float16 block's toRGB has a linear epilogue");
    N3D_CHECK(!epi->residual_up_filter || (epi->residual && H % 2 == 0 && W % 2 == 0), "rgb_combine: residual_up_filter needs a residual and an even output size");
    const int64_t total = (int64_t)N * C * H * W;
    N3dProfScope prof(N3D_K_MISC, stream, 0.0, 4.0 * total * (M + 1));
    N3D_CHECK((int64_t)H * W < (1ll << 31) && (int64_t)N * C <= 65535, "rgb_combine: plane or plane count too large");
    const bool vec = (W & 3) == 0 && (((uintptr_t)partial | (uintptr_t)y) & 15) == 0;
    const int nvec = H * W / (vec ? 4 : 1);
    const dim3 grid((unsigned)std::min((nvec + 255) / 256, 64), (unsigned)(N * C));
    if (vec) hipLaunchKernelGGL(rgb_combine_kernel<4>, grid, dim3(256), 0, stream, partial, y, N, M, C, H, W, *epi);
    else hipLaunchKernelGGL(rgb_combine_kernel<1>, grid, dim3(256), 0, stream, partial, y, N, M, C, H, W, *epi);
    N3D_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------------------
// Transposed 3x3 stride-2 convolution (the up-sampling layers) on the same terms: split8 input staged by LDS-DMA, all four output
// phases from one staged patch exactly as conv2d_up_bf16x3_kernel (tap (ky,kx) feeds phase (ky==1, kx==1) from patch offset
// (ky==2 ? 0 : 1, kx==2 ? 0 : 1); flattened th x tw <= 256 positions over the 8 waves' lanes), channel-interleaved (c8) output
// for the FIR that follows (fir4_c8_split8_kernel).  The register-staged kernel spends, per 16-channel chunk, the SAME staging work as
// the stride-1 kernel for half the MFMAs (54 per wave) and waits on memory 56 % of its cycles (round 1); here staging is 7 DMA
// instructions per wave and chunk (NMT = 2), two LDS buffers per workgroup.

constexpr int UP_PPIX = 9 * 33;                                           // patch capacity: (th+1) x (tw+1) <= 297
constexpr int UP_BCH = (UP_PPIX + 63) / 64, UP_BPAD = UP_BCH * 64;        // 5 pieces = 320 slots per (hi|lo, half)
constexpr int UP_B_SZ = 2 * UP_BPAD;                                      // 640 slots per (hi|lo)
constexpr int UP_B_PIECES = 2 * 2 * UP_BCH;                               // 20

struct ConvUpPsParams {
    const bf16x8* x; const bf16x8* wt16; float* y;
    int N, I, O, OP64, H, W, OH, OW;
    UpTilePlan plan; int tiles_m;
    int64_t xbs, ybs, yrs;       // xbs: 16-byte units; ybs floats; yrs pixels (c8 row pitch)
    const float* row_scale; int64_t row_scale_stride; float const_scale;
    int round_f16;
    int y_nchw;                  // y is float32 NCHW (row pitch yrs floats) instead of c8: the few-position layers (4 x 4 .. 32 x 32 inputs), whose
                                 // FIR reads NCHW — 4-byte stores, irrelevant next to these layers' launch latency
    int thin_last;               // thin edge tiles at the end of the launch order (always, except N3D_UP_THIN_LAST=0 in tuning builds)
    int dbg;                     // N3D_CONV_DBG ablation bits (tuning only, wrong results): 1 skip stores, 4 skip the DMA of chunks > 0
};

// NMT = 32-channel groups per workgroup.  NMT = 2: 64 channels, one workgroup per CU (167 VGPRs: 128 accumulators).  NMT = 1: 32
// channels, 64 accumulators, 78 KB of LDS -> TWO workgroups per CU: the transposed kernel writes 4 output pixels per position
// (a 256 KB tile per 64 channels), and with one workgroup per CU the matrix pipe idles through every tile's stores (pipe busy
// 0.40 against 0.68 for the stride-1 kernel, profiles/r02_mfma_busy_pmc.json); the second workgroup multiplies meanwhile.
constexpr int up_ps_buf_slots(int nmt) { return 2 * (PS_TAPS * 2 * 32 * nmt) + 2 * UP_B_SZ; }      // per buffer, 16-byte slots
constexpr int up_ps_smem_slots(int nmt) { return 2 * up_ps_buf_slots(nmt) + 32 * nmt * 4 / 16; }

// (the body is a __device__ template under two plain kernels: the host pass emits no launch stub for a TEMPLATE kernel whose
// body uses the LDS-DMA builtin, and says nothing)
// NW = waves per workgroup, PG = 32-position groups per wave (NW * PG = 8: the tile is always 256 positions).  <1, 4, 2>: a wave
// covers 64 positions x 32 channels, so the weight fragments it reads serve twice the MFMAs (34 fragment reads per 54 MFMAs instead
// of 26 per 27: the <1, 8, 1> form keeps the LDS pipe ~96 % busy at the MFMA rate it reaches).
template <int NMT, int NW, int PG>
__device__ __forceinline__ void conv2d_up_ps_body(const ConvUpPsParams& p, bf16x8* smem, int lb_in = -1) {
    static_assert(NW * PG == 8, "256 positions per tile");
    constexpr int BM = 32 * NMT, A_SZ = PS_TAPS * 2 * BM;                  // 16-byte slots per (buffer, hi|lo): [tap][half][row]
    constexpr int BUF = up_ps_buf_slots(NMT);                             // NMT = 2: 57,344 B, NMT = 1: 38,912 B per buffer
    constexpr int A_PIECES = PS_TAPS * 2 * NMT;                            // 64-slot pieces: NMT = 2 (tap, hi|lo, half), NMT = 1 (tap, hi|lo)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int lb = lb_in;
    if (lb_in < 0) {
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

    const int plane_bytes = (p.I / 8) * HW * 16;
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt16, 0, PS_TAPS * KC * 4 * p.OP64 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.xbs), 0, 2 * plane_bytes, 0x00020000);
    constexpr int NA = (A_PIECES + NW - 1) / NW, NB = (UP_B_PIECES + NW - 1) / NW;
    int ldsA[NA], sofA[NA], ldsB[NB], sofB[NB], voffB[NB];
    // NMT = 2: a piece = 64 consecutive rows of one (tap, hi|lo, half); NMT = 1: the two halves x 32 rows of one (tap, hi|lo)
    const int voffA = NMT == 2 ? (m0 + lane) * 16 : ((lane >> 5) * p.OP64 + m0 + (lane & 31)) * 16;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int pa = wn + NW * j;
        const int t = NMT == 2 ? pa >> 2 : pa >> 1, hl = NMT == 2 ? (pa >> 1) & 1 : pa & 1, hf = NMT == 2 ? pa & 1 : 0;
        ldsA[j] = hl * A_SZ + (t * 2 + hf) * BM;
        sofA[j] = ((t * KC) * 4 + hl * 2 + hf) * p.OP64 * 16;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int q = wn + NW * j, hl = min(q, UP_B_PIECES - 1) / (2 * UP_BCH), hf = (q / UP_BCH) & 1, c = q % UP_BCH;
        ldsB[j] = 2 * A_SZ + hl * UP_B_SZ + hf * UP_BPAD + c * 64;
        sofB[j] = hl * plane_bytes + hf * HW * 16;
        const int pp = c * 64 + lane;                                     // patch pixel of this lane (row-major, run-time pitch PW)
        const int pr = pp / PW, iy = y0 - 1 + pr, ix = x0 - 1 + pp % PW;
        const bool ok = pr < prows && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        voffB[j] = ok ? (iy * p.W + ix) * 16 : (int)0x80000000;
    }
    const int strideA = 4 * p.OP64 * 16, strideB = 2 * HW * 16;
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
        b_frag[g] = half * UP_BPAD + q_row[g] * PW + q_col[g];            // + dy*PW + dx
    }
    float* s_rs = reinterpret_cast<float*>(smem + 2 * BUF);
    if (tid < BM) {
        const int o = min(m0 + tid, p.O - 1);
        s_rs[tid] = p.const_scale * (p.row_scale ? p.row_scale[(int64_t)n * p.row_scale_stride + o] : 1.f);
    }
    // one loop body holds both the staging of chunk kc + 1 and the multiplies of chunk kc (kc = -1: prologue)
    for (int kc = -1; kc < KC; ++kc) {
        if (kc + 1 < KC && (!(p.dbg & 4) || kc < 0)) {
            bf16x8* base = smem + ((kc + 1) & 1) * BUF;
#pragma unroll
            for (int j = 0; j < NA; ++j)
                if (j < NA - 1 || wn + NW * j < A_PIECES)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_void*)(base + ldsA[j]), 16, voffA, sofA[j] + (kc + 1) * strideA, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (j < NB - 1 || wn + NW * j < UP_B_PIECES)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x, (lds_void*)(base + ldsB[j]), 16, voffB[j], sofB[j] + (kc + 1) * strideB, 0, 0);
        }
        if (kc >= 0 && wave_on) {
            const bf16x8* A_hi = smem + (kc & 1) * BUF, *A_lo = A_hi + A_SZ, *B_hi = A_hi + 2 * A_SZ, *B_lo = B_hi + UP_B_SZ;
            __builtin_amdgcn_s_setprio(1);
            bf16x8 bh[PG][4], bl[PG][4];
#pragma unroll
            for (int g = 0; g < PG; ++g)
#pragma unroll
                for (int d = 0; d < 4; ++d) { bh[g][d] = B_hi[b_frag[g] + (d >> 1) * PW + (d & 1)]; bl[g][d] = B_lo[b_frag[g] + (d >> 1) * PW + (d & 1)]; }
            bf16x8 ah[2][NMT], al[2][NMT];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) { ah[0][mt] = A_hi[a_frag + mt * 32]; al[0][mt] = A_lo[a_frag + mt * 32]; }
#pragma unroll
            for (int t = 0; t < PS_TAPS; ++t) {
                const int ky = t / 3, kx = t % 3, s = t & 1;
                const int ph = (ky == 1 ? 2 : 0) + (kx == 1 ? 1 : 0);
                const int d = (ky == 2 ? 0 : 2) + (kx == 2 ? 0 : 1);
                if (t + 1 < PS_TAPS) {
#pragma unroll
                    for (int mt = 0; mt < NMT; ++mt) {
                        ah[s ^ 1][mt] = A_hi[(t + 1) * 2 * BM + a_frag + mt * 32];
                        al[s ^ 1][mt] = A_lo[(t + 1) * 2 * BM + a_frag + mt * 32];
                    }
                }
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                    for (int g = 0; g < PG; ++g) {
                        acc[mt][g][ph] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s][mt], bh[g][d], acc[mt][g][ph], 0, 0, 0);
                        acc[mt][g][ph] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][mt], bl[g][d], acc[mt][g][ph], 0, 0, 0);
                        acc[mt][g][ph] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s][mt], bh[g][d], acc[mt][g][ph], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_s_barrier();
    }

    // epilogue: channel-interleaved output (see conv2d_up_bf16x3_kernel's c8 path): 32 16-byte stores per lane and position group
    if (p.dbg & 1) { if (acc[0][0][0][0] == 123.456f) p.y[0] = 1.f; return; }
    float* yb = p.y + (int64_t)n * p.ybs;
#pragma unroll
    for (int g = 0; g < PG; ++g) {
        const int gy = y0 + q_row[g], gx = x0 + q_col[g];
        if (!q_act[g] || gy >= end_y || gx >= end_x) continue;
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
            const int oy = 2 * gy + pa;
            if (oy >= p.OH) continue;
            // the two horizontally adjacent phases back to back: a lane's 16-byte pieces of pixel 2 gx and 2 gx + 1 are 32 bytes apart, the other
            // channel half (lane ^ 32) fills the 16 bytes between — the four pieces of a 64-byte run leave the wave in two consecutive instructions
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const int c8 = (m0 >> 3) + mt * 4 + gg, ol = mt * 32 + 8 * gg + 4 * half;
                    if (p.y_nchw) {                                        // float32 NCHW (few-position layers): the two phases of a row are adjacent floats
                        const f32x16& a0 = acc[mt][g][pa * 2], &a1 = acc[mt][g][pa * 2 + 1];
                        const int ox = 2 * gx;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float* dst = yb + ((int64_t)(m0 + ol + k) * p.OH + oy) * p.yrs + ox;
                            const float v0 = n3d_round16(a0[4 * gg + k] * s_rs[ol + k], p.round_f16), v1 = n3d_round16(a1[4 * gg + k] * s_rs[ol + k], p.round_f16);
                            if (ox + 1 < p.OW && ((p.yrs | p.ybs) & 1) == 0) *reinterpret_cast<float2*>(dst) = float2{v0, v1};      // (ox even, even row pitch: 8-byte aligned)
                            else { dst[0] = v0; if (ox + 1 < p.OW) dst[1] = v1; }
                        }
                        continue;
                    }
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb) {
                        const int ox = 2 * gx + pb;
                        if (ox >= p.OW) continue;
                        const f32x16& a = acc[mt][g][pa * 2 + pb];
                        const f32x4 v = {n3d_round16(a[4 * gg + 0] * s_rs[ol + 0], p.round_f16), n3d_round16(a[4 * gg + 1] * s_rs[ol + 1], p.round_f16),
                                         n3d_round16(a[4 * gg + 2] * s_rs[ol + 2], p.round_f16), n3d_round16(a[4 * gg + 3] * s_rs[ol + 3], p.round_f16)};
                        *reinterpret_cast<f32x4*>(yb + (((int64_t)c8 * p.OH + oy) * p.yrs + ox) * 8 + 4 * half) = v;
                    }
                }
        }
    }
}

__global__ __launch_bounds__(512, 4) void conv2d_up_ps32_bf16x3_kernel(ConvUpPsParams p) {
    __shared__ bf16x8 smem[up_ps_smem_slots(1)];
    conv2d_up_ps_body<1, 8, 1>(p, smem);
}

// tuning builds (N3D_UP_WIDE=1): four waves per workgroup, each 64 positions x 32 channels x 4 phases — the weight fragments a wave reads serve two position
// groups (34 fragment reads per 54 MFMAs instead of 26 per 27), 128 accumulators per lane, two workgroups = eight waves per CU
__global__ __launch_bounds__(256, 2) void conv2d_up_ps32w_bf16x3_kernel(ConvUpPsParams p) {
    __shared__ bf16x8 smem[up_ps_smem_slots(1)];
    conv2d_up_ps_body<1, 4, 2>(p, smem);
}
__global__ __launch_bounds__(512, 4) void conv2d_up_ps32p_bf16x3_kernel(ConvUpPsParams p, int total) {      // persistent form (see conv2d_ps1p_bf16x3_kernel)
    __shared__ bf16x8 smem[up_ps_smem_slots(1)];
    int first, count, step;
    ps_persistent_range(total, first, count, step);
    for (int k = 0; k < count; k += step) {
        conv2d_up_ps_body<1, 8, 1>(p, smem, first + k);
        __syncthreads();
    }
}

__global__ __launch_bounds__(512, 4) void conv2d_up_ps32d_bf16x3_kernel(ConvUpPsParams p, int total, unsigned* q) {      // dynamic persistent form (see conv2d_ps1d_bf16x3_kernel)
    __shared__ bf16x8 smem[up_ps_smem_slots(1) + 1];
    int* s_slot = reinterpret_cast<int*>(smem + up_ps_smem_slots(1));
    int first, count;
    ps_queue_share(total, first, count);
    for (;;) {
        const int t = ps_dequeue(q, blockIdx.x & 7, s_slot);
        if (t >= count) break;
        conv2d_up_ps_body<1, 8, 1>(p, smem, first + t);
    }
    ps_queue_leave(q);
}

int conv2d_up_ps_bf16x3_launch(const n3d_conv2d_desc* d, hipStream_t stream) {
    const bool nchw = d->y_layout == N3D_LAYOUT_NCHW_F32;
    N3D_CHECK(d->ksize == 3 && d->mode == 2 && (d->y_layout == N3D_LAYOUT_C8_F32 || nchw), "conv2d_bf16x3: a split8 input to the transposed kernel: c8 or float32 NCHW output");
    N3D_CHECK(d->style == nullptr && d->ksplit <= 1, "conv2d_bf16x3: a split8 input carries its modulation already (style must be NULL), no split-K");
    N3D_CHECK(d->I % 16 == 0 && d->O % 64 == 0 && d->H >= 4 && d->W >= 4, "conv2d_bf16x3 (split8, transposed): I %% 16 == 0, O %% 64 == 0");
    // 32-channel workgroups, two per CU: measured 4-22 % faster than 64-channel ones on every transposed layer of the benchmark
    // (profiles/r02_conv_ps_ablation.txt); the 4-wave x 64-position form gained 0.3 % (round 3) — neither is instantiated any more.
    N3D_CHECK((int64_t)(d->I / 8) * d->H * d->W * 32 < (1ll << 31), "conv2d_bf16x3: one sample's split8 input exceeds 2 GiB (32-bit buffer offsets)");
    const n3d_epilogue& E = d->epi;
    N3D_CHECK(E.act == N3D_ACT_LINEAR && !E.noise && !E.bias && !E.residual && E.clamp < 0.f && E.gain == 1.f,
              "conv2d_bf16x3: the channel-interleaved output takes the demodulation-only epilogue (the layer epilogue runs behind the FIR)");
    N3D_CHECK(((uintptr_t)d->x & 15) == 0 && d->x_batch_stride % 4 == 0 && (nchw || (((uintptr_t)d->y & 15) == 0 && (d->y_batch_stride & 3) == 0)), "conv2d_bf16x3: misaligned split8 / c8 tensor");
    ConvUpPsParams p;
    p.x = (const bf16x8*)d->x; p.wt16 = (const bf16x8*)d->wt; p.y = d->y;
    p.N = d->N; p.I = d->I; p.O = d->O; p.OP64 = (d->O + 63) / 64 * 64; p.H = d->H; p.W = d->W; p.OH = 2 * d->H + 1; p.OW = 2 * d->W + 1;
    p.plan = up_tile_plan(d->H, d->W, true);
    p.tiles_m = d->O / 32;
    p.xbs = d->x_batch_stride ? d->x_batch_stride / 4 : (int64_t)2 * (d->I / 8) * d->H * d->W;
    p.yrs = d->y_row_stride ? d->y_row_stride : p.OW;
    p.ybs = d->y_batch_stride ? d->y_batch_stride : (int64_t)d->O * p.OH * p.yrs;
    N3D_CHECK(p.yrs >= p.OW, "conv2d_bf16x3: y_row_stride smaller than the output width");
    p.row_scale = E.row_scale; p.row_scale_stride = E.row_scale_stride ? E.row_scale_stride : d->O; p.const_scale = E.const_scale;
    p.round_f16 = E.round_f16;
    p.y_nchw = nchw ? 1 : 0;
    p.dbg = n3d_tune("N3D_CONV_DBG", 0);
    p.thin_last = n3d_tune("N3D_UP_THIN_LAST", 1);
    const int64_t nblk = (int64_t)p.plan.total * p.tiles_m * p.N;
    N3D_CHECK(nblk < (1ll << 31) && nblk > 0, "conv2d_bf16x3: grid too large");
    const double flops = 2.0 * d->N * (double)d->O * d->I * 9 * (double)d->H * d->W;
    const double bytes = 4.0 * ((double)d->N * d->I * d->H * d->W + (double)d->N * d->O * p.OH * p.OW + (double)d->O * d->I * 9);
    N3dProfScope prof(N3D_K_CONV2D_BF16X3, stream, flops, bytes);
    const int persist = n3d_tune("N3D_PS_PERSIST", 0);
    if (n3d_tune("N3D_UP_WIDE", 0)) hipLaunchKernelGGL(conv2d_up_ps32w_bf16x3_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, p);
    else if (persist == 2 && nblk > 512 && d->tickets && d->ticket_count >= 9) hipLaunchKernelGGL(conv2d_up_ps32d_bf16x3_kernel, dim3(512), dim3(512), 0, stream, p, (int)nblk, d->tickets);
    else if (persist == 1 && nblk > 512) hipLaunchKernelGGL(conv2d_up_ps32p_bf16x3_kernel, dim3(512), dim3(512), 0, stream, p, (int)nblk);
    else hipLaunchKernelGGL(conv2d_up_ps32_bf16x3_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------------------
// 3x3 stride-2 convolution (the down-sampling encoder layers, mode 1) on split8 input.  Same polyphase formulation as
// conv2d_s2_bf16x3.hip — the K loop runs over (16-channel chunk, phase (py, px)) with 4 / 2 / 2 / 1 taps per phase, one stage =
// the 17 x 33 patch of ONE phase image + that phase's weights — but the register-staged kernel pays the full staging work of a
// stride-1 chunk (24 four-byte gathers, style multiply, hi / lo split per work item) for a quarter of its MFMAs per stage and its
// matrix pipe is busy 23 % of the time (profiles/r02_mfma_busy_pmc.json).  Here a stage is 36 + 4 x taps LDS-DMA pieces: the
// de-interleave is done by the per-lane SOURCE address (patch pixel (a, b) of phase (py, px) = input pixel (2a + py, 2b + px)), the
// phase and the chunk move through the scalar offset.  Three LDS buffers of 52 KB: the DMA of stage s + 2 is issued before the
// MFMAs of stage s and the wait at the end of the iteration is `vmcnt(pieces of stage s + 2)` — stages carry 48 / 24 / 24 / 12
// MFMAs per wave, a two-deep queue evens that out.  One workgroup per CU, MFMAs transposed (pixels as matrix rows) and the
// epilogue of conv2d_ps_bf16x3_body; optional split-K like the register-staged kernel.
// Patch pixels beyond the image (only ever feeding outputs beyond OH x OW, which are not stored) read whatever follows in
// the sample, or zeros beyond its end (descriptor range check).
constexpr int S2_BM = 64, S2_TH = 16, S2_TW = 32;                                     // (S2_BM: the shipped kernel's channels per workgroup; the body is a template over it)
constexpr int S2_PH = S2_TH + 1, S2_PW = S2_TW + 1, S2_PPIX = S2_PH * S2_PW;         // 17 x 33 = 561
constexpr int S2_BCH = (S2_PPIX + 63) / 64, S2_BPAD = S2_BCH * 64;                    // 9 pieces = 576 slots per (hi|lo, half)
constexpr int S2_B_SZ = 2 * S2_BPAD;                                                  // per hi|lo: [half][pixel]
constexpr int s2_a_sz(int bm) { return 4 * 2 * bm; }                                   // per hi|lo: [slot 4][half][row]
constexpr int s2_buf(int bm) { return 2 * s2_a_sz(bm) + 2 * S2_B_SZ; }                 // BM 64: 3328 slots = 53,248 B; BM 128: 4352 slots = 69,632 B
constexpr int S2_B_PIECES = 2 * 2 * S2_BCH;                                            // 36
// MTW = 32-channel groups per workgroup, NBUF LDS buffers.  <2, 3>: shipped (64 channels, the DMA of stage s + 2 in flight).  <4, 2> (tuning builds, N3D_S2_WIDE=1; VERDICT r5
// item 1c): 128 channels per workgroup — the patch is staged once per 128 output channels instead of per 64, each patch fragment a wave reads feeds twice the MFMAs (12 fragment
// reads per 24 MFMAs and tap instead of 8 per 12), 128 accumulators per lane — but only TWO 68 KB buffers fit: the DMA queue is one stage deep.

struct ConvS2PsParams {
    const bf16x8* x; const bf16x8* wt16; float* y; float* partial;
    int N, I, O, OP64, H, W, OH, OW;
    int tiles_x, tiles_y, tiles_m, ksplit, ic_per_split;
    int64_t xbs, ybs, yrs;
    n3d_epilogue epi;
};

template <int MTW, int NBUF>
__device__ __forceinline__ void conv2d_s2_ps_body(const ConvS2PsParams& p, bf16x8* smem) {
    constexpr int BM = 32 * MTW, A_SZ = s2_a_sz(BM), BUF = s2_buf(BM), AW = BM / 64;      // AW: 64-row weight pieces per (slot, hi|lo, half)
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int lb;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int m0 = (lb % p.tiles_m) * BM; lb /= p.tiles_m;
    const int tile_i = lb % (p.tiles_x * p.tiles_y); lb /= (p.tiles_x * p.tiles_y);
    const int ks = lb % p.ksplit, n = lb / p.ksplit;
    const int y0 = (tile_i / p.tiles_x) * S2_TH, x0 = (tile_i % p.tiles_x) * S2_TW;
    const int KC = p.I / 16, HW = p.H * p.W;
    const int c_begin = ks * (p.ic_per_split / 16), c_end = min(KC, c_begin + p.ic_per_split / 16);
    const int nstage = (c_end - c_begin) * 4;                             // stage st -> chunk c_begin + (st >> 2), phase st & 3

    const int plane_bytes = (p.I / 8) * HW * 16;
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt16, 0, PS_TAPS * KC * 4 * p.OP64 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.xbs), 0, 2 * plane_bytes, 0x00020000);

    // This wave's pieces.  Patch: q = wn + 8 j (j < 5, q < 36) = (hi|lo, half, 64-pixel run).  Weights: piece a = wn + 8 j (j < 2,
    // a < 4 * taps) = (slot, hi|lo, half) slabs of 64 rows.
    constexpr int NB = (S2_B_PIECES + 7) / 8;                             // 5 (the fifth only for waves 0-3)
    int ldsB[NB], sofB[NB], voffB[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int q = wn + 8 * j, hl = q / (2 * S2_BCH), hf = (q / S2_BCH) & 1, c = q % S2_BCH;
        ldsB[j] = 2 * A_SZ + hl * S2_B_SZ + hf * S2_BPAD + c * 64;
        sofB[j] = hl * plane_bytes + hf * HW * 16;
        const int pp = c * 64 + lane;
        const int iy = 2 * (y0 + pp / S2_PW), ix = 2 * (x0 + pp % S2_PW);   // phase (0, 0) pixel; the phase adds (py W + px) * 16 to the scalar offset
        voffB[j] = pp < S2_PPIX && iy < p.H && ix < p.W ? (iy * p.W + ix) * 16 : (int)0x80000000;
    }
    const int voffA = (m0 + lane) * 16;
    const int strideB = 2 * HW * 16, strideA = 4 * p.OP64 * 16;           // scalar-offset step per 16-channel chunk
    // issue stage st into buffer st % 3; returns nothing — the number of DMA instructions of a (phase, wave) pair is dma_count()
    auto copy_stage = [&](int st) {
        const int c = c_begin + (st >> 2), g = st & 3, py = g >> 1, px = g & 1;
        const int nx = px ? 1 : 2, nt = (py ? 1 : 2) * nx;
        bf16x8* base = smem + (st % NBUF) * BUF;
#pragma unroll
        for (int j = 0; j < 2 * AW; ++j) {
            const int a = wn + 8 * j;                                     // piece a = (slot, hi|lo, half, 64-row part)
            if (a < 4 * nt * AW) {
                const int mh = a % AW, a4 = a / AW, slot = a4 >> 2, hl = (a4 >> 1) & 1, hf = a4 & 1;
                const int ky = py ? 1 : 2 * (slot / nx), kx = px ? 1 : 2 * (slot % nx);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_void*)(base + hl * A_SZ + (slot * 2 + hf) * BM + mh * 64), 16, voffA + mh * 64 * 16,
                                                         ((((ky * 3 + kx) * KC + c) * 2 + hl) * 2 + hf) * p.OP64 * 16, 0, 0);
            }
        }
        const int sph = (py * p.W + px) * 16 + c * strideB;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < NB - 1 || wn + 8 * j < S2_B_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x, (lds_void*)(base + ldsB[j]), 16, voffB[j], sofB[j] + sph, 0, 0);
    };
    // DMA instructions this wave issues for a stage of phase g: patch 5 (waves 0-3) or 4, weights 2 / 1 / 1 / (waves 0-3: 1, else 0)
    auto dma_count = [&](int g) { return (wn < 4 ? 5 : 4) + (g == 0 ? 2 : (g == 3 ? (wn < 4 ? 1 : 0) : 1)); };      // (MTW = 2: the three-buffer pipeline's wait counts)

    f32x16 acc[MTW][2];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const int a_frag = half * BM + l31;                                   // + slot*2*BM + mt*32
    const int b_frag = half * S2_BPAD + (wn * 2) * S2_PW + l31;           // + nt*PW + dy*PW + dx
    auto mfma_taps = [&](int st, auto ny_c, auto nx_c) {                  // taps (dy, dx), dy < NY, dx < NX; slot = dy*NX + dx
        constexpr int NY = decltype(ny_c)::value, NX = decltype(nx_c)::value;
        const bf16x8* A_hi = smem + (st % NBUF) * BUF, *A_lo = A_hi + A_SZ, *B_hi = A_hi + 2 * A_SZ, *B_lo = B_hi + S2_B_SZ;
        __builtin_amdgcn_s_setprio(1);
        bf16x8 ah[2], al[2], bh[2], bl[2];                                // (weight fragments two channel groups at a time: MTW = 4 has 128 accumulators and 256 registers)
#pragma unroll
        for (int dy = 0; dy < NY; ++dy)
#pragma unroll
            for (int dx = 0; dx < NX; ++dx) {
                const int slot = dy * NX + dx, boff = dy * S2_PW + dx;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) { bh[nt] = B_hi[b_frag + nt * S2_PW + boff]; bl[nt] = B_lo[b_frag + nt * S2_PW + boff]; }
#pragma unroll
                for (int mp = 0; mp < MTW; mp += 2) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) { ah[m] = A_hi[slot * 2 * BM + a_frag + (mp + m) * 32]; al[m] = A_lo[slot * 2 * BM + a_frag + (mp + m) * 32]; }
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {                  // pixels = matrix rows, as conv2d_ps_bf16x3_body
                            acc[mp + m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[nt], al[m], acc[mp + m][nt], 0, 0, 0);
                            acc[mp + m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[nt], ah[m], acc[mp + m][nt], 0, 0, 0);
                            acc[mp + m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[nt], ah[m], acc[mp + m][nt], 0, 0, 0);
                        }
                    if (MTW > 2) __builtin_amdgcn_sched_barrier(0);       // (keeps the scheduler from hoisting every tap's fragment reads: 128 accumulators leave no room for them)
                }
            }
        __builtin_amdgcn_s_setprio(0);
    };
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    float* s_rs = reinterpret_cast<float*>(smem + NBUF * BUF), *s_bs = s_rs + BM;
    const n3d_epilogue& E = p.epi;
    if (tid < BM) {
        const int o = min(m0 + tid, p.O - 1);
        s_rs[tid] = E.const_scale * (E.row_scale ? E.row_scale[(int64_t)n * (E.row_scale_stride ? E.row_scale_stride : p.O) + o] : 1.f);
        s_bs[tid] = E.bias ? E.bias[o] : 0.f;
    }

    if (NBUF == 2) {                                                      // two buffers: stage st + 1 lands under the MFMAs of stage st
        if (nstage > 0) copy_stage(0);
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_s_barrier();
        for (int st = 0; st < nstage; ++st) {
            if (st + 1 < nstage) copy_stage(st + 1);                      // its buffer's last readers passed the barrier of iteration st - 1
            switch (st & 3) {
                case 0: mfma_taps(st, I2{}, I2{}); break;
                case 1: mfma_taps(st, I2{}, I1{}); break;
                case 2: mfma_taps(st, I1{}, I2{}); break;
                default: mfma_taps(st, I1{}, I1{}); break;
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __builtin_amdgcn_s_barrier();
        }
    } else {
    if (nstage > 0) copy_stage(0);
    if (nstage > 1) copy_stage(1);
    // stage 0 must be complete: at most the pieces of stage 1 may still be in flight (vmcnt counts in order)
    if (nstage > 1) { if (wn < 4) __builtin_amdgcn_s_waitcnt(0x0f70 | 6); else __builtin_amdgcn_s_waitcnt(0x0f70 | 5); }     // stage 1 = phase 1: 6 / 5 pieces
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    __builtin_amdgcn_s_barrier();
    for (int st = 0; st < nstage; ++st) {
        const bool more = st + 2 < nstage;
        if (more) copy_stage(st + 2);                                     // its buffer's last readers passed the barrier of iteration st - 1
        switch (st & 3) {
            case 0: mfma_taps(st, I2{}, I2{}); break;
            case 1: mfma_taps(st, I2{}, I1{}); break;
            case 2: mfma_taps(st, I1{}, I2{}); break;
            default: mfma_taps(st, I1{}, I1{}); break;
        }
        // stage st + 1 complete = everything but the pieces of stage st + 2 (if issued) has landed
        switch (more ? dma_count((st + 2) & 3) : 0) {
            case 0: __builtin_amdgcn_s_waitcnt(0x0f70); break;
            case 4: __builtin_amdgcn_s_waitcnt(0x0f70 | 4); break;
            case 5: __builtin_amdgcn_s_waitcnt(0x0f70 | 5); break;
            case 6: __builtin_amdgcn_s_waitcnt(0x0f70 | 6); break;
            default: __builtin_amdgcn_s_waitcnt(0x0f70 | 7); break;
        }
        __builtin_amdgcn_s_barrier();
    }
    }

    // epilogue (conv2d_ps_bf16x3_body's): lane = one channel of group mt, 16 pixels of the wave's row nt in 4 runs of 4
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int64_t plane = (int64_t)p.OH * p.OW, yplane = (int64_t)p.OH * p.yrs;
    if (p.partial) {                                                      // split-K: raw sums, reduced by conv16_splitk_epilogue_kernel
        const bool vec4 = (p.OW & 3) == 0;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int o = m0 + mt * 32 + l31;
            if (o >= p.O) continue;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int oy = y0 + wn * 2 + nt;
                if (oy >= p.OH) continue;
                float* drow = p.partial + (((int64_t)ks * p.N + n) * p.O + o) * plane + (int64_t)oy * p.OW;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ox = x0 + 8 * g + 4 * half;
                    if (ox >= p.OW) continue;
                    if (vec4) *reinterpret_cast<f32x4*>(drow + ox) = f32x4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
                    else
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (ox + k < p.OW) drow[ox + k] = acc[mt][nt][4 * g + k];
                }
            }
        }
        return;
    }
    const float nstr = E.noise ? E.noise_strength[0] : 0.f;
    const bool lrelu = E.act == N3D_ACT_LRELU;
    const float alpha_eff = lrelu ? E.alpha : 1.f, clamp_eff = E.clamp >= 0.f ? E.clamp : INFINITY;
    const bool vec = ((p.OW | p.yrs | p.ybs) & 3) == 0 && ((uintptr_t)p.y & 15) == 0 &&
                     (!E.residual || ((E.residual_batch_stride & 3) == 0 && ((uintptr_t)E.residual & 15) == 0)) && (!E.noise || ((uintptr_t)E.noise & 15) == 0);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int o = m0 + mt * 32 + l31;
        if (o >= p.O) continue;
        const float rs = s_rs[mt * 32 + l31], bs = s_bs[mt * 32 + l31];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int oy = y0 + wn * 2 + nt;
            if (oy >= p.OH) continue;
            float* drow = p.y + (int64_t)n * p.ybs + (int64_t)o * yplane + (int64_t)oy * p.yrs;
            const float* rrow = E.residual ? E.residual + (int64_t)n * E.residual_batch_stride + (int64_t)o * plane + (int64_t)oy * p.OW : nullptr;
            const float* nrow = E.noise ? E.noise + (int64_t)oy * p.OW : nullptr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ox = x0 + 8 * g + 4 * half;
                if (ox >= p.OW) continue;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = acc[mt][nt][4 * g + k];
                if (vec) {
                    f32x4 nz = {0.f, 0.f, 0.f, 0.f};
                    if (nrow) nz = *reinterpret_cast<const f32x4*>(nrow + ox) * nstr;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float t = v[k] * rs + nz[k] + bs;
                        t = fmaxf(t, t * alpha_eff) * E.gain;
                        v[k] = n3d_round16(fminf(fmaxf(t, -clamp_eff), clamp_eff), E.round_f16);
                    }
                    f32x4 out = {v[0], v[1], v[2], v[3]};
                    if (rrow) out += *reinterpret_cast<const f32x4*>(rrow + ox);
                    *reinterpret_cast<f32x4*>(drow + ox) = out;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (ox + k >= p.OW) break;
                        float t = v[k] * rs + (nrow ? nrow[ox + k] * nstr : 0.f) + bs;
                        t = fmaxf(t, t * alpha_eff) * E.gain;
                        t = n3d_round16(fminf(fmaxf(t, -clamp_eff), clamp_eff), E.round_f16);
                        if (rrow) t += rrow[ox + k];
                        drow[ox + k] = t;
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void conv2d_s2_ps_bf16x3_kernel(ConvS2PsParams p) {
    __shared__ bf16x8 smem[3 * s2_buf(64) + 2 * 64 * 4 / 16];
    conv2d_s2_ps_body<2, 3>(p, smem);
}
__global__ __launch_bounds__(512, 2) void conv2d_s2_ps128_bf16x3_kernel(ConvS2PsParams p) {                   // tuning builds: N3D_S2_WIDE=1
    __shared__ bf16x8 smem[2 * s2_buf(128) + 2 * 128 * 4 / 16];
    conv2d_s2_ps_body<4, 2>(p, smem);
}

int conv16_splitk_epilogue_launch(const float* partial, float* y, int ksplit, int N, int O, int OH, int OW, int64_t ybs, int64_t yrs,
                                  const n3d_epilogue& epi, hipStream_t stream);          // conv2d_bf16x3.hip

int conv2d_s2_ps_bf16x3_launch(const n3d_conv2d_desc* d, hipStream_t stream) {
    N3D_CHECK(d->ksize == 3 && d->mode == 1 && d->H >= 3 && d->W >= 3, "conv2d_bf16x3 (split8, stride 2): 3x3 kernel, input of at least 3 x 3");
    N3D_CHECK(d->style == nullptr, "conv2d_bf16x3: a split8 input carries its modulation already (style must be NULL)");
    N3D_CHECK(d->I % 16 == 0 && d->I >= 16, "conv2d_bf16x3 (split8, stride 2): I %% 16 == 0");
    N3D_CHECK((int64_t)(d->I / 8) * d->H * d->W * 32 < (1ll << 31), "conv2d_bf16x3: one sample's split8 input exceeds 2 GiB (32-bit buffer offsets)");
    const n3d_epilogue& E = d->epi;
    N3D_CHECK(E.act == N3D_ACT_LINEAR || (E.act == N3D_ACT_LRELU && E.alpha >= 0.f && E.alpha <= 1.f), "conv2d_bf16x3 (split8): linear or leaky-ReLU epilogue only");
    N3D_CHECK(!E.residual_up_filter, "conv2d_bf16x3: residual_up_filter is only supported by the 1x1 kernel");
    N3D_CHECK(((uintptr_t)d->x & 15) == 0 && d->x_batch_stride % 4 == 0, "conv2d_bf16x3: split8 input must be 16-byte aligned");
    ConvS2PsParams p;
    p.x = (const bf16x8*)d->x; p.wt16 = (const bf16x8*)d->wt; p.y = d->y; p.partial = d->workspace;
    p.N = d->N; p.I = d->I; p.O = d->O; p.OP64 = (d->O + 63) / 64 * 64; p.H = d->H; p.W = d->W;
    p.OH = (d->H - 3) / 2 + 1; p.OW = (d->W - 3) / 2 + 1;
    const bool wide = n3d_tune("N3D_S2_WIDE", 0) != 0 && p.O % 128 == 0;
    p.tiles_x = cdiv(p.OW, S2_TW); p.tiles_y = cdiv(p.OH, S2_TH); p.tiles_m = cdiv(p.O, wide ? 128 : S2_BM);
    p.xbs = d->x_batch_stride ? d->x_batch_stride / 4 : (int64_t)2 * (d->I / 8) * d->H * d->W;
    p.ybs = d->y_batch_stride; p.yrs = d->y_row_stride ? d->y_row_stride : p.OW;
    N3D_CHECK(p.yrs >= p.OW, "conv2d_bf16x3: y_row_stride smaller than the output width");
    const int max_split = d->I / 16;
    p.ksplit = d->ksplit < 1 ? 1 : (d->ksplit > max_split ? max_split : d->ksplit);
    p.ic_per_split = cdiv(cdiv(d->I, p.ksplit), 16) * 16;
    p.ksplit = cdiv(d->I, p.ic_per_split);
    N3D_CHECK(p.ksplit == 1 || d->workspace != nullptr, "conv2d_bf16x3: ksplit > 1 needs a workspace");
    if (p.ksplit == 1) p.partial = nullptr;
    p.epi = d->epi;
    const int64_t nblk = (int64_t)p.tiles_x * p.tiles_y * p.tiles_m * p.N * p.ksplit;
    N3D_CHECK(nblk < (1ll << 31) && nblk > 0, "conv2d_bf16x3: grid too large");
    const double flops = 2.0 * d->N * (double)d->O * d->I * 9 * (double)p.OH * p.OW;
    const double bytes = 4.0 * ((double)d->N * d->I * d->H * d->W + (double)d->N * d->O * p.OH * p.OW + (double)d->O * d->I * 9);
    N3dProfScope prof(N3D_K_CONV2D_BF16X3, stream, flops, bytes);
    if (wide) hipLaunchKernelGGL(conv2d_s2_ps128_bf16x3_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL(conv2d_s2_ps_bf16x3_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
    N3D_LAUNCH_CHECK();
    if (p.ksplit > 1) return conv16_splitk_epilogue_launch(p.partial, p.y, p.ksplit, p.N, p.O, p.OH, p.OW, p.ybs, p.yrs, p.epi, stream);
    return 0;
}
