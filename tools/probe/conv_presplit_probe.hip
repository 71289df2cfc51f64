// tools/probe/conv_presplit_probe.hip — COMPILE-ONLY study for docs/history/DESIGN_rounds1-4.md §7 item 1 (not part of libn3d.so, never launched by
// the product or the tests): what the stride-1 3x3 K loop of conv2d_bf16x3.hip looks like when the activations arrive ALREADY
// split (bf16 hi / lo planes in the consumer's fragment layout, [I/8][H][W][8] = 16-byte units) and the style has been folded
// into per-sample weights.  Staging is then a pure copy, done by LDS-DMA (`buffer_load_dwordx4 ... lds`): no landing registers,
// no style multiply, no split, no ds_write, no wave roles.
//
//     hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -I include -o /tmp/probe.s tools/probe/conv_presplit_probe.hip
//
// Findings of the round-1 compile (ROCm 7.2): see tools/probe/README_presplit.md.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

struct PresplitParams {
    const bf16x8* xh; const bf16x8* xl;      // [N][I/8][H][W] 16-byte units: 8 consecutive channels of one pixel, hi and lo halves
    const bf16x8* wt16;                      // [N][tap][I/16][hl][half][O][8]: style already multiplied in (per-sample weights)
    float* y;                                // fp32 NCHW output (a real producer would write hi / lo planes for the next layer)
    int N, I, O, H, W, tiles_x, tiles_y, tiles_m;
};

// 8 waves, tile = 64 output channels x (16 x 32) pixels, wave tile 64 x 64 (2 x 2 accumulators) — the geometry of the shipped kernel.
__global__ __launch_bounds__(512, 2) void conv_presplit_probe_kernel(PresplitParams p) {
    constexpr int BM = 64, TH = 16, TW = 32, TAPS = 9;
    constexpr int PH = TH + 2, PW = TW + 2, PPIX = PH * PW;               // 612 patch pixels
    constexpr int BCH = 10, BPAD = BCH * 64;                              // B image per (hi|lo, half): 640 slots = 10 DMA pieces of 64 lanes
    constexpr int A_SZ = TAPS * 2 * BM, B_SZ = 2 * BPAD;
    constexpr int A_PIECES = TAPS * 2 * 2, B_PIECES = 2 * 2 * BCH, PIECES = A_PIECES + B_PIECES;     // 36 + 40 wave-level copies per chunk
    constexpr int PER_WAVE = (PIECES + 7) / 8;                            // 10
    __shared__ bf16x8 A_hi[2 * A_SZ], A_lo[2 * A_SZ];                     // [buf][tap][half][row]
    __shared__ bf16x8 B_hi[2 * B_SZ], B_lo[2 * B_SZ];                     // [buf][half][pixel (padded to 640)]

    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int lb = blockIdx.x;
    const int m0 = (lb % p.tiles_m) * BM; lb /= p.tiles_m;
    const int tile_i = lb % (p.tiles_x * p.tiles_y), n = lb / (p.tiles_x * p.tiles_y);
    const int y0 = (tile_i / p.tiles_x) * TH, x0 = (tile_i % p.tiles_x) * TW;
    const int KC = p.I / 16, HW = p.H * p.W;

    // descriptors: this sample's weights / hi plane / lo plane (range-checked: out-of-image pixels read as zero)
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.wt16 + (int64_t)n * TAPS * KC * 4 * p.O), 0, TAPS * KC * 4 * p.O * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_h = __builtin_amdgcn_make_buffer_rsrc((void*)(p.xh + (int64_t)n * (p.I / 8) * HW), 0, (p.I / 8) * HW * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_l = __builtin_amdgcn_make_buffer_rsrc((void*)(p.xl + (int64_t)n * (p.I / 8) * HW), 0, (p.I / 8) * HW * 16, 0x00020000);

    // this wave's copy pieces (piece = wn + 8 j): per-lane source offsets are chunk-independent, the chunk moves the scalar offset
    int voff[PER_WAVE];
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int pc = wn + 8 * j;
        if (pc < A_PIECES) {
            voff[j] = (m0 + lane) * 16;                                    // 64 consecutive weight rows
        } else {
            const int pp = ((pc - A_PIECES) % BCH) * 64 + lane;            // patch pixel of this lane
            const int iy = y0 - 1 + pp / PW, ix = x0 - 1 + pp % PW;
            const bool ok = pp < PPIX && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            voff[j] = ok ? (iy * p.W + ix) * 16 : (int)0x80000000;
        }
    }
    auto copy_chunk = [&](int kc, int buf) {                              // LDS-DMA of K chunk kc into LDS buffer buf: <= 10 instructions per wave
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int pc = wn + 8 * j;
            if (pc >= PIECES) continue;
            if (pc < A_PIECES) {
                const int t = pc >> 2, hl = (pc >> 1) & 1, hf = pc & 1;
                bf16x8* dst = (hl ? A_lo : A_hi) + buf * A_SZ + (t * 2 + hf) * BM;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_void*)dst, 16, voff[j], ((t * KC + kc) * 4 + hl * 2 + hf) * p.O * 16, 0, 0);
            } else {
                const int q = pc - A_PIECES, hl = q / (2 * BCH), hf = (q / BCH) & 1, c = q % BCH;
                bf16x8* dst = (hl ? B_lo : B_hi) + buf * B_SZ + hf * BPAD + c * 64;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(hl ? r_l : r_h, (lds_void*)dst, 16, voff[j], (kc * 2 + hf) * HW * 16, 0, 0);
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const int a_frag = half * BM + l31;
    const int b_frag0 = half * BPAD + (wn * 2) * PW + l31, b_frag1 = b_frag0 + PW;
    auto mfma_block = [&](int buf) {
        const int bo_a = buf * A_SZ, bo_b = buf * B_SZ;
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
    };

    copy_chunk(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);                                   // vmcnt(0): the first chunk has landed
    __builtin_amdgcn_s_barrier();
    for (int kc = 0; kc < KC; ++kc) {
        if (kc + 1 < KC) copy_chunk(kc + 1, (kc + 1) & 1);                // DMA of the next chunk runs under this chunk's MFMAs
        mfma_block(kc & 1);
        __builtin_amdgcn_s_waitcnt(0x0f70);                               // vmcnt(0) only: this wave's DMA pieces are in LDS ...
        __builtin_amdgcn_s_barrier();                                     // ... and after the barrier everybody's are (raw barrier: no extra drain)
    }
    // bare epilogue (C/D layout: col = lane&31 = pixel, row = (r&3) + 8*(r>>2) + 4*(lane>>5) = channel)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int oy = y0 + wn * 2 + nt, ox = x0 + l31;
        if (oy >= p.H || ox >= p.W) continue;
        float* d0 = p.y + ((int64_t)n * p.O + m0 + 4 * half) * HW + (int64_t)oy * p.W + ox;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) d0[(int64_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * HW] = acc[mt][nt][r];
    }
}
