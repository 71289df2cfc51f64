// conv2d_sk_bf16x3.hip — the 3x3 stride-1 split-bf16 convolution of the FEW-PIXEL layers (4 x 4 .. 16 x 16 at batch 4, up to 32 x 32 at
// batch 1: at most 2048 output pixels in the whole batch) in ONE launch.  Same arithmetic as conv2d_bf16x3.hip (three
// v_mfma_f32_32x32x16_bf16 per 16-channel chunk and tap on hi / lo operand halves, float32 accumulation); replaces the same reference call
// sites (F.conv2d inside modulated_conv2d, tat/networks_stylegan2.py:34-91, and Conv2dLayer :173-183).
//
// Why a separate kernel: these layers are K-deep (9 x 512..1024) on a handful of pixels.  conv2d_bf16x3_kernel fills the chip by splitting K
// over 8-16 WORKGROUPS, which costs a partial-sum round trip through HBM plus a second launch (conv16_splitk_epilogue_kernel), and every
// workgroup's K loop pays the register-staging latency per chunk: 25-32 us per layer launch to launch for ~1-5 us of matrix work
// (profiles/r03_small_layer_sweep.txt).  Here the 8 waves of a workgroup split K INSIDE the workgroup (the scheme of
// conv1x1_bf16x3_ksplit_kernel): a workgroup owns 32 output channels x 32 or 64 pixels for the whole K; wave w takes the chunks
// [w * KC/8, (w+1) * KC/8) and, per chunk,
//   * loads its weight fragments (9 taps x hi|lo) straight from the L2-resident prepared tiles — no LDS staging, no barrier,
//   * stages the 16 channels of the tile's pixel patch (tile rows + one halo row / column on every side, <= 128 patch pixels) ONCE: two
//     pixels per lane, style multiply + hi / lo split, into a PRIVATE 8 KB LDS region of the wave (in-order LDS: no barrier either),
//   * reads the 9 shifted pixel fragments back and issues 27 MFMAs per 32-pixel group;
// the eight partial accumulators are summed through LDS in a fixed order (deterministic) and wave w applies the epilogue to rows r = w (mod 8).
// The grid is XCD-aware the other way round from the large kernels: a 32-channel weight tile (9 * I * 32 * 4 B = 0.6 MB at I = 512) is
// re-read by every pixel tile, so all workgroups of one channel tile sit on ONE XCD and its 4 MB L2 holds the two or so tiles it serves.
//
// Round 6 — K also split ACROSS workgroups, reduced inside the launch (KS > 1).  With K inside one workgroup a 4 x 4 .. 16 x 16 layer is 16-64
// workgroups, each streaming 0.6 MB of weight tiles in dependent chunk round trips: 17-44 us for a 9.4 MB weight stream whose HBM time is ~2 us
// (profiles/r05_layer_trace.txt).  Now the grid is (pixel tile, channel tile, K slice): slice s of KS takes the chunks
// [(8 s + w) nc, ...) for its wave w, nc = KC / (8 KS) — down to ONE chunk (18 KB of fragments) per wave — so that the whole chip pulls the layer's
// weights at once.  The slices meet through HBM: every workgroup writes the LDS-reduced sums of its slice as a 4-16 KB slab with write-through (sc1)
// stores, every wave drains them (s_waitcnt vmcnt(0)), one lane takes a ticket (relaxed agent-scope fetch_add on the tile's arrival counter), and the
// workgroup that draws KS - 1 — the last arriver, whichever it is — re-arms the counter, issues ONE agent-scope acquire, adds the KS slabs IN SLICE
// ORDER (bitwise reproducible whatever the arrival order) and applies the epilogue: no second launch, no spin wait, no assumption about dispatch
// order or placement (cdna_hip_programming.md section 5 item 2 / Guideline 16).  The arrival counters are the CALLER's (n3d_conv2d_desc.tickets:
// zeroed once, one pool per stream; every launch leaves them zero).
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct SkParams {
    const float* x; const bf16x8* wt16; const float* style; float* y;
    int N, I, O, OP64, H, W, HW;         // H, W: OUTPUT image (= the input image for stride 1)
    int IH, IW, HWin;                    // input image (stride 2: IH = 2 H + 1, IW = 2 W + 1, no padding: conv2d_resample.py:108-111 pads in the FIR)
    int R, NS, PR, PW, nslots, tps;      // tile rows per sample, samples per tile, patch rows / pitch, patch pixels, tiles per sample (NS == 1)
    int tiles_p, tiles_m;
    int x_bytes;                          // size of the whole input tensor (buffer descriptor range)
    int64_t xbs, ybs, yrs, style_stride;
    int KS;                               // K slices over workgroups (1 = the whole K inside one workgroup, no slabs)
    float* partial; unsigned* tickets;    // KS > 1: slabs [tile][channel tile][slice][PT * 16 items][64 lanes]; arrival counters [tile][channel tile]
    n3d_epilogue epi;
};

// ---- the seam between the K slices of one output tile (see the header comment)
typedef float sk_f32x2 __attribute__((ext_vector_type(2)));
typedef float sk_f32x4 __attribute__((ext_vector_type(4)));
typedef int sk_i32x2 __attribute__((ext_vector_type(2)));
typedef int sk_i32x4 __attribute__((ext_vector_type(4)));
constexpr int SK_SC1 = 16;                                                 // aux bit of the buffer instructions: sc1 = write-through store / L1-bypassing load
// VW (2 or 4) floats per lane, write-through; `off` = float index into the slab workspace
template <int VW>
__device__ __forceinline__ void sk_slab_store(const __amdgpu_buffer_rsrc_t& r, int off, const float (&v)[VW]) {
    if constexpr (VW == 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sk_i32x2, sk_f32x2{v[0], v[1]}), r, off * 4, 0, SK_SC1);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sk_i32x4, sk_f32x4{v[0], v[1], v[2], v[3]}), r, off * 4, 0, SK_SC1);
}
template <int VW>
__device__ __forceinline__ void sk_slab_add(const __amdgpu_buffer_rsrc_t& r, int off, float (&v)[VW]) {     // v += slab values (sc1 loads: the slabs were stored sc1, no acquire needed)
    if constexpr (VW == 2) { const sk_f32x2 t = __builtin_bit_cast(sk_f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off * 4, 0, SK_SC1)); v[0] += t[0]; v[1] += t[1]; }
    else { const sk_f32x4 t = __builtin_bit_cast(sk_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off * 4, 0, SK_SC1)); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
}
// Every wave calls this after its slab stores.  Returns true in the workgroup that arrived last (the counter is re-armed then, and all waves are past a barrier).
__device__ __forceinline__ bool sk_arrive_last(unsigned* ticket, int KS, unsigned* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // EVERY storing wave drains its write-through stores ...
    __syncthreads();                                                       // ... before ONE lane takes the ticket
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)(KS - 1)) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the stream's next launch: nobody else arrives any more
        *s_flag = old;
    }
    __syncthreads();
    return *s_flag == (unsigned)(KS - 1);
}

constexpr int SK_SLOTS = 128;                                              // patch pixels per wave region (stride 1; the transposed twin below)
constexpr int SK_WAVE_SLOTS = 2 * 2 * SK_SLOTS;                            // [hi|lo][half][pixel] 16-byte slots = 8 KB

// S = 1: stride 1, padding 1.  S = 2 (round 5): stride 2, padding 0 — the <= 17 x 17 down-sampling layers of the mouth encoder (Conv2dLayer
// down = 2 behind its FIR, conv2d_resample.py:108-111), which ran on the fp32-MFMA kernel with split-K 16 + a reduce launch (58 us each): the patch
// of a 32-pixel output tile is (2 R + 1) x (2 W + 1) <= 192 input pixels (three per lane), tap (ky, kx) of output (oy, ox) reads patch pixel
// (2 oy + ky, 2 ox + kx) — the same shifted-fragment scheme with a doubled pitch.
template <int PT, int S = 1>                                               // 32-pixel groups per workgroup tile; stride
__global__ __launch_bounds__(512) void conv2d_sk_bf16x3_kernel(SkParams p) {
    constexpr int NJ = S == 1 ? 2 : 3, SLOTS = 64 * NJ, WAVE_SLOTS = 4 * SLOTS, PAD = S == 1 ? 1 : 0;
    __shared__ bf16x8 smem[8 * WAVE_SLOTS];                                // 64 KB (96 KB for S = 2): the waves' patch regions; afterwards the partial sums
    __shared__ float s_style[2 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int mt_i, tile, ks;
    {
        const int b = blockIdx.x, units = p.tiles_m * p.KS;                // unit = (channel tile, K slice): one slab of weights
        int u;
        if ((units & 7) == 0) {                                            // one unit's workgroups (the pixel tiles re-reading its weights) share an XCD (= blockIdx % 8)
            const int per = units >> 3, q = b >> 3;
            u = (b & 7) + 8 * (q % per); tile = q / per;
        } else { u = b % units; tile = b / units; }
        mt_i = u % p.tiles_m; ks = u / p.tiles_m;
    }
    const int m0 = mt_i * 32;
    const int n0 = p.NS == 1 ? tile / p.tps : tile * p.NS;
    const int y0 = p.NS == 1 ? (tile % p.tps) * p.R : 0;
    const int KC = p.I / 16, nc = KC / (8 * p.KS), c_begin = (ks * 8 + wn) * nc;
    const int PRW = p.PR * p.PW;

    for (int i = tid; i < p.NS * p.I; i += 512) {
        const int s = i / p.I, ch = i - s * p.I;
        s_style[i] = (p.style && n0 + s < p.N) ? p.style[(int64_t)(n0 + s) * p.style_stride + ch] : 1.f;
    }

    // this lane's NJ patch pixels: byte offset of (sample, channel 0, y, x) in x, or out of range (halo, beyond the batch)
    int voff[NJ], srow[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int pp = lane + 64 * j;
        const int s = pp / PRW, rem = pp - s * PRW, pr = rem / p.PW, pc = rem - pr * p.PW;
        const int n = n0 + s, yy = S * y0 - PAD + pr, xx = pc - PAD;
        const bool ok = pp < p.nslots && n < p.N && yy >= 0 && yy < p.IH && xx >= 0 && xx < p.IW;
        voff[j] = ok ? (int)(((int64_t)n * p.xbs + yy * p.IW + xx) * 4) : (int)0x80000000;
        srow[j] = min(s, p.NS - 1) * p.I;
    }
    // this lane's output pixels (one per 32-pixel group) and the patch slot of their centre tap
    int centre[PT], on[PT], oy[PT], ox[PT];
#pragma unroll
    for (int g = 0; g < PT; ++g) {
        const int q = g * 32 + l31;
        const int s = p.NS == 1 ? 0 : q / p.HW, rem = q - s * p.HW, ry = rem / p.W, x = rem - ry * p.W;
        centre[g] = s * PRW + (S * ry + PAD) * p.PW + (S * x + PAD);
        on[g] = n0 + s; oy[g] = y0 + ry; ox[g] = x;
    }
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    bf16x8* Bw = smem + wn * WAVE_SLOTS;                                   // this wave's patch region

    // several accumulators per pixel group — one per product of the operand split (lo*hi, hi*lo, hi*hi; PT = 2: the two cross terms share
    // one, register budget) — so that consecutive MFMAs do not wait for each other's results (27 dependent MFMAs per chunk otherwise);
    // they are added once, cross terms first
    constexpr int NA = PT == 1 ? 3 : 2;
    f32x16 acc[PT][NA];
#pragma unroll
    for (int g = 0; g < PT; ++g)
#pragma unroll
        for (int k = 0; k < NA; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][k][r] = 0.f;

    // software pipeline over the wave's chunks: the activations of chunk c + 1 are requested as soon as chunk c's are converted, and every
    // tap's weight fragments are re-requested for chunk c + 1 right after the MFMAs that consumed them (same registers)
    float raw[NJ][16];
    bf16x8 ah[9], al[9];
    auto load_raw = [&](int c) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int ch = 0; ch < 16; ++ch)
                raw[j][ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, voff[j], (c * 16 + ch) * p.HWin * 4, 0));
    };
    const bf16x8* a0 = p.wt16 + (int64_t)half * p.OP64 + m0 + l31;
    const int64_t a_tap = (int64_t)KC * 4 * p.OP64, a_chunk = (int64_t)4 * p.OP64;
    load_raw(c_begin);
#pragma unroll
    for (int t = 0; t < 9; ++t) { ah[t] = a0[t * a_tap + c_begin * a_chunk]; al[t] = a0[t * a_tap + c_begin * a_chunk + 2 * p.OP64]; }
    __syncthreads();                                                       // s_style (behind the first chunk's requests: the styles' round trip overlaps theirs)

    for (int c = c_begin; c < c_begin + nc; ++c) {
        const bool more = c + 1 < c_begin + nc;
        // modulation + operand split, once per patch pixel, into the wave's own region (NJ patch pixels x 16 channels per lane)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float* st = s_style + srow[j] + c * 16;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                bf16x8 hi, lo;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float v = raw[j][hf * 8 + k] * st[hf * 8 + k];
                    const __bf16 h = (__bf16)v;
                    hi[k] = h;
                    lo[k] = (__bf16)(v - (float)h);
                }
                Bw[(0 * 2 + hf) * SLOTS + lane + 64 * j] = hi;
                Bw[(1 * 2 + hf) * SLOTS + lane + 64 * j] = lo;
            }
        }
        if (more) load_raw(c + 1);
        // 9 taps: shifted pixel fragments from the patch (LDS operations of one wave complete in order: no barrier)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int off = (t / 3 - PAD) * p.PW + (t % 3 - PAD);
#pragma unroll
            for (int g = 0; g < PT; ++g) {
                const bf16x8 bh = Bw[(0 * 2 + half) * SLOTS + centre[g] + off];
                const bf16x8 bl = Bw[(1 * 2 + half) * SLOTS + centre[g] + off];
                acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], bh, acc[g][0], 0, 0, 0);
                acc[g][NA - 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bl, acc[g][NA - 2], 0, 0, 0);
                acc[g][NA - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh, acc[g][NA - 1], 0, 0, 0);
            }
            if (more) { ah[t] = a0[t * a_tap + (c + 1) * a_chunk]; al[t] = a0[t * a_tap + (c + 1) * a_chunk + 2 * p.OP64]; }
        }
    }

    // the eight partial sums through LDS, summed in wave order (C/D layout: col = lane & 31 = pixel, row = (r&3) + 8*(r>>2) + 4*half = channel).  Wave w owns the
    // items (g, r) = w * VW .. w * VW + VW - 1 of the PT x 16 — VW = 2 PT consecutive registers r = consecutive channels: its share of a slab is ONE 8- / 16-byte store per lane
    __syncthreads();                                                       // every wave is done with its patch region
    float* red = reinterpret_cast<float*>(smem);                           // [wave][group][r][lane]: PT x 32 KB
#pragma unroll
    for (int g = 0; g < PT; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wn * PT + g) * 16 + r) * 64 + lane] = NA == 3 ? (acc[g][0][r] + acc[g][1][r]) + acc[g][NA - 1][r] : acc[g][0][r] + acc[g][NA - 1][r];
    __syncthreads();
    constexpr int VW = 2 * PT;
    const int g_w = (wn * VW) >> 4, r_w = (wn * VW) & 15;                  // wave-uniform
    float v[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) {
        v[k] = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v[k] += red[((w * PT + g_w) * 16 + r_w + k) * 64 + lane];
    }
    if (p.KS > 1) {                                                        // this slice's sums -> its slab; the last-arriving slice of the tile goes on
        constexpr int SLAB = PT * 16 * 64;
        const int units = p.tiles_p * p.tiles_m;
        const __amdgpu_buffer_rsrc_t r_slab = __builtin_amdgcn_make_buffer_rsrc((void*)p.partial, 0, units * p.KS * SLAB * 4, 0x00020000);
        const int mine = (tile * p.tiles_m + mt_i) * p.KS * SLAB + (wn * 64 + lane) * VW;
        sk_slab_store<VW>(r_slab, mine + ks * SLAB, v);
        if (!sk_arrive_last(p.tickets + tile * p.tiles_m + mt_i, p.KS, reinterpret_cast<unsigned*>(s_style))) return;
#pragma unroll
        for (int k = 0; k < VW; ++k) v[k] = 0.f;
        for (int k2 = 0; k2 < p.KS; ++k2) sk_slab_add<VW>(r_slab, mine + k2 * SLAB, v);        // slice order: reproducible whatever the arrival order
    }
    int n_, oy_, ox_;                                                      // (g_w is a run-time value: select instead of indexing registers)
    n_ = on[0]; oy_ = oy[0]; ox_ = ox[0];
#pragma unroll
    for (int gg = 1; gg < PT; ++gg)
        if (g_w == gg) { n_ = on[gg]; oy_ = oy[gg]; ox_ = ox[gg]; }
    if (n_ >= p.N) return;
#pragma unroll
    for (int k = 0; k < VW; ++k) {
        const int r = r_w + k, o = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (o >= p.O) continue;
        p.y[(int64_t)n_ * p.ybs + ((int64_t)o * p.H + oy_) * p.yrs + ox_] = n3d_apply_epilogue(v[k], p.epi, n_, o, p.O, oy_, ox_, p.H, p.W);
    }
}

// K slices over workgroups for a launch of `units` = (pixel tiles x channel tiles) single-slice workgroups: double while the grid stays within
// ~one workgroup per CU and every wave keeps at least one 16-channel chunk.  No arrival counters (desc.tickets) -> 1: the whole K inside the workgroup.
constexpr int SK_MAX_TICKETS = 4096;
static int sk_slices(int units, int KC, int target) {
    int ks = 1;
    const int forced = n3d_tune("N3D_SK_KS", 0);
    if (forced > 0) return (KC % (8 * forced) == 0) ? forced : 1;
    target = n3d_tune("N3D_SK_TARGET", target);
    while (units * ks * 2 <= target && KC % (16 * ks) == 0) ks *= 2;
    return ks;
}

// tile plan; false when the layer is not this kernel's
static bool sk_plan(int N, int I, int O, int H, int W, SkParams* out, int* pt_out, bool seam = false) {
    static const bool enabled = n3d_tune("N3D_CONV_SK", 1) != 0;
    if (!enabled) return false;
    if (N < 1 || I % 128 != 0 || I > 1024 || O % 32 != 0 || O < 32 || H < 2 || W < 2 || W > 32) return false;
    const int HW = H * W;
    const int64_t px = (int64_t)N * HW;
    if (px > 2048 || HW > 1024) return false;                              // every pixel tile re-reads the layer's weights from L2
    if (I > 512 && px > n3d_tune("N3D_SK_WIDE_PX", 256)) return false;                                 // (measured: 61 against 45 us at I = 1024, 16 x 16 x 4: 8 chunks per wave in sequence)
    int best = 0;
    SkParams bp;
    for (int pt = 2; pt >= 1; --pt) {
        const int TP = 32 * pt;
        SkParams q;
        if (HW >= TP) { if (TP % W != 0 || HW % TP != 0) continue; q.NS = 1; q.R = TP / W; q.tps = HW / TP; q.tiles_p = N * q.tps; }
        else { if (TP % HW != 0) continue; q.NS = TP / HW; q.R = H; q.tps = 1; q.tiles_p = (N + q.NS - 1) / q.NS; }
        if (q.NS > 2) continue;                                            // s_style holds two samples' styles
        q.PR = q.R + 2; q.PW = W + 2; q.nslots = q.NS * q.PR * q.PW;
        if (q.nslots > SK_SLOTS) continue;
        // 64-pixel tiles (every weight fragment feeds two pixel groups) once they still give the chip a workgroup per CU
        const int pt_force = n3d_tune("N3D_SK_PT", 0);
        if (pt_force && pt != pt_force) continue;
        if (!pt_force && pt == 2 && (int64_t)q.tiles_p * (O / 32) < 256) continue;       // (with K slices too: measured, profiles/r06_sk_seam_sweep.txt)
        best = pt; bp = q;
        break;
    }
    if (!best) return false;
    if (out) { out->NS = bp.NS; out->R = bp.R; out->tps = bp.tps; out->tiles_p = bp.tiles_p; out->PR = bp.PR; out->PW = bp.PW; out->nslots = bp.nslots; }
    if (pt_out) *pt_out = best;
    return true;
}

extern "C" int n3d_conv2d_sk_eligible(int N, int I, int O, int H, int W) { return sk_plan(N, I, O, H, W, nullptr, nullptr) ? 1 : 0; }

// The K slices of one launch, from what the descriptor offers: arrival counters for every output tile (desc.tickets / ticket_count) and a slab
// workspace of units * KS * slab floats (desc.workspace, sized by n3d_conv2d_sk_workspace).  Without counters: KS = 1, exactly the round-5 launch.
static void sk_seam(const n3d_conv2d_desc* d, int units, int slab_floats, int target, int* KS, float** partial, unsigned** tickets) {
    *KS = 1; *partial = nullptr; *tickets = nullptr;
    if (!d->tickets || !d->workspace || units > d->ticket_count || units > SK_MAX_TICKETS) return;
    const int ks = sk_slices(units, d->I / 16, target);
    if (ks <= 1) return;
    *KS = ks; *partial = d->workspace; *tickets = d->tickets;
}

// stride 2 (mode 1): input IH x IW = (2 OH + 1) x (2 OW + 1), 32-pixel output tiles of R rows (or two whole samples), patch <= 192 input pixels
static bool sk_s2_plan(int N, int I, int O, int IH, int IW, SkParams* out) {
    static const bool enabled = n3d_tune("N3D_CONV_SK_S2", 1) != 0;
    if (!enabled) return false;
    if (N < 1 || I % 128 != 0 || I > 512 || O % 32 != 0 || O < 32 || IH < 5 || IW < 5 || (IH & 1) == 0 || (IW & 1) == 0) return false;
    const int H = (IH - 3) / 2 + 1, W = (IW - 3) / 2 + 1, HW = H * W;
    if (W > 16 || (int64_t)N * HW > 1024) return false;                    // (every pixel tile re-reads the layer's weights from L2)
    SkParams q;
    const int TP = 32;
    if (HW >= TP) { if (TP % W != 0 || HW % TP != 0) return false; q.NS = 1; q.R = TP / W; q.tps = HW / TP; q.tiles_p = N * q.tps; }
    else { if (TP % HW != 0) return false; q.NS = TP / HW; q.R = H; q.tps = 1; q.tiles_p = (N + q.NS - 1) / q.NS; }
    if (q.NS > 2) return false;
    q.PR = 2 * q.R + 1; q.PW = 2 * W + 1; q.nslots = q.NS * q.PR * q.PW;
    if (q.nslots > 192) return false;
    if (out) { out->NS = q.NS; out->R = q.R; out->tps = q.tps; out->tiles_p = q.tiles_p; out->PR = q.PR; out->PW = q.PW; out->nslots = q.nslots; out->H = H; out->W = W; out->HW = HW; }
    return true;
}
/* 1 when n3d_conv2d_bf16x3 (ksize 3, mode 1, float32 NCHW in / out, dense input rows) runs this stride-2 layer on the one-launch few-pixel kernel: ksplit / workspace ignored */
extern "C" int n3d_conv2d_sk_s2_eligible(int N, int I, int O, int H, int W) { return sk_s2_plan(N, I, O, H, W, nullptr) ? 1 : 0; }

// called by the stride-2 launcher (conv2d_s2_bf16x3.hip) first; returns 1 when the layer is not this kernel's
int conv2d_sk_s2_bf16x3_try_launch(const n3d_conv2d_desc* d, hipStream_t stream) {
    SkParams p;
    if (d->epi.round_f16 || d->epi.residual_up_filter || d->side_split8 || d->y_layout != N3D_LAYOUT_NCHW_F32 || d->x_layout != N3D_LAYOUT_NCHW_F32 || d->wt_batch_stride) return 1;
    if (d->x_row_stride != 0 && d->x_row_stride != d->W) return 1;        // (pitched FIR output: the general stride-2 kernel)
    if (!sk_s2_plan(d->N, d->I, d->O, d->H, d->W, &p)) return 1;
    const int64_t xbs = d->x_batch_stride;
    const int64_t x_bytes = ((int64_t)(d->N - 1) * xbs + (int64_t)d->I * d->H * d->W) * 4;
    if (x_bytes >= (1ll << 31)) return 1;
    p.x = d->x; p.wt16 = (const bf16x8*)d->wt; p.style = d->style; p.y = d->y;
    p.N = d->N; p.I = d->I; p.O = d->O; p.OP64 = (d->O + 63) / 64 * 64;
    p.IH = d->H; p.IW = d->W; p.HWin = d->H * d->W;
    p.tiles_m = d->O / 32;
    p.x_bytes = (int)x_bytes;
    p.xbs = xbs; p.ybs = d->y_batch_stride ? d->y_batch_stride : (int64_t)d->O * p.HW;
    p.yrs = d->y_row_stride ? d->y_row_stride : p.W;
    N3D_CHECK(p.yrs >= p.W, "conv2d_bf16x3: y_row_stride smaller than the output width");
    p.style_stride = d->style_stride ? d->style_stride : d->I;
    p.epi = d->epi;
    sk_seam(d, p.tiles_p * p.tiles_m, 1 * 16 * 64, 256, &p.KS, &p.partial, &p.tickets);
    const double flops = 2.0 * d->N * (double)d->O * d->I * 9 * (double)p.HW;
    const double bytes = 4.0 * ((double)d->N * d->I * p.HWin + (double)d->N * d->O * p.HW + (double)d->O * d->I * 9);
    N3dProfScope prof(N3D_K_CONV2D_BF16X3, stream, flops, bytes);
    hipLaunchKernelGGL((conv2d_sk_bf16x3_kernel<1, 2>), dim3((unsigned)(p.tiles_p * p.tiles_m * p.KS)), dim3(512), 0, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}

// called by n3d_conv2d_bf16x3 for ksize 3 / mode 0 / float32 NCHW in and out (common fields validated there); returns 1 when the layer is
// not this kernel's (the caller goes on to the general kernels), 0 after a launch, -1 on error
int conv2d_sk_bf16x3_try_launch(const n3d_conv2d_desc* d, hipStream_t stream) {
    SkParams p;
    int pt;
    if (d->epi.round_f16 || d->epi.residual_up_filter || d->side_split8 || d->y_layout != N3D_LAYOUT_NCHW_F32) return 1;
    if (!sk_plan(d->N, d->I, d->O, d->H, d->W, &p, &pt, d->tickets && d->workspace)) return 1;
    const int64_t xbs = d->x_batch_stride;                                 // taken literally, like the other float32-input kernels: 0 = one image for the whole batch (the learned constant, expand()ed)
    const int64_t x_bytes = ((int64_t)(d->N - 1) * xbs + (int64_t)d->I * d->H * d->W) * 4;
    if (x_bytes >= (1ll << 31)) return 1;
    p.x = d->x; p.wt16 = (const bf16x8*)d->wt; p.style = d->style; p.y = d->y;
    p.N = d->N; p.I = d->I; p.O = d->O; p.OP64 = (d->O + 63) / 64 * 64; p.H = d->H; p.W = d->W; p.HW = d->H * d->W;
    p.IH = d->H; p.IW = d->W; p.HWin = p.HW;
    p.tiles_m = d->O / 32;
    p.x_bytes = (int)x_bytes;
    p.xbs = xbs; p.ybs = d->y_batch_stride ? d->y_batch_stride : (int64_t)d->O * d->H * d->W;
    p.yrs = d->y_row_stride ? d->y_row_stride : d->W;
    N3D_CHECK(p.yrs >= d->W, "conv2d_bf16x3: y_row_stride smaller than the output width");
    p.style_stride = d->style_stride ? d->style_stride : d->I;
    p.epi = d->epi;
    sk_seam(d, p.tiles_p * p.tiles_m, pt * 16 * 64, 256, &p.KS, &p.partial, &p.tickets);
    const double flops = 2.0 * d->N * (double)d->O * d->I * 9 * (double)d->H * d->W;
    const double bytes = 4.0 * ((double)d->N * d->I * p.HW + (double)d->N * d->O * p.HW + (double)d->O * d->I * 9);
    N3dProfScope prof(N3D_K_CONV2D_BF16X3, stream, flops, bytes);
    const dim3 grid((unsigned)(p.tiles_p * p.tiles_m * p.KS));
    if (pt == 2) hipLaunchKernelGGL(conv2d_sk_bf16x3_kernel<2>, grid, dim3(512), 0, stream, p);
    else hipLaunchKernelGGL(conv2d_sk_bf16x3_kernel<1>, grid, dim3(512), 0, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// The TRANSPOSED twin (round 4): 3x3 stride-2 transposed convolution of the few-position layers (4 x 4 and 8 x 8 inputs at batch 4, up to 16 x 16 at
// batch 1: up to 512 positions in the batch) in ONE launch — conv_transpose2d inside the up-sampling SynthesisLayer (conv2d_resample.py:114-127) as conv2d_up_bf16x3_kernel
// computes it: over the (H+1) x (W+1) POSITION grid, tap (ky, kx) feeds output phase (ky == 1, kx == 1) of position (gy, gx) = output pixel
// (2 gy + pa, 2 gx + pb) from input pixel (gy - 1 + dy, gx - 1 + dx), dy = (ky != 2), dx = (kx != 2).  The general kernel splits K over 4-8
// workgroups and needs a reduce + epilogue launch (conv16_splitk_epilogue_kernel) — 6 such pairs per forward at batch 4, 9 at batch 1; here, as
// above, a workgroup owns 32 channels x 32 consecutive positions (flattened row-major over the position grid) x 4 phases for the whole K, its 8
// waves split the 16-channel chunks, stage their own patch (the rows the 32 positions touch + one row above, W + 2 columns), and the partial sums
// meet in LDS two phases at a time.
struct SkUpParams {
    const float* x; const bf16x8* wt16; const float* style; float* y;
    int N, I, O, OP64, H, W, HW, GW, P, OH, OW;      // GW = W + 1 position columns, P = (H+1) * (W+1) positions per sample
    int PW, tps, tiles_m;                            // patch pitch W + 2, position tiles per sample
    int x_bytes;
    int64_t xbs, ybs, yrs, style_stride;
    int KS;                                          // K slices over workgroups, as SkParams
    float* partial; unsigned* tickets;               // slabs [tile][channel tile][slice][4 phases * 16 items][64 lanes]
    n3d_epilogue epi;
};

__global__ __launch_bounds__(512) void conv2d_up_sk_bf16x3_kernel(SkUpParams p) {
    __shared__ bf16x8 smem[8 * SK_WAVE_SLOTS];                             // 64 KB: the waves' patch regions; afterwards the partial sums of two phases
    __shared__ float s_style[1024];
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int mt_i, tile, ks;
    {
        const int b = blockIdx.x, units = p.tiles_m * p.KS;
        int u;
        if ((units & 7) == 0) { const int per = units >> 3, q = b >> 3; u = (b & 7) + 8 * (q % per); tile = q / per; }
        else { u = b % units; tile = b / units; }
        mt_i = u % p.tiles_m; ks = u / p.tiles_m;
    }
    const int m0 = mt_i * 32;
    const int n = tile / p.tps, q0 = (tile % p.tps) * 32;                  // sample, first flattened position of the tile
    const int row0 = q0 / p.GW;                                            // first position row: patch row 0 = input row row0 - 1
    const int KC = p.I / 16, nc = KC / (8 * p.KS), c_begin = (ks * 8 + wn) * nc;
    for (int i = tid; i < p.I; i += 512) s_style[i] = p.style ? p.style[(int64_t)n * p.style_stride + i] : 1.f;

    const int rows = min(q0 + 31, p.P - 1) / p.GW - row0 + 2, nslots = rows * p.PW;       // patch rows: the position rows + one
    int voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pp = lane + 64 * j, pr = pp / p.PW, pc = pp - pr * p.PW;
        const int yy = row0 - 1 + pr, xx = pc - 1;
        const bool ok = pp < nslots && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        voff[j] = ok ? (int)(((int64_t)n * p.xbs + yy * p.W + xx) * 4) : (int)0x80000000;
    }
    const int q = q0 + l31;                                                // this lane's position
    const bool q_act = q < p.P;
    const int gy = (q_act ? q : q0) / p.GW, gx = (q_act ? q : q0) - gy * p.GW;
    const int base = (gy - row0) * p.PW + gx;                              // patch slot of input pixel (gy - 1, gx - 1)
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    bf16x8* Bw = smem + wn * SK_WAVE_SLOTS;

    f32x16 acc[4];                                                         // one accumulator per output phase
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ph][r] = 0.f;

    float raw[2][16];
    bf16x8 ah[9], al[9];
    auto load_raw = [&](int c) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ch = 0; ch < 16; ++ch)
                raw[j][ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, voff[j], (c * 16 + ch) * p.HW * 4, 0));
    };
    const bf16x8* a0 = p.wt16 + (int64_t)half * p.OP64 + m0 + l31;
    const int64_t a_tap = (int64_t)KC * 4 * p.OP64, a_chunk = (int64_t)4 * p.OP64;
    load_raw(c_begin);
#pragma unroll
    for (int t = 0; t < 9; ++t) { ah[t] = a0[t * a_tap + c_begin * a_chunk]; al[t] = a0[t * a_tap + c_begin * a_chunk + 2 * p.OP64]; }
    __syncthreads();                                                       // s_style (behind the first chunk's requests)

    for (int c = c_begin; c < c_begin + nc; ++c) {
        const bool more = c + 1 < c_begin + nc;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* st = s_style + c * 16;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                bf16x8 hi, lo;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float v = raw[j][hf * 8 + k] * st[hf * 8 + k];
                    const __bf16 h = (__bf16)v;
                    hi[k] = h;
                    lo[k] = (__bf16)(v - (float)h);
                }
                Bw[(0 * 2 + hf) * SK_SLOTS + lane + 64 * j] = hi;
                Bw[(1 * 2 + hf) * SK_SLOTS + lane + 64 * j] = lo;
            }
        }
        if (more) load_raw(c + 1);
        bf16x8 bh[4], bl[4];                                               // the position's 2 x 2 input pixels (LDS operations of one wave complete in order)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            bh[d] = Bw[(0 * 2 + half) * SK_SLOTS + base + (d >> 1) * p.PW + (d & 1)];
            bl[d] = Bw[(1 * 2 + half) * SK_SLOTS + base + (d >> 1) * p.PW + (d & 1)];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t % 3;
            const int ph = (ky == 1 ? 2 : 0) + (kx == 1 ? 1 : 0), d = (ky == 2 ? 0 : 2) + (kx == 2 ? 0 : 1);
            acc[ph] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], bh[d], acc[ph], 0, 0, 0);
            acc[ph] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bl[d], acc[ph], 0, 0, 0);
            acc[ph] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh[d], acc[ph], 0, 0, 0);
            if (more) { ah[t] = a0[t * a_tap + (c + 1) * a_chunk]; al[t] = a0[t * a_tap + (c + 1) * a_chunk + 2 * p.OP64]; }
        }
    }

    // the eight partial sums through LDS in wave order, two phases (one output row pair... the phases pa = 0, then pa = 1) at a time: 8 x 2 x 16 x 64 floats = 64 KB.
    // Wave w owns (pb, r) = (w >> 2, 4 (w & 3) .. + 3) of each round: four consecutive channels = one 16-byte piece of a slab per lane and round
    float* red = reinterpret_cast<float*>(smem);
    const int pb_w = wn >> 2, r_w = 4 * (wn & 3);
    constexpr int SLAB = 4 * 16 * 64;
    const __amdgpu_buffer_rsrc_t r_slab = __builtin_amdgcn_make_buffer_rsrc((void*)p.partial, 0, p.KS > 1 ? p.N * p.tps * p.tiles_m * p.KS * SLAB * 4 : 0, 0x00020000);
    const int mine = (tile * p.tiles_m + mt_i) * p.KS * SLAB + (wn * 64 + lane) * 4;          // + slice * SLAB + pa * (SLAB / 2)
    float v[2][4];
#pragma unroll
    for (int pa = 0; pa < 2; ++pa) {
        __syncthreads();                                                   // the patch regions (pa = 0) / the previous round's sums (pa = 1) are dead
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wn * 2 + pb) * 16 + r) * 64 + lane] = acc[pa * 2 + pb][r];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[pa][k] = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v[pa][k] += red[((w * 2 + pb_w) * 16 + r_w + k) * 64 + lane];
        }
        if (p.KS > 1) sk_slab_store<4>(r_slab, mine + ks * SLAB + pa * (SLAB / 2), v[pa]);
    }
    if (p.KS > 1) {
        if (!sk_arrive_last(p.tickets + tile * p.tiles_m + mt_i, p.KS, reinterpret_cast<unsigned*>(s_style))) return;
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[pa][k] = 0.f;
            for (int k2 = 0; k2 < p.KS; ++k2) sk_slab_add<4>(r_slab, mine + k2 * SLAB + pa * (SLAB / 2), v[pa]);     // slice order: reproducible
        }
    }
    if (!q_act) return;
#pragma unroll
    for (int pa = 0; pa < 2; ++pa)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r_w + k, o = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int oy = 2 * gy + pa, ox = 2 * gx + pb_w;
            if (o >= p.O || oy >= p.OH || ox >= p.OW) continue;
            p.y[(int64_t)n * p.ybs + ((int64_t)o * p.OH + oy) * p.yrs + ox] = n3d_apply_epilogue(v[pa][k], p.epi, n, o, p.O, oy, ox, p.OH, p.OW);
        }
}

static bool sk_up_plan(int N, int I, int O, int H, int W) {
    if (n3d_tune("N3D_CONV_SK", 1) == 0 || n3d_tune("N3D_CONV_UP_SK", 1) == 0) return false;      // (tuning builds: A/B)
    if (N < 1 || I % 128 != 0 || I > 1024 || O % 32 != 0 || O < 32 || H < 2 || W < 2 || W > 32) return false;
    const int P = (H + 1) * (W + 1);
    // every 32-position tile re-reads the layer's weights from L2: measured (layer trace inside a forward) 4 x 4 at batch 4 53 -> 44 us, 8 x 8 56 -> 41,
    // 16 x 16 at batch 1 61 -> 42 — but 16 x 16 at batch 4 (1156 positions) 77 us against 62 for the pre-split kernel, 32 x 32 at batch 1 76 against 68
    if ((int64_t)N * P > n3d_tune("N3D_SK_UP_MAXP", 512)) return false;
    // the patch of 32 consecutive positions: at most 32 / (W+1) + 2 position rows + 1, W + 2 wide
    const int rows = (32 + W) / (W + 1) + 2;
    return rows * (W + 2) <= SK_SLOTS;
}

extern "C" int n3d_conv2d_up_sk_eligible(int N, int I, int O, int H, int W) { return sk_up_plan(N, I, O, H, W) ? 1 : 0; }

// called by n3d_conv2d_bf16x3 for ksize 3 / mode 2 / float32 NCHW in and out, shared weights; 1 = not this kernel's layer, 0 = launched, -1 = error
int conv2d_up_sk_bf16x3_try_launch(const n3d_conv2d_desc* d, hipStream_t stream) {
    if (d->epi.round_f16 || d->epi.residual || d->epi.residual_up_filter || d->side_split8 || d->y_layout != N3D_LAYOUT_NCHW_F32 || d->wt_batch_stride) return 1;
    if (!sk_up_plan(d->N, d->I, d->O, d->H, d->W)) return 1;
    const int64_t xbs = d->x_batch_stride;
    const int64_t x_bytes = ((int64_t)(d->N - 1) * xbs + (int64_t)d->I * d->H * d->W) * 4;
    if (x_bytes >= (1ll << 31)) return 1;
    SkUpParams p;
    p.x = d->x; p.wt16 = (const bf16x8*)d->wt; p.style = d->style; p.y = d->y;
    p.N = d->N; p.I = d->I; p.O = d->O; p.OP64 = (d->O + 63) / 64 * 64; p.H = d->H; p.W = d->W; p.HW = d->H * d->W;
    p.GW = d->W + 1; p.P = (d->H + 1) * (d->W + 1); p.OH = 2 * d->H + 1; p.OW = 2 * d->W + 1;
    p.PW = d->W + 2; p.tps = (p.P + 31) / 32; p.tiles_m = d->O / 32;
    p.x_bytes = (int)x_bytes;
    p.xbs = xbs; p.ybs = d->y_batch_stride ? d->y_batch_stride : (int64_t)d->O * p.OH * p.OW;
    p.yrs = d->y_row_stride ? d->y_row_stride : p.OW;
    N3D_CHECK(p.yrs >= p.OW, "conv2d_bf16x3: y_row_stride smaller than the output width");
    p.style_stride = d->style_stride ? d->style_stride : d->I;
    p.epi = d->epi;
    sk_seam(d, d->N * p.tps * p.tiles_m, 4 * 16 * 64, 256, &p.KS, &p.partial, &p.tickets);
    const double flops = 2.0 * d->N * (double)d->O * d->I * 9 * (double)d->H * d->W;
    const double bytes = 4.0 * ((double)d->N * d->I * p.HW + (double)d->N * d->O * p.OH * p.OW + (double)d->O * d->I * 9);
    N3dProfScope prof(N3D_K_CONV2D_BF16X3, stream, flops, bytes);
    hipLaunchKernelGGL(conv2d_up_sk_bf16x3_kernel, dim3((unsigned)(d->N * p.tps * p.tiles_m * p.KS)), dim3(512), 0, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}

// Floats of slab workspace (desc.workspace) the few-pixel kernels use for this layer when the descriptor carries arrival counters (desc.tickets), and
// the counters it needs (*tickets_needed): 0 / 0 = the layer is not theirs, or runs with the whole K inside one workgroup.  mode 0 / 1 / 2 as
// n3d_conv2d_desc.mode (mode 1: H x W = the input image).
extern "C" int64_t n3d_conv2d_sk_workspace(int N, int I, int O, int H, int W, int mode, int* tickets_needed) {
    if (tickets_needed) *tickets_needed = 0;
    int units = 0, slab = 0;
    SkParams p; int pt;
    if (mode == 0) { if (!sk_plan(N, I, O, H, W, &p, &pt, true)) return 0; units = p.tiles_p * (O / 32); slab = pt * 16 * 64; }
    else if (mode == 1) { if (!sk_s2_plan(N, I, O, H, W, &p)) return 0; units = p.tiles_p * (O / 32); slab = 16 * 64; }
    else if (mode == 2) { if (!sk_up_plan(N, I, O, H, W)) return 0; units = N * (((H + 1) * (W + 1) + 31) / 32) * (O / 32); slab = 4 * 16 * 64; }
    else return 0;
    if (units > SK_MAX_TICKETS) return 0;
    const int ks = sk_slices(units, I / 16, 256);
    if (ks <= 1) return 0;
    if (tickets_needed) *tickets_needed = units;
    return (int64_t)units * ks * slab;
}
