// upfirdn2d for gfx950: zero-insert upsample -> pad/crop -> 2-D FIR -> decimate, per (n,c) plane, with an
// optional fused epilogue (noise + bias + activation + clamp + residual) so that the up-sampling
// StyleGAN2 layer needs one pass over its activation instead of three.
// HBM-bound (algorithmic bytes = input + output): each workgroup produces a TILE_H x TILE_W output tile of
// one plane from an LDS-staged input footprint, so every input element is fetched from HBM once per tile
// and the fh*fw taps are served from LDS; rows are read/written as contiguous 64-lane segments.
// Replaces upfirdn2d_plugin.upfirdn2d (reference torch_utils/ops/upfirdn2d.cpp:20, upfirdn2d.cu:33-207);
// index arithmetic follows upfirdn2d.cu:47-67: out[o] = sum_i x[i] * f[fw-1-(i*up+pad0-o*down)] (flip: f[k]).
#include "common.h"

#define UF_TILE_W 64
#define UF_TILE_H 8
#define UF_MAX_TAPS 1024
#define UF_MAX_FOOT 4608   // floats of LDS input footprint per tile

struct UfParams {
    const float* x; const float* f; float* y;
    int N, C, H, W, OH, OW, fh, fw, upx, upy, downx, downy, padx0, pady0, flip;
    float gain;
    int64_t xbs, ybs, xrs, yrs;               // xrs / yrs: input / output row pitch in floats (plane pitch = rows * pitch)
    int has_epi;
    n3d_epilogue epi;
    int tiles_x, tiles_y, foot_w, foot_h;
};

__device__ __forceinline__ int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
__device__ __forceinline__ int ceil_div(int a, int b) { return -floor_div(-a, b); }

__global__ __launch_bounds__(256) void upfirdn2d_kernel(UfParams p) {
    __shared__ float s_f[UF_MAX_TAPS];
    __shared__ float s_x[UF_MAX_FOOT];
    const int tile = blockIdx.x;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int c = blockIdx.y, n = blockIdx.z;
    const int ox0 = tx * UF_TILE_W, oy0 = ty * UF_TILE_H;

    // taps, pre-flipped so that the inner loop is a plain correlation over the (virtual) upsampled+padded signal
    for (int t = threadIdx.x; t < p.fh * p.fw; t += blockDim.x) {
        const int ky = t / p.fw, kx = t % p.fw;
        s_f[t] = p.flip ? p.f[t] : p.f[(p.fh - 1 - ky) * p.fw + (p.fw - 1 - kx)];
    }
    // input footprint of this tile: rows iy in [iy_lo, iy_lo + foot_h), cols likewise
    const int ix_lo = ceil_div(ox0 * p.downx - p.padx0, p.upx);
    const int iy_lo = ceil_div(oy0 * p.downy - p.pady0, p.upy);
    const float* xp = p.x + (int64_t)n * p.xbs + (int64_t)c * p.H * p.xrs;
    for (int e = threadIdx.x; e < p.foot_h * p.foot_w; e += blockDim.x) {
        const int r = e / p.foot_w, q = e % p.foot_w;
        const int iy = iy_lo + r, ix = ix_lo + q;
        s_x[e] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? xp[(int64_t)iy * p.xrs + ix] : 0.f;
    }
    __syncthreads();

    float* yp = p.y + (int64_t)n * p.ybs + (int64_t)c * p.OH * p.yrs;
    for (int e = threadIdx.x; e < UF_TILE_H * UF_TILE_W; e += blockDim.x) {
        const int oy = oy0 + e / UF_TILE_W, ox = ox0 + e % UF_TILE_W;
        if (oy >= p.OH || ox >= p.OW) continue;
        // padded/upsampled coordinate q = i*up + pad0 ; window [o*down, o*down + f - 1]
        const int qy0 = oy * p.downy, qx0 = ox * p.downx;
        const int iy_a = ceil_div(qy0 - p.pady0, p.upy), iy_b = floor_div(qy0 + p.fh - 1 - p.pady0, p.upy);
        const int ix_a = ceil_div(qx0 - p.padx0, p.upx), ix_b = floor_div(qx0 + p.fw - 1 - p.padx0, p.upx);
        float v = 0.f;
        for (int iy = iy_a; iy <= iy_b; ++iy) {
            const int ky = iy * p.upy + p.pady0 - qy0;
            const float* frow = s_f + ky * p.fw;
            const float* xrow = s_x + (iy - iy_lo) * p.foot_w - ix_lo;
            for (int ix = ix_a; ix <= ix_b; ++ix) v += xrow[ix] * frow[ix * p.upx + p.padx0 - qx0];
        }
        v *= p.gain;
        if (p.has_epi) v = n3d_apply_epilogue(v, p.epi, n, c, p.C, oy, ox, p.OH, p.OW);
        yp[(int64_t)oy * p.yrs + ox] = v;
    }
}


// ---- specialised path: compile-time up/down factors and filter size (the generator only uses the 4x4 [1,3,3,1] filter
// with (up,down) in {(1,1),(2,1),(1,2)}).  All index arithmetic folds to shifts/constants, every thread produces
// FT_PER_T outputs at one x (consecutive lanes = consecutive x: conflict-free LDS reads, coalesced 256-byte stores) and
// the taps live in registers.  HBM-bound: the block reads its input footprint once and writes its outputs once.
#define FT_W 64
#define FT_H 32
#define FT_PER_T (FT_W * FT_H / 256)

template <int UP, int DOWN, int FS>
__global__ __launch_bounds__(256) void upfirdn2d_fast_kernel(UfParams p) {
    constexpr int FOOT_W = ((FT_W - 1) * DOWN + FS - 1) / UP + 2;
    constexpr int FOOT_H = ((FT_H - 1) * DOWN + FS - 1) / UP + 2;
    __shared__ float s_x[FOOT_H * FOOT_W];
    const int tile = blockIdx.x;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int c = blockIdx.y, n = blockIdx.z;
    const int ox0 = tx * FT_W, oy0 = ty * FT_H;
    float f[FS][FS];                                     // pre-flipped taps (uniform -> scalar registers)
#pragma unroll
    for (int ky = 0; ky < FS; ++ky)
#pragma unroll
        for (int kx = 0; kx < FS; ++kx) f[ky][kx] = p.flip ? p.f[ky * FS + kx] : p.f[(FS - 1 - ky) * FS + (FS - 1 - kx)];
    const int ix_lo = ceil_div(ox0 * DOWN - p.padx0, UP);
    const int iy_lo = ceil_div(oy0 * DOWN - p.pady0, UP);
    const float* xp = p.x + (int64_t)n * p.xbs + (int64_t)c * p.H * p.xrs;
    for (int e = threadIdx.x; e < FOOT_H * FOOT_W; e += 256) {
        const int r = e / FOOT_W, q = e % FOOT_W;
        const int iy = iy_lo + r, ix = ix_lo + q;
        s_x[e] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? xp[(int64_t)iy * p.xrs + ix] : 0.f;
    }
    __syncthreads();
    float* yp = p.y + (int64_t)n * p.ybs + (int64_t)c * p.OH * p.yrs;
    const int lx = threadIdx.x % FT_W, ly0 = threadIdx.x / FT_W;
    const int ox = ox0 + lx;
    const int qx0 = ox * DOWN - p.padx0;                 // upsampled-domain coordinate of tap kx = 0
    const int kx0 = ((-qx0) % UP + UP) % UP;              // first tap that lands on a real (non-inserted-zero) sample
#pragma unroll
    for (int j = 0; j < FT_PER_T; ++j) {
        const int oy = oy0 + ly0 * FT_PER_T + j;            // consecutive rows per thread: the FS-row windows overlap,
        const int qy0 = oy * DOWN - p.pady0;                 // so the unrolled loop re-uses LDS reads across j
        const int ky0 = ((-qy0) % UP + UP) % UP;
        float v = 0.f;
#pragma unroll
        for (int a = 0; a < (FS + UP - 1) / UP; ++a) {
            const int ky = ky0 + a * UP;
            if (ky >= FS) continue;
            const int ry = (qy0 + ky) / UP - iy_lo;       // exact division
            const float* row = s_x + ry * FOOT_W - ix_lo;
#pragma unroll
            for (int b = 0; b < (FS + UP - 1) / UP; ++b) {
                const int kx = kx0 + b * UP;
                if (kx >= FS) continue;
                float w;
                if (UP == 1) w = f[a][b];
                else w = f[ky & (FS - 1)][kx & (FS - 1)];
                v += row[(qx0 + kx) / UP] * w;
            }
        }
        if (oy < p.OH && ox < p.OW) {
            v *= p.gain;
            if (p.has_epi) v = n3d_apply_epilogue(v, p.epi, n, c, p.C, oy, ox, p.OH, p.OW);
            yp[(int64_t)oy * p.yrs + ox] = v;
        }
    }
}

// ---- 16-byte path for the FIR that follows a transposed convolution (up = down = 1, 4x4 taps, pad 1): the input rows
// have a pitch that is a multiple of 4 floats and the output width is a multiple of 4, so the footprint is fetched with
// aligned global_load_dwordx4 -> ds_write_b128, every thread produces a 4-wide x RPT-high patch from ds_read_b128 rows
// (7 input floats per row for 4 outputs) and stores float4s.  Input columns beyond W (the pitch padding) are never used:
// they are replaced by zeros with a select, so uninitialised padding cannot leak NaNs.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TW, int PADX0>
__global__ __launch_bounds__(256) void fir4_vec_kernel(UfParams p) {
    constexpr int TH = 32, CG = TW / 4, RG = 256 / CG, RPT = TH / RG, FW4 = (TW + 8) / 4, FH = TH + 3;
    __shared__ f32x4 s_x[FH * FW4];
    const int tile = blockIdx.x;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int c = blockIdx.y, n = blockIdx.z;
    const int ox0 = tx * TW, oy0 = ty * TH;
    float f[4][4];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) f[ky][kx] = p.flip ? p.f[ky * 4 + kx] : p.f[(3 - ky) * 4 + (3 - kx)];
    const float* xp = p.x + (int64_t)n * p.xbs + (int64_t)c * p.H * p.xrs;
    const int iy_lo = oy0 - p.pady0;
    for (int e = threadIdx.x; e < FH * FW4; e += 256) {
        const int r = e / FW4, q = e % FW4;
        const int iy = iy_lo + r, col = ox0 - 4 + 4 * q;          // LDS column 0 = input column ox0 - 4
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < p.H && col >= 0 && col < p.W) {
            v = *reinterpret_cast<const f32x4*>(xp + (int64_t)iy * p.xrs + col);
            if (col + 1 >= p.W) v.y = 0.f;
            if (col + 2 >= p.W) v.z = 0.f;
            if (col + 3 >= p.W) v.w = 0.f;
        }
        s_x[e] = v;
    }
    __syncthreads();
    const int cg = threadIdx.x % CG, rg = threadIdx.x / CG;
    const int ox = ox0 + 4 * cg;
    float acc[RPT][4];
#pragma unroll
    for (int j = 0; j < RPT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
#pragma unroll
    for (int rr = 0; rr < RPT + 3; ++rr) {
        const f32x4* row = s_x + (rg * RPT + rr) * FW4 + cg;
        const f32x4 a = row[0], b = row[1], d = row[2];
        // input columns ox - PADX0 .. ox - PADX0 + 6
        const float in[7] = {PADX0 == 1 ? a.w : a.z, PADX0 == 1 ? b.x : a.w, PADX0 == 1 ? b.y : b.x, PADX0 == 1 ? b.z : b.y,
                             PADX0 == 1 ? b.w : b.z, PADX0 == 1 ? d.x : b.w, PADX0 == 1 ? d.y : d.x};
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int ky = rr - j;
            if (ky < 0 || ky > 3) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) acc[j][i] += in[i + kx] * f[ky][kx];
        }
    }
    if (ox >= p.OW) return;
    const n3d_epilogue& E = p.epi;
    float sc = 1.f, bias = 0.f, nstr = 0.f;
    if (p.has_epi) {
        sc = E.const_scale;
        if (E.row_scale) sc *= E.row_scale[(int64_t)n * (E.row_scale_stride ? E.row_scale_stride : p.C) + c];
        if (E.bias) bias = E.bias[c];
        if (E.noise) nstr = E.noise_strength[0];
    }
    float* yp = p.y + (int64_t)n * p.ybs + (int64_t)c * p.OH * p.yrs;     // (the pitch padding, if any, receives taps of zero padding)
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int oy = oy0 + rg * RPT + j;
        if (oy >= p.OH) continue;
        const int64_t po = (int64_t)oy * p.OW + ox;
        f32x4 v = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
        v *= p.gain;
        if (p.has_epi) {
            v *= sc;
            if (E.noise) v += *reinterpret_cast<const f32x4*>(E.noise + po) * nstr;
            v += bias;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = (E.act == N3D_ACT_LRELU) ? (v[i] > 0.f ? v[i] : v[i] * E.alpha) : n3d_act(v[i], E.act, E.alpha);
                t *= E.gain;
                if (E.clamp >= 0.f) t = fminf(fmaxf(t, -E.clamp), E.clamp);
                v[i] = t;
            }
            if (E.residual) v += *reinterpret_cast<const f32x4*>(E.residual + (int64_t)n * E.residual_batch_stride + (int64_t)c * p.OH * p.OW + po);
        }
        *reinterpret_cast<f32x4*>(yp + (int64_t)oy * p.yrs + ox) = v;
    }
}

// ---- the 4x4 FIR behind a transposed convolution (up = down = 1, pad 1) from the channel-interleaved "c8" layout to the
// "split8" layout (include/n3d.h) for the 3x3 convolution that follows.  The epilogue (noise, bias, leaky ReLU, gain, clamp) is
// applied as in fir4_vec_kernel, then the value is multiplied by the NEXT layer's style s[n][c] (modulation,
// tat/networks_stylegan2.py:70 `x * styles`), split into bf16 hi / lo halves and stored as 16-byte units of 8 consecutive
// channels per pixel:   y[n][hl][c/8][oy][ox][c%8],  hi = bf16(v), lo = bf16(v - hi)
// — exactly the operands conv2d_bf16x3_kernel builds for itself while staging (same multiply, same two conversions), so the
// consumer (conv2d_ps_bf16x3.hip) multiplies bit-identical bf16 pairs.
// Both layouts keep the 8 channels of a pixel together, so ONE work item owns a pixel column segment with all its channels:
// lane = pixel on the load side (32 contiguous bytes per lane), in LDS (two conflict-free 16-byte planes) and on the store
// side (one 16-byte unit per lane and plane, 1 KB contiguous per wave instruction).  Workgroup = 64 x 16 outputs x 8 channels,
// 512 work items, each 2 rows of one column (5 footprint rows shared by its 2 outputs), the two channel halves one after the other.  HBM-bound: input 4 B + output 4 B per element.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
struct FirSplitParams {
    const float* x; const float* f; bf16x8_t* y;
    int N, C, H, W, OH, OW, flip;
    float gain;
    int64_t xbs, xrs;             // batch stride in floats, row pitch in pixels
    const float* out_scale; int64_t out_scale_stride;
    int has_epi;
    n3d_epilogue epi;
    int tiles_x;
    int pad;                      // zero padding on every side: 1 (behind a transposed convolution) or 2 (in front of a stride-2 one)
};

// FAST: linear / leaky-ReLU (0 <= alpha <= 1) epilogues as straight-line code; the generic instantiation carries the full switch
// NCHW_IN: the input is float32 NCHW (the FIR in front of a stride-2 convolution: n3d_fir4_split8_nchw) instead of c8; only the
// footprint loads differ (4-byte loads, transposed into the same two LDS planes).
#ifndef FIR_TW
#define FIR_TW 64     // tile width: 64 (512 work items, 40.7 KB of LDS: 3 workgroups per CU) or 32 (256 work items, 21 KB: 7)
#endif
template <bool FAST, bool NCHW_IN>
__global__ __launch_bounds__(FIR_TW * 8, FIR_TW == 64 ? 4 : 8) void fir4_c8_split8_kernel(FirSplitParams p) {
    constexpr int TW = FIR_TW, NT = TW * 8, TH = 16, RPT = 2, FW = TW + 3, FH = TH + 3, CH = 8;
    static_assert(NT / (2 * FW) == 3, "the unit cursor below advances by three footprint rows + a remainder per step");
    __shared__ f32x4 s_ab[2 * FH * FW];                                   // plane 0: channels 0-3, plane 1: channels 4-7 of every footprint pixel
    f32x4* const s_a = s_ab; f32x4* const s_b = s_ab + FH * FW;
    const int tile = blockIdx.x;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int c8 = blockIdx.y, n = blockIdx.z;
    const int ox0 = tx * TW, oy0 = ty * TH;
    float f[4][4];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) f[ky][kx] = p.flip ? p.f[ky * 4 + kx] : p.f[(3 - ky) * 4 + (3 - kx)];
    // footprint -> LDS: 16-byte units, channel half fastest (consecutive lanes read consecutive bytes).  Buffer loads with a
    // 32-bit offset: units outside the image get an offset beyond the descriptor's range and read as zero (the FIR's padding)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.xbs + (int64_t)c8 * p.H * p.xrs * 8), 0,
                                                                          (int)(p.H * p.xrs * 32), 0x00020000);
    if (NCHW_IN && p.pad == 2 && ((p.W | p.xrs) & 1) == 0) {
        // even pad / width / pitch: the footprint rows start on even columns -> 8-byte loads of two neighbouring pixels
        constexpr int PW2 = (FW + 1) / 2, PAIRS = CH * FH * PW2, PPT = (PAIRS + NT - 1) / NT;      // 34 pairs per row, 5168 -> 11 per work item
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        f32x2_t st[PPT];
        int sl[PPT];
        float* s_f = reinterpret_cast<float*>(s_ab);
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int e = threadIdx.x + NT * j;
            const int ch = e / (FH * PW2), r = (e % (FH * PW2)) / PW2, q = 2 * (e % PW2);
            const int iy = oy0 - 2 + r, ix = ox0 - 2 + q;
            const bool ok = e < PAIRS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;      // ix even, W even: the pair is inside or outside as a whole
            const int off = ok ? ((ch * p.H + iy) * (int)p.xrs + ix) * 4 : (int)0x80000000;
            st[j] = __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0));
            sl[j] = e < PAIRS ? (((ch >> 2) * FH * FW + r * FW + q) * 4 + (ch & 3)) | (q + 1 < FW ? 0 : (int)0x40000000) : -1;
        }
#pragma unroll
        for (int j = 0; j < PPT; ++j)
            if (sl[j] >= 0) {
                const int a = sl[j] & 0x3fffffff;
                s_f[a] = st[j].x;
                if (!(sl[j] & 0x40000000)) s_f[a + 4] = st[j].y;        // the 68th column of a 67-wide footprint row does not exist
            }
    } else if (NCHW_IN) {
        // 8 channel planes of H x xrs floats: element e -> (channel, footprint row, column), column fastest (coalesced rows of 67
        // floats); written as single floats into the (pixel, channel) slots of the two planes.  All loads first, then the writes.
        constexpr int ELEMS = CH * FH * FW, EPT = (ELEMS + NT - 1) / NT;  // 10,184 -> 20 per work item
        float st[EPT];
        int sl[EPT];
        float* s_f = reinterpret_cast<float*>(s_ab);
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int e = threadIdx.x + NT * j;
            const int ch = e / (FH * FW), r = (e % (FH * FW)) / FW, q = e % FW;
            const int iy = oy0 - p.pad + r, ix = ox0 - p.pad + q;
            const bool ok = e < ELEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const int off = ok ? ((ch * p.H + iy) * (int)p.xrs + ix) * 4 : (int)0x80000000;
            st[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
            sl[j] = e < ELEMS ? (((ch >> 2) * FH * FW + r * FW + q) * 4 + (ch & 3)) : -1;
        }
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            if (sl[j] >= 0) s_f[sl[j]] = st[j];
    } else {
    constexpr int ROWU = FW * 2, UNITS = FH * ROWU, LPT = (UNITS + NT - 1) / NT;    // 134 units per row, 2546 per tile, 5 per work item
    f32x4 stage[LPT];
    int slot[LPT];                                                        // LDS slot (plane-relative) or -1
    {
        int r = threadIdx.x / ROWU, cu = threadIdx.x % ROWU;              // this work item's first unit: (row, unit in row)
#pragma unroll
        for (int j = 0; j < LPT; ++j) {                                   // all loads in flight before the first LDS write
            const int hf = cu & 1, q = cu >> 1;
            const int iy = oy0 - p.pad + r, ix = ox0 - p.pad + q;
            const bool in_tile = threadIdx.x + NT * j < UNITS;
            const bool ok = in_tile && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const int off = ok ? ((iy * (int)p.xrs + ix) * 8 + 4 * hf) * 4 : (int)0x80000000;
            stage[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
            slot[j] = in_tile ? (hf * FH * FW + r * FW + q) : -1;
            cu += NT - 3 * ROWU; r += 3;                                  // advance by 512 units = 3 rows + 110 units
            if (cu >= ROWU) { cu -= ROWU; r += 1; }
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j)
            if (slot[j] >= 0) s_a[slot[j]] = stage[j];                    // s_b follows s_a (one array, two planes)
    }
    }
    __syncthreads();
    const int lx = threadIdx.x % TW, ry = threadIdx.x / TW;
    const int ox = ox0 + lx, oyb = oy0 + ry * RPT;
    if (ox >= p.OW || oyb >= p.OH) return;
    const n3d_epilogue& E = p.epi;
    const float nstr = (p.has_epi && E.noise) ? E.noise_strength[0] : 0.f;
    // the layer epilogue for linear / leaky ReLU (0 <= alpha <= 1): lrelu(v) = max(v, alpha v); no clamp = clamp at infinity
    const float alpha_eff = (p.has_epi && E.act == N3D_ACT_LRELU) ? E.alpha : 1.f;
    const float gain_eff = p.has_epi ? E.gain : 1.f, clamp_eff = (p.has_epi && E.clamp >= 0.f) ? E.clamp : INFINITY;
    float nz[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) nz[j] = (p.has_epi && E.noise && oyb + j < p.OH) ? E.noise[(int64_t)(oyb + j) * p.OW + ox] * nstr : 0.f;
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    const int64_t plane = (int64_t)p.OH * p.OW;                           // 16-byte units per (hl, c8) plane
    bf16x4_t* yh = reinterpret_cast<bf16x4_t*>(p.y + (((int64_t)n * 2 + 0) * (p.C / CH) + c8) * plane);
    bf16x4_t* yl = reinterpret_cast<bf16x4_t*>(p.y + (((int64_t)n * 2 + 1) * (p.C / CH) + c8) * plane);
    // The two channel halves of a unit one after the other, as a REAL loop: unrolled into one basic block the scheduler hoists
    // both halves' LDS reads to the top (~250 VGPRs, 2 waves per SIMD, which made the first version latency-bound at 2 TB/s).
    // The first half's packed results (8 registers) wait for the second: every unit is stored whole, 16 bytes per lane and plane.
    bf16x4_t hi0[RPT], lo0[RPT];
#pragma unroll 1
    for (int hf = 0; hf < 2; ++hf) {
        const f32x4* sp = s_ab + hf * FH * FW;
        float acc[RPT][4];
#pragma unroll
        for (int j = 0; j < RPT; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][c] = 0.f;
#pragma unroll
        for (int rr = 0; rr < RPT + 3; ++rr) {
            f32x4 in[4];
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) in[kx] = sp[(ry * RPT + rr) * FW + lx + kx];
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int ky = rr - j;
                if (ky < 0 || ky > 3) continue;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) acc[j][c] += in[kx][c] * f[ky][kx];
            }
        }
        bf16x4_t hi[RPT], lo[RPT];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int cc = c8 * CH + 4 * hf + c;
            float sc = 1.f, bias = 0.f;
            if (p.has_epi) {
                sc = E.const_scale;
                if (E.row_scale) sc *= E.row_scale[(int64_t)n * (E.row_scale_stride ? E.row_scale_stride : p.C) + cc];
                if (E.bias) bias = E.bias[cc];
            }
            const float os = p.out_scale ? p.out_scale[(int64_t)n * p.out_scale_stride + cc] : 1.f;
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                float v = acc[j][c] * p.gain;
                v = v * sc + nz[j] + bias;
                v = FAST ? fmaxf(v, v * alpha_eff) : n3d_act(v, p.has_epi ? E.act : N3D_ACT_LINEAR, E.alpha);
                v *= gain_eff;
                v = n3d_round16(fminf(fmaxf(v, -clamp_eff), clamp_eff), p.has_epi && E.round_f16);      // fp16 block: the activation is stored as float16 ...
                v *= os;                                                                        // ... and modulated by the next layer afterwards
                const __bf16 h = (__bf16)v;
                hi[j][c] = h;
                lo[j][c] = (__bf16)(v - (float)h);
            }
        }
        if (hf == 0) {                                                    // keep the first half; the unit is stored whole (16 bytes) with the second
#pragma unroll
            for (int j = 0; j < RPT; ++j) { hi0[j] = hi[j]; lo0[j] = lo[j]; }
        } else {
            bf16x8_t* yh8 = reinterpret_cast<bf16x8_t*>(yh), *yl8 = reinterpret_cast<bf16x8_t*>(yl);
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int oy = oyb + j;
                if (oy >= p.OH) break;
                const int64_t u = (int64_t)oy * p.OW + ox;                // 16-byte units
                yh8[u] = __builtin_shufflevector(hi0[j], hi[j], 0, 1, 2, 3, 4, 5, 6, 7);
                yl8[u] = __builtin_shufflevector(lo0[j], lo[j], 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    }
}

// Separable form of fir4_c8_split8_kernel for filters f = outer(a, a) (the model's [1,3,3,1] filter), c8 input, padding 1: a vertical
// pass per input column straight from global memory (7 rows -> 4 output rows per work item), the column sums shared through LDS, then
// the horizontal pass + the layer epilogue + the hi / lo split: 8 instead of 16 multiply-adds per output and channel and no input
// staging (the h8 twin of this kernel, sr_f16.hip: fir4_h8_sep_kernel, went from 2.4 to 4.9 TB/s that way).  The float32 sum is
// evaluated in another order than the 16-tap form — differences of a float32 ulp.  Tile: 61 x 16 outputs, 256 work items.
struct FirSplitSepParams {
    const float* x; const float* f1d; bf16x8_t* y;
    int N, C, H, W, OH, OW, flip, tiles_x, pad;
    float gain;
    int64_t xbs, xrs;
    const float* out_scale; int64_t out_scale_stride;
    int has_epi;
    n3d_epilogue epi;
};
// NCHW_IN: the same from 8 float32 channel planes (padding p.pad = 1 or 2): the filter in front of a stride-2 convolution — the
// 16-tap form staged 10,184 single floats per tile through LDS and ran at 2.2 TB/s.
template <bool NCHW_IN, int RPT = 4>
__global__ __launch_bounds__(256) void fir4_split8_sep_kernel(FirSplitSepParams p) {
    constexpr int SW = 61, COLS = 64, TH = 4 * RPT;
    __shared__ f32x4 s_v[2 * TH * COLS];                                  // [channel half][row][column]
    const int lx = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
    const int c8 = blockIdx.y, n = blockIdx.z;
    const int ox0 = tx * SW, oy0 = ty * TH;
    float a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = p.flip ? p.f1d[k] : p.f1d[3 - k];
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.xbs + (int64_t)c8 * p.H * p.xrs * 8), 0,
                                                                          (int)(p.H * p.xrs * 32), 0x00020000);
    {
        const int ix = ox0 - p.pad + lx;
        f32x4 lo4[RPT + 3], hi4[RPT + 3];
#pragma unroll
        for (int r = 0; r < RPT + 3; ++r) {
            const int iy = oy0 - p.pad + RPT * g + r;
            const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            if (NCHW_IN) {                                                // 8 planes of H x xrs floats; a wave reads 256 contiguous bytes per (channel, row)
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    const int o0 = ok ? ((ch * p.H + iy) * (int)p.xrs + ix) * 4 : (int)0x80000000;
                    const int o1 = ok ? (((ch + 4) * p.H + iy) * (int)p.xrs + ix) * 4 : (int)0x80000000;
                    lo4[r][ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, o0, 0, 0));
                    hi4[r][ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, o1, 0, 0));
                }
            } else {
                const int off = ok ? (iy * (int)p.xrs + ix) * 32 : (int)0x80000000;
                lo4[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
                hi4[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 16, 0));
            }
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            f32x4 v0 = lo4[j] * a[0], v1 = hi4[j] * a[0];
#pragma unroll
            for (int ky = 1; ky < 4; ++ky) { v0 += lo4[j + ky] * a[ky]; v1 += hi4[j + ky] * a[ky]; }
            s_v[(RPT * g + j) * COLS + lx] = v0;
            s_v[(TH + RPT * g + j) * COLS + lx] = v1;
        }
    }
    __syncthreads();
    const int ox = ox0 + lx;
    if (lx >= SW || ox >= p.OW) return;
    const n3d_epilogue& E = p.epi;
    const float nstr = (p.has_epi && E.noise) ? E.noise_strength[0] : 0.f;
    const float alpha_eff = (p.has_epi && E.act == N3D_ACT_LRELU) ? E.alpha : 1.f;
    const float gain_eff = p.has_epi ? E.gain : 1.f, clamp_eff = (p.has_epi && E.clamp >= 0.f) ? E.clamp : INFINITY;
    const int r16 = p.has_epi && E.round_f16;
    float sc[8], bs[8], os[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int cc = c8 * 8 + c;
        sc[c] = 1.f; bs[c] = 0.f;
        if (p.has_epi) {
            sc[c] = E.const_scale;
            if (E.row_scale) sc[c] *= E.row_scale[(int64_t)n * (E.row_scale_stride ? E.row_scale_stride : p.C) + cc];
            if (E.bias) bs[c] = E.bias[cc];
        }
        os[c] = p.out_scale ? p.out_scale[(int64_t)n * p.out_scale_stride + cc] : 1.f;
    }
    const int64_t plane = (int64_t)p.OH * p.OW;
    bf16x8_t* yh8 = p.y + (((int64_t)n * 2 + 0) * (p.C / 8) + c8) * plane;
    bf16x8_t* yl8 = p.y + (((int64_t)n * 2 + 1) * (p.C / 8) + c8) * plane;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int oy = oy0 + RPT * g + j;
        if (oy >= p.OH) break;
        const f32x4* r0 = s_v + (RPT * g + j) * COLS + lx, *r1 = r0 + TH * COLS;
        f32x4 s0 = r0[0] * a[0], s1 = r1[0] * a[0];
#pragma unroll
        for (int kx = 1; kx < 4; ++kx) { s0 += r0[kx] * a[kx]; s1 += r1[kx] * a[kx]; }
        const float nz = (p.has_epi && E.noise) ? E.noise[(int64_t)oy * p.OW + ox] * nstr : 0.f;
        bf16x8_t hi, lo;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = (c < 4 ? s0[c] : s1[c - 4]) * p.gain;
            v = v * sc[c] + nz + bs[c];
            v = fmaxf(v, v * alpha_eff) * gain_eff;
            v = n3d_round16(__builtin_amdgcn_fmed3f(v, -clamp_eff, clamp_eff), r16);
            v *= os[c];
            const __bf16 h = (__bf16)v;
            hi[c] = h;
            lo[c] = (__bf16)(v - (float)h);
        }
        const int64_t u = (int64_t)oy * p.OW + ox;
        yh8[u] = hi;
        yl8[u] = lo;
    }
}

static int fir4_split8_impl(const float* x, const float* f, void* y, int N, int C, int H, int W, int64_t x_row_stride, int64_t x_batch_stride,
                           int pad, bool nchw_in, int flip, float gain, const n3d_epilogue* epi, const float* out_scale, int64_t out_scale_stride,
                           hipStream_t stream) {
    N3D_CHECK(N >= 0 && C > 0 && C % 8 == 0 && H > 1 && W > 1, "fir4_split8: bad shape (C %% 8 == 0)");
    N3D_CHECK(pad == 1 || pad == 2, "fir4_split8: padding 1 or 2");
    const int64_t xrs = x_row_stride ? x_row_stride : W;
    N3D_CHECK(xrs >= W && (nchw_in || ((x_batch_stride & 3) == 0 && ((uintptr_t)x & 15) == 0)), "fir4_split8: misaligned c8 input");
    const int OH = H + 2 * pad - 3, OW = W + 2 * pad - 3;                  // 4 taps
    N3D_CHECK(!epi || (!epi->residual && !epi->residual_up_filter), "fir4_split8: no residual input");
    N3D_CHECK(!epi || !epi->noise || epi->noise_strength, "fir4_split8: noise without noise_strength");
    N3D_CHECK(!epi || (epi->act >= N3D_ACT_LINEAR && epi->act <= N3D_ACT_SWISH), "fir4_split8: unknown activation");
    if (N == 0) return 0;
    N3D_CHECK(x && f && y && ((uintptr_t)y & 15) == 0, "fir4_split8: null or misaligned tensor");
    N3D_CHECK(C / 8 <= 65535 && N <= 65535, "fir4_split8: N and C/8 must be <= 65535");
    N3D_CHECK((int64_t)H * xrs * 32 < (1ll << 31), "fir4_split8: 8 channel planes exceed 2 GiB (32-bit buffer offsets)");
    FirSplitParams p;
    p.x = x; p.f = f; p.y = (bf16x8_t*)y; p.N = N; p.C = C; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.flip = flip; p.gain = gain;
    p.xbs = x_batch_stride ? x_batch_stride : (int64_t)C * H * xrs; p.xrs = xrs; p.pad = pad;
    p.out_scale = out_scale; p.out_scale_stride = out_scale_stride ? out_scale_stride : C;
    p.has_epi = epi != nullptr;
    if (epi) p.epi = *epi;
    p.tiles_x = cdiv(OW, FIR_TW);
    N3dProfScope prof(N3D_K_UPFIRDN2D, stream, 2.0 * N * C * (double)OH * OW * 16, 4.0 * N * C * ((double)H * W + (double)OH * OW));
    const bool fast = !epi || epi->act == N3D_ACT_LINEAR || (epi->act == N3D_ACT_LRELU && epi->alpha >= 0.f && epi->alpha <= 1.f);
    const dim3 grid(p.tiles_x * cdiv(OH, 16), C / 8, N);
    if (nchw_in) {
        if (fast) hipLaunchKernelGGL((fir4_c8_split8_kernel<true, true>), grid, dim3(FIR_TW * 8), 0, stream, p);
        else hipLaunchKernelGGL((fir4_c8_split8_kernel<false, true>), grid, dim3(FIR_TW * 8), 0, stream, p);
    } else {
        if (fast) hipLaunchKernelGGL((fir4_c8_split8_kernel<true, false>), grid, dim3(FIR_TW * 8), 0, stream, p);
        else hipLaunchKernelGGL((fir4_c8_split8_kernel<false, false>), grid, dim3(FIR_TW * 8), 0, stream, p);
    }
    N3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int n3d_fir4_split8(const float* x, const float* f, void* y, int N, int C, int H, int W, int64_t x_row_stride, int64_t x_batch_stride,
                               int flip, float gain, const n3d_epilogue* epi, const float* out_scale, int64_t out_scale_stride,
                               n3d_stream_t stream_) {
    return fir4_split8_impl(x, f, y, N, C, H, W, x_row_stride, x_batch_stride, 1, false, flip, gain, epi, out_scale, out_scale_stride, (hipStream_t)stream_);
}

extern "C" int n3d_fir4_split8_sep(const float* x, const float* f1d, void* y, int N, int C, int H, int W, int64_t x_row_stride, int64_t x_batch_stride,
                                   int flip, float gain, const n3d_epilogue* epi, const float* out_scale, int64_t out_scale_stride, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && C > 0 && C % 8 == 0 && H > 3 && W > 3, "fir4_split8_sep: bad shape (C %% 8 == 0)");
    const int64_t xrs = x_row_stride ? x_row_stride : W;
    N3D_CHECK(xrs >= W && (x_batch_stride & 3) == 0 && ((uintptr_t)x & 15) == 0, "fir4_split8_sep: misaligned c8 input");
    N3D_CHECK(!epi || (!epi->residual && !epi->residual_up_filter), "fir4_split8_sep: no residual input");
    N3D_CHECK(!epi || !epi->noise || epi->noise_strength, "fir4_split8_sep: noise without noise_strength");
    N3D_CHECK(!epi || epi->act == N3D_ACT_LINEAR || (epi->act == N3D_ACT_LRELU && epi->alpha >= 0.f && epi->alpha <= 1.f), "fir4_split8_sep: linear or leaky-ReLU epilogue only");
    if (N == 0) return 0;
    N3D_CHECK(x && f1d && y && ((uintptr_t)y & 15) == 0, "fir4_split8_sep: null or misaligned tensor");
    N3D_CHECK(C / 8 <= 65535 && N <= 65535 && (int64_t)H * xrs * 32 < (1ll << 31), "fir4_split8_sep: tensor too large");
    FirSplitSepParams p;
    p.x = x; p.f1d = f1d; p.y = (bf16x8_t*)y; p.N = N; p.C = C; p.H = H; p.W = W; p.OH = H - 1; p.OW = W - 1; p.flip = flip; p.gain = gain;
    p.xbs = x_batch_stride ? x_batch_stride : (int64_t)C * H * xrs; p.xrs = xrs;
    p.out_scale = out_scale; p.out_scale_stride = out_scale_stride ? out_scale_stride : C;
    p.has_epi = epi != nullptr;
    if (epi) p.epi = *epi;
    p.tiles_x = cdiv(p.OW, 61); p.pad = 1;
    N3dProfScope prof(N3D_K_UPFIRDN2D, stream, 2.0 * N * C * (double)p.OH * p.OW * 8, 4.0 * N * C * ((double)H * W + (double)p.OH * p.OW));
    // (8 rows per work item — 11 input rows for 8 outputs instead of 7 for 4, 64 KB of LDS — measured 8-12 % slower: 2 workgroups per CU)
    hipLaunchKernelGGL((fir4_split8_sep_kernel<false, 4>), dim3(p.tiles_x * cdiv(p.OH, 16), C / 8, N), dim3(256), 0, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int n3d_fir4_split8_nchw_sep(const float* x, const float* f1d, void* y, int N, int C, int H, int W, int64_t x_row_stride, int64_t x_batch_stride,
                                        int pad, int flip, float gain, const n3d_epilogue* epi, const float* out_scale, int64_t out_scale_stride,
                                        n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && C > 0 && C % 8 == 0 && H > 1 && W > 1, "fir4_split8_nchw_sep: bad shape (C %% 8 == 0)");
    N3D_CHECK(pad == 1 || pad == 2, "fir4_split8_nchw_sep: padding 1 or 2");
    const int64_t xrs = x_row_stride ? x_row_stride : W;
    N3D_CHECK(xrs >= W, "fir4_split8_nchw_sep: row stride smaller than the width");
    N3D_CHECK(!epi || (!epi->residual && !epi->residual_up_filter), "fir4_split8_nchw_sep: no residual input");
    N3D_CHECK(!epi || !epi->noise || epi->noise_strength, "fir4_split8_nchw_sep: noise without noise_strength");
    N3D_CHECK(!epi || epi->act == N3D_ACT_LINEAR || (epi->act == N3D_ACT_LRELU && epi->alpha >= 0.f && epi->alpha <= 1.f), "fir4_split8_nchw_sep: linear or leaky-ReLU epilogue only");
    if (N == 0) return 0;
    N3D_CHECK(x && f1d && y && ((uintptr_t)y & 15) == 0, "fir4_split8_nchw_sep: null or misaligned tensor");
    N3D_CHECK(C / 8 <= 65535 && N <= 65535 && (int64_t)H * xrs * 32 < (1ll << 31), "fir4_split8_nchw_sep: tensor too large");
    FirSplitSepParams p;
    p.x = x; p.f1d = f1d; p.y = (bf16x8_t*)y; p.N = N; p.C = C; p.H = H; p.W = W; p.OH = H + 2 * pad - 3; p.OW = W + 2 * pad - 3; p.flip = flip; p.gain = gain;
    p.xbs = x_batch_stride ? x_batch_stride : (int64_t)C * H * xrs; p.xrs = xrs; p.pad = pad;
    p.out_scale = out_scale; p.out_scale_stride = out_scale_stride ? out_scale_stride : C;
    p.has_epi = epi != nullptr;
    if (epi) p.epi = *epi;
    p.tiles_x = cdiv(p.OW, 61);
    N3dProfScope prof(N3D_K_UPFIRDN2D, stream, 2.0 * N * C * (double)p.OH * p.OW * 8, 4.0 * N * C * ((double)H * W + (double)p.OH * p.OW));
    hipLaunchKernelGGL((fir4_split8_sep_kernel<true, 4>), dim3(p.tiles_x * cdiv(p.OH, 16), C / 8, N), dim3(256), 0, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int n3d_fir4_split8_nchw(const float* x, const float* f, void* y, int N, int C, int H, int W, int64_t x_row_stride, int64_t x_batch_stride,
                                    int pad, int flip, float gain, const n3d_epilogue* epi, const float* out_scale, int64_t out_scale_stride,
                                    n3d_stream_t stream_) {
    return fir4_split8_impl(x, f, y, N, C, H, W, x_row_stride, x_batch_stride, pad, true, flip, gain, epi, out_scale, out_scale_stride, (hipStream_t)stream_);
}

static int upfirdn2d_impl(const float* x, const float* f, float* y, int N, int C, int H, int W, int64_t xrs, int64_t yrs, int fh, int fw, int upx,
                          int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                          int64_t xbs, int64_t ybs, const n3d_epilogue* epi, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && C > 0 && H > 0 && W > 0, "upfirdn2d: bad input shape");
    N3D_CHECK(xrs >= W, "upfirdn2d: input row pitch smaller than the width");
    N3D_CHECK(fh >= 1 && fw >= 1 && fh * fw <= UF_MAX_TAPS, "upfirdn2d: filter %dx%d unsupported", fh, fw);
    N3D_CHECK(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1, "upfirdn2d: up/down must be >= 1");
    const int OW = (W * upx + padx0 + padx1 - fw + downx) / downx;
    const int OH = (H * upy + pady0 + pady1 - fh + downy) / downy;
    N3D_CHECK(OW >= 1 && OH >= 1, "upfirdn2d: output would be empty");   // upfirdn2d.cpp:39
    N3D_CHECK(!epi || !epi->residual_up_filter || (epi->residual && OH % 2 == 0 && OW % 2 == 0), "upfirdn2d: residual_up_filter needs a residual and an even output size");
    if (N == 0) return 0;
    N3D_CHECK(x && f && y, "upfirdn2d: null tensor");
    N3D_CHECK(C <= 65535 && N <= 65535, "upfirdn2d: N and C must be <= 65535");
    UfParams p;
    p.x = x; p.f = f; p.y = y; p.N = N; p.C = C; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.fh = fh; p.fw = fw;
    p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy; p.padx0 = padx0; p.pady0 = pady0; p.flip = flip;
    p.gain = gain; p.xbs = xbs; p.ybs = ybs; p.xrs = xrs; p.yrs = yrs ? yrs : OW;
    N3D_CHECK(p.yrs >= OW, "upfirdn2d: output row pitch smaller than the output width");
    N3D_CHECK(p.yrs == OW || !epi || (!epi->noise && !epi->residual), "upfirdn2d: a pitched output cannot take per-pixel epilogue inputs");
    N3D_CHECK(!epi || !epi->round_f16, "upfirdn2d: round_f16 is supported by n3d_fir4_split8 only");
    p.has_epi = epi != nullptr;
    if (epi) p.epi = *epi;
    p.tiles_x = cdiv(OW, UF_TILE_W); p.tiles_y = cdiv(OH, UF_TILE_H);
    // footprint extent: i ranges over ceil((o0*down - pad)/up) .. floor(((o0+T-1)*down + f-1 - pad)/up)
    p.foot_w = ((UF_TILE_W - 1) * downx + fw - 1) / upx + 2;
    p.foot_h = ((UF_TILE_H - 1) * downy + fh - 1) / upy + 2;
    N3D_CHECK(p.foot_w * p.foot_h <= UF_MAX_FOOT, "upfirdn2d: tile footprint %dx%d exceeds LDS budget", p.foot_h, p.foot_w);
    N3dProfScope prof(N3D_K_UPFIRDN2D, stream, 2.0 * N * C * (double)OH * OW * fh * fw / (upx * upy),
                      4.0 * N * C * ((double)H * W + (double)OH * OW));
    const bool fast = fh == 4 && fw == 4 && upx == upy && downx == downy && ((upx == 1 && downx <= 2) || (upx == 2 && downx == 1));
    const bool aligned = ((xrs | xbs | ybs | p.yrs) & 3) == 0 && ((OW & 3) == 0 || !epi) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 &&
                         (!epi || (!epi->residual_up_filter && (!epi->noise || ((uintptr_t)epi->noise & 15) == 0) &&
                                   (!epi->residual || (((uintptr_t)epi->residual & 15) == 0 && (epi->residual_batch_stride & 3) == 0))));
    if (fast && upx == 1 && downx == 1 && (padx0 == 1 || padx0 == 2) && aligned) {
        const int tw = OW >= 128 ? 128 : 64;
        p.tiles_x = cdiv(OW, tw); p.tiles_y = cdiv(OH, 32);
        const dim3 grid(p.tiles_x * p.tiles_y, C, N);
        if (tw == 128 && padx0 == 1) hipLaunchKernelGGL((fir4_vec_kernel<128, 1>), grid, dim3(256), 0, stream, p);
        else if (tw == 128) hipLaunchKernelGGL((fir4_vec_kernel<128, 2>), grid, dim3(256), 0, stream, p);
        else if (padx0 == 1) hipLaunchKernelGGL((fir4_vec_kernel<64, 1>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((fir4_vec_kernel<64, 2>), grid, dim3(256), 0, stream, p);
    } else if (fast) {
        p.tiles_x = cdiv(OW, FT_W); p.tiles_y = cdiv(OH, FT_H);
        dim3 grid(p.tiles_x * p.tiles_y, C, N);
        if (upx == 1 && downx == 1) hipLaunchKernelGGL((upfirdn2d_fast_kernel<1, 1, 4>), grid, dim3(256), 0, stream, p);
        else if (upx == 2) hipLaunchKernelGGL((upfirdn2d_fast_kernel<2, 1, 4>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((upfirdn2d_fast_kernel<1, 2, 4>), grid, dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL(upfirdn2d_kernel, dim3(p.tiles_x * p.tiles_y, C, N), dim3(256), 0, stream, p);
    }
    N3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int n3d_upfirdn2d(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw, int upx,
                             int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                             int64_t xbs, int64_t ybs, const n3d_epilogue* epi, n3d_stream_t stream) {
    return upfirdn2d_impl(x, f, y, N, C, H, W, W, 0, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain, xbs, ybs,
                          epi, stream);
}

extern "C" int n3d_upfirdn2d_pitched(const float* x, const float* f, float* y, int N, int C, int H, int W, int64_t x_row_stride,
                                     int64_t y_row_stride, int fh, int fw, int upx, int upy, int downx, int downy, int padx0,
                                     int padx1, int pady0, int pady1, int flip, float gain, int64_t xbs, int64_t ybs,
                                     const n3d_epilogue* epi, n3d_stream_t stream) {
    return upfirdn2d_impl(x, f, y, N, C, H, W, x_row_stride ? x_row_stride : W, y_row_stride, fh, fw, upx, upy, downx, downy, padx0,
                          padx1, pady0, pady1, flip, gain, xbs, ybs, epi, stream);
}
