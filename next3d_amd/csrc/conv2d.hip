// conv2d for gfx950: fp32 implicit-GEMM convolution on the matrix cores (v_mfma_f32_32x32x2_f32) — the one
// dense contraction of the Next3D generator (~765 GFLOP/frame).  Bound: fp32 MFMA (157 TFLOP/s).
//
//   C[o, p] = sum_{tap, i} Wt[tap][i][o] * (style[n,i] * X[n, i, y(p)*S + dy(tap) - P, x(p)*S + dx(tap) - P])
//
// One 256-thread workgroup (4 waves) owns BM output channels x (TH x TW = 128) output pixels of one
// sample.  Per stage of ICB input channels it stages
//   As[tap][ic][BM]  : weight slab, K-major (the layout n3d_conv2d_prep_weight produces) -> 16-byte global loads,
//   Bs[ic][PH][PWP]  : the input patch WITH halo, loaded once and re-used by all k*k taps (9x fewer activation
//                      reads than an im2col GEMM), modulated by style[n,i] on the way in (modulated_conv2d's
//                      per-sample weights w*s become per-sample activations x*s: the weight slab is shared by
//                      the whole batch),
// into LDS, while the next stage's global loads are already in flight in registers.  Every lane then feeds the
// MFMA with one A and one B float per k-step of 2 channels: lanes 0-31 / 32-63 read two adjacent channel rows,
// 32 consecutive floats each -> conflict-free ds_read_b32.  fp32 MFMA issues one 32x32x2 per 64 cycles per SIMD,
// so LDS bandwidth is far from binding; what matters is keeping 4 independent accumulators per wave busy.
// The epilogue (demodulation, noise, bias, activation, clamp, residual) runs on the accumulators in registers.
//
// Replaces F.conv2d / F.conv_transpose2d as called by conv2d_resample (reference
// torch_utils/ops/conv2d_resample.py:96-136) from modulated_conv2d (training_avatar_texture/networks_stylegan2.py:34-91).
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __noinline__ float conv_act_generic(float v, int act, float alpha) { return n3d_act(v, act, alpha); }

struct ConvParams {
    const float* x; const float* wt; const float* style; float* y; float* partial;
    int N, I, O, OP, H, W, OH, OW;   // OH/OW: full output dims; OP = O rounded up to 4 (row pitch of wt)
    int GH, GW;                      // per-phase output grid (mode 2: H+1, W+1; else OH, OW)
    int tiles_x, tiles_y, nphase, ksplit, ic_per_split;
    int dbg;                         // N3D_CONV_DBG ablation bits (tuning only): 1 skip stores, 2 skip MFMA, 4 skip stage loads
    int64_t xbs, ybs, style_stride, yrs;      // yrs: output row pitch in floats
    int64_t wbs;                              // floats between consecutive samples' prepared weights (0 = shared by the batch)
    n3d_epilogue epi;
};

// MODE 0: stride 1 pad KS/2 | MODE 1: stride 2 pad 0 | MODE 2: transposed stride 2 (4 polyphase sub-convolutions)
template <int MT, int NT, int WM, int WN, int TH, int TW, int ICB, int KS, int MODE>
__global__ __launch_bounds__(256) void conv2d_mfma_kernel(ConvParams p) {
    constexpr int BM = WM * MT * 32;
    constexpr int BN = WN * NT * 32;
    static_assert(BN == TH * TW, "pixel tile must match the GEMM N tile");
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int S = (MODE == 1) ? 2 : 1;
    constexpr int P = (MODE == 1) ? 0 : KS / 2;
    constexpr int PH = (TH - 1) * S + KS;
    constexpr int PW = (TW - 1) * S + KS;
    constexpr int PWP = PW + ((PW % 2 == 0) ? 1 : 0) + ((TW < 32) ? 2 : 0);   // odd-ish row pitch spreads rows over banks
    constexpr int TAPS = KS * KS;
    constexpr int A_ELEMS = TAPS * ICB * BM;
    constexpr int B_ELEMS = ICB * PH * PW;
    constexpr int A_VEC = A_ELEMS / 4;
    constexpr int A_PER_T = (A_VEC + 255) / 256;
    constexpr int B_PER_T = (B_ELEMS + 255) / 256;

    __shared__ float As[A_ELEMS];
    __shared__ float Bs[ICB * PH * PWP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = blockIdx.x;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int m0 = blockIdx.y * BM;
    int z = blockIdx.z;
    const int ks = z % p.ksplit; z /= p.ksplit;
    const int phase = z % p.nphase;
    const int n = z / p.nphase;
    const int y0 = ty * TH, x0 = tx * TW;

    // tap table for this block (uniform): weight plane + patch offset of tap t
    int ntaps = TAPS, pa = 0, pb = 0, nkx = KS;
    if (MODE == 2) {
        pa = phase >> 1; pb = phase & 1;
        const int nky = pa ? 1 : 2;
        nkx = pb ? 1 : 2;
        ntaps = nky * nkx;
    }
    auto tap_wplane = [&](int t) -> int {
        if (MODE != 2) return t;
        const int ti = t / nkx, tj = t % nkx;
        const int ky = pa ? 1 : 2 * ti, kx = pb ? 1 : 2 * tj;
        return ky * 3 + kx;
    };
    auto tap_boff = [&](int t) -> int {
        if (MODE != 2) return (t / KS) * PWP + (t % KS);
        const int ti = t / nkx, tj = t % nkx;
        const int dy = pa ? 1 : 1 - ti, dx = pb ? 1 : 1 - tj;   // ky=0 -> input row y (dy=1), ky=2 -> row y-1 (dy=0)
        return dy * PWP + dx;
    };

    const int ic_begin = ks * p.ic_per_split;
    const int ic_end = min(p.I, ic_begin + p.ic_per_split);
    const int nstage = (ic_end - ic_begin + ICB - 1) / ICB;

    // ---- staging state, computed ONCE per thread: the stage loop only bumps two base pointers.
    // A element v = tid + 256 j  -> As[v*4 .. v*4+3]  <- wt[(plane(t)*I + ic0 + ic)*OP + m0 + mv ..]
    //   rows with t >= ntaps or m >= OP are never consumed (their accumulator rows are never stored), so they only
    //   need a SAFE address, not a zero.  B element e = tid + 256 j -> Bs[(ic*PH + r)*PWP + q] <- x[n, ic0+ic, iy0+r, ix0+q]
    //   (zero outside the image = the convolution's zero padding), times style[n, ic0+ic].
    __shared__ float s_style[1024];
    const float* xn = p.x + (int64_t)n * p.xbs;
    const int iy0 = y0 * S - P, ix0 = x0 * S - P;
    const int HW = p.H * p.W;
    for (int i = tid; i < ic_end - ic_begin; i += 256)       // no style = multiply by 1 (keeps the stage loop branch-free)
        s_style[i] = p.style ? p.style[(int64_t)n * p.style_stride + ic_begin + i] : 1.f;

    int a_goff[A_PER_T], a_ic[A_PER_T];
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
        const int v = tid + j * 256;
        const int row = v / (BM / 4), mv = (v % (BM / 4)) * 4;
        const int t = row / ICB, ic = row % ICB;
        const bool ok = v < A_VEC && t < ntaps && (m0 + mv) < p.OP;
        a_ic[j] = ok ? ic : ICB;                                   // ICB = "never valid": forces the safe address
        a_goff[j] = ok ? (tap_wplane(t) * p.I + ic) * p.OP + m0 + mv : 0;
    }
    int b_goff[B_PER_T], b_loff[B_PER_T], b_ic[B_PER_T];
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j) {
        const int e = tid + j * 256;
        const int ic = e / (PH * PW), rem = e % (PH * PW);
        const int r = rem / PW, q = rem % PW;
        const int iy = iy0 + r, ix = ix0 + q;
        const bool ok = e < B_ELEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        b_ic[j] = ok ? ic : ICB;
        b_goff[j] = ok ? ic * HW + iy * p.W + ix : 0;
        b_loff[j] = e < B_ELEMS ? (ic * PH + r) * PWP + q : -1;
    }
    const float* a_base = p.wt + (int64_t)n * p.wbs + (int64_t)ic_begin * p.OP;
    const float* b_base = xn + (int64_t)ic_begin * HW;

    f32x4 ra[A_PER_T];
    float rb[B_PER_T];

    // load_stage only ISSUES the global loads (raw values stay in registers while the MFMA block runs); the style
    // modulation / zero padding is applied in store_stage, after the MFMAs — consuming a loaded value any earlier puts an
    // s_waitcnt vmcnt(0) in front of the MFMA block and serialises the two phases (measured: load + MFMA time added up).
    auto load_stage = [&](int st) {
        const int nch = min(ICB, ic_end - ic_begin - st * ICB);     // valid channels of this stage (tail stage: < ICB)
        const float* ab = a_base + (int64_t)st * ICB * p.OP;
        const float* bb = b_base + (int64_t)st * ICB * HW;
#pragma unroll
        for (int j = 0; j < A_PER_T; ++j) ra[j] = *reinterpret_cast<const f32x4*>(ab + (a_ic[j] < nch ? a_goff[j] : 0));
#pragma unroll
        for (int j = 0; j < B_PER_T; ++j) rb[j] = bb[b_ic[j] < nch ? b_goff[j] : 0];
    };
    auto store_stage = [&](int st) {
        const int nch = min(ICB, ic_end - ic_begin - st * ICB);
#pragma unroll
        for (int j = 0; j < A_PER_T; ++j) {
            const int v = tid + j * 256;
            if (v < A_VEC) *reinterpret_cast<f32x4*>(&As[v * 4]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < B_PER_T; ++j) {
            const bool ok = b_ic[j] < nch;
            const float val = rb[j] * s_style[st * ICB + (ok ? b_ic[j] : 0)];
            if (b_loff[j] >= 0) Bs[b_loff[j]] = ok ? val : 0.f;
        }
    };

    // per-lane fragment addressing
    const int half = lane >> 5, l31 = lane & 31;
    int a_off[MT], b_off[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a_off[mt] = (wm * MT + mt) * 32 + l31;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int pix = (wn * NT + nt) * 32 + l31;
        b_off[nt] = ((pix / TW) * S) * PWP + (pix % TW) * S;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    __syncthreads();                     // s_style visible
    if (nstage > 0) {
        load_stage(0);
        store_stage(0);
    }
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        if (st + 1 < nstage) load_stage(st + 1);
        for (int t = 0; t < ntaps; ++t) {
            const float* At = As + (t * ICB + half) * BM;
            const float* Bt = Bs + half * (PH * PWP) + tap_boff(t);
#pragma unroll
            for (int kk = 0; kk < ICB / 2; ++kk) {
                float a[MT], b[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = At[kk * 2 * BM + a_off[mt]];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[nt] = Bt[kk * 2 * (PH * PWP) + b_off[nt]];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        __syncthreads();
        if (st + 1 < nstage) {
            store_stage(st + 1);
            __syncthreads();
        }
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (channel)
    const n3d_epilogue& E = p.epi;
    const int64_t plane = (int64_t)p.OH * p.OW;
    if (p.partial) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int pix = (wn * NT + nt) * 32 + l31;
            const int gy = y0 + pix / TW, gx = x0 + pix % TW;
            int oy = gy, ox = gx;
            bool ok = gy < p.GH && gx < p.GW;
            if (MODE == 2) { oy = 2 * gy + pa; ox = 2 * gx + pb; ok = ok && oy < p.OH && ox < p.OW; }
            if (!ok) continue;
            float* dst = p.partial + ((int64_t)ks * p.N + n) * p.O * plane + (int64_t)oy * p.OW + ox;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (o < p.O) dst[(int64_t)o * plane] = acc[mt][nt][r];
                }
        }
        return;
    }
    // per-channel factors first (row scale, bias), then one pass over the pixels
    float rs[MT][16], bs[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int oc = o < p.O ? o : p.O - 1;
            rs[mt][r] = E.const_scale * (E.row_scale ? E.row_scale[(int64_t)n * (E.row_scale_stride ? E.row_scale_stride : p.O) + oc] : 1.f);
            bs[mt][r] = E.bias ? E.bias[oc] : 0.f;
        }
    const float nstr = E.noise ? E.noise_strength[0] : 0.f;
    const bool lrelu = E.act == N3D_ACT_LRELU, linear = E.act == N3D_ACT_LINEAR;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int pix = (wn * NT + nt) * 32 + l31;
        const int gy = y0 + pix / TW, gx = x0 + pix % TW;
        int oy = gy, ox = gx;
        bool ok = gy < p.GH && gx < p.GW;
        if (MODE == 2) { oy = 2 * gy + pa; ox = 2 * gx + pb; ok = ok && oy < p.OH && ox < p.OW; }
        if (!ok) continue;
        const int64_t po = (int64_t)oy * p.OW + ox;
        const float nz = E.noise ? E.noise[po] * nstr : 0.f;
        const int64_t yplane = (int64_t)p.OH * p.yrs;
        float* dst = p.y + (int64_t)n * p.ybs + (int64_t)oy * p.yrs + ox;
        const bool res_up = E.residual && E.residual_up_filter;
        const float* res = E.residual ? E.residual + (int64_t)n * E.residual_batch_stride + (res_up ? 0 : po) : nullptr;
        n3d_up2_taps up2;
        const int64_t lplane = (int64_t)(p.OH >> 1) * (p.OW >> 1);
        if (res_up) up2 = n3d_up2_setup(E.residual_up_filter, oy, ox, p.OH >> 1, p.OW >> 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (o >= p.O) continue;
                float v = acc[mt][nt][r] * rs[mt][r] + nz + bs[mt][r];
                if (lrelu) v = v > 0.f ? v : v * E.alpha;
                else if (!linear) v = conv_act_generic(v, E.act, E.alpha);
                v *= E.gain;
                if (E.clamp >= 0.f) v = fminf(fmaxf(v, -E.clamp), E.clamp);
                if (res_up) v += n3d_up2_apply(up2, res + (int64_t)o * lplane);
                else if (res) v += res[(int64_t)o * plane];
                dst[(int64_t)o * yplane] = v;
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Transposed 3x3 stride-2 convolution (the up-sampling layers, conv2d_resample.py:114-131) with all four output
// phases computed by ONE workgroup from ONE staged input patch.  Output pixel (2y+a, 2x+b) receives
//   a=0: ky in {0,2} from input rows {y, y-1};  a=1: ky=1 from row y   (same for b / kx / columns),
// i.e. the nine taps split 4/2/2/1 over the phases.  A workgroup owns 64 output channels x (TH x TW = 128) INPUT-grid
// positions x 4 phases = 512 output pixels; per k-step of 2 channels a wave issues 9 taps x 2 position tiles = 18 MFMAs
// from 9 A reads + 8 B reads, so every stage carries the same 144 MFMAs per wave as the stride-1 kernel and the
// patch / weight slab is staged once for all phases (the per-phase launch left the 1-tap phase with 16 MFMAs per
// stage: 31-72 TFLOP/s measured vs ~130 for stride 1).
template <int TH, int TW>
__global__ __launch_bounds__(256, 2) void conv2d_up_mfma_kernel(ConvParams p) {
    constexpr int BM = 64, ICB = 16, NT = 2;
    static_assert(TH * TW == 128, "128 input-grid positions per workgroup");
    constexpr int PH = TH + 1, PW = TW + 1;
    constexpr int PWP = PW + ((PW % 2 == 0) ? 1 : 0) + ((TW < 32) ? 2 : 0);
    constexpr int A_ELEMS = 9 * ICB * BM, B_ELEMS = ICB * PH * PW;
    constexpr int A_VEC = A_ELEMS / 4;
    constexpr int A_PER_T = (A_VEC + 255) / 256, B_PER_T = (B_ELEMS + 255) / 256;

    __shared__ float As[A_ELEMS];
    __shared__ float Bs[ICB * PH * PWP];
    __shared__ float s_style[1024];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.x;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int m0 = blockIdx.y * BM;
    const int ks = blockIdx.z % p.ksplit, n = blockIdx.z / p.ksplit;
    const int y0 = ty * TH, x0 = tx * TW;
    const int ic_begin = ks * p.ic_per_split;
    const int ic_end = min(p.I, ic_begin + p.ic_per_split);
    const int nstage = (ic_end - ic_begin + ICB - 1) / ICB;
    const int HW = p.H * p.W;
    const float* xn = p.x + (int64_t)n * p.xbs;

    for (int i = tid; i < ic_end - ic_begin; i += 256) s_style[i] = p.style ? p.style[(int64_t)n * p.style_stride + ic_begin + i] : 1.f;

    // Register budget: 128 accumulators + the prefetched stage (47 VGPRs).  Staging addresses are therefore recomputed
    // from `tid` every stage (constant divisors -> a few VALU ops that run beside the MFMAs) instead of being kept.
    const float* a_base = p.wt + (int64_t)n * p.wbs + (int64_t)ic_begin * p.OP;
    const float* b_base = xn + (int64_t)ic_begin * HW;
    f32x4 ra[A_PER_T];
    float rb[B_PER_T];
    auto load_stage = [&](int st) {
        const int nch = min(ICB, ic_end - ic_begin - st * ICB);
        const float* ab = a_base + (int64_t)st * ICB * p.OP;
        const float* bb = b_base + (int64_t)st * ICB * HW;
#pragma unroll
        for (int j = 0; j < A_PER_T; ++j) {
            const int v = tid + j * 256;
            const int row = v / (BM / 4), mv = (v % (BM / 4)) * 4;
            const int t = row / ICB, ic = row % ICB;
            const bool ok = v < A_VEC && (m0 + mv) < p.OP && ic < nch;
            ra[j] = *reinterpret_cast<const f32x4*>(ab + (ok ? (t * p.I + ic) * p.OP + m0 + mv : 0));
        }
#pragma unroll
        for (int j = 0; j < B_PER_T; ++j) {
            const int e = tid + j * 256;
            const int ic = e / (PH * PW), rem = e % (PH * PW);
            const int iy = y0 - 1 + rem / PW, ix = x0 - 1 + rem % PW;
            const bool ok = e < B_ELEMS && ic < nch && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            rb[j] = bb[ok ? ic * HW + iy * p.W + ix : 0];          // raw: consumed only in store_stage (see the stride-1 kernel)
        }
    };
    auto store_stage = [&](int st) {
        const int nch = min(ICB, ic_end - ic_begin - st * ICB);
#pragma unroll
        for (int j = 0; j < A_PER_T; ++j) {
            const int v = tid + j * 256;
            if (v < A_VEC) *reinterpret_cast<f32x4*>(&As[v * 4]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < B_PER_T; ++j) {
            const int e = tid + j * 256;
            const int ic = e / (PH * PW), rem = e % (PH * PW);
            const int iy = y0 - 1 + rem / PW, ix = x0 - 1 + rem % PW;
            const bool ok = ic < nch && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            if (e < B_ELEMS) Bs[(ic * PH + rem / PW) * PWP + rem % PW] = ok ? rb[j] * s_style[st * ICB + ic] : 0.f;
        }
    };

    // fragment addressing: A rows wm*32 + l31; B position tiles (wn*NT + nt) of 32 positions each
    const int a_off = wm * 32 + l31;
    int b_off[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int pos = (wn * NT + nt) * 32 + l31;
        b_off[nt] = (pos / TW) * PWP + (pos % TW);
    }
    f32x16 acc[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][ph][r] = 0.f;

    __syncthreads();
    if (nstage > 0) { load_stage(0); store_stage(0); }
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        if (st + 1 < nstage && !(p.dbg & 4)) load_stage(st + 1);
        if (!(p.dbg & 2))
#pragma unroll 2
        for (int kk = 0; kk < ICB / 2; ++kk) {
            const float* Ak = As + (2 * kk + half) * BM + a_off;
            const float* Bk = Bs + (2 * kk + half) * (PH * PWP);
            float a[9], b[NT][4];
#pragma unroll
            for (int t = 0; t < 9; ++t) a[t] = Ak[t * ICB * BM];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int d = 0; d < 4; ++d) b[nt][d] = Bk[b_off[nt] + (d >> 1) * PWP + (d & 1)];     // d = dy*2 + dx
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ph = (ky == 1 ? 2 : 0) + (kx == 1 ? 1 : 0);
                    const int d = (ky == 2 ? 0 : 2) + (kx == 2 ? 0 : 1);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt][ph] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ky * 3 + kx], b[nt][d], acc[nt][ph], 0, 0, 0);
                }
        }
        __syncthreads();
        if (st + 1 < nstage) { store_stage(st + 1); __syncthreads(); }
    }

    // epilogue
    if (p.dbg & 1) { if (acc[0][0][0] == 123.456f) p.y[0] = 1.f; return; }
    const n3d_epilogue& E = p.epi;
    const int64_t plane = (int64_t)p.OH * p.OW;
    float rs[16], bs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int oc = o < p.O ? o : p.O - 1;
        rs[r] = E.const_scale * (E.row_scale ? E.row_scale[(int64_t)n * (E.row_scale_stride ? E.row_scale_stride : p.O) + oc] : 1.f);
        bs[r] = E.bias ? E.bias[oc] : 0.f;
    }
    const float nstr = E.noise ? E.noise_strength[0] : 0.f;
    const bool lrelu = E.act == N3D_ACT_LRELU, linear = E.act == N3D_ACT_LINEAR;
    // Stores: phases (a,0) and (a,1) of one position are horizontally adjacent output pixels (2x, 2x+1): write them as
    // ONE 8-byte store per lane so a half-wave covers a dense 256-byte row segment (stride-2 dword stores left every
    // cache line half written: the kernel was store-bound at ~250 GB/s).  Rows have odd pitch (2W+1), so the pair is
    // only 4-byte aligned; global memory accepts dword-aligned 8-byte stores.
    struct __attribute__((packed, aligned(4))) pair_t { float v0, v1; };
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int pos = (wn * NT + nt) * 32 + l31;
        const int gy = y0 + pos / TW, gx = x0 + pos % TW;
        if (gy >= p.GH || gx >= p.GW) continue;
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
            const int oy = 2 * gy + pa, ox = 2 * gx;
            if (oy >= p.OH) continue;
            const bool two = ox + 1 < p.OW;
            const int64_t po = (int64_t)oy * p.OW + ox;
            if (p.partial) {
                float* dst = p.partial + ((int64_t)ks * p.N + n) * p.O * plane + po;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (o >= p.O) continue;
                    if (two) *reinterpret_cast<pair_t*>(dst + (int64_t)o * plane) = pair_t{acc[nt][pa * 2][r], acc[nt][pa * 2 + 1][r]};
                    else dst[(int64_t)o * plane] = acc[nt][pa * 2][r];
                }
                continue;
            }
            const float nz0 = E.noise ? E.noise[po] * nstr : 0.f;
            const float nz1 = (E.noise && two) ? E.noise[po + 1] * nstr : 0.f;
            const int64_t yplane = (int64_t)p.OH * p.yrs;
            float* dst = p.y + (int64_t)n * p.ybs + (int64_t)oy * p.yrs + ox;
            const float* res = E.residual ? E.residual + (int64_t)n * E.residual_batch_stride + po : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (o >= p.O) continue;
                float v[2] = {acc[nt][pa * 2][r] * rs[r] + nz0 + bs[r], acc[nt][pa * 2 + 1][r] * rs[r] + nz1 + bs[r]};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (lrelu) v[q] = v[q] > 0.f ? v[q] : v[q] * E.alpha;
                    else if (!linear) v[q] = conv_act_generic(v[q], E.act, E.alpha);
                    v[q] *= E.gain;
                    if (E.clamp >= 0.f) v[q] = fminf(fmaxf(v[q], -E.clamp), E.clamp);
                    if (res && (q == 0 || two)) v[q] += res[(int64_t)o * plane + q];
                }
                if (two) *reinterpret_cast<pair_t*>(dst + (int64_t)o * yplane) = pair_t{v[0], v[1]};
                else dst[(int64_t)o * yplane] = v[0];
            }
        }
    }
}

template <int TH, int TW>
static int launch_conv_up(ConvParams& p, int ksplit_req, hipStream_t stream) {
    constexpr int BM = 64, ICB = 16;
    p.tiles_x = cdiv(p.GW, TW);
    p.tiles_y = cdiv(p.GH, TH);
    p.nphase = 1;
    const int max_split = cdiv(p.I, ICB);
    p.ksplit = ksplit_req < 1 ? 1 : (ksplit_req > max_split ? max_split : ksplit_req);
    p.ic_per_split = cdiv(cdiv(p.I, p.ksplit), ICB) * ICB;
    p.ksplit = cdiv(p.I, p.ic_per_split);
    if (p.ksplit == 1) p.partial = nullptr;
    const int64_t gz = (int64_t)p.N * p.ksplit;
    if (gz > 65535) return n3d_set_error("conv2d: grid.z %lld too large", (long long)gz);
    dim3 grid(p.tiles_x * p.tiles_y, cdiv(p.O, BM), (unsigned)gz);
    hipLaunchKernelGGL((conv2d_up_mfma_kernel<TH, TW>), grid, dim3(256), 0, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}

// split-K second pass: y = epilogue(sum_ks partial[ks])
__global__ __launch_bounds__(256) void conv2d_splitk_epilogue_kernel(const float* __restrict__ partial, float* __restrict__ y,
                                                                      int ksplit, int N, int O, int OH, int OW, int64_t ybs,
                                                                      int64_t yrs, n3d_epilogue epi) {
    const int64_t plane = (int64_t)OH * OW;
    const int64_t total = (int64_t)N * O * plane;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int k = 0; k < ksplit; ++k) v += partial[(int64_t)k * total + i];
        const int64_t pl = i / plane;
        const int pix = (int)(i % plane);
        const int n = (int)(pl / O), o = (int)(pl % O);
        const int oy = pix / OW, ox = pix % OW;
        v = n3d_apply_epilogue(v, epi, n, o, O, oy, ox, OH, OW);
        y[(int64_t)n * ybs + ((int64_t)o * OH + oy) * yrs + ox] = v;
    }
}

// w [O,I,k,k] -> wt [k*k][I][O];  wsq[o,i] = sum_k w^2
__global__ __launch_bounds__(256) void conv2d_prep_weight_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                                 float* __restrict__ wsq, int O, int I, int KK) {
    const int OP = (O + 3) & ~3;                           // row pitch: 16-byte rows, zero padded
    const int64_t total = (int64_t)OP * I;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / OP), o = (int)(e % OP);    // o fastest -> coalesced writes
        const float* src = w + ((int64_t)o * I + i) * KK;
        float s = 0.f;
        for (int t = 0; t < KK; ++t) {
            const float v = o < O ? src[t] : 0.f;
            wt[((int64_t)t * I + i) * OP + o] = v;
            s += v * v;
        }
        if (wsq && o < O) wsq[(int64_t)o * I + i] = s;
    }
}

extern "C" int n3d_conv2d_prep_weight(const float* w, float* wt, float* wsq, int O, int I, int ksize, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(w && wt && O > 0 && I > 0 && (ksize == 1 || ksize == 3), "conv2d_prep_weight: bad arguments");
    const int64_t total = (int64_t)O * I;
    const int grid = (int)(cdiv64(total, 256) > 4096 ? 4096 : cdiv64(total, 256));
    N3dProfScope prof(N3D_K_MISC, stream, 0.0, 8.0 * total * ksize * ksize);
    hipLaunchKernelGGL(conv2d_prep_weight_kernel, dim3(grid), dim3(256), 0, stream, w, wt, wsq, O, I, ksize * ksize);
    N3D_LAUNCH_CHECK();
    return 0;
}

template <int MT, int NT, int WM, int WN, int TH, int TW, int ICB, int KS, int MODE>
static int launch_conv(ConvParams& p, int ksplit_req, hipStream_t stream) {
    constexpr int BM = WM * MT * 32;
    p.tiles_x = cdiv(p.GW, TW);
    p.tiles_y = cdiv(p.GH, TH);
    const int max_split = cdiv(p.I, ICB);
    p.ksplit = ksplit_req < 1 ? 1 : (ksplit_req > max_split ? max_split : ksplit_req);
    p.ic_per_split = cdiv(cdiv(p.I, p.ksplit), ICB) * ICB;
    p.ksplit = cdiv(p.I, p.ic_per_split);
    if (p.ksplit == 1) p.partial = nullptr;
    const int64_t gz = (int64_t)p.N * p.nphase * p.ksplit;
    if (gz > 65535) return n3d_set_error("conv2d: grid.z %lld too large", (long long)gz);
    dim3 grid(p.tiles_x * p.tiles_y, cdiv(p.O, BM), (unsigned)gz);
    hipLaunchKernelGGL((conv2d_mfma_kernel<MT, NT, WM, WN, TH, TW, ICB, KS, MODE>), grid, dim3(256), 0, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}

extern "C" int n3d_conv2d(const n3d_conv2d_desc* d, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(d != nullptr, "conv2d: null descriptor");
    N3D_CHECK(d->N >= 0 && d->I > 0 && d->O > 0 && d->H > 0 && d->W > 0, "conv2d: bad shape");
    N3D_CHECK(d->ksize == 1 || d->ksize == 3, "conv2d: ksize %d unsupported (1 or 3)", d->ksize);
    N3D_CHECK(d->mode >= 0 && d->mode <= 2, "conv2d: unknown mode %d", d->mode);
    N3D_CHECK(d->mode == 0 || d->ksize == 3, "conv2d: strided / transposed modes need ksize 3");
    N3D_CHECK(d->epi.act >= N3D_ACT_LINEAR && d->epi.act <= N3D_ACT_SWISH, "conv2d: unknown activation %d", d->epi.act);
    N3D_CHECK(d->epi.noise == nullptr || d->epi.noise_strength != nullptr, "conv2d: noise without noise_strength");
    if (d->N == 0) return 0;
    N3D_CHECK(!d->rgb_partial, "conv2d: the fused toRGB (rgb_partial) is an option of n3d_conv2d_bf16x3");
    N3D_CHECK(d->x && d->wt && d->y, "conv2d: null tensor");
    N3D_CHECK(((uintptr_t)d->wt & 15) == 0, "conv2d: wt must be 16-byte aligned");
    ConvParams p;
    p.x = d->x; p.wt = d->wt; p.style = d->style; p.y = d->y; p.partial = d->workspace;
    p.N = d->N; p.I = d->I; p.O = d->O; p.OP = (d->O + 3) & ~3; p.H = d->H; p.W = d->W;
    p.xbs = d->x_batch_stride; p.ybs = d->y_batch_stride; p.epi = d->epi;
    p.style_stride = d->style_stride ? d->style_stride : d->I;
    N3D_CHECK((d->wt_batch_stride & 15) == 0, "conv2d: wt_batch_stride must be a multiple of 16 bytes");
    p.wbs = d->wt_batch_stride / 4;
    p.nphase = 1;
    p.dbg = n3d_tune("N3D_CONV_DBG", 0);
    if (d->mode == 0) { p.OH = d->H; p.OW = d->W; p.GH = p.OH; p.GW = p.OW; }
    else if (d->mode == 1) {
        N3D_CHECK(d->H >= 3 && d->W >= 3, "conv2d: input too small for stride-2 3x3");
        p.OH = (d->H - 3) / 2 + 1; p.OW = (d->W - 3) / 2 + 1; p.GH = p.OH; p.GW = p.OW;
    } else { p.OH = 2 * d->H + 1; p.OW = 2 * d->W + 1; p.GH = d->H + 1; p.GW = d->W + 1; p.nphase = 4; }
    p.yrs = d->y_row_stride ? d->y_row_stride : p.OW;
    N3D_CHECK(p.yrs >= p.OW, "conv2d: y_row_stride smaller than the output width");
    N3D_CHECK(d->x_row_stride == 0 || d->x_row_stride == d->W, "conv2d: the fp32 kernels take dense input rows");
    N3D_CHECK(!d->epi.round_f16, "conv2d: round_f16 is not supported by the fp32 kernels");
    N3D_CHECK(d->x_layout == N3D_LAYOUT_NCHW_F32 && d->y_layout == N3D_LAYOUT_NCHW_F32, "conv2d: the fp32 kernels read and write float32 NCHW (x_layout / y_layout 0)");
    N3D_CHECK(!d->side_split8, "conv2d: side_split8 is written by the split-bf16 1x1 kernel only");
    N3D_CHECK(!d->epi.residual_up_filter || (d->epi.residual && d->mode != 2 && p.OH % 2 == 0 && p.OW % 2 == 0),
              "conv2d: residual_up_filter needs a residual, an even output size and a non-transposed mode");
    N3D_CHECK(d->ksplit <= 1 || d->workspace != nullptr, "conv2d: ksplit > 1 needs a workspace");
    N3D_CHECK(d->I <= 1024 * (d->ksplit < 1 ? 1 : d->ksplit), "conv2d: more than 1024 input channels per K-split");
    N3D_CHECK((int64_t)9 * d->I * ((d->O + 3) & ~3) < (1 << 27), "conv2d: weight tensor too large for 27-bit staging offsets");
    const bool wide = p.GW > 16;
    const double flops = 2.0 * d->N * (double)d->O * d->I * d->ksize * d->ksize *
                         (d->mode == 2 ? (double)d->H * d->W : (double)p.OH * p.OW);
    const double bytes = 4.0 * ((double)d->N * d->I * d->H * d->W + (double)d->N * d->O * p.OH * p.OW +
                                (double)d->O * d->I * d->ksize * d->ksize);
    N3dProfScope prof(N3D_K_CONV2D, stream, flops, bytes);
    int rc;
    if (d->ksize == 3) {
        if (d->mode == 0)      rc = wide ? launch_conv<2, 2, 2, 2, 4, 32, 8, 3, 0>(p, d->ksplit, stream) : launch_conv<2, 2, 2, 2, 8, 16, 8, 3, 0>(p, d->ksplit, stream);
        else if (d->mode == 1) rc = wide ? launch_conv<2, 2, 2, 2, 4, 32, 8, 3, 1>(p, d->ksplit, stream) : launch_conv<2, 2, 2, 2, 8, 16, 8, 3, 1>(p, d->ksplit, stream);
        else                   rc = wide ? launch_conv_up<4, 32>(p, d->ksplit, stream) : launch_conv_up<8, 16>(p, d->ksplit, stream);
    } else {
        if (d->O > 64) rc = wide ? launch_conv<2, 2, 2, 2, 4, 32, 32, 1, 0>(p, d->ksplit, stream) : launch_conv<2, 2, 2, 2, 8, 16, 32, 1, 0>(p, d->ksplit, stream);
        else           rc = wide ? launch_conv<1, 1, 1, 4, 4, 32, 32, 1, 0>(p, d->ksplit, stream) : launch_conv<1, 1, 1, 4, 8, 16, 32, 1, 0>(p, d->ksplit, stream);
    }
    if (rc) return rc;
    if (p.ksplit > 1) {
        const int64_t total = (int64_t)p.N * p.O * p.OH * p.OW;
        const int grid = (int)(cdiv64(total, 256) > 2048 ? 2048 : cdiv64(total, 256));
        hipLaunchKernelGGL(conv2d_splitk_epilogue_kernel, dim3(grid), dim3(256), 0, stream, (const float*)p.partial, p.y,
                           p.ksplit, p.N, p.O, p.OH, p.OW, p.ybs, p.yrs, p.epi);
        N3D_LAUNCH_CHECK();
    }
    return 0;
}
