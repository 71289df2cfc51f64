// 3x3 stride-2 convolution (the down-sampling encoder layers, mode 1) on the split-bf16 path for gfx950.
//
//     y[o, oy, ox] = sum_{i, ky, kx} w[o,i,ky,kx] * x[i, 2 oy + ky, 2 ox + kx]          (padding 0: x is the FIR-padded image)
//
// A strided gather cannot feed the MFMA B fragments from one LDS patch without 2-way bank conflicts (pixels 2 apart),
// so the input is treated as its FOUR POLYPHASE images x_{py,px}[a,b] = x[2a+py, 2b+px]:
//     y[oy,ox] = sum_{py,px} sum_{dy in D(py), dx in D(px)} w[2dy+py, 2dx+px] * x_{py,px}[oy+dy, ox+dx],   D(0)={0,1}, D(1)={0}
// i.e. four stride-1 sub-convolutions with 4 / 2 / 2 / 1 taps.  The K loop runs over (16-channel chunk, phase): one stage
// = the (TH+1) x (TW+1) patch of ONE phase image (de-interleaved while staging: the global loads of a stage are 8-byte
// strided, the sibling phase picks the other half of the lines up from L2) + the weights of that phase's taps only.
// Everything else is the stride-1 kernel of conv2d_bf16x3.hip: 8 waves, tile = 64 channels x (16 x 32) output pixels,
// double-buffered LDS, ping-pong staging/MFMA roles, split-bf16 operands (three bf16 MFMAs per fragment pair), fused
// epilogue, optional split-K over the input channels.
// Replaces the ATen conv2d(stride=2) issued by conv2d_resample (torch_utils/ops/conv2d_resample.py:108-111) for
// Conv2dLayer(down=2) (tat/networks_stylegan2.py:173-183) in the StyleUNet encoder (networks_stylegan2_styleunet.py:97-115).
#include <type_traits>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ConvS2Params {
    const float* x; const bf16x8* wt16; const float* style; float* y; float* partial;
    int N, I, O, OP64, H, W, OH, OW;
    int tiles_x, tiles_y, tiles_m, ksplit, ic_per_split;
    int64_t xbs, ybs, style_stride, yrs;
    int xrs;                                  // input row pitch in floats (>= W; the FIR in front writes 16-byte-aligned rows)
    n3d_epilogue epi;
};

__device__ __noinline__ float convs2_act_generic(float v, int act, float alpha) { return n3d_act(v, act, alpha); }

__device__ __forceinline__ void s2_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v[i];
        hi[i] = h;
        lo[i] = (__bf16)(v[i] - (float)h);
    }
}

__global__ __launch_bounds__(512, 2) void conv2d_s2_bf16x3_kernel(ConvS2Params p) {
    constexpr int NW = 8, NT_ = 512;
    constexpr int BM = 64, TH = 16, TW = 32, ICB = 16;
    constexpr int PH = TH + 1, PW = TW + 1, PPIX = PH * PW;               // 17 x 33 = 561 patch pixels of one phase image
    constexpr int B_ITEMS = 2 * PPIX;
    constexpr int B_PER_T = (B_ITEMS + NT_ - 1) / NT_;                    // 3
    constexpr int A_SZ = 4 * 2 * BM, B_SZ = 2 * PPIX;                     // A: [slot 4][half][row]
    __shared__ bf16x8 A_hi[2 * A_SZ], A_lo[2 * A_SZ];
    __shared__ bf16x8 B_hi[2 * B_SZ], B_lo[2 * B_SZ];
    __shared__ float s_style[1024];

    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    int lb;                                                               // XCD-aware logical block id, M-tile fastest (conv2d_bf16x3.hip)
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int mt_i = lb % p.tiles_m; lb /= p.tiles_m;
    const int tile_i = lb % (p.tiles_x * p.tiles_y); lb /= (p.tiles_x * p.tiles_y);
    const int tx = tile_i % p.tiles_x, ty = tile_i / p.tiles_x;
    const int m0 = mt_i * BM;
    const int ks = lb % p.ksplit, n = lb / p.ksplit;
    const int y0 = ty * TH, x0 = tx * TW;
    const int ic_begin = ks * p.ic_per_split;
    const int ic_end = min(p.I, ic_begin + p.ic_per_split);
    const int nstage = (ic_end - ic_begin) / ICB * 4;                     // stage = (chunk, phase): st -> chunk st >> 2, phase st & 3
    const int KC = p.I / ICB;
    const int HW = p.H * p.xrs;               // plane pitch

    for (int i = tid; i < ic_end - ic_begin; i += NT_) s_style[i] = p.style ? p.style[(int64_t)n * p.style_stride + ic_begin + i] : 1.f;

    // A staging: thread owns (row, half, hl) of tap slot a_s0 and a_s0 + 2; slot j of phase (py,px): dy = j / nx, dx = j % nx,
    // nx = px ? 1 : 2, tap (ky,kx) = (py ? 1 : 2 dy, px ? 1 : 2 dx)
    const int a_row = tid & 63, a_q = (tid >> 6) & 3, a_half = a_q & 1, a_hl = a_q >> 1, a_s0 = tid >> 8;
    const int64_t a_tap_stride = (int64_t)KC * 4 * p.OP64, a_chunk_stride = (int64_t)4 * p.OP64;
    const bf16x8* a_src = p.wt16 + ((int64_t)(ic_begin / ICB) * 4 + a_hl * 2 + a_half) * p.OP64 + m0 + a_row;
    bf16x8* a_dst = (a_hl ? A_lo : A_hi) + a_half * BM + a_row + a_s0 * 2 * BM;
    // B staging: work item e = tid + 512 j -> (half, patch pixel); global offset of the phase-(0,0) pixel
    int b_goff[B_PER_T], b_iy[B_PER_T], b_ix[B_PER_T];
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j) {
        const int e = tid + j * NT_;
        const int hf = e / PPIX, pp = e % PPIX;
        b_iy[j] = e < B_ITEMS ? 2 * (y0 + pp / PW) : p.H;                 // rows/cols at or beyond H/W are never loaded
        b_ix[j] = 2 * (x0 + pp % PW);
        b_goff[j] = (hf * 8 * HW + b_iy[j] * p.xrs + b_ix[j]) * 4;          // byte offset of the phase-(0,0) pixel inside the sample
    }
    // buffer loads: descriptor per sample (SGPRs) + 32-bit lane offset + (channel, phase) offset in an SGPR
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.xbs), 0, p.I * HW * 4, 0x00020000);

    bf16x8 ra[2];
    float rb[B_PER_T][8];
    bool rb_ok[B_PER_T];
    auto ntaps_of = [](int g) { return (g & 1 ? 1 : 2) * (g & 2 ? 1 : 2); };
    auto load_stage = [&](int st) {                       // issue only; consumed in store_stage (after the MFMA block)
        const int c = st >> 2, g = st & 3, py = g >> 1, px = g & 1;
        const int nx = px ? 1 : 2, nt = ntaps_of(g);
        const bf16x8* as = a_src + c * a_chunk_stride;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = a_s0 + 2 * jj;
            if (j < nt) {
                const int ky = py ? 1 : 2 * (j / nx), kx = px ? 1 : 2 * (j % nx);
                ra[jj] = as[(ky * 3 + kx) * a_tap_stride];
            }
        }
#pragma unroll
        for (int j = 0; j < B_PER_T; ++j) {
            rb_ok[j] = b_iy[j] + py < p.H && b_ix[j] + px < p.W;
            const int voff = rb_ok[j] ? b_goff[j] : (int)0x80000000;          // beyond the buffer: the load returns 0
#pragma unroll
            for (int ch = 0; ch < 8; ++ch)
                rb[j][ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, voff, (((ic_begin + c * ICB + ch) * p.H + py) * p.xrs + px) * 4, 0));
        }
    };
    auto store_stage = [&](int st) {
        const int c = st >> 2, g = st & 3, nt = ntaps_of(g);
        const int bo_a = (st & 1) ? A_SZ : 0, bo_b = (st & 1) ? B_SZ : 0;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            if (a_s0 + 2 * jj < nt) a_dst[bo_a + jj * 2 * 2 * BM] = ra[jj];
#pragma unroll
        for (int j = 0; j < B_PER_T; ++j) {
            const int e = tid + j * NT_;
            if (e >= B_ITEMS) continue;
            const int hf = e / PPIX;
            float v[8];
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) v[ch] = rb[j][ch] * s_style[c * ICB + hf * 8 + ch];
            bf16x8 hi, lo;
            s2_split8(v, hi, lo);
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
    const int a_frag = half * BM + l31;                                   // + slot*2*BM + mt*32
    const int b_frag = half * PPIX + (wn * 2) * PW + l31;                 // + nt*PW + dy*PW + dx
    auto mfma_taps = [&](int st, auto ny_c, auto nx_c) {                  // taps (dy, dx), dy < NY, dx < NX; slot = dy*NX + dx
        constexpr int NY = decltype(ny_c)::value, NX = decltype(nx_c)::value;
        const int bo_a = (st & 1) ? A_SZ : 0, bo_b = (st & 1) ? B_SZ : 0;
        __builtin_amdgcn_s_setprio(1);
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int dy = 0; dy < NY; ++dy)
#pragma unroll
            for (int dx = 0; dx < NX; ++dx) {
                const int slot = dy * NX + dx, boff = dy * PW + dx;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) { ah[mt] = A_hi[bo_a + slot * 2 * BM + a_frag + mt * 32]; al[mt] = A_lo[bo_a + slot * 2 * BM + a_frag + mt * 32]; }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) { bh[nt] = B_hi[bo_b + b_frag + nt * PW + boff]; bl[nt] = B_lo[bo_b + b_frag + nt * PW + boff]; }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    }
            }
        __builtin_amdgcn_s_setprio(0);
    };
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    auto mfma_block = [&](int st) {
        switch (st & 3) {                                                 // phase (py, px): NY = py ? 1 : 2, NX = px ? 1 : 2
            case 0: mfma_taps(st, I2{}, I2{}); break;
            case 1: mfma_taps(st, I2{}, I1{}); break;
            case 2: mfma_taps(st, I1{}, I2{}); break;
            default: mfma_taps(st, I1{}, I1{}); break;
        }
    };
    // registers hold the raw data of stage st+1 at the top of iteration st; waves 0-3 stage before their MFMA block, 4-7 after
    if (nstage > 1) load_stage(1);
    const bool stage_first = wn < NW / 2;
    for (int st = 0; st < nstage; ++st) {
        if (stage_first) {
            if (st + 1 < nstage) store_stage(st + 1);
            if (st + 2 < nstage) load_stage(st + 2);
            mfma_block(st);
        } else {
            mfma_block(st);
            if (st + 1 < nstage) store_stage(st + 1);
            if (st + 2 < nstage) load_stage(st + 2);
        }
        __syncthreads();
    }

    // epilogue (C/D layout: col = lane&31 = pixel, row = (r&3) + 8*(r>>2) + 4*(lane>>5) = channel)
    const n3d_epilogue& E = p.epi;
    const int64_t plane = (int64_t)p.OH * p.OW;
    const float nstr = E.noise ? E.noise_strength[0] : 0.f;
    const bool lrelu = E.act == N3D_ACT_LRELU, linear = E.act == N3D_ACT_LINEAR;
    // per-channel factors through LDS (the K loop is over: s_style is free) -> no dependent global loads in the store loop
    float* s_rs = s_style, *s_bs = s_style + BM;
    if (tid < BM) {
        const int o = min(m0 + tid, p.O - 1);
        s_rs[tid] = E.const_scale * (E.row_scale ? E.row_scale[(int64_t)n * (E.row_scale_stride ? E.row_scale_stride : p.O) + o] : 1.f);
        s_bs[tid] = E.bias ? E.bias[o] : 0.f;
    }
    __syncthreads();
    const bool simple = !p.partial && (linear || (lrelu && E.alpha >= 0.f && E.alpha <= 1.f)) && !E.residual && m0 + BM <= p.O;
    const float alpha_eff = lrelu ? E.alpha : 1.f, clamp_eff = E.clamp >= 0.f ? E.clamp : INFINITY;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int oy = y0 + wn * 2 + nt, ox = x0 + l31;
        if (oy >= p.OH || ox >= p.OW) continue;
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
        if (simple) {       // straight-line common case: leaky ReLU = max(v, alpha v) (linear: alpha 1), no clamp = clamp at +inf
            float* d0 = dst + (int64_t)(m0 + 4 * half) * yplane;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ol = mt * 32 + (r & 3) + 8 * (r >> 2);
                    float v = acc[mt][nt][r] * s_rs[ol + 4 * half] + nz + s_bs[ol + 4 * half];
                    v = fmaxf(v, v * alpha_eff) * E.gain;
                    d0[(int64_t)ol * yplane] = fminf(fmaxf(v, -clamp_eff), clamp_eff);
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
                float v = acc[mt][nt][r] * s_rs[o - m0] + nz + s_bs[o - m0];
                if (lrelu) v = v > 0.f ? v : v * E.alpha;
                else if (!linear) v = convs2_act_generic(v, E.act, E.alpha);
                v *= E.gain;
                if (E.clamp >= 0.f) v = fminf(fmaxf(v, -E.clamp), E.clamp);
                if (res) v += res[(int64_t)o * plane];
                dst[(int64_t)o * yplane] = v;
            }
    }
}

// split-K reduction + epilogue kernel of conv2d_bf16x3.hip
int conv16_splitk_epilogue_launch(const float* partial, float* y, int ksplit, int N, int O, int OH, int OW, int64_t ybs, int64_t yrs,
                                  const n3d_epilogue& epi, hipStream_t stream);

// called by n3d_conv2d_bf16x3 for ksize == 3, mode == 1 (descriptor already validated for the common fields)
int conv2d_sk_s2_bf16x3_try_launch(const n3d_conv2d_desc* d, hipStream_t stream);   // conv2d_sk_bf16x3.hip (few-pixel stride-2 layers; 1 = not its layer)

int conv2d_s2_bf16x3_launch(const n3d_conv2d_desc* d, hipStream_t stream) {
    N3D_CHECK(d->H >= 3 && d->W >= 3, "conv2d_bf16x3: stride-2 input smaller than the kernel");
    {                                                                      // few-pixel layers (<= 17 x 17 inputs at batch 4): K split inside the workgroup, one launch
        const int r = conv2d_sk_s2_bf16x3_try_launch(d, stream);
        if (r <= 0) return r;
    }
    ConvS2Params p;
    p.x = d->x; p.wt16 = (const bf16x8*)d->wt; p.style = d->style; p.y = d->y; p.partial = d->workspace;
    p.N = d->N; p.I = d->I; p.O = d->O; p.OP64 = (d->O + 63) / 64 * 64; p.H = d->H; p.W = d->W;
    p.OH = (d->H - 3) / 2 + 1; p.OW = (d->W - 3) / 2 + 1;
    p.xbs = d->x_batch_stride; p.ybs = d->y_batch_stride; p.epi = d->epi;
    p.style_stride = d->style_stride ? d->style_stride : d->I;
    p.yrs = d->y_row_stride ? d->y_row_stride : p.OW;
    p.xrs = d->x_row_stride ? (int)d->x_row_stride : d->W;
    N3D_CHECK(p.xrs >= d->W, "conv2d_bf16x3: x_row_stride smaller than the input width");
    N3D_CHECK(p.yrs >= p.OW, "conv2d_bf16x3: y_row_stride smaller than the output width");
    N3D_CHECK(!d->epi.residual_up_filter, "conv2d_bf16x3: residual_up_filter is only supported by the 1x1 kernel");
    N3D_CHECK((int64_t)d->I * d->H * (d->x_row_stride ? d->x_row_stride : d->W) * 4 < (1ll << 31), "conv2d_bf16x3: one sample's input exceeds 2 GiB (32-bit buffer offsets)");
    p.tiles_x = cdiv(p.OW, 32); p.tiles_y = cdiv(p.OH, 16); p.tiles_m = cdiv(p.O, 64);
    const int max_split = d->I / 16;
    p.ksplit = d->ksplit < 1 ? 1 : (d->ksplit > max_split ? max_split : d->ksplit);
    p.ic_per_split = cdiv(cdiv(d->I, p.ksplit), 16) * 16;
    p.ksplit = cdiv(d->I, p.ic_per_split);
    N3D_CHECK(p.ksplit == 1 || d->workspace != nullptr, "conv2d_bf16x3: ksplit > 1 needs a workspace");
    N3D_CHECK(p.ic_per_split <= 1024, "conv2d_bf16x3: more than 1024 input channels per K-split");
    if (p.ksplit == 1) p.partial = nullptr;
    const int64_t nblk = (int64_t)p.tiles_x * p.tiles_y * p.tiles_m * p.N * p.ksplit;
    N3D_CHECK(nblk < (1ll << 31), "conv2d_bf16x3: grid too large");
    const double flops = 2.0 * d->N * (double)d->O * d->I * 9 * (double)p.OH * p.OW;
    const double bytes = 4.0 * ((double)d->N * d->I * d->H * d->W + (double)d->N * d->O * p.OH * p.OW + (double)d->O * d->I * 9);
    N3dProfScope prof(N3D_K_CONV2D_BF16X3, stream, flops, bytes);
    hipLaunchKernelGGL(conv2d_s2_bf16x3_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, p);
    N3D_LAUNCH_CHECK();
    if (p.ksplit > 1) return conv16_splitk_epilogue_launch(p.partial, p.y, p.ksplit, p.N, p.O, p.OH, p.OW, p.ybs, p.yrs, p.epi, stream);
    return 0;
}
