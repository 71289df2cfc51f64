// filtered_lrelu for gfx950: bias -> x`up` up-sampling FIR -> leaky ReLU * gain, clamp -> x`down` down-sampling FIR in ONE
// kernel; the up-sampled intermediate (up^2 times the input) lives only in LDS.
//
// Replaces filtered_lrelu_plugin.filtered_lrelu (reference torch_utils/ops/filtered_lrelu.cpp:20, kernel
// filtered_lrelu.cu:143-144; semantics = _filtered_lrelu_ref, filtered_lrelu.py:123-155).  StyleGAN3's alias-free
// non-linearity: the Next3D generator imports its module (superresolution.py:22) but never executes it, so this kernel is
// written for correctness and a sane memory pattern, not tuned: one workgroup = one (sample, channel) plane tile of
// TH x TW outputs;
//   1. the input patch the tile depends on (+ bias) -> LDS (coalesced rows, zeros outside the image);
//   2. every work item evaluates intermediate samples  m = clamp(lrelu(sum_taps fu * zero_stuffed(x)) * gain)  — only the
//      taps that hit a real (non-stuffed) sample are visited (every up-th tap) — into LDS;
//   3. one output per work item:  y = sum_taps fd * m[oy*down + ty][ox*down + tx].
// HBM traffic = input patch + output tile (the reference's fused kernel has the same property; its un-fused fallback moves
// the up^2-times larger intermediate four times).
#include "common.h"

struct FlrParams {
    const float* x; const float* fu; const float* fd; const float* b; float* y;
    int N, C, H, W, OH, OW;
    int up, down, fuh, fuw, fdh, fdw, px0, py0;
    int MH, MW, IH, IW;          // per-tile extents: intermediate (MH x MW), input patch (IH x IW)
    int tiles_x, tiles_y;
    float gain, slope, clamp;    // clamp < 0: none
    int flip;
};

constexpr int FLR_TH = 8, FLR_TW = 32, FLR_NT = 256;

__device__ __forceinline__ int floordiv(int a, int b) { const int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }

__global__ __launch_bounds__(FLR_NT) void filtered_lrelu_kernel(FlrParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;                                   // [IH][IW]
    float* s_mid = s_in + p.IH * p.IW;                    // [MH][MW]
    float* s_fu = s_mid + p.MH * p.MW;                    // [fuh][fuw], stored so that tap t multiplies sample (pos + t)
    float* s_fd = s_fu + p.fuh * p.fuw;
    const int tid = threadIdx.x;
    // 1-D grid (N * C planes x tiles: no 65535 limit on the plane count)
    const int tiles = p.tiles_x * p.tiles_y, tile = (int)(blockIdx.x % tiles);
    const int tx_i = tile % p.tiles_x, ty_i = tile / p.tiles_x;
    const int nc = (int)(blockIdx.x / tiles), c = nc % p.C;
    const int oy0 = ty_i * FLR_TH, ox0 = tx_i * FLR_TW;
    const int my0 = oy0 * p.down, mx0 = ox0 * p.down;     // top-left intermediate sample of the tile
    const int iy0 = floordiv(my0 - p.py0, p.up), ix0 = floordiv(mx0 - p.px0, p.up);   // first input sample any tap can reach

    // filters: upfirdn2d correlates with the FLIPPED taps unless flip_filter (upfirdn2d.py:203-205); the up-sampling gain
    // up^2 (filtered_lrelu.py:150) is folded into fu
    const float ug = (float)(p.up * p.up);
    for (int i = tid; i < p.fuh * p.fuw; i += FLR_NT) s_fu[i] = (p.fu ? p.fu[p.flip ? i : p.fuh * p.fuw - 1 - i] : 1.f) * ug;
    for (int i = tid; i < p.fdh * p.fdw; i += FLR_NT) s_fd[i] = p.fd ? p.fd[p.flip ? i : p.fdh * p.fdw - 1 - i] : 1.f;
    const float bias = p.b ? p.b[c] : 0.f;
    const float* plane = p.x + (int64_t)nc * p.H * p.W;
    for (int i = tid; i < p.IH * p.IW; i += FLR_NT) {
        const int iy = iy0 + i / p.IW, ix = ix0 + i % p.IW;
        s_in[i] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? plane[(int64_t)iy * p.W + ix] + bias : 0.f;
    }
    __syncthreads();

    for (int e = tid; e < p.MH * p.MW; e += FLR_NT) {
        const int my = my0 + e / p.MW, mx = mx0 + e % p.MW;
        // tap ty reads zero-stuffed sample uu = my + ty - py0; only uu % up == 0 carries data
        int ty0 = (p.up - ((my - p.py0) % p.up + p.up) % p.up) % p.up;
        int tx0 = (p.up - ((mx - p.px0) % p.up + p.up) % p.up) % p.up;
        float acc = 0.f;
        for (int ty = ty0; ty < p.fuh; ty += p.up) {
            const int r = (my + ty - p.py0) / p.up - iy0;            // exact division; 0 <= r < IH by construction
            for (int tx = tx0; tx < p.fuw; tx += p.up) {
                const int q = (mx + tx - p.px0) / p.up - ix0;
                acc += s_fu[ty * p.fuw + tx] * s_in[r * p.IW + q];
            }
        }
        float v = (acc < 0.f ? acc * p.slope : acc) * p.gain;
        if (p.clamp >= 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
        s_mid[e] = v;
    }
    __syncthreads();

    const int oy = oy0 + tid / FLR_TW, ox = ox0 + tid % FLR_TW;
    if (oy >= p.OH || ox >= p.OW) return;
    const float* m = s_mid + (tid / FLR_TW) * p.down * p.MW + (tid % FLR_TW) * p.down;
    float acc = 0.f;
    for (int ty = 0; ty < p.fdh; ++ty)
        for (int tx = 0; tx < p.fdw; ++tx) acc += s_fd[ty * p.fdw + tx] * m[ty * p.MW + tx];
    p.y[((int64_t)nc * p.OH + oy) * p.OW + ox] = acc;
}

extern "C" int n3d_filtered_lrelu(const float* x, const float* fu, const float* fd, const float* b, float* y, int N, int C, int H, int W,
                                  int fuh, int fuw, int fdh, int fdw, int up, int down, int px0, int px1, int py0, int py1, float gain,
                                  float slope, float clamp, int flip, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && C > 0 && H > 0 && W > 0 && up >= 1 && down >= 1, "filtered_lrelu: bad shape");
    N3D_CHECK(fuh >= 1 && fuw >= 1 && fdh >= 1 && fdw >= 1, "filtered_lrelu: bad filter size");
    N3D_CHECK((fu != nullptr || (fuh == 1 && fuw == 1)) && (fd != nullptr || (fdh == 1 && fdw == 1)), "filtered_lrelu: a missing filter is a single unit tap");
    N3D_CHECK(gain > 0.f && slope >= 0.f, "filtered_lrelu: gain must be positive and slope non-negative");
    const int mid_h = H * up + py0 + py1 - fuh + 1, mid_w = W * up + px0 + px1 - fuw + 1;       // size after the up FIR
    N3D_CHECK(mid_h >= fdh && mid_w >= fdw, "filtered_lrelu: the up-sampled image is smaller than the filters");
    FlrParams p;
    p.OH = (mid_h - fdh + down) / down; p.OW = (mid_w - fdw + down) / down;
    if (N == 0) return 0;
    N3D_CHECK(x && y, "filtered_lrelu: null tensor");
    N3D_CHECK((int64_t)N * C * cdiv(p.OW, FLR_TW) * cdiv(p.OH, FLR_TH) < (1ll << 31), "filtered_lrelu: grid too large");
    p.x = x; p.fu = fu; p.fd = fd; p.b = b; p.y = y;
    p.N = N; p.C = C; p.H = H; p.W = W;
    p.up = up; p.down = down; p.fuh = fuh; p.fuw = fuw; p.fdh = fdh; p.fdw = fdw; p.px0 = px0; p.py0 = py0;
    p.MH = (FLR_TH - 1) * down + fdh; p.MW = (FLR_TW - 1) * down + fdw;
    p.IH = (p.MH + fuh - 2) / up + 2; p.IW = (p.MW + fuw - 2) / up + 2;      // span of floor((m + t - pad) / up) over the tile, any alignment
    p.tiles_x = cdiv(p.OW, FLR_TW); p.tiles_y = cdiv(p.OH, FLR_TH);
    p.gain = gain; p.slope = slope; p.clamp = clamp; p.flip = flip;
    const size_t lds = sizeof(float) * ((size_t)p.IH * p.IW + (size_t)p.MH * p.MW + (size_t)fuh * fuw + (size_t)fdh * fdw);
    N3D_CHECK(lds <= 64 * 1024, "filtered_lrelu: filters / factors too large for one LDS tile (%zu bytes)", lds);
    const double taps = (double)((fuh + up - 1) / up) * ((fuw + up - 1) / up);
    N3dProfScope prof(N3D_K_UPFIRDN2D, stream, 2.0 * N * C * ((double)mid_h * mid_w * taps + (double)p.OH * p.OW * fdh * fdw),
                      4.0 * N * C * ((double)H * W + (double)p.OH * p.OW));
    hipLaunchKernelGGL(filtered_lrelu_kernel, dim3((unsigned)((int64_t)p.tiles_x * p.tiles_y * N * C)), dim3(FLR_NT), lds, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}
