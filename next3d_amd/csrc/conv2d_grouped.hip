// Per-sample ("grouped") weight preparation for the OPERATOR boundary (B1).
//
// The reference's fused modulated_conv2d (training_avatar_texture/networks_stylegan2.py:82-88) folds the batch into the channel
// dimension — x.reshape(1, N*I, H, W), w.reshape(N*O, I, k, k) — and calls conv2d_resample(..., groups = N), which ends in ATen's
// grouped F.conv2d / F.conv_transpose2d (torch_utils/ops/conv2d_resample.py:96-136, conv2d_gradfix.py:37-45).  Behind the same
// call, next3d_amd/torch_utils/ops/conv2d_gradfix.py runs ONE launch for the whole batch on the matrix-core kernels with
// n3d_conv2d_desc.wt_batch_stride; those kernels stream K-major tiles, so the [G*O, I, k, k] (or, transposed, [G*I, O, k, k])
// tensor the reference built is re-tiled here, all groups in one launch: read once, written once — HBM-bound, 2 x the weight bytes
// (a 512 -> 512 3x3 layer at batch 4: 37.7 MB each way).
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct GroupedPrepParams {
    const void* w; void* wt;
    int w_f16, kind, G, O, I, KK, OP;      // OP: O padded to 4 (kind 0) / 64 (kind 1) / O (kind 2)
    int tiles_o, tiles_i;
    int64_t sg, so, si;
};

// One workgroup re-tiles a (32 output channels) x (32 input channels) x (all taps) block of one group through LDS:
//   read : the source walks its FASTEST dimensions across the lanes — [.., I, k, k] rows of 32 x kk contiguous elements per output
//          channel (F.conv2d layout, stride_i == kk) or [.., O, k, k] rows per input channel (F.conv_transpose2d layout, stride_o == kk);
//   write: 16-byte units with the output channel fastest — 512-byte runs of the tiled layouts, 128-byte runs of the float32 one.
constexpr int GP_T = 32;
__global__ __launch_bounds__(256) void grouped_prep_kernel(GroupedPrepParams p) {
    __shared__ float tile[GP_T * (GP_T * 9 + 1)];
    const int KK = p.KK, S = GP_T * KK + 1;                               // + 1: the write phase's lanes walk o — one bank apart
    int b = blockIdx.x;
    const int ti = b % p.tiles_i; b /= p.tiles_i;
    const int to = b % p.tiles_o;
    const int g = b / p.tiles_o;
    const int o0 = to * GP_T, i0 = ti * GP_T;
    const bool o_major = p.so >= p.si;                                    // source rows: per output channel (conv) or per input channel (transposed)
    for (int e = threadIdx.x; e < GP_T * GP_T * KK; e += 256) {
        const int t = e % KK, r = e / KK;
        const int ol = o_major ? r / GP_T : r % GP_T, il = o_major ? r % GP_T : r / GP_T;
        const int o = o0 + ol, i = i0 + il;
        float v = 0.f;
        if (o < p.O && i < p.I) {
            const int64_t src = (int64_t)g * p.sg + (int64_t)o * p.so + (int64_t)i * p.si + t;
            v = p.w_f16 ? (float)reinterpret_cast<const _Float16*>(p.w)[src] : reinterpret_cast<const float*>(p.w)[src];
        }
        tile[ol * S + il * KK + t] = v;
    }
    __syncthreads();
    if (p.kind == 0) {                                                    // float32 K-major [G][KK][I][OP]: float4 over 4 output channels
        for (int e = threadIdx.x; e < KK * GP_T * (GP_T / 4); e += 256) {
            const int o4 = e % (GP_T / 4), il = (e / (GP_T / 4)) % GP_T, t = e / (GP_T * (GP_T / 4));
            const int o = o0 + 4 * o4, i = i0 + il;
            if (o >= p.OP || i >= p.I) continue;
            const float* s = tile + (4 * o4) * S + il * KK + t;
            reinterpret_cast<float4*>(reinterpret_cast<float*>(p.wt) + (((int64_t)g * KK + t) * p.I + i) * p.OP + o)[0] = make_float4(s[0], s[S], s[2 * S], s[3 * S]);
        }
        return;
    }
    // kind 1: bf16 [G][KK][I/16][hl 2][half 2][OP64][8] (conv2d_bf16x3.hip: conv16_prep_weight_kernel's layout per group)
    // kind 2: f16  [G][KK][I/16][half 2][O][8]          (sr_f16.hip: modulate_row's layout per group)
    for (int e = threadIdx.x; e < KK * (GP_T / 8) * GP_T; e += 256) {
        const int ol = e % GP_T, c8l = (e / GP_T) % (GP_T / 8), t = e / (GP_T * (GP_T / 8));
        const int o = o0 + ol, c8 = i0 / 8 + c8l;
        if (o >= p.OP || c8 * 8 >= p.I) continue;
        const float* s = tile + ol * S + (c8l * 8) * KK + t;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = s[k * KK];
        const int kc = c8 >> 1, half = c8 & 1;
        if (p.kind == 1) {
            bf16x8 hi, lo;
#pragma unroll
            for (int k = 0; k < 8; ++k) { const __bf16 h = (__bf16)v[k]; hi[k] = h; lo[k] = (__bf16)(v[k] - (float)h); }
            bf16x8* dst = reinterpret_cast<bf16x8*>(p.wt) + ((((int64_t)g * KK + t) * (p.I / 16) + kc) * 4 + half) * p.OP + o;
            dst[0] = hi;
            dst[(int64_t)2 * p.OP] = lo;                                   // hl = 1
        } else {
            f16x8 h;
#pragma unroll
            for (int k = 0; k < 8; ++k) h[k] = (_Float16)v[k];
            reinterpret_cast<f16x8*>(p.wt)[((((int64_t)g * KK + t) * (p.I / 16) + kc) * 2 + half) * p.OP + o] = h;
        }
    }
}

extern "C" int n3d_conv2d_prep_weight_grouped(const void* w, int w_dtype, void* wt, int wt_kind, int G, int O, int I, int ksize,
                                              int64_t stride_g, int64_t stride_o, int64_t stride_i, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(w_dtype == N3D_F32 || w_dtype == N3D_F16, "conv2d_prep_weight_grouped: float32 or float16 weights");
    N3D_CHECK(wt_kind >= 0 && wt_kind <= 2, "conv2d_prep_weight_grouped: wt_kind 0 (float32 K-major), 1 (split-bf16 tiles) or 2 (float16 tiles)");
    N3D_CHECK(G >= 0 && O > 0 && I > 0 && (ksize == 1 || ksize == 3), "conv2d_prep_weight_grouped: bad shape (1x1 or 3x3)");
    N3D_CHECK(wt_kind == 0 || I % 16 == 0, "conv2d_prep_weight_grouped: the tiled layouts need I %% 16 == 0");
    N3D_CHECK(wt_kind != 2 || ksize == 3, "conv2d_prep_weight_grouped: float16 tiles are 3x3 (n3d_conv2d_f16)");
    N3D_CHECK(stride_g >= 0 && stride_o > 0 && stride_i > 0, "conv2d_prep_weight_grouped: bad strides");
    if (G == 0) return 0;
    N3D_CHECK(w && wt && ((uintptr_t)wt & 15) == 0, "conv2d_prep_weight_grouped: null or misaligned tensor");
    GroupedPrepParams p;
    p.w = w; p.wt = wt; p.w_f16 = w_dtype == N3D_F16; p.kind = wt_kind; p.G = G; p.O = O; p.I = I; p.KK = ksize * ksize;
    p.OP = wt_kind == 0 ? (O + 3) / 4 * 4 : (wt_kind == 1 ? (O + 63) / 64 * 64 : O);
    p.sg = stride_g; p.so = stride_o; p.si = stride_i;
    p.tiles_o = cdiv(p.OP, GP_T); p.tiles_i = cdiv(I, GP_T);
    const double elems = (double)G * O * I * p.KK;
    N3dProfScope prof(N3D_K_MISC, stream, 0.0, elems * ((w_dtype == N3D_F16 ? 2.0 : 4.0) + (wt_kind == 2 ? 2.0 : 4.0)));
    const int64_t nblk = (int64_t)G * p.tiles_o * p.tiles_i;
    N3D_CHECK(nblk < (1ll << 31), "conv2d_prep_weight_grouped: grid too large");
    hipLaunchKernelGGL(grouped_prep_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}
