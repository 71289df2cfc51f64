// Small element-wise kernels of the generator forward path (HBM-bound or launch-bound; 16-byte accesses where aligned).
//   normalize_2nd_moment : x * rsqrt(mean(x^2, dim=1) + eps)            (training_avatar_texture/networks_stylegan2.py:27-29)
//   truncate_ws          : w_avg + psi * (ws - w_avg) on the first `cutoff` latents, broadcast of one w to num_ws
//                          (MappingNetwork.forward :255-267)
//   fma                  : a * b + c with NCHW / per-(n,c) / per-pixel broadcasting     (torch_utils/ops/fma.py:17-28)
//   to_uint8             : (img * 127.5 + 128).clamp(0, 255) -> uint8              (gen_samples_next3d.py:201)
//   cast                 : float16 <-> float32 (the reference's `x.to(dtype)` at the fp16 block boundaries,
//                          training_avatar_texture/networks_stylegan2.py:548-552, :57-59)
#include "common.h"

// one wave per row.  T = double: the scripts' z (torch.from_numpy(np.random.RandomState(seed).randn(...)): float64) — every element is rounded to
// float32 first, as MappingNetwork.forward does (`z.to(torch.float32)`, tat/networks_stylegan2.py:239), then the float32 arithmetic: bit-identical to
// a separate conversion pass
template <typename T>
__global__ __launch_bounds__(256) void normalize_2nd_moment_kernel(const T* __restrict__ x, float* __restrict__ y, int rows, int D,
                                                                   int64_t y_stride, float eps) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const T* xr = x + (int64_t)r * D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) { const float v = (float)xr[i]; s += v * v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float k = rsqrtf(s / (float)D + eps);
    float* yr = y + (int64_t)r * y_stride;
    for (int i = lane; i < D; i += 64) yr[i] = (float)xr[i] * k;
}

// ws[n, j, :] = j < cutoff ? lerp(w_avg, w[n], psi) : w[n]
__global__ __launch_bounds__(256) void truncate_ws_kernel(const float* __restrict__ w, const float* __restrict__ w_avg, float* __restrict__ ws,
                                                          int N, int num_ws, int D, int cutoff, float psi) {
    const int64_t total = (int64_t)N * num_ws * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D), j = (int)((i / D) % num_ws), n = (int)(i / ((int64_t)D * num_ws));
        const float v = w[(int64_t)n * D + d];
        ws[i] = (j < cutoff && w_avg) ? w_avg[d] + psi * (v - w_avg[d]) : v;
    }
}

// y[n,c,p] = a[n,c,p] * b[(n*C + c) * bs_nc + p * bs_p] + c[...] ; strides 0 broadcast
__global__ __launch_bounds__(256) void fma_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                  float* __restrict__ y, int64_t NC, int64_t P, int64_t b_nc, int64_t b_p, int64_t c_nc,
                                                  int64_t c_p) {
    const int64_t total = NC * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t nc = i / P, pp = i % P;
        y[i] = a[i] * b[nc * b_nc + pp * b_p] + c[nc * c_nc + pp * c_p];
    }
}

__global__ __launch_bounds__(256) void to_uint8_kernel(const float* __restrict__ x, unsigned char* __restrict__ y, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        uchar4 o;
        o.x = (unsigned char)fminf(fmaxf(v.x * 127.5f + 128.f, 0.f), 255.f);
        o.y = (unsigned char)fminf(fmaxf(v.y * 127.5f + 128.f, 0.f), 255.f);
        o.z = (unsigned char)fminf(fmaxf(v.z * 127.5f + 128.f, 0.f), 255.f);
        o.w = (unsigned char)fminf(fmaxf(v.w * 127.5f + 128.f, 0.f), 255.f);
        reinterpret_cast<uchar4*>(y)[i] = o;
    }
}

// The video scripts' frame grid in ONE pass (gen_videos_next3d.py:35-49 does conversion, reshape, two permutes and a copy):
// frames [B,C,H,W] float32 -> uint8 canvas, frame b at tile row b / cols, tile column b % cols.  One thread per 4 consecutive x of one
// (frame, y): a 16-byte load per channel; hwc = 1 writes [rows*H, cols*W, C] (4 pixels x C bytes contiguous per thread), hwc = 0 writes [C, rows*H, cols*W].
__global__ __launch_bounds__(256) void layout_grid_u8_kernel(const float* __restrict__ x, unsigned char* __restrict__ y, int B, int C, int H, int W4,
                                                             int cols, int rows, int hwc) {
    const int64_t total = (int64_t)B * H * W4;
    const int W = W4 * 4, CW = cols * W;
    const int64_t CH = (int64_t)rows * H;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int xq = (int)(i % W4), yy = (int)((i / W4) % H), b = (int)(i / ((int64_t)W4 * H));
        const int64_t oy = (int64_t)(b / cols) * H + yy, ox = (int64_t)(b % cols) * W + xq * 4;
        const float* src = x + ((int64_t)b * C * H + yy) * W + xq * 4;
        for (int c = 0; c < C; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)c * H * W);
            const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned char u = (unsigned char)fminf(fmaxf(f[k] * 127.5f + 128.f, 0.f), 255.f);
                if (hwc) y[(oy * CW + ox + k) * C + c] = u;
                else y[((int64_t)c * CH + oy) * CW + ox + k] = u;
            }
        }
    }
}

// 8 elements per thread (16-byte accesses on the half side, 2 x 16 on the float side); the tail is done element-wise
template <typename SRC, typename DST>
__global__ __launch_bounds__(256) void cast_kernel(const SRC* __restrict__ x, DST* __restrict__ y, int64_t n) {
    const int64_t n8 = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        SRC v[8];
        DST o[8];
        __builtin_memcpy(v, x + i * 8, sizeof(v));
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (DST)v[k];          // f32 -> f16: round to nearest even (v_cvt_f16_f32), as ATen
        __builtin_memcpy(y + i * 8, o, sizeof(o));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) y[n8 * 8 + threadIdx.x] = (DST)x[n8 * 8 + threadIdx.x];
}

// float32 NCHW -> split8 (include/n3d.h): y[n][hl][c/8][p][c%8] = split(x[n][c][p] * scale[n][c]).  One work item = one pixel x
// 8 channels: 8 plane reads (each coalesced across the wave), one 16-byte unit per plane (hi, lo) written (contiguous across the
// wave).  HBM-bound: 4 B in + 4 B out per element.
typedef __bf16 ew_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void split8_from_nchw_kernel(const float* __restrict__ x, const float* __restrict__ scale, ew_bf16x8* __restrict__ y,
                                                               int C8, int64_t HW, int64_t xbs, int64_t scale_stride) {
    const int c8 = blockIdx.y, n = blockIdx.z;
    const float* xp = x + (int64_t)n * xbs + (int64_t)c8 * 8 * HW;
    float sc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) sc[c] = scale ? scale[(int64_t)n * scale_stride + c8 * 8 + c] : 1.f;
    ew_bf16x8* yh = y + (((int64_t)n * 2 + 0) * C8 + c8) * HW;
    ew_bf16x8* yl = y + (((int64_t)n * 2 + 1) * C8 + c8) * HW;
    for (int64_t px = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; px < HW; px += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = xp[(int64_t)c * HW + px];
        ew_bf16x8 hi, lo;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float t = v[c] * sc[c];
            const __bf16 h = (__bf16)t;
            hi[c] = h;
            lo[c] = (__bf16)(t - (float)h);
        }
        yh[px] = hi;
        yl[px] = lo;
    }
}

static inline int grid_for(int64_t n) { const int64_t g = cdiv64(n, 256); return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g)); }

extern "C" {

int n3d_normalize_2nd_moment(const float* x, float* y, int rows, int D, int64_t y_stride, float eps, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(rows >= 0 && D > 0 && y_stride >= D, "normalize_2nd_moment: bad shape");
    if (rows == 0) return 0;
    N3D_CHECK(x && y, "normalize_2nd_moment: null tensor");
    N3dProfScope prof(N3D_K_MISC, stream, 3.0 * rows * D, 8.0 * rows * D);
    hipLaunchKernelGGL(normalize_2nd_moment_kernel<float>, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, y, rows, D, y_stride, eps);
    N3D_LAUNCH_CHECK();
    return 0;
}

int n3d_normalize_2nd_moment_f64(const double* x, float* y, int rows, int D, int64_t y_stride, float eps, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(rows >= 0 && D > 0 && y_stride >= D, "normalize_2nd_moment_f64: bad shape");
    if (rows == 0) return 0;
    N3D_CHECK(x && y, "normalize_2nd_moment_f64: null tensor");
    N3dProfScope prof(N3D_K_MISC, stream, 3.0 * rows * D, 12.0 * rows * D);
    hipLaunchKernelGGL(normalize_2nd_moment_kernel<double>, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, y, rows, D, y_stride, eps);
    N3D_LAUNCH_CHECK();
    return 0;
}

int n3d_truncate_ws(const float* w, const float* w_avg, float* ws, int N, int num_ws, int D, int cutoff, float psi, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && num_ws > 0 && D > 0, "truncate_ws: bad shape");
    if (N == 0) return 0;
    N3D_CHECK(w && ws, "truncate_ws: null tensor");
    N3dProfScope prof(N3D_K_MISC, stream, 2.0 * N * num_ws * D, 4.0 * N * D * (num_ws + 1));
    hipLaunchKernelGGL(truncate_ws_kernel, dim3(grid_for((int64_t)N * num_ws * D)), dim3(256), 0, stream, w, w_avg, ws, N, num_ws, D, cutoff, psi);
    N3D_LAUNCH_CHECK();
    return 0;
}

int n3d_fma(const float* a, const float* b, const float* c, float* y, int64_t NC, int64_t P, int64_t b_nc, int64_t b_p, int64_t c_nc,
            int64_t c_p, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(NC >= 0 && P >= 0, "fma: bad shape");
    if (NC * P == 0) return 0;
    N3D_CHECK(a && b && c && y, "fma: null tensor");
    N3dProfScope prof(N3D_K_MISC, stream, 2.0 * NC * P, 8.0 * NC * P);
    hipLaunchKernelGGL(fma_kernel, dim3(grid_for(NC * P)), dim3(256), 0, stream, a, b, c, y, NC, P, b_nc, b_p, c_nc, c_p);
    N3D_LAUNCH_CHECK();
    return 0;
}

int n3d_to_uint8(const float* x, unsigned char* y, int64_t numel, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(numel >= 0 && numel % 4 == 0, "to_uint8: numel must be a multiple of 4");
    if (numel == 0) return 0;
    N3D_CHECK(x && y && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 3) == 0, "to_uint8: null or misaligned tensor");
    N3dProfScope prof(N3D_K_MISC, stream, 2.0 * numel, 5.0 * numel);
    hipLaunchKernelGGL(to_uint8_kernel, dim3(grid_for(numel / 4)), dim3(256), 0, stream, x, y, numel / 4);
    N3D_LAUNCH_CHECK();
    return 0;
}

// synthesis' input hand-over in ONE launch: v [N][V + L][3] (vertices then landmarks, any batch stride) -> verts [N][V][3], lms [N][L][3];
// c [N][25] -> cam2world [N][16], intrinsics [N][9] (the dense tensors n3d_rasterize_views / n3d_render_rays read) — four torch copies before
__global__ __launch_bounds__(256) void unpack_inputs_kernel(const float* __restrict__ v, int64_t v_bs, const float* __restrict__ c, int64_t c_bs,
                                                            float* __restrict__ verts, float* __restrict__ lms, float* __restrict__ cam, float* __restrict__ intr,
                                                            int N, int V, int L) {
    const int per = (V + L) * 3 + 25;
    const int64_t total = (int64_t)N * per;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i / per), e = (int)(i % per);
        if (e < V * 3) verts[(int64_t)n * V * 3 + e] = v[n * v_bs + e];
        else if (e < (V + L) * 3) lms[(int64_t)n * L * 3 + (e - V * 3)] = v[n * v_bs + e];
        else if (e < (V + L) * 3 + 16) cam[n * 16 + (e - (V + L) * 3)] = c[n * c_bs + (e - (V + L) * 3)];
        else intr[n * 9 + (e - (V + L) * 3 - 16)] = c[n * c_bs + (e - (V + L) * 3)];
    }
}

int n3d_unpack_inputs(const float* v, int64_t v_batch_stride, const float* c, int64_t c_batch_stride, float* verts, float* lms, float* cam2world,
                      float* intrinsics, int N, int V, int L, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && V >= 1 && L >= 0, "unpack_inputs: bad shape");
    if (N == 0) return 0;
    N3D_CHECK(v && c && verts && (lms || L == 0) && cam2world && intrinsics, "unpack_inputs: null tensor");
    const int64_t total = (int64_t)N * ((V + L) * 3 + 25);
    N3dProfScope prof(N3D_K_MISC, stream, 0.0, 8.0 * total);
    hipLaunchKernelGGL(unpack_inputs_kernel, dim3(grid_for(total)), dim3(256), 0, stream, v, v_batch_stride, c, c_batch_stride, verts, lms, cam2world, intrinsics, N, V, L);
    N3D_LAUNCH_CHECK();
    return 0;
}

int n3d_layout_grid_u8(const float* frames, unsigned char* canvas, int B, int C, int H, int W, int cols, int rows, int hwc, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(B >= 0 && C >= 1 && H >= 1 && W >= 4 && W % 4 == 0 && cols >= 1 && rows >= 1, "layout_grid_u8: bad shape (W %% 4 == 0)");
    N3D_CHECK((int64_t)cols * rows == B, "layout_grid_u8: %d frames do not fill a %d x %d grid", B, cols, rows);
    if (B == 0) return 0;
    N3D_CHECK(frames && canvas && ((uintptr_t)frames & 15) == 0, "layout_grid_u8: null or misaligned tensor");
    const int64_t numel = (int64_t)B * C * H * W;
    N3dProfScope prof(N3D_K_MISC, stream, 2.0 * numel, 5.0 * numel);
    hipLaunchKernelGGL(layout_grid_u8_kernel, dim3(grid_for(numel / (4 * C))), dim3(256), 0, stream, frames, canvas, B, C, H, W / 4, cols, rows, hwc);
    N3D_LAUNCH_CHECK();
    return 0;
}

int n3d_split8_from_nchw(const float* x, const float* scale, void* y, int N, int C, int64_t HW, int64_t x_batch_stride, int64_t scale_stride,
                         n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && C > 0 && C % 8 == 0 && HW > 0, "split8_from_nchw: bad shape (C %% 8 == 0)");
    if (N == 0) return 0;
    N3D_CHECK(x && y && ((uintptr_t)y & 15) == 0, "split8_from_nchw: null or misaligned tensor");
    N3D_CHECK(C / 8 <= 65535 && N <= 65535, "split8_from_nchw: N and C/8 must be <= 65535");
    N3dProfScope prof(N3D_K_MISC, stream, 2.0 * N * C * (double)HW, 8.0 * N * C * (double)HW);
    const int gx = (int)(cdiv64(HW, 256) > 1024 ? 1024 : cdiv64(HW, 256));
    hipLaunchKernelGGL(split8_from_nchw_kernel, dim3(gx, C / 8, N), dim3(256), 0, stream, x, scale, (ew_bf16x8*)y, C / 8, HW,
                       x_batch_stride ? x_batch_stride : (int64_t)C * HW, scale_stride ? scale_stride : C);
    N3D_LAUNCH_CHECK();
    return 0;
}

int n3d_cast(const void* x, void* y, int64_t numel, int src_dtype, int dst_dtype, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(numel >= 0, "cast: bad size");
    N3D_CHECK((src_dtype == N3D_F32 && dst_dtype == N3D_F16) || (src_dtype == N3D_F16 && dst_dtype == N3D_F32),
              "cast: float32 <-> float16 only");
    if (numel == 0) return 0;
    N3D_CHECK(x && y && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "cast: null or misaligned tensor");
    N3dProfScope prof(N3D_K_MISC, stream, 0.0, 6.0 * numel);
    const int grid = grid_for(cdiv64(numel, 8));
    if (src_dtype == N3D_F32) hipLaunchKernelGGL((cast_kernel<float, _Float16>), dim3(grid), dim3(256), 0, stream, (const float*)x, (_Float16*)y, numel);
    else hipLaunchKernelGGL((cast_kernel<_Float16, float>), dim3(grid), dim3(256), 0, stream, (const _Float16*)x, (float*)y, numel);
    N3D_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
