// bias_act for gfx950: y = clamp(act(x + b[(i / step_b) % size_b]) * gain).
// HBM-bound elementwise op: 16-byte loads/stores, grid-stride, 64-wide waves.
// Replaces bias_act_plugin.bias_act (reference torch_utils/ops/bias_act.cpp:36, bias_act.cu:27-150), forward only.
#include <hip/hip_fp16.h>

#include "common.h"

template <typename T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<__half> { using type = uint2; };

__device__ __forceinline__ float ld(const float* p) { return *p; }
__device__ __forceinline__ float ld(const __half* p) { return __half2float(*p); }
__device__ __forceinline__ void st(float* p, float v) { *p = v; }
__device__ __forceinline__ void st(__half* p, float v) { *p = __float2half(v); }

template <typename T>
__global__ __launch_bounds__(256) void bias_act_kernel(const T* __restrict__ x, const T* __restrict__ b, T* __restrict__ y,
                                                       int64_t numel, int size_b, int64_t step_b, int act, float alpha,
                                                       float gain, float clamp) {
    const int64_t nvec = numel >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    using V = typename Vec4<T>::type;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        V in = reinterpret_cast<const V*>(x)[v];
        T* e = reinterpret_cast<T*>(&in);
        V out;
        T* oe = reinterpret_cast<T*>(&out);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = v * 4 + k;
            float f = ld(&e[k]);
            if (b) f += ld(&b[(i / step_b) % size_b]);
            f = n3d_act(f, act, alpha) * gain;
            if (clamp >= 0.f) f = fminf(fmaxf(f, -clamp), clamp);
            st(&oe[k], f);
        }
        reinterpret_cast<V*>(y)[v] = out;
    }
    // tail (numel % 4)
    for (int64_t i = (nvec << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
        float f = ld(&x[i]);
        if (b) f += ld(&b[(i / step_b) % size_b]);
        f = n3d_act(f, act, alpha) * gain;
        if (clamp >= 0.f) f = fminf(fmaxf(f, -clamp), clamp);
        st(&y[i], f);
    }
}

// The common geometry — the bias runs along a dimension whose inner extent (step_b) is a multiple of 4, e.g. dim 1 of an NCHW tensor —
// as a 2-D grid: blockIdx.y = row (one bias value per row), blockIdx.x walks the row's 16-byte vectors.  No 64-bit divisions per
// element (the generic kernel above pays four per vector) and the activation is a template parameter: the operator boundary issues
// one bias_act per convolution layer (103 per generator forward), most of them linear / leaky-ReLU on large feature maps.
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bias_act_rows_kernel(const T* __restrict__ x, const T* __restrict__ b, T* __restrict__ y, int64_t row_vecs,
                                                            int size_b, int act, float alpha, float gain, float clamp) {
    using V = typename Vec4<T>::type;
    const int64_t row = blockIdx.y;
    const float bias = b ? ld(&b[row % size_b]) : 0.f;
    const V* xr = reinterpret_cast<const V*>(x) + row * row_vecs;
    V* yr = reinterpret_cast<V*>(y) + row * row_vecs;
    const bool clamped = clamp >= 0.f;          // (wave-uniform; without a clamp a NaN activation must stay NaN, as in bias_act_kernel and the reference: fminf / fmaxf drop NaNs)
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < row_vecs; v += (int64_t)gridDim.x * 256) {
        V in = xr[v];
        T* e = reinterpret_cast<T*>(&in);
        V out;
        T* oe = reinterpret_cast<T*>(&out);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f = ld(&e[k]) + bias;
            if (ACT == N3D_ACT_LRELU) f = (f > 0.f ? f : f * alpha);
            else if (ACT != N3D_ACT_LINEAR) f = n3d_act(f, act, alpha);
            f *= gain;
            if (clamped) f = fminf(fmaxf(f, -clamp), clamp);
            st(&oe[k], f);
        }
        yr[v] = out;
    }
}

template <typename T>
static void bias_act_rows_launch(const T* x, const T* b, T* y, int64_t rows, int64_t row_vecs, int size_b, int act, float alpha, float gain, float clamp,
                                 hipStream_t stream) {
    int gx = (int)(cdiv64(row_vecs, 256 * 4) < 1 ? 1 : cdiv64(row_vecs, 256 * 4));      // ~4 vectors per thread
    if (gx > 1024) gx = 1024;
    const dim3 grid(gx, (unsigned)rows);
    if (act == N3D_ACT_LINEAR) hipLaunchKernelGGL((bias_act_rows_kernel<T, N3D_ACT_LINEAR>), grid, dim3(256), 0, stream, x, b, y, row_vecs, size_b, act, alpha, gain, clamp);
    else if (act == N3D_ACT_LRELU) hipLaunchKernelGGL((bias_act_rows_kernel<T, N3D_ACT_LRELU>), grid, dim3(256), 0, stream, x, b, y, row_vecs, size_b, act, alpha, gain, clamp);
    else hipLaunchKernelGGL((bias_act_rows_kernel<T, 0>), grid, dim3(256), 0, stream, x, b, y, row_vecs, size_b, act, alpha, gain, clamp);
}

extern "C" int n3d_bias_act(const void* x, const void* b, void* y, int64_t numel, int size_b, int64_t step_b, int dtype,
                            int act, float alpha, float gain, float clamp, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(numel >= 0, "bias_act: negative numel");
    N3D_CHECK(act >= N3D_ACT_LINEAR && act <= N3D_ACT_SWISH, "bias_act: unknown activation %d", act);
    N3D_CHECK(dtype == N3D_F32 || dtype == N3D_F16, "bias_act: unsupported dtype %d", dtype);
    N3D_CHECK(b == nullptr || (size_b > 0 && step_b > 0), "bias_act: bad bias geometry");
    if (numel == 0) return 0;
    N3D_CHECK(x && y, "bias_act: null tensor");
    const int esize = dtype == N3D_F32 ? 4 : 2;
    N3D_CHECK(((uintptr_t)x % (4 * esize)) == 0 && ((uintptr_t)y % (4 * esize)) == 0, "bias_act: x/y must be 4-element aligned");
    if (!b) { size_b = 1; step_b = 1; }
    const int block = 256;
    int64_t want = cdiv64(cdiv64(numel, 4), block);
    const int grid = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    N3dProfScope prof(N3D_K_BIAS_ACT, stream, (double)numel, 2.0 * esize * (double)numel);
    if (!b && numel % 4 == 0) step_b = numel;                      // no bias: one "row"
    if (step_b % 4 == 0 && step_b >= 64 && numel % step_b == 0 && numel / step_b <= 65535) {       // rows of step_b elements, one bias value each
        const int64_t rows = numel / step_b;
        if (dtype == N3D_F32) bias_act_rows_launch<float>((const float*)x, (const float*)b, (float*)y, rows, step_b / 4, size_b, act, alpha, gain, clamp, stream);
        else bias_act_rows_launch<__half>((const __half*)x, (const __half*)b, (__half*)y, rows, step_b / 4, size_b, act, alpha, gain, clamp, stream);
        N3D_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == N3D_F32)
        hipLaunchKernelGGL(bias_act_kernel<float>, dim3(grid), dim3(block), 0, stream, (const float*)x, (const float*)b,
                           (float*)y, numel, size_b, step_b, act, alpha, gain, clamp);
    else
        hipLaunchKernelGGL(bias_act_kernel<__half>, dim3(grid), dim3(block), 0, stream, (const __half*)x, (const __half*)b,
                           (__half*)y, numel, size_b, step_b, act, alpha, gain, clamp);
    N3D_LAUNCH_CHECK();
    return 0;
}
