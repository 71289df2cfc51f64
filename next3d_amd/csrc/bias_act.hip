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
    if (dtype == N3D_F32)
        hipLaunchKernelGGL(bias_act_kernel<float>, dim3(grid), dim3(block), 0, stream, (const float*)x, (const float*)b,
                           (float*)y, numel, size_b, step_b, act, alpha, gain, clamp);
    else
        hipLaunchKernelGGL(bias_act_kernel<__half>, dim3(grid), dim3(block), 0, stream, (const __half*)x, (const __half*)b,
                           (__half*)y, numel, size_b, step_b, act, alpha, gain, clamp);
    N3D_LAUNCH_CHECK();
    return 0;
}
