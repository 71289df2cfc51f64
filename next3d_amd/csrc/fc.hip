// Small fully-connected layers (style affines, mapping network, demodulation coefficients) for gfx950.
// These are GEMVs with N <= a few samples: weight-read bound.  One 64-lane wave owns one output
// feature, streams its weight row once with 16-byte loads and keeps up to 8 sample accumulators in
// registers; cross-lane reduction with DPP shuffles.  No LDS needed.
// Replaces FullyConnectedLayer.forward (reference training_avatar_texture/networks_stylegan2.py:114-127)
// and the demodulation reduction of modulated_conv2d (:72-76) in the factored form
//   d[n,o] = rsqrt(sum_i s[n,i]^2 * (sum_k w[o,i,k]^2) + 1e-8).
#include "common.h"

#define FC_MAX_N 8

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool PRE_SQUARE>
__global__ __launch_bounds__(256) void fc_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                 const float* __restrict__ b, float* __restrict__ y, int n0, int nn, int N,
                                                 int I, int O, float wgain, float bgain, int act, float alpha, float gain,
                                                 int post_rsqrt) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= O) return;
    const float* wr = w + (int64_t)o * I;
    float acc[FC_MAX_N];
#pragma unroll
    for (int n = 0; n < FC_MAX_N; ++n) acc[n] = 0.f;
    if ((I & 3) == 0) {
        for (int i = lane * 4; i < I; i += 256) {
            const float4 wv = *reinterpret_cast<const float4*>(wr + i);
#pragma unroll
            for (int n = 0; n < FC_MAX_N; ++n) {
                if (n < nn) {
                    float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)(n0 + n) * I + i);
                    if (PRE_SQUARE) { xv.x *= xv.x; xv.y *= xv.y; xv.z *= xv.z; xv.w *= xv.w; }
                    acc[n] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
                }
            }
        }
    } else {
        for (int i = lane; i < I; i += 64) {
            const float wv = wr[i];
#pragma unroll
            for (int n = 0; n < FC_MAX_N; ++n) {
                if (n < nn) {
                    float xv = x[(int64_t)(n0 + n) * I + i];
                    if (PRE_SQUARE) xv *= xv;
                    acc[n] += xv * wv;
                }
            }
        }
    }
    const float bias = b ? b[o] * bgain : 0.f;
#pragma unroll
    for (int n = 0; n < FC_MAX_N; ++n) {
        if (n < nn) {
            float v = wave_sum(acc[n]);
            v = v * wgain + bias;
            v = n3d_act(v, act, alpha) * gain;
            if (post_rsqrt) v = rsqrtf(v + 1e-8f);
            if (lane == 0) y[(int64_t)(n0 + n) * O + o] = v;
        }
    }
}

// one wave per (job, output row); up to FC_MAX_N samples per pass
__global__ __launch_bounds__(256) void fc_multi_kernel(const n3d_fc_job* __restrict__ jobs, const int* __restrict__ rows, int total_rows,
                                                       const float* __restrict__ x_base, float* __restrict__ y_base, int n0, int nn) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= total_rows) return;
    const n3d_fc_job J = jobs[rows[2 * r]];
    const int o = rows[2 * r + 1];
    const float* wr = J.w + (int64_t)o * J.I;
    const float* xb = x_base + J.x_off + (int64_t)n0 * J.x_stride;
    float acc[FC_MAX_N];
#pragma unroll
    for (int n = 0; n < FC_MAX_N; ++n) acc[n] = 0.f;
    for (int i = lane * 4; i < J.I; i += 256) {          // I % 4 == 0 checked on the host
        const float4 wv = *reinterpret_cast<const float4*>(wr + i);
#pragma unroll
        for (int n = 0; n < FC_MAX_N; ++n) {
            if (n < nn) {
                float4 xv = *reinterpret_cast<const float4*>(xb + (int64_t)n * J.x_stride + i);
                if (J.pre_square) { xv.x *= xv.x; xv.y *= xv.y; xv.z *= xv.z; xv.w *= xv.w; }
                acc[n] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
            }
        }
    }
    const float bias = J.b ? J.b[o] * J.bgain : 0.f;
#pragma unroll
    for (int n = 0; n < FC_MAX_N; ++n) {
        if (n < nn) {
            float v = wave_sum(acc[n]);
            v = v * J.wgain + bias;
            v = n3d_act(v, J.act, J.alpha) * J.gain;
            if (J.post_rsqrt) v = rsqrtf(v + 1e-8f);
            if (lane == 0) y_base[J.y_off + (int64_t)(n0 + n) * J.y_stride + o] = v;
        }
    }
}

extern "C" int n3d_fc_multi(const n3d_fc_job* jobs, const int* rows, int total_rows, const float* x_base, float* y_base, int N,
                            n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(total_rows >= 0 && N >= 0, "fc_multi: bad sizes");
    if (total_rows == 0 || N == 0) return 0;
    N3D_CHECK(jobs && rows && x_base && y_base, "fc_multi: null tensor");
    N3dProfScope prof(N3D_K_FC, stream, 0.0, 0.0);
    for (int n0 = 0; n0 < N; n0 += FC_MAX_N) {
        const int nn = (N - n0) < FC_MAX_N ? (N - n0) : FC_MAX_N;
        hipLaunchKernelGGL(fc_multi_kernel, dim3(cdiv(total_rows, 4)), dim3(256), 0, stream, jobs, rows, total_rows, x_base, y_base, n0, nn);
        N3D_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int n3d_fc(const float* x, const float* w, const float* b, float* y, int N, int I, int O, float wgain,
                      float bgain, int act, float alpha, float gain, int pre_square, int post_rsqrt, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && I > 0 && O > 0, "fc: bad sizes N=%d I=%d O=%d", N, I, O);
    N3D_CHECK(act >= N3D_ACT_LINEAR && act <= N3D_ACT_SWISH, "fc: unknown activation %d", act);
    if (N == 0) return 0;
    N3D_CHECK(x && w && y, "fc: null tensor");
    N3D_CHECK((I & 3) != 0 || (((uintptr_t)x | (uintptr_t)w) & 15) == 0, "fc: x/w must be 16-byte aligned");
    N3dProfScope prof(N3D_K_FC, stream, 2.0 * N * (double)I * O, 4.0 * ((double)I * O + (double)N * (I + O)));
    for (int n0 = 0; n0 < N; n0 += FC_MAX_N) {
        const int nn = (N - n0) < FC_MAX_N ? (N - n0) : FC_MAX_N;
        if (pre_square)
            hipLaunchKernelGGL(fc_kernel<true>, dim3(cdiv(O, 4)), dim3(256), 0, stream, x, w, b, y, n0, nn, N, I, O, wgain,
                               bgain, act, alpha, gain, post_rsqrt);
        else
            hipLaunchKernelGGL(fc_kernel<false>, dim3(cdiv(O, 4)), dim3(256), 0, stream, x, w, b, y, n0, nn, N, I, O, wgain,
                               bgain, act, alpha, gain, post_rsqrt);
        N3D_LAUNCH_CHECK();
    }
    return 0;
}
