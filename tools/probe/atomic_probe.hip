// Probe: do global atomics lose updates while another stream runs the 8-wave split-bf16 conv kernels?
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" {
// every thread t: atomicMin(z64[t % M], key(t)), atomicMin(z32[t % M], (uint)t), atomicAdd(count, 1)
__global__ void probe_kernel(unsigned long long* z64, unsigned int* z32, unsigned int* count, int M, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    // a little arithmetic so the kernel lives long enough to overlap
    float a = (float)t;
    for (int i = 0; i < 64; ++i) a = a * 1.0001f + 0.5f;
    const unsigned long long key = ((unsigned long long)(unsigned)(T - t) << 32) | (unsigned)t | (a < 0.f ? 1ull : 0ull);
    atomicMin(&z64[t % M], key);
    atomicMin(&z32[t % M], (unsigned)(T - t));
    atomicAdd(count, 1u);
}
__global__ void probe_clear(unsigned long long* z64, unsigned int* z32, unsigned int* count, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) { z64[i] = ~0ull; z32[i] = ~0u; }
    if (i == 0) *count = 0;
}
int probe_run(void* z64, void* z32, void* count, int M, int T, void* stream) {
    hipLaunchKernelGGL(probe_clear, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)z64, (unsigned int*)z32, (unsigned int*)count, M);
    hipLaunchKernelGGL(probe_kernel, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)z64, (unsigned int*)z32, (unsigned int*)count, M, T);
    return (int)hipGetLastError();
}
}
extern "C" {
// arithmetic probe: IEEE divisions + fma chain + transcendental per thread; output must not depend on what else runs
__global__ void probe_math(float* out, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float a = 1.0f + (float)(t % 977) * 0.37f, b = 3.0f + (float)(t % 613) * 0.11f, acc = 0.f;
    for (int i = 0; i < 200; ++i) {
        const float q = (a + (float)i) / (b + 0.25f * (float)i);      // IEEE division sequence
        acc = acc * 0.999f + q;
        acc += sqrtf(q) - floorf(q);
    }
    out[t] = acc;
}
int probe_math_run(void* out, int T, void* stream) {
    hipLaunchKernelGGL(probe_math, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, (float*)out, T);
    return (int)hipGetLastError();
}
}
extern "C" {
// gather probe: out[t] = sum of 9 dependent gathers data[3*idx[3f+k] + c] (the rasteriser's access pattern)
__global__ void probe_gather(const float* __restrict__ data, const int* __restrict__ idx, float* __restrict__ out, int F, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int f = t % F;
    float acc = 0.f;
    for (int k = 0; k < 3; ++k) {
        const float* v = data + 3 * (long)idx[3 * f + k];
        acc += v[0] * 1.f + v[1] * 2.f + v[2] * 3.f;
    }
    out[t] = acc;
}
int probe_gather_run(const void* data, const void* idx, void* out, int F, int T, void* stream) {
    hipLaunchKernelGGL(probe_gather, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)data, (const int*)idx, (float*)out, F, T);
    return (int)hipGetLastError();
}
}
extern "C" {
// many non-returning 64-bit atomicMin per thread (the rasteriser's pattern: one per covered pixel, then the thread exits)
__global__ void probe_many(unsigned long long* z64, int M, int T, int per) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    unsigned int s = (unsigned)t * 2654435761u + 12345u;
    for (int i = 0; i < per; ++i) {
        s = s * 1664525u + 1013904223u;
        const int slot = (int)((s >> 8) % (unsigned)M);
        const unsigned long long key = ((unsigned long long)(s >> 4) << 32) | (unsigned)t;
        atomicMin(&z64[slot], key);
    }
}
int probe_many_run(void* z64, int M, int T, int per, void* stream) {
    hipLaunchKernelGGL(probe_many, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)z64, M, T, per);
    return (int)hipGetLastError();
}
}
