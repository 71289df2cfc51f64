// Probe for the rasteriser co-residency corruption: do plain (vector-L1-served) gather loads of an IMMUTABLE table
// return wrong values while workgroups of another stream with a large LDS allocation come and go on the same CUs?
//   reader   : many small workgroups; every lane walks a pseudo-random index sequence over a 4 MB table whose entry i holds
//              hash(i), with plain `global_load_dword` (L1-served) or agent-scope (sc1, L2-served) loads, and counts mismatches.
//   co-runner: on a second stream, short workgroups that allocate `lds_bytes` of LDS, scribble over it and exit — launched back to
//              back so that LDS is allocated / released on the reader's CUs the whole time.  Variants: none, 32 KB x 256 threads,
//              150 KB x 512 threads (the footprint of the 8-wave convolution workgroups), 150 KB x 512 threads with MFMA work.
// Build / run:  hipcc -O3 --offload-arch=gfx950 tools/l1_lds_probe.hip -o /tmp/l1_lds_probe && /tmp/l1_lds_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <bool AGENT>
__global__ __launch_bounds__(256) void reader_kernel(const uint32_t* table, uint32_t n_mask, int iters, unsigned long long* errors, uint32_t* first_bad) {
    uint32_t s = hash32(blockIdx.x * 256 + threadIdx.x + 1);
    unsigned long long bad = 0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const uint32_t idx = (s >> 8) & n_mask;
        const uint32_t v = AGENT ? __hip_atomic_load(table + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : table[idx];
        if (v != hash32(idx)) { if (!bad) { first_bad[0] = idx; first_bad[1] = v; first_bad[2] = hash32(idx); } ++bad; }
    }
    if (bad) atomicAdd(errors, bad);
}

template <bool MFMA>
__global__ void corunner_kernel(int lds_dwords, float* sink) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < lds_dwords; i += blockDim.x) lds[i] = 0xdeadbeefu ^ (uint32_t)i;
    __syncthreads();
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < lds_dwords; i += blockDim.x * 7) acc += lds[i];
    if (MFMA) {
        f32x16 c = {0};
        bf16x8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(float)(threadIdx.x + k); b[k] = (__bf16)(float)(acc & 7); }
        for (int r = 0; r < 64; ++r) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        acc += (uint32_t)c[0];
    }
    if (acc == 0x12345678u) sink[0] = 1.f;
}

int main() {
    const uint32_t n = 1u << 20;                       // 4 MB table
    uint32_t* h = (uint32_t*)malloc(n * 4);
    for (uint32_t i = 0; i < n; ++i) h[i] = hash32(i);
    uint32_t *table, *first_bad; unsigned long long* errors; float* sink;
    hipMalloc(&table, n * 4); hipMalloc(&first_bad, 16); hipMalloc(&errors, 8); hipMalloc(&sink, 4);
    hipMemcpy(table, h, n * 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    hipStream_t s_read, s_co;
    hipStreamCreate(&s_read); hipStreamCreate(&s_co);
    hipFuncSetAttribute((const void*)corunner_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)corunner_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct { const char* name; int lds_bytes, threads; bool mfma; } co[] = {
        {"no co-runner", 0, 0, false}, {"co-runner 32 KB LDS x 256 threads", 32 * 1024, 256, false},
        {"co-runner 150 KB LDS x 512 threads", 150 * 1024, 512, false}, {"co-runner 150 KB LDS x 512 threads + MFMA", 150 * 1024, 512, true}};
    for (int agent = 0; agent < 2; ++agent)
        for (auto& c : co) {
            unsigned long long total = 0; uint32_t fb[4] = {0, 0, 0, 0};
            for (int rep = 0; rep < 6; ++rep) {
                hipMemsetAsync(errors, 0, 8, s_read); hipMemsetAsync(first_bad, 0, 16, s_read);
                hipStreamSynchronize(s_read);
                if (agent) hipLaunchKernelGGL(reader_kernel<true>, dim3(2048), dim3(256), 0, s_read, table, n - 1, 20000, errors, first_bad);
                else hipLaunchKernelGGL(reader_kernel<false>, dim3(2048), dim3(256), 0, s_read, table, n - 1, 20000, errors, first_bad);
                if (c.threads)
                    for (int k = 0; k < 400; ++k) {
                        if (c.mfma) hipLaunchKernelGGL(corunner_kernel<true>, dim3(256), dim3(c.threads), c.lds_bytes, s_co, c.lds_bytes / 4, sink);
                        else hipLaunchKernelGGL(corunner_kernel<false>, dim3(256), dim3(c.threads), c.lds_bytes, s_co, c.lds_bytes / 4, sink);
                    }
                hipDeviceSynchronize();
                unsigned long long e; hipMemcpy(&e, errors, 8, hipMemcpyDeviceToHost);
                if (e && !total) hipMemcpy(fb, first_bad, 12, hipMemcpyDeviceToHost);
                total += e;
            }
            printf("%-7s loads, %-45s: %llu mismatching loads of %.2e", agent ? "agent" : "plain", c.name, total, 6.0 * 2048 * 256 * 20000);
            if (total) printf("   (first: table[%u] read %08x, holds %08x)", fb[0], fb[1], fb[2]);
            printf("\n");
        }
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return 0;
}
