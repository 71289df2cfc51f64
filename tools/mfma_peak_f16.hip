// Standalone probe (the f16 twin of tools/mfma_peak.hip): what the matrix pipe sustains with v_mfma_f32_32x32x16_f16 and the float16
// convolution kernels' instruction mix (conv2d_f16.hip), so that `roofline_f16` can be priced against a MEASURED ceiling as well.
//   mode 0: 4 independent accumulators, one MFMA each per step, operands in registers
//   mode 1: + ds_read_b128 fragment reads at READS per 4 MFMAs (4 = the <2,2> tile without row reuse, 2 = the <2,4> tile with it)
//   mode 2: mode 1 + one s_barrier per 36 MFMAs (a 16-channel chunk of the <2,2> tile)
// Build (build container): hipcc -O3 --offload-arch=gfx950 tools/mfma_peak_f16.hip -o tools/probe/mfma_peak_f16.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int READS>
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters, int rnd) {
    __shared__ f16x8 lds[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 512) {
        f16x8 v;
        for (int k = 0; k < 8; ++k) { unsigned h = (unsigned)(i * 8 + k) * 2654435761u; h ^= h >> 13; v[k] = (_Float16)(rnd ? ((int)(h & 0xffff) - 32768) * (1.f / 16384.f) : 0.001f * ((i + k) & 15)); }
        lds[i] = v;
    }
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 a[2], b[2];
    for (int m = 0; m < 2; ++m) { a[m] = lds[lane + 64 * m]; b[m] = lds[lane + 256 + 64 * m]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (MODE >= 1) {
                const int o = ((it * 9 + t) & 3) * 512 + lane;
                a[0] = lds[o]; b[0] = lds[o + 256];
                if (READS >= 4) { a[1] = lds[o + 64]; b[1] = lds[o + 320]; }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[nt], acc[mt * 2 + nt], 0, 0, 0);
        }
        if (MODE >= 2) __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int a_ = 0; a_ < 4; ++a_) for (int r = 0; r < 16; ++r) s += acc[a_][r];
    if (s == 123.456f) out[0] = s;
}

template <int MODE, int READS>
static void run(int wgs, int iters, float* d, int rnd) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<MODE, READS>), dim3(wgs), dim3(512), 0, 0, d, iters, rnd);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<MODE, READS>), dim3(wgs), dim3(512), 0, 0, d, iters, rnd);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double flops = (double)wgs * 8 * iters * 36 * 32768.0;
    const double cyc_per_simd = (double)wgs * 8 * iters * 36 * 32.0 / (256.0 * 4.0);       // 32 cycles per MFMA (8 passes x 4)
    printf("%s mode %d reads/4mfma %d  %5d workgroups x %4d iters: %8.1f us  %7.1f TFLOP/s f16  pipe-limited clock >= %.3f GHz\n",
           rnd ? "random" : "tiny  ", MODE, MODE ? READS : 0, wgs, iters, ms * 1e3, flops / ms / 1e9, cyc_per_simd / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    float* d; hipMalloc(&d, 64);
    for (int rnd = 0; rnd < 2; ++rnd)
        for (int wgs : {256, 512, 2048})
            for (int iters : {96, 768}) {
                run<0, 4>(wgs, iters, d, rnd); run<1, 4>(wgs, iters, d, rnd); run<1, 2>(wgs, iters, d, rnd); run<2, 4>(wgs, iters, d, rnd); run<2, 2>(wgs, iters, d, rnd);
            }
    return 0;
}
