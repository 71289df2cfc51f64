// Standalone probe: what the bf16 matrix pipe of this chip sustains with the convolution kernels' instruction mix, so that
// their MFMA-only ablation (tools/conv_ps_abl.py, dbg5) can be priced against a MEASURED ceiling instead of the data-sheet one.
//   mode 0: 12 independent-chain MFMAs (4 accumulators x 3) per iteration, operands in registers
//   mode 1: + 8 ds_read_b128 per 12 MFMAs (the pre-split kernel's ratio), conflict-free lane-linear addresses
//   mode 2: mode 1 + one s_barrier per 108 MFMAs (a 16-channel chunk)
// Build (build container): hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/probe/mfma_peak.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters, int rnd) {
    __shared__ bf16x8 lds[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 512) { bf16x8 v; for (int k = 0; k < 8; ++k) { unsigned h = (unsigned)(i * 8 + k) * 2654435761u; h ^= h >> 13; v[k] = (__bf16)(rnd ? ((int)(h & 0xffff) - 32768) * (1.f / 16384.f) : 0.001f * ((i + k) & 15)); } lds[i] = v; }
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 ah[2], al[2], bh[2], bl[2];
    for (int m = 0; m < 2; ++m) { ah[m] = lds[lane + 64 * m]; al[m] = lds[lane + 128 + 64 * m]; bh[m] = lds[lane + 256 + 64 * m]; bl[m] = lds[lane + 384 + 64 * m]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (MODE >= 1) {
                const int o = ((it * 9 + t) & 3) * 512 + lane;
#pragma unroll
                for (int m = 0; m < 2; ++m) { ah[m] = lds[o + 64 * m]; al[m] = lds[o + 128 + 64 * m]; bh[m] = lds[o + 256 + 64 * m]; bl[m] = lds[o + 384 + 64 * m]; }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[nt], al[mt], acc[mt * 2 + nt], 0, 0, 0);
                    acc[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[nt], ah[mt], acc[mt * 2 + nt], 0, 0, 0);
                    acc[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[nt], ah[mt], acc[mt * 2 + nt], 0, 0, 0);
                }
        }
        if (MODE >= 2) __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
static void run(int wgs, int iters, float* d, int rnd) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(512), 0, 0, d, iters, rnd);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(512), 0, 0, d, iters, rnd);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double flops = (double)wgs * 8 * iters * 108 * 32768.0;
    const double cyc_per_simd = (double)wgs * 8 * iters * 108 * 32.0 / (256.0 * 4.0);      // 32 cycles per MFMA (8 passes x 4)
    printf("%s mode %d  %5d workgroups x %4d iters: %8.1f us  %7.1f TFLOP/s bf16 (%6.1f fp32-equivalent / 3)  pipe-limited clock >= %.3f GHz\n",
           rnd ? "random" : "tiny  ", MODE, wgs, iters, ms * 1e3, flops / ms / 1e9, flops / ms / 1e9 / 3, cyc_per_simd / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    float* d; hipMalloc(&d, 64);
    for (int rnd = 0; rnd < 2; ++rnd) {
        for (int wgs : {256, 512, 2048})
            for (int iters : {32, 256}) { run<0>(wgs, iters, d, rnd); run<1>(wgs, iters, d, rnd); run<2>(wgs, iters, d, rnd); }
        for (int iters : {8, 16, 64, 128}) run<2>(256, iters, d, rnd);                  // fixed cost of a one-wave grid
        for (int wgs : {128, 248, 264, 384, 768, 1024}) run<2>(wgs, 32, d, rnd);
    }
    return 0;
}
