// Winograd F(2x2, 3x3) for the 64 x 64 x 512 -> 512 layer at batch 4: a TRAFFIC-MODEL probe, not a convolution.  It issues exactly the
// memory traffic, LDS traffic, MFMA count and barriers the fused split-bf16 Winograd kernel of docs/history/DESIGN_rounds1-4.md 3.1h would issue — per workgroup 64 output channels x
// 64 tiles (= 8 x 32 output pixels) with all 16 transform-domain accumulators resident (8 waves x 2 positions x 2 x 2 blocks of 32 x 32 = 128 accumulator
// registers per lane), per 16-channel chunk 64 KB of transformed input V by LDS-DMA (double-buffered) and 8 x 1 KB of transformed-weight fragments U per wave straight
// from global memory, 24 MFMAs per wave (3 per product: hi*hi, hi*lo, lo*hi), one barrier — on operands of the right size and layout with random contents, and
// writes 64 KB of "output" per workgroup.  Its time is a LOWER bound for the real kernel (no input transform in the producer, no output transform, no epilogue).
//   mode 0: everything      mode 1: no U loads after the first chunk      mode 2: no V DMA after the first chunk      mode 3: neither (MFMA + LDS reads + barriers)
// Build (build container): hipcc -O3 --offload-arch=gfx950 tools/winograd_probe.hip -o tools/probe/winograd_probe.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int N = 4, I = 512, O = 512, H = 64, W = 64;
constexpr int KC = I / 16, OB = O / 64, TB = (H / 2) * (W / 2) / 64;       // 32 chunks, 8 channel blocks, 16 tile blocks per image
constexpr int V_SLOTS = 16 * 2 * 2 * 64;                                  // 16-byte slots per chunk: [xi][hi|lo][k half][tile] = 4096 (64 KB)
constexpr int U_SLOTS = 16 * 2 * 2 * 64;                                  // per (channel block, chunk): [xi][hi|lo][k half][row]     = 4096 (64 KB)

template <int MODE, int MAP>
__global__ __launch_bounds__(512, 2) void wino_probe(const bf16x8* __restrict__ V, const bf16x8* __restrict__ U, float* __restrict__ y) {
    __shared__ bf16x8 smem[2 * V_SLOTS];
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lb;
    { const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7; lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3); }
    // MAP 0: channel block fastest within an XCD's run of workgroups — the 8 workgroups sharing a V block sit in one XCD's L2 (every XCD then needs all of U);
    // MAP 1: channel block = XCD — an XCD's L2 holds ITS 2.1 MB of U, and every XCD streams all of V (from the Infinity Cache)
    const int ob = MAP == 0 ? lb % OB : (int)(blockIdx.x & 7), rest = MAP == 0 ? lb / OB : (int)(blockIdx.x >> 3), tb = rest % TB, n = rest / TB;
    if (MAP == 1) lb = (n * TB + tb) * OB + ob;
    const bf16x8* Vb = V + ((int64_t)(n * TB + tb) * KC) * V_SLOTS;
    const bf16x8* Ub = U + ((int64_t)ob * KC) * U_SLOTS;
    const __amdgpu_buffer_rsrc_t r_v = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, KC * V_SLOTS * 16, 0x00020000);
    auto copy_v = [&](int kc, int buf) {
#pragma unroll
        for (int j = 0; j < 8; ++j)                                        // 64 pieces of 1 KB, 8 per wave
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_v, (lds_void*)(smem + buf * V_SLOTS + (wn * 8 + j) * 64), 16, lane * 16, (kc * V_SLOTS + (wn * 8 + j) * 64) * 16, 0, 0);
    };
    bf16x8 ua[2][2][2][2];                                                // [buffer][xi of this wave][channel half][hi|lo]
    auto load_u = [&](int kc, int s) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
                    ua[s][x][m][hl] = Ub[(int64_t)kc * U_SLOTS + (((wn * 2 + x) * 2 + hl) * 2 + (lane >> 5)) * 64 + m * 32 + (lane & 31)];
    };
    f32x16 acc[2][2][2];
#pragma unroll
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a >> 2][(a >> 1) & 1][a & 1][r] = 0.f;
    copy_v(0, 0); load_u(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __builtin_amdgcn_s_barrier();
    for (int kc = 0; kc < KC; ++kc) {
        const int s = kc & 1;
        if (kc + 1 < KC) {
            if (MODE == 0 || MODE == 1) copy_v(kc + 1, s ^ 1);
            if (MODE == 0 || MODE == 2) load_u(kc + 1, s ^ 1);
            else { for (int x = 0; x < 2; ++x) for (int m = 0; m < 2; ++m) for (int hl = 0; hl < 2; ++hl) ua[s ^ 1][x][m][hl] = ua[s][x][m][hl]; }
        }
        const bf16x8* B = smem + s * V_SLOTS;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bh[t] = B[(((wn * 2 + x) * 2 + 0) * 2 + (lane >> 5)) * 64 + t * 32 + (lane & 31)];
                bl[t] = B[(((wn * 2 + x) * 2 + 1) * 2 + (lane >> 5)) * 64 + t * 32 + (lane & 31)];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[x][m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[s][x][m][1], bh[t], acc[x][m][t], 0, 0, 0);
                    acc[x][m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[s][x][m][0], bl[t], acc[x][m][t], 0, 0, 0);
                    acc[x][m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[s][x][m][0], bh[t], acc[x][m][t], 0, 0, 0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_s_barrier();
    }
    // "output": the real kernel exchanges the 16 positions of a tile through LDS (2 x 128 KB) and writes 64 channels x 256 pixels; here every wave reduces its
    // accumulators to its share of the workgroup's 64 KB and stores it coalesced
    float* yb = y + (int64_t)lb * (64 * 256);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = acc[q >> 2][(q >> 1) & 1][q & 1][k] + acc[q >> 2][(q >> 1) & 1][q & 1][4 + k] + acc[q >> 2][(q >> 1) & 1][q & 1][8 + k] + acc[q >> 2][(q >> 1) & 1][q & 1][12 + k];
        *reinterpret_cast<f32x4*>(yb + (q * 512 + tid) * 4) = o;
    }
}

template <int MODE, int MAP>
static double run(const bf16x8* V, const bf16x8* U, float* y) {
    const int wgs = N * TB * OB;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((wino_probe<MODE, MAP>), dim3(wgs), dim3(512), 0, 0, V, U, y);
    (void)hipEventRecord(e0, 0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((wino_probe<MODE, MAP>), dim3(wgs), dim3(512), 0, 0, V, U, y);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3;
}

int main() {
    const size_t vb = (size_t)N * TB * KC * V_SLOTS * 16, ub = (size_t)OB * KC * U_SLOTS * 16, yb = (size_t)N * TB * OB * 64 * 256 * 4;
    bf16x8 *V, *U; float* y;
    (void)hipMalloc(&V, vb); (void)hipMalloc(&U, ub); (void)hipMalloc(&y, yb);
    {   // random operands (the matrix pipe's clock depends on them: profiles/r02_mfma_peak_probe.txt)
        unsigned short* h = (unsigned short*)malloc(vb > ub ? vb : ub);
        unsigned s = 12345u;
        for (size_t i = 0; i < vb / 2; ++i) { s = s * 1664525u + 1013904223u; h[i] = (unsigned short)((0x3f80u + ((s >> 16) & 0x7fu)) | ((s >> 8) & 0x8000u)); }
        (void)hipMemcpy(V, h, vb, hipMemcpyHostToDevice);
        for (size_t i = 0; i < ub / 2; ++i) { s = s * 1664525u + 1013904223u; h[i] = (unsigned short)((0x3f80u + ((s >> 16) & 0x7fu)) | ((s >> 8) & 0x8000u)); }
        (void)hipMemcpy(U, h, ub, hipMemcpyHostToDevice);
        free(h);
    }
    const double gf = 2.0 * N * O * I * 9.0 * H * W / 1e9;               // the DIRECT convolution's flops: what a Winograd kernel would be credited with
    printf("Winograd F(2x2,3x3) traffic model, %d x %d x %d -> %d at batch %d: %d workgroups, V %.0f MB, U %.1f MB, per workgroup %d chunks x (64 KB V + 64 KB U), 24 MFMAs per wave and chunk\n",
           H, W, I, O, N, N * TB * OB, vb / 1e6, ub / 1e6, KC);
    for (int map = 0; map < 2; ++map) {
        const double t0 = map ? run<0, 1>(V, U, y) : run<0, 0>(V, U, y), t1 = map ? run<1, 1>(V, U, y) : run<1, 0>(V, U, y),
                     t2 = map ? run<2, 1>(V, U, y) : run<2, 0>(V, U, y), t3 = map ? run<3, 1>(V, U, y) : run<3, 0>(V, U, y);
        printf("workgroup -> XCD mapping %d (%s)\n", map, map ? "channel block = XCD: U resident in each L2, V streamed by every XCD" : "the V block's 8 channel blocks on one XCD: V fetched once, all of U needed by every XCD");
        printf("  all traffic                       %7.1f us  = %6.1f TFLOP/s direct-convolution-equivalent\n", t0, gf / t0 * 1e3);
        printf("  no U loads after chunk 0          %7.1f us\n  no V DMA after chunk 0            %7.1f us\n  MFMA + LDS reads + barriers only  %7.1f us\n", t1, t2, t3);
    }
    printf("(the direct pre-split kernel runs this layer in 183-190 us = 407-423 TFLOP/s inside a forward: profiles/r04_layer_trace.txt)\n");
    return 0;
}
