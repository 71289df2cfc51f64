// Stand-alone (no Python, no torch) reproducer of the rasteriser co-residency finding of DESIGN.md 3.3: the rasteriser's kernels,
// built with PLAIN vector-L1-served table loads (RASTER_VARIANT 0: tools/probe/libraster_v0.so), return different results from run to
// run while 8-wave split-bf16 convolution workgroups of ANOTHER stream are resident on the same CUs; the shipped build
// (RASTER_VARIANT 15, agent-scope loads: libraster_v15.so or libn3d.so itself) does not.  Everything goes through the C ABI of
// include/n3d.h: n3d_rasterize_views from the raster library under test, n3d_conv2d_prep_weight_bf16x3 / n3d_conv2d_bf16x3 from libn3d.so.
//
//   hipcc -O2 --offload-arch=gfx950 tools/probe/raster_coresidency_repro.cpp -o tools/probe/raster_repro -ldl
//   python tools/probe/dump_mesh.py                                   (inputs: the demo mesh at batch 4 -> tools/probe/raster_inputs.bin)
//   tools/build_raster_variants.sh                                     (build container: tools/probe/libraster_v{0,15,...}.so)
//   tools/probe/raster_repro tools/probe/libraster_v0.so next3d_amd/libn3d.so tools/probe/raster_inputs.bin 36
// Prints, per run, the number of grid / alpha words that differ from a reference rasterisation done ALONE on the device, and a summary.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/n3d.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef int (*rasterize_views_fn)(const float*, const float*, const float*, const int*, const float*, const float*, int, int, float*, unsigned long long*,
                                  float*, float*, float*, int, int, int, int, int, int, int, float, float, float, float, int, int, n3d_stream_t);
typedef int (*prep_fn)(const float*, void*, int, int, int, n3d_stream_t);
typedef int (*conv_fn)(const n3d_conv2d_desc*, n3d_stream_t);

// ---- synthetic co-runners (argv[5] = 1..5) instead of the convolution: which property of an 8-wave convolution workgroup does it take?
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// mode 1: range-checked buffer loads with OUT-OF-RANGE lanes (how the convolutions fetch their halo: offset 0x80000000 reads as zero)
// mode 2: the same loads, every lane in range
template <bool OOB>
__global__ __launch_bounds__(512) void corun_buffer_loads(const float* src, int nfloats, float* sink, int iters) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nfloats * 4, 0x00020000);
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int idx = ((blockIdx.x * 512 + threadIdx.x) * 37 + it * 8191) % nfloats;
        const int voff = (OOB && ((lane + it) & 3) == 0) ? (int)0x80000000 : idx * 4;
        acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// mode 3: 150 KB of LDS per workgroup + MFMA, no global traffic inside the loop
__global__ __launch_bounds__(512) void corun_lds_mfma(float* sink, int iters) {
    __shared__ bf16x8 sm[9600];
    for (int i = threadIdx.x; i < 9600; i += 512) { bf16x8 v; for (int k = 0; k < 8; ++k) v[k] = (__bf16)(float)(i + k); sm[i] = v; }
    __syncthreads();
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        const bf16x8 a = sm[(threadIdx.x * 7 + it * 13) % 9600], b = sm[(threadIdx.x * 3 + it * 29) % 9600];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    if (acc[0] == 12345.678f) sink[0] = acc[0];
}
// mode 4: streaming 16-byte stores (the convolutions' epilogues)
__global__ __launch_bounds__(512) void corun_stores(float4* dst, int n4, int iters) {
    for (int it = 0; it < iters; ++it) dst[((size_t)(blockIdx.x * 512 + threadIdx.x) + (size_t)it * 131072) % n4] = make_float4(it, 1.f, 2.f, 3.f);
}
// mode 5: LDS-DMA (buffer_load ... lds), how the pre-split kernels stage their operands, with out-of-range lanes
typedef __attribute__((address_space(3))) void lds_void;
__global__ __launch_bounds__(512) void corun_lds_dma(const float* src, int nfloats, float* sink, int iters) {
    __shared__ float4 sm[8 * 64 * 4];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nfloats * 4, 0x00020000);
    const int lane = threadIdx.x & 63, wn = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
        const int idx = (((blockIdx.x * 512 + threadIdx.x) * 5 + it * 4099) % (nfloats / 4)) * 16;
        const int voff = ((lane + it) & 7) == 0 ? (int)0x80000000 : idx;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(sm + (wn * 4 + (it & 3)) * 64), 16, voff, 0, 0, 0);
        if ((it & 3) == 3) __builtin_amdgcn_s_waitcnt(0x0f70);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (sm[threadIdx.x].x == 12345.678f) sink[0] = 1.f;
}

// mode 6: dense MFMA issue from registers only — 12 independent accumulators per wave like the convolutions' inner loop, no LDS, no memory
__global__ __launch_bounds__(512) void corun_mfma_dense(float* sink, int iters, float seed) {
    f32x16 acc[12];
    bf16x8 a[2], b[2];
    for (int k = 0; k < 8; ++k) { a[0][k] = (__bf16)(seed + threadIdx.x * 0.001f + k); a[1][k] = (__bf16)(seed * 0.5f + k); b[0][k] = (__bf16)(0.25f * k + threadIdx.x * 0.002f); b[1][k] = (__bf16)(1.f - 0.1f * k); }
    for (int q = 0; q < 12; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 12; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q & 1], b[(q >> 1) & 1], acc[q], 0, 0, 0);
    }
    float t = 0.f;
    for (int q = 0; q < 12; ++q) t += acc[q][0];
    if (t == 12345.678f) sink[0] = t;
}

// victim B (argv[6] = 1): a plain gather kernel instead of the rasteriser — every lane sums 24 table entries at pseudo-random indices
__global__ __launch_bounds__(256) void victim_gather(const float* __restrict__ table, int n, float* __restrict__ out, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    unsigned h = (unsigned)i * 2654435761u;
    float acc = 0.f;
    for (int k = 0; k < 24; ++k) { h = h * 1664525u + 1013904223u; acc += table[(h >> 8) % (unsigned)n] * (float)(k + 1); }
    out[i] = acc;
}

template <typename T> static T* to_device(const std::vector<T>& h) {
    T* d; HIPCHECK(hipMalloc(&d, h.size() * sizeof(T))); HIPCHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <raster lib> <libn3d.so> <raster_inputs.bin> [runs]\n", argv[0]); return 1; }
    const int runs = argc > 4 ? atoi(argv[4]) : 36;
    const int mode = argc > 5 ? atoi(argv[5]) : 0;      // 0: libn3d.so's convolution; 1-6: the synthetic co-runners above
    const int victim = argc > 6 ? atoi(argv[6]) : 0;    // 0: the rasteriser (library under test); 1: victim_gather
    const int nco = argc > 7 ? atoi(argv[7]) : 6;       // co-runner launches queued before and after the victim
    void* hr = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    void* hc = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!hr || !hc) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    auto rasterize = (rasterize_views_fn)dlsym(hr, "n3d_rasterize_views");
    auto prep = (prep_fn)dlsym(hc, "n3d_conv2d_prep_weight_bf16x3");
    auto conv = (conv_fn)dlsym(hc, "n3d_conv2d_bf16x3");
    if (!rasterize || !prep || !conv) { fprintf(stderr, "missing symbol\n"); return 1; }

    FILE* fh = fopen(argv[3], "rb");
    if (!fh) { perror(argv[3]); return 1; }
    int hdr[9];
    if (fread(hdr, 4, 9, fh) != 9) return 1;
    const int N = hdr[0], V = hdr[1], Lm = hdr[2], F = hdr[3], views = hdr[4], H = hdr[5], W = hdr[6], MH = hdr[7], MW = hdr[8];
    auto rd_f = [&](size_t n) { std::vector<float> v(n); if (fread(v.data(), 4, n, fh) != n) exit(3); return v; };
    std::vector<float> verts = rd_f((size_t)N * V * 3), lms = rd_f((size_t)N * Lm * 3), rot = rd_f((size_t)views * 9);
    std::vector<int> faces((size_t)F * 3);
    if (fread(faces.data(), 4, faces.size(), fh) != faces.size()) return 3;
    std::vector<float> face_uv = rd_f((size_t)F * 9), mask = rd_f((size_t)MH * MW);
    fclose(fh);

    float *d_verts = to_device(verts), *d_lms = to_device(lms), *d_rot = to_device(rot), *d_fuv = to_device(face_uv), *d_mask = to_device(mask);
    int* d_faces = to_device(faces);
    const size_t NV = (size_t)N * views, npix = NV * H * W;
    float *tv, *grid, *alpha, *lm2d; unsigned long long* zbuf;
    HIPCHECK(hipMalloc(&tv, NV * V * 3 * 4)); HIPCHECK(hipMalloc(&zbuf, npix * 8)); HIPCHECK(hipMalloc(&grid, npix * 2 * 4));
    HIPCHECK(hipMalloc(&alpha, npix * 4)); HIPCHECK(hipMalloc(&lm2d, (size_t)N * Lm * 2 * 4));

    // the co-resident work: a 512 -> 512 channel 3x3 layer at 64 x 64, batch 4 (8-wave workgroups, ~150 KB of LDS each), float32 NCHW in / out
    const int CN = 4, CI = 512, CO = 512, CH = 64, CW = 64;
    std::vector<float> hx((size_t)CN * CI * CH * CW), hw((size_t)CO * CI * 9);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.02f;
    float *d_x = to_device(hx), *d_w = to_device(hw), *d_y; void* d_wt16;
    HIPCHECK(hipMalloc(&d_y, (size_t)CN * CO * CH * CW * 4));
    HIPCHECK(hipMalloc(&d_wt16, (size_t)9 * CI * 512 * 4));
    hipStream_t sa, sb;
    HIPCHECK(hipStreamCreate(&sa)); HIPCHECK(hipStreamCreate(&sb));
    if (prep(d_w, d_wt16, CO, CI, 3, sb) != 0) { fprintf(stderr, "prep failed\n"); return 1; }
    n3d_conv2d_desc cd; memset(&cd, 0, sizeof(cd));
    cd.x = d_x; cd.wt = (const float*)d_wt16; cd.y = d_y; cd.N = CN; cd.I = CI; cd.O = CO; cd.H = CH; cd.W = CW; cd.ksize = 3; cd.mode = 0; cd.ksplit = 1;
    cd.x_batch_stride = (int64_t)CI * CH * CW; cd.y_batch_stride = (int64_t)CO * CH * CW;
    cd.epi.const_scale = 1.f; cd.epi.act = N3D_ACT_LINEAR; cd.epi.gain = 1.f; cd.epi.clamp = -1.f;

    auto raster = [&](hipStream_t st) {
        return rasterize(d_verts, d_lms, d_rot, d_faces, d_fuv, d_mask, MH, MW, tv, zbuf, grid, alpha, lm2d, N, V, Lm, F, views, H, W, 0.f, -0.01f, -0.01f, 5.f, 1, 1, st);
    };
    std::vector<uint32_t> ref_g(npix * 2), ref_a(npix), got_g(npix * 2), got_a(npix);
    if (raster(sa) != 0) { fprintf(stderr, "rasterize failed\n"); return 1; }
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(ref_g.data(), grid, npix * 8, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(ref_a.data(), alpha, npix * 4, hipMemcpyDeviceToHost));
    int bad_runs = 0; long long bad_words = 0;
    float4* d_dst; HIPCHECK(hipMalloc(&d_dst, (size_t)64 << 20));
    auto corun = [&]() {
        if (mode == 0) { if (conv(&cd, sb) != 0) { fprintf(stderr, "conv failed\n"); exit(1); } }
        else if (mode == 1) hipLaunchKernelGGL(corun_buffer_loads<true>, dim3(512), dim3(512), 0, sb, d_x, 1 << 23, d_y, 4000);
        else if (mode == 2) hipLaunchKernelGGL(corun_buffer_loads<false>, dim3(512), dim3(512), 0, sb, d_x, 1 << 23, d_y, 4000);
        else if (mode == 3) hipLaunchKernelGGL(corun_lds_mfma, dim3(256), dim3(512), 0, sb, d_y, 6000);
        else if (mode == 4) hipLaunchKernelGGL(corun_stores, dim3(512), dim3(512), 0, sb, d_dst, (64 << 20) / 16, 2000);
        else if (mode == 5) hipLaunchKernelGGL(corun_lds_dma, dim3(512), dim3(512), 0, sb, d_x, 1 << 23, d_y, 4000);
        else hipLaunchKernelGGL(corun_mfma_dense, dim3(512), dim3(512), 0, sb, d_y, 3000, 1.5f);
    };
    if (victim == 1) {                                   // the plain gather victim: 4 M lanes x 24 gathers from a 4 MB table (= d_x's first floats)
        const int total = 1 << 22, tn = 1 << 20;
        float* d_out; HIPCHECK(hipMalloc(&d_out, (size_t)total * 4));
        std::vector<uint32_t> ref(total), got(total);
        hipLaunchKernelGGL(victim_gather, dim3(total / 256), dim3(256), 0, sa, d_x, tn, d_out, total);
        HIPCHECK(hipDeviceSynchronize());
        HIPCHECK(hipMemcpy(ref.data(), d_out, (size_t)total * 4, hipMemcpyDeviceToHost));
        int bad = 0; long long words = 0;
        for (int r = 0; r < runs; ++r) {
            for (int k = 0; k < nco; ++k) corun();
            hipLaunchKernelGGL(victim_gather, dim3(total / 256), dim3(256), 0, sa, d_x, tn, d_out, total);
            for (int k = 0; k < nco; ++k) corun();
            HIPCHECK(hipDeviceSynchronize());
            HIPCHECK(hipMemcpy(got.data(), d_out, (size_t)total * 4, hipMemcpyDeviceToHost));
            long long d = 0;
            for (int i = 0; i < total; ++i) d += got[i] != ref[i];
            bad += d != 0; words += d;
        }
        printf("victim_gather, co-runner mode %d: %d of %d co-resident runs differ (%lld words)\n", mode, bad, runs, words);
        return 0;
    }
    for (int r = 0; r < runs; ++r) {
        for (int k = 0; k < nco; ++k) corun();                                   // keeps the CUs full of 8-wave workgroups
        if (raster(sa) != 0) return 1;
        for (int k = 0; k < nco; ++k) corun();
        HIPCHECK(hipDeviceSynchronize());
        HIPCHECK(hipMemcpy(got_g.data(), grid, npix * 8, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(got_a.data(), alpha, npix * 4, hipMemcpyDeviceToHost));
        long long d = 0;
        for (size_t i = 0; i < npix * 2; ++i) d += got_g[i] != ref_g[i];
        for (size_t i = 0; i < npix; ++i) d += got_a[i] != ref_a[i];
        printf("run %2d: %lld words differ from the solo rasterisation\n", r, d);
        bad_runs += d != 0; bad_words += d;
    }
    printf("%s, co-runner mode %d: %d of %d co-resident runs differ (%lld words)\n", argv[1], mode, bad_runs, runs, bad_words);
    return 0;
}
