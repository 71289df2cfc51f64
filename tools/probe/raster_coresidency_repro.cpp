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

#include <unordered_set>
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
    if (getenv("RASTER_CLASSIFY")) { printf("--- rocm-smi --showrasinfo all (before) ---\n"); fflush(stdout); if (system("rocm-smi --showrasinfo all 2>&1 | grep -v '^$' | head -60") != 0) printf("(rocm-smi failed)\n"); }
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
    // round 5: WHICH stage goes wrong, and WHAT the wrong words are.  n3d_rasterize_views leaves its intermediates in
    // caller-owned buffers: tv (transformed vertices: raster_transform_kernel's output, the z-buffer kernel's gather table) and zbuf (64-bit
    // z|face keys: raster_faces_kernel's atomicMin target, raster_resolve_kernel's input) — both are compared with the solo run's, and every
    // differing word is classified: equal to ANOTHER entry of the reference table (a mis-addressed / stale gather), one bit flipped (an SRAM
    // upset), the cleared value, or something else.
    const bool classify = getenv("RASTER_CLASSIFY") != nullptr;
    std::vector<uint32_t> ref_tv(NV * V * 3), got_tv(NV * V * 3);
    std::vector<unsigned long long> ref_z(npix), got_z(npix);
    HIPCHECK(hipMemcpy(ref_tv.data(), tv, ref_tv.size() * 4, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(ref_z.data(), zbuf, npix * 8, hipMemcpyDeviceToHost));
    long long tv_bad = 0, z_bad = 0, z_face = 0, z_zonly = 0, z_cleared = 0, z_to_cleared = 0, z_face_invalid = 0, z_onebit = 0, z_neighbour = 0, z_got_closer = 0, z_got_farther = 0;
    long long ga_runs_without_z = 0, printed = 0;
    // raw gathered words of raster_resolve_kernel (libraster_v64 / v79: n3d_raster_debug_buffer): 18 per pixel = 3 vertex indices, 9 vertex
    // coordinates, 6 uv coordinates of the winning face
    typedef int (*dbg_fn)(void*);
    auto set_dbg = (dbg_fn)dlsym(hr, "n3d_raster_debug_buffer");
    uint32_t* d_dbg = nullptr;
    std::vector<uint32_t> ref_dbg, got_dbg;
    std::unordered_set<uint32_t> tv_words, fuv_words;
    long long w_total = 0, w_idx = 0, w_vert = 0, w_fuv = 0, w_other_entry = 0, w_onebit = 0, w_zero = 0, w_else = 0, px_bad = 0, px_idx_consistent = 0, px_whole_vertex_of_other_index = 0, dbg_printed = 0;
    long long w_idx_valid = 0, w_idx_same_face_rotated = 0;
    if (classify && set_dbg) {
        HIPCHECK(hipMalloc(&d_dbg, npix * 18 * 4)); HIPCHECK(hipMemset(d_dbg, 0, npix * 18 * 4));
        if (set_dbg(d_dbg) != 0) { fprintf(stderr, "n3d_raster_debug_buffer failed\n"); return 1; }
        if (raster(sa) != 0) return 1;
        HIPCHECK(hipDeviceSynchronize());
        ref_dbg.resize(npix * 18); got_dbg.resize(npix * 18);
        HIPCHECK(hipMemcpy(ref_dbg.data(), d_dbg, npix * 18 * 4, hipMemcpyDeviceToHost));
        for (uint32_t w : ref_tv) tv_words.insert(w);
        for (float f : face_uv) { uint32_t w; memcpy(&w, &f, 4); fuv_words.insert(w); }
        // sanity: the solo dump equals what the host reads from the reference tables
        long long host_bad = 0;
        for (size_t i = 0; i < npix; ++i) {
            if (ref_z[i] == ~0ull) continue;
            const uint32_t f = (uint32_t)ref_z[i]; const size_t nv = i / ((size_t)H * W);
            for (int k = 0; k < 3; ++k) {
                const int vi = faces[3 * f + k];
                host_bad += ref_dbg[i * 18 + k] != (uint32_t)vi;
                for (int c = 0; c < 3; ++c) host_bad += ref_dbg[i * 18 + 3 + 3 * k + c] != ref_tv[(nv * V + vi) * 3 + c];
            }
        }
        printf("raw-word dump: solo run vs the tables read on the host: %lld words differ\n", host_bad);
    }
    // ... and of raster_faces_kernel (the z-buffer pass): 12 words per (view, face) = 3 vertex indices + 9 vertex coordinates
    auto set_dbg2 = (dbg_fn)dlsym(hr, "n3d_raster_debug_buffer_faces");
    uint32_t* d_dbg2 = nullptr;
    const size_t nfr = NV * (size_t)F, nfw = nfr * 14;     // 12 gathered words per record, then (updates issued, checksum of their keys) per record
    std::vector<uint32_t> ref_f2, got_f2;
    long long f_bad = 0, fw_total = 0, fw_idx = 0, fw_coord = 0, fw_zero = 0, fw_other = 0, fw_onebit = 0, fw_else = 0, f_consistent = 0, f_printed = 0, f_idx_valid = 0, f_all_words_wrong = 0;
    long long f_run_lanes = 0, f_runs_count = 0, issued_total = 0, issued_count_bad = 0, issued_keys_bad = 0;
    if (classify && set_dbg2) {
        HIPCHECK(hipMalloc(&d_dbg2, nfw * 4)); HIPCHECK(hipMemset(d_dbg2, 0, nfw * 4));
        if (set_dbg2(d_dbg2) != 0) return 1;
        if (raster(sa) != 0) return 1;
        HIPCHECK(hipDeviceSynchronize());
        ref_f2.resize(nfw); got_f2.resize(nfw);
        HIPCHECK(hipMemcpy(ref_f2.data(), d_dbg2, nfw * 4, hipMemcpyDeviceToHost));
        long long host_bad = 0;
        for (size_t nvf = 0; nvf < NV * (size_t)F; ++nvf) {
            const size_t nv = nvf / F, f = nvf % F;
            for (int k = 0; k < 3; ++k) {
                const int vi = faces[3 * f + k];
                host_bad += ref_f2[nvf * 12 + k] != (uint32_t)vi;
                for (int c = 0; c < 3; ++c) host_bad += ref_f2[nvf * 12 + 3 + 3 * k + c] != ref_tv[(nv * V + vi) * 3 + c];
            }
        }
        printf("raw-word dump of raster_faces_kernel: solo run vs the tables read on the host: %lld words differ\n", host_bad);
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
        if (classify) {
            HIPCHECK(hipMemcpy(got_tv.data(), tv, got_tv.size() * 4, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(got_z.data(), zbuf, npix * 8, hipMemcpyDeviceToHost));
            long long t = 0, z = 0;
            for (size_t i = 0; i < got_tv.size(); ++i) t += got_tv[i] != ref_tv[i];
            for (size_t i = 0; i < npix; ++i) {
                if (got_z[i] == ref_z[i]) continue;
                ++z;
                const unsigned long long g = got_z[i], e = ref_z[i];
                const uint32_t gf = (uint32_t)g, ef = (uint32_t)e, gz = (uint32_t)(g >> 32), ez = (uint32_t)(e >> 32);
                if (e == ~0ull) ++z_cleared;                                     // solo: no face here; co-resident: some face won
                if (g == ~0ull) ++z_to_cleared;                                  // co-resident: NO face won where one should have
                if (gf != ef) ++z_face; else ++z_zonly;
                if (g != ~0ull && gf >= (uint32_t)F) ++z_face_invalid;
                if (__builtin_popcountll(g ^ e) == 1) ++z_onebit;
                if (g != ~0ull && e != ~0ull) { if (gz < ez) ++z_got_closer; else if (gz > ez) ++z_got_farther; }
                // is the winning key of a NEIGHBOURING pixel (the face that should have lost here, or that pixel's own result)?
                const size_t x = i % W, y = (i / W) % H;
                bool nb = false;
                for (int dy = -1; dy <= 1 && !nb; ++dy) for (int dx = -1; dx <= 1 && !nb; ++dx) {
                    if ((!dx && !dy) || (int)x + dx < 0 || (int)x + dx >= W || (int)y + dy < 0 || (int)y + dy >= H) continue;
                    nb = (uint32_t)ref_z[i + dy * W + dx] == gf;
                }
                z_neighbour += nb;
                if (printed < 24) { printf("  zbuf[%zu] (view %zu, y %zu, x %zu): solo %016llx  co-resident %016llx%s\n", i, i / ((size_t)H * W), y, x, e, g, nb ? "  (face of a neighbouring pixel)" : ""); ++printed; }
            }
            tv_bad += t; z_bad += z;
            ga_runs_without_z += (d != 0 && z == 0);
            if (d_dbg2) {
                HIPCHECK(hipMemcpy(got_f2.data(), d_dbg2, nfw * 4, hipMemcpyDeviceToHost));
                long long prev_bad = -2, run_len = 0;
                for (size_t nvf = 0; nvf < NV * (size_t)F; ++nvf) {
                    const uint32_t* g = &got_f2[nvf * 12]; const uint32_t* e = &ref_f2[nvf * 12];
                    if (!memcmp(g, e, 48)) continue;
                    ++f_bad;
                    if ((long long)nvf == prev_bad + 1) ++run_len; else { if (run_len) { f_run_lanes += run_len; ++f_runs_count; } run_len = 1; }
                    prev_bad = (long long)nvf;
                    const size_t nv = nvf / F;
                    bool consistent = true, allwrong = true;
                    for (int k = 0; k < 3; ++k)
                        for (int c = 0; c < 3; ++c) consistent &= g[k] < (uint32_t)V && g[3 + 3 * k + c] == ref_tv[(nv * V + g[k]) * 3 + c];
                    f_consistent += consistent;
                    for (int k = 0; k < 12; ++k) {
                        if (g[k] == e[k]) { allwrong = false; continue; }
                        ++fw_total;
                        if (k < 3) { ++fw_idx; f_idx_valid += g[k] < (uint32_t)V; } else ++fw_coord;
                        if (g[k] == 0) ++fw_zero;
                        else if (k >= 3 && tv_words.count(g[k])) ++fw_other;
                        else if (__builtin_popcount(g[k] ^ e[k]) == 1) ++fw_onebit;
                        else if (k >= 3) ++fw_else;
                    }
                    f_all_words_wrong += allwrong;
                    if (f_printed < 16) {
                        printf("  face record %zu (view %zu, face %zu):", nvf, nv, nvf % F);
                        for (int k = 0; k < 12; ++k) if (g[k] != e[k]) printf(" [%d] %08x -> %08x", k, e[k], g[k]);
                        printf("\n"); ++f_printed;
                    }
                }
                if (run_len) { f_run_lanes += run_len; ++f_runs_count; }
                for (size_t nvf = 0; nvf < nfr; ++nvf) {
                    issued_total += got_f2[nfr * 12 + nvf * 2];
                    if (got_f2[nfr * 12 + nvf * 2] != ref_f2[nfr * 12 + nvf * 2]) ++issued_count_bad;
                    else if (got_f2[nfr * 12 + nvf * 2 + 1] != ref_f2[nfr * 12 + nvf * 2 + 1]) ++issued_keys_bad;
                }
            }
            if (d_dbg) {
                HIPCHECK(hipMemcpy(got_dbg.data(), d_dbg, npix * 18 * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < npix; ++i) {
                    if (got_z[i] != ref_z[i] || ref_z[i] == ~0ull) continue;             // same winning face: the resolve kernel's own gathers
                    const uint32_t* g = &got_dbg[i * 18]; const uint32_t* e = &ref_dbg[i * 18];
                    if (!memcmp(g, e, 72)) continue;
                    ++px_bad;
                    const size_t nv = i / ((size_t)H * W);
                    bool consistent = true;                                            // do the vertex words match the (possibly wrong) index words?
                    for (int k = 0; k < 3; ++k)
                        for (int c = 0; c < 3; ++c) consistent &= g[k] < (uint32_t)V && g[3 + 3 * k + c] == ref_tv[(nv * V + g[k]) * 3 + c];
                    px_idx_consistent += consistent && (g[0] != e[0] || g[1] != e[1] || g[2] != e[2]);
                    for (int k = 0; k < 18; ++k) {
                        if (g[k] == e[k]) continue;
                        ++w_total;
                        if (k < 3) { ++w_idx; w_idx_valid += g[k] < (uint32_t)V; w_idx_same_face_rotated += (g[k] == e[0] || g[k] == e[1] || g[k] == e[2]); }
                        else if (k < 12) ++w_vert; else ++w_fuv;
                        const bool other = k < 3 ? false : (k < 12 ? tv_words.count(g[k]) > 0 : fuv_words.count(g[k]) > 0);
                        if (other) ++w_other_entry;
                        else if (__builtin_popcount(g[k] ^ e[k]) == 1) ++w_onebit;
                        else if (g[k] == 0) ++w_zero;
                        else if (k >= 3) ++w_else;
                    }
                    // a whole vertex (3 words) replaced by ANOTHER vertex of the table?
                    for (int k = 0; k < 3; ++k) {
                        if (!memcmp(g + 3 + 3 * k, e + 3 + 3 * k, 12)) continue;
                        bool found = false;
                        for (size_t vi = 0; vi < (size_t)NV * V && !found; ++vi) found = !memcmp(g + 3 + 3 * k, &ref_tv[vi * 3], 12);
                        px_whole_vertex_of_other_index += found;
                    }
                    if (dbg_printed < 12) {
                        printf("  pixel %zu (face %u): idx solo %u %u %u | got %u %u %u;  differing words:", i, (uint32_t)ref_z[i], e[0], e[1], e[2], g[0], g[1], g[2]);
                        for (int k = 3; k < 18; ++k) if (g[k] != e[k]) printf(" [%d] %08x -> %08x", k, e[k], g[k]);
                        printf("\n"); ++dbg_printed;
                    }
                }
            }
            printf("        transformed-vertex words differing: %lld; z-buffer keys differing: %lld\n", t, z);
        }
    }
    printf("%s, co-runner mode %d: %d of %d co-resident runs differ (%lld words)\n", argv[1], mode, bad_runs, runs, bad_words);
    if (classify) { printf("--- rocm-smi --showrasinfo all (after) ---\n"); fflush(stdout); if (system("rocm-smi --showrasinfo all 2>&1 | grep -v '^$' | head -60") != 0) printf("(rocm-smi failed)\n"); }
    if (classify) {
        printf("classification over %d runs: transformed vertices (raster_transform_kernel's output) differing: %lld words\n", runs, tv_bad);
        printf("  z-buffer keys differing: %lld — another FACE won: %lld, same face / other depth: %lld; face index out of range: %lld; exactly one bit flipped: %lld\n",
               z_bad, z_face, z_zonly, z_face_invalid, z_onebit);
        printf("  solo had no face, co-resident has one: %lld; solo had a face, co-resident has none: %lld; winning depth closer / farther than solo: %lld / %lld\n",
               z_cleared, z_to_cleared, z_got_closer, z_got_farther);
        printf("  the co-resident winner is the solo winner of one of the 8 neighbouring pixels: %lld of %lld\n", z_neighbour, z_bad);
        printf("  runs whose grid / alpha differ although every z-buffer key equals the solo run's: %lld\n", ga_runs_without_z);
        if (d_dbg2) {
            printf("raw gathered words of raster_faces_kernel (one record per view and face): %lld records with a wrong word, %lld wrong words\n", f_bad, fw_total);
            printf("  vertex-index words %lld (valid index: %lld), coordinate words %lld; zero: %lld; another entry of the vertex table: %lld; one bit flipped: %lld; anything else: %lld\n",
                   fw_idx, f_idx_valid, fw_coord, fw_zero, fw_other, fw_onebit, fw_else);
            printf("  records whose coordinate words equal the vertices of the index words READ (a wrong index, followed faithfully): %lld; records with all 12 words wrong: %lld\n", f_consistent, f_all_words_wrong);
            printf("  wrong records come in runs of consecutive lanes: %lld runs, mean length %.1f\n", f_runs_count, f_runs_count ? (double)f_run_lanes / f_runs_count : 0.0);
            printf("  z-buffer updates ISSUED by raster_faces_kernel (atomicMin calls): %lld in all; lanes that issued another NUMBER of updates than solo: %lld; same number, other keys: %lld\n",
                   issued_total, issued_count_bad, issued_keys_bad);
        }
        if (d_dbg) {
            printf("raw gathered words of raster_resolve_kernel on pixels whose z-buffer key is RIGHT: %lld pixels with a wrong word, %lld wrong words\n", px_bad, w_total);
            printf("  by table: vertex indices %lld (of them a valid index: %lld, another corner of the SAME face: %lld), vertex coordinates %lld, face uv %lld\n", w_idx, w_idx_valid, w_idx_same_face_rotated, w_vert, w_fuv);
            printf("  wrong coordinate / uv words that are ANOTHER entry of the same table: %lld; one bit flipped: %lld; zero: %lld; anything else: %lld\n", w_other_entry, w_onebit, w_zero, w_else);
            printf("  pixels whose wrong index words are consistent with the vertex words read (the INDEX gather went wrong, the vertex gather followed it): %lld\n", px_idx_consistent);
            printf("  wrong vertices that are, as a whole (x, y, z), another vertex of the table: %lld\n", px_whole_vertex_of_other_index);
        }
    }
    return 0;
}
