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

template <typename T> static T* to_device(const std::vector<T>& h) {
    T* d; HIPCHECK(hipMalloc(&d, h.size() * sizeof(T))); HIPCHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <raster lib> <libn3d.so> <raster_inputs.bin> [runs]\n", argv[0]); return 1; }
    const int runs = argc > 4 ? atoi(argv[4]) : 36;
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
    for (int r = 0; r < runs; ++r) {
        for (int k = 0; k < 6; ++k) if (conv(&cd, sb) != 0) { fprintf(stderr, "conv failed\n"); return 1; }      // keeps the CUs full of 8-wave workgroups
        if (raster(sa) != 0) return 1;
        for (int k = 0; k < 6; ++k) conv(&cd, sb);
        HIPCHECK(hipDeviceSynchronize());
        HIPCHECK(hipMemcpy(got_g.data(), grid, npix * 8, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(got_a.data(), alpha, npix * 4, hipMemcpyDeviceToHost));
        long long d = 0;
        for (size_t i = 0; i < npix * 2; ++i) d += got_g[i] != ref_g[i];
        for (size_t i = 0; i < npix; ++i) d += got_a[i] != ref_a[i];
        printf("run %2d: %lld words differ from the solo rasterisation\n", r, d);
        bad_runs += d != 0; bad_words += d;
    }
    printf("%s: %d of %d co-resident runs differ (%lld words)\n", argv[1], bad_runs, runs, bad_words);
    return 0;
}
