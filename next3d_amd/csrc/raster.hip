// FLAME-mesh neural-texture rasterisation for gfx950 (compiled with -ffp-contract=off: the face-selection
// arithmetic mirrors oracle/raster_ref.c operation for operation so pix_to_face is bit-identical).
//
//   raster_transform : per view, vertices -> PyTorch3D NDC (y-flip, R, orth shift/scale, proj, +10, x/y negate)
//   raster_faces     : one thread per (sample*view, face): conservative pixel bbox, strict-inside test, 64-bit
//                      atomicMin of (z bits << 32 | face) -> lowest z wins, lowest face index on ties
//   raster_resolve   : per pixel: barycentric uv, vis, bilinear uv_face_mask lookup -> (grid uv, alpha)
//   fill_holes       : the cv2.floodFill step of fill_mouth on the GPU (one workgroup per 256x256 image, LDS state,
//                      alternating row/column sweeps until stable) — removes 4*N device->host->device round trips
//   texture_project  : grid_sample(textures, uv) for front / (side1 + side2) / top planes
//   mouth_bbox       : landmark -> integer crop box, kept on the device (no host sync, no dynamic shapes)
//   resize_aa        : antialiased bilinear resize (ATen _upsample_bilinear2d_aa) with per-sample device-side boxes
//
// Replaces TriPlaneGenerator.rasterize (reference training_avatar_texture/triplane_next3d.py:190-230),
// Pytorch3dRasterizer.forward (volumetric_rendering/renderer.py:401-440 -> third-party pytorch3d rasterize_meshes),
// fill_mouth (renderer.py:583-602 -> third-party cv2.floodFill), gen_mouth_mask (triplane_next3d.py:330-344) and
// F.interpolate(..., mode='bilinear', antialias=True) (triplane_next3d.py:152,161; superresolution.py:282-286).
#include <stdlib.h>

#include "common.h"

#define K_EPS 1e-8f

// RASTER_VARIANT — how the rasteriser's kernels read and hand over their small, heavily re-read tables.
// Round-1 finding: while an 8-wave split-bf16 convolution workgroup of ANOTHER stream is resident on the same
// CUs, raster_faces / raster_resolve built with plain table loads return different results from run to run (12 of 12 runs); with
// agent-scope (sc1) dword loads (variant 4 and up) never.  Rounds 2-4 read that as "wrong words from vector-L1-served gathers".  Round 5
// (profiles/r05_raster_coresidency_classified.txt; probe bit 64 dumps every gathered word and the number of z-buffer updates a lane issues):
// the GATHERED WORDS ARE RIGHT in the failing runs (0 wrong of ~46 M per run; RAS counters clean; never a bit flip) — what differs is how
// many atomicMin updates ~0.2 % of the lanes issue, so faces go missing from the z-buffer; and a build whose table loads are PLAIN,
// L1-served single-dword loads (bit 256) never fails either.  What every failing build shares is per-lane 96-bit gathers
// (global_load_dwordx3 on 12-byte records) in flight in these two kernels, not the cache that serves them; the mechanism below the
// instruction level is still unknown.  The shipped configuration keeps dword sc1 loads for every gathered table and write-through hand-overs
// (no 96-bit gather, nothing measurable in cost on a few hundred KB), and the co-residency test keeps it that way.
//   bit 0: z-buffer clear stores sc1   bit 1: vertex (tv) stores sc1   bit 2: vertex / index / uv-table loads sc1
//   bit 3: z-buffer loads in resolve sc1   bit 4: (probe) faces / resolve transform their vertices themselves from `verts`
#ifndef RASTER_VARIANT
#define RASTER_VARIANT 15
#endif
#if RASTER_VARIANT & 64
__device__ unsigned int* g_raster_dbg = nullptr;        // probe build only: raw gathered words of raster_resolve_kernel
__device__ unsigned int* g_raster_dbg2 = nullptr;       // ... and of raster_faces_kernel
extern "C" int n3d_raster_debug_buffer(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_raster_dbg), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
extern "C" int n3d_raster_debug_buffer_faces(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_raster_dbg2), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#endif
template <typename T> __device__ __forceinline__ T ld_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void st_agent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
//   bit 5 (probe only): plain single-dword vertex loads the compiler cannot merge into 12-byte dwordx3 loads
//   bit 8 of the high byte (256, probe only): EVERY table load as a plain single-dword load the compiler cannot merge (no global_load_dwordx3 anywhere,
//   still L1-served) — separates "96-bit loads" from "L1-served loads" as the property the failing builds share
template <typename T> __device__ __forceinline__ T ld_plain1(const T* p) {
    static_assert(sizeof(T) == 4, "dword tables");
    unsigned v; asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return __builtin_bit_cast(T, v);
}
template <typename T> __device__ __forceinline__ T ld_tab(const T* p) {
    if (RASTER_VARIANT & 4) return ld_agent(p);
    if constexpr (sizeof(T) == 4) { if (RASTER_VARIANT & 256) return ld_plain1(p); }
    return *p;
}
__device__ __forceinline__ float ld_vert(const float* p) {
    if (RASTER_VARIANT & 4) return ld_agent(p);
    if (RASTER_VARIANT & (32 | 256)) { float v; asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
    return *p;
}
struct XfParams { const float* verts; const float* rot; int V, views; float sx, sy, sz, scale; };
// one transformed vertex (the arithmetic of raster_transform_kernel, operation for operation)
__device__ __forceinline__ void xf_vertex(const XfParams& q, int nv, int vi, float& ox, float& oy, float& oz) {
    const int view = nv % q.views, n = nv / q.views;
    const float* p = q.verts + ((int64_t)n * q.V + vi) * 3;
    const float* R = q.rot + view * 9;
    const float x = p[0], y = -p[1], z = p[2];
    float tx = (x * R[0] + y * R[3] + z * R[6] + q.sx) * q.scale;
    float ty = (x * R[1] + y * R[4] + z * R[7] + q.sy) * q.scale;
    float tz = (x * R[2] + y * R[5] + z * R[8] + q.sz) * q.scale;
    ty = -ty; tz = -tz; tz = tz + 10.f;
    ox = -tx; oy = -ty; oz = tz;
}
__device__ __forceinline__ void get_vertex(const float* vn, const XfParams& q, int nv, int vi, float& x, float& y, float& z) {
    if (RASTER_VARIANT & 16) { xf_vertex(q, nv, vi, x, y, z); return; }
    const float* v = vn + 3 * (int64_t)vi;
    x = ld_vert(v); y = ld_vert(v + 1); z = ld_vert(v + 2);
}

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}
__device__ __forceinline__ float pix_to_ndc(int i, int S) { return -1.0f + (2.0f * (float)i + 1.0f) / (float)S; }

// verts [N,V,3]; rot [NV_VIEWS,3,3] (angle2matrix on the host); out tv [N*views, V, 3] in PyTorch3D NDC.
// lms [N,Lm,3] -> lm2d [N,Lm,2] for view 0 only (the mouth box uses the front view).
__global__ __launch_bounds__(256) void raster_transform_kernel(const float* __restrict__ verts, const float* __restrict__ lms,
                                                               const float* __restrict__ rot, float* __restrict__ tv,
                                                               float* __restrict__ lm2d, int N, int V, int Lm, int views,
                                                               float sx, float sy, float sz, float scale) {
    const int64_t total = (int64_t)N * views * V;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const int v = (int)(i % V), view = (int)((i / V) % views), n = (int)(i / ((int64_t)V * views));
        const float* p = verts + ((int64_t)n * V + v) * 3;
        const float* R = rot + view * 9;
        const float x = p[0], y = -p[1], z = p[2];
        float tx = (x * R[0] + y * R[3] + z * R[6] + sx) * scale;
        float ty = (x * R[1] + y * R[4] + z * R[7] + sy) * scale;
        float tz = (x * R[2] + y * R[5] + z * R[8] + sz) * scale;
        ty = -ty; tz = -tz; tz = tz + 10.f;
        float* o = tv + i * 3;                   // Pytorch3dRasterizer.forward negates x,y (renderer.py:403)
        if (RASTER_VARIANT & 2) { st_agent(o, -tx); st_agent(o + 1, -ty); st_agent(o + 2, tz); }
        else { o[0] = -tx; o[1] = -ty; o[2] = tz; }
    }
    const int64_t j = i - total;
    if (j >= 0 && j < (int64_t)N * Lm) {
        const float* p = lms + j * 3;
        const float* R = rot;
        const float x = p[0], y = -p[1], z = p[2];
        const float tx = (x * R[0] + y * R[3] + z * R[6] + sx) * scale;
        const float ty = (x * R[1] + y * R[4] + z * R[7] + sy) * scale;
        lm2d[j * 2 + 0] = tx; lm2d[j * 2 + 1] = -ty;
    }
}

// faces_bs: ints between the face tables of consecutive images (0: one shared table); cull: cull_backfaces
__global__ __launch_bounds__(256) void raster_faces_kernel(const float* tv, const int* __restrict__ faces_base, int64_t faces_bs, int cull,
                                                           unsigned long long* zbuf, int NV, int V, int F, int H, int W, XfParams xf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)NV * F) return;
    const int f = (int)(i % F), nv = (int)(i / F);
    const int* faces = faces_base + (int64_t)nv * faces_bs;
    const float* vn = tv + (int64_t)nv * V * 3;
    // the reference rasterises faces[..., [0,2,1]] (triplane_next3d.py:207): `faces` is passed already swapped
    float x0, y0, z0, x1, y1, z1, x2, y2, z2;
    const int i0 = ld_tab(faces + 3 * f + 0), i1 = ld_tab(faces + 3 * f + 1), i2 = ld_tab(faces + 3 * f + 2);
    get_vertex(vn, xf, nv, i0, x0, y0, z0);
    get_vertex(vn, xf, nv, i1, x1, y1, z1);
    get_vertex(vn, xf, nv, i2, x2, y2, z2);
#if RASTER_VARIANT & 64
    if (g_raster_dbg2) {                               // probe build only: the RAW words this lane gathered for its face, 12 per (view, face)
        unsigned int* o = g_raster_dbg2 + i * 12;
        o[0] = (unsigned)i0; o[1] = (unsigned)i1; o[2] = (unsigned)i2;
        const float w[9] = {x0, y0, z0, x1, y1, z1, x2, y2, z2};
        for (int k = 0; k < 9; ++k) o[3 + k] = __float_as_uint(w[k]);
    }
#endif
    const float zmax = fmaxf(z0, fmaxf(z1, z2)), zmin = fminf(z0, fminf(z1, z2));
    const float face_area = edge_fn(x0, y0, x1, y1, x2, y2);
    const bool zero_area = (face_area <= K_EPS) && (face_area >= -K_EPS);
    // zmin < kEpsilon: PyTorch3D's CheckPointOutsideBoundingBox ("z_invalid") puts every pixel outside the box of a face with a vertex at / behind the camera plane
    if (zmax < 0.0f || (cull && face_area < 0.0f) || zero_area || zmin < K_EPS) return;      // cull_backfaces (renderer.py:397: True)
    const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
    const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
    if (!(xmax >= -2.f && xmin <= 2.f && ymax >= -2.f && ymin <= 2.f)) return;   // far off-screen / NaN
    int xi_lo = (int)floorf((float)(W - 1) - ((xmax + 1.0f) * (float)W - 1.0f) * 0.5f) - 1;
    int xi_hi = (int)ceilf((float)(W - 1) - ((xmin + 1.0f) * (float)W - 1.0f) * 0.5f) + 1;
    int yi_lo = (int)floorf((float)(H - 1) - ((ymax + 1.0f) * (float)H - 1.0f) * 0.5f) - 1;
    int yi_hi = (int)ceilf((float)(H - 1) - ((ymin + 1.0f) * (float)H - 1.0f) * 0.5f) + 1;
    xi_lo = max(xi_lo, 0); yi_lo = max(yi_lo, 0); xi_hi = min(xi_hi, W - 1); yi_hi = min(yi_hi, H - 1);
    const float area = edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS;
#if RASTER_VARIANT & 64
    unsigned int n_atomics = 0, key_xor = 0;           // probe build: how many z-buffer updates this lane ISSUED, and a checksum of their keys
#endif
    for (int yi = yi_lo; yi <= yi_hi; ++yi) {
        const float yf = pix_to_ndc(H - 1 - yi, H);
        for (int xi = xi_lo; xi <= xi_hi; ++xi) {
            const float xf = pix_to_ndc(W - 1 - xi, W);
            if (xf > xmax || xf < xmin || yf > ymax || yf < ymin) continue;
            const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / area;
            const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / area;
            const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / area;
            const float pz = w0 * z0 + w1 * z1 + w2 * z2;
            if (pz < 0.0f) continue;
            if (!((w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f))) continue;
            const unsigned long long key = ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned)f;
            atomicMin(&zbuf[((int64_t)nv * H + yi) * W + xi], key);
#if RASTER_VARIANT & 64
            ++n_atomics; key_xor ^= __float_as_uint(pz) * 2654435761u + (unsigned)(yi * W + xi);
#endif
        }
    }
#if RASTER_VARIANT & 64
    if (g_raster_dbg2) { g_raster_dbg2[(int64_t)NV * F * 12 + i * 2] = n_atomics; g_raster_dbg2[(int64_t)NV * F * 12 + i * 2 + 1] = key_xor; }
#endif
}

__device__ __forceinline__ float bilinear_1ch(const float* __restrict__ img, int H, int W, float gx, float gy) {
    // grid_sampler_2d, bilinear / zeros / align_corners=False, accumulation order nw, ne, sw, se
    const float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f, iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    if (!(fx0 > -2.f && fx0 < (float)W + 1.f && fy0 > -2.f && fy0 < (float)H + 1.f)) return 0.f;
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
    float acc = 0.f;
    if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) acc += ld_tab(img + (int64_t)y0 * W + x0) * (wx0 * wy0);
    if (x0 + 1 >= 0 && x0 + 1 < W && y0 >= 0 && y0 < H) acc += ld_tab(img + (int64_t)y0 * W + x0 + 1) * (wx1 * wy0);
    if (x0 >= 0 && x0 < W && y0 + 1 >= 0 && y0 + 1 < H) acc += ld_tab(img + (int64_t)(y0 + 1) * W + x0) * (wx0 * wy1);
    if (x0 + 1 >= 0 && x0 + 1 < W && y0 + 1 >= 0 && y0 + 1 < H) acc += ld_tab(img + (int64_t)(y0 + 1) * W + x0 + 1) * (wx1 * wy1);
    return acc;
}

struct BilinearTaps { int off[4]; float w[4]; bool ok[4]; };
__device__ __forceinline__ BilinearTaps bilinear_setup(int H, int W, float gx, float gy) {
    BilinearTaps t;
    const float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f, iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const bool sane = fx0 > -2.f && fx0 < (float)W + 1.f && fy0 > -2.f && fy0 < (float)H + 1.f;
    const int x0 = sane ? (int)fx0 : -4, y0 = sane ? (int)fy0 : -4;
    const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
    const float ww[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
        t.ok[k] = xx >= 0 && xx < W && yy >= 0 && yy < H;
        t.off[k] = t.ok[k] ? yy * W + xx : 0;
        t.w[k] = ww[k];
    }
    return t;
}
__device__ __forceinline__ float bilinear_apply(const BilinearTaps& t, const float* __restrict__ img) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (t.ok[k]) acc += img[t.off[k]] * t.w[k];
    return acc;
}

// The texture projection gathers single floats through per-lane addresses — the access class whose vector-L1-served form returned wrong
// values in the rasteriser's kernels while 8-wave MFMA workgroups shared the CU.  TEXPROJ_AGENT = 1 (default): the same
// agent-scope loads here, so that this kernel, too, is safe by construction and not only by soak test (A/B: tools/build_variant.sh
// noagent raster.hip -DTEXPROJ_AGENT=0 -ffp-contract=off; profiles/r04_texproj_agent_ab.txt).
#ifndef TEXPROJ_AGENT
#define TEXPROJ_AGENT 1
#endif
__device__ __forceinline__ float bilinear_apply_tex(const BilinearTaps& t, const float* img) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (t.ok[k]) acc += (TEXPROJ_AGENT ? ld_agent(img + t.off[k]) : img[t.off[k]]) * t.w[k];
    return acc;
}

// per pixel: uv = sum_k bary_k * face_uv[f][k], vis; alpha = grid_sample(uv_face_mask, uv) * vis   (renderer.py:425-437,
// triplane_next3d.py:211-214).  grid [NV,H,W,2], alpha [NV,H,W]
__global__ __launch_bounds__(256) void raster_resolve_kernel(const float* tv, const int* __restrict__ faces,
                                                             const float* __restrict__ face_uv,   // [F,3,3] (vertex order swapped)
                                                             const unsigned long long* zbuf,
                                                             const float* __restrict__ uv_mask, int MH, int MW,
                                                             float* __restrict__ grid, float* __restrict__ alpha, int NV, int V,
                                                             int F, int H, int W, XfParams xfp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)NV * H * W) return;
    const int xi = (int)(i % W), yi = (int)((i / W) % H), nv = (int)(i / ((int64_t)H * W));
    const unsigned long long key = (RASTER_VARIANT & 8) ? ld_agent(zbuf + i) : zbuf[i];
    float u = 0.f, v = 0.f, vis = 0.f;
    if (key != 0xFFFFFFFFFFFFFFFFull) {
        const int f = (int)(key & 0xFFFFFFFFull);
        const float* vn = tv + (int64_t)nv * V * 3;
        float ax, ay, az, bx, by, bz, cx, cy, cz;
        const int i0 = ld_tab(faces + 3 * f + 0), i1 = ld_tab(faces + 3 * f + 1), i2 = ld_tab(faces + 3 * f + 2);
        get_vertex(vn, xfp, nv, i0, ax, ay, az);
        get_vertex(vn, xfp, nv, i1, bx, by, bz);
        get_vertex(vn, xfp, nv, i2, cx, cy, cz);
        const float xf = pix_to_ndc(W - 1 - xi, W), yf = pix_to_ndc(H - 1 - yi, H);
        const float area = edge_fn(cx, cy, ax, ay, bx, by) + K_EPS;
        const float w0 = edge_fn(xf, yf, bx, by, cx, cy) / area;
        const float w1 = edge_fn(xf, yf, cx, cy, ax, ay) / area;
        const float w2 = edge_fn(xf, yf, ax, ay, bx, by) / area;
        const float* a = face_uv + (int64_t)f * 9;
        const float u0 = ld_tab(a + 0), u1 = ld_tab(a + 3), u2 = ld_tab(a + 6), v0 = ld_tab(a + 1), v1 = ld_tab(a + 4), v2 = ld_tab(a + 7);
        u = (w0 * u0 + w1 * u1) + w2 * u2;
        v = (w0 * v0 + w1 * v1) + w2 * v2;
        vis = 1.f;
#if RASTER_VARIANT & 64
        // probe build only (tools/probe/raster_coresidency_repro.cpp, RASTER_CLASSIFY): the RAW words this lane gathered, 18 per pixel
        if (g_raster_dbg) {
            unsigned int* o = g_raster_dbg + i * 18;
            o[0] = (unsigned)i0; o[1] = (unsigned)i1; o[2] = (unsigned)i2;
            const float w[15] = {ax, ay, az, bx, by, bz, cx, cy, cz, u0, v0, u1, v1, u2, v2};
            for (int k = 0; k < 15; ++k) o[3 + k] = __float_as_uint(w[k]);
        }
#endif
    }
    grid[i * 2 + 0] = u; grid[i * 2 + 1] = v;
    alpha[i] = bilinear_1ch(uv_mask, MH, MW, u, v) * vis;
}

// fill_mouth on the device: flood from (0,0) through pixels with seed <= a*255 <= seed + 254 (4-connected), then
// res = clip(a + ((m*2-1)*-1+1)/2, 0, 1) with m = filled ? 255/127.5-1 : a*255/127.5-1   (renderer.py:583-602)
// Bit-parallel: thread t owns row t AND column t of the 256 x 256 image as 256-bit masks (4 x u64) in registers.  Flooding
// along a line inside runs of passable pixels is O(1) — seeds added to the mask ripple a carry to the end of their run,
// ((m + x) ^ m) & m | x, the other direction on the bit-reversed words — so one iteration = flood all rows, transpose
// (256 ballots), flood all columns, transpose back; the number of iterations is the number of direction changes of the
// longest path (a handful), not its length.  264 us (per-pixel LDS sweeps) -> ~50 us for 16 images.
#define FH_WORDS 4          // 256 columns / rows
__device__ __forceinline__ void flood_up(const unsigned long long (&m)[FH_WORDS], unsigned long long (&x)[FH_WORDS]) {
    unsigned long long carry = 0;
#pragma unroll
    for (int w = 0; w < FH_WORDS; ++w) {
        const unsigned long long xs = x[w] & m[w];
        const unsigned long long s1 = m[w] + xs;
        const unsigned long long c1 = s1 < xs ? 1ull : 0ull;
        const unsigned long long s2 = s1 + carry;
        const unsigned long long c2 = s2 < s1 ? 1ull : 0ull;
        x[w] = ((s2 ^ m[w]) & m[w]) | xs;
        carry = c1 | c2;
    }
}
__device__ __forceinline__ void rev256(unsigned long long (&v)[FH_WORDS]) {
    const unsigned long long a = __brevll(v[0]), b = __brevll(v[1]), c = __brevll(v[2]), d = __brevll(v[3]);
    v[0] = d; v[1] = c; v[2] = b; v[3] = a;
}
// flood `x` inside the runs of `m` in both directions (rm = bit-reversed m)
__device__ __forceinline__ void flood_line(const unsigned long long (&m)[FH_WORDS], const unsigned long long (&rm)[FH_WORDS],
                                           unsigned long long (&x)[FH_WORDS]) {
    flood_up(m, x);
    rev256(x);
    flood_up(rm, x);
    rev256(x);
}
// 64 x 64 bit-block transpose inside a wave (lane = row, bit = column): six butterfly rounds of a xor-shuffle
__device__ __forceinline__ unsigned long long transpose64(unsigned long long x, int lane) {
    unsigned long long m = 0x00000000FFFFFFFFull;
#pragma unroll
    for (int j = 32; j != 0; j >>= 1) {
        const unsigned int lo = __shfl_xor((unsigned int)x, j, 64), hi = __shfl_xor((unsigned int)(x >> 32), j, 64);
        const unsigned long long y = ((unsigned long long)hi << 32) | lo;
        x = (lane & j) ? ((x & ~m) | ((y >> j) & m)) : ((x & m) | ((y & m) << j));
        m ^= m << (j >> 1);
    }
    return x;
}
// 256 x 256 bit transpose across the block: thread t holds line t (`in`), afterwards `out` = the t-th cross line.
// Wave wr transposes its four 64 x 64 blocks (wr, wc) in registers and hands block (wc, wr) over through LDS.
__device__ __forceinline__ void transpose256(const unsigned long long (&in)[FH_WORDS], unsigned long long (&out)[FH_WORDS],
                                             unsigned long long (*s_t)[FH_WORDS], int t, bool active) {
    const int lane = t & 63, wr = t >> 6;
    if (active) {
#pragma unroll
        for (int wc = 0; wc < FH_WORDS; ++wc) s_t[wc * 64 + lane][wr] = transpose64(in[wc], lane);
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int w = 0; w < FH_WORDS; ++w) out[w] = s_t[t][w];
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void fill_holes_kernel(float* __restrict__ alpha, int H, int W, int binarize_view, int views) {
    __shared__ unsigned long long s_t[256][FH_WORDS];
    const int img = blockIdx.x;
    float* a = alpha + (int64_t)img * H * W;
    const int tid = threadIdx.x, t = tid & 255, q = tid >> 8;  // 1024 threads share the per-pixel phases (4 rows at a time);
    const float seed = a[0] * 255.f;                           // threads 0-255 (4 waves) run the flood itself
    const float vmin = seed - 0.f, vmax = seed + 254.f;
    // passable masks: coalesced row reads, one ballot per wave = one 64-bit word of a row
    for (int y0 = 0; y0 < 256; y0 += 32) {                     // 8 independent loads in flight per thread
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int y = y0 + 4 * j + q; v[j] = (y < H && t < W) ? a[(int64_t)y * W + t] * 255.f : -1.f; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int y = y0 + 4 * j + q;
            const unsigned long long word = __ballot(y < H && t < W && v[j] >= vmin && v[j] <= vmax);
            if ((t & 63) == 0) s_t[y][t >> 6] = word;
        }
    }
    __syncthreads();
    const bool fl = tid < 256;                                 // the flood group; the other waves only keep the barriers company
    unsigned long long pr[FH_WORDS] = {0, 0, 0, 0}, pc[FH_WORDS] = {0, 0, 0, 0}, rpr[FH_WORDS], rpc[FH_WORDS];   // passable: row t, column t
    if (fl) {
#pragma unroll
        for (int w = 0; w < FH_WORDS; ++w) pr[w] = s_t[t][w];
        if (t == 0) pr[0] |= 1ull;                             // the seed pixel is always filled
    }
    __syncthreads();
    transpose256(pr, pc, s_t, t, fl);
#pragma unroll
    for (int w = 0; w < FH_WORDS; ++w) { rpr[w] = pr[w]; rpc[w] = pc[w]; }
    rev256(rpr); rev256(rpc);
    unsigned long long rr[FH_WORDS] = {(fl && t == 0) ? 1ull : 0ull, 0, 0, 0};           // reached: row t
    for (int iter = 0; iter < 1024; ++iter) {
        unsigned long long rc[FH_WORDS] = {0, 0, 0, 0}, back[FH_WORDS] = {0, 0, 0, 0};
        if (fl) flood_line(pr, rpr, rr);
        transpose256(rr, rc, s_t, t, fl);
        if (fl) flood_line(pc, rpc, rc);
        transpose256(rc, back, s_t, t, fl);
        bool changed = false;
#pragma unroll
        for (int w = 0; w < FH_WORDS; ++w) { changed |= (back[w] & ~rr[w]) != 0; rr[w] |= back[w]; }
        if (!__syncthreads_or(fl && changed)) break;
    }
    if (fl) {
#pragma unroll
        for (int w = 0; w < FH_WORDS; ++w) s_t[t][w] = rr[w];
    }
    __syncthreads();
    const bool binarize = (img % views) == binarize_view;     // alpha_side = alpha[1].bool() | alpha[1].bool()
    for (int e = tid; e < H * W; e += 1024) {
        const int y = e / W, x = e % W;
        const float av = a[e];
        const bool filled = (s_t[y][x >> 6] >> (x & 63)) & 1ull;
        const float ci = filled ? 255.f : av * 255.f;
        const float m = ci / 127.5f - 1.f;
        const float mm = ((m * 2.f - 1.f) * -1.f + 1.f) / 2.f;
        float r = fminf(fmaxf(av + mm, 0.f), 1.f);
        if (binarize) r = (r != 0.f) ? 1.f : 0.f;
        a[e] = r;
    }
}

// the four taps of a pixel as two 8-byte pairs (row 0: taps 0 / 1, row 1: taps 2 / 3): byte offset of the pair inside a channel plane and which half holds which tap
struct TexPairs { int boff[2]; bool first_in_y[2], second_in_x[2]; };
__device__ __forceinline__ TexPairs tex_pairs(const BilinearTaps& t) {
    TexPairs q;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const bool ok0 = t.ok[2 * r], ok1 = t.ok[2 * r + 1];
        // both inside: the pair starts at tap 0.  Only tap 0 inside (x = TW - 1): the pair (x - 1, x), tap 0 in its second half.  Only tap 1 inside (x = 0): the pair (0, 1), tap 1
        // in its first half.  Neither: any legal pair (its values are not used).
        const int first = ok0 ? (ok1 ? t.off[2 * r] : t.off[2 * r] - 1) : (ok1 ? t.off[2 * r + 1] : 0);
        q.boff[r] = first * 4;
        q.first_in_y[r] = ok0 && !ok1;
        q.second_in_x[r] = !ok0 && ok1;
    }
    return q;
}
__device__ __forceinline__ float tex_pairs_apply(__amdgpu_buffer_rsrc_t r_tex, const TexPairs& q, const BilinearTaps& t, int plane_off) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const f32x2v pr = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(r_tex, q.boff[r] + plane_off, 0, TEXPROJ_AGENT ? 16 : 0));
        const float v0 = q.first_in_y[r] ? pr.y : pr.x, v1 = q.second_in_x[r] ? pr.x : pr.y;
        if (t.ok[2 * r]) acc += v0 * t.w[2 * r];
        if (t.ok[2 * r + 1]) acc += v1 * t.w[2 * r + 1];
    }
    return acc;
}

// out_plane[n][c][y][x] = sum over the plane's views of grid_sample(textures[n][c], grid[n*views+view][y][x])
// `planes` > 1: the caller's plane list (view_a[k], view_b[k], out[k]) is processed in ONE launch (blockIdx.y = plane).
struct TexProjPlanes { float* out[4]; int view_a[4]; int view_b[4]; };
__global__ __launch_bounds__(256) void texture_project_kernel(const float* __restrict__ tex, const float* __restrict__ grid,
                                                              TexProjPlanes pl, int N, int C, int TH, int TW, int H, int W, int views) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * H * W) return;
    const int k = blockIdx.y, view_a = pl.view_a[k], view_b = pl.view_b[k];
    const int pix = (int)(i % ((int64_t)H * W)), n = (int)(i / ((int64_t)H * W));
    const float* tn = tex + (int64_t)n * C * TH * TW;
    float* on = pl.out[k] + (int64_t)n * C * H * W + pix;
    const float* ga = grid + (((int64_t)n * views + view_a) * H * W + pix) * 2;
    const float ua = ga[0], va = ga[1];
    float ub = 0.f, vb = 0.f;
    if (view_b >= 0) {
        const float* gb = grid + (((int64_t)n * views + view_b) * H * W + pix) * 2;
        ub = gb[0]; vb = gb[1];
    }
    // tap offsets and weights once per pixel (bilinear_1ch's arithmetic, same accumulation order nw, ne, sw, se)
    BilinearTaps ta = bilinear_setup(TH, TW, ua, va), tb = bilinear_setup(TH, TW, ub, vb);
    if (TW < 2 || (int64_t)C * TH * TW * 4 >= (1ll << 31)) {               // (degenerate widths / planes beyond 32-bit buffer offsets: one load per tap)
        for (int c = 0; c < C; ++c) {
            const float* tc = tn + (int64_t)c * TH * TW;
            float v = bilinear_apply_tex(ta, tc);
            if (view_b >= 0) v = v + bilinear_apply_tex(tb, tc);
            on[(int64_t)c * H * W] = v;
        }
        return;
    }
    // Round 6: the two x-adjacent taps of a row are 8 contiguous bytes: ONE 8-byte load per tap row (buffer loads need dword alignment only) instead of two dword gathers —
    // the kernel is bound by the texture path's per-lane address rate (64 scattered addresses per load instruction), and this halves the instructions: 138 -> 93 us per
    // 3-plane batch-4 call (profiles/r06_texproj_pairs_ab.txt).  A row with one tap outside the texture loads the pair that holds its inside tap; same values, same sums in the same order.
    const __amdgpu_buffer_rsrc_t r_tex = __builtin_amdgcn_make_buffer_rsrc((void*)tn, 0, C * TH * TW * 4, 0x00020000);
    TexPairs pa = tex_pairs(ta), pb = tex_pairs(tb);
    const int plane_bytes = TH * TW * 4;
    for (int c = 0; c < C; ++c) {
        float v = tex_pairs_apply(r_tex, pa, ta, c * plane_bytes);
        if (view_b >= 0) v = v + tex_pairs_apply(r_tex, pb, tb, c * plane_bytes);
        on[(int64_t)c * H * W] = v;
    }
}

// gen_mouth_mask (triplane_next3d.py:330-344): bbox[n] = (y0, y1, x0, x1)
__global__ void mouth_bbox_kernel(const float* __restrict__ lm2d, int* __restrict__ bbox, int N, int Lm) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* lm = lm2d + (int64_t)n * Lm * 2;
    float c0min = INFINITY, c0max = -INFINITY, c1min = INFINITY, c1max = -INFINITY;
    for (int k = 48; k < 60; ++k) {
        const float a = lm[k * 2 + 0] * 128.f + 128.f, b = lm[k * 2 + 1] * 128.f + 128.f;
        c0min = fminf(c0min, a); c0max = fmaxf(c0max, a); c1min = fminf(c1min, b); c1max = fmaxf(c1max, b);
    }
    const float l0 = lm[48 * 2 + 0] * 128.f + 128.f, l1 = lm[48 * 2 + 1] * 128.f + 128.f;
    const float r0 = lm[54 * 2 + 0] * 128.f + 128.f, r1 = lm[54 * 2 + 1] * 128.f + 128.f;
    const float avg0 = (l0 + r0) * 0.5f, avg1 = (l1 + r1) * 0.5f;
    const float ext = fmaxf(c0max - c0min, c1max - c1min) * 1.2f;
    const int res = (int)ext;                    // astype(int): truncation
    const int h = res >= 0 ? res / 2 : -((-res + 1) / 2);     // python floor division
    bbox[n * 4 + 0] = (int)((double)avg1 - (double)h);
    bbox[n * 4 + 1] = (int)((double)avg1 + (double)h);
    bbox[n * 4 + 2] = (int)((double)avg0 - (double)h);
    bbox[n * 4 + 3] = (int)((double)avg0 + (double)h);
}

// antialiased bilinear resize with optional per-sample boxes (y0, y1, x0, x1) on source and/or destination.
struct ResizeParams {
    const float* src; float* dst;
    const int* src_box; const int* dst_box;      // [N,4] device ints or NULL (= full tensor)
    int N, C, SH, SW, DH, DW;
    int dst_square;                              // paste mode: destination box is (y0, y0+s, x0, x0+s), s = y1-y0
    int64_t src_bs;                              // floats between consecutive samples of src (C * SH * SW when dense): a channel-slice view needs no copy
};

__device__ __forceinline__ float tri_filter(float x) { x = fabsf(x); return x < 1.f ? 1.f - x : 0.f; }

__device__ __forceinline__ void aa_range(int i, int in_size, int out_size, int& xmin, int& xsize, float& center, float& invscale,
                                         float& total) {
    const float scale = (float)in_size / (float)out_size;
    const float support = scale >= 1.f ? scale : 1.f;
    invscale = scale >= 1.f ? 1.f / scale : 1.f;
    center = scale * ((float)i + 0.5f);
    xmin = max((int)(center - support + 0.5f), 0);
    xsize = min((int)(center + support + 0.5f), in_size) - xmin;
    total = 0.f;
    for (int j = 0; j < xsize; ++j) total += tri_filter(((float)(j + xmin) - center + 0.5f) * invscale);
}

__global__ __launch_bounds__(256) void resize_aa_kernel(ResizeParams p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.N * p.C * p.DH * p.DW) return;
    const int dx = (int)(i % p.DW), dy = (int)((i / p.DW) % p.DH);
    const int c = (int)((i / ((int64_t)p.DW * p.DH)) % p.C), n = (int)(i / ((int64_t)p.DW * p.DH * p.C));
    int sy0 = 0, sx0 = 0, sh = p.SH, sw = p.SW, oy0 = 0, ox0 = 0, oh = p.DH, ow = p.DW;
    if (p.src_box) {
        const int* b = p.src_box + n * 4;
        sy0 = b[0]; sx0 = b[2]; sh = b[1] - b[0]; sw = b[3] - b[2];
    }
    if (p.dst_box) {
        const int* b = p.dst_box + n * 4;
        oy0 = b[0]; ox0 = b[2]; oh = b[1] - b[0]; ow = p.dst_square ? oh : b[3] - b[2];
    }
    const int ry = dy - oy0, rx = dx - ox0;
    if (ry < 0 || ry >= oh || rx < 0 || rx >= ow || sh <= 0 || sw <= 0) return;
    // clamp the source box to the tensor (python slicing semantics)
    const int cy0 = max(sy0, 0), cx0 = max(sx0, 0);
    sh = min(sy0 + sh, p.SH) - cy0; sw = min(sx0 + sw, p.SW) - cx0;
    if (sh <= 0 || sw <= 0) return;
    int ymin, ysize, xmin, xsize;
    float cyc, cxc, yinv, xinv, ytot, xtot;
    aa_range(ry, sh, oh, ymin, ysize, cyc, yinv, ytot);
    aa_range(rx, sw, ow, xmin, xsize, cxc, xinv, xtot);
    const float* s = p.src + (int64_t)n * p.src_bs + (int64_t)c * p.SH * p.SW;
    float acc = 0.f;
    constexpr int MAXT = 16;
    if (xsize <= MAXT) {            // the horizontal weights do not depend on the row: compute (and divide) them once
        float wx[MAXT];
#pragma unroll
        for (int jx = 0; jx < MAXT; ++jx) wx[jx] = jx < xsize ? tri_filter(((float)(jx + xmin) - cxc + 0.5f) * xinv) / xtot : 0.f;
        for (int jy = 0; jy < ysize; ++jy) {
            const float wy = tri_filter(((float)(jy + ymin) - cyc + 0.5f) * yinv) / ytot;
            const float* row = s + (int64_t)(cy0 + ymin + jy) * p.SW + cx0 + xmin;
            float h = 0.f;
#pragma unroll
            for (int jx = 0; jx < MAXT; ++jx) if (jx < xsize) h += row[jx] * wx[jx];
            acc += h * wy;
        }
    } else {
        for (int jy = 0; jy < ysize; ++jy) {
            const float wy = tri_filter(((float)(jy + ymin) - cyc + 0.5f) * yinv) / ytot;
            const float* row = s + (int64_t)(cy0 + ymin + jy) * p.SW + cx0 + xmin;
            float h = 0.f;
            for (int jx = 0; jx < xsize; ++jx) h += row[jx] * (tri_filter(((float)(jx + xmin) - cxc + 0.5f) * xinv) / xtot);
            acc += h * wy;
        }
    }
    p.dst[((int64_t)n * p.C + c) * p.DH * p.DW + (int64_t)dy * p.DW + dx] = acc;
}

// The same resize with a DESTINATION box (the paste of the mouth plane back into the front plane: 256^2 -> an s x s box, s ~ 40, known on the device only).  resize_aa_kernel
// spends a thread on every pixel of the destination tensor (8.4 M for batch 4, 2 % of them inside the boxes, 20 active lanes per wave) and lets each of them derive its own
// 2 x ~13 normalised weights (26 divisions): 109 us per call.  Here a workgroup owns a band of box rows of one (sample, channel), derives the weights of its rows and of all box
// columns ONCE into LDS, and its 256 threads walk the band's pixels: the same weights (same expressions), the same sums in the same order — identical results.
// grid (RS_BANDS, N * C).  Falls back to the per-pixel form (weights on the fly) when the box's weight tables would not fit (RS_TAPS floats per axis).
constexpr int RS_BANDS = 4, RS_TAPS = 3072, RS_DIM = 512;
__global__ __launch_bounds__(256) void resize_aa_paste_kernel(ResizeParams p) {
    __shared__ float s_wx[RS_TAPS], s_wy[RS_TAPS];
    __shared__ int s_x0[RS_DIM], s_xn[RS_DIM], s_y0[RS_DIM], s_yn[RS_DIM];
    const int tid = threadIdx.x, band = blockIdx.x, n = blockIdx.y / p.C, c = blockIdx.y % p.C;
    const int* b = p.dst_box + n * 4;
    const int oy0 = b[0], ox0 = b[2], oh = b[1] - b[0], ow = p.dst_square ? oh : b[3] - b[2];
    int sy0 = 0, sx0 = 0, sh = p.SH, sw = p.SW;
    if (p.src_box) { const int* sb = p.src_box + n * 4; sy0 = sb[0]; sx0 = sb[2]; sh = sb[1] - sb[0]; sw = sb[3] - sb[2]; }
    if (oh <= 0 || ow <= 0 || sh <= 0 || sw <= 0) return;
    const int cy0 = max(sy0, 0), cx0 = max(sx0, 0);                       // clamp the source box to the tensor (python slicing semantics)
    sh = min(sy0 + sh, p.SH) - cy0; sw = min(sx0 + sw, p.SW) - cx0;
    if (sh <= 0 || sw <= 0) return;
    const int r0 = (int)((int64_t)oh * band / RS_BANDS), r1 = (int)((int64_t)oh * (band + 1) / RS_BANDS);      // this workgroup's box rows
    if (r1 <= r0) return;
    const float* src = p.src + (int64_t)n * p.src_bs + (int64_t)c * p.SH * p.SW;
    float* dst = p.dst + ((int64_t)n * p.C + c) * p.DH * p.DW;
    // support of an axis: 2 * max(scale, 1) + 1 taps per pixel at most
    const float scx = (float)sw / (float)ow, scy = (float)sh / (float)oh;
    const int tx_max = (int)(2.f * fmaxf(scx, 1.f)) + 2, ty_max = (int)(2.f * fmaxf(scy, 1.f)) + 2;
    const bool tables = ow <= RS_DIM && (r1 - r0) <= RS_DIM && (int64_t)ow * tx_max <= RS_TAPS && (int64_t)(r1 - r0) * ty_max <= RS_TAPS;
    if (tables) {
        for (int rx = tid; rx < ow; rx += 256) {
            int xmin, xsize; float cxc, xinv, xtot;
            aa_range(rx, sw, ow, xmin, xsize, cxc, xinv, xtot);
            s_x0[rx] = xmin; s_xn[rx] = xsize;
            for (int jx = 0; jx < xsize; ++jx) s_wx[rx * tx_max + jx] = tri_filter(((float)(jx + xmin) - cxc + 0.5f) * xinv) / xtot;
        }
        for (int ry = r0 + tid; ry < r1; ry += 256) {
            int ymin, ysize; float cyc, yinv, ytot;
            aa_range(ry, sh, oh, ymin, ysize, cyc, yinv, ytot);
            s_y0[ry - r0] = ymin; s_yn[ry - r0] = ysize;
            for (int jy = 0; jy < ysize; ++jy) s_wy[(ry - r0) * ty_max + jy] = tri_filter(((float)(jy + ymin) - cyc + 0.5f) * yinv) / ytot;
        }
        __syncthreads();
    }
    for (int e = tid; e < (r1 - r0) * ow; e += 256) {
        const int ry = r0 + e / ow, rx = e % ow, dy = oy0 + ry, dx = ox0 + rx;
        if (dy < 0 || dy >= p.DH || dx < 0 || dx >= p.DW) continue;       // the part of the box outside the tensor is not written
        float acc = 0.f;
        if (tables) {
            const int xmin = s_x0[rx], xsize = s_xn[rx], ymin = s_y0[ry - r0], ysize = s_yn[ry - r0];
            const float* wx = s_wx + rx * tx_max, *wy = s_wy + (ry - r0) * ty_max;
            for (int jy = 0; jy < ysize; ++jy) {
                const float* row = src + (int64_t)(cy0 + ymin + jy) * p.SW + cx0 + xmin;
                float h = 0.f;
                for (int jx = 0; jx < xsize; ++jx) h += row[jx] * wx[jx];
                acc += h * wy[jy];
            }
        } else {
            int ymin, ysize, xmin, xsize;
            float cyc, cxc, yinv, xinv, ytot, xtot;
            aa_range(ry, sh, oh, ymin, ysize, cyc, yinv, ytot);
            aa_range(rx, sw, ow, xmin, xsize, cxc, xinv, xtot);
            for (int jy = 0; jy < ysize; ++jy) {
                const float wyv = tri_filter(((float)(jy + ymin) - cyc + 0.5f) * yinv) / ytot;
                const float* row = src + (int64_t)(cy0 + ymin + jy) * p.SW + cx0 + xmin;
                float h = 0.f;
                for (int jx = 0; jx < xsize; ++jx) h += row[jx] * (tri_filter(((float)(jx + xmin) - cxc + 0.5f) * xinv) / xtot);
                acc += h * wyv;
            }
        }
        dst[(int64_t)dy * p.DW + dx] = acc;
    }
}

__global__ __launch_bounds__(256) void raster_clear_kernel(unsigned long long* __restrict__ zbuf, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if (RASTER_VARIANT & 1) st_agent(zbuf + i, 0xFFFFFFFFFFFFFFFFull);
    else zbuf[i] = 0xFFFFFFFFFFFFFFFFull;
}

// pytorch3d.renderer.mesh.rasterize_meshes' outputs from the z-buffer (the B1 shim next3d_amd/shims/pytorch3d: the reference's own
// Pytorch3dRasterizer.forward, vr/renderer.py:401-440, consumes them): pix_to_face = PACKED index n * F + f (-1 empty), zbuf = the
// interpolated depth (-1 empty), bary [.,3] (-1 empty) — barycentrics recomputed with the operation order of raster_faces_kernel.
__global__ __launch_bounds__(256) void raster_meshes_resolve_kernel(const float* tv, const int* __restrict__ faces_base, int64_t faces_bs,
                                                                    const unsigned long long* zbuf, long long* __restrict__ pix_to_face,
                                                                    float* __restrict__ zout, float* __restrict__ bary, int N, int V, int F, int H, int W) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * H * W) return;
    const int xi = (int)(i % W), yi = (int)((i / W) % H), n = (int)(i / ((int64_t)H * W));
    const unsigned long long key = ld_agent(zbuf + i);
    long long pf = -1; float z = -1.f, b0 = -1.f, b1 = -1.f, b2 = -1.f;
    if (key != 0xFFFFFFFFFFFFFFFFull) {
        const int f = (int)(key & 0xFFFFFFFFull);
        const int* faces = faces_base + (int64_t)n * faces_bs;
        const float* vn = tv + (int64_t)n * V * 3;
        const float* a = vn + 3 * (int64_t)ld_tab(faces + 3 * f + 0), *b = vn + 3 * (int64_t)ld_tab(faces + 3 * f + 1), *c = vn + 3 * (int64_t)ld_tab(faces + 3 * f + 2);
        const float ax = ld_vert(a), ay = ld_vert(a + 1), bx = ld_vert(b), by = ld_vert(b + 1), cx = ld_vert(c), cy = ld_vert(c + 1);
        const float xf = pix_to_ndc(W - 1 - xi, W), yf = pix_to_ndc(H - 1 - yi, H);
        const float area = edge_fn(cx, cy, ax, ay, bx, by) + K_EPS;
        b0 = edge_fn(xf, yf, bx, by, cx, cy) / area;
        b1 = edge_fn(xf, yf, cx, cy, ax, ay) / area;
        b2 = edge_fn(xf, yf, ax, ay, bx, by) / area;
        z = __uint_as_float((unsigned)(key >> 32));
        pf = (long long)n * F + f;
    }
    pix_to_face[i] = pf; zout[i] = z;
    bary[i * 3 + 0] = b0; bary[i * 3 + 1] = b1; bary[i * 3 + 2] = b2;
}

// cv2.floodFill(img, mask, seed = (0, 0), new_val, lo, up, FLOODFILL_FIXED_RANGE) on float32 images of up to 256 x 256, in place
// (the B1 shim next3d_amd/shims/cv2; the reference's call: vr/renderer.py:593): the bit-parallel flood of fill_holes_kernel with the
// plain cv2 write-back — every pixel 4-connected to the seed through values in [seed - lo, seed + up] becomes new_val.
__global__ __launch_bounds__(1024) void flood_fill_kernel(float* __restrict__ imgs, int H, int W, float new_val, float lo, float up) {
    __shared__ unsigned long long s_t[256][FH_WORDS];
    float* a = imgs + (int64_t)blockIdx.x * H * W;
    const int tid = threadIdx.x, t = tid & 255, q = tid >> 8;
    const float seed = a[0];
    const float vmin = seed - lo, vmax = seed + up;
    for (int y0 = 0; y0 < 256; y0 += 32) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int y = y0 + 4 * j + q; v[j] = (y < H && t < W) ? a[(int64_t)y * W + t] : 0.f; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int y = y0 + 4 * j + q;
            const unsigned long long word = __ballot(y < H && t < W && v[j] >= vmin && v[j] <= vmax);
            if ((t & 63) == 0) s_t[y][t >> 6] = word;
        }
    }
    __syncthreads();
    const bool fl = tid < 256;
    unsigned long long pr[FH_WORDS] = {0, 0, 0, 0}, pc[FH_WORDS] = {0, 0, 0, 0}, rpr[FH_WORDS], rpc[FH_WORDS];
    if (fl) {
#pragma unroll
        for (int w = 0; w < FH_WORDS; ++w) pr[w] = s_t[t][w];
        if (t == 0) pr[0] |= 1ull;
    }
    __syncthreads();
    transpose256(pr, pc, s_t, t, fl);
#pragma unroll
    for (int w = 0; w < FH_WORDS; ++w) { rpr[w] = pr[w]; rpc[w] = pc[w]; }
    rev256(rpr); rev256(rpc);
    unsigned long long rr[FH_WORDS] = {(fl && t == 0) ? 1ull : 0ull, 0, 0, 0};
    for (int iter = 0; iter < 1024; ++iter) {
        unsigned long long rc[FH_WORDS] = {0, 0, 0, 0}, back[FH_WORDS] = {0, 0, 0, 0};
        if (fl) flood_line(pr, rpr, rr);
        transpose256(rr, rc, s_t, t, fl);
        if (fl) flood_line(pc, rpc, rc);
        transpose256(rc, back, s_t, t, fl);
        bool changed = false;
#pragma unroll
        for (int w = 0; w < FH_WORDS; ++w) { changed |= (back[w] & ~rr[w]) != 0; rr[w] |= back[w]; }
        if (!__syncthreads_or(fl && changed)) break;
    }
    if (fl) {
#pragma unroll
        for (int w = 0; w < FH_WORDS; ++w) s_t[t][w] = rr[w];
    }
    __syncthreads();
    for (int e = tid; e < H * W; e += 1024) {
        const int y = e / W, x = e % W;
        if ((s_t[y][x >> 6] >> (x & 63)) & 1ull) a[e] = new_val;
    }
}

extern "C" {

int n3d_rasterize_views(const float* verts, const float* lms, const float* rot, const int* faces, const float* face_uv,
                        const float* uv_mask, int mask_h, int mask_w, float* tv_ws, unsigned long long* zbuf_ws, float* grid,
                        float* alpha, float* lm2d, int N, int V, int Lm, int F, int views, int H, int W, float shift_x,
                        float shift_y, float shift_z, float scale, int fill, int binarize_view, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && V > 0 && F > 0 && views > 0 && H > 0 && W > 0 && Lm >= 0, "rasterize_views: bad shape");
    N3D_CHECK(!fill || (H <= 256 && W <= 256), "rasterize_views: fill_holes supports images up to 256x256");
    if (N == 0) return 0;
    N3D_CHECK(verts && rot && faces && face_uv && uv_mask && tv_ws && zbuf_ws && grid && alpha && (Lm == 0 || (lms && lm2d)),
              "rasterize_views: null tensor");
    const int NV = N * views;
    N3dProfScope prof(N3D_K_RASTER, stream, 0.0, 4.0 * NV * (double)H * W * 6);
    // z-buffer clear as an ordinary kernel on the launch stream
    hipLaunchKernelGGL(raster_clear_kernel, dim3((unsigned)cdiv64((int64_t)NV * H * W, 256)), dim3(256), 0, stream, zbuf_ws,
                       (int64_t)NV * H * W);
    N3D_LAUNCH_CHECK();
    const int64_t nt = (int64_t)NV * V + (int64_t)N * Lm;
    hipLaunchKernelGGL(raster_transform_kernel, dim3((unsigned)cdiv64(nt, 256)), dim3(256), 0, stream, verts, lms, rot, tv_ws, lm2d, N,
                       V, Lm, views, shift_x, shift_y, shift_z, scale);
    N3D_LAUNCH_CHECK();
    const XfParams xf = {verts, rot, V, views, shift_x, shift_y, shift_z, scale};
    hipLaunchKernelGGL(raster_faces_kernel, dim3((unsigned)cdiv64((int64_t)NV * F, 256)), dim3(256), 0, stream, (const float*)tv_ws,
                       faces, (int64_t)0, 1, zbuf_ws, NV, V, F, H, W, xf);
    N3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(raster_resolve_kernel, dim3((unsigned)cdiv64((int64_t)NV * H * W, 256)), dim3(256), 0, stream,
                       (const float*)tv_ws, faces, face_uv, (const unsigned long long*)zbuf_ws, uv_mask, mask_h, mask_w, grid, alpha,
                       NV, V, F, H, W, xf);
    N3D_LAUNCH_CHECK();
    if (fill) {
        hipLaunchKernelGGL(fill_holes_kernel, dim3(NV), dim3(1024), 0, stream, alpha, H, W, binarize_view, views);
        N3D_LAUNCH_CHECK();
    }
    return 0;
}

int n3d_rasterize_meshes(const float* verts_ndc, const int* faces, int64_t faces_batch_stride, unsigned long long* zbuf_ws, long long* pix_to_face,
                         float* zbuf, float* bary, int N, int V, int F, int H, int W, int cull_backfaces, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && V > 0 && F > 0 && H > 0 && W > 0 && H == W, "rasterize_meshes: bad shape (square images)");
    if (N == 0) return 0;
    N3D_CHECK(verts_ndc && faces && zbuf_ws && pix_to_face && zbuf && bary, "rasterize_meshes: null tensor");
    N3dProfScope prof(N3D_K_RASTER, stream, 0.0, 4.0 * N * (double)H * W * 8);
    hipLaunchKernelGGL(raster_clear_kernel, dim3((unsigned)cdiv64((int64_t)N * H * W, 256)), dim3(256), 0, stream, zbuf_ws, (int64_t)N * H * W);
    N3D_LAUNCH_CHECK();
    const XfParams xf = {verts_ndc, nullptr, V, 1, 0.f, 0.f, 0.f, 1.f};    // (only read by the RASTER_VARIANT & 16 probe build)
    hipLaunchKernelGGL(raster_faces_kernel, dim3((unsigned)cdiv64((int64_t)N * F, 256)), dim3(256), 0, stream, verts_ndc, faces, faces_batch_stride,
                       cull_backfaces ? 1 : 0, zbuf_ws, N, V, F, H, W, xf);
    N3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(raster_meshes_resolve_kernel, dim3((unsigned)cdiv64((int64_t)N * H * W, 256)), dim3(256), 0, stream, verts_ndc, faces,
                       faces_batch_stride, (const unsigned long long*)zbuf_ws, pix_to_face, zbuf, bary, N, V, F, H, W);
    N3D_LAUNCH_CHECK();
    return 0;
}

int n3d_flood_fill(float* images, int N, int H, int W, float new_val, float lo_diff, float up_diff, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && H > 0 && W > 0 && H <= 256 && W <= 256, "flood_fill: images of up to 256 x 256");
    N3D_CHECK(lo_diff >= 0.f && up_diff >= 0.f, "flood_fill: lo_diff / up_diff must be >= 0");
    if (N == 0) return 0;
    N3D_CHECK(images, "flood_fill: null tensor");
    N3dProfScope prof(N3D_K_RASTER, stream, 0.0, 8.0 * N * (double)H * W);
    hipLaunchKernelGGL(flood_fill_kernel, dim3(N), dim3(1024), 0, stream, images, H, W, new_val, lo_diff, up_diff);
    N3D_LAUNCH_CHECK();
    return 0;
}

int n3d_texture_project(const float* textures, const float* grid, float* out, int N, int C, int TH, int TW, int H, int W,
                        int views, int view_a, int view_b, n3d_stream_t stream_) {
    float* outs[1] = {out};
    return n3d_texture_project_planes(textures, grid, outs, &view_a, &view_b, 1, N, C, TH, TW, H, W, views, stream_);
}

int n3d_texture_project_planes(const float* textures, const float* grid, float* const* outs, const int* view_a, const int* view_b,
                               int planes, int N, int C, int TH, int TW, int H, int W, int views, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && C > 0 && planes >= 1 && planes <= 4 && outs && view_a && view_b, "texture_project: bad arguments");
    if (N == 0) return 0;
    N3D_CHECK(textures && grid, "texture_project: null tensor");
    TexProjPlanes pl = {};
    for (int k = 0; k < planes; ++k) {
        N3D_CHECK(outs[k] && view_a[k] >= 0 && view_a[k] < views && view_b[k] < views, "texture_project: bad plane %d", k);
        pl.out[k] = outs[k]; pl.view_a[k] = view_a[k]; pl.view_b[k] = view_b[k];
    }
    N3dProfScope prof(N3D_K_RASTER, stream, 8.0 * planes * N * C * (double)H * W, 4.0 * planes * N * C * ((double)H * W + (double)TH * TW));
    hipLaunchKernelGGL(texture_project_kernel, dim3((unsigned)cdiv64((int64_t)N * H * W, 256), planes), dim3(256), 0, stream, textures, grid,
                       pl, N, C, TH, TW, H, W, views);
    N3D_LAUNCH_CHECK();
    return 0;
}

int n3d_mouth_bbox(const float* lm2d, int* bbox, int N, int Lm, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(N >= 0 && Lm >= 60, "mouth_bbox: need >= 60 landmarks");
    if (N == 0) return 0;
    N3D_CHECK(lm2d && bbox, "mouth_bbox: null tensor");
    hipLaunchKernelGGL(mouth_bbox_kernel, dim3(cdiv(N, 64)), dim3(64), 0, stream, lm2d, bbox, N, Lm);
    N3D_LAUNCH_CHECK();
    return 0;
}

int n3d_resize_aa_strided(const float* src, int64_t src_batch_stride, float* dst, const int* src_box, const int* dst_box, int N, int C, int SH, int SW, int DH,
                          int DW, int dst_square, n3d_stream_t stream_);
int n3d_resize_aa(const float* src, float* dst, const int* src_box, const int* dst_box, int N, int C, int SH, int SW, int DH,
                  int DW, int dst_square, n3d_stream_t stream_) {
    return n3d_resize_aa_strided(src, 0, dst, src_box, dst_box, N, C, SH, SW, DH, DW, dst_square, stream_);
}
int n3d_resize_aa_strided(const float* src, int64_t src_batch_stride, float* dst, const int* src_box, const int* dst_box, int N, int C, int SH, int SW, int DH,
                          int DW, int dst_square, n3d_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    N3D_CHECK(src_batch_stride == 0 || src_batch_stride >= (int64_t)C * SH * SW, "resize_aa: source batch stride smaller than one sample");
    N3D_CHECK(N >= 0 && C > 0 && SH > 0 && SW > 0 && DH > 0 && DW > 0, "resize_aa: bad shape");
    if (N == 0) return 0;
    N3D_CHECK(src && dst, "resize_aa: null tensor");
    ResizeParams p;
    p.src = src; p.dst = dst; p.src_box = src_box; p.dst_box = dst_box; p.N = N; p.C = C; p.SH = SH; p.SW = SW; p.DH = DH; p.DW = DW;
    p.dst_square = dst_square;
    p.src_bs = src_batch_stride ? src_batch_stride : (int64_t)C * SH * SW;
    N3dProfScope prof(N3D_K_MISC, stream, 0.0, 4.0 * N * C * ((double)SH * SW + (double)DH * DW));
    if (dst_box && (int64_t)N * C < 65536)                                // a destination box: one workgroup per band of box rows of a (sample, channel)
        hipLaunchKernelGGL(resize_aa_paste_kernel, dim3(RS_BANDS, (unsigned)(N * C)), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(resize_aa_kernel, dim3((unsigned)cdiv64((int64_t)N * C * DH * DW, 256)), dim3(256), 0, stream, p);
    N3D_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
