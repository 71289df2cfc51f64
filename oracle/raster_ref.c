/*
 * oracle/raster_ref.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, float32, no FMA contraction) of the two THIRD-PARTY
 * pieces the Next3D generator forward calls but does not ship:
 *
 *   1. pytorch3d.renderer.mesh.rasterize_meshes — call site
 *      /root/reference/training_avatar_texture/volumetric_rendering/renderer.py:415-424
 *      with the settings of renderer.py:389-397 (image 256x256, blur_radius=0,
 *      faces_per_pixel=1, perspective_correct=False, cull_backfaces=True).
 *      PyTorch3D is not vendored and not version-pinned by the reference
 *      (environment.yml lists neither pytorch3d nor opencv), so this follows the
 *      published algorithm of PyTorch3D >= 0.6 (rasterize_meshes.cu /
 *      rasterization_utils.cuh: PixToNonSquareNdc, EdgeFunctionForward,
 *      BarycentricCoordsForward, CheckPixelInsideFace).  PARITY UNPINNED for this
 *      boundary: there is no golden vector from a real PyTorch3D build.
 *
 *   2. cv2.floodFill(img32f, mask, (0,0), 255, lo=0, up=254, FLOODFILL_FIXED_RANGE)
 *      — call site renderer.py:593 (fill_mouth :583-602).  4-connectivity, a pixel
 *      joins when  seed-lo <= v <= seed+up  (seed = ORIGINAL value at (0,0)).
 *      PARITY UNPINNED likewise (OpenCV absent from tree and container).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Build:  gcc -O2 -ffp-contract=off -shared -fPIC -o oracle/liboracle_raster.so oracle/raster_ref.c
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define K_EPS 1e-8f

static inline float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    /* EdgeFunctionForward(p, v0=a, v1=b) */
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

static inline float pix_to_ndc(int i, int S) {
    /* PixToNonSquareNdc for a square image: -1 + (2 i + 1) / S */
    return -1.0f + (2.0f * (float)i + 1.0f) / (float)S;
}

/*
 * verts : [N, V, 3] float32, already in PyTorch3D NDC (+X left, +Y up), i.e. AFTER the
 *         reference's `fixed_vertices[..., :2] = -fixed_vertices[..., :2]` (renderer.py:403).
 * faces : [F, 3] int32 vertex indices (shared by all meshes of the batch).
 * out pix_to_face [N,H,W] int64 : PACKED index n*F+f, -1 where empty.
 * out zbuf        [N,H,W] f32   : -1 where empty.
 * out bary        [N,H,W,3] f32 : -1 where empty.
 */
void oracle_rasterize_meshes(const float* verts, const int32_t* faces, int N, int V, int F, int H, int W,
                             int cull_backfaces, int64_t* pix_to_face, float* zbuf, float* bary) {
    for (int64_t i = 0; i < (int64_t)N * H * W; ++i) { pix_to_face[i] = -1; zbuf[i] = -1.0f; }
    for (int64_t i = 0; i < (int64_t)N * H * W * 3; ++i) bary[i] = -1.0f;

    for (int n = 0; n < N; ++n) {
        const float* vn = verts + (size_t)n * V * 3;
        for (int f = 0; f < F; ++f) {
            const float* v0 = vn + 3 * (size_t)faces[3 * f + 0];
            const float* v1 = vn + 3 * (size_t)faces[3 * f + 1];
            const float* v2 = vn + 3 * (size_t)faces[3 * f + 2];
            const float x0 = v0[0], y0 = v0[1], z0 = v0[2];
            const float x1 = v1[0], y1 = v1[1], z1 = v1[2];
            const float x2 = v2[0], y2 = v2[1], z2 = v2[2];
            const float zmax = fmaxf(z0, fmaxf(z1, z2));
            const float zmin = fminf(z0, fminf(z1, z2));
            const float face_area = edge_fn(x0, y0, x1, y1, x2, y2); /* EdgeFunctionForward(v0,v1,v2) */
            const int is_back = face_area < 0.0f;
            const int zero_area = (face_area <= K_EPS) && (face_area >= -K_EPS);
            /* z_invalid: PyTorch3D's CheckPointOutsideBoundingBox also reports EVERY pixel as outside the box of a face with a vertex at or behind the
             * camera plane (zlims.x < kEpsilon: "faces with at least one vertex behind the camera won't render correctly and should be removed or
             * clipped before calling the rasterizer") — such a face never covers a pixel.  Irrelevant on the Next3D path (z += 10,
             * triplane_next3d.py:201) but part of rasterize_meshes' published behaviour. */
            const int z_invalid = zmin < K_EPS;
            if (zmax < 0.0f || (cull_backfaces && is_back) || zero_area || z_invalid) continue;

            const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
            const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
            /* pixel xi has NDC x = pix_to_ndc(W-1-xi); conservative pixel bounds (+-1) */
            int xi_lo = (int)floorf((float)(W - 1) - ((xmax + 1.0f) * (float)W - 1.0f) * 0.5f) - 1;
            int xi_hi = (int)ceilf((float)(W - 1) - ((xmin + 1.0f) * (float)W - 1.0f) * 0.5f) + 1;
            int yi_lo = (int)floorf((float)(H - 1) - ((ymax + 1.0f) * (float)H - 1.0f) * 0.5f) - 1;
            int yi_hi = (int)ceilf((float)(H - 1) - ((ymin + 1.0f) * (float)H - 1.0f) * 0.5f) + 1;
            if (xi_lo < 0) xi_lo = 0;
            if (yi_lo < 0) yi_lo = 0;
            if (xi_hi > W - 1) xi_hi = W - 1;
            if (yi_hi > H - 1) yi_hi = H - 1;

            const float area = edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS; /* BarycentricCoordsForward */
            for (int yi = yi_lo; yi <= yi_hi; ++yi) {
                const float yf = pix_to_ndc(H - 1 - yi, H);
                for (int xi = xi_lo; xi <= xi_hi; ++xi) {
                    const float xf = pix_to_ndc(W - 1 - xi, W);
                    /* CheckPointOutsideBoundingBox with blur 0 */
                    if (xf > xmax || xf < xmin || yf > ymax || yf < ymin) continue;
                    const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / area;
                    const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / area;
                    const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / area;
                    const float pz = w0 * z0 + w1 * z1 + w2 * z2;
                    if (pz < 0.0f) continue;
                    const int inside = (w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f);
                    if (!inside) continue; /* blur_radius == 0: dist >= 0 rejects */
                    const size_t pix = ((size_t)n * H + yi) * W + xi;
                    /* K = 1: keep the smallest z; strict '<' so the lowest face index wins ties */
                    if (pix_to_face[pix] < 0 || pz < zbuf[pix]) {
                        pix_to_face[pix] = (int64_t)n * F + f;
                        zbuf[pix] = pz;
                        bary[3 * pix + 0] = w0;
                        bary[3 * pix + 1] = w1;
                        bary[3 * pix + 2] = w2;
                    }
                }
            }
        }
    }
}

/*
 * In-place cv2.floodFill restatement on one float32 image [H,W]:
 * seed (0,0), newVal, fixed range [seed-lo, seed+up], 4-connected.
 */
void oracle_floodfill_fixed_range(float* img, int H, int W, float new_val, float lo, float up) {
    const float seed = img[0];
    const float vmin = seed - lo, vmax = seed + up;
    uint8_t* visited = (uint8_t*)calloc((size_t)H * W, 1);
    int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * (size_t)H * W);
    size_t sp = 0;
    stack[sp++] = 0;
    visited[0] = 1;
    while (sp) {
        const int32_t p = stack[--sp];
        const int y = p / W, x = p % W;
        const int ny[4] = {y - 1, y + 1, y, y};
        const int nx[4] = {x, x, x - 1, x + 1};
        for (int k = 0; k < 4; ++k) {
            if (ny[k] < 0 || ny[k] >= H || nx[k] < 0 || nx[k] >= W) continue;
            const int32_t q = ny[k] * W + nx[k];
            if (visited[q]) continue;
            const float v = img[q];
            if (v >= vmin && v <= vmax) { visited[q] = 1; stack[sp++] = q; }
        }
    }
    for (size_t i = 0; i < (size_t)H * W; ++i)
        if (visited[i]) img[i] = new_val;
    free(stack);
    free(visited);
}
