/*
 * n3d.h — C ABI of libn3d.so, the MI355X (gfx950) implementation of the Next3D generator-forward
 * hot path.  This is the drop-in boundary: every entry point replaces one native plugin function
 * or one ATen/third-party call of the reference (file:line cited per function; paths relative to
 * the reference root, `tat/` = training_avatar_texture/, `vr/` = tat/volumetric_rendering/).
 *
 * Conventions (mirroring the reference plugins):
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless noted;
 *   - inputs are borrowed, outputs are CALLER-allocated; the library never allocates, frees or
 *     synchronises; every kernel is enqueued on the `stream` argument (a hipStream_t);
 *   - return 0 on success, negative on error; n3d_last_error() gives the message (the Python
 *     wrapper raises RuntimeError, the analogue of TORCH_CHECK in bias_act.cpp:39-55);
 *   - tensors are float32, NCHW-contiguous unless a stride argument says otherwise.
 */
#ifndef N3D_H_
#define N3D_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* n3d_stream_t; /* hipStream_t */

#define N3D_ABI_VERSION 1

/* activation ids = the reference's cuda_idx (torch_utils/ops/bias_act.py:23-33) */
enum { N3D_ACT_LINEAR = 1, N3D_ACT_RELU = 2, N3D_ACT_LRELU = 3, N3D_ACT_TANH = 4, N3D_ACT_SIGMOID = 5,
       N3D_ACT_ELU = 6, N3D_ACT_SELU = 7, N3D_ACT_SOFTPLUS = 8, N3D_ACT_SWISH = 9 };
enum { N3D_F32 = 0, N3D_F16 = 1 };

int n3d_abi_version(void);
const char* n3d_last_error(void);

/* Per-kernel-family timing for bench.py's roofline leg.  While enabled, every entry point brackets its
 * kernel launches with HIP events recorded ON THE LAUNCH STREAM and books its algorithmic flops / bytes.
 * n3d_prof_read synchronises the recorded events and returns the totals for one family since the last reset. */
enum { N3D_K_BIAS_ACT = 0, N3D_K_UPFIRDN2D = 1, N3D_K_CONV2D = 2, N3D_K_FC = 3, N3D_K_RENDER = 4, N3D_K_RASTER = 5,
       N3D_K_MISC = 6, N3D_K_COUNT = 7 };
int n3d_prof_enable(int on);
int n3d_prof_reset(void);
int n3d_prof_read(int family, double* total_ms, int64_t* launches, double* flops, double* bytes);

/* ---- bias_act: replaces bias_act_plugin.bias_act forward (torch_utils/ops/bias_act.cpp:36,
 *      kernel bias_act.cu:27-150).  y = clamp(act(x + b[(i/step_b) % size_b]) * gain).
 *      b may be NULL; clamp < 0 disables clamping; x and y may alias. */
int n3d_bias_act(const void* x, const void* b, void* y, int64_t numel, int size_b, int64_t step_b, int dtype,
                 int act, float alpha, float gain, float clamp, n3d_stream_t stream);

/* Fused per-element epilogue shared by upfirdn2d / conv2d (all optional, applied in this order):
 *   v = v * row_scale[n*O + o] * const_scale;  v += noise[oy*OW+ox] * (*noise_strength);  v += bias[o];
 *   v = act(v) * gain;  clamp;  v += residual[n, o, oy, ox]                                            */
typedef struct {
    const float* row_scale;       /* [N,O] or NULL  (demodulation coefficients, tat/networks_stylegan2.py:72-79) */
    const float* noise;           /* [OH,OW] or NULL (noise_const, :320-321)                                     */
    const float* noise_strength;  /* device scalar, required when noise != NULL                                  */
    const float* bias;            /* [O] or NULL                                                                 */
    const float* residual;        /* [N,O,OH,OW] (batch stride residual_batch_stride) or NULL                    */
    int64_t residual_batch_stride;
    float const_scale;            /* e.g. Conv2dLayer.weight_gain (tat/networks_stylegan2.py:174)                */
    int act;                      /* N3D_ACT_*                                                                   */
    float alpha, gain, clamp;     /* clamp < 0: none                                                             */
} n3d_epilogue;

/* ---- upfirdn2d: replaces upfirdn2d_plugin.upfirdn2d (torch_utils/ops/upfirdn2d.cpp:20, kernels
 *      upfirdn2d.cu:33,102).  x [N,C,H,W] -> y [N,C,OH,OW], f [fh,fw] float32 (2-D taps),
 *      OH = (H*upy + pady0 + pady1 - fh + downy) / downy (same for W).  `epi` may be NULL. */
int n3d_upfirdn2d(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw, int upx,
                  int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                  int64_t x_batch_stride, int64_t y_batch_stride, const n3d_epilogue* epi, n3d_stream_t stream);

/* ---- conv2d weight preparation (done once per model): w [O,I,k,k] -> wt [k*k][I][O] (K-major, the layout
 *      the MFMA kernel streams) and, when wsq != NULL, wsq[o*I+i] = sum_k w[o,i,k]^2 (for demodulation). */
int n3d_conv2d_prep_weight(const float* w, float* wt, float* wsq, int O, int I, int ksize, n3d_stream_t stream);

/* ---- conv2d: the one dense contraction (fp32 MFMA implicit GEMM).  Replaces the ATen conv2d /
 *      conv_transpose2d calls issued by conv2d_resample (torch_utils/ops/conv2d_resample.py:96-136) inside
 *      modulated_conv2d (tat/networks_stylegan2.py:34-91) and Conv2dLayer (:173-183).
 *        mode 0: stride 1, padding ksize/2            y [N,O,H,W]
 *        mode 1: stride 2, padding 0 (ksize 3)        y [N,O,(H-3)/2+1,(W-3)/2+1]   (conv2d_resample.py:108-111)
 *        mode 2: transposed, stride 2 (ksize 3)       y [N,O,2H+1,2W+1]              (conv2d_resample.py:114-131)
 *      style [N,I] (or NULL) multiplies input channels (modulation); `epi` is the fused epilogue.
 *      ksplit > 1 splits the input channels over extra workgroups: partial sums go to `workspace`
 *      (ksplit*N*O*OH*OW floats) and a second kernel reduces + applies the epilogue. */
typedef struct {
    const float* x;      /* [N,I,H,W], batch stride x_batch_stride */
    const float* wt;     /* prepared weights [k*k][I][O]           */
    const float* style;  /* [N,I] or NULL                          */
    float* y;            /* [N,O,OH,OW], batch stride y_batch_stride */
    float* workspace;    /* required when ksplit > 1               */
    int N, I, O, H, W;
    int ksize, mode, ksplit;
    int64_t x_batch_stride, y_batch_stride;
    n3d_epilogue epi;
} n3d_conv2d_desc;
int n3d_conv2d(const n3d_conv2d_desc* desc, n3d_stream_t stream);

/* ---- fully connected: replaces addmm / matmul+bias_act of FullyConnectedLayer.forward
 *      (tat/networks_stylegan2.py:114-127).  y[n,o] = post(act(sum_i pre(x[n,i]) * w[o,i] * wgain + b[o]*bgain)).
 *      pre_square: use x^2 (demodulation: sum_i s^2 * wsq);  post_rsqrt: y = rsqrt(y + 1e-8). */
int n3d_fc(const float* x, const float* w, const float* b, float* y, int N, int I, int O, float wgain, float bgain,
           int act, float alpha, float gain, int pre_square, int post_rsqrt, n3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* N3D_H_ */
