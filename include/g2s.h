/*
 * g2s.h — C ABI of libg2s.so, the MI355X (gfx950) native kernels behind the GAN2Shape
 * inner loop.  This header is the drop-in boundary: every entry point replaces one native
 * (CUDA) interface the reference binds by name.  Reference citations are relative to
 * /root/reference (alessioGalatolo/GAN-2D-to-3D).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless stated;
 *   - the caller owns and allocates every buffer (the library never allocates device memory,
 *     never synchronises, never throws);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - returns G2S_OK (0) or a negative error code; g2s_last_error() gives a thread-local message;
 *   - re-entrant.  Thread-local state: the error string, the two tuning overrides
 *     g2s_modconv_tune / g2s_raster_tune (off by default; they select among launch configurations
 *     that produce the same results, and only affect the calling thread) and g2s_set_precleared
 *     (off by default: the caller has cleared the accumulators of the next call).  Process-global state:
 *     ONE flag, g2s_set_deterministic (off by default).  No environment variable is read.
 */
#ifndef G2S_H
#define G2S_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G2S_ABI_VERSION 1

#define G2S_OK 0
#define G2S_ERR_INVALID (-1)   /* bad argument (shape / pointer / dtype / unsupported combination) */
#define G2S_ERR_LAUNCH (-2)    /* hipLaunchKernel / hipMemsetAsync reported an error              */
#define G2S_ERR_WORKSPACE (-3) /* workspace too small                                              */

#define G2S_F32 0
#define G2S_F16 1

typedef void *g2s_stream_t; /* hipStream_t */

int g2s_abi_version(void);
const char *g2s_last_error(void);

/* Reproducible launches (no reference counterpart: torch.use_deterministic_algorithms is the
 * analogue a maintainer would reach for).  on = 1: every convolution launch of the process takes a
 * partition in which ONE workgroup owns each output element — direct kernel split-K 1
 * (csrc/modconv.hip), weight-gradient GEMM without its pixel split (csrc/conv_wgrad_core.h),
 * Winograd split-K / stream-K only with stored slices (workspace given), else whole tiles — so the
 * forward values of every network are bit-identical from run to run and independent of how many
 * workgroups race.  Slower (small layers no longer fill 256 CUs); results differ from the default
 * mode by fp32 summation order only.  What remains order-dependent at the 1e-7 level: the scalar
 * loss accumulators and the gradient scatter of the rasterizer / geometry backward kernels (float
 * atomics over tiles).  Takes effect for launches issued after the call, from any thread. */
int g2s_set_deterministic(int on);
int g2s_get_deterministic(void);

/* Caller-cleared accumulators (no reference counterpart; per THREAD, returns the previous value).  The backward
 * functions below that accumulate into small outputs or workspaces clear them first with a memset of their own —
 * one more launch each (14 of step 3's).  While this flag is on, the calling thread guarantees that those buffers
 * are already zero (the Python side carves them from the step's one cleared pool, zeropool.py) and the memsets are
 * skipped: g2s_warp_verts_bwd / g2s_inv_warp_grid_bwd (gRt), g2s_smooth_loss_fwd (loss), g2s_shading_bwd (glight),
 * g2s_depth_head_bwd (gsum), g2s_grid_sample_bwd (gx, or the deterministic mode's fixed-point workspace),
 * g2s_raster_depth_bwd(_ex) (grad_verts, or its fixed-point workspace).  Set it around ONE call and restore it. */
int g2s_set_precleared(int on);

/* ------------------------------------------------------------------------------------------
 * Differentiable depth rasterizer.
 * Replaces neural_renderer.Renderer.render_depth(vertices, faces) as called from
 * GAN2Shape/renderer/renderer.py:47-54 (constructor: camera_mode='projection', K, R=I, t=0,
 * fill_back=True, anti_aliasing=True) and :116-125 (warp_canon_depth).  neural_renderer is an
 * external, un-vendored CUDA package (README.md:32-37); semantics per SURVEY.md Appendix A.
 *
 * verts      [B, n_verts, 3] f32, camera-space xyz (after the view transform)
 * faces      [n_faces, 3] i32 vertex ids shared by the whole batch, or NULL = the implicit
 *            regular-grid topology of renderer/utils.py:76-80 get_face_idx(b, S, S) with
 *            n_verts == S*S and n_faces == 2*(S-1)*(S-1)
 * K          HOST pointer, 9 floats row-major, pinhole intrinsics (third row must be 0 0 1)
 * orig_size  neural_renderer `orig_size` (== S here)
 * S          output image size; the raster runs at ssaa*S (ssaa 1 or 2; 2 == anti_aliasing)
 * fill_back  1: each face is also rendered with reversed vertex order (face id + n_faces)
 * depth_out  [B, S, S] f32  after vertical flip and ssaa x ssaa average pooling (un-clamped;
 *            background = far)
 * face_idx_out [B, ssaa*S, ssaa*S] i32 winning face id (-1 background) in UNFLIPPED raster rows,
 *            or NULL when no backward is needed
 * bary_out   [B, ssaa*S, ssaa*S, 3] f32 clamped+renormalised barycentric weights of the winner,
 *            or NULL (must be NULL iff face_idx_out is NULL)
 * workspace  >= g2s_raster_workspace_bytes(B, n_verts, n_faces, S) bytes of device scratch
 * ---------------------------------------------------------------------------------------- */
size_t g2s_raster_workspace_bytes(int B, int n_verts, int n_faces, int S);

int g2s_raster_depth_fwd(const float *verts, const int32_t *faces, int B, int n_verts, int n_faces,
                         int S, const float *K, float orig_size, int ssaa, int fill_back,
                         float near, float far, float *depth_out, int32_t *face_idx_out,
                         float *bary_out, void *workspace, size_t workspace_bytes,
                         g2s_stream_t stream);

/* Tuning hook (tools/bench_raster.py): waves_per_tile = 4 runs the 4 waves of a workgroup on ONE
 * 8x8-sample tile, 1 gives every wave its own tile, 0 restores the built-in choice (4 when
 * B * tiles <= 4096).  Calling thread only; outputs are bit-identical either way. */
int g2s_raster_tune(int waves_per_tile);

/* Backward of the above w.r.t. verts (neural_renderer backward_depth_map + vertices_to_faces +
 * projection backward).
 * grad_depth [B, S, S] f32 (gradient of depth_out); grad_verts [B, n_verts, 3] f32 is zero-filled
 * by the callee, then accumulated with float atomics. */
int g2s_raster_depth_bwd(const float *verts, const int32_t *faces, const float *grad_depth,
                         const int32_t *face_idx, const float *bary, int B, int n_verts,
                         int n_faces, int S, const float *K, float orig_size, int ssaa,
                         float *grad_verts, g2s_stream_t stream);

/* The same with the scratch deterministic mode needs (g2s_set_deterministic): the per-vertex sums are
 * then accumulated as 2^-40 fixed point with 64-bit integer atomics in `workspace`
 * (>= g2s_raster_bwd_workspace_bytes(B, n_verts) bytes of device memory), whose result does not depend
 * on the order the tiles arrive in — the gradient is bit-identical from run to run — and converted to
 * grad_verts at the end (range +-8.4e6 per component, resolution 9e-13; non-finite contributions are
 * dropped).  In deterministic mode a NULL / short workspace is G2S_ERR_WORKSPACE — also through
 * g2s_raster_depth_bwd, which passes none; otherwise the workspace is ignored. */
size_t g2s_raster_bwd_workspace_bytes(int B, int n_verts);
int g2s_raster_depth_bwd_ex(const float *verts, const int32_t *faces, const float *grad_depth,
                            const int32_t *face_idx, const float *bary, int B, int n_verts,
                            int n_faces, int S, const float *K, float orig_size, int ssaa,
                            float *grad_verts, void *workspace, size_t workspace_bytes,
                            g2s_stream_t stream);

/* Texture path: nr.Renderer.render_rgb(vertices, faces, textures [B,F,ts,ts,ts,C]) as the reference's
 * visualisation helpers call it (GAN2Shape/renderer/renderer.py:196,230,248,272,275).  Forward only.
 * Second pass over the maps g2s_raster_depth_fwd saves (run it with the constructor's near / far,
 * renderer.py:51): trilinear read of the winning face's texture cube at perspective-corrected
 * barycentric coordinates (eps = rasterizer_eps, 1e-3 in the package), `background` (HOST pointer, C
 * floats) where no face covers the sample, then the vertical flip + ssaa x ssaa average.
 * rgb_out [B, C, S, S].  PARITY UNPINNED like the depth path. */
int g2s_raster_rgb_fwd(const float *verts, const int32_t *faces, const int32_t *face_idx,
                       const float *bary, const float *textures, int B, int n_verts, int n_faces, int S,
                       int ssaa, int ts, int C, const float *background, float eps, float *rgb_out,
                       g2s_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * fused bias + activation.
 * Replaces fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)
 * (GAN2Shape/stylegan2/stylegan2-pytorch/op/fused_bias_act.cpp:11-20,
 *  op/fused_bias_act_kernel.cu:19-99).
 *   y[i] = f(x[i] + bias[(i / step_b) % size_b]) * scale
 *   act*10+grad: 10,11 -> identity ; 12 -> 0 ; 30 -> lrelu(alpha) ; 31 -> ref>0 ? x : x*alpha ;
 *   32 -> 0 ; anything else -> identity.
 * bias / ref may be NULL (== the reference's 0-element tensors). dtype G2S_F32 or G2S_F16.
 * ---------------------------------------------------------------------------------------- */
int g2s_fused_bias_act(const void *x, const void *bias, const void *ref, void *y, int64_t n,
                       int64_t step_b, int64_t size_b, int act, int grad, float alpha,
                       float scale, int dtype, g2s_stream_t stream);

/* nn.MaxPool2d(2, 2) of the VGG16 trunk (lpips/pretrained_networks.py:97-135): x [planes, H, W] ->
 * y [planes, H/2, W/2]; backward from the saved INPUT (first maximum of a window in row-major order
 * wins, NaN wins: torch's rule), gx fully written.  H even, W a multiple of 8, 16-byte aligned. */
int g2s_maxpool2x2_fwd(const float *x, float *y, int64_t planes, int H, int W, g2s_stream_t stream);
int g2s_maxpool2x2_bwd(const float *x, const float *gy, float *gx, int64_t planes, int H, int W,
                       g2s_stream_t stream);

/* Residual joins of the frozen nets in one pass: y = (a + b + bias[c]) * scale, c = (i / hw) % C.
 * ToRGB: conv + bias + upsample(skip) (stylegan2-pytorch/model.py:371-377); discriminator ResBlock:
 * (out + skip) / sqrt(2) (model.py:693-697).  a, b, y [n] f32; b and bias [C] may be NULL. */
int g2s_add_bias_scale(const float *a, const float *b, const float *bias, float *y, int64_t n, int64_t hw, int C,
                       float scale, g2s_stream_t stream);

/* The mask pyramid of DiscriminatorLoss (GAN2Shape/losses.py:24-30): n_levels (1..4) successive
 * avg_pool2d(., 2, 2) of x [planes, H, W] in one launch; levels: HOST array of n_levels device pointers,
 * level l [planes, H / 2^(l+1), W / 2^(l+1)].  H, W multiples of 2^n_levels.  ATen's arithmetic per window. */
int g2s_avg_pyramid(const float *x, float *const *levels, int n_levels, int64_t planes, int H, int W,
                    g2s_stream_t stream);

/* Entry of the offset encoder's residual block (GAN2Shape/networks.py:170-194): relu_out = relu(x) for the
 * residual path and pool_out = avg_pool2d(x, 2, 2) for the identity path in one pass over x [planes, H, W]
 * (H, W even); backward gx = g_relu * (x > 0) + g_pool / 4 in one pass (either gradient may be NULL). */
int g2s_res_split_fwd(const float *x, float *relu_out, float *pool_out, int64_t planes, int H, int W,
                      g2s_stream_t stream);
int g2s_res_split_bwd(const float *x, const float *g_relu, const float *g_pool, float *gx, int64_t planes, int H,
                      int W, g2s_stream_t stream);

/* The depth head (GAN2Shape/model.py:337-345 get_clamped_depth with :323-324 rescale_depth) on the depth net's
 * raw map, raw [n = B*H*W] f32:  t = tanh(raw - mean[0]);  d = (1 + t) / 2 * hi + (1 - t) / 2 * lo;
 * clamp_border: out = d * (1 - b) + b * border_depth with b = 1.02 in the two left / right columns of every
 * row of W, else 0.  mean: device float (the whole-batch mean of raw, reduced by the caller — under data
 * parallelism the all-reduced one).  _bwd: g_raw = gc - mean(gc), gc = g * d out / d (raw - mean) — the
 * centring's own backward included (mean taken over these n elements); gsum: one device float of scratch. */
int g2s_depth_head_fwd(const float *raw, const float *mean, float *out, int64_t n, int W, float lo, float hi,
                       int clamp_border, float border_depth, g2s_stream_t stream);
int g2s_depth_head_bwd(const float *raw, const float *mean, const float *g, float *g_raw, float *gsum, int64_t n,
                       int W, float lo, float hi, int clamp_border, float border_depth, g2s_stream_t stream);

/* The reconstruction warp: torch.nn.functional.grid_sample(x, grid, mode='bilinear', padding_mode='zeros',
 * align_corners=True) as GAN2Shape/model.py:147,267 calls it, optionally followed by .clamp(lo, hi)
 * (model.py:150,270) in the same pass (clamp = 1).  ATen's arithmetic (corner order nw, ne, sw, se).
 *   x [B, C, IH, IW], grid [B, H, W, 2] (x then y in [-1, 1]), y / gy [B, C, H, W], all f32.
 * Backward: gx [B, C, IH, IW] (zero-filled by the callee, then scattered) and ggrid [B, H, W, 2]; either
 * may be NULL.  With clamp = 1 the gradient passes where lo <= sample <= hi (torch's rule), the sample
 * being recomputed.  The scatter uses float atomics; in deterministic mode (g2s_set_deterministic) it
 * accumulates 2^-40 fixed point with 64-bit integer atomics in `workspace`
 * (>= g2s_grid_sample_bwd_workspace_bytes) and converts at the end — bit-identical from run to run, which
 * ATen's grid_sampler_2d_backward cannot offer; there a NULL / short workspace is G2S_ERR_WORKSPACE. */
int g2s_grid_sample_fwd(const float *x, const float *grid, float *y, int B, int C, int IH, int IW, int H, int W,
                        int clamp, float lo, float hi, g2s_stream_t stream);
size_t g2s_grid_sample_bwd_workspace_bytes(int B, int C, int IH, int IW);
int g2s_grid_sample_bwd(const float *gy, const float *x, const float *grid, float *gx, float *ggrid, int B, int C,
                        int IH, int IW, int H, int W, int clamp, float lo, float hi, void *workspace,
                        size_t workspace_bytes, g2s_stream_t stream);

/* torch.clamp(x, lo, hi) of the step's image / depth clamps (GAN2Shape/model.py:150,270, renderer.py:123-124)
 * and its backward as ONE launch each (autograd's ClampBackward is five: ge, le, logical_and, where, fill).
 * backward = 0: y = min(max(x, lo), hi);  backward = 1: y = g where lo <= x <= hi, else 0 (torch's rule;
 * NaN in x: forward gives NaN like torch, backward 0).  x, g, y [n] f32. */
int g2s_clamp(const float *x, const float *g, float *y, int64_t n, float lo, float hi, int backward,
              g2s_stream_t stream);

/* StyledConv tail in one pass (stylegan2-pytorch/model.py:349-355: NoiseInjection then
 * FusedLeakyReLU): y = lrelu_alpha(x + noise_w * noise[hw] + bias[c]) * scale.
 * x,y [B, C, HW] f32; noise [HW] f32 (one map broadcast over B and C) or NULL;
 * noise_w is a DEVICE pointer to 1 float. */
int g2s_noise_bias_act(const float *x, const float *noise, const float *noise_w,
                       const float *bias, float *y, int B, int C, int HW, float alpha,
                       float scale, g2s_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * upfirdn2d.
 * Replaces upfirdn2d_op.upfirdn2d(input[M,H,W,1], kernel[kh,kw], up_x, up_y, down_x, down_y,
 * pad_x0, pad_x1, pad_y0, pad_y1) (op/upfirdn2d.cpp:12-22, op/upfirdn2d_kernel.cu:209-369).
 * x [major, in_h, in_w] (minor == 1), k [kh, kw] f32 on device, y [major, out_h, out_w] with
 * out = (in*up + pad0 + pad1 - k + down) / down.  Negative pads crop.
 * ---------------------------------------------------------------------------------------- */
int g2s_upfirdn2d(const void *x, const float *k, void *y, int major, int in_h, int in_w, int kh,
                  int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                  int pad_y0, int pad_y1, int dtype, g2s_stream_t stream);
/* upfirdn2d (f32, one up / down factor for both axes) with the StyledConv tail of an UP-SAMPLING layer in its store
 * (stylegan2-pytorch/model.py:264-275: conv_transpose2d, then Blur; :349-355: noise, bias, leaky ReLU):
 *     y[m, oy, ox] = gain * leaky_relu(upfirdn2d(x)[m, oy, ox] + noise_w[0] * noise[oy, ox] + bias[m % channels], alpha)
 * x [major = B * channels, in_h, in_w]; noise [out_h, out_w]; noise_w a DEVICE pointer to 1 float. */
int g2s_upfirdn2d_nba(const float *x, const float *k, float *y, int major, int channels, int in_h, int in_w, int kh,
                      int kw, int up, int down, int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float *bias,
                      const float *noise, const float *noise_w, float alpha, float gain, g2s_stream_t stream);



/* ------------------------------------------------------------------------------------------
 * Modulated convolution (StyleGAN2 generator) as an fp32-MFMA implicit GEMM.
 * Replaces the grouped F.conv2d / F.conv_transpose2d inside ModulatedConv2d.forward
 * (stylegan2-pytorch/model.py:250-291) and their autograd data-gradients, using the
 * input-scaling formulation
 *     y[b,o] = out_scale[b,o] * sum_{i,ky,kx} w[o,i,ky,kx] * (in_scale[b,i] * x[b,i, ...])
 * (in_scale = modulated style, out_scale = demodulation factor), which equals the reference's
 * per-sample-weight formulation up to fp32 rounding.
 *
 * w is always the forward layout [Cout, Cin, k, k] (already multiplied by 1/sqrt(fan_in)).
 * transpose = 0: x has Cin channels, y has Cout channels, reduction over (i, ky, kx):
 *   G2S_CONV_PLAIN : y[b,o,Y,X]            = sum x[b,i,Y+ky-k/2,X+kx-k/2] w[o,i,ky,kx]  (model.py:286-289)
 *   G2S_CONV_UP2   : y[b,o,2y+ky,2x+kx]   += x[b,i,y,x] w[o,i,ky,kx]   out (H-1)*2+k    (model.py:264-274;
 *                    the Blur that follows is g2s_upfirdn2d)
 *   G2S_CONV_DOWN2 : y[b,o,Y,X]            = sum x[b,i,2Y+ky,2X+kx] w[o,i,ky,kx]  out (H-k)/2+1 (model.py:277-283)
 * transpose = 1: the ADJOINT of the same mode's transpose=0 map w.r.t. x — x has Cout channels,
 *   y has Cin channels, reduction over (o, ky, kx), no weight repacking:
 *   G2S_CONV_PLAIN : y[b,i,Y,X]            = sum x[b,o,Y-ky+k/2,X-kx+k/2] w[o,i,ky,kx]
 *   G2S_CONV_DOWN2 : y[b,i,2Y+ky,2X+kx]   += x[b,o,Y,X] w[o,i,ky,kx]            (adjoint of DOWN2)
 *   G2S_CONV_UP2   : y[b,i,Y,X]            = sum x[b,o,2Y+ky,2X+kx] w[o,i,ky,kx] (adjoint of UP2;
 *                    here H, W are the sizes of x and must be odd: out (H-k)/2+1)
 * x [B, C_in_of_this_call, H, W] f32; in_scale [B, C_in_of_this_call] or NULL;
 * out_scale [B, C_out_of_this_call] or NULL; y fully overwritten.  k in {1, 3}.
 * ---------------------------------------------------------------------------------------- */
#define G2S_CONV_PLAIN 0
#define G2S_CONV_UP2 1
#define G2S_CONV_DOWN2 2

int g2s_modconv(const float *x, const float *w, const float *in_scale, const float *out_scale,
                float *y, int B, int Cin, int Cout, int H, int W, int k, int mode, int transpose,
                g2s_stream_t stream);

/* Plain convolution on the same kernel with a fused epilogue:
 *   y = act(conv(x, w) + bias[o]),  act = 0: identity, 1: leaky-ReLU(alpha) * gain (alpha = 0, gain = 1: ReLU).
 * Used for the frozen VGG16 of the LPIPS loss (conv3x3 + bias + ReLU, lpips/pretrained_networks.py:97-135).
 * Same geometry modes as g2s_modconv with transpose = 0; bias may be NULL. */
int g2s_conv_bias_act(const float *x, const float *w, const float *bias, float *y, int B, int Cin,
                      int Cout, int H, int W, int k, int mode, int act, float alpha, float gain,
                      g2s_stream_t stream);

/* g2s_modconv and g2s_conv_bias_act behind one entry point (in_scale / out_scale / bias may each be
 * NULL, act as above), plus y_is_zero: the caller's promise that y is already all zeros — e.g. a slice
 * of a pool cleared ONCE per training step (gan-2d-to-3d_amd/zeropool.py).  Launches that add partial
 * sums into y (split-K slices of small layers; polyphase classes that leave holes) then skip their own
 * clear: one graph node less each (36 of step 1's 319 nodes were such clears).
 * g2s_modconv_needs_zero answers, without launching anything, whether that signature adds into a
 * cleared output (1) or overwrites y (0) under the calling thread's current tuning state, so that
 * only those outputs are taken from the pool.  fused = a bias / activation epilogue is present. */
int g2s_modconv_ex(const float *x, const float *w, const float *in_scale, const float *out_scale,
                   const float *bias, float *y, int B, int Cin, int Cout, int H, int W, int k, int mode,
                   int transpose, int act, float alpha, float gain, int y_is_zero, g2s_stream_t stream);
int g2s_modconv_needs_zero(int B, int Cin, int Cout, int H, int W, int k, int mode, int transpose,
                           int has_scales, int fused);
/* g2s_modconv_ex with the WHOLE StyledConv tail in the epilogue (stylegan2-pytorch/model.py:321-355: ModulatedConv2d,
 * NoiseInjection, FusedLeakyReLU — the reference runs conv, `image + weight * noise` and fused_bias_act as three ops):
 *     y = gain * leaky_relu(out_scale[b,o] * conv(in_scale * x)[b,o,h,w] + noise_w[0] * noise[h,w] + bias[o], alpha)
 * noise [OH, OW] f32 (one map for all samples and channels), noise_w a DEVICE pointer to 1 float, bias [Cout].
 * A split-K launch finishes with g2s_noise_bias_act in place. */
int g2s_modconv_nba(const float *x, const float *w, const float *in_scale, const float *out_scale, const float *bias,
                    const float *noise, const float *noise_w, float *y, int B, int Cin, int Cout, int H, int W, int k,
                    int mode, int transpose, float alpha, float gain, int y_is_zero, g2s_stream_t stream);

/* General 2-D convolution on the same fp32-MFMA implicit-GEMM kernel: the trained nets of the step
 * (depth / albedo / viewpoint / lighting / offset-encoder nets, GAN2Shape/networks.py:23-244:
 * nn.Conv2d k in {1,3,4,5} stride 1/2, nn.ConvTranspose2d k4 stride 1/2), forward and data-gradient.
 *   adjoint = 0: y[b,m,oy,ox]               = sum_{c,ky,kx} x[b,c,oy*s+ky-p,ox*s+kx-p] W(m,c,ky,kx),
 *                out ((H+2p-k)/s+1) x ((W+2p-k)/s+1)                    (Conv2d forward; ConvTranspose2d dgrad)
 *   adjoint = 1: y[b,m,iy*s+ky-p,ix*s+kx-p] += x[b,c,iy,ix] W(m,c,ky,kx), out out_h x out_w, 0 = (H-1)s-2p+k
 *                (up to s-1 more, as a strided Conv2d's input can be)    (ConvTranspose2d forward; Conv2d dgrad)
 *   W(m,c,ky,kx) = w[(m*Cr + c)*k*k + ky*k + kx] if w_m_major (w is [M,Cr,k,k]) else w[(c*M + m)*k*k + ...]
 * x [B,Cr,H,W], y [B,M,...] f32, fully overwritten; bias [M] or NULL, act as g2s_conv_bias_act.
 * k 1..5, stride 1 or 2, 0 <= pad < k.
 * g2s_conv2d_wgrad: dw[a,g,ky,kx] = sum_{b,py,px} A[b,a,py,px] * G[b,g,py*s+ky-p,px*s+kx-p]
 *   (zero outside G), A [B,Ca,PH,PW], G [B,Cg,GH,GW], dw [Ca,Cg,k,k] fully overwritten.
 *   Conv2d weight gradient: A = grad_out, G = input; ConvTranspose2d: A = input, G = grad_out.
 *   Summation over pixels is split over workgroups and added with float atomics.
 * y_is_zero / dw_is_zero != 0: the caller hands over zero-filled outputs (e.g. slices of one cleared
 *   arena per network pass), so the split-K paths skip their own clear of the output. */
int g2s_conv2d(const float *x, const float *w, const float *bias, float *y, int B, int Cr, int M, int H,
               int W, int k, int stride, int pad, int adjoint, int w_m_major, int out_h, int out_w,
               int act, float alpha, float gain, int y_is_zero, g2s_stream_t stream);
int g2s_conv2d_wgrad(const float *A, const float *G, float *dw, int B, int Ca, int Cg, int PH, int PW,
                     int GH, int GW, int k, int stride, int pad, int dw_is_zero, g2s_stream_t stream);
/* `groups` independent convolutions in one launch — two structurally identical trained nets (depth +
 * albedo, viewpoint + lighting: GAN2Shape/networks.py:53-167 differ only in their last layer) run
 * with their channels side by side: x [B, groups*Cr, H, W], y [B, groups*M, ...], w and bias hold the
 * groups back to back ([groups][M][Cr][k][k], or [groups][Cr][M][k][k] when w_m_major = 0); for the
 * weight gradient A [B, groups*Ca, ...], G [B, groups*Cg, ...], dw [groups][Ca][Cg][k][k].  Per
 * group the arithmetic is that of g2s_conv2d / g2s_conv2d_wgrad. */
int g2s_conv2d_grouped(const float *x, const float *w, const float *bias, float *y, int B, int Cr, int M,
                       int H, int W, int k, int stride, int pad, int adjoint, int w_m_major, int out_h,
                       int out_w, int act, float alpha, float gain, int y_is_zero, int groups,
                       g2s_stream_t stream);
int g2s_conv2d_wgrad_grouped(const float *A, const float *G, float *dw, int B, int Ca, int Cg, int PH,
                             int PW, int GH, int GW, int k, int stride, int pad, int dw_is_zero, int groups,
                             g2s_stream_t stream);

/* Backward of one convolution layer of the trained nets as ONE launch: the data-gradient
 *   g2s_conv2d[_grouped](gy, w, NULL, gx, B, Cr, M, H, W, k, stride, pad, adjoint, w_m_major, out_h, out_w,
 *                        0, 0, 1, gx_is_zero, [groups,] stream)
 * and the weight-gradient
 *   g2s_conv2d_wgrad[_grouped](A, G, dw, B, Ca, Cg, PH, PW, GH, GW, k, stride, pad, dw_is_zero, [groups,] stream)
 * of the same layer (k, stride, pad, B, groups shared) in one grid — the two are independent and
 * latency-bound at these sizes, so the layer costs the longer of the two.  Results are those of
 * the two separate calls.  (torch's ConvolutionBackward of nn.Conv2d / nn.ConvTranspose2d,
 * GAN2Shape/networks.py:23-244.) */
int g2s_conv2d_bwd(const float *gy, const float *w, float *gx, int B, int Cr, int M, int H, int W, int k,
                   int stride, int pad, int adjoint, int w_m_major, int out_h, int out_w, int gx_is_zero,
                   const float *A, const float *G, float *dw, int Ca, int Cg, int PH, int PW, int GH, int GW,
                   int dw_is_zero, int groups, g2s_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Adam step of a list of parameter tensors in ONE launch (torch.optim.Adam(params, lr, betas,
 * weight_decay) of GAN2Shape/trainer.py:163-171: classic L2 weight decay, amsgrad off):
 *   g' = g + wd p;  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2;  t = *step + 1;
 *   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps);  then *step = t.
 * tensors  DEVICE array [n_tensors] of {p, m, v, n, step, ticket} — everything that persists from
 *          step to step: p, m, v f32 with n elements each; step a DEVICE f32 scalar per tensor
 *          (steps taken so far, as torch counts them per parameter; advanced by the launch, so a
 *          captured graph replays the right bias correction); ticket a DEVICE int per tensor, zero
 *          before and after every launch;
 * chunk0   DEVICE array [n_tensors + 1]: prefix sums of ceil(n / g2s_adam_chunk()) — tensor i owns
 *          the chunks chunk0[i] .. chunk0[i + 1] - 1; n_chunks = chunk0[n_tensors];
 * grads    HOST array [n_tensors] of device pointers to the gradients (they move from step to step;
 *          passed to the kernel by value).  n_tensors <= G2S_ADAM_MAX_TENSORS per call. */
#define G2S_ADAM_MAX_TENSORS 128
typedef struct g2s_adam_tensor {
    float *p, *m, *v;
    int64_t n;
    float *step;
    int *ticket;
} g2s_adam_tensor;
int64_t g2s_adam_chunk(void);
int g2s_adam_step(const g2s_adam_tensor *tensors, const int *chunk0, const float *const *grads, int n_tensors,
                  int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay,
                  g2s_stream_t stream);

/* fp16-OPERAND form of g2s_modconv / g2s_conv_bias_act (BASELINE config 5, "fp16 MFMA path"): same
 * arguments and geometry; x, w, y stay fp32 in memory, both GEMM operands are rounded to fp16 on
 * their way into LDS and multiplied by v_mfma_f32_32x32x8_f16 with fp32 accumulation.  bias (NULL ok)
 * and act (0 none / 1 leaky-ReLU(alpha) * gain) form the epilogue.  1x1 and 3x3 kernels only.
 * Results differ from the fp32 kernels at the 1e-3 level (fp16 rounding of the operands). */
int g2s_modconv_f16(const float *x, const float *w, const float *in_scale, const float *out_scale,
                    const float *bias, float *y, int B, int Cin, int Cout, int H, int W, int k, int mode,
                    int transpose, int act, float alpha, float gain, g2s_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 3x3 stride-1 convolution (padding 1) as Winograd F(2x2, 3x3) on the fp32 matrix cores
 * (csrc/winograd.hip): the same map as g2s_modconv(mode = G2S_CONV_PLAIN, k = 3) — the generator's
 * plain StyledConvs (stylegan2-pytorch/model.py:285-289), the discriminator's ResBlock conv1
 * (model.py:679-697), the VGG16 trunk of LPIPS (lpips/pretrained_networks.py:97-135) and their
 * data-gradients — with 2.25x fewer multiplications; results equal up to fp32 rounding.
 *
 * g2s_wino_weights: U = G g G^T for every (output, reduction) channel pair of w [Cout, Cin, 3, 3],
 *   written to U (g2s_wino_weights_floats(M, Cr) floats, tiled layout private to the library).
 *   transpose = 0: M = Cout, Cr = Cin (forward).  transpose = 1: M = Cin, Cr = Cout, taps flipped
 *   (the data-gradient of the forward map).  Done once per weight tensor by the caller.
 * g2s_conv3x3_wino: y[b,m] = act(out_scale[b,m] * sum_c conv3x3(in_scale[b,c] * x[b,c], w(m,c)) + bias[m])
 *   x [B, Cr, H, W], y [B, M, H, W]; in_scale [B, Cr], out_scale [B, M], bias [M] may be NULL;
 *   act 0: none, 1: leaky-ReLU(alpha) * gain.  splitk = 0: partition chosen by the library — whole
 *   tiles, or "stream-K": equal runs of (tile, K tile) units over 256 workgroups when whole tiles
 *   would fill the last round of CUs badly; splitk > 0: that K split of every tile; splitk < 0:
 *   stream-K over -splitk workgroups.  Partial sums meet by float atomics in a cleared y and the
 *   bias / activation then runs as one deferred elementwise launch — or, for split-K with a
 *   workspace `ws` (device scratch of ws_floats >= slices x B*M*H*W floats, <= 8 slices; NULL = none;
 *   contents undefined afterwards; one per stream), the slices store their partial sums side by
 *   side in ws and one elementwise pass adds them with the bias / activation: no clear, no atomics
 *   (this kernel's workgroups reach their epilogue together, so their atomics cannot hide behind
 *   matrix work: 35 of 72 us on 8 x 512 x 16 x 16 in 4 slices), deterministic sums.
 * ---------------------------------------------------------------------------------------- */
size_t g2s_wino_weights_floats(int M, int Cr);
int g2s_wino_weights(const float *w, float *U, int Cout, int Cin, int transpose, g2s_stream_t stream);
int g2s_conv3x3_wino(const float *x, const float *U, const float *in_scale, const float *out_scale,
                     const float *bias, float *y, int B, int Cr, int M, int H, int W, int act,
                     float alpha, float gain, int splitk, float *ws, int64_t ws_floats, g2s_stream_t stream);
/* The same with the StyledConv tail in the epilogue (see g2s_modconv_nba): act = leaky-ReLU, bias / noise / noise_w
 * required; the split-K and stream-K reduce passes add the noise term with the bias. */
int g2s_conv3x3_wino_nba(const float *x, const float *U, const float *in_scale, const float *out_scale,
                         const float *bias, const float *noise, const float *noise_w, float *y, int B, int Cr, int M,
                         int H, int W, float alpha, float gain, int splitk, float *ws, int64_t ws_floats,
                         g2s_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The same map as Winograd F(4x4, 3x3) (csrc/winograd4.hip): 36 multiplications per 16 outputs — 1.78x fewer than
 * g2s_conv3x3_wino, 4x fewer than the direct form — for the large maps of the frozen networks.  Interpolation points
 * 0, +-1, +-2, inf; U in double arithmetic, one rounding; fp32 everywhere else (error bound and measurements: DESIGN §4.3b).
 * g2s_wino4_supported: 1 when the kernel takes the shape (H, W multiples of 4, W / 4 divides 32, M a multiple of 64,
 *   Cr a multiple of 4); the caller falls back to g2s_conv3x3_wino otherwise.
 * g2s_wino4_weights / _floats: as g2s_wino_weights, in this kernel's own tiled layout.
 * g2s_conv3x3_wino4: y[b,m] = act(out_scale[b,m] * sum_c conv3x3(in_scale[b,c] * x[b,c], w(m,c)) + noise_w[0] *
 *   noise[h,w] + bias[m]); in_scale, out_scale, bias, noise (+ noise_w) may be NULL.  splitk = 0: whole tiles, or K
 *   slices through the workspace `ws` (as above) when whole tiles would leave CUs idle; splitk > 0: that split (needs
 *   ws; falls back to whole tiles without one).  No atomics on any path: sums have a fixed order.
 * ---------------------------------------------------------------------------------------- */
size_t g2s_wino4_weights_floats(int M, int Cr);
int g2s_wino4_weights(const float *w, float *U, int Cout, int Cin, int transpose, g2s_stream_t stream);
int g2s_wino4_supported(int B, int Cr, int M, int H, int W);
int g2s_conv3x3_wino4(const float *x, const float *U, const float *in_scale, const float *out_scale,
                      const float *bias, const float *noise, const float *noise_w, float *y, int B, int Cr, int M,
                      int H, int W, int act, float alpha, float gain, int splitk, float *ws, int64_t ws_floats,
                      g2s_stream_t stream);

/* Measurement aid (csrc/probe.hip, tools/bench_mfma_peak.py), not part of the training path: `blocks` workgroups of
 * `waves` wavefronts (1..8), each issuing iters x 8 independent v_mfma_f32_32x32x2_f32 from registers — what the fp32
 * matrix pipe sustains against the nominal 157.3 TFLOP/s the rooflines use.  FLOP = blocks * waves * iters * 8 * 4096. */
int g2s_mfma_probe(float *out, int blocks, int waves, int iters, g2s_stream_t stream);
/* The same with the Winograd inner loop's operand traffic: 8 x 16-byte LDS reads per lane per 16 MFMAs; acc_agpr = 1
 * forces the accumulators into AccVGPRs.  FLOP = blocks * waves * iters * 16 * 4096. */
int g2s_mfma_lds_probe(float *out, int blocks, int waves, int iters, int acc_agpr, g2s_stream_t stream);

/* Tuning hook (tools/tune_modconv.py): force the tile configuration (0: 128x128, 1: 128x64,
 * 2: 64x64, 3: 32x128, 4: 64x128 output channels x pixels) and/or the split-K factor of the calling thread's following
 * g2s_modconv / g2s_conv_bias_act / g2s_conv2d launches (slices of the deepest polyphase class of a
 * strided scatter; shallower classes get proportionally fewer); -1 restores the built-in choice
 * (measured tables csrc/modconv_tuned.inc / conv2d_tuned.inc, else a heuristic), tile = -2 selects
 * the heuristic alone.  Results do not depend on the choice beyond the fp32 summation order. */
int g2s_modconv_tune(int tile, int splitk);

/* ------------------------------------------------------------------------------------------
 * Row-wise fused reductions around the modulated convolution (csrc/rowops.hip).
 * g2s_rows_dot_scale: a, b, out are [rows, n] f32; s, inv, dot are [rows].
 *   dot[r]   = (sum_i a[r,i] * b[r,i]) * (inv ? 1 / inv[r] : 1)      (dot may be NULL; needs a)
 *   out[r,i] = b[r,i] * s[r]                                          (out may be NULL or alias b)
 * g2s_demod_fwd: demod[b,o] = rsqrt(sum_i wsq[o,i] * s[b,i]^2 + eps)  (model.py:254-258)
 * g2s_demod_bwd: gs[b,i] = -s[b,i] * sum_o gd[b,o] * demod[b,o]^3 * wsq[o,i]
 * ---------------------------------------------------------------------------------------- */
int g2s_rows_dot_scale(const float *a, const float *b, const float *s, const float *inv, float *out,
                       float *dot, int rows, int n, g2s_stream_t stream);
int g2s_demod_fwd(const float *wsq, const float *s, float *demod, int B, int Cin, int Cout,
                  float eps, g2s_stream_t stream);
int g2s_demod_bwd(const float *wsq, const float *s, const float *demod, const float *gd, float *gs,
                  int B, int Cin, int Cout, g2s_stream_t stream);
/* The same plus a gradient that reaches s from elsewhere (the convolution's own in_scale path):
 * gs = gs_add + (the above); gs_add [B, Cin] may be NULL, may alias gs.  Saves autograd's accumulation launch. */
int g2s_demod_bwd_add(const float *wsq, const float *s, const float *demod, const float *gd, const float *gs_add,
                      float *gs, int B, int Cin, int Cout, g2s_stream_t stream);
/* out[c] = sum over b and i of g[b, c, i] — the bias gradient of nn.Conv2d(bias=True) (the offset encoder's residual
 * blocks, GAN2Shape/networks.py:170-244; autograd's sum over dims (0, 2, 3)).  g [B, C, n], out [C] f32; fixed order. */
int g2s_channel_sum(const float *g, float *out, int B, int C, int n, g2s_stream_t stream);
/* g2s_demod_fwd / g2s_demod_bwd_add of up to G2S_DEMOD_MAX_LAYERS layers in ONE launch each (the styled layers of
 * the frozen generator, synthesis.py): HOST arrays of `layers` device pointers / sizes; layer l has wsq [Cout_l,
 * Cin_l], s [B, Cin_l], demod [B, Cout_l], gd [B, Cout_l]; the backward ADDS its result to gs [B, Cin_l] in place. */
#define G2S_DEMOD_MAX_LAYERS 24
int g2s_demod_fwd_multi(const void *const *wsq, const void *const *s, const void *const *demod, const int *Cin,
                        const int *Cout, int layers, int B, float eps, g2s_stream_t stream);
int g2s_demod_bwd_multi(const void *const *wsq, const void *const *s, const void *const *demod, const void *const *gd,
                        const void *const *gs, const int *Cin, const int *Cout, int layers, int B, g2s_stream_t stream);
/* Backward of the frozen generator, one pass per activation x = gain * leaky_relu(yconv + noise_w * noise + bias)
 * between two layers (stylegan2-pytorch/model.py:321-355,545-627; gan-2d-to-3d_amd/synthesis.py).  x is the output
 * of the producer's StyledConv tail and the input of its consumers: the next modulated convolution (g1 = gradient
 * w.r.t. s1 * x from its data-gradient GEMM) and, where present, ToRGB (g2, s2; all three NULL otherwise).
 * Rows r = (b, c) of n = H * W:
 *     dot1[r] = sum_i x g1,  dot2[r] = sum_i x g2                               (style gradients of the consumers)
 *     out[r,i] = (g1 s1[r] + g2 s2[r]) * gain * (x > 0 ? 1 : slope)             (op/fused_act.py:33-38 with ref = x;
 *                                                                                out may be NULL)
 *     gdot[r]  = sum_i out * yconv / demod[r],  yconv = (x > 0 ? x : x / slope) / gain - noise_w[0] noise[i] - bias[c]
 *                (d loss / d demodulation of the producer; gdot NULL: skipped, noise / bias / demod unused)
 * x, g1, g2, out [rows, n] f32; s1, s2, demod, dot1, dot2, gdot [rows]; bias [channels], c = r % channels; noise [n]. */
int g2s_synth_bwd_rows(const float *x, const float *g1, const float *s1, const float *g2, const float *s2,
                       const float *noise, const float *noise_w, const float *bias, const float *demod, float *out,
                       float *dot1, float *dot2, float *gdot, int rows, int channels, int n, float slope, float gain,
                       g2s_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * LPIPS per-layer tail (csrc/lpips.hip): unit-normalise both feature maps over channels, weighted
 * squared difference, spatial mean — lpips/networks_basic.py:64-92 + lpips/__init__.py:40-42.
 * f0, f1 [N, C, HW] f32; w [C] (the 1x1 `lin` weights); out [N] is ACCUMULATED into
 * (out[n] += mean_hw sum_c w_c (f0n - f1n)^2): zero it before the first layer.
 * _bwd: g0 [N, C, HW] = d(sum_n gout[n] * out[n]) / d f0   (f1 is the target branch).
 * ---------------------------------------------------------------------------------------- */
int g2s_lpips_layer_fwd(const float *f0, const float *f1, const float *w, float *out, int N, int C,
                        int HW, g2s_stream_t stream);
int g2s_lpips_layer_bwd(const float *f0, const float *f1, const float *w, const float *gout,
                        float *g0, int N, int C, int HW, g2s_stream_t stream);
/* The same inside the VGG trunk's hand-written backward (gan2shape_amd/lpips.py): f0 is the ReLU output
 * of the slice's last convolution (pretrained_networks.py:97-135), so the gradient that reaches it is
 * g_in (from the next slice through its max pool; NULL for the last slice) + this layer's tail, passed
 * only where f0 > 0 when relu_gate = 1 — one launch instead of tail + add + ReLU backward. */
int g2s_lpips_layer_bwd_ex(const float *f0, const float *f1, const float *w, const float *gout,
                           const float *g_in, int relu_gate, float *g0, int N, int C, int HW,
                           g2s_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Masked L1 (csrc/losses.hip): the numerator of PhotometricLoss and of every DiscriminatorLoss level
 * (GAN2Shape/losses.py:6-51: (|a - b| * mask.expand_as(loss)).sum() / mask.expand_as(loss).sum()).
 * x, y [B, C, HW] f32; w [B, HW] (the mask at this resolution) or NULL (weight 1); HW % 4 == 0.
 * _fwd: *num += sum |x - y| * w   (num: a ZEROED device float).  The denominator C * sum(w) only
 *       needs the small mask and is left to the caller.
 * _bwd: gx = sign(x - y) * w * coef[0]   (coef: device float = incoming gradient / denominator).
 * ---------------------------------------------------------------------------------------- */
int g2s_weighted_l1_fwd(const float *x, const float *y, const float *w, float *num, int B, int C, int HW,
                        g2s_stream_t stream);
int g2s_weighted_l1_bwd(const float *x, const float *y, const float *w, const float *coef, float *gx, int B,
                        int C, int HW, g2s_stream_t stream);
/* The loss one division away, in the same two passes:
 * _fwd2: numden[0] += sum |x - y| * w,  numden[1] += sum over (b, c, p) of w — the reference's
 *        mask.expand_as(loss).sum(); w == NULL: the element count.  numden: two ZEROED device floats.
 * _bwd2: gx = gadd + sign(x - y) * w * g[0] / den[0]   (g: incoming gradient, den: numden + 1, both device
 *        floats; gadd [B, C, HW] another gradient reaching x, or NULL). */
int g2s_weighted_l1_fwd2(const float *x, const float *y, const float *w, float *numden, int B, int C, int HW,
                         g2s_stream_t stream);
int g2s_weighted_l1_bwd2(const float *x, const float *y, const float *w, const float *g, const float *den,
                         const float *gadd, float *gx, int B, int C, int HW, g2s_stream_t stream);
/* One launch per level of the discriminator-feature loss's backward (losses._DFeatureL1; GAN2Shape/losses.py:11-36 on
 * stylegan2-pytorch/model.py:679-697):  gx = (gadd + gadd2) * add_scale + [x != NULL] sign(x - y) w g[0] / den[0] — the
 * residual join of the block above with this level's masked-L1 gradient — and, optionally, gx_gate = gx * gain *
 * (gate_ref > 0 ? 1 : slope), the gradient behind the block's activated conv2.  Any of gadd2, gx, gx_gate may be NULL. */
int g2s_weighted_l1_bwd3(const float *x, const float *y, const float *w, const float *g, const float *den,
                         const float *gadd, const float *gadd2, float add_scale, float *gx, const float *gate_ref,
                         float slope, float gain, float *gx_gate, int B, int C, int HW, g2s_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Fused renderer geometry / loss glue (csrc/geometry.hip).  All f32, device pointers; K is a HOST
 * pointer to 9 floats.  rays [H*W, 3] = K^-1 (u, v, 1)^T per pixel (renderer.py:74-80); R [B,3,3],
 * t [B,3]; rot_center_depth = z of the rotation centre (renderer.py:64-69).
 *   g2s_view_transform_*  view [B,6] -> R = Rz Ry Rx, t  (model.py:330-335 + renderer/utils.py:33-73)
 *   g2s_warp_verts_*      verts [B,H*W,3] = R (d ray - c) + c + t          (renderer.py:90-95)
 *   g2s_inv_warp_grid_*   grid [B,H,W,2] = normalise(K (R^T (d ray - t - c) + c))  (renderer.py:97-114)
 *   g2s_smooth_loss_*     mean|dx2| + mean|dxdy| + mean|dydx| + mean|dy2| of p [N,H,W]  (losses.py:54-79)
 * Backward entry points write gdepth / gview / gp fully; gRt [B,12] (9 of R, 3 of t; may be NULL)
 * is zero-filled by the callee then accumulated with one float atomic per workgroup.
 * ---------------------------------------------------------------------------------------- */
int g2s_view_transform_fwd(const float *view, float rot_scale, float txy_scale, float tz_scale,
                           float *R, float *t, int B, g2s_stream_t stream);
int g2s_view_transform_bwd(const float *view, float rot_scale, float txy_scale, float tz_scale,
                           const float *gR, const float *gt, float *gview, int B, g2s_stream_t stream);
int g2s_warp_verts_fwd(const float *depth, const float *rays, const float *R, const float *t,
                       float rot_center_depth, float *verts, int B, int P, g2s_stream_t stream);
int g2s_warp_verts_bwd(const float *depth, const float *rays, const float *R, const float *gverts,
                       float rot_center_depth, float *gdepth, float *gRt, int B, int P,
                       g2s_stream_t stream);
int g2s_inv_warp_grid_fwd(const float *depth, const float *rays, const float *R, const float *t,
                          const float *K, float rot_center_depth, float *grid, int B, int H, int W,
                          g2s_stream_t stream);
int g2s_inv_warp_grid_bwd(const float *depth, const float *rays, const float *R, const float *t,
                          const float *K, float rot_center_depth, const float *ggrid, float *gdepth,
                          float *gRt, int B, int H, int W, g2s_stream_t stream);
/*   g2s_normal_*      normal [B,H,W,3] from depth (renderer.py:127-139), gradient to depth
 *   g2s_shading_*     light [B,4] (a, b, dx, dy raw), normal [Bn,H,W,3], albedo [Ba,3,H,W] (Bn, Ba in
 *                     {1, B}) -> diffuse [B,1,H,W], texture [B,3,H,W] (model.py:347-360); backward
 *                     writes per-sample gnormal [B,H,W,3], galbedo [B,3,H,W] and glight [B,4]. */
int g2s_normal_fwd(const float *depth, const float *rays, float *normal, int B, int H, int W,
                   g2s_stream_t stream);
int g2s_normal_bwd(const float *depth, const float *rays, const float *gnormal, float *gdepth, int B,
                   int H, int W, g2s_stream_t stream);
int g2s_shading_fwd(const float *normal, const float *light, const float *albedo, float *diffuse,
                    float *texture, int B, int Bn, int Ba, int P, g2s_stream_t stream);
int g2s_shading_bwd(const float *normal, const float *light, const float *albedo,
                    const float *gdiffuse, const float *gtexture, float *gnormal, float *galbedo,
                    float *glight, int B, int Bn, int Ba, int P, g2s_stream_t stream);
int g2s_smooth_loss_fwd(const float *p, float *loss, int N, int H, int W, g2s_stream_t stream);
int g2s_smooth_loss_bwd(const float *p, const float *gloss, float *gp, int N, int H, int W,
                        g2s_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm + (leaky-)ReLU of the depth / albedo nets (csrc/groupnorm.hip), replacing the
 * nn.GroupNorm -> nn.ReLU / nn.LeakyReLU(0.2) pairs of GAN2Shape/networks.py:88-127.
 *   y = act((x - mean_g) * rstd_g * gamma[c] + beta[c]),  rstd_g = 1 / sqrt(var_g + eps) (biased var)
 * x, y, gy, dx [B, C, HW] f32 (HW % 4 == 0); gamma, beta, dgamma, dbeta [C]; mean, rstd [B, G]
 * (written by fwd, read by bwd); act 0: none, 1: leaky-ReLU(alpha) (alpha = 0: ReLU), the
 * backward taking the activation's slope from the sign of y.  workspace: at least
 * g2s_groupnorm_workspace_floats(B, C, HW, G) floats, scratch of one call.  No atomics.
 * ---------------------------------------------------------------------------------------- */
size_t g2s_groupnorm_workspace_floats(int B, int C, int HW, int G);
int g2s_groupnorm_act_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean,
                          float *rstd, float *workspace, int B, int C, int HW, int G, float eps,
                          int act, float alpha, g2s_stream_t stream);
int g2s_groupnorm_act_bwd(const float *gy, const float *y, const float *x, const float *gamma,
                          const float *mean, const float *rstd, float *dx, float *dgamma, float *dbeta,
                          float *workspace, int B, int C, int HW, int G, int act, float alpha,
                          g2s_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* G2S_H */
