/*
 * stmi.h -- C-ABI of libstmi355.so: gfx950 (MI355X) kernels for the
 * style-transfer hot path of loveFeynman/style-transfer.
 *
 * The reference has no FFI of its own: every FLOP of its hot path runs inside
 * TensorFlow ops called from Python.  Each entry point below replaces one such
 * TensorFlow op call site (cited as reference file:line, paths relative to the
 * reference's src/ directory) and is what a ctypes binding in the reference's
 * Python would call instead.
 *
 * Conventions
 *   - plain C symbols, plain pointers and sizes; no C++ or torch types.
 *   - every function returns 0 (STMI_OK) or a negative error code and leaves a
 *     thread-local message readable through stmi_last_error().
 *   - all tensor pointers are DEVICE pointers owned by the caller; activations
 *     are NHWC, contiguous, in `dtype` (STMI_F32 or STMI_BF16) unless a
 *     parameter is declared `float*`.
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous and
 *     never synchronise the device.  Functions are re-entrant.
 *   - TensorFlow 'SAME' padding everywhere: out = ceil(in/stride),
 *     pad_before = max((out-1)*stride + k - in, 0) / 2.
 */
#ifndef STMI_H
#define STMI_H

#ifdef __cplusplus
extern "C" {
#endif

#define STMI_OK 0
#define STMI_ERR_INVALID (-1)
#define STMI_ERR_HIP (-2)

#define STMI_F32 0
#define STMI_BF16 1

#define STMI_ACT_NONE 0
#define STMI_ACT_RELU 1
#define STMI_ACT_SIGMOID 2

/* which GEMM a packed weight buffer feeds */
#define STMI_PACK_FWD 0   /* conv forward; transposed-conv backward-data */
#define STMI_PACK_DGRAD 1 /* conv backward-data; transposed-conv forward */
#define STMI_PACK_FWD_ROWOUT 2 /* conv forward producing (kx, co) row patches, kw*co <= 32: packed like
                                * the STMI_PACK_FWD weights of a [kh,1,ci,32] conv (see row patches below) */
#define STMI_PACK_FWD_RGB 3    /* forward weights of a 3-channel-input k x k stride-1 convolution for
                                  stmi_conv2d_rgb_nhwc_fwd: co rows (co a multiple of 16) of
                                  round_up(4*k*(k+1), 32) K elements, k = (ky*(k+1) + kx)*4 + c,
                                  zero where kx >= k or c == 3 */
#define STMI_PACK_DGRAD_RGB 4  /* backward-data weights of a k x k stride-1 convolution ci -> 3 for
                                  stmi_conv2d_rgb_nhwc_dgrad: ci rows of the same K layout, built from the
                                  flipped kernel with its channel axes exchanged */

/* ---- library ---------------------------------------------------------- */
int stmi_version(void);
const char* stmi_last_error(void);
/* writes e.g. "gfx950:sramecc+:xnack-" of the current device */
int stmi_device_arch(char* buf, int buflen);

/* Profiling aid: the name a profiler shows for the kernel instantiation that this thread's
 * most recent convolution forward / backward-data call launched (empty before the first). */
int stmi_last_conv_kernel(char* buf, int buflen);
/* The same for this thread's most recent weight-gradient / Gram call: "wgrad_glds_kernel" (LDS-DMA operands,
 * channel counts that are multiples of 64 bytes) or "wgrad_kernel" (register-staged, any channel count). */
int stmi_last_wgrad_kernel(char* buf, int buflen);

/* ---- weight packing ---------------------------------------------------- */
/* Weights live as fp32 HWIO [kh,kw,ci,co] (the layout of the reference's
 * tf.Variable, model_utilities.py:236, and of the transposed .mat kernels,
 * vgg.py:92-94).  The GEMM kernels read a K-contiguous, zero-padded copy in
 * the compute dtype.  stmi_conv_packed_elems gives its size in elements. */
long stmi_conv_packed_elems(int kh, int kw, int ci, int co, int mode);
int stmi_pack_conv_weights(const float* w_hwio, void* packed, int kh, int kw, int ci, int co,
                           int mode, int dtype, void* stream);

/* Packs n weight tensors that live at element offsets src_off of one fp32 buffer into element
 * offsets dst_off of one packed buffer, in one launch (descs is a HOST array, read during the call). */
typedef struct {
    long src_off, dst_off;
    int kh, kw, ci, co, mode, reserved;
} stmi_pack_desc;
int stmi_pack_conv_weights_multi(const float* src_base, void* dst_base, const stmi_pack_desc* descs, int n,
                                 int dtype, void* stream);

/* ---- convolution ------------------------------------------------------- */
/* tf.nn.conv2d(x, W, [1,s,s,1], 'SAME')            (model_utilities.py:236-237)
 * + tf.nn.bias_add                                 (model_utilities.py:163-164)
 * + tf.nn.relu                                     (vgg.py:48, :98)
 * x [n,h,w,ci] -> y [n,ceil(h/s),ceil(w/s),co].  bias may be NULL.  If y_pre is
 * not NULL it receives the value before the ReLU (the pre-ReLU VGG taps of
 * vgg.py:104-105).
 * ws / ws_floats: optional fp32 workspace (16-byte aligned; NULL / 0: none).  A 3x3 stride-1 layer whose output has
 * too few tiles to fill the chip (VGG conv5_x on 16 x 16 maps: 64 tiles of 144 K steps for 256 CUs) then splits
 * its K loop over several blocks per tile: every slice leaves raw fp32 sums in a plane of the workspace and a second
 * launch of the same call adds the planes in slice order and applies bias / ReLU / copies.  Results differ from the
 * unsplit launch only in the order of fp32 additions and are bit-reproducible.  stmi_conv2d_ksplit_ws_floats gives
 * the floats with which a geometry splits (0: it never does; dgrad != 0 asks for stmi_conv2d_nhwc_dgrad); a smaller
 * workspace means fewer slices or none, never an error. */
long stmi_conv2d_ksplit_ws_floats(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype, int dgrad);

/* conv_block / residual_block at inference (full_graph_stylizer_network.py:47-57: conv -> instance norm -> ReLU ->
 * conv -> instance norm): the SECOND convolution WITHOUT the normalised tensor in between ever being written:
 * y = conv3x3(relu(IN(x))), where x is the first convolution's raw output, in_tile_stats the tile sums that
 * convolution left (in_tiles rows per sample), in_scale (> 0, every element) and in_shift the norm's parameters.
 * tile_stats receives the tile sums of y for the norm behind the call (stmi_conv2d_absorb_norm_tiles rows per
 * sample); in_stats [n][ci][2] receives (mean, rstd) of the absorbed norm.
 * How: with a = scale * rstd, b = shift - mean * a, t = -b / a,  relu(a x + b) = a (max(x, t) - t); the kernel reads
 * max(x, t[n][ci]) (one packed maximum per fragment register), takes t outside the image (the zero padding of the
 * normalised tensor), multiplies with per-sample weights W a and starts its accumulators from the constant term
 * -sum(W a t); weights, clamp values and constant come from a small launch in front (from the values as rounded to
 * bf16, so that what the clamp flattens is an exact zero).  bf16 only: 3x3 stride 1 with 128 input and output
 * channels on planes large enough for the persistent kernel (the 1080p residual layers), 3x3 stride 2 with 32 / 64
 * input channels on even sides (the down-sampling layers).  stmi_conv2d_absorb_norm_ws_bytes: bytes of the 16-byte
 * aligned workspace, 0 when the geometry is not supported. */
long stmi_conv2d_absorb_norm_ws_bytes(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int stmi_conv2d_absorb_norm_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int stmi_conv2d_nhwc_fwd_absorb_norm(const void* x, const float* in_tile_stats, int in_tiles, const float* in_scale,
                                     const float* in_shift, float eps, float* in_stats, const void* w_packed_fwd,
                                     void* y, float* tile_stats, void* ws, int n, int h, int w, int ci, int co,
                                     int kh, int kw, int stride, int dtype, void* stream);

/* vgg.py:99-103 (conv -> ReLU -> tf.nn.max_pool 2x2 / 2 'SAME'): stmi_conv2d_nhwc_fwd that also leaves the pooled
 * tensor y_pool [n, (ho+1)/2, (wo+1)/2, co] (= stmi_maxpool2x2_nhwc_fwd of y, bit for bit).  3x3 stride-1 layers
 * with at least 128 bytes of channels pool the tile they store inside the convolution's launch; every other
 * geometry runs the two launches. */
int stmi_conv2d_nhwc_fwd_pool(const void* x, const void* w_packed_fwd, const float* bias, void* y, void* y_pool,
                              int n, int h, int w, int ci, int co, int kh, int kw, int stride,
                              int relu, int dtype, void* stream);
int stmi_conv2d_nhwc_fwd(const void* x, const void* w_packed_fwd, const float* bias, void* y, void* y_pre,
                         float* ws, long ws_floats,
                         int n, int h, int w, int ci, int co, int kh, int kw, int stride,
                         int relu, int dtype, void* stream);
/* The same convolution (model_utilities.py:236-237) that also leaves per-tile sums of what it wrote, from
 * which the instance norm that follows it (model_utilities.py:238-239, :282-288) takes its mean and
 * variance without reading y again.  stmi_conv2d_tile_stats_tiles gives the number of tiles per sample
 * for a geometry (0: the kernel that runs this geometry emits no statistics; use stmi_conv2d_nhwc_fwd and
 * stmi_instance_norm_nhwc_stats); tile_stats holds n * 4 * co * tiles floats, laid out
 * [sample][tile][plane][channel], planes = shift k, sum(y - k), sum((y - k)^2), pixel count.  Consumed by
 * stmi_instance_norm_nhwc_apply_tiles.  y and y_pre must be 16-byte aligned. */
int stmi_conv2d_tile_stats_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
/* 1 when the statistics rows of this geometry cost the convolution next to nothing: planes large enough that ONE
 * sample already runs on the persistent patch kernel, which sums the rows it stores (on the matrix cores, or in
 * its loading waves while the computing waves store).  With them the convolution must be plain: no bias, ReLU
 * or second output; bf16 layers with 128-channel tiles leave TWO rows per tile, which
 * stmi_conv2d_tile_stats_tiles accounts for.  Otherwise the rows cost the convolution's epilogue up to a
 * microsecond per block and pay only while stmi_instance_norm_nhwc_apply_tiles can fold them without a
 * finalize launch (at most 32 per sample).  No counterpart in the reference (a tuning query). */
int stmi_conv2d_tile_stats_free(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int stmi_conv2d_nhwc_fwd_tile_stats(const void* x, const void* w_packed_fwd, const float* bias, void* y, void* y_pre,
                                    float* tile_stats,
                                    int n, int h, int w, int ci, int co, int kh, int kw, int stride,
                                    int relu, int dtype, void* stream);

/* The same convolution for a THREE-channel fp32 image, stride 1 (the stylizer's first layer,
 * full_graph_stylizer_network.py:47; VGG conv1_1, vgg.py:92-96), with the input transformed while it
 * is read: conv(x*scale - mean[c]) - scale = 1, mean = NULL is the plain cast; scale = 255/2 with the
 * mean pixel is vgg_preprocess_input (vgg.py:11-17); the zero padding applies to the transformed
 * image, as in the reference where the preprocessing is a separate op.  x_f32 [n,h,w,3] fp32 ->
 * y [n,h,w,co] in dtype; W packed with STMI_PACK_FWD_RGB.  Built for (k, co) = (9, 32), (3, 64), (3, 32). */
int stmi_conv2d_rgb_nhwc_fwd(const float* x_f32, float scale, const float* mean, const void* w_packed_rgb,
                             const float* bias, void* y, void* y_pre,
                             int n, int h, int w, int co, int k, int relu, int dtype, void* stream);

/* stmi_conv2d_rgb_nhwc_fwd (no bias, no ReLU) that also leaves the per-tile sums of y described at
 * stmi_conv2d_tile_stats_tiles for the instance norm that follows the stylizer's first layer
 * (model_utilities.py:235-242: conv, then instance_norm).  stmi_conv2d_rgb_tile_stats_tiles: rows per sample
 * (one per 8 x 32 block of output pixels; 0: none for this (k, co) - built for (9, 32)); tile_stats holds
 * n * tiles * 4 * co floats. */
int stmi_conv2d_rgb_tile_stats_tiles(int h, int w, int co, int k);
int stmi_conv2d_rgb_nhwc_fwd_tile_stats(const float* x_f32, float scale, const float* mean, const void* w_packed_rgb,
                                        void* y, float* tile_stats,
                                        int n, int h, int w, int co, int k, int dtype, void* stream);

/* The stylizer's first layer on 8-BIT frames as a decoder leaves them (image_utilities.py:229-236 divides by 255
 * on the host; here scale = 1/255 does it while the frame is read, so a frame crosses the host link as 3 bytes a
 * pixel instead of 12): x_u8 [n,h,w,3] unsigned char -> y [n,h,w,co] = conv(x * scale - mean[c]) in dtype, no bias
 * or ReLU; W packed with STMI_PACK_FWD_RGB.  tile_stats: NULL, or the per-tile sums described at
 * stmi_conv2d_rgb_nhwc_fwd_tile_stats.  Built for (k, co) = (9, 32). */
int stmi_conv2d_rgb_u8_nhwc_fwd(const unsigned char* x_u8, float scale, const float* mean, const void* w_packed_rgb,
                                void* y, float* tile_stats,
                                int n, int h, int w, int co, int k, int dtype, void* stream);

/* k x k stride-1 SAME convolution ci -> 3 in one launch (the stylizer's last layer,
 * full_graph_stylizer_network.py:78; with the flipped, transposed kernel it is the backward-data of VGG
 * conv1_1): x [n,h,w,ci] -> y [n,h,w,3] = conv(x) + addend (optional, shaped like y).  W packed with
 * STMI_PACK_FWD_ROWOUT; same values as stmi_conv2d_nhwc_fwd on those weights followed by
 * stmi_rows_to_pixels_sum, except that the intermediate stays in fp32.  stmi_conv2d_to3_supported
 * returns 1 for the (k, ci, dtype) it is built for: (9, 32), (3, 32) in both dtypes, (3, 64) in bf16. */
int stmi_conv2d_to3_supported(int k, int ci, int dtype);
int stmi_conv2d_to3_nhwc_fwd(const void* x, const void* w_packed_rowout, const void* addend, void* y,
                             int n, int h, int w, int ci, int k, int dtype, void* stream);

/* stmi_conv2d_to3_nhwc_fwd absorbing the instance norm + ReLU in front of it, as stmi_conv2d_nhwc_fwd_absorb_norm
 * does for the 3x3 layers (full_graph_stylizer_network.py:76-78: the last up-sampling block's norm and the 9x9
 * output layer): x is the raw output of the layer in front, in_tile_stats the tile sums it left; the patch goes to
 * LDS already clamped.  bf16, 9x9, 32 input channels; no tile sums of y (three channels: its norm runs its own
 * pass).  stmi_conv2d_to3_absorb_norm_ws_bytes: bytes of the 16-byte aligned workspace, 0: not supported. */
long stmi_conv2d_to3_absorb_norm_ws_bytes(int n, int h, int w, int ci, int k, int dtype);
int stmi_conv2d_to3_nhwc_fwd_absorb_norm(const void* x, const float* in_tile_stats, int in_tiles, const float* in_scale,
                                         const float* in_shift, float eps, float* in_stats, const void* w_packed_rowout,
                                         void* y, void* ws, int n, int h, int w, int ci, int k, int dtype, void* stream);

/* Gradient w.r.t. the input of a k x k stride-1 convolution that WRITES three channels (the stylizer's
 * last layer, full_graph_stylizer_network.py:78): dy [n,h,w,3] in dtype -> dx [n,h,w,ci]; W packed with
 * STMI_PACK_DGRAD_RGB.  Built for (k, ci) = (9, 32). */
int stmi_conv2d_rgb_nhwc_dgrad(const void* dy, const void* w_packed_dgrad_rgb, void* dx,
                               int n, int h, int w, int ci, int k, int dtype, void* stream);

/* Gradient of the above w.r.t. x (what tf.gradients emits as Conv2DBackpropInput
 * for the ops at model_utilities.py:237 and :164):
 *   dx = alpha * backprop_input(dy) ; dx = 0 where relu_mask <= 0 ; dx += addend.
 * dy [n,ho,wo,co] -> dx [n,h,w,ci].  relu_mask and addend are optional tensors
 * shaped like dx (addend may alias dx).  ws / ws_floats: optional split-K workspace as at stmi_conv2d_nhwc_fwd. */
int stmi_conv2d_nhwc_dgrad(const void* dy, const void* w_packed_dgrad, void* dx,
                           const void* relu_mask, const void* addend, float alpha,
                           float* ws, long ws_floats,
                           int n, int h, int w, int ci, int co, int kh, int kw, int stride,
                           int dtype, void* stream);

/* Backward-data of a 3x3 stride-1 layer (as stmi_conv2d_nhwc_dgrad) that also leaves, per tile of dx, the sums the
 * backward pass of the instance norm BELOW needs (the norm whose output gradient is dx, model_utilities.py:238-239
 * seen from the back): sum(g * act'(z)) and sum(g * act'(z) * xhat) over the tile's pixels, g = dx after mask and
 * addend, xhat = (bn_x - mean) * rstd, z = bn_scale * xhat + bn_shift; bn_x is that norm's input (shaped like dx),
 * bn_stats its [n,c,2] statistics.  rows: fp32 [n, tiles, 2, ci], tiles from stmi_conv2d_dgrad_in_sums_tiles (0: the
 * kernel for this geometry leaves no sums).  Consumed by stmi_instance_norm_nhwc_bwd_rows, which then needs no
 * reduction pass over x and dy. */
int stmi_conv2d_dgrad_in_sums_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int stmi_conv2d_nhwc_dgrad_in_sums(const void* dy, const void* w_packed_dgrad, void* dx, const void* relu_mask,
                                   const void* addend, float alpha,
                                   const void* bn_x, const float* bn_stats, const float* bn_scale, const float* bn_shift,
                                   int bn_act, float* rows,
                                   int n, int h, int w, int ci, int co, int kh, int kw, int stride,
                                   int dtype, void* stream);
/* Gradient w.r.t. W (Conv2DBackpropFilter): dw fp32 [kh,kw,ci,co].  accumulate == 0: dw is
 * overwritten; accumulate != 0: the gradient is ADDED to what dw holds.
 * The sum over pixels is cut into slices that run on different blocks; the slices' partial results go to
 * planes of `ws` (fp32, 16-byte aligned, stmi_conv2d_wgrad_ws_floats(...) floats; may be NULL when that is 0)
 * and are added up in slice order by a second launch of the same call: no float atomics, the result is
 * bit-reproducible from run to run. */
long stmi_conv2d_wgrad_ws_floats(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int stmi_conv2d_nhwc_wgrad(const void* x, const void* dy, float* dw, float* ws,
                           int n, int h, int w, int ci, int co, int kh, int kw, int stride,
                           int accumulate, int dtype, void* stream);

/* tf.nn.conv2d_transpose(x, W, [n,s*h,s*w,co], [1,s,s,1], 'SAME') with
 * W [kh,kw,co,ci]                                   (model_utilities.py:246-248)
 * x [n,h,w,ci] -> y [n,s*h,s*w,co].  W is packed with STMI_PACK_DGRAD using
 * (ci_pack, co_pack) = (co, ci): the op is the backward-data of a conv whose
 * HWIO weight is W. */
int stmi_conv2d_transpose_nhwc_fwd(const void* x, const void* w_packed_dgrad, void* y,
                                   int n, int h, int w, int ci, int co, int kh, int kw, int stride,
                                   int dtype, void* stream);
/* The same transposed convolution, also leaving the per-tile sums of y (layout and consumer as described at
 * stmi_conv2d_tile_stats_tiles) for the instance norm that follows it in deconv_block
 * (model_utilities.py:245-253: conv2d_transpose, then instance_norm): the norm then needs no statistics pass
 * over y.  stmi_conv2d_transpose_tile_stats_tiles: rows per sample, 0 when the kernel that runs this geometry
 * leaves none (then use stmi_conv2d_transpose_nhwc_fwd); tile_stats holds n * tiles * 4 * co floats. */
int stmi_conv2d_transpose_tile_stats_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int stmi_conv2d_transpose_nhwc_fwd_tile_stats(const void* x, const void* w_packed_dgrad, void* y, float* tile_stats,
                                              int n, int h, int w, int ci, int co, int kh, int kw, int stride,
                                              int dtype, void* stream);

/* stmi_conv2d_transpose_nhwc_fwd_tile_stats absorbing the instance norm + ReLU in front of it, as
 * stmi_conv2d_nhwc_fwd_absorb_norm does for the forward layers (full_graph_stylizer_network.py:72-76: the first
 * up-sampling block's norm and the second block's transposed convolution): x is the raw output of the layer in
 * front, in_tile_stats the tile sums it left.  bf16, 3x3 stride 2, 64 input channels.
 * stmi_conv2d_transpose_absorb_norm_ws_bytes: bytes of the 16-byte aligned workspace, 0: not supported. */
long stmi_conv2d_transpose_absorb_norm_ws_bytes(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int stmi_conv2d_transpose_nhwc_fwd_absorb_norm(const void* x, const float* in_tile_stats, int in_tiles,
                                               const float* in_scale, const float* in_shift, float eps, float* in_stats,
                                               const void* w_packed_dgrad, void* y, float* tile_stats, void* ws,
                                               int n, int h, int w, int ci, int co, int kh, int kw, int stride,
                                               int dtype, void* stream);
/* dy [n,s*h,s*w,co] -> dx [n,h,w,ci]; W packed with STMI_PACK_FWD, (ci_pack, co_pack) = (co, ci). */
int stmi_conv2d_transpose_nhwc_dgrad(const void* dy, const void* w_packed_fwd, void* dx,
                                     const void* addend,
                                     int n, int h, int w, int ci, int co, int kh, int kw, int stride,
                                     int dtype, void* stream);
/* dw fp32 [kh,kw,co,ci]; accumulate and ws (stmi_conv2d_transpose_wgrad_ws_floats floats) as above. */
long stmi_conv2d_transpose_wgrad_ws_floats(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int stmi_conv2d_transpose_nhwc_wgrad(const void* x, const void* dy, float* dw, float* ws,
                                     int n, int h, int w, int ci, int co, int kh, int kw, int stride,
                                     int accumulate, int dtype, void* stream);

/* ---- instance norm ----------------------------------------------------- */
/* tf.nn.moments(x,[1,2]) ; scale*(x-mu)/(var+eps)**.5 + shift   (model_utilities.py:282-288)
 * followed by the block's activation (model_utilities.py:240-241) and, for the
 * second conv of a residual block, the skip add (model_utilities.py:278).
 *
 * stats  : stats[n,c,2] (fp32) = (mean, 1/sqrt(var+eps)) per sample and channel; ws is
 *          scratch of stmi_instance_norm_ws_floats floats (per-block partial sums).
 * apply  : y = act(scale*xhat + shift) [+ residual]; optional fp32 copy y_f32.
 * bwd    : from dy (gradient w.r.t. y) to dx, and the parameter gradients dscale[c],
 *          dshift[c] (fp32; written, or added to what they hold when accumulate != 0;
 *          both NULL: not computed here, see stmi_instance_norm_param_grads), summed over
 *          samples in a fixed order by a few extra blocks of the same launch.  The
 *          activation gradient is recomputed from x, so y is not needed.
 * Nothing in these passes uses float atomics: results are bit-reproducible, and a sample's
 * result does not depend on the batch it is part of. */
long stmi_instance_norm_ws_floats(int n, int h, int w, int c);
int stmi_instance_norm_nhwc_stats(const void* x, float* ws, float* stats, float eps,
                                  int n, int h, int w, int c, int dtype, void* stream);
/* stats + apply as one call of two launches: the apply blocks fold the statistics pass's partial sums
 * themselves (no separate finalize launch); stats[n,c,2] is written as by stmi_instance_norm_nhwc_stats
 * (same values up to the order of a few fp32 additions).  y must not alias x (the apply blocks read x's
 * first pixel, the statistics' shift, while others already write y). */
int stmi_instance_norm_nhwc_fwd(const void* x, float* ws, float* stats, float eps,
                                const float* scale, const float* shift, const void* residual,
                                void* y, float* y_f32, int act, int n, int h, int w, int c,
                                int dtype, void* stream);
/* The same normalisation with the statistics taken from the per-tile sums that
 * stmi_conv2d_nhwc_fwd_tile_stats left behind while it wrote x: the separate statistics pass over x
 * (one read of the tensor) is not needed.  `tiles` is the row length stmi_conv2d_tile_stats_tiles
 * returned.  stats[n,c,2] is WRITTEN (mean, 1/sqrt(var+eps)) for the backward pass. */
int stmi_instance_norm_nhwc_apply_tiles(const void* x, const float* tile_stats, int tiles, float* stats, float eps,
                                        const float* scale, const float* shift, const void* residual,
                                        void* y, float* y_f32, int act, int n, int h, int w, int c,
                                        int dtype, void* stream);
/* stmi_instance_norm_nhwc_stats without a workspace argument and without any allocation.  It has no
 * counterpart in the reference: it exists for callers that cannot allocate device memory (code under
 * HIP-graph capture, serving loops and harnesses that forbid hipMalloc).  The per-block partial sums
 * go to a fixed 16 MiB array in device memory that every GPU the process uses holds until the process
 * ends; n * c must not exceed STMI_IN_STATIC_MAX_NC (otherwise STMI_ERR_INVALID and a message).
 * Results are bit-identical to stmi_instance_norm_nhwc_stats.  All calls on one GPU share that GPU's
 * array: calls that overlap in time on the same GPU (different streams or threads) race.  The first
 * call on a device looks up the array's device address and must happen outside graph capture. */
#define STMI_IN_STATIC_MAX_NC 8192
int stmi_instance_norm_nhwc_stats_static(const void* x, float* stats, float eps,
                                         int n, int h, int w, int c, int dtype, void* stream);
int stmi_instance_norm_nhwc_apply(const void* x, const float* stats, const float* scale, const float* shift,
                                  const void* residual, void* y, float* y_f32,
                                  int act, int n, int h, int w, int c,
                                  int dtype, void* stream);
/* stmi_instance_norm_nhwc_apply for the LAST layer of a service that hands 8-bit frames back: y_u8 [n,h,w,c] =
 * (unsigned char)(act(scale*xhat + shift) * 255), truncated as the reference's save path does
 * ((img * 255).astype(uint8), image_utilities.py:195-196, :223-227) and clamped to 0 .. 255.  Built for c = 3. */
int stmi_instance_norm_nhwc_apply_u8(const void* x, const float* stats, const float* scale, const float* shift,
                                     unsigned char* y_u8, int act, int n, int h, int w, int c,
                                     int dtype, void* stream);
int stmi_instance_norm_nhwc_bwd(const void* x, const void* dy, const float* stats,
                                const float* scale, const float* shift,
                                void* dx, float* dscale, float* dshift, int accumulate, float* ws,
                                float eps, int act, int n, int h, int w, int c,
                                int dtype, void* stream);

/* stmi_instance_norm_nhwc_bwd with the reduction pass replaced by `nrows` partial rows per sample
 * ([n, nrows, 2, c]: sum(g), sum(g * xhat)) that stmi_conv2d_nhwc_dgrad_in_sums left behind. */
int stmi_instance_norm_nhwc_bwd_rows(const void* x, const void* dy, const float* stats,
                                     const float* scale, const float* shift,
                                     void* dx, float* dscale, float* dshift, int accumulate,
                                     const float* rows, int nrows,
                                     int act, int n, int h, int w, int c, int dtype, void* stream);

/* The parameter gradients of the norm from the partial rows of its backward reduction, as a launch of its
 * own (a trainer runs it beside the data-gradient chain): dshift[c] = sum over samples and rows of
 * rows[.., 0, c], dscale[c] = of rows[.., 1, c], in one fixed order; accumulate != 0 adds to what they hold.
 * rows: fp32 [n, nrows, 2, c] -- what stmi_conv2d_nhwc_dgrad_in_sums left, or the first n * nrows * 2 * c
 * floats of the workspace stmi_instance_norm_nhwc_bwd used, nrows = stmi_instance_norm_bwd_rows_count(...)
 * (the reduction pass's rows per sample for that geometry). */
int stmi_instance_norm_bwd_rows_count(int n, int h, int w, int c, int dtype);
int stmi_instance_norm_param_grads(const float* rows, int nrows, int n, int c, float* dscale, float* dshift,
                                   int accumulate, void* stream);

/* ---- VGG pieces -------------------------------------------------------- */
/* x*255/2 - mean                                    (vgg.py:11-17)
 * x fp32 [count/c, c] -> y in dtype; y = x*scale - mean[c]. */
int stmi_scale_shift(const float* x, const float* mean, void* y, float scale, long count, int c,
                     int dtype, void* stream);
/* tf.nn.max_pool2d(2,2,'SAME')                      (model_utilities.py:166-167)
 * x [n,h,w,c] -> y [n,ceil(h/2),ceil(w/2),c] */
int stmi_maxpool2x2_same_fwd(const void* x, void* y, int n, int h, int w, int c, int dtype, void* stream);
/* MaxPoolGrad fused with the ReLU gradient of the layer that fed the pool:
 * dx = dy routed to the first maximum of each window, zero where x <= 0
 * (x is the post-ReLU pool input). */
int stmi_maxpool2x2_same_bwd(const void* x, const void* dy, void* dx, int n, int h, int w, int c,
                             int dtype, void* stream);

/* ---- Gram matrix ------------------------------------------------------- */
/* tf.linalg.einsum('bijc,bijd->bcd', F, F) / (H*W)   (model_utilities.py:380-384)
 * f [n,h,w,c] -> g fp32 [n,c,c], overwritten.  ws: workspace of stmi_gram_ws_floats(...) floats for the
 * partial sums of the pixel slices (as at stmi_conv2d_nhwc_wgrad; NULL when the query returns 0). */
long stmi_gram_ws_floats(int n, int h, int w, int c, int dtype);
int stmi_gram_nhwc_fwd(const void* f, float* g, float* ws, int n, int h, int w, int c, int dtype, void* stream);
/* The same, ADDED into what g holds. */
int stmi_gram_nhwc_fwd_add(const void* f, float* g, float* ws, int n, int h, int w, int c, int dtype, void* stream);
/* df[p][c] (+)= sum_d F[p][d] * S[b][c][d], S = (dG + dG^T)/(H*W) in dtype, in the
 * zero-padded K-contiguous layout stmi_gram_mse_fwd_bwd writes
 * (stmi_gram_packed_elems elements).  accumulate != 0 adds into df.  c % 8 == 0. */
long stmi_gram_packed_elems(int n, int c);
int stmi_gram_nhwc_bwd(const void* f, const void* s, void* df, int accumulate,
                       int n, int h, int w, int c, int dtype, void* stream);

/* ---- losses ------------------------------------------------------------ */
/* A loss call leaves its term as PARTIAL SUMS, one per block of its launch, in a "loss slot": a caller-owned
 * fp32 buffer of STMI_LOSS_SLOT_FLOATS floats, 16-byte aligned, which the call overwrites entirely (unused
 * entries are cleared).  stmi_loss_slots_sum then adds the slots of each term up in a fixed order:
 *   out[t] = sum over the slots first[t] .. first[t+1]-1   for t in [0, nterms)
 * (`first`: HOST array of nterms + 1 slot indices into `slots`, read during the call; nterms <=
 * STMI_LOSS_TERMS_MAX).  No float atomics: a loss value is bit-reproducible from run to run.  A step's slots can
 * be summed by ONE launch, off the critical path, once all loss kernels of the step have been issued.
 *
 * tf.reduce_mean((a-b)**2) terms of sum_mse          (model_utilities.py:387-388)
 * The slot receives loss_scale * sum((a-b)^2); grad (may be NULL for stmi_mse_fwd_bwd) = grad_scale * (a-b).
 * gram form: g fp32 [n,c,c], target fp32 [c,c] broadcast over n; s_out in dtype,
 * stmi_gram_packed_elems(n,c) elements. */
#define STMI_LOSS_SLOT_FLOATS 1024
#define STMI_LOSS_TERMS_MAX 8
int stmi_gram_mse_fwd_bwd(const float* g, const float* target, void* s_out, float* slot,
                          float loss_scale, float grad_scale, int n, int c, int dtype, void* stream);
int stmi_mse_fwd_bwd(const void* a, const void* b, void* grad, float* slot,
                     float loss_scale, float grad_scale, long count, int dtype, void* stream);
/* total_variation_loss                               (model_utilities.py:369-377)
 * img fp32 [n,h,w,c]; the slot receives weight*(mean(dx^2)+mean(dy^2)); grad (dtype) =
 * weight * d/d img, overwritten. */
int stmi_tv_loss_fwd_bwd(const float* img, void* grad, float* slot, float weight,
                         int n, int h, int w, int c, int dtype, void* stream);
/* add_n over the terms of sum_mse / total_loss      (model_utilities.py:387-399) */
int stmi_loss_slots_sum(const float* slots, const int* first, int nterms, float* out, void* stream);

/* ---- step begin -------------------------------------------------------- */
/* Clears up to STMI_ZERO_MULTI_MAX fp32 buffers (ptrs[i], counts[i] floats) in ONE launch and, when
 * step_dev is not NULL, advances that device step counter by one (then stmi_adam_tf1 is called with
 * t < 0 in the same step).  No reference counterpart: in the reference a session run starts from
 * fresh gradient accumulators; here the accumulators of a step (losses, flat gradient, Gram matrices)
 * are caller-owned buffers, and clearing them one launch each would cost more than the work. */
#define STMI_ZERO_MULTI_MAX 16
int stmi_zero_multi(float* const* ptrs, const long* counts, int n, int* step_dev, void* stream);

/* ---- optimizer --------------------------------------------------------- */
/* tf.train.AdamOptimizer update (call sites full_graph_stylizer_network.py:125, :278):
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t*m/(sqrt(v)+eps).
 * step_dev (device int32) gives t; it is incremented first when the host
 * argument t >= 0 and used as it stands when t < 0 (a further slice of the same
 * parameter set within one step).  Pass step_dev = NULL to use the host value
 * t >= 1 instead.  grad_scale multiplies g first (1/world for data-parallel
 * sums). */
int stmi_adam_tf1(float* p, const float* g, float* m, float* v, long count,
                  float lr, float beta1, float beta2, float eps, float grad_scale,
                  int* step_dev, int t, void* stream);

/* ---- row patches for layers with < 8 channels on one side (the 9x9 RGB layers) ----------
 * A kw x c window of an NHWC tensor is kw*c contiguous elements.  Copying it once per pixel
 * into a 32-wide row turns the [kh,kw] convolution into kh vertical taps with 32 reduction
 * channels (or 32 outputs), which the GEMM kernels run without padding waste:
 *   conv c->co   : rows = pixels_to_rows(x, dx0=-pad_left); y = conv2d_fwd(rows, [kh,1,32,co]) with the
 *                  STMI_PACK_FWD weights of the original kernel; dW = rows_wgrad_to_hwio(mode 0) of
 *                  conv2d_wgrad(rows, dy, [kh,1,32,co])
 *   conv ci->c   : t = conv2d_fwd(x, [kh,1,ci,32]) with STMI_PACK_FWD_ROWOUT weights;
 *                  y = rows_to_pixels_sum(t, dx0=-pad_left);
 *                  r = pixels_to_rows(dy, dx0=pad_left-(kw-1)); dx = conv2d_dgrad(r, [kh,1,ci,32]) with the
 *                  STMI_PACK_DGRAD weights of the original kernel; dW = rows_wgrad_to_hwio(mode 1) of
 *                  conv2d_wgrad(x, r, [kh,1,ci,32])
 * pixels_to_rows   : rows[n,y,x,e] = x[n,y,x+dx0+e/c,e%c] for e < kw*c and an in-range column, else 0
 * rows_to_pixels_sum: y[n,yy,x,co] = sum_kx rows[n,yy,x+dx0+kx,kx*c+co] over in-range columns
 *                    (+ addend[n,yy,x,co] when addend is not NULL) */
int stmi_pixels_to_rows(const void* x, void* rows, int n, int h, int w, int c, int kw, int dx0,
                        int dtype, void* stream);
int stmi_rows_to_pixels_sum(const void* rows, const void* addend, void* y, int n, int h, int w, int c, int kw, int dx0,
                            int dtype, void* stream);
/* d fp32 is the [kh,1,32,co] (mode 0) or [kh,1,ci,32] (mode 1) gradient of the row-patch conv;
 * dw fp32 [kh,kw,ci,co] is the gradient of the original kernel, overwritten. */
int stmi_rows_wgrad_to_hwio(const float* d, float* dw, int kh, int kw, int ci, int co, int mode, void* stream);

/* ---- small utilities --------------------------------------------------- */
/* dtype conversion between fp32 and `dtype`; to_f32 != 0 converts dtype -> fp32 */
int stmi_cast(const void* src, void* dst, long count, int dtype, int to_f32, void* stream);
/* tf.nn.relu / tf.nn.sigmoid applied on their own (model_utilities.py:240-241 when a block
 * has no instance norm to fuse them into); y may alias x. */
int stmi_activation(const void* x, void* y, long count, int act, int dtype, void* stream);
/* tf.clip_by_value(x, lo, hi) in place, fp32           (model_utilities.py:365-366) */
int stmi_clip(float* x, long count, float lo, float hi, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STMI_H */
