/* srganfd.h -- C ABI of the MI355X (gfx950) SR-GAN-FD hot path.
 *
 * The reference (MiNeves00/SR-GAN-FD) has no native code and no FFI: its hot path is the
 * torch.nn forward/backward of BSRGAN/model.py (RRDBNet generator :311-384, U-Net discriminator
 * :91-167, VGG content loss :501-554) driven by BSRGAN/train_bsrgan.py:387-483.  This header is
 * therefore the boundary SURVEY.md 8(b) proposes: extern "C" entry points taking raw device
 * pointers, sizes, a plain-C argument struct and a hipStream_t (as void*), returning an int status
 * (0 = ok, <0 = error; never throws).  The caller owns every buffer, workspaces included.
 * Each entry point names the reference code it replaces.
 */
#ifndef SRGANFD_H
#define SRGANFD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SRGANFD_OK 0
#define SRGANFD_EINVAL (-1)   /* bad shape / argument            */
#define SRGANFD_EHIP (-2)     /* HIP runtime error (see last_error) */
#define SRGANFD_ENOSPC (-3)   /* workspace too small             */

/* element type of activations / packed weights */
#define SRGANFD_BF16 0
#define SRGANFD_F32 1
#define SRGANFD_F16 2   /* IEEE half: what the reference's amp.autocast() computes in on its GPU path (train_bsrgan.py:415-467) */

/* epilogue activation */
#define SRGANFD_ACT_NONE 0
#define SRGANFD_ACT_LRELU 1
#define SRGANFD_ACT_RELU 2

const char* srganfd_last_error(void);
/* Bumped whenever an exported signature or struct changes; the binding (sr_gan_fd_amd/_abi.py) reads this constant from this
 * header and refuses a library whose srganfd_abi_version() differs (a stale A/B build selected with SRGANFD_LIB, a prebuilt .so). */
#define SRGANFD_ABI_VERSION 7
int srganfd_abi_version(void);
/* dry run: entry points validate their arguments and build plans but launch nothing (used by the
 * CPU-only host-logic tests; never set in production). */
void srganfd_set_dry_run(int on);

/* A channel-slice view of an NHWC activation buffer: element (n,y,x,c) lives at
 * ptr[((n*H + y)*W + x)*cstride + c0 + c]. */
typedef struct {
  void* ptr;
  int32_t cstride; /* channels of the underlying buffer */
  int32_t c0;      /* first channel of the view */
  int32_t planar;  /* 0: NHWC, a pixel's cstride channels are contiguous.  1: per image, every group of 32 channels is its own
                      (H, W, 32) plane -- element (p, c) of an image at (c / 32) * H*W*32 + p * 32 + c % 32 (same bytes per image).
                      Only srganfd_conv2d and srganfd_conv2d_wgrad take planar views (the dense-block buffers: each 32-channel
                      chunk pass then reads whole contiguous 128-byte lines); c0 must be a multiple of 32. */
  int32_t pad_;
} srganfd_view;

/* Fused convolution (implicit GEMM on MFMA).  Replaces one nn.Conv2d call of
 * BSRGAN/model.py:42-46,102-135,325-355 together with the element-wise ops the reference runs
 * after it (LeakyReLU :48, `mul 0.2 / add identity` :59-60,85-86, torch.cat :55-58 -- the output is
 * written in place into a channel slice of the dense-block buffer, nearest upsample :372-374 --
 * fused into the input gather).  The same entry point runs the data-gradient pass (weights packed
 * transposed/flipped by srganfd_pack_weights) with the LeakyReLU-derivative mask in the epilogue.
 *
 *   v   = alpha * conv(x)[c] + bias[c]
 *   v   = act(v)
 *   v   = post_scale * v + r1_scale * r1[c] + r2_scale * r2[c]
 *   v  *= (mask[c] > 0 ? 1 : mask_slope)          if mask.ptr != NULL
 *   y[c] = v                                       for c < cout_store
 */
typedef struct {
  int32_t dtype;            /* SRGANFD_BF16 | SRGANFD_F16 | SRGANFD_F32 */
  int32_t n, h_in, w_in;    /* stored input dims */
  int32_t up;               /* 1: logical input = nearest-x2 upsample of the stored input */
  int32_t ksize, stride, pad; /* 3,1,1 | 4,2,1 | 1,1,0 */
  int32_t cin;              /* multiple of 32 */
  int32_t cout;             /* packed (padded) output channels, multiple of 32 */
  int32_t cout_store;       /* channels actually written (<= cout) */
  int32_t h_out, w_out;
  srganfd_view x, y, r1, r2, mask;
  const void* w_packed;     /* from srganfd_pack_weights */
  const float* bias;        /* [cout_store] fp32 or NULL */
  const float* alpha_dev;   /* optional device scalar multiplied into alpha (spectral norm 1/sigma) */
  float alpha, slope, post_scale, r1_scale, r2_scale, mask_slope;
  int32_t act;
  int32_t y_f32;             /* 1: y is fp32 regardless of dtype (final SR / logits) */
  /* Strided output (all 0 = dense).  Used by the data gradient of the 4x4 stride-2 convs
   * (model.py:103-114): each of the 4 output parity classes is a 2x2-tap stride-1 conv over dy
   * whose result pixel (oy,ox) is stored at (oy*out_sy+out_oy, ox*out_sx+out_ox) of an
   * out_h_full x out_w_full image; pad_y/pad_x replace `pad`; r1/r2/mask use the same addressing. */
  int32_t out_sy, out_sx, out_oy, out_ox, out_h_full, out_w_full, pad_y, pad_x;
  /* optional second output: post_scale * act(alpha*conv + bias) BEFORE the residual adds (the U-Net skip
   * adds of model.py:153,157,161 keep the LeakyReLU output so its derivative's sign is exact in backward) */
  srganfd_view y2;
  /* 0 / 1: one parity class per launch (out_oy, out_ox, pad_y, pad_x say which).  4: ALL FOUR classes of a 2x2 output stride in this
   * launch (16-bit dtypes, ksize 2 or 1, stride 1, out_sy = out_sx = 2): class (py,px) = (c >> 1, c & 1) uses out_oy = py, out_ox = px,
   * pad_y - py * class_pad_step, pad_x - px * class_pad_step (pad_y / pad_x = class (0,0)'s) and the packed operand at
   * w_packed + c * ksize^2 * cin * cout * 2 bytes -- the four packs follow each other; out_oy / out_ox are ignored.
   * class_pad_step = 1: data gradient of a 4x4 stride-2 pad-1 conv (pad 1 - p); 0: the 3x3 stride-2 / 2x2 stride-2 forms whose classes
   * share one window.  The four workgroups that read one patch of x are neighbours on one XCD, so x comes from HBM once, not four times. */
  int32_t out_classes, class_pad_step;
} srganfd_conv_args;

int srganfd_conv2d(const srganfd_conv_args* a, void* stream);
/* Name of the kernel template srganfd_conv2d dispatches these arguments to (validates them, launches nothing): the class label
 * of bench.py's per-kernel timing and of the rocprofv3 summaries under profiles/. */
int srganfd_conv2d_describe(const srganfd_conv_args* a, char* out, size_t out_len);

/* A whole dense block as ONE launch with LDS-resident activations (csrc/dense_chain.hip): replaces the n_layers srganfd_conv2d
 * launches of _ResidualDenseBlock.forward (BSRGAN/model.py:51-62; ESRGAN/model.py:49-60, Real_ESRGAN/model.py:131-142,
 * A-ESRGAN/model.py:441-452 are the same code) -- or of its data-gradient pass, which is the same dense structure over the stacked
 * output gradients -- when an image is at most one 16 x 16-pixel tile per compute unit (the reference's crop sizes: 32 ... 72 pixels
 * at batch 8-16; faster than the separate launches while the whole batch is at most one tile per compute unit; the library tiles
 * 8 x 16, else 12 x 16, instead when the batch is one pass that way too).  `layers` are exactly
 * the arguments those launches would get, in order: layer i (0-based) is a 16-bit 3x3 stride-1 pad-1 conv reading channels
 * [0, 64 + 32 i) of ONE buffer; all but the last write 32 channels at [64 + 32 i, 96 + 32 i) of that same buffer (bias / activation /
 * mask as given), the last one (64 output channels, r1 / r2 as given) writes anywhere else.  Per layer either a mask or residuals
 * (r2 only with r1); every channel offset a multiple of 32.  Results follow srganfd_conv2d's formula with the same accumulation order;
 * the whole batch is one launch (a workgroup walks the groups of images whose tiles are resident at once).  `workspace`:
 * srganfd_dense_chain_workspace_bytes() bytes of device memory that the caller zeroes ONCE at allocation and gives to one stream's
 * launches at a time (hand-off flags are epoch-valued from a counter in its header: nothing is zeroed per launch and a captured graph
 * replays correctly; word 0 counts hand-off waits that gave up after ~2 s -- always 0 in a correct run, results are wrong otherwise).
 * ONE such launch at a time per device: every workgroup of a pass must be resident (one per compute unit, all of its LDS), so two
 * concurrent launches -- two streams, or two processes sharing the GPU -- whose grids together exceed the compute units can starve
 * each other until the waits give up.  Other kernels beside it only delay it.
 * srganfd_dense_chain_check validates `layers` and the size limits (tiles of one image <= compute units, 16384 tiles per call)
 * without launching. */
int srganfd_dense_chain(const srganfd_conv_args* layers, int32_t n_layers, void* workspace, size_t workspace_bytes, void* stream);
int srganfd_dense_chain_check(const srganfd_conv_args* layers, int32_t n_layers);
size_t srganfd_dense_chain_workspace_bytes(void);

/* Weight packing: NCHW fp32 parameters -> MFMA B-fragment order (dtype bf16/f32).
 * A packed operand is a logical matrix W[tap][k][n]; its k range is assembled from up to 5
 * segments, each taken from one parameter tensor either in forward orientation
 * (W[t][k][n] = src[n+co_off][k-k_lo+ci_off][t]) or in data-gradient orientation
 * (W[t][k][n] = src[k-k_lo+co_off][n+ci_off][KT-1-t]).  The 5-segment form is the backward of
 * _ResidualDenseBlock (BSRGAN/model.py:51-62) expressed as a dense block over the stacked output
 * gradients.  Job tables live in DEVICE memory and hold offsets only (floats from `params`,
 * floats from `scalars`, bytes from `packed`), so one table serves every step. */
typedef struct {
  int64_t src_off;        /* floats from params base; tensor is (co_src, ci_src, k, k) */
  int64_t scale_off;      /* floats from scalars base (e.g. 1/sigma of spectral norm), -1 = none */
  int32_t co_src, ci_src;
  int32_t k_lo, k_len;    /* rows [k_lo, k_lo+k_len) of the packed operand */
  int32_t co_off, ci_off;
  int32_t transposed;     /* 0 forward, 1 data-gradient (flip taps, swap roles),
                             2+2*py+px: parity class (py,px) of a 4x4 stride-2 data gradient, packed as a
                             2x2-tap operand: tap (a,b) <- source tap (ty,tx), ty = py ? 2-2a : 3-2a;
                             6+2*py+px: same for a 3x3 stride-2 pad-1 conv (A-ESRGAN/model.py:287-291);
                             10+2*a+b: tap (a,b) of a 2x2 stride-2 conv (:236) as a 1x1 operand */
  float scale;
} srganfd_pack_seg;

typedef struct {
  int64_t dst_off;        /* bytes from packed base, multiple of 16 */
  int32_t dtype, ksize;
  int32_t k, n;           /* packed K and N, multiples of 32 */
  int32_t nseg;
  int32_t layout;         /* 0: B fragments of v_mfma_f32_32x32x16; 1: of v_mfma_f32_16x16x32 -- srganfd_pack_layout(dtype, ksize, n) says
                             which one the consuming kernel reads under the current srganfd_set_mfma16 level */
  srganfd_pack_seg seg[5];
} srganfd_pack_job;

size_t srganfd_packed_bytes(int32_t dtype, int32_t ksize, int32_t k, int32_t n);
/* MFMA form of the 16-bit convolutions: 3 = v_mfma_f32_16x16x32 for every kernel shape (the product library's only level). */
int srganfd_get_mfma16(void);
/* the `layout` a pack job must carry for an operand of this dtype / kernel size / packed output width n under the current setting */
int srganfd_pack_layout(int32_t dtype, int32_t ksize, int32_t n);
/* max_elems = max over jobs of ksize*ksize*k*n (grid sizing; host knows it) */
int srganfd_pack_weights(const srganfd_pack_job* jobs_dev, int32_t njobs, int64_t max_elems,
                         const float* params, const float* scalars, void* packed, void* stream);

/* Weight/bias-gradient pass (replaces the weight and bias outputs of ATen convolution_backward
 * for the convs above).  One launch covers several convolutions that read the same activation
 * buffer x and the same output-gradient buffer dy (the five convs of a dense block).  The plan
 * (wave tasks, LDS tile groups, slab table) is built on the host from the conv list, uploaded by
 * the caller once, and reused every step. */
typedef struct {
  int32_t ci_lo, cin;     /* x channels [ci_lo, ci_lo+cin) relative to the x view; cin multiple of 32 */
  int32_t co_lo, cout;    /* dy channels, relative to the dy view; multiple of 32 */
  int64_t dw_off;         /* floats from grads base: (co_dst, ci_dst, k, k) fp32 gradient */
  int64_t db_off;         /* floats from grads base: bias gradient or -1 */
  int32_t co_dst, ci_dst; /* real dims of the parameter (<= cout, cin: padded channels are dropped) */
  float alpha;            /* scale on the summed gradient */
  float beta;             /* 0 overwrite, 1 accumulate */
  int64_t alpha_off;      /* floats from scalars base multiplied into alpha, -1 = none */
} srganfd_wgrad_conv;

typedef struct {
  int32_t dtype;
  int32_t n, h_in, w_in, up, ksize, stride, pad, h_out, w_out;
  int32_t x_channels, dy_channels;  /* extents of the two views */
  int32_t nconv;
  int32_t splits;                    /* pixel-tile splits (0 = auto) */
} srganfd_wgrad_shape;

size_t srganfd_wgrad_plan_bytes(const srganfd_wgrad_shape* s, const srganfd_wgrad_conv* convs);
int srganfd_wgrad_plan_build(const srganfd_wgrad_shape* s, const srganfd_wgrad_conv* convs,
                             void* plan_host, size_t plan_bytes, size_t* workspace_bytes);
/* plan_host: the buffer filled by plan_build; plan_dev: the same bytes in device memory */
int srganfd_conv2d_wgrad(const void* plan_host, const void* plan_dev, srganfd_view x, srganfd_view dy,
                         float* grads, const float* scalars, void* workspace, size_t workspace_bytes,
                         void* stream);
/* The same in two steps, for launches whose slab reductions can share one kernel (the 69 dense blocks of the generator: the
 * reduction is latency-bound at ~20 us whatever it reduces): _partial runs the MFMA kernel only and leaves the fp32 partial slabs in
 * `workspace` (one workspace per pending launch); _reduce_batch finishes up to 8 pending launches of the same kernel size --
 * identical results to srganfd_conv2d_wgrad (same summation order per element). */
typedef struct {
  const void* plan_host; const void* plan_dev;
  float* grads; const float* scalars; const void* workspace;
} srganfd_wgrad_reduce_job;
int srganfd_conv2d_wgrad_partial(const void* plan_host, const void* plan_dev, srganfd_view x, srganfd_view dy,
                                 void* workspace, size_t workspace_bytes, void* stream);
int srganfd_wgrad_reduce_batch(const srganfd_wgrad_reduce_job* jobs, int32_t njobs, void* stream);

/* ---- boundary layout conversion (the reference's modules take/return NCHW fp32) ---- */
/* BSRGAN.forward input (model.py:366): NCHW fp32 -> NHWC dtype view, zero padded to cpad channels */
int srganfd_nchw_to_nhwc(const float* src, int32_t n, int32_t c, int32_t h, int32_t w, srganfd_view dst,
                         int32_t dtype, int32_t cpad, const float* ch_mean, const float* ch_std, void* stream);
/* (ch_mean, ch_std: optional per-channel (x - mean) / std of ContentLoss.normalize, model.py:542-543) */
/* NHWC view (dtype) -> NCHW fp32; clamp01 = torch.clamp_(out, 0, 1) of model.py:379 */
int srganfd_nhwc_to_nchw(srganfd_view src, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w,
                         float* dst, int32_t clamp01, void* stream);
/* backward of that clamp + relayout: dst[p][c] = (0 <= pre[p][c] <= 1) ? dsr[n][c][p] : 0 ; pre is fp32 NHWC */
int srganfd_clamp_grad_to_nhwc(const float* dsr_nchw, srganfd_view pre_f32, int32_t n, int32_t c, int32_t h,
                               int32_t w, srganfd_view dst, int32_t dtype, int32_t cpad, void* stream);

/* resampling: op 0 = backward of nearest x2 (model.py:372,374), 1 = bilinear x2 forward
 * (align_corners=False, model.py:150,154,158), 2 = its backward, 3 = 2x2 max-pool (VGG-19).
 * 4 = ReLU copy.  (h, w) are the LOW-resolution dims for ops 0-2 and the input dims for ops 3-4;
 * a = source view, b = destination view. */
int srganfd_resample(int32_t op, srganfd_view a, srganfd_view b, int32_t dtype, int32_t n, int32_t h,
                     int32_t w, int32_t c, void* stream);
/* Backward of `lrelu(conv(bilinear_x2(u))) + skip` chains (the U-Net decoder, model.py:150-161) in one pass: dx_raw = adjoint of the
 * bilinear x2 upsampling applied to dy ((h, w) = LOW-resolution dims), dx_masked = dx_raw * (act > 0 ? 1 : slope) with act = the
 * LeakyReLU output of the upsampled layer.  dx_raw.ptr may be NULL; dx_masked may alias nothing it reads. */
int srganfd_resample_bwd_lrelu(srganfd_view dy, srganfd_view dx_raw, srganfd_view act, srganfd_view dx_masked,
                               int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, float slope, void* stream);
/* out = dy * (act - skip > 0 ? 1 : slope): LeakyReLU backward where only act = lrelu(z) + skip was
 * stored (U-Net skip adds, model.py:153,157,161); skip.ptr may be NULL (plain LeakyReLU backward). */
int srganfd_lrelu_bwd(srganfd_view dy, srganfd_view act, srganfd_view skip, srganfd_view out, int32_t dtype,
                      int64_t npix, int32_t c, float slope, void* stream);
/* y = alpha * x + beta * y over npix pixels x c channels */
int srganfd_axpby(srganfd_view x, srganfd_view y, int32_t dtype, int64_t npix, int32_t c, float alpha,
                  float beta, void* stream);

/* ---- losses (train_bsrgan.py:297,301,417,427,450-452).  workspace: >= 2048 floats.
 * *out = (accumulate ? *out : 0) + weight * mean(...).  Deterministic two-stage reductions. */
#define SRGANFD_LOSS_WS_FLOATS 2049
/* grad = grad_scale * (grad_scale_dev ? *grad_scale_dev : 1) * d(mean)/d(input): grad_scale_dev is the device-resident loss scale of
 * f16 training (srganfd_loss_scale_update), NULL otherwise. */
int srganfd_l1_loss(const float* a, const float* b, int64_t numel, float weight, float* out,
                    int32_t accumulate, float* grad_a /* or NULL */, float grad_scale,
                    const float* grad_scale_dev /* or NULL */, float* workspace, void* stream);
int srganfd_l1_loss_views(srganfd_view a, srganfd_view b, int32_t dtype, int64_t npix, int32_t c,
                          int32_t relu_first, float weight, float* out, int32_t accumulate,
                          float* workspace, void* stream);
int srganfd_bce_logits(const float* logits, int64_t numel, float target, float weight, float* loss_out,
                       int32_t accumulate, float* sigmoid_mean_out /* or NULL */, float* grad /* or NULL */,
                       float grad_scale, const float* grad_scale_dev /* or NULL */, float* workspace, void* stream);
/* Relativistic-average BCE of ESRGAN (train_esrgan.py:378-380,404,412): *loss_out (+)= weight * mean_i BCEWithLogits(x_i - mean(other), target).
 * grad_x[i] (+)= g * (sigmoid(x_i - mean(other)) - target) / numel, grad_other[j] (+)= -g * mean_i(sigmoid(x_i - mean(other)) - target) / numel_other
 * with g = grad_scale * (grad_scale_dev ? *grad_scale_dev : 1); either gradient may be NULL; accumulate_* != 0 adds to what the buffer holds
 * (the generator's adversarial term reaches sr_output through both of its halves).  Deterministic two-stage reductions. */
int srganfd_bce_logits_relativistic(const float* x, int64_t numel, const float* other, int64_t numel_other, float target, float weight,
                                    float* loss_out, int32_t accumulate, float* grad_x /* or NULL */, int32_t accumulate_x,
                                    float* grad_other /* or NULL */, int32_t accumulate_other, float grad_scale,
                                    const float* grad_scale_dev /* or NULL */, float* workspace, void* stream);
/* *out = sigmoid(mean(logits)): the D(x) probability as ESRGAN / Real-ESRGAN log it (train_esrgan.py:430-431,
 * train_realesrgan.py:475-476; BSRGAN logs mean(sigmoid), srganfd_bce_logits' sigmoid_mean_out). */
int srganfd_sigmoid_of_mean(const float* logits, int64_t numel, float* out, float* workspace, void* stream);

/* ---- spectral norm (torch/nn/utils/spectral_norm.py:62-114 as applied at model.py:104-132).
 * One power iteration in place on u, v when training; sigma = u^T W v; workspace >= ceil(rows / 32) * cols + rows floats. */
int srganfd_spectral_norm(const float* w_orig, float* u, float* v, int32_t rows, int32_t cols,
                          int32_t training, float eps, float* sigma_out, float* inv_sigma_out,
                          float* workspace, void* stream);
/* The same for several layers at once (the discriminators normalise 8 / 20 layers before every forward pass: four dependent
 * 5-14 us kernels per layer are launch latency, not work).  jobs: HOST array; each layer is computed exactly as by a call of its
 * own (bit-identical); each job needs its own workspace of ceil(rows / 32) * cols + rows floats. */
#define SRGANFD_SN_BATCH 8        /* layers per launch; longer job lists run in groups */
typedef struct srganfd_sn_job {
  const float* w_orig; float* u; float* v; float* sigma_out; float* inv_sigma_out; float* workspace;
  int32_t rows, cols;
} srganfd_sn_job;
int srganfd_spectral_norm_batch(const srganfd_sn_job* jobs, int32_t njobs, int32_t training, float eps, void* stream);
/* dW_orig = beta*dW_orig + (G - <G,W_orig>/sigma * u v^T)/sigma ; workspace >= 1025 floats */
int srganfd_spectral_norm_grad(const float* g_weight, const float* w_orig, const float* u, const float* v,
                               const float* inv_sigma, float* dw_orig, int32_t rows, int32_t cols,
                               float beta, float* workspace, void* stream);
/* Several layers at once (jobs: HOST array, each with its own 1025-float workspace), bit-identical to per-layer calls. */
typedef struct srganfd_sn_grad_job {
  const float* g_weight; const float* w_orig; const float* u; const float* v; const float* inv_sigma;
  float* dw_orig; float* workspace;
  int32_t rows, cols;
} srganfd_sn_grad_job;
int srganfd_spectral_norm_grad_batch(const srganfd_sn_grad_job* jobs, int32_t njobs, float beta, void* stream);

/* ---- fused Adam + EMA over flat buffers (torch.optim.Adam maths, train_bsrgan.py:311-323,436,466;
 * AveragedModel with the reference's avg_fn, :290-291,470).  ema_mode: 0 none, 1 copy (first
 * update), 2 ema = (1-decay)*ema + decay*param.  grad_scale multiplies the gradient first
 * (1/world_size after an all-reduce(sum), times 1/loss_scale in f16 mode).  skip_flag (device float or NULL): when *skip_flag != 0
 * the parameter / moment update is skipped and only the EMA advances -- torch.cuda.amp.GradScaler.step() on a non-finite gradient
 * followed by the unconditional ema update of train_bsrgan.py:466-470. */
int srganfd_adam_ema(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema,
                     int64_t numel, float lr, float beta1, float beta2, float eps, float weight_decay,
                     int32_t step, float grad_scale, float ema_decay, int32_t ema_mode, const float* skip_flag,
                     const float* grad_scale_dev /* or NULL: multiplies grad_scale (1 / loss scale from the scaler state) */, void* stream);
/* torch.amp.GradScaler.update() on a device-resident state (torch keeps its scale on the device as well, so that neither the host nor a
 * captured graph carries a stale value): state = 8 words {float scale, float 1 / scale, int32 growth tracker, int32 optimizer steps, int32 skipped steps, 0, 0, 0}.
 * *found_inf != 0: scale *= backoff_factor, tracker = 0; else tracker += 1 and, at growth_interval, scale *= growth_factor (kept if that
 * would overflow), tracker = 0.  Loss kernels read state[0] as grad_scale_dev, the Adam kernels state[1]; train_bsrgan.py:109,436-437,466-467. */
int srganfd_loss_scale_update(float* state, const float* found_inf, float growth_factor, float backoff_factor,
                              int32_t growth_interval, void* stream);
/* *flag = 1.0 if any element of x is inf or NaN, else 0.0 (accumulate != 0: keeps an earlier 1.0): the found_inf of
 * GradScaler.unscale_ (train_bsrgan.py:436,466), computed on the flat (all-reduced) gradient. */
int srganfd_nonfinite_flag(const float* x, int64_t numel, float* flag, int32_t accumulate, void* stream);
/* Same update with the step count in device memory: *step_dev is advanced by one and the bias corrections are computed on
 * the device (bc_dev: 2 floats of scratch), so that a captured hipGraph of the iteration can be replayed. */
int srganfd_adam_ema_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema,
                         int64_t numel, float lr, float beta1, float beta2, float eps, float weight_decay,
                         int32_t* step_dev, float* bc_dev, float grad_scale, float ema_decay, int32_t ema_mode,
                         const float* skip_flag, const float* grad_scale_dev /* or NULL */, void* stream);

/* ---- thin-side 3x3 convolutions (stride 1, pad 1): 1..4 channels against 64 (csrc/conv_thin.hip) ----
 * The generator's conv1 / conv4 (BSRGAN/model.py:325,355), the discriminators' conv1 / conv4 (:102,135; A-ESRGAN/model.py:287,307;
 * ESRGAN/model.py:92) and VGG-19 features.0 (:522-524), with their data and weight gradients.  16-bit dtypes only.  The thin tensor
 * is NHWC with a pitch of 4 channels ("NHWC4", 8 bytes per pixel; channels >= cs must be zero), the 64-channel tensor an ordinary
 * view.  `weight` is the layer's RAW fp32 parameter (Cout, Cin, 3, 3): no packed copy exists for these layers.
 *   w_big_is_cout  1: the 64-channel side is the weight's Cout (a 1..4 -> 64 conv), 0: it is its Cin (a 64 -> 1..4 conv)
 *   flip           1: kernel rotated by 180 degrees (the launch is the DATA GRADIENT of the conv that owns `weight`) */
typedef struct srganfd_thin_args {
  int32_t dtype, n, h, w, cs, w_big_is_cout, flip, act;
  float slope, mask_slope;
  const float* weight;
  const float* bias;        /* of the launch's OUTPUT channels (64 for thin_in, cs for thin_out), or NULL */
  srganfd_view big;         /* thin_in: output; thin_out: input; thin_wgrad: the 64-channel operand (dy of a 1..4 -> 64 conv, x of a 64 -> 1..4 conv) */
  srganfd_view mask;        /* thin_in only: y *= (mask > 0 ? 1 : mask_slope), or NULL */
  const void* thin;         /* thin_in: input; thin_wgrad: the thin operand (x resp. dy); NHWC4, 16-bit */
  float* thin_out;          /* thin_out: fp32 output */
  int32_t thin_out_pitch;   /* its pixel pitch in floats: 4 (channels >= cs are written as zeros) or, with cs == 1, 1 */
  int32_t pad_;
} srganfd_thin_args;
/* big[p][b] = act(bias[b] + sum_{tap,s} W(b,s,tap) * thin[p + tap][s]) (* mask): one K = 36 MFMA pair per 16 pixels x 16 channels */
int srganfd_conv2d_thin_in(const srganfd_thin_args* a, void* stream);
/* thin_out[p][s] = bias[s] + sum_{tap,b} W(b,s,tap) * big[p + tap][b] */
int srganfd_conv2d_thin_out(const srganfd_thin_args* a, void* stream);
/* dw (raw layout of `weight`'s tensor) = sum_p big[p][b] * thin[p +- tap][s]; db = the conv's bias gradient (64 values when
 * w_big_is_cout, else cs) or NULL.  Deterministic (slabs + ordered reduction); workspace >= srganfd_conv2d_thin_wgrad_workspace(). */
size_t srganfd_conv2d_thin_wgrad_workspace(void);
int srganfd_conv2d_thin_wgrad(const srganfd_thin_args* a, float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream);

/* ---- A-ESRGAN attention U-Net discriminator (A-ESRGAN/model.py:228-345) ---- */
/* F.interpolate(size=..., mode="bilinear", align_corners=False) (model.py:245,250): bwd=0: a (hi x wi) -> b (ho x wo);
 * bwd=1: a = dy (ho x wo) -> b = dx (hi x wi), deterministic gather. */
int srganfd_resize_bilinear(int32_t bwd, srganfd_view a, srganfd_view b, int32_t dtype, int32_t n, int32_t hi,
                            int32_t wi, int32_t ho, int32_t wo, int32_t c, void* stream);
/* out = relu(a + b)  (model.py:246) */
int srganfd_add_relu(srganfd_view a, srganfd_view b, srganfd_view out, int32_t dtype, int64_t npix, int32_t c, void* stream);
/* in-place sigmoid of an fp32 map (model.py:248) and its backward out = ds * s * (1 - s) */
int srganfd_sigmoid(float* x, int64_t numel, void* stream);
int srganfd_sigmoid_bwd(const float* ds, const float* s, float* out, int64_t numel, void* stream);
/* attention gate y = gate[p] * x[p][c] (model.py:252).  bwd=1: y is dy; dx = gate * dy, dgate[p] = sum_c dy * x */
int srganfd_gate_mul(int32_t bwd, srganfd_view x, const float* gate, srganfd_view y, srganfd_view dx, float* dgate,
                     int32_t dtype, int64_t npix, int32_t c, void* stream);
/* nn.BatchNorm2d (model.py:233).  save: 4*c floats [mean | invstd | scale | shift]; workspace: 2048*c + 3*c floats.
 * training=1: batch statistics, running stats updated with `momentum` (unbiased variance); 0: running stats. */
int srganfd_batchnorm_fwd(srganfd_view x, srganfd_view y, int32_t dtype, int64_t npix, int32_t c, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                          int32_t training, float* save, float* workspace, void* stream);
/* dgamma/dbeta = acc * old + new; dx from the training-mode formula */
int srganfd_batchnorm_bwd(srganfd_view x, srganfd_view dy, srganfd_view dx, int32_t dtype, int64_t npix, int32_t c,
                          const float* gamma, const float* save, float* dgamma, float* dbeta, float acc,
                          float* workspace, void* stream);

/* Data-parallel SyncBatchNorm (the reference wraps its BatchNorm discriminator in DistributedDataParallel without converting it,
 * A-ESRGAN/model.py:233, so each rank normalises with its own statistics; this is the optional whole-batch form).
 * Two-phase calls around one all-reduce of the first srganfd_batchnorm_partial_floats(c) floats of the workspace:
 *   phase 1: this rank's partial sums ((sum x, sum x^2) forward, (sum dy, sum dy*xhat) backward) into `workspace`;
 *   caller:  sums that table over the ranks (forward in place; backward into `workspace_global`, `workspace` keeps its own);
 *   phase 2: statistics / dx coefficients from the summed table with total_npix = pixels of the whole batch, then the apply pass.
 * dgamma/dbeta are this rank's sums (the gradient all-reduce adds the ranks).  c <= 256, training mode; `act` may be empty
 * (ptr NULL) or the LeakyReLU(act_slope) output whose derivative is folded into dy; phase 0 = the single-call form. */
int64_t srganfd_batchnorm_partial_floats(int32_t c);
int srganfd_batchnorm_fwd_sync(srganfd_view x, srganfd_view y, int32_t dtype, int64_t npix, int32_t c, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* save,
                               float* workspace, float act_slope, int32_t phase, int64_t total_npix, void* stream);
int srganfd_batchnorm_bwd_sync(srganfd_view x, srganfd_view dy, srganfd_view dx, int32_t dtype, int64_t npix, int32_t c,
                               const float* gamma, const float* save, float* dgamma, float* dbeta, float acc, float* workspace,
                               const float* workspace_global, srganfd_view act, float act_slope, int32_t phase, int64_t total_npix,
                               void* stream);

/* BatchNorm2d followed by LeakyReLU(act_slope) in one pass (ESRGAN/model.py:98-126: conv -> BatchNorm2d -> LeakyReLU(0.2));
 * backward takes `act` = that LeakyReLU's output and folds its derivative into dy.  Any channel count that splits into
 * blocks of 256 (last block a power-of-two number of 16-byte chunks); workspace as above for 256 channels. */
int srganfd_batchnorm_act_fwd(srganfd_view x, srganfd_view y, int32_t dtype, int64_t npix, int32_t c, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                              int32_t training, float* save, float* workspace, float act_slope, void* stream);
int srganfd_batchnorm_act_bwd(srganfd_view x, srganfd_view dy, srganfd_view dx, int32_t dtype, int64_t npix, int32_t c,
                              const float* gamma, const float* save, float* dgamma, float* dbeta, float acc,
                              float* workspace, srganfd_view act, float act_slope, void* stream);

/* ---- differentiable VGG tap (ESRGAN/model.py:281-292: F.l1_loss between one feature node of SR and GT) ----
 * out = scale * (*upstream or 1) * sign(a - b): the gradient of weight * mean|a - b| when scale = weight / numel. */
int srganfd_l1_grad_views(srganfd_view a, srganfd_view b, srganfd_view out, int32_t dtype, int64_t npix, int32_t c,
                          const float* upstream /* device scalar or NULL */, float scale, void* stream);
/* MaxPool2d(2,2) backward fused with the derivative of the ReLU in front of it: x = the ReLU output (pre-pool,
 * n x h x w x c), dy = gradient of the pooled map, dx = gradient w.r.t. the ReLU input (first maximum in row-major
 * window order takes the gradient, as in ATen; zero where that maximum is not positive). */
int srganfd_maxpool2_relu_bwd(srganfd_view x, srganfd_view dy, srganfd_view dx, int32_t dtype, int32_t n, int32_t h,
                              int32_t w, int32_t c, void* stream);
/* fp32 NHWC -> NCHW with dst[n][c] = src[..][c] / ch_div[c] (backward of transforms.Normalize's division by std) */
int srganfd_nhwc_to_nchw_scaled(srganfd_view src_f32, int32_t n, int32_t c, int32_t h, int32_t w, float* dst,
                                const float* ch_div, void* stream);

/* ---- validation / data side ----
 * random_crop (imgproc.py:846-886): the batch's common window (top, left, ph x pw) of NCHW fp32 images in one copy. */
int srganfd_crop_nchw(const float* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t top, int32_t left,
                      int32_t ph, int32_t pw, void* stream);
/* uint8 ingest (dataset.py:64-96 + imgproc.py:331-358 per batch on the device): decoded HWC uint8 images (n, h, w, 3) -> the window
 * (top, left, ph x pw) as NCHW fp32 = value / scale (255 for the reference's [0, 1] range), channels swapped 0 <-> 2 when swap_rb
 * (cv2's BGR -> RGB, dataset.py:81). */
int srganfd_u8hwc_to_nchw(const unsigned char* src, float* dst, int32_t n, int32_t h, int32_t w, int32_t top, int32_t left,
                          int32_t ph, int32_t pw, int32_t swap_rb, float scale, void* stream);
/* PSNR per image (image_quality_assessment.py:361-395): NCHW fp32 in [0,1], crop_border pixels dropped on every side,
 * y_only = BT.601 luma of RGB first (imgproc.py:757-767); out: n doubles (dB); workspace: n * 64 doubles. */
int srganfd_psnr(const float* a, const float* b, int32_t n, int32_t c, int32_t h, int32_t w, int32_t crop_border,
                 int32_t y_only, double* out, double* workspace, void* stream);
/* SSIM per image (image_quality_assessment.py:420-494, the SSIM module at :497-532): same inputs / crop / luma as
 * srganfd_psnr; `window` = window_size x window_size fp64 filter in device memory (the reference's outer product of
 * cv2.getGaussianKernel(11, 1.5)), window_size <= 16, valid padding; moments and map in fp64, out: n floats (the mean of
 * the map over channels and pixels, cast like the reference's .float()).  workspace: srganfd_ssim_workspace_doubles(). */
int64_t srganfd_ssim_workspace_doubles(int32_t n, int32_t c, int32_t h, int32_t w, int32_t crop_border, int32_t y_only,
                                       int32_t window_size);
int srganfd_ssim(const float* a, const float* b, int32_t n, int32_t c, int32_t h, int32_t w, int32_t crop_border,
                 int32_t y_only, const double* window, int32_t window_size, float* out, double* workspace, void* stream);

/* ---- Real-ESRGAN on-device degradation (Real_ESRGAN/imgproc.py) ----
 * filter2d_torch (imgproc.py:1092-1124): NCHW fp32 image (b,c,h,w), reflect padding k/2, cross-correlation of every
 * channel of image n with kernels[n] (kernel_batch == b) or the one shared kernel (kernel_batch == 1); k odd, <= 51
 * ("Wrong kernel size." for an even k, like the reference's ValueError). */
int srganfd_filter2d(const float* image, const float* kernels, int32_t kernel_batch, int32_t b, int32_t c, int32_t h,
                     int32_t w, int32_t k, float* out, void* stream);
/* the same filter when kernels[n] is an outer product: taps = per image (or shared) [k vertical taps | k horizontal taps];
 * a horizontal pass through LDS then the vertical one, 2k instead of k*k multiply-adds per pixel. */
int srganfd_filter2d_separable(const float* image, const float* taps, int32_t kernel_batch, int32_t b, int32_t c, int32_t h,
                               int32_t w, int32_t k, float* out, void* stream);
/* USMSharp.forward (imgproc.py:1529-1540): blur with the shared k x k kernel, residual, |residual|*255 > threshold mask,
 * blurred mask, blend -- two fused filter passes.  separable != 0: `kernel` holds 2k taps as for
 * srganfd_filter2d_separable (USMSharp's kernel is the outer product of a 1-D Gaussian).  workspace: 2 * b*c*h*w floats. */
int srganfd_usm_sharp(const float* image, const float* kernel, int32_t separable, int32_t b, int32_t c, int32_t h, int32_t w,
                      int32_t k, float weight, float threshold, float* out, float* workspace, void* stream);
/* DiffJPEG.forward (imgproc.py:1465-1497): RGB NCHW fp32 in [0,1] -> JPEG round trip (4:2:0, zero-padded to multiples of
 * 16, cropped back).  quality: b floats in device memory, converted IN PLACE to the compression factor
 * (imgproc.py:1127-1144, :1476-1480) unless quality_is_factor; differentiable = the cubic rounding of :1183-1195.
 * tables: srganfd_diff_jpeg_table_floats() floats in device memory, filled on the host by srganfd_diff_jpeg_tables(). */
int32_t srganfd_diff_jpeg_table_floats(void);
int srganfd_diff_jpeg_tables(float* host_out);
int srganfd_diff_jpeg(const float* image, int32_t b, int32_t c, int32_t h, int32_t w, float* quality,
                      int32_t quality_is_factor, int32_t differentiable, const float* tables, float* out, void* stream);
/* F.interpolate as degradation_process calls it (imgproc.py:2374, :2415-2418, :2440-2442, :2454-2456): mode 0 "area"
 * (adaptive average pooling), 1 "bilinear", 2 "bicubic" (A = -0.75), align_corners unset.  `planes` = b*c NCHW fp32 planes.
 * rscale_h/w: 1 / scale_factor when the caller passed scale_factor= (torch maps coordinates with it), 0 = in / out. */
int srganfd_resize(const float* src, int32_t planes, int32_t h, int32_t w, int32_t out_h, int32_t out_w, int32_t mode,
                   float rscale_h, float rscale_w, float* dst, void* stream);
/* random_add_gaussian_noise_torch after its draws (imgproc.py:849-866, :1046-1060): randn_color (b,c,h,w), randn_gray_hw
 * (h,w) or NULL, sigma[b] (range 255), gray_flag[b] in {0,1}; out = image + mixed noise, then clip / rounds. */
int srganfd_gaussian_noise(const float* image, const float* randn_color, const float* randn_gray_hw, const float* sigma,
                           const float* gray_flag, int32_t b, int32_t c, int32_t h, int32_t w, int32_t clip, int32_t rounds,
                           float* out, void* stream);
/* random_add_poisson_noise_torch around its draws (imgproc.py:886-919, :1077-1089).  prepare: image rounded to 8 bits
 * (image_q; gray_q = the same of torchvision's grey image when want_gray) and vals[b] = 2^ceil(log2(#distinct levels));
 * workspace: b * 512 uint32.  The caller draws poisson(image_q * vals) (and the grey one); apply mixes, scales, adds, clips. */
int srganfd_poisson_prepare(const float* image, int32_t b, int32_t c, int32_t h, int32_t w, int32_t want_gray,
                            float* image_q, float* gray_q, float* vals, float* vals_gray, void* workspace, void* stream);
int srganfd_poisson_apply(const float* image, const float* image_q, const float* gray_q, const float* poisson_color,
                          const float* poisson_gray, const float* vals, const float* vals_gray, const float* scale,
                          const float* gray_flag, int32_t b, int32_t c, int32_t h, int32_t w, int32_t clip, int32_t rounds,
                          float* out, void* stream);
/* random_crop_torch / random_rotate_torch / random_*_flip_torch (imgproc.py:2081-2320) on one NCHW fp32 tensor: crop the
 * window (top, left, ph x pw) of every plane, then op 0 nothing, 1 / 2 / 3 = 90 / 180 / 270 degrees counter-clockwise
 * (square windows), 4 horizontal flip, 5 vertical flip.  dst: planes x ph x pw. */
int srganfd_crop_rot_flip(const float* src, float* dst, int32_t planes, int32_t h, int32_t w, int32_t top, int32_t left,
                          int32_t ph, int32_t pw, int32_t op, void* stream);
/* last line of degradation_process (imgproc.py:2460): dst = clamp(round(src * 255), 0, 255) / 255 (may alias) */
int srganfd_quantize_u8(const float* src, float* dst, int64_t numel, void* stream);

#ifdef __cplusplus
}
#endif
#endif
