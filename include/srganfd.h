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

/* epilogue activation */
#define SRGANFD_ACT_NONE 0
#define SRGANFD_ACT_LRELU 1
#define SRGANFD_ACT_RELU 2

const char* srganfd_last_error(void);
int srganfd_abi_version(void);

/* A channel-slice view of an NHWC activation buffer: element (n,y,x,c) lives at
 * ptr[((n*H + y)*W + x)*cstride + c0 + c]. */
typedef struct {
  void* ptr;
  int32_t cstride; /* channels of the underlying buffer */
  int32_t c0;      /* first channel of the view */
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
  int32_t dtype;            /* SRGANFD_BF16 | SRGANFD_F32 */
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
} srganfd_conv_args;

int srganfd_conv2d(const srganfd_conv_args* a, void* stream);

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
  int32_t transposed;     /* 0 forward, 1 data-gradient (flip taps, swap roles) */
  float scale;
} srganfd_pack_seg;

typedef struct {
  int64_t dst_off;        /* bytes from packed base, multiple of 16 */
  int32_t dtype, ksize;
  int32_t k, n;           /* packed K and N, multiples of 32 */
  int32_t nseg;
  int32_t pad_;
  srganfd_pack_seg seg[5];
} srganfd_pack_job;

size_t srganfd_packed_bytes(int32_t dtype, int32_t ksize, int32_t k, int32_t n);
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

#ifdef __cplusplus
}
#endif
#endif
