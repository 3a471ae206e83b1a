// abi.hip -- extern "C" surface of libsrganfd_hip.so (see include/srganfd.h).
#include "conv_common.hpp"
#include <stdarg.h>
#include <stdlib.h>

namespace srganfd {
thread_local char g_err[512] = {0};
int g_dry_run = 0;
thread_local char* g_describe = nullptr;
thread_local size_t g_describe_len = 0;
int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int conv2d_impl(const srganfd_conv_args* a, hipStream_t stream);
int dense_chain_impl(const srganfd_conv_args* layers, int n, void* workspace, size_t workspace_bytes, hipStream_t stream);
int dense_chain_check_impl(const srganfd_conv_args* layers, int n);
size_t dense_chain_workspace_bytes_impl();
int pack_weights_impl(const srganfd_pack_job* jobs_dev, int njobs, long long max_elems, const float* params,
                      const float* scalars, void* packed, hipStream_t stream);
size_t wgrad_plan_bytes_impl(const srganfd_wgrad_shape* s, const srganfd_wgrad_conv* convs);
int wgrad_plan_build_impl(const srganfd_wgrad_shape* s, const srganfd_wgrad_conv* convs, void* plan_host, size_t plan_bytes,
                          size_t* workspace_bytes);
int wgrad_reduce_batch_impl(const srganfd_wgrad_reduce_job* jobs, int njobs, hipStream_t stream);
int wgrad_impl(const void* plan_host, const void* plan_dev, srganfd_view x, srganfd_view dy, float* grads, const float* scalars,
               void* workspace, size_t workspace_bytes, hipStream_t stream);
int nchw_to_nhwc_impl(const float* src, int n, int c, int h, int w, srganfd_view dst, int dtype, int cpad, const float* mean, const float* stdv, hipStream_t s);
int lrelu_bwd_impl(srganfd_view dy, srganfd_view act, srganfd_view skip, srganfd_view out, int dtype, size_t npix, int c, float slope, hipStream_t s);
int nhwc_to_nchw_impl(srganfd_view src, int dtype, int n, int c, int h, int w, float* dst, int clamp01, hipStream_t s);
int clamp_grad_impl(const float* dsr, srganfd_view pre, int n, int c, int h, int w, srganfd_view dst, int dtype, int cpad, hipStream_t s);
int resample_impl(int op, srganfd_view a, srganfd_view b, int dtype, int n, int h, int w, int c, hipStream_t s);
int resample_bwd_lrelu_impl(srganfd_view dy, srganfd_view dx_raw, srganfd_view act, srganfd_view dx_masked, int dtype, int n, int h, int w, int c, float slope,
                            hipStream_t s);
int axpby_impl(srganfd_view x, srganfd_view y, int dtype, size_t npix, int c, float a, float b, hipStream_t s);
int l1_loss_impl(const float* a, const float* b, size_t n, float weight, float* out, int accumulate, float* grad, float grad_scale,
                 const float* grad_scale_dev, float* ws, hipStream_t s);
int sigmoid_of_mean_impl(const float* x, size_t n, float* out, float* ws, hipStream_t s);
int l1_views_impl(srganfd_view a, srganfd_view b, int dtype, size_t npix, int c, int relu, float weight, float* out, int accumulate, float* ws, hipStream_t s);
int bce_logits_impl(const float* x, size_t n, float target, float weight, float* loss_out, int accumulate, float* sig_mean_out, float* grad,
                    float grad_scale, const float* grad_scale_dev, float* ws, hipStream_t s);
int bce_logits_relativistic_impl(const float* x, size_t n, const float* other, size_t n_other, float target, float weight, float* loss_out, int accumulate,
                                 float* grad_x, int accumulate_x, float* grad_other, int accumulate_other, float grad_scale,
                                 const float* grad_scale_dev, float* ws, hipStream_t s);
int spectral_norm_grad_batch_impl(const srganfd_sn_grad_job* jobs, int njobs, float beta, hipStream_t s);
int spectral_norm_batch_impl(const srganfd_sn_job* jobs, int njobs, int training, float eps, hipStream_t s);
int spectral_norm_impl(const float* W, float* u, float* v, int rows, int cols, int training, float eps, float* sigma, float* inv_sigma, float* ws, hipStream_t s);
int spectral_norm_grad_impl(const float* G, const float* W, const float* u, const float* v, const float* inv_sigma, float* dW, int rows, int cols,
                            float beta, float* ws, hipStream_t s);
int adam_ema_impl(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float b1, float b2, float eps, float wd, int step,
                  float grad_scale, float ema_decay, int ema_mode, const float* skip_flag, const float* grad_scale_dev, hipStream_t s);
int loss_scale_update_impl(float* state, const float* found_inf, float growth, float backoff, int interval, hipStream_t s);
int nonfinite_flag_impl(const float* x, size_t n, float* flag, int accumulate, hipStream_t s);
int resize_bilinear_impl(int bwd, srganfd_view a, srganfd_view b, int dtype, int n, int hi, int wi, int ho, int wo, int c, hipStream_t s);
int add_relu_impl(srganfd_view a, srganfd_view b, srganfd_view out, int dtype, size_t npix, int c, hipStream_t s);
int l1_grad_views_impl(srganfd_view a, srganfd_view b, srganfd_view out, int dtype, size_t npix, int c, const float* upstream, float scale, hipStream_t s);
int maxpool2_relu_bwd_impl(srganfd_view x, srganfd_view dy, srganfd_view dx, int dtype, int n, int h, int w, int c, hipStream_t s);
int nhwc_to_nchw_scaled_impl(srganfd_view src, int n, int c, int h, int w, float* dst, const float* ch_div, hipStream_t s);
int adam_ema_dev_impl(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float b1, float b2, float eps, float wd,
                      int* step_dev, float* bc_dev, float grad_scale, float ema_decay, int ema_mode, const float* skip_flag,
                      const float* grad_scale_dev, hipStream_t s);
int crop_nchw_impl(const float* src, float* dst, int n, int c, int h, int w, int top, int left, int ph, int pw, hipStream_t s);
int u8hwc_to_nchw_impl(const unsigned char* src, float* dst, int n, int h, int w, int top, int left, int ph, int pw, int swap_rb, float scale, hipStream_t s);
int psnr_impl(const float* a, const float* b, int n, int c, int h, int w, int crop_border, int y_only, double* out, double* ws, hipStream_t s);
int filter2d_impl(const float* src, const float* kernels, int kernel_batch, int b, int c, int h, int w, int k, int mode, const float* x_in,
                  const float* res_in, float weight, float threshold, float* out, float* out2, hipStream_t s, bool separable = false);
void diff_jpeg_tables_host(float* t);
int jpeg_table_floats();
int diff_jpeg_impl(const float* src, int b, int c, int h, int w, float* quality, int quality_is_factor, int differentiable, const float* tables,
                   float* dst, hipStream_t s);
int quantize_u8_impl(const float* src, float* dst, size_t n, hipStream_t s);
int crop_rot_flip_impl(const float* src, float* dst, int planes, int h, int w, int top, int left, int ph, int pw, int op, hipStream_t s);
int resize_impl(const float* src, int planes, int h, int w, int oh, int ow, int mode, float rscale_h, float rscale_w, float* dst, hipStream_t s);
int gaussian_noise_impl(const float* image, const float* n_color, const float* n_gray, const float* sigma, const float* gray, int b, int c, int h, int w,
                        int clip, int rounds, float* out, hipStream_t s);
int poisson_prepare_impl(const float* image, int b, int c, int h, int w, int want_gray, float* img_q, float* gray_q, float* vals, float* vals_gray,
                         unsigned int* presence, hipStream_t s);
int poisson_apply_impl(const float* image, const float* img_q, const float* gray_q, const float* pois, const float* pois_gray, const float* vals,
                       const float* vals_gray, const float* scale, const float* gray, int b, int c, int h, int w, int clip, int rounds, float* out,
                       hipStream_t s);
int64_t ssim_workspace_doubles(int n, int c, int h, int w, int crop_border, int y_only, int ws);
int ssim_impl(const float* a, const float* b, int n, int c, int h, int w, int crop_border, int y_only, const double* window, int ws, float* out,
              double* wsp, hipStream_t s);
int sigmoid_impl(float* x, size_t n, hipStream_t s);
int sigmoid_bwd_impl(const float* ds, const float* sg, float* out, size_t n, hipStream_t s);
int gate_mul_impl(int bwd, srganfd_view x, const float* gate, srganfd_view y, srganfd_view dx, float* dgate, int dtype, size_t npix, int c, hipStream_t s);
int batchnorm_fwd_impl(srganfd_view x, srganfd_view y, int dtype, size_t npix, int c, const float* gamma, const float* beta, float* rm, float* rv,
                       float momentum, float eps, int training, float* save, float* ws, float act_slope, hipStream_t s, int phase = 0,
                       size_t total_npix = 0);
int batchnorm_bwd_impl(srganfd_view x, srganfd_view dy, srganfd_view dx, int dtype, size_t npix, int c, const float* gamma, const float* save,
                       float* dgamma, float* dbeta, float acc, float* ws, srganfd_view act, float act_slope, hipStream_t s, int phase = 0,
                       const float* ws_global = nullptr, size_t total_npix = 0);
long long batchnorm_partial_floats_impl(int c);
int conv2d_thin_in_impl(const srganfd_thin_args* a, hipStream_t s);
int conv2d_thin_out_impl(const srganfd_thin_args* a, hipStream_t s);
size_t conv2d_thin_wgrad_workspace_impl();
int conv2d_thin_wgrad_impl(const srganfd_thin_args* a, float* dw, float* db, void* ws, size_t ws_bytes, hipStream_t s);
}  // namespace srganfd

using namespace srganfd;

extern "C" {

const char* srganfd_last_error(void) { return g_err; }
int srganfd_abi_version(void) { return SRGANFD_ABI_VERSION; }
void srganfd_set_dry_run(int on) { g_dry_run = on ? 1 : 0; }
int srganfd_get_mfma16(void) { return srganfd::g_mfma16; }
int srganfd_pack_layout(int32_t dtype, int32_t ksize, int32_t n) { return srganfd::conv_uses_m16(dtype, ksize, n) ? 1 : 0; }

int srganfd_conv2d(const srganfd_conv_args* a, void* stream) { return conv2d_impl(a, (hipStream_t)stream); }
int srganfd_dense_chain(const srganfd_conv_args* layers, int32_t n_layers, void* workspace, size_t workspace_bytes, void* stream) {
  return dense_chain_impl(layers, n_layers, workspace, workspace_bytes, (hipStream_t)stream);
}
int srganfd_dense_chain_check(const srganfd_conv_args* layers, int32_t n_layers) { return dense_chain_check_impl(layers, n_layers); }
size_t srganfd_dense_chain_workspace_bytes(void) { return dense_chain_workspace_bytes_impl(); }
int srganfd_conv2d_describe(const srganfd_conv_args* a, char* out, size_t out_len) {
  if (!out || !out_len) return set_err(SRGANFD_EINVAL, "conv2d_describe: no buffer");
  out[0] = 0;
  g_describe = out; g_describe_len = out_len;
  const int rc = conv2d_impl(a, nullptr);
  g_describe = nullptr; g_describe_len = 0;
  return rc;
}

int srganfd_conv2d_thin_in(const srganfd_thin_args* a, void* stream) { return conv2d_thin_in_impl(a, (hipStream_t)stream); }
int srganfd_conv2d_thin_out(const srganfd_thin_args* a, void* stream) { return conv2d_thin_out_impl(a, (hipStream_t)stream); }
size_t srganfd_conv2d_thin_wgrad_workspace(void) { return conv2d_thin_wgrad_workspace_impl(); }
int srganfd_conv2d_thin_wgrad(const srganfd_thin_args* a, float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
  return conv2d_thin_wgrad_impl(a, dw, db, workspace, workspace_bytes, (hipStream_t)stream);
}

size_t srganfd_packed_bytes(int32_t dtype, int32_t ksize, int32_t k, int32_t n) {
  if (k <= 0 || n <= 0 || k % 32 || n % 32) return 0;
  return (size_t)ksize * ksize * k * n * (dtype == SRGANFD_F32 ? 4 : 2);
}
int srganfd_pack_weights(const srganfd_pack_job* jobs_dev, int32_t njobs, int64_t max_elems, const float* params,
                         const float* scalars, void* packed, void* stream) {
  return pack_weights_impl(jobs_dev, njobs, max_elems, params, scalars, packed, (hipStream_t)stream);
}

size_t srganfd_wgrad_plan_bytes(const srganfd_wgrad_shape* s, const srganfd_wgrad_conv* convs) {
  return wgrad_plan_bytes_impl(s, convs);
}
int srganfd_wgrad_plan_build(const srganfd_wgrad_shape* s, const srganfd_wgrad_conv* convs, void* plan_host, size_t plan_bytes,
                             size_t* workspace_bytes) {
  return wgrad_plan_build_impl(s, convs, plan_host, plan_bytes, workspace_bytes);
}
int srganfd_conv2d_wgrad_partial(const void* plan_host, const void* plan_dev, srganfd_view x, srganfd_view dy, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  return wgrad_impl(plan_host, plan_dev, x, dy, nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}
int srganfd_wgrad_reduce_batch(const srganfd_wgrad_reduce_job* jobs, int32_t njobs, void* stream) {
  return wgrad_reduce_batch_impl(jobs, njobs, (hipStream_t)stream);
}
int srganfd_conv2d_wgrad(const void* plan_host, const void* plan_dev, srganfd_view x, srganfd_view dy, float* grads,
                         const float* scalars, void* workspace, size_t workspace_bytes, void* stream) {
  if (!grads) return srganfd::set_err(SRGANFD_EINVAL, "wgrad: null gradient pointer (srganfd_conv2d_wgrad_partial is the reduce-later form)");
  return wgrad_impl(plan_host, plan_dev, x, dy, grads, scalars, workspace, workspace_bytes, (hipStream_t)stream);
}

int srganfd_nchw_to_nhwc(const float* src, int32_t n, int32_t c, int32_t h, int32_t w, srganfd_view dst, int32_t dtype, int32_t cpad,
                         const float* ch_mean, const float* ch_std, void* stream) {
  return nchw_to_nhwc_impl(src, n, c, h, w, dst, dtype, cpad, ch_mean, ch_std, (hipStream_t)stream);
}
int srganfd_lrelu_bwd(srganfd_view dy, srganfd_view act, srganfd_view skip, srganfd_view out, int32_t dtype, int64_t npix, int32_t c, float slope,
                      void* stream) {
  return lrelu_bwd_impl(dy, act, skip, out, dtype, (size_t)npix, c, slope, (hipStream_t)stream);
}
int srganfd_nhwc_to_nchw(srganfd_view src, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w, float* dst, int32_t clamp01, void* stream) {
  return nhwc_to_nchw_impl(src, dtype, n, c, h, w, dst, clamp01, (hipStream_t)stream);
}
int srganfd_clamp_grad_to_nhwc(const float* dsr_nchw, srganfd_view pre_f32, int32_t n, int32_t c, int32_t h, int32_t w, srganfd_view dst,
                               int32_t dtype, int32_t cpad, void* stream) {
  return clamp_grad_impl(dsr_nchw, pre_f32, n, c, h, w, dst, dtype, cpad, (hipStream_t)stream);
}
int srganfd_resample_bwd_lrelu(srganfd_view dy, srganfd_view dx_raw, srganfd_view act, srganfd_view dx_masked, int32_t dtype, int32_t n, int32_t h, int32_t w,
                               int32_t c, float slope, void* stream) {
  return resample_bwd_lrelu_impl(dy, dx_raw, act, dx_masked, dtype, n, h, w, c, slope, (hipStream_t)stream);
}
int srganfd_resample(int32_t op, srganfd_view a, srganfd_view b, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  return resample_impl(op, a, b, dtype, n, h, w, c, (hipStream_t)stream);
}
int srganfd_axpby(srganfd_view x, srganfd_view y, int32_t dtype, int64_t npix, int32_t c, float alpha, float beta, void* stream) {
  return axpby_impl(x, y, dtype, (size_t)npix, c, alpha, beta, (hipStream_t)stream);
}
int srganfd_l1_loss(const float* a, const float* b, int64_t numel, float weight, float* out, int32_t accumulate, float* grad_a, float grad_scale,
                    const float* grad_scale_dev, float* workspace, void* stream) {
  return l1_loss_impl(a, b, (size_t)numel, weight, out, accumulate, grad_a, grad_scale, grad_scale_dev, workspace, (hipStream_t)stream);
}
int srganfd_l1_loss_views(srganfd_view a, srganfd_view b, int32_t dtype, int64_t npix, int32_t c, int32_t relu_first, float weight, float* out,
                          int32_t accumulate, float* workspace, void* stream) {
  return l1_views_impl(a, b, dtype, (size_t)npix, c, relu_first, weight, out, accumulate, workspace, (hipStream_t)stream);
}
int srganfd_sigmoid_of_mean(const float* logits, int64_t numel, float* out, float* workspace, void* stream) {
  return sigmoid_of_mean_impl(logits, numel > 0 ? (size_t)numel : 0, out, workspace, (hipStream_t)stream);
}
int srganfd_bce_logits(const float* logits, int64_t numel, float target, float weight, float* loss_out, int32_t accumulate,
                       float* sigmoid_mean_out, float* grad, float grad_scale, const float* grad_scale_dev, float* workspace, void* stream) {
  return bce_logits_impl(logits, (size_t)numel, target, weight, loss_out, accumulate, sigmoid_mean_out, grad, grad_scale, grad_scale_dev, workspace,
                         (hipStream_t)stream);
}
int srganfd_bce_logits_relativistic(const float* x, int64_t numel, const float* other, int64_t numel_other, float target, float weight,
                                    float* loss_out, int32_t accumulate, float* grad_x, int32_t accumulate_x, float* grad_other,
                                    int32_t accumulate_other, float grad_scale, const float* grad_scale_dev, float* workspace, void* stream) {
  return bce_logits_relativistic_impl(x, numel > 0 ? (size_t)numel : 0, other, numel_other > 0 ? (size_t)numel_other : 0, target, weight, loss_out,
                                      accumulate, grad_x, accumulate_x, grad_other, accumulate_other, grad_scale, grad_scale_dev, workspace,
                                      (hipStream_t)stream);
}
int srganfd_spectral_norm(const float* w_orig, float* u, float* v, int32_t rows, int32_t cols, int32_t training, float eps, float* sigma_out,
                          float* inv_sigma_out, float* workspace, void* stream) {
  return spectral_norm_impl(w_orig, u, v, rows, cols, training, eps, sigma_out, inv_sigma_out, workspace, (hipStream_t)stream);
}
int srganfd_spectral_norm_batch(const srganfd_sn_job* jobs, int32_t njobs, int32_t training, float eps, void* stream) {
  return spectral_norm_batch_impl(jobs, njobs, training, eps, (hipStream_t)stream);
}
int srganfd_spectral_norm_grad(const float* g_weight, const float* w_orig, const float* u, const float* v, const float* inv_sigma, float* dw_orig,
                               int32_t rows, int32_t cols, float beta, float* workspace, void* stream) {
  return spectral_norm_grad_impl(g_weight, w_orig, u, v, inv_sigma, dw_orig, rows, cols, beta, workspace, (hipStream_t)stream);
}
int srganfd_spectral_norm_grad_batch(const srganfd_sn_grad_job* jobs, int32_t njobs, float beta, void* stream) {
  return spectral_norm_grad_batch_impl(jobs, njobs, beta, (hipStream_t)stream);
}
int srganfd_adam_ema(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, int64_t numel, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int32_t step, float grad_scale, float ema_decay, int32_t ema_mode,
                     const float* skip_flag, const float* grad_scale_dev, void* stream) {
  return adam_ema_impl(param, grad, exp_avg, exp_avg_sq, ema, (size_t)numel, lr, beta1, beta2, eps, weight_decay, step, grad_scale, ema_decay,
                       ema_mode, skip_flag, grad_scale_dev, (hipStream_t)stream);
}
int srganfd_loss_scale_update(float* state, const float* found_inf, float growth_factor, float backoff_factor, int32_t growth_interval, void* stream) {
  return loss_scale_update_impl(state, found_inf, growth_factor, backoff_factor, growth_interval, (hipStream_t)stream);
}
int srganfd_nonfinite_flag(const float* x, int64_t numel, float* flag, int32_t accumulate, void* stream) {
  return nonfinite_flag_impl(x, (size_t)numel, flag, accumulate, (hipStream_t)stream);
}

int srganfd_resize_bilinear(int32_t bwd, srganfd_view a, srganfd_view b, int32_t dtype, int32_t n, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                            int32_t c, void* stream) {
  return resize_bilinear_impl(bwd, a, b, dtype, n, hi, wi, ho, wo, c, (hipStream_t)stream);
}
int srganfd_adam_ema_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, int64_t numel, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int32_t* step_dev, float* bc_dev, float grad_scale, float ema_decay,
                         int32_t ema_mode, const float* skip_flag, const float* grad_scale_dev, void* stream) {
  return adam_ema_dev_impl(param, grad, exp_avg, exp_avg_sq, ema, (size_t)numel, lr, beta1, beta2, eps, weight_decay, step_dev, bc_dev, grad_scale,
                           ema_decay, ema_mode, skip_flag, grad_scale_dev, (hipStream_t)stream);
}
int srganfd_l1_grad_views(srganfd_view a, srganfd_view b, srganfd_view out, int32_t dtype, int64_t npix, int32_t c, const float* upstream,
                          float scale, void* stream) {
  return l1_grad_views_impl(a, b, out, dtype, (size_t)npix, c, upstream, scale, (hipStream_t)stream);
}
int srganfd_maxpool2_relu_bwd(srganfd_view x, srganfd_view dy, srganfd_view dx, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c,
                              void* stream) {
  return maxpool2_relu_bwd_impl(x, dy, dx, dtype, n, h, w, c, (hipStream_t)stream);
}
int srganfd_nhwc_to_nchw_scaled(srganfd_view src_f32, int32_t n, int32_t c, int32_t h, int32_t w, float* dst, const float* ch_div, void* stream) {
  return nhwc_to_nchw_scaled_impl(src_f32, n, c, h, w, dst, ch_div, (hipStream_t)stream);
}
int srganfd_crop_nchw(const float* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t top, int32_t left, int32_t ph, int32_t pw,
                      void* stream) {
  return crop_nchw_impl(src, dst, n, c, h, w, top, left, ph, pw, (hipStream_t)stream);
}
int srganfd_u8hwc_to_nchw(const unsigned char* src, float* dst, int32_t n, int32_t h, int32_t w, int32_t top, int32_t left, int32_t ph, int32_t pw,
                          int32_t swap_rb, float scale, void* stream) {
  return u8hwc_to_nchw_impl(src, dst, n, h, w, top, left, ph, pw, swap_rb, scale, (hipStream_t)stream);
}
int srganfd_psnr(const float* a, const float* b, int32_t n, int32_t c, int32_t h, int32_t w, int32_t crop_border, int32_t y_only, double* out,
                 double* workspace, void* stream) {
  return psnr_impl(a, b, n, c, h, w, crop_border, y_only, out, workspace, (hipStream_t)stream);
}
int64_t srganfd_ssim_workspace_doubles(int32_t n, int32_t c, int32_t h, int32_t w, int32_t crop_border, int32_t y_only, int32_t window_size) {
  return ssim_workspace_doubles(n, c, h, w, crop_border, y_only, window_size);
}
int srganfd_ssim(const float* a, const float* b, int32_t n, int32_t c, int32_t h, int32_t w, int32_t crop_border, int32_t y_only, const double* window,
                 int32_t window_size, float* out, double* workspace, void* stream) {
  return ssim_impl(a, b, n, c, h, w, crop_border, y_only, window, window_size, out, workspace, (hipStream_t)stream);
}
int srganfd_filter2d(const float* image, const float* kernels, int32_t kernel_batch, int32_t b, int32_t c, int32_t h, int32_t w, int32_t k, float* out,
                     void* stream) {
  return filter2d_impl(image, kernels, kernel_batch, b, c, h, w, k, 0, nullptr, nullptr, 0.f, 0.f, out, nullptr, (hipStream_t)stream);
}
int srganfd_usm_sharp(const float* image, const float* kernel, int32_t separable, int32_t b, int32_t c, int32_t h, int32_t w, int32_t k, float weight,
                      float threshold, float* out, float* workspace, void* stream) {
  if (!workspace) return set_err(SRGANFD_EINVAL, "usm_sharp: workspace of 2 * b*c*h*w floats needed");
  const size_t n = (size_t)b * c * h * w;
  float* residual = workspace;
  float* mask = workspace + n;
  int rc = filter2d_impl(image, kernel, 1, b, c, h, w, k, 1, nullptr, nullptr, weight, threshold, residual, mask, (hipStream_t)stream, separable != 0);
  if (rc != SRGANFD_OK) return rc;
  return filter2d_impl(mask, kernel, 1, b, c, h, w, k, 2, image, residual, weight, threshold, out, nullptr, (hipStream_t)stream, separable != 0);
}
int srganfd_filter2d_separable(const float* image, const float* taps, int32_t kernel_batch, int32_t b, int32_t c, int32_t h, int32_t w, int32_t k,
                               float* out, void* stream) {
  return filter2d_impl(image, taps, kernel_batch, b, c, h, w, k, 0, nullptr, nullptr, 0.f, 0.f, out, nullptr, (hipStream_t)stream, true);
}
int32_t srganfd_diff_jpeg_table_floats(void) { return jpeg_table_floats(); }
int srganfd_diff_jpeg_tables(float* host_out) {
  if (!host_out) return set_err(SRGANFD_EINVAL, "diff_jpeg_tables: null output");
  diff_jpeg_tables_host(host_out);
  return SRGANFD_OK;
}
int srganfd_diff_jpeg(const float* image, int32_t b, int32_t c, int32_t h, int32_t w, float* quality, int32_t quality_is_factor,
                      int32_t differentiable, const float* tables, float* out, void* stream) {
  return diff_jpeg_impl(image, b, c, h, w, quality, quality_is_factor, differentiable, tables, out, (hipStream_t)stream);
}
int srganfd_resize(const float* src, int32_t planes, int32_t h, int32_t w, int32_t out_h, int32_t out_w, int32_t mode, float rscale_h, float rscale_w,
                   float* dst, void* stream) {
  return resize_impl(src, planes, h, w, out_h, out_w, mode, rscale_h, rscale_w, dst, (hipStream_t)stream);
}
int srganfd_gaussian_noise(const float* image, const float* randn_color, const float* randn_gray_hw, const float* sigma, const float* gray_flag, int32_t b,
                           int32_t c, int32_t h, int32_t w, int32_t clip, int32_t rounds, float* out, void* stream) {
  return gaussian_noise_impl(image, randn_color, randn_gray_hw, sigma, gray_flag, b, c, h, w, clip, rounds, out, (hipStream_t)stream);
}
int srganfd_poisson_prepare(const float* image, int32_t b, int32_t c, int32_t h, int32_t w, int32_t want_gray, float* image_q, float* gray_q, float* vals,
                            float* vals_gray, void* workspace, void* stream) {
  return poisson_prepare_impl(image, b, c, h, w, want_gray, image_q, gray_q, vals, vals_gray, (unsigned int*)workspace, (hipStream_t)stream);
}
int srganfd_poisson_apply(const float* image, const float* image_q, const float* gray_q, const float* poisson_color, const float* poisson_gray,
                          const float* vals, const float* vals_gray, const float* scale, const float* gray_flag, int32_t b, int32_t c, int32_t h,
                          int32_t w, int32_t clip, int32_t rounds, float* out, void* stream) {
  return poisson_apply_impl(image, image_q, gray_q, poisson_color, poisson_gray, vals, vals_gray, scale, gray_flag, b, c, h, w, clip, rounds, out,
                            (hipStream_t)stream);
}
int srganfd_crop_rot_flip(const float* src, float* dst, int32_t planes, int32_t h, int32_t w, int32_t top, int32_t left, int32_t ph, int32_t pw,
                          int32_t op, void* stream) {
  return crop_rot_flip_impl(src, dst, planes, h, w, top, left, ph, pw, op, (hipStream_t)stream);
}
int srganfd_quantize_u8(const float* src, float* dst, int64_t numel, void* stream) {
  return quantize_u8_impl(src, dst, numel > 0 ? (size_t)numel : 0, (hipStream_t)stream);
}
int srganfd_add_relu(srganfd_view a, srganfd_view b, srganfd_view out, int32_t dtype, int64_t npix, int32_t c, void* stream) {
  return add_relu_impl(a, b, out, dtype, (size_t)npix, c, (hipStream_t)stream);
}
int srganfd_sigmoid(float* x, int64_t numel, void* stream) { return sigmoid_impl(x, (size_t)numel, (hipStream_t)stream); }
int srganfd_sigmoid_bwd(const float* ds, const float* s, float* out, int64_t numel, void* stream) {
  return sigmoid_bwd_impl(ds, s, out, (size_t)numel, (hipStream_t)stream);
}
int srganfd_gate_mul(int32_t bwd, srganfd_view x, const float* gate, srganfd_view y, srganfd_view dx, float* dgate, int32_t dtype, int64_t npix,
                     int32_t c, void* stream) {
  return gate_mul_impl(bwd, x, gate, y, dx, dgate, dtype, (size_t)npix, c, (hipStream_t)stream);
}
int srganfd_batchnorm_fwd(srganfd_view x, srganfd_view y, int32_t dtype, int64_t npix, int32_t c, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, int32_t training, float* save, float* workspace,
                          void* stream) {
  return batchnorm_fwd_impl(x, y, dtype, (size_t)npix, c, gamma, beta, running_mean, running_var, momentum, eps, training, save, workspace,
                            1.f, (hipStream_t)stream);
}
int srganfd_batchnorm_act_fwd(srganfd_view x, srganfd_view y, int32_t dtype, int64_t npix, int32_t c, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, int32_t training, float* save,
                              float* workspace, float act_slope, void* stream) {
  return batchnorm_fwd_impl(x, y, dtype, (size_t)npix, c, gamma, beta, running_mean, running_var, momentum, eps, training, save, workspace,
                            act_slope, (hipStream_t)stream);
}
int64_t srganfd_batchnorm_partial_floats(int32_t c) { return batchnorm_partial_floats_impl(c); }
int srganfd_batchnorm_fwd_sync(srganfd_view x, srganfd_view y, int32_t dtype, int64_t npix, int32_t c, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, float* save, float* workspace, float act_slope,
                               int32_t phase, int64_t total_npix, void* stream) {
  return batchnorm_fwd_impl(x, y, dtype, (size_t)npix, c, gamma, beta, running_mean, running_var, momentum, eps, 1, save, workspace, act_slope,
                            (hipStream_t)stream, phase, (size_t)total_npix);
}
int srganfd_batchnorm_bwd_sync(srganfd_view x, srganfd_view dy, srganfd_view dx, int32_t dtype, int64_t npix, int32_t c, const float* gamma,
                               const float* save, float* dgamma, float* dbeta, float acc, float* workspace, const float* workspace_global,
                               srganfd_view act, float act_slope, int32_t phase, int64_t total_npix, void* stream) {
  return batchnorm_bwd_impl(x, dy, dx, dtype, (size_t)npix, c, gamma, save, dgamma, dbeta, acc, workspace, act, act_slope, (hipStream_t)stream, phase,
                            workspace_global, (size_t)total_npix);
}
int srganfd_batchnorm_bwd(srganfd_view x, srganfd_view dy, srganfd_view dx, int32_t dtype, int64_t npix, int32_t c, const float* gamma,
                          const float* save, float* dgamma, float* dbeta, float acc, float* workspace, void* stream) {
  srganfd_view none = {nullptr, 0, 0};
  return batchnorm_bwd_impl(x, dy, dx, dtype, (size_t)npix, c, gamma, save, dgamma, dbeta, acc, workspace, none, 1.f, (hipStream_t)stream);
}
int srganfd_batchnorm_act_bwd(srganfd_view x, srganfd_view dy, srganfd_view dx, int32_t dtype, int64_t npix, int32_t c, const float* gamma,
                              const float* save, float* dgamma, float* dbeta, float acc, float* workspace, srganfd_view act,
                              float act_slope, void* stream) {
  return batchnorm_bwd_impl(x, dy, dx, dtype, (size_t)npix, c, gamma, save, dgamma, dbeta, acc, workspace, act, act_slope, (hipStream_t)stream);
}

}  // extern "C"
