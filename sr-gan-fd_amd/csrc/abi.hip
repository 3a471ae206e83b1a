// abi.hip -- extern "C" surface of libsrganfd_hip.so (see include/srganfd.h).
#include "common.hpp"
#include <stdarg.h>

namespace srganfd {
thread_local char g_err[512] = {0};
int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int conv2d_impl(const srganfd_conv_args* a, hipStream_t stream);
int pack_weights_impl(const srganfd_pack_job* jobs_dev, int njobs, long long max_elems, const float* params,
                      const float* scalars, void* packed, hipStream_t stream);
size_t wgrad_plan_bytes_impl(const srganfd_wgrad_shape* s, const srganfd_wgrad_conv* convs);
int wgrad_plan_build_impl(const srganfd_wgrad_shape* s, const srganfd_wgrad_conv* convs, void* plan_host, size_t plan_bytes,
                          size_t* workspace_bytes);
int wgrad_impl(const void* plan_host, const void* plan_dev, srganfd_view x, srganfd_view dy, float* grads, const float* scalars,
               void* workspace, size_t workspace_bytes, hipStream_t stream);
}  // namespace srganfd

using namespace srganfd;

extern "C" {

const char* srganfd_last_error(void) { return g_err; }
int srganfd_abi_version(void) { return 1; }

int srganfd_conv2d(const srganfd_conv_args* a, void* stream) { return conv2d_impl(a, (hipStream_t)stream); }

size_t srganfd_packed_bytes(int32_t dtype, int32_t ksize, int32_t k, int32_t n) {
  if (k <= 0 || n <= 0 || k % 32 || n % 32) return 0;
  return (size_t)ksize * ksize * k * n * (dtype == SRGANFD_BF16 ? 2 : 4);
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
int srganfd_conv2d_wgrad(const void* plan_host, const void* plan_dev, srganfd_view x, srganfd_view dy, float* grads,
                         const float* scalars, void* workspace, size_t workspace_bytes, void* stream) {
  return wgrad_impl(plan_host, plan_dev, x, dy, grads, scalars, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
