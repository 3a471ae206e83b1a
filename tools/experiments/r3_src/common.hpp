// common.hpp -- shared device helpers for the gfx950 (CDNA4) SR-GAN-FD kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "srganfd.h"   // the round-3 header with the experiment exports

namespace srganfd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

typedef unsigned short bf16_t;  // storage type for bf16
typedef _Float16 f16_t;         // IEEE half (the reference's CUDA autocast dtype, train_bsrgan.py:415-467): same MFMA rate and bytes as
                                // bf16, 3 more mantissa bits -- the mode whose SR meets the 1e-3 tolerance; gradients need loss scaling
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// round-to-nearest-even fp32 -> bf16 (plain cast keeps NaN a NaN: MI355X guide, correctness table)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(bf16_t v) {
  return __builtin_bit_cast(float, ((unsigned int)v) << 16);
}

template <typename T> struct Elem;
template <> struct Elem<bf16_t> {
  static constexpr int kDtype = SRGANFD_BF16;
  static constexpr int kStep = 16;  // K per MFMA (v_mfma_f32_32x32x16_bf16)
  __device__ static __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
  __device__ static __forceinline__ bf16_t from_f(float f) { return f2bf(f); }
};
template <> struct Elem<f16_t> {
  static constexpr int kDtype = SRGANFD_F16;
  static constexpr int kStep = 16;  // K per MFMA (v_mfma_f32_32x32x16_f16)
  __device__ static __forceinline__ float to_f(f16_t v) { return (float)v; }
  __device__ static __forceinline__ f16_t from_f(float f) { return (f16_t)f; }   // round-to-nearest-even; overflow -> inf (caught by the loss scaler)
};
template <> struct Elem<float> {
  static constexpr int kDtype = SRGANFD_F32;
  static constexpr int kStep = 2;   // K per MFMA (v_mfma_f32_32x32x2_f32, exact fp32 fma chain)
  __device__ static __forceinline__ float to_f(float v) { return v; }
  __device__ static __forceinline__ float from_f(float f) { return f; }
};

// 8 consecutive 16-bit elements (one 16-byte chunk) <-> 8 floats
template <typename T> __device__ __forceinline__ void unpack8(const u32x4 raw, float* out);
template <> __device__ __forceinline__ void unpack8<bf16_t>(const u32x4 raw, float* out) {
#pragma unroll
  for (int q = 0; q < 4; ++q) { out[2 * q] = __uint_as_float(raw[q] << 16); out[2 * q + 1] = __uint_as_float(raw[q] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void unpack8<f16_t>(const u32x4 raw, float* out) {
  const f16x8 hv = __builtin_bit_cast(f16x8, raw);
#pragma unroll
  for (int q = 0; q < 8; ++q) out[q] = (float)hv[q];
}
template <typename T> __device__ __forceinline__ u32x4 pack8(const float* v);
template <> __device__ __forceinline__ u32x4 pack8<bf16_t>(const float* v) {
  u32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = (unsigned)f2bf(v[2 * q]) | ((unsigned)f2bf(v[2 * q + 1]) << 16);
  return o;
}
template <> __device__ __forceinline__ u32x4 pack8<f16_t>(const float* v) {
  f16x8 hv;
#pragma unroll
  for (int q = 0; q < 8; ++q) hv[q] = (_Float16)v[q];
  return __builtin_bit_cast(u32x4, hv);
}

// MFMA 32x32 C/D layout (dtype independent on gfx950): col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
__device__ __forceinline__ int mfma32_row(int reg, int lane) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

extern thread_local char g_err[512];
#ifdef SRGANFD_EXPERIMENT
extern int g_debug;    // srganfd_set_debug: kernel timing experiments, results are WRONG when non-zero
#endif
extern int g_dry_run;  // srganfd_set_dry_run(1): validate arguments, build plans, launch nothing (host-logic tests on CPU)
int set_err(int code, const char* fmt, ...);
// srganfd_conv2d_describe: the launch functions write the label of the kernel they would run here instead of launching
extern thread_local char* g_describe;
extern thread_local size_t g_describe_len;
template <typename T> inline const char* dtype_name() { return sizeof(T) == 4 ? "f32" : (Elem<T>::kDtype == SRGANFD_F16 ? "f16" : "bf16"); }

#define SRGANFD_HIP_CHECK(expr)                                                          \
  do {                                                                                   \
    if (srganfd::g_dry_run) break;                                                       \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess)                                                                \
      return srganfd::set_err(SRGANFD_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, \
                              hipGetErrorString(_e));                                    \
  } while (0)

#define SRGANFD_LAUNCH(...) do { if (!srganfd::g_dry_run) hipLaunchKernelGGL(__VA_ARGS__); } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace srganfd
