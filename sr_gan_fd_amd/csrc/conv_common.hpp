// conv_common.hpp -- launch arguments and MFMA helpers shared by the fused-convolution kernels
// (conv_igemm.hip: every kernel shape; conv_thin.hip; wgrad.hip).  The kernel-timing experiment harness of rounds 1-3 (debug ablation
// switches, s_memrealtime stamps, ring / stream / chain / pair-fusion kernels) lives in tools/experiments/ with its own copy of these sources.
#pragma once
#include <utility>
#include "common.hpp"

namespace srganfd {

struct ConvK {
  const void* x; void* y; void* y2; const void* r1; const void* r2; const void* mask; const void* w;
  const float* bias; const float* alpha_dev;
  int xC, x_c0, yC, y_c0, y2C, y2_c0, r1C, r1_c0, r2C, r2_c0, mC, m_c0;
  // element (pixel p, channel c) of an operand's image sits at p * ps + (c >> 5) * gs + (c & 31): NHWC ps = C, gs = 32;
  // planar 32-channel groups (srganfd_view.planar) ps = 32, gs = H*W*32.  x additionally: first chunk at x_base, next at + x_cs.
  int x_ps, x_base, x_cs, y_ps, y_gs, y2_ps, y2_gs, r1_ps, r1_gs, r2_ps, r2_gs, m_ps, m_gs;
  int N, Hin, Win, up, pad_y, pad_x, Hout, Wout;
  int osy, osx, ooy, oox, HoutF, WoutF;  // output pixel (oy,ox) is stored at (oy*osy+ooy, ox*osx+oox) of a HoutF x WoutF image
  int nChunks;        // cin / 32
  int nNb;            // cout / (output channels per workgroup); x 4 in a class launch
  int cls_sh;         // -1, or log2(channel blocks per class) of a launch that runs all four output-parity classes (out_classes == 4)
  int cls_pad;        // class (py,px) of such a launch pads by pad_y - py * cls_pad, pad_x - px * cls_pad
  int cout_store;
  int tiles_x, tiles_y;
  int nblocks;        // N * tiles_y * tiles_x * nNb virtual blocks; the grid may be smaller (persistent workgroups, see the kernel)
  unsigned m_nNb, m_tx, m_ty;   // ceil(2^32 / d) for the block-index decode (0: divide), see fast_div
  float alpha, slope, post_scale, r1s, r2s, mask_slope;
  int act, y_f32, fast_epi;
};

template <typename T> struct FragAB;
template <> struct FragAB<bf16_t> { typedef bf16x8 type; };
template <> struct FragAB<f16_t> { typedef f16x8 type; };
template <> struct FragAB<float> { typedef float type; };

template <typename T> __device__ __forceinline__ f32x16 mfma32(typename FragAB<T>::type a, typename FragAB<T>::type b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mfma32<bf16_t>(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma32<f16_t>(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma32<float>(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// v_mfma_f32_16x16x32: K = 32 channels per instruction at half the cycles of 32x32x16.  The micro-architecture guide measures
// 1.12-1.15x the FLOP/s of the 32x32x16 form on random data at equal cycles per FLOP (the chip holds a higher clock on it), and
// this path runs at 1.2 kW of a 1.4 kW cap with the clock pulled down to 2.06 GHz (profiles/r02_power_clock_samples.txt).
// A operand: lane l holds A[row l&15][k = 8*(l>>4) + j]; B: B[k = 8*(l>>4) + j][col l&15]; C/D: col = l&15, row = 4*(l>>4) + reg.
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <typename T> __device__ __forceinline__ f32x4_t mfma16(typename FragAB<T>::type a, typename FragAB<T>::type b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t mfma16<bf16_t>(bf16x8 a, bf16x8 b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mfma16<f16_t>(f16x8 a, f16x8 b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// n / d for block-uniform n with the host's magic number m = ceil(2^32 / d): exact while n * d < 2^32 (the host passes 0 otherwise, and
// for d == 1).  A runtime 32-bit division is a v_rcp_iflag + fix-up sequence of ~20 instructions on the vector pipe.
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned d, unsigned m) { return m ? __umulhi(n, m) : n / d; }
inline unsigned div_magic(unsigned d, unsigned long long n_max) {
  if (d <= 1 || n_max * d >= (1ull << 32)) return 0u;
  return (unsigned)(((1ull << 32) + d - 1) / d);
}
bool conv_uses_m16(int dtype, int ksize, int cout);
extern int g_mfma16;   // 1: the 3x3 16-bit convolutions run on the 16x16x32 form (weights packed in its B-fragment order: srganfd_pack_job.layout)

// compile-time loop (indices as types), for the software-pipelined MFMA phase
template <int I> struct IC { static constexpr int v = I; };
template <class Fn, int... Is> __device__ __forceinline__ void static_for_impl(Fn&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
template <int N, class Fn> __device__ __forceinline__ void static_for(Fn&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// last fragment load issued before MFMA i: the fragment it needs (7 loads / 6 MFMAs per column body) plus kD of read-ahead
// 16x16x32 form, 3x3 stride 1, two rows per wave: fixed issue order of one kernel column's 14 fragment reads and 24 MFMAs.
// reads:  B00 A00 A01 A10 A11 | B01 | B10 A20 A21 | B11 | B20 A30 A31 | B21      (B[ky][channel half], A[patch row][pixel half])
// MFMAs:  group (ky, nh) = 4 MFMAs (row m, pixel half ph): acc[m][ph][nh] += A[m + ky][ph] x B[ky][nh]  -- eight independent
//         accumulators between two uses of the same one.
__host__ __device__ constexpr int m16_bidx(int ky, int nh) { return ky == 0 ? (nh ? 5 : 0) : ky == 1 ? (nh ? 9 : 6) : (nh ? 13 : 10); }
__host__ __device__ constexpr int m16_aidx(int rr, int ph) { return (rr == 0 ? 1 : rr == 1 ? 3 : rr == 2 ? 7 : 11) + ph; }
__host__ __device__ constexpr int m16_need(int j) {
  const int gq = j / 4, t = j % 4, a = m16_aidx((t >> 1) + (gq >> 1), t & 1), b = m16_bidx(gq >> 1, gq & 1);
  return a > b ? a : b;
}
__host__ __device__ constexpr int m16_pipe_hi(int i, int d, int nl) { const int need = 14 * (i / 24) + m16_need(i % 24); return need + d < nl - 1 ? need + d : nl - 1; }
__host__ __device__ constexpr int pipe_hi(int i, int d, int nl) { const int need = 7 * (i / 6) + (i % 6) + 1; return need + d < nl - 1 ? need + d : nl - 1; }

// XCD-aware bijective remap of a 1-D grid: blocks b and b+8 share an XCD (private L2); give each XCD a contiguous range
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
  return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
}

int conv_device_cus();   // compute units of the current device (conv_igemm.hip)

}  // namespace srganfd
