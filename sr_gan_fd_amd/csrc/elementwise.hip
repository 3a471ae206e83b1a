// elementwise.hip -- the bandwidth-bound pieces of the hot path that are not fused into a conv epilogue:
// layout/dtype conversion at the module boundary (NCHW fp32 <-> NHWC bf16/f32), clamp (+ its gradient
// mask), nearest/bilinear x2 resampling gradients, losses (L1, BCE-with-logits), spectral-norm power
// iteration and its gradient, max-pool, and the fused Adam + EMA update over flat parameter buffers.
// All kernels are grid-stride, vectorised where the layout allows, and accumulate in fp32.
// Reductions are two-stage (per-block partials, then one block) -> bitwise reproducible, no atomics.
#include "common.hpp"
#include <initializer_list>

namespace srganfd {

template <typename T> __device__ __forceinline__ float ld(const void* p, size_t i) { return Elem<T>::to_f(((const T*)p)[i]); }
template <typename T> __device__ __forceinline__ void st(void* p, size_t i, float v) { ((T*)p)[i] = Elem<T>::from_f(v); }

__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  return r;  // valid on thread 0
}

// ---- 16-byte vector access (8 bf16 / 4 f32 channels per lane): every view-to-view kernel below has a
// vector form used whenever channel count, view offset and buffer stride are multiples of VecN<T>.
template <typename T> struct VecN { static constexpr int N = 16 / (int)sizeof(T); };
template <typename T> __device__ __forceinline__ void ldv(const void* p, size_t i, float* o) {
  if constexpr (sizeof(T) == 2) {
    unpack8<T>(*(const u32x4*)((const T*)p + i), o);
  } else {
    const f32x4 r = *(const f32x4*)((const float*)p + i);
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
  }
}
// non-temporal form: outputs that are written once and are far larger than the caches (the upsampled U-Net tensors: 0.5-2 GB)
template <typename T> __device__ __forceinline__ void stv_nt(void* p, size_t i, const float* v) {
  if constexpr (sizeof(T) == 2) {
    __builtin_nontemporal_store(pack8<T>(v), (u32x4*)((T*)p + i));
  } else {
    const f32x4 o = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(o, (f32x4*)((float*)p + i));
  }
}
template <typename T> __device__ __forceinline__ void stv(void* p, size_t i, const float* v) {
  if constexpr (sizeof(T) == 2) {
    *(u32x4*)((T*)p + i) = pack8<T>(v);
  } else {
    const f32x4 o = {v[0], v[1], v[2], v[3]};
    *(f32x4*)((float*)p + i) = o;
  }
}

// generic 2-D resampling on vectors: op 0 nearest-x2 backward, 1 bilinear-x2 forward, 2 bilinear-x2 backward, 3 maxpool2, 4 relu
__device__ __forceinline__ void bil_taps(int d, int n, int& i0, int& i1, float& w0, float& w1);
__device__ __forceinline__ int bil_bwd_taps(int k, int n, int* d, float* wt);

// OP 2 only: act != NULL also writes b2 = result * (act > 0 ? 1 : slope) (LeakyReLU' of the layer whose output was upsampled: the
// raw gradient b is the U-Net skip's share, model.py:153,157,161; b may be NULL when only the masked one is wanted)
template <typename T, int OP>
__global__ __launch_bounds__(256) void resample_vec_kernel(const void* __restrict__ a, int aC, int a0, void* b, int bC, int b0, int n, int h, int w, int c,
                                                           const void* __restrict__ act = nullptr, int actC = 0, int act0 = 0, void* b2 = nullptr, int b2C = 0,
                                                           int b20 = 0, float slope = 0.f) {
  constexpr int N = VecN<T>::N;
  const int cv = c / N;
  // output extents: op0/2 -> (h, w) low-res ; op1 -> (2h, 2w) ; op3 -> (h/2, w/2) ; op4 -> (h, w)
  const int oh = OP == 1 ? 2 * h : (OP == 3 ? h / 2 : h), ow = OP == 1 ? 2 * w : (OP == 3 ? w / 2 : w);
  const size_t total = (size_t)n * oh * ow * cv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % cv) * N;
    size_t p = i / cv;
    const int ox = (int)(p % ow); p /= ow;
    const int oy = (int)(p % oh);
    const size_t img = p / oh;
    float acc[N], t[N];
    if constexpr (OP == 0) {          // sum of the 2x2 high-res pixels
      const size_t bq = (img * 2 * h + 2 * oy) * 2 * w + 2 * ox;
      ldv<T>(a, bq * aC + a0 + ch, acc);
      ldv<T>(a, (bq + 1) * aC + a0 + ch, t);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] += t[q];
      ldv<T>(a, (bq + 2 * w) * aC + a0 + ch, t);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] += t[q];
      ldv<T>(a, (bq + 2 * w + 1) * aC + a0 + ch, t);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] += t[q];
    } else if constexpr (OP == 1) {   // bilinear forward
      int ya, yb, xa, xb; float wya, wyb, wxa, wxb;
      bil_taps(oy, h, ya, yb, wya, wyb);
      bil_taps(ox, w, xa, xb, wxa, wxb);
      const size_t r0 = (img * h + ya) * w, r1 = (img * h + yb) * w;
      ldv<T>(a, (r0 + xa) * aC + a0 + ch, t);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] = wya * wxa * t[q];
      ldv<T>(a, (r0 + xb) * aC + a0 + ch, t);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] += wya * wxb * t[q];
      ldv<T>(a, (r1 + xa) * aC + a0 + ch, t);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] += wyb * wxa * t[q];
      ldv<T>(a, (r1 + xb) * aC + a0 + ch, t);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] += wyb * wxb * t[q];
    } else if constexpr (OP == 2) {   // bilinear backward (gather form)
      int dys[6], dxs[6]; float wys[6], wxs[6];
      const int ny = bil_bwd_taps(oy, h, dys, wys), nx = bil_bwd_taps(ox, w, dxs, wxs);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] = 0.f;
      for (int ia = 0; ia < ny; ++ia)
        for (int ib = 0; ib < nx; ++ib) {
          ldv<T>(a, ((img * 2 * h + dys[ia]) * 2 * w + dxs[ib]) * aC + a0 + ch, t);
          const float ww = wys[ia] * wxs[ib];
#pragma unroll
          for (int q = 0; q < N; ++q) acc[q] += ww * t[q];
        }
    } else if constexpr (OP == 3) {   // 2x2 max pool
      const size_t bq = (img * h + 2 * oy) * w + 2 * ox;
      ldv<T>(a, bq * aC + a0 + ch, acc);
      ldv<T>(a, (bq + 1) * aC + a0 + ch, t);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] = fmaxf(acc[q], t[q]);
      ldv<T>(a, (bq + w) * aC + a0 + ch, t);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] = fmaxf(acc[q], t[q]);
      ldv<T>(a, (bq + w + 1) * aC + a0 + ch, t);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] = fmaxf(acc[q], t[q]);
    } else {                           // relu copy
      ldv<T>(a, ((img * h + oy) * w + ox) * aC + a0 + ch, acc);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] = fmaxf(acc[q], 0.f);
    }
    if constexpr (OP == 2) {
      const size_t op_ = (img * oh + oy) * ow + ox;
      if (b) stv<T>(b, op_ * (size_t)bC + b0 + ch, acc);
      if (act) {
        ldv<T>(act, op_ * (size_t)actC + act0 + ch, t);
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] *= t[q] > 0.f ? 1.f : slope;
        stv<T>(b2, op_ * (size_t)b2C + b20 + ch, acc);
      }
    } else {
      stv<T>(b, ((img * oh + oy) * ow + ox) * (size_t)bC + b0 + ch, acc);
    }
  }
}

// ---- bilinear x2 (align_corners=False, model.py:150-158), row-grid forms: blockIdx.y = group of kBilRows low-res rows, blockIdx.z = image, one thread per
// (low-res column, 16-byte channel vector).  The generic kernel above spends its time on 64-bit div/mod chains and per-thread tap tables;
// here the row taps are wave-uniform, the column taps closed-form, and every address is 32-bit arithmetic on top of one 64-bit row base.
// Same products and the same accumulation order per output as resample_vec_kernel<T, 1 / 2>: results are bit-identical.
// Each thread walks kBilRows consecutive low-res rows with a sliding window of source rows in registers: the forward pass reads
// (R + 2) x 3 vectors for 4R stores (2x2 high-res block per low-res pixel), the adjoint (2R + 2) x 4 for R.
static constexpr int kBilRows = 4;
// raw 16-byte vector (8 halves or 4 floats), widened to floats where it is used
template <typename T> __device__ __forceinline__ u32x4 ldraw(const void* p, size_t i) { return *(const u32x4*)((const T*)p + i); }
template <typename T> __device__ __forceinline__ void widen(const u32x4 raw, float* o) {
  if constexpr (sizeof(T) == 2) unpack8<T>(raw, o);
  else { o[0] = __uint_as_float(raw[0]); o[1] = __uint_as_float(raw[1]); o[2] = __uint_as_float(raw[2]); o[3] = __uint_as_float(raw[3]); }
}
template <typename T>
__global__ __launch_bounds__(256) void bilinear_up2_block_kernel(const void* __restrict__ a, int aC, int a0, void* b, int bC, int b0, int h, int w, int c, int cv_shift) {
  constexpr int N = VecN<T>::N, R = kBilRows;
  const int cv = c / N;
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (unsigned)(w * cv)) return;
  const int kx = cv_shift >= 0 ? (int)(i >> cv_shift) : (int)(i / (unsigned)cv), ch = ((int)i - kx * cv) * N;
  const int ky0 = blockIdx.y * R;
  const size_t img = blockIdx.z;
  const int xs[3] = {max(kx - 1, 0), kx, min(kx + 1, w - 1)};
  float t[3][3][N];                                         // window slot (row - ky0 + 1) % 3
  auto load_row = [&](float (*dst)[N], int y) {
    const size_t row = (img * h + y) * (size_t)w;
#pragma unroll
    for (int q = 0; q < 3; ++q) ldv<T>(a, (row + xs[q]) * aC + a0 + ch, dst[q]);
  };
  load_row(t[0], max(ky0 - 1, 0));
  load_row(t[1], ky0);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ky = ky0 + r;
    if (ky >= h) break;                                     // wave-uniform
    load_row(t[(r + 2) % 3], min(ky + 1, h - 1));
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      // even output row: taps (k-1: .25, k: .75); odd: (k: .75, k+1: .25)
      const int ra = (r + dy) % 3, rb = (r + dy + 1) % 3;
      const float wya = dy ? 0.75f : 0.25f, wyb = dy ? 0.25f : 0.75f;
      const size_t orow = (img * 2 * h + 2 * ky + dy) * (size_t)(2 * w) + 2 * kx;
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int qa = dx, qb = dx + 1;
        const float wxa = dx ? 0.75f : 0.25f, wxb = dx ? 0.25f : 0.75f;
        float acc[N];
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] = wya * wxa * t[ra][qa][q];
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] += wya * wxb * t[ra][qb][q];
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] += wyb * wxa * t[rb][qa][q];
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] += wyb * wxb * t[rb][qb][q];
        // non-temporal: 142 -> 107 us (512 channels, 64^2 -> 128^2), 268 -> 199 us (256 channels), 525 -> 505 us (128 channels), bit-equal outputs
        stv_nt<T>(b, (orow + dx) * (size_t)bC + b0 + ch, acc);
      }
    }
  }
}
// adjoint (gather form): low-res pixel k collects high-res 2k-1 .. 2k+2 with (.25, .75, .75, .25); at the borders the clamped taps fold
// into the edge pixel (weight 1) and the out-of-range tap is dropped -- the table bil_bwd_taps builds, in closed form.
__device__ __forceinline__ void bil_bwd_taps4(int k, int n, int* d, float* wt, bool* on) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int dd = 2 * k - 1 + j;
    on[j] = dd >= 0 && dd < 2 * n;
    d[j] = dd;
    wt[j] = (j == 0 || j == 3) ? 0.25f : 0.75f;
  }
  if (k == 0) wt[1] = 1.0f;
  if (k == n - 1) wt[2] = 1.0f;
}
template <typename T, int R = kBilRows, bool NT = false>
__global__ __launch_bounds__(256) void bilinear_up2_bwd_rows_kernel(const void* __restrict__ a, int aC, int a0, void* b, int bC, int b0, int h, int w, int c, int cv_shift,
                                                                    const void* __restrict__ act, int actC, int act0, void* b2, int b2C, int b20, float slope) {
  constexpr int N = VecN<T>::N;
  const int cv = c / N;
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (unsigned)(w * cv)) return;
  const int kx = cv_shift >= 0 ? (int)(i >> cv_shift) : (int)(i / (unsigned)cv), ch = ((int)i - kx * cv) * N;
  const int ky0 = blockIdx.y * R;
  const size_t img = blockIdx.z;
  int dxs[4]; float wxs[4]; bool onx[4];
  bil_bwd_taps4(kx, w, dxs, wxs, onx);
  u32x4 win[4][4];                                          // high-res row 2*ky0 - 1 + m lives in slot m % 4, as loaded (16 bytes per tap)
  auto load_row = [&](u32x4* dst, int d) {
    if (d < 0 || d >= 2 * h) return;                        // wave-uniform
    const size_t row = (img * 2 * h + d) * (size_t)(2 * w);
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
      if (onx[ib]) dst[ib] = ldraw<T>(a, (row + dxs[ib]) * aC + a0 + ch);
  };
  load_row(win[0], 2 * ky0 - 1);
  load_row(win[1], 2 * ky0);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ky = ky0 + r;
    if (ky >= h) break;                                     // wave-uniform
    load_row(win[(2 * r + 2) % 4], 2 * ky + 1);
    load_row(win[(2 * r + 3) % 4], 2 * ky + 2);
    int dys[4]; float wys[4]; bool ony[4];
    bil_bwd_taps4(ky, h, dys, wys, ony);
    float acc[N], t[N];
#pragma unroll
    for (int q = 0; q < N; ++q) acc[q] = 0.f;
#pragma unroll
    for (int ia = 0; ia < 4; ++ia) {
      if (!ony[ia]) continue;                               // wave-uniform
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) {
        if (onx[ib]) {
          widen<T>(win[(2 * r + ia) % 4][ib], t);
          const float ww = wys[ia] * wxs[ib];
#pragma unroll
          for (int q = 0; q < N; ++q) acc[q] += ww * t[q];
        }
      }
    }
    const size_t op_ = (img * h + ky) * (size_t)w + kx;
    if (b) { if constexpr (NT) stv_nt<T>(b, op_ * (size_t)bC + b0 + ch, acc); else stv<T>(b, op_ * (size_t)bC + b0 + ch, acc); }
    if (act) {
      ldv<T>(act, op_ * (size_t)actC + act0 + ch, t);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] *= t[q] > 0.f ? 1.f : slope;
      if constexpr (NT) stv_nt<T>(b2, op_ * (size_t)b2C + b20 + ch, acc); else stv<T>(b2, op_ * (size_t)b2C + b20 + ch, acc);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void lrelu_bwd_vec_kernel(const void* __restrict__ dy, int dC, int d0, const void* __restrict__ act, int aC, int a0,
                                                            const void* __restrict__ skip, int sC, int s0, void* out, int oC, int o0, size_t npix, int c, float slope) {
  constexpr int N = VecN<T>::N;
  const int cv = c / N;
  const size_t total = npix * cv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % cv) * N;
    const size_t p = i / cv;
    float g[N], z[N], sk[N];
    ldv<T>(dy, p * dC + d0 + ch, g);
    ldv<T>(act, p * aC + a0 + ch, z);
    if (skip) {
      ldv<T>(skip, p * sC + s0 + ch, sk);
#pragma unroll
      for (int q = 0; q < N; ++q) z[q] -= sk[q];
    }
#pragma unroll
    for (int q = 0; q < N; ++q) g[q] *= z[q] > 0.f ? 1.f : slope;
    stv<T>(out, p * oC + o0 + ch, g);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void axpby_vec_kernel(const void* __restrict__ x, int xC, int x0, void* y, int yC, int y0, size_t npix, int c, float a, float b) {
  constexpr int N = VecN<T>::N;
  const int cv = c / N;
  const size_t total = npix * cv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % cv) * N;
    const size_t p = i / cv;
    float vx[N], vy[N];
    ldv<T>(x, p * xC + x0 + ch, vx);
    if (b != 0.f) ldv<T>(y, p * yC + y0 + ch, vy);
#pragma unroll
    for (int q = 0; q < N; ++q) vx[q] = a * vx[q] + (b != 0.f ? b * vy[q] : 0.f);
    stv<T>(y, p * yC + y0 + ch, vx);
  }
}

// ---- NCHW fp32 -> NHWC T view, zero padded to cpad channels (BSRGAN.forward input, model.py:366) ----
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, void* dst, int dC, int d0, int n, int c, int hw, int cpad,
                                    const float* __restrict__ mean, const float* __restrict__ stdv) {
  const size_t total = (size_t)n * hw * cpad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpad);
    const size_t p = i / cpad;
    const size_t img = p / hw, pix = p % hw;
    float v = 0.f;
    if (ch < c) {
      v = src[(img * c + ch) * hw + pix];
      if (mean) v = (v - mean[ch]) / stdv[ch];
    }
    st<T>(dst, p * dC + d0 + ch, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_vec_kernel(const float* __restrict__ src, void* dst, int dC, int d0, int n, int c, int hw, int cpad,
                                                               const float* __restrict__ mean, const float* __restrict__ stdv) {
  constexpr int N = VecN<T>::N;
  const int cv = cpad / N;
  const size_t total = (size_t)n * hw * cv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ck = (int)(i % cv) * N;
    const size_t p = i / cv;
    const size_t img = p / hw, pix = p % hw;
    float v[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const int ch = ck + q;
      float t = 0.f;
      if (ch < c) {
        t = src[(img * c + ch) * hw + pix];
        if (mean) t = (t - mean[ch]) / stdv[ch];
      }
      v[q] = t;
    }
    stv<T>(dst, p * dC + d0 + ck, v);
  }
}

// ---- NHWC view (T or fp32) -> NCHW fp32, optional clamp to [0,1] (model.py:379) ----
template <typename T>
__global__ void nhwc_to_nchw_kernel(const void* __restrict__ src, int sC, int s0, float* __restrict__ dst, int n, int c, int hw, int clamp01) {
  const size_t total = (size_t)n * c * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i % hw;
    const size_t t = i / hw;
    const int ch = (int)(t % c);
    const size_t img = t / c;
    float v = ld<T>(src, (img * hw + pix) * sC + s0 + ch);
    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.0f);
    dst[i] = v;
  }
}

// ---- gradient of clamp_(0,1) + NCHW fp32 -> NHWC T (zero padded): d pre = (0 <= pre <= 1) ? d sr : 0 ----
template <typename T>
__global__ void clamp_grad_kernel(const float* __restrict__ dsr, const float* __restrict__ pre, int pC, int p0, void* dst, int dC, int d0,
                                  int n, int c, int hw, int cpad) {
  const size_t total = (size_t)n * hw * cpad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpad);
    const size_t p = i / cpad;
    const size_t img = p / hw, pix = p % hw;
    float v = 0.f;
    if (ch < c) {
      const float q = pre[p * pC + p0 + ch];
      if (q >= 0.f && q <= 1.f) v = dsr[(img * c + ch) * hw + pix];
    }
    st<T>(dst, p * dC + d0 + ch, v);
  }
}

// the generator's case (3 image channels, fp32 pre-clamp SR with a 4-channel pitch, 16-bit gradient padded to 32 channels): one thread
// per pixel, one 16-byte read of the pre-clamp pixel, three coalesced plane reads, four 16-byte stores (the 29 padding channels are
// zeros the data-gradient conv multiplies by padded weights)
template <typename T>
__global__ __launch_bounds__(256) void clamp_grad_rgb16_kernel(const float* __restrict__ dsr, const f32x4* __restrict__ pre, u32x4* __restrict__ dst,
                                                               size_t npix, size_t hw, int c) {
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    const size_t img = p / hw, pix = p % hw;
    const f32x4 q = pre[p];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < c && q[k] >= 0.f && q[k] <= 1.f) v[k] = dsr[(img * c + k) * hw + pix];
    const float v8[8] = {v[0], v[1], v[2], v[3], 0.f, 0.f, 0.f, 0.f};
    const u32x4 w0 = pack8<T>(v8);
    const u32x4 z = {0u, 0u, 0u, 0u};
    dst[p * 4 + 0] = w0; dst[p * 4 + 1] = z; dst[p * 4 + 2] = z; dst[p * 4 + 3] = z;
  }
}

// the same into a 4-channel pitch ("NHWC4", 8 bytes per pixel): the thin-side kernels' operand (conv_thin.hip)
template <typename T>
__global__ __launch_bounds__(256) void clamp_grad_rgb4_kernel(const float* __restrict__ dsr, const f32x4* __restrict__ pre, unsigned long long* __restrict__ dst,
                                                              size_t npix, size_t hw, int c) {
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    const size_t img = p / hw, pix = p % hw;
    const f32x4 q = pre[p];
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < c && q[k] >= 0.f && q[k] <= 1.f) v[k] = dsr[(img * c + k) * hw + pix];
    const u32x4 w0 = pack8<T>(v);
    dst[p] = (unsigned long long)w0[0] | ((unsigned long long)w0[1] << 32);
  }
}

// NCHW fp32 (c <= 4 planes) -> NHWC4 16-bit: one thread per pixel, c coalesced plane reads, one 8-byte store
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ src, unsigned long long* __restrict__ dst, size_t npix, size_t hw, int c,
                                                            const float* __restrict__ mean, const float* __restrict__ stdv) {
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    const size_t img = p / hw, pix = p % hw;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < c) {
        float t = src[(img * c + k) * hw + pix];
        if (mean) t = (t - mean[k]) / stdv[k];
        v[k] = t;
      }
    const u32x4 w0 = pack8<T>(v);
    dst[p] = (unsigned long long)w0[0] | ((unsigned long long)w0[1] << 32);
  }
}

// ---- backward of F.interpolate(scale_factor=2, mode="nearest") (model.py:372,374): 2x2 sum ----
template <typename T>
__global__ void up2_nearest_bwd_kernel(const void* __restrict__ dy, int yC, int y0, void* dx, int xC, int x0, int n, int h, int w, int c) {
  const size_t total = (size_t)n * h * w * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    size_t p = i / c;
    const int x = (int)(p % w); p /= w;
    const int y = (int)(p % h);
    const size_t img = p / h;
    const size_t b = ((img * 2 * h + 2 * y) * 2 * w + 2 * x);
    const float s = ld<T>(dy, b * yC + y0 + ch) + ld<T>(dy, (b + 1) * yC + y0 + ch) + ld<T>(dy, (b + 2 * w) * yC + y0 + ch) +
                    ld<T>(dy, (b + 2 * w + 1) * yC + y0 + ch);
    st<T>(dx, (i / c) * xC + x0 + ch, s);
  }
}

// ---- bilinear x2, align_corners=False (model.py:150,154,158) forward and backward ----
// dst(2k)   = 0.25*src(k-1) + 0.75*src(k)   (src index clamped to [0, n-1])
// dst(2k+1) = 0.75*src(k)   + 0.25*src(k+1)
__device__ __forceinline__ void bil_taps(int d, int n, int& i0, int& i1, float& w0, float& w1) {
  const int k = d >> 1;
  if (d & 1) { i0 = k; i1 = min(k + 1, n - 1); w0 = 0.75f; w1 = 0.25f; }
  else { i0 = max(k - 1, 0); i1 = k; w0 = 0.25f; w1 = 0.75f; }
}
template <typename T>
__global__ void up2_bilinear_fwd_kernel(const void* __restrict__ x, int xC, int x0, void* y, int yC, int y0, int n, int h, int w, int c) {
  const size_t total = (size_t)n * 4 * h * w * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    size_t p = i / c;
    const int ox = (int)(p % (2 * w)); p /= (2 * w);
    const int oy = (int)(p % (2 * h));
    const size_t img = p / (2 * h);
    int ya, yb, xa, xb; float wya, wyb, wxa, wxb;
    bil_taps(oy, h, ya, yb, wya, wyb);
    bil_taps(ox, w, xa, xb, wxa, wxb);
    const size_t r0 = (img * h + ya) * w, r1 = (img * h + yb) * w;
    const float v = wya * (wxa * ld<T>(x, (r0 + xa) * xC + x0 + ch) + wxb * ld<T>(x, (r0 + xb) * xC + x0 + ch)) +
                    wyb * (wxa * ld<T>(x, (r1 + xa) * xC + x0 + ch) + wxb * ld<T>(x, (r1 + xb) * xC + x0 + ch));
    st<T>(y, (i / c) * yC + y0 + ch, v);
  }
}
// gather form of the transpose: src pixel k receives from dst 2k-1 (0.25), 2k (0.75), 2k+1 (0.75), 2k+2 (0.25),
// plus the clamped border contributions (dst 0 -> src 0 with the 0.25 that would go to src -1; same at the top).
__device__ __forceinline__ int bil_bwd_taps(int k, int n, int* d, float* wt) {
  int cnt = 0;
  for (int dd = 2 * k - 2; dd <= 2 * k + 3; ++dd) {
    if (dd < 0 || dd >= 2 * n) continue;
    int i0, i1; float w0, w1;
    bil_taps(dd, n, i0, i1, w0, w1);
    float ww = 0.f;
    if (i0 == k) ww += w0;
    if (i1 == k) ww += w1;
    if (ww != 0.f) { d[cnt] = dd; wt[cnt] = ww; ++cnt; }
  }
  return cnt;
}
template <typename T>
__global__ void up2_bilinear_bwd_kernel(const void* __restrict__ dy, int yC, int y0, void* dx, int xC, int x0, int n, int h, int w, int c) {
  const size_t total = (size_t)n * h * w * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    size_t p = i / c;
    const int x = (int)(p % w); p /= w;
    const int y = (int)(p % h);
    const size_t img = p / h;
    int dys[6], dxs[6]; float wys[6], wxs[6];
    const int ny = bil_bwd_taps(y, h, dys, wys), nx = bil_bwd_taps(x, w, dxs, wxs);
    float s = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b)
        s += wys[a] * wxs[b] * ld<T>(dy, ((img * 2 * h + dys[a]) * 2 * w + dxs[b]) * yC + y0 + ch);
    st<T>(dx, (i / c) * xC + x0 + ch, s);
  }
}

// ---- y = a*x + b*y on channel-slice views ----
template <typename T>
__global__ void axpby_kernel(const void* __restrict__ x, int xC, int x0, void* y, int yC, int y0, size_t npix, int c, float a, float b) {
  const size_t total = npix * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    const size_t p = i / c;
    const float v = a * ld<T>(x, p * xC + x0 + ch) + (b != 0.f ? b * ld<T>(y, p * yC + y0 + ch) : 0.f);
    st<T>(y, p * yC + y0 + ch, v);
  }
}

// ---- 2x2 max pool (+ the preceding ReLU is already in the conv epilogue) for VGG-19 features ----
template <typename T>
__global__ void maxpool2_kernel(const void* __restrict__ x, int xC, int x0, void* y, int yC, int y0, int n, int h, int w, int c) {
  const int ho = h / 2, wo = w / 2;
  const size_t total = (size_t)n * ho * wo * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    size_t p = i / c;
    const int ox = (int)(p % wo); p /= wo;
    const int oy = (int)(p % ho);
    const size_t img = p / ho;
    const size_t b = (img * h + 2 * oy) * w + 2 * ox;
    const float v = fmaxf(fmaxf(ld<T>(x, b * xC + x0 + ch), ld<T>(x, (b + 1) * xC + x0 + ch)),
                          fmaxf(ld<T>(x, (b + w) * xC + x0 + ch), ld<T>(x, (b + w + 1) * xC + x0 + ch)));
    st<T>(y, (i / c) * yC + y0 + ch, v);
  }
}

// ---- ReLU copy (VGG taps observed pre-ReLU) and LeakyReLU backward with the sign recovered from act - skip ----
template <typename T>
__global__ void relu_copy_kernel(const void* __restrict__ x, int xC, int x0, void* y, int yC, int y0, int n, int h, int w, int c) {
  const size_t total = (size_t)n * h * w * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    const size_t p = i / c;
    st<T>(y, p * yC + y0 + ch, fmaxf(ld<T>(x, p * xC + x0 + ch), 0.f));
  }
}
template <typename T>
__global__ void lrelu_bwd_kernel(const void* __restrict__ dy, int dC, int d0, const void* __restrict__ act, int aC, int a0,
                                 const void* __restrict__ skip, int sC, int s0, void* out, int oC, int o0, size_t npix, int c, float slope) {
  const size_t total = npix * c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    const size_t p = i / c;
    float z = ld<T>(act, p * aC + a0 + ch);
    if (skip) z -= ld<T>(skip, p * sC + s0 + ch);
    st<T>(out, p * oC + o0 + ch, ld<T>(dy, p * dC + d0 + ch) * (z > 0.f ? 1.f : slope));
  }
}

// ---- losses.  out[slot] (+)= weight * mean(...) ; two-stage deterministic reduction ----
// L1 (nn.L1Loss, train_bsrgan.py:297,450) on flat fp32 arrays, optional gradient wrt a.
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, float gscale,
                                                         const float* __restrict__ gscale_dev, float* __restrict__ grad, float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  if (gscale_dev) gscale *= *gscale_dev;      // the loss scale lives in device memory (srganfd_loss_scale_update), as torch's GradScaler keeps it
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float d = a[i] - b[i];
    s += fabsf(d);
    if (grad) grad[i] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
  }
  const float r = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
// L1 between two NHWC T views (VGG feature taps, model.py:548-550), no gradient (detached in the reference)
template <typename T>
__global__ __launch_bounds__(256) void l1_views_partial_kernel(const void* __restrict__ a, int aC, int a0, const void* __restrict__ b, int bC, int b0,
                                                               size_t npix, int c, int relu, float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  const size_t total = npix * c;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    const size_t p = i / c;
    float va = ld<T>(a, p * aC + a0 + ch), vb = ld<T>(b, p * bC + b0 + ch);
    if (relu) { va = fmaxf(va, 0.f); vb = fmaxf(vb, 0.f); }
    s += fabsf(va - vb);
  }
  const float r = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
template <typename T>
__global__ __launch_bounds__(256) void l1_views_vec_partial_kernel(const void* __restrict__ a, int aC, int a0, const void* __restrict__ b, int bC, int b0,
                                                                   size_t npix, int c, int relu, float* __restrict__ partial) {
  constexpr int N = VecN<T>::N;
  __shared__ float sh[4];
  float s = 0.f;
  const int cv = c / N;
  const size_t total = npix * cv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % cv) * N;
    const size_t p = i / cv;
    float va[N], vb[N];
    ldv<T>(a, p * aC + a0 + ch, va);
    ldv<T>(b, p * bC + b0 + ch, vb);
#pragma unroll
    for (int q = 0; q < N; ++q) {
      float x = va[q], y = vb[q];
      if (relu) { x = fmaxf(x, 0.f); y = fmaxf(y, 0.f); }
      s += fabsf(x - y);
    }
  }
  const float r = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
// BCE-with-logits against a constant label map (train_bsrgan.py:301,403-404): loss and sigmoid mean
__global__ __launch_bounds__(256) void bce_partial_kernel(const float* __restrict__ x, size_t n, float target, float gscale,
                                                          const float* __restrict__ gscale_dev, float* __restrict__ grad, float* __restrict__ partial,
                                                          float* __restrict__ partial_sig) {
  __shared__ float sh[4];
  float s = 0.f, sg = 0.f;
  if (gscale_dev) gscale *= *gscale_dev;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = x[i];
    s += fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v)));
    const float sig = 1.f / (1.f + expf(-v));
    sg += sig;
    if (grad) grad[i] = (sig - target) * gscale;
  }
  float r = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
  r = block_reduce_sum(sg, sh);
  if (threadIdx.x == 0) partial_sig[blockIdx.x] = r;
}
__global__ __launch_bounds__(256) void finish_sum_kernel(const float* __restrict__ partial, int nblk, float scale, float* out, int accumulate) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
  const float r = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) *out = (accumulate ? *out : 0.f) + r * scale;
}
// sigmoid(mean(logits)): the D(x) probability as ESRGAN / Real-ESRGAN log it (train_esrgan.py:430-431, train_realesrgan.py:475-476;
// BSRGAN / A-ESRGAN log mean(sigmoid(logits)) instead -- bce_partial_kernel's second output)
__global__ __launch_bounds__(256) void sum_partial_kernel(const float* __restrict__ x, size_t n, float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += x[i];
  const float r = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
// ---- relativistic-average BCE (ESRGAN/train_esrgan.py:378-380,404,412): mean_i BCE(x_i - mean(other), target) ----
// stage 0: mean(other) -> ws[2 * kRedBlocks] (sum_partial_kernel + this finish); stage 1: per-element loss, d/dx_i, partial sums of the loss
// and of (sigmoid - target); stage 2: finishes -- loss, and d/d(other_j) = -(1/n_other) * mean_i(sigmoid_i - target), the same for every j.
__global__ __launch_bounds__(256) void finish_mean_kernel(const float* __restrict__ partial, int nblk, float inv_n, float* out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
  const float r = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) *out = r * inv_n;
}
__global__ __launch_bounds__(256) void bce_rel_partial_kernel(const float* __restrict__ x, size_t n, const float* __restrict__ other_mean, float target,
                                                              float gscale, const float* __restrict__ gscale_dev, float* __restrict__ grad_x,
                                                              int accumulate_x, float* __restrict__ partial, float* __restrict__ partial_d) {
  __shared__ float sh[4];
  float s = 0.f, sd = 0.f;
  const float m = *other_mean;
  if (gscale_dev) gscale *= *gscale_dev;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = x[i] - m;
    s += fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v)));
    const float d = 1.f / (1.f + expf(-v)) - target;
    sd += d;
    if (grad_x) grad_x[i] = (accumulate_x ? grad_x[i] : 0.f) + d * gscale;
  }
  float r = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
  r = block_reduce_sum(sd, sh);
  if (threadIdx.x == 0) partial_d[blockIdx.x] = r;
}
// grad_other[j] (+)= -gscale * sum_d / n_other for every j (gscale already carries weight / n_x)
__global__ __launch_bounds__(256) void bce_rel_other_kernel(const float* __restrict__ partial_d, int nblk, float gscale, const float* __restrict__ gscale_dev,
                                                            float* __restrict__ grad_other, size_t n_other, int accumulate) {
  __shared__ float sh[4];
  __shared__ float tot;
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial_d[i];      // every block re-reduces the (<= 1024) partials: same order, same value
  const float r = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) tot = r;
  __syncthreads();
  if (gscale_dev) gscale *= *gscale_dev;
  const float g = -gscale * tot / (float)n_other;
  for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n_other; j += (size_t)gridDim.x * 256)
    grad_other[j] = (accumulate ? grad_other[j] : 0.f) + g;
}
__global__ __launch_bounds__(256) void finish_sigmoid_mean_kernel(const float* __restrict__ partial, int nblk, float inv_n, float* out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
  const float r = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) *out = 1.f / (1.f + expf(-r * inv_n));
}

// ---- spectral norm (torch/nn/utils/spectral_norm.py:62-114 as applied at model.py:104-132) ----
// W is (rows=Cout, cols=Cin*k*k) row-major fp32.
// W^T u in row chunks of kSnRows: block (x, y) sums rows [y*kSnRows, ...) of 256 columns into part[y][k]; the normalise kernel adds the
// chunks in order (deterministic).  One thread per column over ALL rows left the chip with <= 18 workgroups for 71 us per layer.
static constexpr int kSnRows = 32;
// Up to kSnBatch layers per launch (blockIdx.z / .y picks the layer): eight layers x four dependent 5-14 us kernels are launch latency,
// not work.  Every layer is summed exactly as in a launch of its own, so batching does not change a bit.
static constexpr int kSnBatch = SRGANFD_SN_BATCH;
struct SnJobs { srganfd_sn_job j[kSnBatch]; };
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const SnJobs jobs) {
  const srganfd_sn_job& J = jobs.j[blockIdx.z];
  const int rows = J.rows, cols = J.cols;
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * kSnRows, r1 = min(rows, r0 + kSnRows);
  if (k >= cols || r0 >= rows) return;
  const float* __restrict__ W = J.w_orig; const float* __restrict__ u = J.u;
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += W[(size_t)r * cols + k] * u[r];
  J.workspace[(size_t)blockIdx.y * cols + k] = s;
}
__global__ __launch_bounds__(1024) void sn_normalize_kernel(const SnJobs jobs, float eps) {
  __shared__ float sh[16];
  __shared__ float inv;
  const srganfd_sn_job& J = jobs.j[blockIdx.x];
  const int n = J.cols, nparts = (J.rows + kSnRows - 1) / kSnRows;
  const float* __restrict__ part = J.workspace; float* __restrict__ out = J.v;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    float v = 0.f;
    for (int p = 0; p < nparts; ++p) v += part[(size_t)p * n + i];
    out[i] = v;                      // raw W^T u, scaled in place below
    s += v * v;
  }
  const float r = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) inv = 1.f / fmaxf(sqrtf(r), eps);
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 1024) out[i] *= inv;
}
__device__ __forceinline__ float* sn_t(const srganfd_sn_job& J) { return J.workspace + (size_t)((J.rows + kSnRows - 1) / kSnRows) * J.cols; }
__global__ __launch_bounds__(256) void sn_w_v_kernel(const SnJobs jobs) {
  __shared__ float sh[4];
  const srganfd_sn_job& J = jobs.j[blockIdx.y];
  const int r = blockIdx.x, cols = J.cols;
  if (r >= J.rows) return;
  const float* __restrict__ W = J.w_orig; const float* __restrict__ v = J.v;
  float s = 0.f;
  for (int k = threadIdx.x; k < cols; k += 256) s += W[(size_t)r * cols + k] * v[k];
  const float tot = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) sn_t(J)[r] = tot;
}
// u = normalize(t) (only if update_u), sigma = u . t, inv_sigma = 1/sigma
__global__ __launch_bounds__(256) void sn_finish_kernel(const SnJobs jobs, float eps, int update_u) {
  __shared__ float sh[4];
  __shared__ float inv;
  const srganfd_sn_job& J = jobs.j[blockIdx.x];
  const int rows = J.rows;
  const float* __restrict__ t = sn_t(J); float* __restrict__ u = J.u;
  float s = 0.f;
  if (update_u) {
    for (int i = threadIdx.x; i < rows; i += 256) s += t[i] * t[i];
    const float r = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) inv = 1.f / fmaxf(sqrtf(r), eps);
    __syncthreads();
    for (int i = threadIdx.x; i < rows; i += 256) u[i] = t[i] * inv;
    __syncthreads();
  }
  s = 0.f;
  for (int i = threadIdx.x; i < rows; i += 256) s += u[i] * t[i];
  const float sig = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) { *J.sigma_out = sig; *J.inv_sigma_out = 1.f / sig; }
}
// gradient through weight = W_orig / sigma, sigma = u^T W_orig v (u, v constants):
//   dW_orig = (G - <G, W_orig>/sigma * u v^T) / sigma        with G = dL/d(weight)
// Batched like the forward kernels: blockIdx.y picks the layer; every layer keeps the grid (number of partial sums, element stride) a
// launch of its own would have, so the sums are bit-identical.  kSnGradBlocks = the loss entry points' kRedBlocks.
static constexpr int kSnGradBlocks = 1024;
struct SnGradJobs { srganfd_sn_grad_job j[kSnBatch]; };
__device__ __forceinline__ unsigned sn_grad_blocks(size_t n) { const size_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > kSnGradBlocks ? kSnGradBlocks : g)); }
__global__ __launch_bounds__(256) void sn_dot_partial_kernel(const SnGradJobs jobs) {
  __shared__ float sh[4];
  const srganfd_sn_grad_job& J = jobs.j[blockIdx.y];
  const size_t n = (size_t)J.rows * J.cols;
  const unsigned g = sn_grad_blocks(n);
  if (blockIdx.x >= g) return;
  const float* __restrict__ G = J.g_weight; const float* __restrict__ W = J.w_orig;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)g * 256) s += G[i] * W[i];
  const float r = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) J.workspace[blockIdx.x] = r;
}
__global__ __launch_bounds__(256) void sn_dot_finish_kernel(const SnGradJobs jobs) {
  __shared__ float sh[4];
  const srganfd_sn_grad_job& J = jobs.j[blockIdx.x];
  const int nblk = (int)sn_grad_blocks((size_t)J.rows * J.cols);
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += J.workspace[i];
  const float r = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) J.workspace[kSnGradBlocks] = 0.f + r * 1.f;
}
__global__ __launch_bounds__(256) void sn_grad_kernel(const SnGradJobs jobs, float beta) {
  const srganfd_sn_grad_job& J = jobs.j[blockIdx.y];
  const int cols = J.cols;
  const size_t n = (size_t)J.rows * cols;
  const float* __restrict__ G = J.g_weight; const float* __restrict__ u = J.u; const float* __restrict__ v = J.v;
  float* __restrict__ dW = J.dw_orig;
  const float is = *J.inv_sigma, coef = J.workspace[kSnGradBlocks] * is;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / cols), k = (int)(i % cols);
    const float g = (G[i] - coef * u[r] * v[k]) * is;
    dW[i] = g + (beta != 0.f ? beta * dW[i] : 0.f);
  }
}

// ---- fused Adam (torch.optim.Adam maths, train_bsrgan.py:311-323) + EMA (train_bsrgan.py:290-291,470) ----
__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                       float* __restrict__ ema, size_t n, float lr, float b1, float b2, float eps, float wd,
                                                       float bc1, float bc2_sqrt, float gscale, float ema_decay, int ema_mode,
                                                       const float* __restrict__ skip, const float* __restrict__ gscale_dev) {
  if (gscale_dev) gscale *= *gscale_dev;      // 1 / loss scale, from the device-resident scaler state
  // loss-scaled (f16) training: a non-finite gradient skips the parameter update (GradScaler.step, train_bsrgan.py:436,466); the
  // EMA still advances -- the reference calls update_parameters() after every iteration (:470)
  const bool skipped = skip && *skip != 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (skipped) {
      if (ema_mode == 1) ema[i] = p[i];
      else if (ema_mode == 2) ema[i] = (1.f - ema_decay) * ema[i] + ema_decay * p[i];
      continue;
    }
    float gi = g[i] * gscale;
    float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
    if (ema_mode == 1) ema[i] = pi;                                               // first update: copy
    else if (ema_mode == 2) ema[i] = (1.f - ema_decay) * ema[i] + ema_decay * pi; // reference avg_fn
  }
}

// Step counter and bias corrections in device memory (hipGraph replays cannot change kernel arguments): one thread
// advances *step and writes bc = {1 - b1^t, sqrt(1 - b2^t)}; the Adam kernel then reads them.
__global__ void adam_step_kernel(int* __restrict__ step, float b1, float b2, float* __restrict__ bc, const float* __restrict__ skip) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !(skip && *skip != 0.f)) {      // a skipped step does not count (torch: state["step"] unchanged)
    const int t = *step + 1;
    *step = t;
    bc[0] = (float)(1.0 - pow((double)b1, (double)t));
    bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
  }
}
__global__ __launch_bounds__(256) void adam_ema_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                           float* __restrict__ ema, size_t n, float lr, float b1, float b2, float eps, float wd,
                                                           const float* __restrict__ bc, float gscale, float ema_decay, int ema_mode,
                                                           const float* __restrict__ skip, const float* __restrict__ gscale_dev) {
  const float bc1 = bc[0], bc2_sqrt = bc[1];
  if (gscale_dev) gscale *= *gscale_dev;
  const bool skipped = skip && *skip != 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (skipped) {
      if (ema_mode == 1) ema[i] = p[i];
      else if (ema_mode == 2) ema[i] = (1.f - ema_decay) * ema[i] + ema_decay * p[i];
      continue;
    }
    float gi = g[i] * gscale;
    float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
    if (ema_mode == 1) ema[i] = pi;
    else if (ema_mode == 2) ema[i] = (1.f - ema_decay) * ema[i] + ema_decay * pi;
  }
}

// ---- A-ESRGAN attention gates (A-ESRGAN/model.py:239-254): general bilinear resize, relu(a+b), sigmoid,
// gate multiply, BatchNorm2d (training statistics, running stats, backward) ----
// F.interpolate(mode="bilinear", align_corners=False) with an explicit output size (ATen area_pixel source index)
__device__ __forceinline__ void resize_taps(int d, int in, float scale, int& i0, int& i1, float& w0, float& w1) {
  float src = scale * ((float)d + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  w1 = src - (float)i0;
  w0 = 1.f - w1;
}
template <typename T>
__global__ __launch_bounds__(256) void resize_fwd_kernel(const void* __restrict__ a, int aC, int a0, void* b, int bC, int b0, int n, int hi, int wi,
                                                         int ho, int wo, int c) {
  constexpr int N = VecN<T>::N;
  const int cv = c / N;
  const float sy = (float)hi / (float)ho, sx = (float)wi / (float)wo;
  const size_t total = (size_t)n * ho * wo * cv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % cv) * N;
    size_t p = i / cv;
    const int ox = (int)(p % wo); p /= wo;
    const int oy = (int)(p % ho);
    const size_t img = p / ho;
    int ya, yb, xa, xb; float wya, wyb, wxa, wxb;
    resize_taps(oy, hi, sy, ya, yb, wya, wyb);
    resize_taps(ox, wi, sx, xa, xb, wxa, wxb);
    float acc[N], t[N];
    ldv<T>(a, ((img * hi + ya) * wi + xa) * (size_t)aC + a0 + ch, t);
#pragma unroll
    for (int q = 0; q < N; ++q) acc[q] = wya * wxa * t[q];
    ldv<T>(a, ((img * hi + ya) * wi + xb) * (size_t)aC + a0 + ch, t);
#pragma unroll
    for (int q = 0; q < N; ++q) acc[q] += wya * wxb * t[q];
    ldv<T>(a, ((img * hi + yb) * wi + xa) * (size_t)aC + a0 + ch, t);
#pragma unroll
    for (int q = 0; q < N; ++q) acc[q] += wyb * wxa * t[q];
    ldv<T>(a, ((img * hi + yb) * wi + xb) * (size_t)aC + a0 + ch, t);
#pragma unroll
    for (int q = 0; q < N; ++q) acc[q] += wyb * wxb * t[q];
    stv<T>(b, ((img * ho + oy) * wo + ox) * (size_t)bC + b0 + ch, acc);
  }
}
// backward as a deterministic gather: input pixel k collects every output pixel whose two taps include k
__device__ __forceinline__ int resize_bwd_range(int k, int in, int out, float scale, int& lo) {
  // outputs d with src(d) in (k-1, k+1): d in ((k-0.5)/scale - 0.5 - 1, (k+1.5)/scale - 0.5 + 1)
  int l = (int)floorf(((float)k - 0.5f) / scale - 0.5f) - 1, h = (int)ceilf(((float)k + 1.5f) / scale - 0.5f) + 1;
  if (k == 0) l = 0;   // clamped sources
  if (l < 0) l = 0;
  if (h > out - 1) h = out - 1;
  lo = l;
  return h;
}
template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const void* __restrict__ dy, int yC, int y0, void* dx, int xC, int x0, int n, int hi, int wi,
                                                         int ho, int wo, int c) {
  constexpr int N = VecN<T>::N;
  const int cv = c / N;
  const float sy = (float)hi / (float)ho, sx = (float)wi / (float)wo;
  const size_t total = (size_t)n * hi * wi * cv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % cv) * N;
    size_t p = i / cv;
    const int x = (int)(p % wi); p /= wi;
    const int y = (int)(p % hi);
    const size_t img = p / hi;
    int ylo, xlo;
    const int yhi = resize_bwd_range(y, hi, ho, sy, ylo), xhi = resize_bwd_range(x, wi, wo, sx, xlo);
    float acc[N], t[N];
#pragma unroll
    for (int q = 0; q < N; ++q) acc[q] = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      int ya, yb; float wya, wyb;
      resize_taps(oy, hi, sy, ya, yb, wya, wyb);
      const float wy = (ya == y ? wya : 0.f) + (yb == y ? wyb : 0.f);
      if (wy == 0.f) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        int xa, xb; float wxa, wxb;
        resize_taps(ox, wi, sx, xa, xb, wxa, wxb);
        const float wx = (xa == x ? wxa : 0.f) + (xb == x ? wxb : 0.f);
        if (wx == 0.f) continue;
        ldv<T>(dy, ((img * ho + oy) * wo + ox) * (size_t)yC + y0 + ch, t);
        const float ww = wy * wx;
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] += ww * t[q];
      }
    }
    stv<T>(dx, ((img * hi + y) * wi + x) * (size_t)xC + x0 + ch, acc);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void add_relu_kernel(const void* __restrict__ a, int aC, int a0, const void* __restrict__ b, int bC, int b0,
                                                       void* out, int oC, int o0, size_t npix, int c) {
  constexpr int N = VecN<T>::N;
  const int cv = c / N;
  const size_t total = npix * cv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % cv) * N;
    const size_t p = i / cv;
    float va[N], vb[N];
    ldv<T>(a, p * aC + a0 + ch, va);
    ldv<T>(b, p * bC + b0 + ch, vb);
#pragma unroll
    for (int q = 0; q < N; ++q) va[q] = fmaxf(va[q] + vb[q], 0.f);
    stv<T>(out, p * oC + o0 + ch, va);
  }
}
__global__ __launch_bounds__(256) void sigmoid_kernel(float* x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = 1.f / (1.f + expf(-x[i]));
}
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ s, float* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float v = s[i]; out[i] = ds[i] * v * (1.f - v); }
}
// y[p][c] = gate[p] * x[p][c]
template <typename T>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const void* __restrict__ x, int xC, int x0, const float* __restrict__ gate, void* y, int yC, int y0,
                                                       size_t npix, int c) {
  constexpr int N = VecN<T>::N;
  const int cv = c / N;
  const size_t total = npix * cv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % cv) * N;
    const size_t p = i / cv;
    float v[N];
    ldv<T>(x, p * xC + x0 + ch, v);
    const float gv = gate[p];
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] *= gv;
    stv<T>(y, p * yC + y0 + ch, v);
  }
}
// dx[p][c] = gate[p] * dy[p][c] ; dgate[p] = sum_c dy[p][c] * x[p][c]   (c/N lanes of a wave per pixel, c/N a power of two <= 64)
template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const void* __restrict__ x, int xC, int x0, const float* __restrict__ gate,
                                                       const void* __restrict__ dy, int dC, int d0, void* dx, int oC, int o0, float* __restrict__ dgate,
                                                       size_t npix, int c) {
  constexpr int N = VecN<T>::N;
  const int cv = c / N;                       // lanes per pixel
  const size_t total = npix * cv;
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t iters = (total + stride - 1) / stride;
  for (size_t it = 0; it < iters; ++it) {
    const size_t i = it * stride + (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = i < total;
    const int ch = ok ? (int)(i % cv) * N : 0;
    const size_t p = ok ? i / cv : 0;
    float vx[N], vd[N];
    float part = 0.f;
    if (ok) {
      ldv<T>(x, p * xC + x0 + ch, vx);
      ldv<T>(dy, p * dC + d0 + ch, vd);
      const float gv = gate[p];
#pragma unroll
      for (int q = 0; q < N; ++q) { part += vd[q] * vx[q]; vd[q] *= gv; }
      stv<T>(dx, p * oC + o0 + ch, vd);
    }
    for (int o = cv >> 1; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (ok && (i % cv) == 0) dgate[p] = part;
  }
}
// BatchNorm2d.  Statistics: 16-byte loads, thread = one channel chunk, pixels strided over the grid (coalesced);
// each block writes partial[block][2][C]; the finish kernel (one 1024-thread block) reduces them in a fixed order
// (deterministic), turns them into mean / invstd / (scale, shift) and updates the running statistics.
template <typename T>
__global__ __launch_bounds__(256) void bn_partial_kernel(const void* __restrict__ x, int xC, int x0, const void* __restrict__ g, int gC, int g0,
                                                         const float* __restrict__ save, size_t npix, int c, float* __restrict__ partial,
                                                         const void* __restrict__ act, int actC, int act0, float act_slope) {
  // act (optional, backward only): output of the LeakyReLU that followed the BatchNorm; dy is scaled by its derivative
  // forward statistics (g == nullptr): sum x, sum x^2.  backward (g = dy): sum dy, sum dy * xhat (xhat from save)
  constexpr int N = VecN<T>::N;
  __shared__ float sh[2][256 * N];
  const int cv = c / N;                                 // 16-byte chunks per pixel (host: 256 % cv == 0)
  const int lanes = 256 / cv;                           // pixels per block pass
  const int chunk = threadIdx.x % cv, pl = threadIdx.x / cv, ch = chunk * N;
  float s0[N], s1[N], mean[N], invstd[N];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    s0[q] = 0.f; s1[q] = 0.f;
    mean[q] = (g && save) ? save[ch + q] : 0.f;
    invstd[q] = (g && save) ? save[c + ch + q] : 1.f;
  }
  const size_t step = (size_t)gridDim.x * lanes;
  if (g) {
    for (size_t p = (size_t)blockIdx.x * lanes + pl; p < npix; p += step) {
      float xv[N], dv[N];
      ldv<T>(x, p * xC + x0 + ch, xv);
      ldv<T>(g, p * gC + g0 + ch, dv);
      if (act) {
        float av[N];
        ldv<T>(act, p * actC + act0 + ch, av);
#pragma unroll
        for (int q = 0; q < N; ++q) dv[q] *= av[q] > 0.f ? 1.f : act_slope;
      }
#pragma unroll
      for (int q = 0; q < N; ++q) { s0[q] += dv[q]; s1[q] += dv[q] * (xv[q] - mean[q]) * invstd[q]; }
    }
  } else {
    size_t p = (size_t)blockIdx.x * lanes + pl;
    for (; p + step < npix; p += 2 * step) {            // two loads in flight
      float xa[N], xb[N];
      ldv<T>(x, p * xC + x0 + ch, xa);
      ldv<T>(x, (p + step) * xC + x0 + ch, xb);
#pragma unroll
      for (int q = 0; q < N; ++q) { s0[q] += xa[q] + xb[q]; s1[q] += xa[q] * xa[q] + xb[q] * xb[q]; }
    }
    if (p < npix) {
      float xa[N];
      ldv<T>(x, p * xC + x0 + ch, xa);
#pragma unroll
      for (int q = 0; q < N; ++q) { s0[q] += xa[q]; s1[q] += xa[q] * xa[q]; }
    }
  }
#pragma unroll
  for (int q = 0; q < N; ++q) { sh[0][(pl * cv + chunk) * N + q] = s0[q]; sh[1][(pl * cv + chunk) * N + q] = s1[q]; }
  __syncthreads();
  if ((int)threadIdx.x < c) {
    float a0 = 0.f, a1 = 0.f;
    for (int l = 0; l < lanes; ++l) { a0 += sh[0][l * c + threadIdx.x]; a1 += sh[1][l * c + threadIdx.x]; }
    partial[((size_t)blockIdx.x * 2 + 0) * c + threadIdx.x] = a0;
    partial[((size_t)blockIdx.x * 2 + 1) * c + threadIdx.x] = a1;
  }
}
// sums partial[b][2][c] over b with all 1024 threads (fixed order), result in tot[2][256]
__device__ __forceinline__ void bn_reduce_partials(const float* __restrict__ partial, int nblk, int c, float (*tot)[256]) {
  __shared__ float sh[2][1024];
  const int ch = threadIdx.x % c, l = threadIdx.x / c, lanes = 1024 / c;
  float a0 = 0.f, a1 = 0.f;
  if (l < lanes) {
#pragma unroll 8
    for (int b = l; b < nblk; b += lanes) { a0 += partial[((size_t)b * 2 + 0) * c + ch]; a1 += partial[((size_t)b * 2 + 1) * c + ch]; }
  }
  sh[0][threadIdx.x] = a0; sh[1][threadIdx.x] = a1;
  __syncthreads();
  if ((int)threadIdx.x < c) {
    float t0 = 0.f, t1 = 0.f;
    for (int k = 0; k < lanes; ++k) { t0 += sh[0][k * c + threadIdx.x]; t1 += sh[1][k * c + threadIdx.x]; }
    tot[0][threadIdx.x] = t0; tot[1][threadIdx.x] = t1;
  }
  __syncthreads();
}
__global__ __launch_bounds__(1024) void bn_fwd_finish_kernel(const float* __restrict__ partial, int nblk, int c, float npix, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps,
                                                             int training, float* __restrict__ save) {
  __shared__ float tot[2][256];
  if (training) bn_reduce_partials(partial, nblk, c, tot);
  const int ch = threadIdx.x;
  if (ch >= c) return;
  float mean, var;
  if (training) {
    mean = tot[0][ch] / npix;
    var = fmaxf(tot[1][ch] / npix - mean * mean, 0.f);
    running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mean;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * var * (npix / (npix - 1.f));
  } else {
    mean = running_mean[ch]; var = running_var[ch];
  }
  const float invstd = rsqrtf(var + eps);
  const float sc = gamma[ch] * invstd;
  save[ch] = mean; save[c + ch] = invstd; save[2 * c + ch] = sc; save[3 * c + ch] = beta[ch] - mean * sc;
}
// dx = dy*A + x*B + C0 per channel; coefficient triple + parameter gradients from the partial sums
// partial_global (data-parallel SyncBN, else NULL): the same table summed over the ranks.  The parameter gradients are this rank's
// sums (the flat-gradient all-reduce adds the ranks later); the dx coefficients use the sums and the pixel count of the whole batch.
__global__ __launch_bounds__(1024) void bn_bwd_finish_kernel(const float* __restrict__ partial, int nblk, int c, float npix, const float* __restrict__ gamma,
                                                             const float* __restrict__ save, float* dgamma, float* dbeta, float acc, float* __restrict__ coef,
                                                             const float* __restrict__ partial_global) {
  __shared__ float tot[2][256];
  bn_reduce_partials(partial, nblk, c, tot);
  const int ch = threadIdx.x;
  float db = 0.f, dg = 0.f;
  if (ch < c) {
    db = tot[0][ch]; dg = tot[1][ch];
    dgamma[ch] = dg + (acc != 0.f ? acc * dgamma[ch] : 0.f);
    dbeta[ch] = db + (acc != 0.f ? acc * dbeta[ch] : 0.f);
  }
  if (partial_global) {
    __syncthreads();
    bn_reduce_partials(partial_global, nblk, c, tot);
    if (ch < c) { db = tot[0][ch]; dg = tot[1][ch]; }
  }
  if (ch >= c) return;
  const float mean = save[ch], invstd = save[c + ch], gi = gamma[ch] * invstd;
  coef[ch] = gi;                                              // A
  coef[c + ch] = -gi * invstd * dg / npix;                    // B
  coef[2 * c + ch] = gi * (-db / npix + mean * invstd * dg / npix);  // C0
}
// out = a*ca[c] + b*cb[c] + c0[c]  (b, cb optional): BatchNorm apply (forward: a=x, ca=scale, c0=shift) and backward.
// Thread = one fixed channel chunk (coefficients live in registers), pixels strided over the grid.
template <typename T>
__global__ __launch_bounds__(256) void chan_affine_kernel(const void* __restrict__ a, int aC, int a0, const void* __restrict__ b, int bC, int b0,
                                                          void* out, int oC, int o0, const float* __restrict__ ca, const float* __restrict__ cb,
                                                          const float* __restrict__ c0, size_t npix, int c, float post_slope,
                                                          const void* __restrict__ act, int actC, int act0, float act_slope) {
  // post_slope: LeakyReLU applied to the result (1 = none).  act (optional): `a` is scaled by LeakyReLU'(act) first.
  constexpr int N = VecN<T>::N;
  const int cv = c / N, lanes = 256 / cv;               // host: 256 % cv == 0
  const int ch = (threadIdx.x % cv) * N, pl = threadIdx.x / cv;
  float fa[N], fb[N], f0[N];
#pragma unroll
  for (int q = 0; q < N; ++q) { fa[q] = ca[ch + q]; fb[q] = b ? cb[ch + q] : 0.f; f0[q] = c0[ch + q]; }
  const size_t step = (size_t)gridDim.x * lanes;
  for (size_t p = (size_t)blockIdx.x * lanes + pl; p < npix; p += step) {
    float va[N], vb[N];
    ldv<T>(a, p * aC + a0 + ch, va);
    if (act) {
      ldv<T>(act, p * actC + act0 + ch, vb);
#pragma unroll
      for (int q = 0; q < N; ++q) va[q] *= vb[q] > 0.f ? 1.f : act_slope;
    }
    if (b) {
      ldv<T>(b, p * bC + b0 + ch, vb);
#pragma unroll
      for (int q = 0; q < N; ++q) va[q] = va[q] * fa[q] + vb[q] * fb[q] + f0[q];
    } else {
#pragma unroll
      for (int q = 0; q < N; ++q) va[q] = va[q] * fa[q] + f0[q];
    }
    if (post_slope != 1.f) {
#pragma unroll
      for (int q = 0; q < N; ++q) va[q] = va[q] > 0.f ? va[q] : va[q] * post_slope;
    }
    stv<T>(out, p * oC + o0 + ch, va);
  }
}

// ------------------------------------------------------------------------------------------------
// validation / data side
// random_crop (imgproc.py:846-886): one (top, left) for the whole batch -> ONE strided copy instead of B slice copies
__global__ __launch_bounds__(256) void crop_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int planes, int h, int w, int top,
                                                        int left, int ph, int pw) {
  const size_t total = (size_t)planes * ph * pw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % pw);
    const size_t t = i / pw;
    const int y = (int)(t % ph);
    const size_t pl = t / ph;
    dst[i] = src[(pl * h + top + y) * w + left + x];
  }
}
// PSNR (image_quality_assessment.py:361-395): border crop, optional BT.601 luma in fp32 exactly as rgb_to_ycbcr_torch
// (imgproc.py:757-767: matmul, + 16, / 255), then the squared error of the x255 values accumulated in fp64.
// grid (blocks_per_image, n): partial[img][block]; the finish kernel sums them in a fixed order.
__global__ __launch_bounds__(256) void psnr_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, int c, int h, int w, int cb,
                                                           int y_only, double* __restrict__ partial) {
  __shared__ double sh[256];
  const int img = blockIdx.y;
  const int hh = h - 2 * cb, ww = w - 2 * cb;
  const size_t plane = (size_t)h * w;
  const float* pa = a + (size_t)img * c * plane;
  const float* pb = b + (size_t)img * c * plane;
  double acc = 0.0;
  const size_t npix = (size_t)hh * ww;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
    const int y = (int)(i / ww) + cb, x = (int)(i % ww) + cb;
    const size_t o = (size_t)y * w + x;
    if (y_only) {
      // torch.matmul of a (.., 3) row with the (3, 1) weight: fp32 fused multiply-add chain in channel order
      float ya = pa[o] * 65.481f; ya = fmaf(pa[plane + o], 128.553f, ya); ya = fmaf(pa[2 * plane + o], 24.966f, ya); ya = (ya + 16.0f) / 255.f;
      float yb = pb[o] * 65.481f; yb = fmaf(pb[plane + o], 128.553f, yb); yb = fmaf(pb[2 * plane + o], 24.966f, yb); yb = (yb + 16.0f) / 255.f;
      const double d = (double)ya * 255.0 - (double)yb * 255.0;
      acc += d * d + 1e-8;
    } else {
      for (int k = 0; k < c; ++k) {
        const double d = (double)pa[k * plane + o] * 255.0 - (double)pb[k * plane + o] * 255.0;
        acc += d * d + 1e-8;
      }
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(size_t)img * gridDim.x + blockIdx.x] = sh[0];
}
__global__ void psnr_finish_kernel(const double* __restrict__ partial, int nblk, double count, double* __restrict__ out) {
  const int img = blockIdx.x;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int k = 0; k < nblk; ++k) s += partial[(size_t)img * nblk + k];
    out[img] = 10.0 * log10(255.0 * 255.0 / (s / count));
  }
}

// SSIM (image_quality_assessment.py:420-494): border crop, optional BT.601 luma in fp32 (imgproc.py:757-767), x255 in
// fp64, then the five window-filtered moments (valid padding, any ws x ws window handed over by the caller) and the
// SSIM map, all in fp64 like the reference; the map is averaged over every channel and pixel of an image.
// grid (tiles_x * tiles_y, channels, n): a 16x16 output tile per block, its (16+ws-1)^2 inputs staged once in LDS.
static constexpr int kSsimTile = 16, kSsimMaxWin = 16;
__global__ __launch_bounds__(256) void ssim_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, int c, int h, int w, int cb,
                                                           int y_only, const double* __restrict__ window, int ws, int tiles_x,
                                                           double* __restrict__ partial) {
  constexpr int kIn = kSsimTile + kSsimMaxWin - 1;
  __shared__ double sa[kIn * kIn], sb[kIn * kIn], sw[kSsimMaxWin * kSsimMaxWin], red[256];
  const int img = blockIdx.z, ch = blockIdx.y;
  const int hh = h - 2 * cb, ww = w - 2 * cb;           // cropped image
  const int oh = hh - ws + 1, ow = ww - ws + 1;          // SSIM map
  const int ty0 = (blockIdx.x / tiles_x) * kSsimTile, tx0 = (blockIdx.x % tiles_x) * kSsimTile;
  const size_t plane = (size_t)h * w;
  const float* pa = a + (size_t)img * c * plane;
  const float* pb = b + (size_t)img * c * plane;
  const int in = kSsimTile + ws - 1;
  for (int i = threadIdx.x; i < ws * ws; i += 256) sw[i] = window[i];
  for (int i = threadIdx.x; i < in * in; i += 256) {
    const int iy = i / in, ix = i % in;
    const int y = ty0 + iy, x = tx0 + ix;
    double va = 0.0, vb = 0.0;
    if (y < hh && x < ww) {
      const size_t o = (size_t)(y + cb) * w + (x + cb);
      if (y_only) {
        float ya = pa[o] * 65.481f; ya = fmaf(pa[plane + o], 128.553f, ya); ya = fmaf(pa[2 * plane + o], 24.966f, ya); ya = (ya + 16.0f) / 255.f;
        float yb = pb[o] * 65.481f; yb = fmaf(pb[plane + o], 128.553f, yb); yb = fmaf(pb[2 * plane + o], 24.966f, yb); yb = (yb + 16.0f) / 255.f;
        va = (double)ya * 255.0; vb = (double)yb * 255.0;
      } else {
        va = (double)pa[ch * plane + o] * 255.0; vb = (double)pb[ch * plane + o] * 255.0;
      }
    }
    sa[iy * in + ix] = va; sb[iy * in + ix] = vb;
  }
  __syncthreads();
  const int ly = threadIdx.x / kSsimTile, lx = threadIdx.x % kSsimTile;
  double val = 0.0;
  if (ty0 + ly < oh && tx0 + lx < ow) {
    double ma = 0.0, mb = 0.0, saa = 0.0, sbb = 0.0, sab = 0.0;
    for (int ky = 0; ky < ws; ++ky)
      for (int kx = 0; kx < ws; ++kx) {
        const double g = sw[ky * ws + kx];
        const double xa = sa[(ly + ky) * in + lx + kx], xb = sb[(ly + ky) * in + lx + kx];
        ma = fma(g, xa, ma); mb = fma(g, xb, mb);
        saa = fma(g, xa * xa, saa); sbb = fma(g, xb * xb, sbb); sab = fma(g, xa * xb, sab);
      }
    const double c1 = (0.01 * 255.0) * (0.01 * 255.0), c2 = (0.03 * 255.0) * (0.03 * 255.0);
    const double ma2 = ma * ma, mb2 = mb * mb, mab = ma * mb;
    const double num = (2.0 * mab + c1) * (2.0 * (sab - mab) + c2);
    const double den = (ma2 + mb2 + c1) * ((saa - ma2) + (sbb - mb2) + c2);
    val = num / den;
  }
  red[threadIdx.x] = val;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[((size_t)img * gridDim.y + ch) * gridDim.x + blockIdx.x] = red[0];
}
// one block per image: fixed-order tree over its (channels * tiles) partials, then the mean (cast to fp32 like .float())
__global__ __launch_bounds__(256) void ssim_finish_kernel(const double* __restrict__ partial, int per_img, double count, float* __restrict__ out) {
  __shared__ double red[256];
  const double* p = partial + (size_t)blockIdx.x * per_img;
  double s = 0.0;
  for (int k = threadIdx.x; k < per_img; k += 256) s += p[k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(red[0] / count);
}

// ---- differentiable VGG tap (ESRGAN/model.py:281-292): gradient of mean |a - b| w.r.t. a, max-pool backward with the
// preceding ReLU's derivative folded in, and the relayout that also undoes the 1/std of the input normalisation ----
template <typename T>
__global__ __launch_bounds__(256) void l1_grad_views_kernel(const void* __restrict__ a, int aC, int a0, const void* __restrict__ b, int bC, int b0,
                                                            void* out, int oC, int o0, size_t npix, int c, const float* __restrict__ upstream, float scale) {
  const float sc = scale * (upstream ? *upstream : 1.f);
  const size_t total = npix * c;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    const size_t p = i / c;
    const float d = ld<T>(a, p * aC + a0 + ch) - ld<T>(b, p * bC + b0 + ch);
    st<T>(out, p * oC + o0 + ch, d > 0.f ? sc : (d < 0.f ? -sc : 0.f));     // torch: sign(0) = 0
  }
}
// x: pre-pool activation (a ReLU output), dy: gradient of the pooled map, dx: gradient w.r.t. the ReLU's INPUT.
// The gradient goes to the first maximum of each 2x2 window in row-major order (ATen max_pool2d) and is zero where
// that maximum is not positive (ReLU').
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_relu_bwd_kernel(const void* __restrict__ x, int xC, int x0, const void* __restrict__ dy, int yC, int y0,
                                                                void* dx, int dC, int d0, int n, int h, int w, int c) {
  const int ho = h / 2, wo = w / 2;
  const size_t total = (size_t)n * ho * wo * c;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    size_t p = i / c;
    const int ox = (int)(p % wo); p /= wo;
    const int oy = (int)(p % ho);
    const size_t img = p / ho;
    const size_t b = (img * h + 2 * oy) * w + 2 * ox;
    const size_t q[4] = {b, b + 1, b + w, b + w + 1};
    float best = ld<T>(x, q[0] * xC + x0 + ch);
    int arg = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float v = ld<T>(x, q[k] * xC + x0 + ch);
      if (v > best) { best = v; arg = k; }
    }
    const float g = best > 0.f ? ld<T>(dy, (i / c) * yC + y0 + ch) : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) st<T>(dx, q[k] * dC + d0 + ch, k == arg ? g : 0.f);
  }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_scaled_kernel(const float* __restrict__ src, int sC, int s0, float* __restrict__ dst, int n, int c,
                                                                  int hw, const float* __restrict__ ch_div) {
  const size_t total = (size_t)n * c * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t pix = i % hw;
    const size_t t = i / hw;
    const int ch = (int)(t % c);
    const size_t img = t / c;
    dst[i] = src[(img * hw + pix) * sC + s0 + ch] / ch_div[ch];
  }
}

// ------------------------------------------------------------------------------------------------
static inline unsigned grid_for(size_t total, int block = 256, unsigned cap = 8192) {
  size_t g = (total + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}
// CALL names the element type as TT
#define DISPATCH_T(dtype, CALL)                                                              \
  if ((dtype) == SRGANFD_BF16) { using TT = bf16_t; CALL; } else if ((dtype) == SRGANFD_F16) { using TT = f16_t; CALL; } \
  else if ((dtype) == SRGANFD_F32) { using TT = float; CALL; }                               \
  else return set_err(SRGANFD_EINVAL, "bad dtype %d", (int)(dtype));

int nchw_to_nhwc_impl(const float* src, int n, int c, int h, int w, srganfd_view dst, int dtype, int cpad, const float* mean, const float* stdv, hipStream_t s) {
  if (!src || !dst.ptr || n <= 0 || c <= 0 || cpad < c || dst.c0 + cpad > dst.cstride) return set_err(SRGANFD_EINVAL, "nchw_to_nhwc: bad args");
  const size_t total = (size_t)n * h * w * cpad;
  if (dtype != SRGANFD_F32 && c <= 4 && cpad == 4 && dst.cstride == 4 && dst.c0 == 0 && ((uintptr_t)dst.ptr & 7) == 0) {
    const size_t npix = (size_t)n * h * w;
    if (dtype == SRGANFD_BF16) SRGANFD_LAUNCH(nchw_to_nhwc4_kernel<bf16_t>, dim3(grid_for(npix)), dim3(256), 0, s, src, (unsigned long long*)dst.ptr, npix, (size_t)h * w, c, mean, stdv);
    else SRGANFD_LAUNCH(nchw_to_nhwc4_kernel<f16_t>, dim3(grid_for(npix)), dim3(256), 0, s, src, (unsigned long long*)dst.ptr, npix, (size_t)h * w, c, mean, stdv);
    SRGANFD_HIP_CHECK(hipGetLastError());
    return SRGANFD_OK;
  }
  {
    const int vn = dtype == SRGANFD_F32 ? 4 : 8;
    if (cpad % vn == 0 && dst.c0 % vn == 0 && dst.cstride % vn == 0 && ((uintptr_t)dst.ptr & 15) == 0) {
      DISPATCH_T(dtype,
                 SRGANFD_LAUNCH(nchw_to_nhwc_vec_kernel<TT>, dim3(grid_for(total / vn, 256, 65536)), dim3(256), 0, s, src, dst.ptr, dst.cstride, dst.c0, n, c, h * w, cpad, mean, stdv));
      SRGANFD_HIP_CHECK(hipGetLastError());
      return SRGANFD_OK;
    }
  }
  DISPATCH_T(dtype,
             SRGANFD_LAUNCH(nchw_to_nhwc_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, s, src, dst.ptr, dst.cstride, dst.c0, n, c, h * w, cpad, mean, stdv));
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int nhwc_to_nchw_impl(srganfd_view src, int dtype, int n, int c, int h, int w, float* dst, int clamp01, hipStream_t s) {
  if (!src.ptr || !dst || src.c0 + c > src.cstride) return set_err(SRGANFD_EINVAL, "nhwc_to_nchw: bad args");
  const size_t total = (size_t)n * h * w * c;
  DISPATCH_T(dtype,
             SRGANFD_LAUNCH(nhwc_to_nchw_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, s, src.ptr, src.cstride, src.c0, dst, n, c, h * w, clamp01));
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int clamp_grad_impl(const float* dsr, srganfd_view pre, int n, int c, int h, int w, srganfd_view dst, int dtype, int cpad, hipStream_t s) {
  if (!dsr || !pre.ptr || !dst.ptr || dst.c0 + cpad > dst.cstride) return set_err(SRGANFD_EINVAL, "clamp_grad: bad args");
  const size_t total = (size_t)n * h * w * cpad;
  if (dtype != SRGANFD_F32 && c <= 4 && cpad == 4 && pre.cstride == 4 && pre.c0 == 0 && dst.cstride == 4 && dst.c0 == 0 &&
      ((uintptr_t)pre.ptr & 15) == 0 && ((uintptr_t)dst.ptr & 7) == 0) {
    const size_t npix = (size_t)n * h * w;
    if (dtype == SRGANFD_BF16) SRGANFD_LAUNCH(clamp_grad_rgb4_kernel<bf16_t>, dim3(grid_for(npix)), dim3(256), 0, s, dsr, (const f32x4*)pre.ptr, (unsigned long long*)dst.ptr, npix, (size_t)h * w, c);
    else SRGANFD_LAUNCH(clamp_grad_rgb4_kernel<f16_t>, dim3(grid_for(npix)), dim3(256), 0, s, dsr, (const f32x4*)pre.ptr, (unsigned long long*)dst.ptr, npix, (size_t)h * w, c);
    SRGANFD_HIP_CHECK(hipGetLastError());
    return SRGANFD_OK;
  }
  if (dtype != SRGANFD_F32 && c <= 4 && cpad == 32 && pre.cstride == 4 && pre.c0 == 0 && dst.cstride == 32 && dst.c0 == 0 &&
      ((uintptr_t)pre.ptr & 15) == 0 && ((uintptr_t)dst.ptr & 15) == 0) {
    const size_t npix = (size_t)n * h * w;
    if (dtype == SRGANFD_BF16) SRGANFD_LAUNCH(clamp_grad_rgb16_kernel<bf16_t>, dim3(grid_for(npix)), dim3(256), 0, s, dsr, (const f32x4*)pre.ptr, (u32x4*)dst.ptr, npix, (size_t)h * w, c);
    else SRGANFD_LAUNCH(clamp_grad_rgb16_kernel<f16_t>, dim3(grid_for(npix)), dim3(256), 0, s, dsr, (const f32x4*)pre.ptr, (u32x4*)dst.ptr, npix, (size_t)h * w, c);
    SRGANFD_HIP_CHECK(hipGetLastError());
    return SRGANFD_OK;
  }
  DISPATCH_T(dtype,
             SRGANFD_LAUNCH(clamp_grad_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, s, dsr, (const float*)pre.ptr, pre.cstride, pre.c0, dst.ptr, dst.cstride, dst.c0, n, c, h * w, cpad));
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
// op: 0 nearest-x2 backward, 1 bilinear-x2 forward, 2 bilinear-x2 backward, 3 maxpool2 ; (h, w) = low-res dims (op 3: input dims)
int resample_impl(int op, srganfd_view a, srganfd_view b, int dtype, int n, int h, int w, int c, hipStream_t s) {
  if (!a.ptr || !b.ptr || a.c0 + c > a.cstride || b.c0 + c > b.cstride) return set_err(SRGANFD_EINVAL, "resample: bad args");
  const size_t lo = (size_t)n * h * w * c;
#define RS(K, TOTAL) DISPATCH_T(dtype, \
    SRGANFD_LAUNCH(K<TT>, dim3(grid_for(TOTAL)), dim3(256), 0, s, a.ptr, a.cstride, a.c0, b.ptr, b.cstride, b.c0, n, h, w, c))
  const int vn = dtype == SRGANFD_F32 ? 4 : 8;
  const bool vec = c % vn == 0 && a.c0 % vn == 0 && b.c0 % vn == 0 && a.cstride % vn == 0 && b.cstride % vn == 0 &&
                   ((uintptr_t)a.ptr & 15) == 0 && ((uintptr_t)b.ptr & 15) == 0;
#define RSV(OP, TOTAL) DISPATCH_T(dtype, \
    SRGANFD_LAUNCH((resample_vec_kernel<TT, OP>), dim3(grid_for((TOTAL) / vn, 256, 65536)), dim3(256), 0, s, a.ptr, a.cstride, a.c0, b.ptr, b.cstride, b.c0, n, h, w, c))
  // row-grid forms of the two bilinear ops: grid (column blocks, low-res rows, images)
  const int cv = c / vn, cv_shift = (cv & (cv - 1)) == 0 ? __builtin_ctz(cv) : -1;
  const bool rows_ok = h <= 65535 && n <= 65535 && (size_t)w * cv < (1u << 31);
  const dim3 rows_grid((unsigned)(((size_t)w * cv + 255) / 256), (unsigned)((h + kBilRows - 1) / kBilRows), (unsigned)n);
  if (vec) {
    if (op == 0) { RSV(0, lo); }
    else if (rows_ok && op == 1) {
      DISPATCH_T(dtype, SRGANFD_LAUNCH(bilinear_up2_block_kernel<TT>, rows_grid, dim3(256), 0, s, a.ptr, a.cstride, a.c0, b.ptr, b.cstride, b.c0, h, w, c, cv_shift));
    } else if (rows_ok && op == 2) {
      DISPATCH_T(dtype, SRGANFD_LAUNCH(bilinear_up2_bwd_rows_kernel<TT>, rows_grid, dim3(256), 0, s, a.ptr, a.cstride, a.c0, b.ptr, b.cstride, b.c0, h, w, c, cv_shift,
                                       (const void*)nullptr, 0, 0, (void*)nullptr, 0, 0, 0.f));
    }
    else if (op == 1) { RSV(1, lo * 4); }
    else if (op == 2) { RSV(2, lo); }
    else if (op == 3) { RSV(3, lo / 4); }
    else if (op == 4) { RSV(4, lo); }
    else return set_err(SRGANFD_EINVAL, "resample: bad op %d", op);
  }
  else if (op == 0) { RS(up2_nearest_bwd_kernel, lo); }
  else if (op == 1) { RS(up2_bilinear_fwd_kernel, lo * 4); }
  else if (op == 2) { RS(up2_bilinear_bwd_kernel, lo); }
  else if (op == 3) { RS(maxpool2_kernel, lo / 4); }
  else if (op == 4) { RS(relu_copy_kernel, lo); }
  else return set_err(SRGANFD_EINVAL, "resample: bad op %d", op);
#undef RS
#undef RSV
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
// bilinear-x2 backward fused with the LeakyReLU' of the upsampled layer: dx_raw (optional) = adjoint of the upsampling applied to dy,
// dx_masked = dx_raw * (act > 0 ? 1 : slope).  16-byte-vectorised views only (the discriminators' channel counts).
int resample_bwd_lrelu_impl(srganfd_view dy, srganfd_view dx_raw, srganfd_view act, srganfd_view dx_masked, int dtype, int n, int h, int w, int c, float slope,
                            hipStream_t s) {
  if (!dy.ptr || !act.ptr || !dx_masked.ptr) return set_err(SRGANFD_EINVAL, "resample_bwd_lrelu: null view");
  const int vn = dtype == SRGANFD_F32 ? 4 : 8;
  auto ok = [&](const srganfd_view& v) { return !v.ptr || (v.c0 % vn == 0 && v.cstride % vn == 0 && ((uintptr_t)v.ptr & 15) == 0 && v.c0 + c <= v.cstride && !v.planar); };
  if (c % vn || !ok(dy) || !ok(dx_raw) || !ok(act) || !ok(dx_masked)) return set_err(SRGANFD_EINVAL, "resample_bwd_lrelu: views must be 16-byte aligned NHWC slices");
  const size_t lo = (size_t)n * h * w * c;
  const int cv = c / vn, cv_shift = (cv & (cv - 1)) == 0 ? __builtin_ctz(cv) : -1;
  if (h <= 65535 && n <= 65535 && (size_t)w * cv < (1u << 31)) {
    // non-temporal stores for results one pass cannot keep in the 256 MiB Infinity Cache anyway: 867 -> 805 us (128 channels, 512^2 -> 256^2, batch 32),
    // 427 -> 414 (256 channels), bit-equal; 8 / 16 rows per thread instead of 4 measured 3-10 % slower (tools/r4/bil_bwd_bench.py)
    const bool nt = lo * (size_t)(dtype == SRGANFD_F32 ? 4 : 2) >= ((size_t)192 << 20);
#define BB(NTT) DISPATCH_T(dtype, SRGANFD_LAUNCH((bilinear_up2_bwd_rows_kernel<TT, kBilRows, NTT>), dim3((unsigned)(((size_t)w * cv + 255) / 256), (unsigned)((h + kBilRows - 1) / kBilRows), (unsigned)n), dim3(256), 0, s, dy.ptr, \
                                     dy.cstride, dy.c0, dx_raw.ptr, dx_raw.cstride, dx_raw.c0, h, w, c, cv_shift, (const void*)act.ptr, act.cstride, act.c0, \
                                     dx_masked.ptr, dx_masked.cstride, dx_masked.c0, slope))
    if (nt) { BB(true); } else { BB(false); }
#undef BB
  } else
  DISPATCH_T(dtype, SRGANFD_LAUNCH((resample_vec_kernel<TT, 2>), dim3(grid_for(lo / vn, 256, 65536)), dim3(256), 0, s, dy.ptr, dy.cstride, dy.c0, dx_raw.ptr, dx_raw.cstride,
                                   dx_raw.c0, n, h, w, c, (const void*)act.ptr, act.cstride, act.c0, dx_masked.ptr, dx_masked.cstride, dx_masked.c0, slope));
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int lrelu_bwd_impl(srganfd_view dy, srganfd_view act, srganfd_view skip, srganfd_view out, int dtype, size_t npix, int c, float slope, hipStream_t s) {
  if (!dy.ptr || !act.ptr || !out.ptr) return set_err(SRGANFD_EINVAL, "lrelu_bwd: null");
  {
    const int vn = dtype == SRGANFD_F32 ? 4 : 8;
    auto ok = [&](const srganfd_view& v) { return !v.ptr || (v.c0 % vn == 0 && v.cstride % vn == 0 && ((uintptr_t)v.ptr & 15) == 0); };
    if (c % vn == 0 && ok(dy) && ok(act) && ok(skip) && ok(out)) {
      DISPATCH_T(dtype,
                 SRGANFD_LAUNCH(lrelu_bwd_vec_kernel<TT>, dim3(grid_for(npix * c / vn, 256, 65536)), dim3(256), 0, s, dy.ptr, dy.cstride, dy.c0, act.ptr, act.cstride, act.c0,
                                skip.ptr, skip.cstride, skip.c0, out.ptr, out.cstride, out.c0, npix, c, slope));
      SRGANFD_HIP_CHECK(hipGetLastError());
      return SRGANFD_OK;
    }
  }
  DISPATCH_T(dtype,
             SRGANFD_LAUNCH(lrelu_bwd_kernel<TT>, dim3(grid_for(npix * c)), dim3(256), 0, s, dy.ptr, dy.cstride, dy.c0, act.ptr, act.cstride, act.c0,
                                skip.ptr, skip.cstride, skip.c0, out.ptr, out.cstride, out.c0, npix, c, slope));
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int axpby_impl(srganfd_view x, srganfd_view y, int dtype, size_t npix, int c, float a, float b, hipStream_t s) {
  if (!x.ptr || !y.ptr) return set_err(SRGANFD_EINVAL, "axpby: null");
  {
    const int vn = dtype == SRGANFD_F32 ? 4 : 8;
    auto ok = [&](const srganfd_view& v) { return v.c0 % vn == 0 && v.cstride % vn == 0 && ((uintptr_t)v.ptr & 15) == 0; };
    if (c % vn == 0 && ok(x) && ok(y)) {
      DISPATCH_T(dtype,
                 SRGANFD_LAUNCH(axpby_vec_kernel<TT>, dim3(grid_for(npix * c / vn, 256, 65536)), dim3(256), 0, s, x.ptr, x.cstride, x.c0, y.ptr, y.cstride, y.c0, npix, c, a, b));
      SRGANFD_HIP_CHECK(hipGetLastError());
      return SRGANFD_OK;
    }
  }
  DISPATCH_T(dtype,
             SRGANFD_LAUNCH(axpby_kernel<TT>, dim3(grid_for(npix * c)), dim3(256), 0, s, x.ptr, x.cstride, x.c0, y.ptr, y.cstride, y.c0, npix, c, a, b));
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

static constexpr int kRedBlocks = 1024;  // workspace floats needed by the loss entry points: 2 * kRedBlocks

int l1_loss_impl(const float* a, const float* b, size_t n, float weight, float* out, int accumulate, float* grad, float grad_scale,
                 const float* grad_scale_dev, float* ws, hipStream_t s) {
  if (!a || !b || !out || !ws || n == 0) return set_err(SRGANFD_EINVAL, "l1_loss: bad args");
  const unsigned g = grid_for(n, 256, kRedBlocks);
  SRGANFD_LAUNCH(l1_partial_kernel, dim3(g), dim3(256), 0, s, a, b, n, grad_scale / (float)n, grad_scale_dev, grad, ws);
  SRGANFD_LAUNCH(finish_sum_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, (int)g, weight / (float)n, out, accumulate);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int l1_views_impl(srganfd_view a, srganfd_view b, int dtype, size_t npix, int c, int relu, float weight, float* out, int accumulate, float* ws, hipStream_t s) {
  if (!a.ptr || !b.ptr || !out || !ws) return set_err(SRGANFD_EINVAL, "l1_views: bad args");
  const size_t n = npix * c;
  const unsigned g = grid_for(n, 256, kRedBlocks);
  const int vn = dtype == SRGANFD_F32 ? 4 : 8;
  if (c % vn == 0 && a.c0 % vn == 0 && b.c0 % vn == 0 && a.cstride % vn == 0 && b.cstride % vn == 0 && ((uintptr_t)a.ptr & 15) == 0 && ((uintptr_t)b.ptr & 15) == 0) {
    DISPATCH_T(dtype,
               SRGANFD_LAUNCH(l1_views_vec_partial_kernel<TT>, dim3(g), dim3(256), 0, s, a.ptr, a.cstride, a.c0, b.ptr, b.cstride, b.c0, npix, c, relu, ws));
  } else
  DISPATCH_T(dtype,
             SRGANFD_LAUNCH(l1_views_partial_kernel<TT>, dim3(g), dim3(256), 0, s, a.ptr, a.cstride, a.c0, b.ptr, b.cstride, b.c0, npix, c, relu, ws));
  SRGANFD_LAUNCH(finish_sum_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, (int)g, weight / (float)n, out, accumulate);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int bce_logits_impl(const float* x, size_t n, float target, float weight, float* loss_out, int accumulate, float* sig_mean_out, float* grad,
                    float grad_scale, const float* grad_scale_dev, float* ws, hipStream_t s) {
  if (!x || !loss_out || !ws || n == 0) return set_err(SRGANFD_EINVAL, "bce: bad args");
  const unsigned g = grid_for(n, 256, kRedBlocks);
  SRGANFD_LAUNCH(bce_partial_kernel, dim3(g), dim3(256), 0, s, x, n, target, grad_scale / (float)n, grad_scale_dev, grad, ws, ws + kRedBlocks);
  SRGANFD_LAUNCH(finish_sum_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, (int)g, weight / (float)n, loss_out, accumulate);
  if (sig_mean_out) SRGANFD_LAUNCH(finish_sum_kernel, dim3(1), dim3(256), 0, s, (const float*)(ws + kRedBlocks), (int)g, 1.f / (float)n, sig_mean_out, 0);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int sigmoid_of_mean_impl(const float* x, size_t n, float* out, float* ws, hipStream_t s) {
  if (!x || !out || !ws || n == 0) return set_err(SRGANFD_EINVAL, "sigmoid_of_mean: bad args");
  const unsigned g = grid_for(n, 256, kRedBlocks);
  SRGANFD_LAUNCH(sum_partial_kernel, dim3(g), dim3(256), 0, s, x, n, ws);
  SRGANFD_LAUNCH(finish_sigmoid_mean_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, (int)g, 1.f / (float)n, out);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
// workspace: 2 * kRedBlocks + 1 floats (SRGANFD_LOSS_WS_FLOATS)
int bce_logits_relativistic_impl(const float* x, size_t n, const float* other, size_t n_other, float target, float weight, float* loss_out, int accumulate,
                                 float* grad_x, int accumulate_x, float* grad_other, int accumulate_other, float grad_scale,
                                 const float* grad_scale_dev, float* ws, hipStream_t s) {
  if (!x || !other || !loss_out || !ws || n == 0 || n_other == 0) return set_err(SRGANFD_EINVAL, "bce_relativistic: bad args");
  float* mean = ws + 2 * kRedBlocks;
  const unsigned go = grid_for(n_other, 256, kRedBlocks), g = grid_for(n, 256, kRedBlocks);
  SRGANFD_LAUNCH(sum_partial_kernel, dim3(go), dim3(256), 0, s, other, n_other, ws);
  SRGANFD_LAUNCH(finish_mean_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, (int)go, 1.f / (float)n_other, mean);
  SRGANFD_LAUNCH(bce_rel_partial_kernel, dim3(g), dim3(256), 0, s, x, n, (const float*)mean, target, grad_scale / (float)n, grad_scale_dev, grad_x,
                 accumulate_x, ws, ws + kRedBlocks);
  SRGANFD_LAUNCH(finish_sum_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, (int)g, weight / (float)n, loss_out, accumulate);
  if (grad_other)
    SRGANFD_LAUNCH(bce_rel_other_kernel, dim3(grid_for(n_other, 256, 256)), dim3(256), 0, s, (const float*)(ws + kRedBlocks), (int)g, grad_scale / (float)n,
                   grad_scale_dev, grad_other, n_other, accumulate_other);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
// each job's workspace: ceil(rows / 32) * cols + rows floats
int spectral_norm_batch_impl(const srganfd_sn_job* jobs, int njobs, int training, float eps, hipStream_t s) {
  if (!jobs || njobs <= 0) return set_err(SRGANFD_EINVAL, "spectral_norm: no jobs");
  for (int b = 0; b < njobs; b += kSnBatch) {
    const int nb = std::min(kSnBatch, njobs - b);
    SnJobs J;
    int max_rows = 0, max_cols = 0;
    for (int i = 0; i < nb; ++i) {
      const srganfd_sn_job& q = jobs[b + i];
      if (!q.w_orig || !q.u || !q.v || !q.sigma_out || !q.inv_sigma_out || !q.workspace || q.rows <= 0 || q.cols <= 0)
        return set_err(SRGANFD_EINVAL, "spectral_norm: bad args");
      J.j[i] = q; max_rows = std::max(max_rows, q.rows); max_cols = std::max(max_cols, q.cols);
    }
    for (int i = nb; i < kSnBatch; ++i) J.j[i] = J.j[0];            // never indexed: the grids stop at nb
    if (training) {
      SRGANFD_LAUNCH(sn_wt_u_kernel, dim3((max_cols + 255) / 256, (max_rows + kSnRows - 1) / kSnRows, nb), dim3(256), 0, s, J);
      SRGANFD_LAUNCH(sn_normalize_kernel, dim3(nb), dim3(1024), 0, s, J, eps);
    }
    SRGANFD_LAUNCH(sn_w_v_kernel, dim3(max_rows, nb), dim3(256), 0, s, J);
    SRGANFD_LAUNCH(sn_finish_kernel, dim3(nb), dim3(256), 0, s, J, eps, training);
  }
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int spectral_norm_impl(const float* W, float* u, float* v, int rows, int cols, int training, float eps, float* sigma, float* inv_sigma, float* ws, hipStream_t s) {
  srganfd_sn_job q;
  q.w_orig = W; q.u = u; q.v = v; q.sigma_out = sigma; q.inv_sigma_out = inv_sigma; q.workspace = ws; q.rows = rows; q.cols = cols;
  return spectral_norm_batch_impl(&q, 1, training, eps, s);
}
// each job's workspace: kRedBlocks + 1 floats
int spectral_norm_grad_batch_impl(const srganfd_sn_grad_job* jobs, int njobs, float beta, hipStream_t s) {
  static_assert(kSnGradBlocks == kRedBlocks, "workspace contract of srganfd_spectral_norm_grad");
  if (!jobs || njobs <= 0) return set_err(SRGANFD_EINVAL, "spectral_norm_grad: no jobs");
  for (int b = 0; b < njobs; b += kSnBatch) {
    const int nb = std::min(kSnBatch, njobs - b);
    SnGradJobs J;
    size_t max_n = 0;
    for (int i = 0; i < nb; ++i) {
      const srganfd_sn_grad_job& q = jobs[b + i];
      if (!q.g_weight || !q.w_orig || !q.u || !q.v || !q.inv_sigma || !q.dw_orig || !q.workspace || q.rows <= 0 || q.cols <= 0)
        return set_err(SRGANFD_EINVAL, "spectral_norm_grad: bad args");
      J.j[i] = q; max_n = std::max(max_n, (size_t)q.rows * q.cols);
    }
    for (int i = nb; i < kSnBatch; ++i) J.j[i] = J.j[0];
    SRGANFD_LAUNCH(sn_dot_partial_kernel, dim3(grid_for(max_n, 256, kRedBlocks), nb), dim3(256), 0, s, J);
    SRGANFD_LAUNCH(sn_dot_finish_kernel, dim3(nb), dim3(256), 0, s, J);
    SRGANFD_LAUNCH(sn_grad_kernel, dim3(grid_for(max_n), nb), dim3(256), 0, s, J, beta);
  }
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int spectral_norm_grad_impl(const float* G, const float* W, const float* u, const float* v, const float* inv_sigma, float* dW, int rows, int cols,
                            float beta, float* ws, hipStream_t s) {
  srganfd_sn_grad_job q;
  q.g_weight = G; q.w_orig = W; q.u = u; q.v = v; q.inv_sigma = inv_sigma; q.dw_orig = dW; q.workspace = ws; q.rows = rows; q.cols = cols;
  return spectral_norm_grad_batch_impl(&q, 1, beta, s);
}
// flag = 1 if any element of x is inf or NaN (the found_inf of torch.cuda.amp.GradScaler.unscale_, train_bsrgan.py:436,466)
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float* __restrict__ x, size_t n, float* __restrict__ flag) {
  bool bad = false;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = ((const f32x4*)x)[i];
    bad |= !(fabsf(v[0]) <= 3.402823466e38f) | !(fabsf(v[1]) <= 3.402823466e38f) | !(fabsf(v[2]) <= 3.402823466e38f) | !(fabsf(v[3]) <= 3.402823466e38f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= !(fabsf(x[n4 * 4 + threadIdx.x]) <= 3.402823466e38f);
  if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 1.f;      // every writer stores the same value
}
int nonfinite_flag_impl(const float* x, size_t n, float* flag, int accumulate, hipStream_t s) {
  if (!x || !flag || n == 0 || ((uintptr_t)x & 15)) return set_err(SRGANFD_EINVAL, "nonfinite_flag: bad args");
  if (!accumulate) SRGANFD_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(float), s));
  SRGANFD_LAUNCH(nonfinite_flag_kernel, dim3(grid_for(n / 4 + 1, 256, 2048)), dim3(256), 0, s, x, n, flag);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

int adam_ema_impl(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float b1, float b2, float eps, float wd, int step,
                  float grad_scale, float ema_decay, int ema_mode, const float* skip_flag, const float* grad_scale_dev, hipStream_t s) {
  if (!p || !g || !m || !v || n == 0 || step < 1 || (ema_mode && !ema)) return set_err(SRGANFD_EINVAL, "adam: bad args");
  const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
  SRGANFD_LAUNCH(adam_ema_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, g, m, v, ema, n, lr, b1, b2, eps, wd, (float)bc1, (float)sqrt(bc2),
                     grad_scale, ema_decay, ema_mode, skip_flag, grad_scale_dev);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

// torch.amp.GradScaler.update() (torch/amp/grad_scaler.py, _amp_update_scale_) on a device-resident state, so that neither the host nor a
// captured graph ever carries a stale scale: state = {scale, 1 / scale, growth tracker, optimizer steps, skipped steps}.  The three
// counters are int32 words of the same 8-word state (a float stops counting at 2^24 steps; torch's tracker is an int32 tensor too).
__global__ void loss_scale_update_kernel(float* __restrict__ st, const float* __restrict__ found_inf, float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int* sti = reinterpret_cast<int*>(st);
  float scale = st[0];
  int tracker = sti[2];
  sti[3] += 1;
  if (*found_inf != 0.f) {
    scale *= backoff; tracker = 0; sti[4] += 1;
  } else {
    tracker += 1;
    if (tracker >= interval) {
      const float grown = scale * growth;
      if (fabsf(grown) <= 3.402823466e38f) scale = grown;       // torch keeps the scale when growing it would overflow
      tracker = 0;
    }
  }
  st[0] = scale; st[1] = 1.f / scale; sti[2] = tracker;
}
int loss_scale_update_impl(float* state, const float* found_inf, float growth, float backoff, int interval, hipStream_t s) {
  if (!state || !found_inf || interval < 1 || !(growth >= 1.f) || !(backoff > 0.f && backoff <= 1.f)) return set_err(SRGANFD_EINVAL, "loss_scale_update: bad args");
  SRGANFD_LAUNCH(loss_scale_update_kernel, dim3(1), dim3(64), 0, s, state, found_inf, growth, backoff, interval);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

int adam_ema_dev_impl(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float b1, float b2, float eps, float wd,
                      int* step_dev, float* bc_dev, float grad_scale, float ema_decay, int ema_mode, const float* skip_flag,
                      const float* grad_scale_dev, hipStream_t s) {
  if (!p || !g || !m || !v || n == 0 || !step_dev || !bc_dev || (ema_mode && !ema)) return set_err(SRGANFD_EINVAL, "adam(dev): bad args");
  SRGANFD_LAUNCH(adam_step_kernel, dim3(1), dim3(64), 0, s, step_dev, b1, b2, bc_dev, skip_flag);
  SRGANFD_LAUNCH(adam_ema_dev_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, g, m, v, ema, n, lr, b1, b2, eps, wd, (const float*)bc_dev, grad_scale,
                 ema_decay, ema_mode, skip_flag, grad_scale_dev);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

static bool vec_ok(int dtype, int c, std::initializer_list<srganfd_view> vs) {
  const int vn = dtype == SRGANFD_F32 ? 4 : 8;
  if (c % vn) return false;
  for (const auto& v : vs)
    if (v.ptr && (v.c0 % vn || v.cstride % vn || ((uintptr_t)v.ptr & 15))) return false;
  return true;
}
int resize_bilinear_impl(int bwd, srganfd_view a, srganfd_view b, int dtype, int n, int hi, int wi, int ho, int wo, int c, hipStream_t s) {
  if (!a.ptr || !b.ptr || !vec_ok(dtype, c, {a, b})) return set_err(SRGANFD_EINVAL, "resize_bilinear: views must be 16-byte aligned channel multiples");
  const int vn = dtype == SRGANFD_F32 ? 4 : 8;
  if (!bwd) {
    const size_t total = (size_t)n * ho * wo * c / vn;
    DISPATCH_T(dtype,
               SRGANFD_LAUNCH(resize_fwd_kernel<TT>, dim3(grid_for(total, 256, 65536)), dim3(256), 0, s, a.ptr, a.cstride, a.c0, b.ptr, b.cstride, b.c0, n, hi, wi, ho, wo, c));
  } else {
    const size_t total = (size_t)n * hi * wi * c / vn;
    DISPATCH_T(dtype,
               SRGANFD_LAUNCH(resize_bwd_kernel<TT>, dim3(grid_for(total, 256, 65536)), dim3(256), 0, s, a.ptr, a.cstride, a.c0, b.ptr, b.cstride, b.c0, n, hi, wi, ho, wo, c));
  }
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int add_relu_impl(srganfd_view a, srganfd_view b, srganfd_view out, int dtype, size_t npix, int c, hipStream_t s) {
  if (!a.ptr || !b.ptr || !out.ptr || !vec_ok(dtype, c, {a, b, out})) return set_err(SRGANFD_EINVAL, "add_relu: bad views");
  const int vn = dtype == SRGANFD_F32 ? 4 : 8;
  DISPATCH_T(dtype,
             SRGANFD_LAUNCH(add_relu_kernel<TT>, dim3(grid_for(npix * c / vn, 256, 65536)), dim3(256), 0, s, a.ptr, a.cstride, a.c0, b.ptr, b.cstride, b.c0, out.ptr, out.cstride, out.c0, npix, c));
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int sigmoid_impl(float* x, size_t n, hipStream_t s) {
  if (!x) return set_err(SRGANFD_EINVAL, "sigmoid: null");
  SRGANFD_LAUNCH(sigmoid_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, n);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int sigmoid_bwd_impl(const float* ds, const float* sg, float* out, size_t n, hipStream_t s) {
  if (!ds || !sg || !out) return set_err(SRGANFD_EINVAL, "sigmoid_bwd: null");
  SRGANFD_LAUNCH(sigmoid_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, ds, sg, out, n);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int gate_mul_impl(int bwd, srganfd_view x, const float* gate, srganfd_view y, srganfd_view dx, float* dgate, int dtype, size_t npix, int c, hipStream_t s) {
  const int vn = dtype == SRGANFD_F32 ? 4 : 8;
  const int cv = c / vn;
  if (!x.ptr || !gate || !y.ptr || !vec_ok(dtype, c, {x, y, dx}) || cv > 64 || (cv & (cv - 1))) return set_err(SRGANFD_EINVAL, "gate_mul: bad args");
  if (!bwd) {
    DISPATCH_T(dtype,
               SRGANFD_LAUNCH(gate_fwd_kernel<TT>, dim3(grid_for(npix * cv, 256, 65536)), dim3(256), 0, s, x.ptr, x.cstride, x.c0, gate, y.ptr, y.cstride, y.c0, npix, c));
  } else {
    if (!dx.ptr || !dgate) return set_err(SRGANFD_EINVAL, "gate_mul(bwd): null");
    DISPATCH_T(dtype,
               SRGANFD_LAUNCH(gate_bwd_kernel<TT>, dim3(grid_for(npix * cv, 256, 65536)), dim3(256), 0, s, x.ptr, x.cstride, x.c0, gate, y.ptr, y.cstride, y.c0, dx.ptr, dx.cstride, dx.c0, dgate, npix, c));
  }
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
static constexpr int kBnBlocks = 1024;  // workspace: kBnBlocks * 2 * c floats (+ 3c for the backward coefficients)
long long batchnorm_partial_floats_impl(int c) { return (long long)kBnBlocks * 2 * c; }
static inline bool bn_chunks_ok(int dtype, int c) { const int cv = c / (dtype == SRGANFD_F32 ? 4 : 8); return cv > 0 && 256 % cv == 0; }
static inline unsigned bn_grid(size_t npix, int dtype, int c) { const int lanes = 256 / (c / (dtype == SRGANFD_F32 ? 4 : 8)); return grid_for((npix + lanes - 1) / lanes, 1, 16384); }
// Channels are processed in blocks of <= 256 (the statistics kernels map one thread to one channel); `save` is
// [block][mean | invstd | scale | shift] and is only read back by batchnorm_bwd_impl with the same blocking.
static inline srganfd_view sub_view(srganfd_view v, int cb) { if (v.ptr) v.c0 += cb; return v; }
// phase (data-parallel SyncBN): 0 = statistics, finish and apply in one call; 1 = this rank's partial sums into ws only (the caller
// all-reduces the first batchnorm_partial_floats(c) floats of ws over the ranks); 2 = finish + apply from ws with total_npix pixels.
int batchnorm_fwd_impl(srganfd_view x, srganfd_view y, int dtype, size_t npix, int c, const float* gamma, const float* beta, float* rm, float* rv,
                       float momentum, float eps, int training, float* save, float* ws, float act_slope, hipStream_t s, int phase = 0,
                       size_t total_npix = 0) {
  if (!x.ptr || !y.ptr || !gamma || !beta || !rm || !rv || !save || !ws || c <= 0 || !vec_ok(dtype, c, {x, y}))
    return set_err(SRGANFD_EINVAL, "batchnorm_fwd: bad args (16-byte aligned views)");
  if (phase && (c > 256 || !training)) return set_err(SRGANFD_EINVAL, "batchnorm_fwd: the two-phase form takes training mode and at most 256 channels");
  const float count = (float)(phase == 2 ? total_npix : npix);
  for (int cb = 0; cb < c; cb += 256) {
    const int cc = c - cb < 256 ? c - cb : 256;
    if (!bn_chunks_ok(dtype, cc)) return set_err(SRGANFD_EINVAL, "batchnorm_fwd: channel block of %d is not a power-of-two number of 16-byte chunks", cc);
    const srganfd_view xs = sub_view(x, cb), ys = sub_view(y, cb);
    float* sv = save + 4 * cb;
    if (training && phase != 2) {
      DISPATCH_T(dtype,
                 SRGANFD_LAUNCH(bn_partial_kernel<TT>, dim3(kBnBlocks), dim3(256), 0, s, xs.ptr, xs.cstride, xs.c0, (const void*)nullptr, 0, 0, (const float*)nullptr, npix, cc, ws, (const void*)nullptr, 0, 0, 1.f));
    }
    if (phase == 1) continue;
    SRGANFD_LAUNCH(bn_fwd_finish_kernel, dim3(1), dim3(1024), 0, s, (const float*)ws, kBnBlocks, cc, count, gamma + cb, beta + cb, rm + cb, rv + cb, momentum, eps, training, sv);
    DISPATCH_T(dtype,
               SRGANFD_LAUNCH(chan_affine_kernel<TT>, dim3(bn_grid(npix, dtype, cc)), dim3(256), 0, s, xs.ptr, xs.cstride, xs.c0, (const void*)nullptr, 0, 0,
                              ys.ptr, ys.cstride, ys.c0, (const float*)(sv + 2 * cc), (const float*)nullptr, (const float*)(sv + 3 * cc), npix, cc, act_slope, (const void*)nullptr, 0, 0, 1.f));
  }
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
// phase as in batchnorm_fwd_impl; phase 2 takes ws_global = the partial table summed over the ranks (ws keeps this rank's own)
int batchnorm_bwd_impl(srganfd_view x, srganfd_view dy, srganfd_view dx, int dtype, size_t npix, int c, const float* gamma, const float* save,
                       float* dgamma, float* dbeta, float acc, float* ws, srganfd_view act, float act_slope, hipStream_t s, int phase = 0,
                       const float* ws_global = nullptr, size_t total_npix = 0) {
  if (!x.ptr || !dy.ptr || !dx.ptr || !gamma || !save || !dgamma || !dbeta || !ws || c <= 0 || !vec_ok(dtype, c, {x, dy, dx, act}))
    return set_err(SRGANFD_EINVAL, "batchnorm_bwd: bad args");
  if (phase && c > 256) return set_err(SRGANFD_EINVAL, "batchnorm_bwd: the two-phase form takes at most 256 channels");
  if (phase == 2 && !ws_global) return set_err(SRGANFD_EINVAL, "batchnorm_bwd: phase 2 needs the all-reduced table");
  const float count = (float)(phase == 2 ? total_npix : npix);
  for (int cb = 0; cb < c; cb += 256) {
    const int cc = c - cb < 256 ? c - cb : 256;
    if (!bn_chunks_ok(dtype, cc)) return set_err(SRGANFD_EINVAL, "batchnorm_bwd: channel block of %d is not a power-of-two number of 16-byte chunks", cc);
    const srganfd_view xs = sub_view(x, cb), dys = sub_view(dy, cb), dxs = sub_view(dx, cb), as = sub_view(act, cb);
    const float* sv = save + 4 * cb;
    float* coef = ws + (size_t)kBnBlocks * 2 * cc;
    if (phase != 2) {
      DISPATCH_T(dtype,
                 SRGANFD_LAUNCH(bn_partial_kernel<TT>, dim3(kBnBlocks), dim3(256), 0, s, xs.ptr, xs.cstride, xs.c0, (const void*)dys.ptr, dys.cstride, dys.c0, sv, npix, cc, ws, (const void*)as.ptr, as.cstride, as.c0, act_slope));
    }
    if (phase == 1) continue;
    SRGANFD_LAUNCH(bn_bwd_finish_kernel, dim3(1), dim3(1024), 0, s, (const float*)ws, kBnBlocks, cc, count, gamma + cb, sv, dgamma + cb, dbeta + cb, acc, coef,
                   phase == 2 ? ws_global : (const float*)nullptr);
    DISPATCH_T(dtype,
               SRGANFD_LAUNCH(chan_affine_kernel<TT>, dim3(bn_grid(npix, dtype, cc)), dim3(256), 0, s, dys.ptr, dys.cstride, dys.c0, (const void*)xs.ptr, xs.cstride, xs.c0,
                              dxs.ptr, dxs.cstride, dxs.c0, (const float*)coef, (const float*)(coef + cc), (const float*)(coef + 2 * cc), npix, cc, 1.f, (const void*)as.ptr, as.cstride, as.c0, act_slope));
  }
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

// ---- uint8 ingest (dataset.py:64-96): what the reference does per image on the host -- cv2.imread(...).astype(float32) / 255,
// crop, BGR -> RGB, image_to_tensor's HWC -> CHW (imgproc.py:331-358) -- for a whole batch of decoded uint8 HWC images on the device:
// a quarter of the host-to-device bytes, and the float batch never exists in host memory.  One thread per output pixel: three byte reads
// of one pixel (a wave reads 192 contiguous bytes), three coalesced plane stores.
__global__ __launch_bounds__(256) void u8hwc_to_nchw_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int n, int h, int w, int top, int left,
                                                            int ph, int pw, int swap_rb, float scale) {
  const size_t total = (size_t)n * ph * pw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % pw);
    const size_t t = i / pw;
    const int y = (int)(t % ph);
    const size_t img = t / ph;
    const unsigned char* p = src + ((img * h + top + y) * (size_t)w + left + x) * 3;
    const float c0 = (float)p[0] / scale, c1 = (float)p[1] / scale, c2 = (float)p[2] / scale;
    float* d = dst + (img * 3 * ph + y) * (size_t)pw + x;
    const size_t plane = (size_t)ph * pw;
    d[0] = swap_rb ? c2 : c0;
    d[plane] = c1;
    d[2 * plane] = swap_rb ? c0 : c2;
  }
}
int u8hwc_to_nchw_impl(const unsigned char* src, float* dst, int n, int h, int w, int top, int left, int ph, int pw, int swap_rb, float scale, hipStream_t s) {
  if (!src || !dst || n <= 0 || ph <= 0 || pw <= 0 || top < 0 || left < 0 || top + ph > h || left + pw > w || !(scale > 0.f))
    return set_err(SRGANFD_EINVAL, "u8hwc_to_nchw: bad args / window outside the image");
  SRGANFD_LAUNCH(u8hwc_to_nchw_kernel, dim3(grid_for((size_t)n * ph * pw)), dim3(256), 0, s, src, dst, n, h, w, top, left, ph, pw, swap_rb, scale);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

static constexpr int kPsnrBlocks = 64;   // workspace: n * kPsnrBlocks doubles
int crop_nchw_impl(const float* src, float* dst, int n, int c, int h, int w, int top, int left, int ph, int pw, hipStream_t s) {
  if (!src || !dst || n <= 0 || c <= 0 || top < 0 || left < 0 || ph <= 0 || pw <= 0 || top + ph > h || left + pw > w)
    return set_err(SRGANFD_EINVAL, "crop: window %dx%d at (%d,%d) outside %dx%d", ph, pw, top, left, h, w);
  SRGANFD_LAUNCH(crop_nchw_kernel, dim3(grid_for((size_t)n * c * ph * pw)), dim3(256), 0, s, src, dst, n * c, h, w, top, left, ph, pw);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int psnr_impl(const float* a, const float* b, int n, int c, int h, int w, int crop_border, int y_only, double* out, double* ws, hipStream_t s) {
  if (!a || !b || !out || !ws || n <= 0 || c <= 0 || crop_border < 0 || h - 2 * crop_border <= 0 || w - 2 * crop_border <= 0 || (y_only && c != 3))
    return set_err(SRGANFD_EINVAL, "psnr: bad args (Y channel needs 3-channel RGB input)");
  SRGANFD_LAUNCH(psnr_partial_kernel, dim3(kPsnrBlocks, n), dim3(256), 0, s, a, b, c, h, w, crop_border, y_only, ws);
  const double count = (double)(y_only ? 1 : c) * (h - 2 * crop_border) * (w - 2 * crop_border);
  SRGANFD_LAUNCH(psnr_finish_kernel, dim3(n), dim3(64), 0, s, (const double*)ws, kPsnrBlocks, count, out);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

int64_t ssim_workspace_doubles(int n, int c, int h, int w, int crop_border, int y_only, int ws) {
  const int oh = h - 2 * crop_border - ws + 1, ow = w - 2 * crop_border - ws + 1;
  if (n <= 0 || c <= 0 || oh <= 0 || ow <= 0) return 0;
  const int64_t tiles = (int64_t)((oh + kSsimTile - 1) / kSsimTile) * ((ow + kSsimTile - 1) / kSsimTile);
  return (int64_t)n * (y_only ? 1 : c) * tiles;
}
int ssim_impl(const float* a, const float* b, int n, int c, int h, int w, int crop_border, int y_only, const double* window, int ws, float* out,
              double* wsp, hipStream_t s) {
  const int oh = h - 2 * crop_border - ws + 1, ow = w - 2 * crop_border - ws + 1;
  if (!a || !b || !out || !wsp || !window || n <= 0 || c <= 0 || crop_border < 0 || ws < 1 || ws > kSsimMaxWin || oh <= 0 || ow <= 0 ||
      (y_only && c != 3) || n > 65535 || c > 65535)
    return set_err(SRGANFD_EINVAL, "ssim: bad args (window 1..%d inside the cropped image; Y channel needs 3-channel RGB input)", kSsimMaxWin);
  const int tiles_x = (ow + kSsimTile - 1) / kSsimTile, tiles_y = (oh + kSsimTile - 1) / kSsimTile;
  const int ce = y_only ? 1 : c;
  SRGANFD_LAUNCH(ssim_partial_kernel, dim3(tiles_x * tiles_y, ce, n), dim3(256), 0, s, a, b, c, h, w, crop_border, y_only, window, ws, tiles_x, wsp);
  SRGANFD_LAUNCH(ssim_finish_kernel, dim3(n), dim3(256), 0, s, (const double*)wsp, ce * tiles_x * tiles_y, (double)ce * oh * ow, out);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

int l1_grad_views_impl(srganfd_view a, srganfd_view b, srganfd_view out, int dtype, size_t npix, int c, const float* upstream, float scale, hipStream_t s) {
  if (!a.ptr || !b.ptr || !out.ptr || npix == 0 || c <= 0) return set_err(SRGANFD_EINVAL, "l1_grad_views: bad args");
  DISPATCH_T(dtype,
             SRGANFD_LAUNCH(l1_grad_views_kernel<TT>, dim3(grid_for(npix * c)), dim3(256), 0, s, a.ptr, a.cstride, a.c0, b.ptr, b.cstride, b.c0, out.ptr, out.cstride, out.c0, npix, c, upstream, scale));
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int maxpool2_relu_bwd_impl(srganfd_view x, srganfd_view dy, srganfd_view dx, int dtype, int n, int h, int w, int c, hipStream_t s) {
  if (!x.ptr || !dy.ptr || !dx.ptr || n <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1) || c <= 0) return set_err(SRGANFD_EINVAL, "maxpool2_relu_bwd: bad args");
  const size_t total = (size_t)n * (h / 2) * (w / 2) * c;
  DISPATCH_T(dtype,
             SRGANFD_LAUNCH(maxpool2_relu_bwd_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, s, x.ptr, x.cstride, x.c0, dy.ptr, dy.cstride, dy.c0, dx.ptr, dx.cstride, dx.c0, n, h, w, c));
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int nhwc_to_nchw_scaled_impl(srganfd_view src, int n, int c, int h, int w, float* dst, const float* ch_div, hipStream_t s) {
  if (!src.ptr || !dst || !ch_div || n <= 0 || c <= 0) return set_err(SRGANFD_EINVAL, "nhwc_to_nchw_scaled: bad args");
  SRGANFD_LAUNCH(nhwc_to_nchw_scaled_kernel, dim3(grid_for((size_t)n * c * h * w)), dim3(256), 0, s, (const float*)src.ptr, src.cstride, src.c0, dst, n, c, h * w, ch_div);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

}  // namespace srganfd
