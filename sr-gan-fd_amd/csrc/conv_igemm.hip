// conv_igemm.hip -- fused NHWC convolution as an implicit GEMM on the CDNA4 matrix cores.
//
// Replaces every nn.Conv2d of the reference's hot path (BSRGAN/model.py:42-46 dense-block convs,
// :102-135 discriminator, :325-355 generator head/tail) together with the element-wise work the
// reference runs as separate ATen ops around it: bias, LeakyReLU (:48), `mul 0.2 + identity`
// (:59-60, :85-86), torch.cat (:55-58 -- the output goes straight into a channel slice of the
// dense-block buffer), nearest x2 upsample (:372-374 -- folded into the gather).  With weights
// packed in data-gradient orientation the same kernel is the dgrad pass (mask = LeakyReLU').
//
// Mapping (MI355X-first, not a cuDNN tiling):
//   * one workgroup = 4 wavefronts (64 lanes) = a (4*MR) x 32 output-pixel tile x (32*NR) channels;
//     wave w owns MR image rows, each row of 32 pixels is the M side of a 32x32 MFMA tile.
//   * K = taps x input channels is walked in chunks of 32 channels: the (rows+halo) x (32+halo)
//     x 32ch input patch and the 9 x 32 x (32*NR) weight slab are staged in LDS once per chunk and
//     every one of the 9 taps re-reads the SAME patch at a shifted address (implicit im2col).
//   * A fragments: ds_read_b128 (bf16) of 8 consecutive channels of one pixel; the 16-byte chunk
//     index is XOR-swizzled with (pixel>>2)&3 so the 16-lane read groups of ds_read_b128 are
//     bank-conflict free.  B fragments are pre-packed in lane order -> linear ds_read_b128.
//   * global -> register -> LDS staging with the next chunk's loads issued before the MFMA phase
//     (issue-early / write-late), so HBM/L2 latency hides under the matrix work.
//   * bf16: v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  f32: v_mfma_f32_32x32x2_f32, an exact
//     fp32 fma chain, used as the parity mode against the CPU oracle.
#include "common.hpp"

namespace srganfd {

struct ConvK {
  const void* x; void* y; void* y2; const void* r1; const void* r2; const void* mask; const void* w;
  const float* bias; const float* alpha_dev;
  int xC, x_c0, yC, y_c0, y2C, y2_c0, r1C, r1_c0, r2C, r2_c0, mC, m_c0;
  int N, Hin, Win, up, pad_y, pad_x, Hout, Wout;
  int osy, osx, ooy, oox, HoutF, WoutF;  // output pixel (oy,ox) is stored at (oy*osy+ooy, ox*osx+oox) of a HoutF x WoutF image
  int nChunks;        // cin / 32
  int nNb;            // cout / (32*NR)
  int cout_store;
  int tiles_x, tiles_y;
  float alpha, slope, post_scale, r1s, r2s, mask_slope;
  int act, y_f32, fast_epi;
};

template <typename T> struct FragAB;
template <> struct FragAB<bf16_t> { typedef bf16x8 type; };
template <> struct FragAB<float> { typedef float type; };

template <typename T> __device__ __forceinline__ f32x16 mfma32(typename FragAB<T>::type a, typename FragAB<T>::type b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mfma32<bf16_t>(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma32<float>(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <typename T, int KS, int STRIDE, int MR, int NR>
struct ConvCfg {
  static constexpr int KT = KS * KS;
  static constexpr int TW = 32, TH = 4 * MR;
  static constexpr int PR = (TH - 1) * STRIDE + KS;
  static constexpr int PC = (TW - 1) * STRIDE + KS;
  static constexpr int KC = 32;
  static constexpr int E16 = 16 / (int)sizeof(T);            // elements per 16 bytes
  static constexpr int CPP = KC / E16;                        // 16-byte chunks per pixel
  static constexpr int PIXB = KC * (int)sizeof(T);            // bytes per pixel in LDS
  static constexpr int XBYTES = PR * PC * PIXB;
  static constexpr int KSTEPS = KC / Elem<T>::kStep;
  static constexpr int FRAGB = (int)sizeof(typename FragAB<T>::type);
  static constexpr int WN_BYTES = KT * KSTEPS * 64 * FRAGB;  // one 32-channel n-tile, one chunk
  static constexpr int NX = PR * PC * CPP;
  static constexpr int XI = (NX + 255) / 256;
  static constexpr int NW = NR * WN_BYTES / 16;
  static constexpr int WI = (NW + 255) / 256;
  static constexpr int STAGE_BYTES = XBYTES + NR * WN_BYTES;
  static constexpr int NB = 32 * NR;                           // output channels per workgroup
  static constexpr int EPI_BYTES = TH * TW * NB * 4;           // fp32 tile for the vectorised epilogue
  static constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  static constexpr int NROWS = (MR - 1) * STRIDE + KS;         // patch rows one wave touches
};

template <typename T> __device__ __forceinline__ int lds_x_chunk_off(int pix, int c16);
// bf16: 4 chunks of 16 B per pixel; chunk index XOR (pix>>2)&3 (see header)
template <> __device__ __forceinline__ int lds_x_chunk_off<bf16_t>(int pix, int c16) {
  return pix * 64 + ((c16 ^ ((pix >> 2) & 3)) << 4);
}
// f32: dword index XOR (pix & 31): 32 lanes reading one channel of 32 consecutive pixels hit 32 banks
__device__ __forceinline__ int lds_x_f32_off(int pix, int k) { return pix * 128 + ((k ^ (pix & 31)) << 2); }

template <typename T, int KS, int STRIDE, int MR, int NR>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1)) void conv_igemm_kernel(const ConvK a) {
  using C = ConvCfg<T, KS, STRIDE, MR, NR>;
  using Frag = typename FragAB<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsX = smem;
  char* ldsW = smem + C::XBYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  // XCD-aware bijective remap: blocks b and b+8 share an XCD (L2), give each XCD a contiguous range
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  }
  const int nb = bid % a.nNb;
  int t = bid / a.nNb;
  const int tx = t % a.tiles_x; t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int n = t / a.tiles_y;
  const int oy0 = ty * C::TH, ox0 = tx * C::TW;

  const int Hl = a.Hin << a.up, Wl = a.Win << a.up;
  const T* __restrict__ xg = (const T*)a.x;

  // per-thread source offsets (elements) of the X staging items; -1 = zero padding
  int xoff[C::XI];
#pragma unroll
  for (int i = 0; i < C::XI; ++i) {
    const int item = tid + i * 256;
    const int pix = item / C::CPP, c16 = item % C::CPP;
    const int py = pix / C::PC, px = pix % C::PC;
    const int gy = oy0 * STRIDE - a.pad_y + py, gx = ox0 * STRIDE - a.pad_x + px;
    const bool ok = item < C::NX && gy >= 0 && gy < Hl && gx >= 0 && gx < Wl;
    xoff[i] = ok ? ((n * a.Hin + (gy >> a.up)) * a.Win + (gx >> a.up)) * a.xC + a.x_c0 + c16 * C::E16 : -1;
  }
  const u32x4* __restrict__ wg = (const u32x4*)a.w;

  // bf16: register prefetch of the next chunk (issue-early / write-late).  f32 (parity mode) stages
  // synchronously: its 2x larger tiles would not fit the register budget next to the accumulators.
  constexpr bool kPrefetch = sizeof(T) == 2;
  constexpr int XR = kPrefetch ? C::XI : 1, WR = kPrefetch ? C::WI : 1;
  u32x4 xr[XR];
  u32x4 wr[WR];
  auto load_x = [&](int i, int chunk) -> u32x4 {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (xoff[i] >= 0) v = *(const u32x4*)(xg + xoff[i] + chunk * C::KC);
    return v;
  };
  auto load_w = [&](int i, int chunk) -> u32x4 {
    const int item = tid + i * 256;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (item < C::NW) {
      const int nn = item / (C::WN_BYTES / 16), rem = item % (C::WN_BYTES / 16);
      const size_t src16 = ((size_t)(nb * NR + nn) * a.nChunks + chunk) * (C::WN_BYTES / 16) + rem;
      v = wg[src16];
    }
    return v;
  };
  auto store_x = [&](int i, u32x4 v) {
    const int item = tid + i * 256;
    if (item < C::NX) {
      const int pix = item / C::CPP, c16 = item % C::CPP;
      if constexpr (sizeof(T) == 2) {
        *(u32x4*)(ldsX + lds_x_chunk_off<bf16_t>(pix, c16)) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) *(unsigned int*)(ldsX + lds_x_f32_off(pix, c16 * 4 + e)) = v[e];
      }
    }
  };
  auto store_w = [&](int i, u32x4 v) {
    const int item = tid + i * 256;
    if (item < C::NW) *(u32x4*)(ldsW + item * 16) = v;
  };
  auto prefetch = [&](int chunk) {
    if constexpr (kPrefetch) {
#pragma unroll
      for (int i = 0; i < C::XI; ++i) xr[i] = load_x(i, chunk);
#pragma unroll
      for (int i = 0; i < C::WI; ++i) wr[i] = load_w(i, chunk);
    }
  };
  auto commit = [&](int chunk) {
    if constexpr (kPrefetch) {
#pragma unroll
      for (int i = 0; i < C::XI; ++i) store_x(i, xr[i]);
#pragma unroll
      for (int i = 0; i < C::WI; ++i) store_w(i, wr[i]);
    } else {
#pragma unroll 4
      for (int i = 0; i < C::XI; ++i) store_x(i, load_x(i, chunk));
#pragma unroll 4
      for (int i = 0; i < C::WI; ++i) store_w(i, load_w(i, chunk));
    }
  };

  f32x16 acc[MR][NR];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int nn = 0; nn < NR; ++nn)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][nn][i] = 0.f;

  prefetch(0);
  for (int chunk = 0; chunk < a.nChunks; ++chunk) {
    __syncthreads();
    commit(chunk);
    __syncthreads();
    if (chunk + 1 < a.nChunks) prefetch(chunk + 1);

    // For a fixed kernel column kx and k-step s, the MR output rows of this wave and the KS kernel rows
    // touch only NROWS = (MR-1)*S + KS patch rows: read each A fragment ONCE and reuse it for every
    // (output row, kernel row) pair that needs it (6 LDS reads instead of 12 per 12 MFMAs at MR=4, KS=3).
    auto col_body = [&](int kx, int s) {
      Frag av[C::NROWS];
#pragma unroll
      for (int rr = 0; rr < C::NROWS; ++rr) {
        const int pix = (wave * MR * STRIDE + rr) * C::PC + r * STRIDE + kx;
        if constexpr (sizeof(T) == 2) av[rr] = *(const Frag*)(ldsX + lds_x_chunk_off<bf16_t>(pix, 2 * s + h));
        else av[rr] = *(const Frag*)(ldsX + lds_x_f32_off(pix, 2 * s + h));
      }
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
        Frag bq[NR];
#pragma unroll
        for (int nn = 0; nn < NR; ++nn)
          bq[nn] = *(const Frag*)(ldsW + ((nn * C::KT + ky * KS + kx) * C::KSTEPS + s) * 64 * C::FRAGB + lane * C::FRAGB);
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int nn = 0; nn < NR; ++nn) acc[m][nn] = mfma32<T>(av[m * STRIDE + ky], bq[nn], acc[m][nn]);
      }
    };
    if constexpr (sizeof(T) == 2) {
#pragma unroll 1
      for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
        for (int s2 = 0; s2 < C::KSTEPS; ++s2) col_body(kx, s2);
      }
    } else {
#pragma unroll 1
      for (int kx = 0; kx < KS; ++kx) {
#pragma unroll 2
        for (int s2 = 0; s2 < C::KSTEPS; ++s2) col_body(kx, s2);
      }
    }
  }

  // ---- epilogue (see srganfd.h for the formula) ----
  float alpha = a.alpha;
  if (a.alpha_dev) alpha *= *a.alpha_dev;
  if (a.fast_epi) {
    // Vectorised epilogue: (1) every lane applies the per-channel part (alpha, bias, activation, scale) to
    // its accumulators and drops them as fp32 into an LDS tile [pixel][channel]; (2) the workgroup
    // re-reads the tile 16 output bytes per lane, adds residuals / applies the LeakyReLU' mask with
    // 16-byte global loads, and issues 16-byte stores (one pixel's channels are contiguous in NHWC).
    float* tile = (float*)smem;
    __syncthreads();   // all waves are done with the staging buffers
#pragma unroll
    for (int nn = 0; nn < NR; ++nn) {
      const int co = (nb * NR + nn) * 32 + r;
      const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float v = alpha * acc[m][nn][i] + bv;
          if (a.act == SRGANFD_ACT_LRELU) v = v > 0.f ? v : v * a.slope;
          else if (a.act == SRGANFD_ACT_RELU) v = fmaxf(v, 0.f);
          tile[((wave * MR + m) * 32 + mfma32_row(i, lane)) * C::NB + nn * 32 + r] = v * a.post_scale;
        }
    }
    __syncthreads();
    constexpr int CP = C::NB / C::E16;                 // 16-byte output chunks per pixel
    constexpr int ITEMS = C::TH * 32 * CP;
    const int cbase = nb * C::NB;
#pragma unroll 2
    for (int item = tid; item < ITEMS; item += 256) {
      const int pix = item / CP, ck = item % CP;
      const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
      if (oy < a.Hout && ox < a.Wout) {
        const size_t p = ((size_t)n * a.HoutF + oy * a.osy + a.ooy) * a.WoutF + ox * a.osx + a.oox;
        float v[C::E16];
        const f32x4* tp = (const f32x4*)(tile + pix * C::NB + ck * C::E16);
#pragma unroll
        for (int q = 0; q < C::E16 / 4; ++q) {
          const f32x4 t4 = tp[q];
          v[4 * q] = t4[0]; v[4 * q + 1] = t4[1]; v[4 * q + 2] = t4[2]; v[4 * q + 3] = t4[3];
        }
        const int cch = cbase + ck * C::E16;
        auto load16 = [&](const void* base, int Cs, int c0, float* out) {
          const T* src = (const T*)base + p * Cs + c0 + cch;
          if constexpr (sizeof(T) == 2) {
            const u32x4 raw = *(const u32x4*)src;
            const unsigned w0 = raw[0], w1 = raw[1], w2 = raw[2], w3 = raw[3];
            out[0] = __uint_as_float(w0 << 16); out[1] = __uint_as_float(w0 & 0xffff0000u);
            out[2] = __uint_as_float(w1 << 16); out[3] = __uint_as_float(w1 & 0xffff0000u);
            out[4] = __uint_as_float(w2 << 16); out[5] = __uint_as_float(w2 & 0xffff0000u);
            out[6] = __uint_as_float(w3 << 16); out[7] = __uint_as_float(w3 & 0xffff0000u);
          } else {
            const f32x4 rf = *(const f32x4*)src;
            out[0] = rf[0]; out[1] = rf[1]; out[2] = rf[2]; out[3] = rf[3];
          }
        };
        auto store16 = [&](void* base, int Cs, int c0, const float* vv) {
          T* dstp = (T*)base + p * Cs + c0 + cch;
          if constexpr (sizeof(T) == 2) {
            u32x4 o;
            o[0] = (unsigned)f2bf(vv[0]) | ((unsigned)f2bf(vv[1]) << 16);
            o[1] = (unsigned)f2bf(vv[2]) | ((unsigned)f2bf(vv[3]) << 16);
            o[2] = (unsigned)f2bf(vv[4]) | ((unsigned)f2bf(vv[5]) << 16);
            o[3] = (unsigned)f2bf(vv[6]) | ((unsigned)f2bf(vv[7]) << 16);
            *(u32x4*)dstp = o;
          } else {
            f32x4 o = {vv[0], vv[1], vv[2], vv[3]};
            *(f32x4*)dstp = o;
          }
        };
        if (a.y2) store16(a.y2, a.y2C, a.y2_c0, v);   // activation before the skip add (exact LeakyReLU' sign for backward)
        float t[C::E16];
        if (a.r1) { load16(a.r1, a.r1C, a.r1_c0, t);
#pragma unroll
          for (int q = 0; q < C::E16; ++q) v[q] += a.r1s * t[q]; }
        if (a.r2) { load16(a.r2, a.r2C, a.r2_c0, t);
#pragma unroll
          for (int q = 0; q < C::E16; ++q) v[q] += a.r2s * t[q]; }
        if (a.mask) { load16(a.mask, a.mC, a.m_c0, t);
#pragma unroll
          for (int q = 0; q < C::E16; ++q) v[q] *= t[q] > 0.f ? 1.f : a.mask_slope; }
        store16(a.y, a.yC, a.y_c0, v);
      }
    }
    return;
  }
  // generic epilogue (padded channel counts, fp32 output): scalar stores
  T* __restrict__ yg = (T*)a.y;
  const T* __restrict__ r1g = (const T*)a.r1;
  const T* __restrict__ r2g = (const T*)a.r2;
  const T* __restrict__ mg = (const T*)a.mask;
#pragma unroll
  for (int nn = 0; nn < NR; ++nn) {
    const int co = (nb * NR + nn) * 32 + r;
    const bool cok = co < a.cout_store;
    const float bv = (a.bias && cok) ? a.bias[co] : 0.f;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int oy = oy0 + wave * MR + m;
      if (!cok || oy >= a.Hout) continue;
      const size_t prow = ((size_t)n * a.HoutF + oy * a.osy + a.ooy) * a.WoutF + a.oox;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ox = ox0 + mfma32_row(i, lane);
        if (ox < a.Wout) {
          const size_t p = prow + ox * a.osx;
          float v = alpha * acc[m][nn][i] + bv;
          if (a.act == SRGANFD_ACT_LRELU) v = v > 0.f ? v : v * a.slope;
          else if (a.act == SRGANFD_ACT_RELU) v = v > 0.f ? v : 0.f;
          v *= a.post_scale;
          if (a.y2) ((T*)a.y2)[p * a.y2C + a.y2_c0 + co] = Elem<T>::from_f(v);
          if (r1g) v += a.r1s * Elem<T>::to_f(r1g[p * a.r1C + a.r1_c0 + co]);
          if (r2g) v += a.r2s * Elem<T>::to_f(r2g[p * a.r2C + a.r2_c0 + co]);
          if (mg) v *= (Elem<T>::to_f(mg[p * a.mC + a.m_c0 + co]) > 0.f) ? 1.f : a.mask_slope;
          if (a.y_f32) ((float*)a.y)[p * a.yC + a.y_c0 + co] = v;
          else yg[p * a.yC + a.y_c0 + co] = Elem<T>::from_f(v);
        }
      }
    }
  }
}

template <typename T, int KS, int STRIDE, int MR, int NR>
static int launch_conv(const ConvK& k, int cout, hipStream_t stream) {
  using C = ConvCfg<T, KS, STRIDE, MR, NR>;
  auto kern = conv_igemm_kernel<T, KS, STRIDE, MR, NR>;
  static bool attr_done = false;
  if (!attr_done) {
    SRGANFD_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    attr_done = true;
  }
  ConvK kk = k;
  kk.nNb = cout / (32 * NR);
  kk.tiles_x = ceil_div(k.Wout, C::TW);
  kk.tiles_y = ceil_div(k.Hout, C::TH);
  const long long nblk = (long long)k.N * kk.tiles_x * kk.tiles_y * kk.nNb;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return set_err(SRGANFD_EINVAL, "conv2d: bad grid %lld", nblk);
  SRGANFD_LAUNCH(kern, dim3((unsigned)nblk), dim3(256), C::LDS_BYTES, stream, kk);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

template <typename T>
static int dispatch_conv(const srganfd_conv_args* a, const ConvK& k, hipStream_t s) {
  const bool wide = (a->cout % 64) == 0;
  if (a->ksize == 3 && a->stride == 1) {
    if (wide) return launch_conv<T, 3, 1, 2, 2>(k, a->cout, s);
    return launch_conv<T, 3, 1, 4, 1>(k, a->cout, s);
  }
  if (a->ksize == 4 && a->stride == 2) return launch_conv<T, 4, 2, 1, 1>(k, a->cout, s);
  if (a->ksize == 2 && a->stride == 1) return wide ? launch_conv<T, 2, 1, 2, 2>(k, a->cout, s) : launch_conv<T, 2, 1, 4, 1>(k, a->cout, s);
  if (a->ksize == 1 && a->stride == 1) return launch_conv<T, 1, 1, 2, 1>(k, a->cout, s);
  return set_err(SRGANFD_EINVAL, "conv2d: unsupported ksize=%d stride=%d", a->ksize, a->stride);
}

int conv2d_impl(const srganfd_conv_args* a, hipStream_t stream) {
  if (!a || !a->x.ptr || !a->y.ptr || !a->w_packed) return set_err(SRGANFD_EINVAL, "conv2d: null pointer");
  if (a->cin <= 0 || a->cin % 32 || a->cout <= 0 || a->cout % 32 || a->cout_store <= 0 || a->cout_store > a->cout)
    return set_err(SRGANFD_EINVAL, "conv2d: cin=%d cout=%d cout_store=%d (need multiples of 32)", a->cin, a->cout, a->cout_store);
  if (a->n <= 0 || a->h_in <= 0 || a->w_in <= 0 || a->h_out <= 0 || a->w_out <= 0)
    return set_err(SRGANFD_EINVAL, "conv2d: bad dims");
  const int hl = a->h_in << (a->up ? 1 : 0), wl = a->w_in << (a->up ? 1 : 0);
  const bool sub = a->out_sy > 1 || a->out_sx > 1;  // parity-class launch of a stride-2 transposed conv
  if (!sub) {
    const int ho = (hl + 2 * a->pad - a->ksize) / a->stride + 1, wo = (wl + 2 * a->pad - a->ksize) / a->stride + 1;
    if (ho != a->h_out || wo != a->w_out)
      return set_err(SRGANFD_EINVAL, "conv2d: h_out/w_out %dx%d inconsistent with input (expect %dx%d)", a->h_out, a->w_out, ho, wo);
  } else if (a->out_h_full < (a->h_out - 1) * a->out_sy + a->out_oy + 1 || a->out_w_full < (a->w_out - 1) * a->out_sx + a->out_ox + 1) {
    return set_err(SRGANFD_EINVAL, "conv2d: strided output does not fit the full image");
  }
  const int align = a->dtype == SRGANFD_BF16 ? 8 : 4;
  if (a->x.cstride % align || a->x.c0 % align) return set_err(SRGANFD_EINVAL, "conv2d: x view not 16-byte aligned");
  if (a->x.c0 + a->cin > a->x.cstride) return set_err(SRGANFD_EINVAL, "conv2d: x view exceeds buffer channels");
  if (a->y.c0 + a->cout_store > a->y.cstride) return set_err(SRGANFD_EINVAL, "conv2d: y view exceeds buffer channels");
  if ((size_t)a->n * hl * wl * (size_t)a->x.cstride >= 0x7fffffffULL)
    return set_err(SRGANFD_EINVAL, "conv2d: input too large for 32-bit element offsets");
  ConvK k;
  k.x = a->x.ptr; k.y = a->y.ptr; k.y2 = a->y2.ptr; k.y2C = a->y2.cstride; k.y2_c0 = a->y2.c0; k.r1 = a->r1.ptr; k.r2 = a->r2.ptr; k.mask = a->mask.ptr; k.w = a->w_packed;
  k.bias = a->bias; k.alpha_dev = a->alpha_dev;
  k.xC = a->x.cstride; k.x_c0 = a->x.c0; k.yC = a->y.cstride; k.y_c0 = a->y.c0;
  k.r1C = a->r1.cstride; k.r1_c0 = a->r1.c0; k.r2C = a->r2.cstride; k.r2_c0 = a->r2.c0;
  k.mC = a->mask.cstride; k.m_c0 = a->mask.c0;
  k.N = a->n; k.Hin = a->h_in; k.Win = a->w_in; k.up = a->up ? 1 : 0; k.pad_y = sub ? a->pad_y : a->pad; k.pad_x = sub ? a->pad_x : a->pad;
  k.osy = sub ? a->out_sy : 1; k.osx = sub ? a->out_sx : 1; k.ooy = sub ? a->out_oy : 0; k.oox = sub ? a->out_ox : 0;
  k.HoutF = sub ? a->out_h_full : a->h_out; k.WoutF = sub ? a->out_w_full : a->w_out;
  k.Hout = a->h_out; k.Wout = a->w_out; k.nChunks = a->cin / 32; k.nNb = 0; k.cout_store = a->cout_store;
  k.tiles_x = k.tiles_y = 0;
  k.alpha = a->alpha; k.slope = a->slope; k.post_scale = a->post_scale; k.r1s = a->r1_scale; k.r2s = a->r2_scale;
  k.mask_slope = a->mask_slope; k.act = a->act; k.y_f32 = a->y_f32 ? 1 : 0;
  auto aligned = [&](const srganfd_view& v) { return !v.ptr || (v.cstride % align == 0 && v.c0 % align == 0 && ((uintptr_t)v.ptr & 15) == 0); };
  k.fast_epi = (!a->y_f32 && a->cout_store == a->cout && aligned(a->y) && aligned(a->y2) && aligned(a->r1) && aligned(a->r2) && aligned(a->mask)) ? 1 : 0;
  if (a->dtype == SRGANFD_BF16) return dispatch_conv<bf16_t>(a, k, stream);
  if (a->dtype == SRGANFD_F32) return dispatch_conv<float>(a, k, stream);
  return set_err(SRGANFD_EINVAL, "conv2d: bad dtype %d", a->dtype);
}

}  // namespace srganfd
