// conv_thin.hip -- the 3x3 stride-1 convolutions of the hot path that have a THIN side: 1..4 channels against 64.
//
// Reference layers: generator conv1 3->64 / conv4 64->3 (BSRGAN/model.py:325,355), discriminator conv1 3->64 / conv4 64->1
// (:102,135), VGG-19 features.0 3->64 (ContentLoss, :522-524), A-ESRGAN conv0 / conv9 (A-ESRGAN/model.py:287,307), ESRGAN's
// discriminator features.0 (ESRGAN/model.py:92), and the data / weight gradients of all of them.  On conv_igemm's 32-channel chunks
// these layers ran on tensors that are 29/32 or 31/32 zeros (8 % of a GAN step at 17-60 TFLOP/s of useful work).  They are
// HBM-streaming problems -- 128 bytes per pixel on the 64-channel side, 8-16 on the thin side, 1.7 kFLOP per pixel -- so the kernels
// here are built around the stream, not around a GEMM tile:
//
//   * thin_in  (1..4 -> 64 channels; also the data gradient of a 64 -> 1..4 conv, with the LeakyReLU' mask): all 9 taps x 4 channels
//     are ONE K: k = 4 * tap + c, 36 products = two v_mfma_f32_16x16x32 steps (taps 0-7, tap 8) per 16 pixels x
//     16 output channels instead of 9 taps x 32 padded channels.  Operands are swapped (A = weights, B = pixels) so that a lane ends
//     up with 8 CONSECUTIVE output channels of one pixel: bias (as the accumulator's initial value), activation, mask and the 16-byte
//     store all happen in registers.  No LDS, no barrier: a wave owns a 16-column strip and walks down the image; its B fragments are
//     8-byte loads of the thin tensor (L1 / L2 hits: every thin pixel is needed by 9 taps), its weights live in 24 registers.
//   * thin_out (64 -> 1..4 channels; also the data gradient of a 1..4 -> 64 conv): the three kernel COLUMNS are folded into the MFMA's
//     M side: row (kx, co) of D' = sum over (ky, ci) of W[co][ky][kx][ci] * X[y + ky][x'][ci] for 16 patch columns x' -- K = 3 x 64 =
//     192 = six 16x16x32 MFMAs per 14 output pixels instead of eighteen on 31/32-zero columns -- and the output is
//     out[x] = D'[kx=0][x] + D'[1][x+1] + D'[2][x+2], three ds_bpermute'd registers.  B fragments are 16-byte global loads of the
//     64-channel tensor (read exactly once per strip row, kept in registers for the three output rows that use them), ring-buffered
//     four rows ahead.  No LDS, no barrier.
//   * thin_wgrad: dW[b][tap][s] = sum_p BIG[p][b] * THIN[p + tap][s] as a 64 x 48 output (M = the 64 channels, N = 4 * tap + s, plus
//     a column of ones that yields the bias gradient), K = pixels.  Both operands need K (pixels) along a lane's register, i.e. a
//     transpose of NHWC data: tiles go through a wave-private LDS image and come back through ds_read_b64_tr_b16 (for the thin
//     operand the transposing read's per-lane row addresses ARE the tap shifts).  Persistent waves, fp32 slabs, deterministic
//     two-stage reduction (no float atomics).
//
// All three take the RAW fp32 weight tensor (Cout, Cin, 3, 3) -- the fragments are built from it in the prologue -- so these layers
// have no packed copies to keep fresh.  16-bit (f16 / bf16) only: the exact-fp32 parity mode stays on conv_igemm's f32 kernels.
#include "conv_common.hpp"

namespace srganfd {

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2v;
typedef __attribute__((address_space(3))) s16x4 lds_tr_t;

struct ThinK {
  const void* thin;         // NHWC4 16-bit tensor (thin_in input / thin_wgrad small operand)
  float* thin_f32;          // thin_out output (fp32)
  int thin_pitch;           // its pixel pitch in floats (4, or 1 when cs == 1)
  const void* big; const void* mask;
  int bigC, big_c0, big_ps, big_gs;      // element (pixel p, channel c) at p * ps + (c >> 5) * gs + (c & 31) inside an image of H*W*bigC elements
  int mC, m_c0, m_ps, m_gs;
  const float* w; const float* bias;
  int cs, big_is_cout, flip;
  int N, H, W;
  float neg, mask_slope;    // activation as one select: v * (v > 0 ? 1 : neg)
  int nt;                   // non-temporal output stores (large outputs)
  int tiles_x, tiles_y;
};

constexpr int kThinOob = 0x7fffffff;

template <typename T> __device__ __forceinline__ unsigned short to_bits16(float f) {
  if constexpr (Elem<T>::kDtype == SRGANFD_F16) return __builtin_bit_cast(unsigned short, (_Float16)f);
  else return f2bf(f);
}

// raw weight of (64-channel side b, thin side s, tap t), zero outside the thin side's channels
__device__ __forceinline__ float thin_weight(const float* __restrict__ wl, int big_is_cout, int cs, int flip, int b, int s, int t) {
  if (s >= cs || t > 8) return 0.f;
  const int tt = flip ? 8 - t : t;
  return big_is_cout ? wl[(b * cs + s) * 9 + tt] : wl[(s * 64 + b) * 9 + tt];
}

__device__ __forceinline__ void* thin_uniform_ptr(const void* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return (void*)(((unsigned long long)hi << 32) | lo);
}

// ------------------------------------------------------------------------------------------------------------------------------
// thin_in: y[p][b] = act(bias[b] + sum_{t,s} w(b,s,t) * x[p + t - 1][s]) (* LeakyReLU'(mask[p][b])), b = 0..63
// workgroup = 4 waves = 64 columns x TH rows; wave = 16 columns
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int kThinInRows = 32;

// The 16 x 64 output tile of a wave row goes through a wave-private 2 KiB LDS image and leaves as two stores of 8 WHOLE pixels (1 KiB
// contiguous each); straight from the accumulator layout a store instruction covers 16 half pixels (64 of a pixel's 128 bytes), which
// measured 371 vs 326 us (3 -> 64 at 512 x 512, batch 32) and 472 vs 420 us with the mask, which is read in the store layout too.
template <typename T, bool MASK, bool NT>
__global__ __launch_bounds__(256) void thin_in_kernel(const ThinK a) {
  using Frag = typename FragAB<T>::type;
  __shared__ float wl[64 * 4 * 9];
  __shared__ __attribute__((aligned(16))) char tl_all[4 * 2048];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const int nw = 64 * a.cs * 9;
    for (int i = tid; i < nw; i += 256) wl[i] = a.w[i];
  }
  __syncthreads();
  const int n16 = lane & 15, kg = lane >> 4;
  // A fragments: D row m = lane & 15 of tile t is output channel 32 * (t >> 1) + 8 * (m >> 2) + 4 * (t & 1) + (m & 3), so that after
  // the MFMA lane (pixel, kg) holds channels 32 u + 8 kg + [0, 8) in tiles 2u, 2u + 1
  Frag wa[4], wb[4];
  f32x4_t bv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int b = 32 * (t >> 1) + 8 * (n16 >> 2) + 4 * (t & 1) + (n16 & 3);
    unsigned short e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = to_bits16<T>(thin_weight(wl, a.big_is_cout, a.cs, a.flip, b, j & 3, 2 * kg + (j >> 2)));
    const u32x4 pk = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16),
                      (unsigned)e[6] | ((unsigned)e[7] << 16)};
    wa[t] = __builtin_bit_cast(Frag, pk);
    // tap 8 rides in a second 16x16x32 step (k = 0..3 of lane group 0; the other 28 products are zeros).  NOT a 16x16x16 MFMA: hipcc
    // (ROCm 7.2) schedules a v_mfma_f32_16x16x16 three instructions behind the v_mfma_f32_16x16x32 whose accumulator it continues, and on
    // gfx950 that pair loses registers 0-1 of the first result (seen as outputs that equal the bias); two MFMAs of one form chain fine
    unsigned short e8[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) e8[j] = to_bits16<T>(kg == 0 ? thin_weight(wl, a.big_is_cout, a.cs, a.flip, b, j, 8) : 0.f);
    const u32x4 pk8 = {(unsigned)e8[0] | ((unsigned)e8[1] << 16), (unsigned)e8[2] | ((unsigned)e8[3] << 16), 0u, 0u};
    wb[t] = __builtin_bit_cast(Frag, pk8);
    // accumulator rows of this lane: channels 32 (t >> 1) + 8 kg + 4 (t & 1) + r
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[t][r] = a.bias ? a.bias[32 * (t >> 1) + 8 * kg + 4 * (t & 1) + r] : 0.f;
  }
  // tile of this workgroup
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int img = bid / a.tiles_y;
  const int ox = tx * 64 + wave * 16 + n16;
  const int y0 = ty * kThinInRows, y1 = min(y0 + kThinInRows, a.H);
  if (tx * 64 + wave * 16 >= a.W) return;              // whole wave outside the image (ragged width)
  // thin operand: 8-byte loads through a buffer descriptor of this image (padding = offset beyond the range -> zeros)
  const T* timg = (const T*)a.thin + (size_t)img * a.H * a.W * 4;
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(thin_uniform_ptr(timg), (short)0, (int)((unsigned)a.H * (unsigned)a.W * 8u), 0x00020000);
  // taps of this lane: K32 fragment taps 2 kg, 2 kg + 1; K16 fragment tap 8
  const int tA = 2 * kg, tB = 2 * kg + 1;
  const int dyA = tA / 3 - 1, dxA = tA % 3 - 1, dyB = tB / 3 - 1, dxB = tB % 3 - 1;
  const int xA = ox + dxA, xB = ox + dxB, xC = ox + 1;
  const bool okA = xA >= 0 && xA < a.W, okB = xB >= 0 && xB < a.W, okC = xC < a.W;
  auto ldx = [&](int gy, int gx, bool okx) -> u32x2v {
    typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned v2u;
    const bool ok = okx && gy >= 0 && gy < a.H;
    const v2u r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, ok ? (gy * a.W + gx) * 8 : kThinOob, 0, 0);
    return __builtin_bit_cast(u32x2v, r);
  };
  const T* yimg = (const T*)a.big + (size_t)img * a.H * a.W * a.bigC;
  const T* mimg = MASK ? (const T*)a.mask + (size_t)img * a.H * a.W * a.mC : nullptr;
  // store layout: piece u (0, 1) of a lane is pixel (lane >> 3) + 8 u of the wave's 16, channels c0 + 8 (lane & 7) .. + 7
  char* tl = tl_all + wave * 2048;
  const int sp8 = lane >> 3, ss8 = lane & 7;
  const int ox_t0 = tx * 64 + wave * 16 + sp8;             // image column of piece 0 (piece 1: + 8)
  const int cy = a.big_c0 + 8 * ss8, cm = a.m_c0 + 8 * ss8;
  const int yoff = (cy >> 5) * a.big_gs + (cy & 31), moff = (cm >> 5) * a.m_gs + (cm & 31);
  // rows of loads in flight ahead of the row being computed (4 and 8 measured the same: the kernel is bound by its stores, and 2 leaves
  // registers for four / three waves per SIMD)
  constexpr int D = 2;
  u32x2v fa[D], fb[D], fc[D];
  u32x4 mk[D][2];
  auto issue = [&](int slot, int oy) {
    fa[slot] = ldx(oy + dyA, xA, okA);
    fb[slot] = ldx(oy + dyB, xB, okB);
    fc[slot] = ldx(oy + 1, xC, okC);
    if constexpr (MASK) {
      if (oy < y1) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (ox_t0 + 8 * u < a.W) mk[slot][u] = *(const u32x4*)(mimg + (size_t)(oy * a.W + ox_t0 + 8 * u) * a.m_ps + moff);
      }
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) issue(d, y0 + d);
  for (int oyb = y0; oyb < y1; oyb += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int oy = oyb + d;
      const u32x2v A0 = fa[d], B0 = fb[d], C0 = fc[d];
      u32x4 m0, m1;
      if constexpr (MASK) { m0 = mk[d][0]; m1 = mk[d][1]; }
      issue(d, oy + D);
      if (oy >= y1) continue;
      const u32x4 xk = {A0.x, A0.y, B0.x, B0.y};
      const Frag xf = __builtin_bit_cast(Frag, xk);
      const Frag xs = __builtin_bit_cast(Frag, u32x4{C0.x, C0.y, 0u, 0u});
      f32x4_t acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16<T>(wa[t], xf, bv[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16<T>(wb[t], xs, acc[t]);
      {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float t = acc[2 * u + (q >> 2)][q & 3];
            v[q] = t * (t > 0.f ? 1.f : a.neg);
          }
          // accumulator layout -> LDS image [pixel][8 slots of 16 bytes], slot XOR (pixel & 7): conflict-free 16-byte writes and reads
          *(u32x4*)(tl + n16 * 128 + (((4 * u + kg) ^ (n16 & 7)) << 4)) = pack8<T>(v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int pp = sp8 + 8 * u;
          u32x4 o = *(const u32x4*)(tl + pp * 128 + ((ss8 ^ (pp & 7)) << 4));
          if constexpr (MASK) {
            float v[8], mv[8];
            unpack8<T>(o, v);
            unpack8<T>(u ? m1 : m0, mv);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] *= mv[q] > 0.f ? 1.f : a.mask_slope;
            o = pack8<T>(v);
          }
          if (ox_t0 + 8 * u < a.W) {
            u32x4* dst = (u32x4*)((T*)yimg + (size_t)(oy * a.W + ox_t0 + 8 * u) * a.big_ps + yoff);
            // (NT is a template parameter: behind a run-time `if` the two stores are merged into one plain store)
            if constexpr (NT) __builtin_nontemporal_store(o, dst);
            else *dst = o;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// thin_out: y[p][s] = bias[s] + sum_{t,b} w(b,s,t) * x[p + t - 1][b], s < cs <= 4, b = 0..63
// workgroup = 4 waves = 4 adjacent strips of 14 output columns x TH rows
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int kThinOutRows = 32;

template <typename T>
__global__ __launch_bounds__(256) void thin_out_kernel(const ThinK a) {
  using Frag = typename FragAB<T>::type;
  __shared__ float wl[64 * 4 * 9];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const int nw = 64 * a.cs * 9;
    for (int i = tid; i < nw; i += 256) wl[i] = a.w[i];
  }
  __syncthreads();
  const int n16 = lane & 15, kg = lane >> 4;
  // A fragments [ky][half]: row m = (kx = m >> 2, co = m & 3), k = 8 kg + j -> input channel 32 half + 8 kg + j
  Frag wa[3][2];
  {
    const int kx = n16 >> 2, co = n16 & 3;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = to_bits16<T>(kx < 3 ? thin_weight(wl, a.big_is_cout, a.cs, a.flip, 32 * hf + 8 * kg + j, co, ky * 3 + kx) : 0.f);
        const u32x4 pk = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16),
                          (unsigned)e[6] | ((unsigned)e[7] << 16)};
        wa[ky][hf] = __builtin_bit_cast(Frag, pk);
      }
  }
  float bs[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bs[r] = (a.bias && r < a.cs) ? a.bias[r] : 0.f;
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int img = bid / a.tiles_y;
  const int x0 = (tx * 4 + wave) * 14;                   // first output column of this wave's strip
  if (x0 >= a.W) return;
  const int y0 = ty * kThinOutRows, y1 = min(y0 + kThinOutRows, a.H);
  const int gx = x0 - 1 + n16;                            // image column of this lane's patch column
  const bool okx = gx >= 0 && gx < a.W;
  const T* ximg = (const T*)a.big + (size_t)img * a.H * a.W * a.bigC;
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(thin_uniform_ptr(ximg), (short)0, (int)((unsigned)a.H * (unsigned)a.W * (unsigned)a.bigC * 2u), 0x00020000);
  int coff[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int c = a.big_c0 + 32 * hf + 8 * kg;
    coff[hf] = ((c >> 5) * a.big_gs + (c & 31)) * 2;
  }
  auto ldrow = [&](int gy, u32x4* out) {
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned v4u;
    const bool ok = okx && gy >= 0 && gy < a.H;
    const int base = ok ? (gy * a.W + gx) * a.big_ps * 2 : kThinOob;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const v4u r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? base + coff[hf] : kThinOob, 0, 0);
      out[hf] = __builtin_bit_cast(u32x4, r);
    }
  };
  constexpr int D = 4;
  u32x4 ring[D][2];
  const int pr0 = y0 - 1, pr1 = y1;                        // patch rows pr0 .. pr1 inclusive
#pragma unroll
  for (int d = 0; d < D; ++d) ldrow(pr0 + d, ring[d]);
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  // gather lanes of the kx = 1 / kx = 2 partial sums: output column i (lane i, kg == 0) needs row group 1 of column i + 1 and row group 2 of i + 2
  const int src1 = 4 * min(16 + n16 + 1, 63), src2 = 4 * min(32 + n16 + 2, 63);
  const int ox = x0 + n16;
  const bool st_ok = kg == 0 && n16 < 14 && ox < a.W;
  float* yimg = a.thin_f32 + (size_t)img * a.H * a.W * a.thin_pitch;
  for (int prb = pr0; prb <= pr1; prb += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int pr = prb + d;
      const Frag b0 = __builtin_bit_cast(Frag, ring[d][0]), b1 = __builtin_bit_cast(Frag, ring[d][1]);
      ldrow(pr + D, ring[d]);
      if (pr > pr1) continue;
      // patch row pr: kernel row 2 of output row pr - 1, row 1 of pr, row 0 of pr + 1
      acc0 = mfma16<T>(wa[2][0], b0, acc0);
      acc0 = mfma16<T>(wa[2][1], b1, acc0);
      acc1 = mfma16<T>(wa[1][0], b0, acc1);
      acc1 = mfma16<T>(wa[1][1], b1, acc1);
      f32x4_t acc2 = {0.f, 0.f, 0.f, 0.f};
      acc2 = mfma16<T>(wa[0][0], b0, acc2);
      acc2 = mfma16<T>(wa[0][1], b1, acc2);
      const int oy = pr - 1;
      if (oy >= y0) {            // (oy < y1 holds: pr <= pr1 = y1)
        float o[4];
        const float c0 = acc0[0], c1 = acc0[1], c2 = acc0[2], c3 = acc0[3];
        const float cv[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v1 = __shfl(cv[r], src1 >> 2, 64);
          const float v2 = __shfl(cv[r], src2 >> 2, 64);
          o[r] = (cv[r] + v1) + v2 + bs[r];
        }
        if (st_ok) {
          float* dst = yimg + (size_t)(oy * a.W + ox) * a.thin_pitch;
          if (a.thin_pitch == 4) *(f32x4*)dst = f32x4{o[0], o[1], o[2], o[3]};
          else
#pragma unroll
            for (int r = 0; r < 4; ++r) if (r < a.cs) dst[r] = o[r];
        }
      }
      acc0 = acc1;
      acc1 = acc2;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// thin_wgrad: D[b][4 * tap + s] = sum_p BIG[p][b] * THIN[p + tap - 1][s]  (+ column 36: sum_p BIG[p][b]; + sum_p THIN[p][s])
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int kTwWaves = 8;                       // waves per workgroup
constexpr int kTwBigBytes = 32 * 128;             // BIG tile of one K step: 32 pixels x 64 channels, 16-bit
constexpr int kTwPatchCols = 40;                  // THIN patch row pitch in pixels (34 used)
constexpr int kTwPatchBytes = 3 * kTwPatchCols * 8;
constexpr int kTwWaveBytes = kTwBigBytes + kTwPatchBytes + 32;      // + a {1,0,0,0} cell and a zero cell
constexpr int kTwSlab = 64 * 48 + 16;             // floats per workgroup slab: D (64 x 48) + thin-side channel sums (4) + pad
constexpr int kTwLds = (kTwWaves * kTwWaveBytes > 4 * 64 * 48 * 4 ? kTwWaves * kTwWaveBytes : 4 * 64 * 48 * 4) + 64;

struct ThinWgK {
  const void* big; const void* thin; float* slabs;
  int bigC, big_c0, big_ps, big_gs;
  int N, H, W, cbx;         // cbx = column blocks of 32 per row
  long long nunits;         // N * H * cbx K steps
};

// swizzled byte offset of 8-byte chunk `ch` (0..15: channels 4 ch .. 4 ch + 3) of pixel `pix` (0..31) in the BIG tile
__device__ __forceinline__ int tw_big_off(int pix, int ch) { return pix * 128 + ((ch ^ (4 * (((pix >> 1) & 1) | (((pix >> 3) & 1) << 1)))) << 3); }

template <typename T>
__global__ __launch_bounds__(64 * kTwWaves) void thin_wgrad_kernel(const ThinWgK a) {
  using Frag = typename FragAB<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* lbig = smem + wave * kTwWaveBytes;
  char* lthin = lbig + kTwBigBytes;
  char* lone = lthin + kTwPatchBytes;              // {1, 0, 0, 0} in the element type, then 8 zero bytes
  if (lane == 0) {
    *(unsigned*)(lone) = (unsigned)to_bits16<T>(1.f);
    *(unsigned*)(lone + 4) = 0u; *(unsigned*)(lone + 8) = 0u; *(unsigned*)(lone + 12) = 0u;
  }
  const int i16 = lane & 15, kg = lane >> 4, q = i16 >> 2, p4 = i16 & 3;
  // transposing-read addresses (bytes from the wave's LDS base): A = BIG^T, m-tile mt, first half (pixels 8 kg + q), second + 4 pixels
  int aoff[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) aoff[mt][hh] = tw_big_off(8 * kg + 4 * hh + q, 4 * mt + p4);
  // B = THIN at tap 4 nt + p4: patch row ky, column pixel + kx
  int boff[3][2];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    const int tap = 4 * nt + p4;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (tap <= 8) boff[nt][hh] = kTwBigBytes + ((tap / 3) * kTwPatchCols + (8 * kg + 4 * hh + q + tap % 3)) * 8;
      else boff[nt][hh] = kTwBigBytes + kTwPatchBytes + (tap == 9 ? 0 : 8);     // ones column (bias gradient) / zeros
    }
  }
  f32x4_t acc[4][3];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float tsum[4] = {0.f, 0.f, 0.f, 0.f};
  // staging roles: BIG piece j (0..3): pixel 8 j + (lane >> 3), 16-byte slot lane & 7; THIN items lane, lane + 64 (< 102): row item / 34, col item % 34
  const int bpix = lane >> 3, bslot = lane & 7;
  const long long nw = (long long)gridDim.x * kTwWaves;
  const long long gw = (long long)blockIdx.x * kTwWaves + wave;
  u32x4 rb[4];
  u32x2v rt[2];
  auto load_unit = [&](long long u) {
    const int cb = (int)(u % a.cbx);
    const long long t = u / a.cbx;
    const int row = (int)(t % a.H), img = (int)(t / a.H);
    const int c0 = cb * 32;
    const T* bimg = (const T*)a.big + (size_t)img * a.H * a.W * a.bigC;
    const T* timg = (const T*)a.thin + (size_t)img * a.H * a.W * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int px = c0 + 8 * j + bpix;
      const int c = a.big_c0 + 8 * bslot;
      u32x4 v = {0u, 0u, 0u, 0u};
      // (non-temporal: read once, 229 -> 218 us; the same hint on thin_out's ring loads cost 254 -> 321 us -- its strips re-read 2 of 16 columns
      // and 2 of 34 rows from the caches -- and on thin_in's mask loads 410 -> 423)
      if (px < a.W) v = __builtin_nontemporal_load((const u32x4*)(bimg + (size_t)(row * a.W + px) * a.big_ps + (c >> 5) * a.big_gs + (c & 31)));
      rb[j] = v;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int item = lane + 64 * j;
      const int pr = item / 34, pc = item % 34;
      const int gy = row - 1 + pr, gx = c0 - 1 + pc;
      u32x2v v = {0u, 0u};
      if (item < 102 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = *(const u32x2v*)(timg + (size_t)(gy * a.W + gx) * 4);
      rt[j] = v;
    }
  };
  long long u = gw;
  if (u < a.nunits) load_unit(u);
  for (; u < a.nunits; u += nw) {
    // commit the staged unit to the wave-private LDS image (DS operations of one wave execute in order: the transposing reads of
    // the previous unit are done with the image before these writes land)
#pragma unroll
    for (int j = 0; j < 4; ++j) *(u32x4*)(lbig + tw_big_off(8 * j + bpix, 2 * bslot)) = rb[j];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int item = lane + 64 * j;
      if (item < 102) {
        *(u32x2v*)(lthin + ((item / 34) * kTwPatchCols + item % 34) * 8) = rt[j];
        if (item / 34 == 1 && item % 34 >= 1 && item % 34 <= 32) {      // centre row, own pixels: the thin side's channel sums
          float f[8];
          unpack8<T>(u32x4{rt[j].x, rt[j].y, 0u, 0u}, f);
#pragma unroll
          for (int s = 0; s < 4; ++s) tsum[s] += f[s];
        }
      }
    }
    if (u + nw < a.nunits) load_unit(u + nw);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    Frag bf[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t*)(lbig + boff[nt][0]));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t*)(lbig + boff[nt][1]));
      const u32x2v l = __builtin_bit_cast(u32x2v, lo), h2 = __builtin_bit_cast(u32x2v, hi);
      bf[nt] = __builtin_bit_cast(Frag, u32x4{l.x, l.y, h2.x, h2.y});
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t*)(lbig + aoff[mt][0]));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t*)(lbig + aoff[mt][1]));
      const u32x2v l = __builtin_bit_cast(u32x2v, lo), h2 = __builtin_bit_cast(u32x2v, hi);
      const Frag af = __builtin_bit_cast(Frag, u32x4{l.x, l.y, h2.x, h2.y});
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = mfma16<T>(af, bf[nt], acc[mt][nt]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- workgroup reduction (fixed order: ((w0 + w4) + (w2 + w6)) + ((w1 + w5) + (w3 + w7))), one slab per workgroup ----
  __syncthreads();
  float* red = (float*)smem;          // [4][64 * 48]
  // D layout: lane (column n = i16, rows 4 kg + r): acc[mt][nt][r] = D[16 mt + 4 kg + r][16 nt + i16]
  auto red_idx = [&](int mt, int nt, int r) { return (16 * mt + 4 * kg + r) * 48 + 16 * nt + i16; };
  for (int stride = 4; stride >= 1; stride >>= 1) {
    if (wave >= stride && wave < 2 * stride) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(wave - stride) * 3072 + red_idx(mt, nt, r)] = acc[mt][nt][r];
    }
    __syncthreads();
    if (wave < stride) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mt][nt][r] += red[wave * 3072 + red_idx(mt, nt, r)];
    }
    __syncthreads();
  }
  float* slab = a.slabs + (size_t)blockIdx.x * kTwSlab;
  if (wave == 0) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[red_idx(mt, nt, r)] = acc[mt][nt][r];
  }
  // thin-side channel sums: butterfly inside the wave (fixed order), the eight wave totals through LDS
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float t = tsum[s];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
    if (lane == 0) red[s * kTwWaves + wave] = t;
  }
  __syncthreads();
  if (tid < 4) {
    float t = 0.f;
    for (int i = 0; i < kTwWaves; ++i) t += red[tid * kTwWaves + i];
    slab[3072 + tid] = t;
  }
}

// second stage: sum the slabs in order and scatter into the raw (Cout, Cin, 3, 3) gradient (+ bias gradient)
__global__ __launch_bounds__(256) void thin_wgrad_reduce_kernel(const float* __restrict__ slabs, int nslabs, float* __restrict__ dw, float* __restrict__ db,
                                                                int cs, int big_is_cout) {
  __shared__ float part[256];
  // block handles 64 consecutive slab elements; 4 slab quarters per element
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), qd = threadIdx.x >> 6;
  float t = 0.f;
  if (e < kTwSlab) {
    const int per = (nslabs + 3) / 4;
    const int lo = qd * per, hi = min(nslabs, lo + per);
    for (int s = lo; s < hi; ++s) t += slabs[(size_t)s * kTwSlab + e];
  }
  part[threadIdx.x] = t;
  __syncthreads();
  if (qd != 0 || e >= kTwSlab) return;
  const float v = (part[threadIdx.x] + part[threadIdx.x + 64]) + (part[threadIdx.x + 128] + part[threadIdx.x + 192]);
  if (e < 3072) {
    const int b = e / 48, n = e % 48;
    const int tap = n >> 2, s = n & 3;
    if (tap <= 8) {
      if (s < cs) {
        if (big_is_cout) dw[(b * cs + s) * 9 + tap] = v;
        else dw[(s * 64 + b) * 9 + (8 - tap)] = v;
      }
    } else if (n == 36 && big_is_cout && db) {
      db[b] = v;
    }
  } else if (!big_is_cout && db && e - 3072 < cs) {
    db[e - 3072] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------------
static int thin_check(const srganfd_thin_args* a, const char* what, bool need_thin_in, bool need_thin_out) {
  if (!a) return set_err(SRGANFD_EINVAL, "%s: null argument", what);
  if (a->dtype != SRGANFD_F16 && a->dtype != SRGANFD_BF16) return set_err(SRGANFD_EINVAL, "%s: 16-bit dtypes only (the f32 parity mode runs srganfd_conv2d)", what);
  if (a->n <= 0 || a->h <= 0 || a->w <= 0 || a->cs < 1 || a->cs > 4) return set_err(SRGANFD_EINVAL, "%s: bad dims (n %d h %d w %d thin channels %d)", what, a->n, a->h, a->w, a->cs);
  if (!a->big.ptr || !a->weight) return set_err(SRGANFD_EINVAL, "%s: null pointer", what);
  if (need_thin_in && !a->thin) return set_err(SRGANFD_EINVAL, "%s: thin operand is null", what);
  if (need_thin_out && (!a->thin_out || (a->thin_out_pitch != 4 && !(a->thin_out_pitch == 1 && a->cs == 1))))
    return set_err(SRGANFD_EINVAL, "%s: fp32 output needs a pixel pitch of 4 (or 1 for one channel)", what);
  for (const srganfd_view* v : {&a->big, &a->mask}) {
    if (!v->ptr) continue;
    if (v->c0 % 8 || v->cstride % 8 || ((uintptr_t)v->ptr & 15) || v->c0 + 64 > v->cstride) return set_err(SRGANFD_EINVAL, "%s: 64-channel view not 16-byte aligned / out of range", what);
    if (v->planar && (v->c0 % 32 || v->cstride % 32)) return set_err(SRGANFD_EINVAL, "%s: a planar view needs c0 and cstride multiples of 32", what);
    if ((size_t)a->h * a->w * (size_t)v->cstride * 2 >= 0x7fffffffULL) return set_err(SRGANFD_EINVAL, "%s: one image exceeds 2 GiB", what);
  }
  if (((uintptr_t)a->thin & 7) || (((uintptr_t)a->thin_out & 15) && a->thin_out_pitch == 4) || (size_t)a->h * a->w * 8 >= 0x7fffffffULL) return set_err(SRGANFD_EINVAL, "%s: thin tensor misaligned", what);
  return SRGANFD_OK;
}

static void thin_fill(const srganfd_thin_args* a, ThinK& k) {
  const long long ipix = (long long)a->h * a->w;
  k.thin = a->thin; k.thin_f32 = a->thin_out; k.thin_pitch = a->thin_out_pitch;
  k.big = a->big.ptr; k.mask = a->mask.ptr;
  k.bigC = a->big.cstride; k.big_c0 = a->big.c0; k.big_ps = a->big.planar ? 32 : a->big.cstride; k.big_gs = a->big.planar ? (int)(ipix * 32) : 32;
  k.mC = a->mask.cstride; k.m_c0 = a->mask.c0; k.m_ps = a->mask.planar ? 32 : a->mask.cstride; k.m_gs = a->mask.planar ? (int)(ipix * 32) : 32;
  k.w = a->weight; k.bias = a->bias; k.cs = a->cs; k.big_is_cout = a->w_big_is_cout ? 1 : 0; k.flip = a->flip ? 1 : 0;
  k.N = a->n; k.H = a->h; k.W = a->w;
  k.neg = a->act == SRGANFD_ACT_LRELU ? a->slope : (a->act == SRGANFD_ACT_RELU ? 0.f : 1.f);
  k.mask_slope = a->mask_slope;
  // Non-temporal stores for outputs that one pass cannot keep in the caches anyway (> the 256 MiB Infinity Cache): the write-only
  // 3 -> 64 launch at 512 x 512, batch 32 (1.07 GB) ran 321 -> 240 us with them, the masked launch unchanged (same-box, alternating).
  // (The same switch in conv_igemm's epilogue moved neither training step: 58.8 / 141.0 ms with and without.)
  k.nt = (long long)a->n * ipix * a->big.cstride * 2 >= (192LL << 20) ? 1 : 0;
}

int conv2d_thin_in_impl(const srganfd_thin_args* a, hipStream_t s) {
  const int rc = thin_check(a, "conv2d_thin_in", true, false);
  if (rc != SRGANFD_OK) return rc;
  ThinK k;
  thin_fill(a, k);
  k.tiles_x = ceil_div(a->w, 64); k.tiles_y = ceil_div(a->h, kThinInRows);
  const long long grid = (long long)a->n * k.tiles_x * k.tiles_y;
  if (grid > 0x7fffffffLL) return set_err(SRGANFD_EINVAL, "conv2d_thin_in: grid too large");
  if (g_describe) { snprintf(g_describe, g_describe_len, "thin_in_kernel<%s%s>", a->dtype == SRGANFD_F16 ? "f16" : "bf16", a->mask.ptr ? ",mask" : ""); return SRGANFD_OK; }
#define THIN_IN_LAUNCH(TT, MK) do { if (k.nt) SRGANFD_LAUNCH((thin_in_kernel<TT, MK, true>), dim3((unsigned)grid), dim3(256), 0, s, k); \
                                    else SRGANFD_LAUNCH((thin_in_kernel<TT, MK, false>), dim3((unsigned)grid), dim3(256), 0, s, k); } while (0)
  if (a->dtype == SRGANFD_F16) {
    if (a->mask.ptr) THIN_IN_LAUNCH(f16_t, true); else THIN_IN_LAUNCH(f16_t, false);
  } else {
    if (a->mask.ptr) THIN_IN_LAUNCH(bf16_t, true); else THIN_IN_LAUNCH(bf16_t, false);
  }
#undef THIN_IN_LAUNCH
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

int conv2d_thin_out_impl(const srganfd_thin_args* a, hipStream_t s) {
  const int rc = thin_check(a, "conv2d_thin_out", false, true);
  if (rc != SRGANFD_OK) return rc;
  if (a->big.planar) return set_err(SRGANFD_EINVAL, "conv2d_thin_out: NHWC input views only");
  ThinK k;
  thin_fill(a, k);
  k.tiles_x = ceil_div(ceil_div(a->w, 14), 4); k.tiles_y = ceil_div(a->h, kThinOutRows);
  const long long grid = (long long)a->n * k.tiles_x * k.tiles_y;
  if (grid > 0x7fffffffLL) return set_err(SRGANFD_EINVAL, "conv2d_thin_out: grid too large");
  if (g_describe) { snprintf(g_describe, g_describe_len, "thin_out_kernel<%s>", a->dtype == SRGANFD_F16 ? "f16" : "bf16"); return SRGANFD_OK; }
  if (a->dtype == SRGANFD_F16) SRGANFD_LAUNCH((thin_out_kernel<f16_t>), dim3((unsigned)grid), dim3(256), 0, s, k);
  else SRGANFD_LAUNCH((thin_out_kernel<bf16_t>), dim3((unsigned)grid), dim3(256), 0, s, k);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

static int thin_wgrad_grid() { return 2 * conv_device_cus(); }
size_t conv2d_thin_wgrad_workspace_impl() { return (size_t)thin_wgrad_grid() * kTwSlab * sizeof(float); }

int conv2d_thin_wgrad_impl(const srganfd_thin_args* a, float* dw, float* db, void* ws, size_t ws_bytes, hipStream_t s) {
  const int rc = thin_check(a, "conv2d_thin_wgrad", true, false);
  if (rc != SRGANFD_OK) return rc;
  if (!dw || !ws || ws_bytes < conv2d_thin_wgrad_workspace_impl()) return set_err(SRGANFD_EINVAL, "conv2d_thin_wgrad: null gradient / workspace below srganfd_conv2d_thin_wgrad_workspace()");
  ThinWgK k;
  const long long ipix = (long long)a->h * a->w;
  k.big = a->big.ptr; k.thin = a->thin; k.slabs = (float*)ws;
  k.bigC = a->big.cstride; k.big_c0 = a->big.c0; k.big_ps = a->big.planar ? 32 : a->big.cstride; k.big_gs = a->big.planar ? (int)(ipix * 32) : 32;
  k.N = a->n; k.H = a->h; k.W = a->w; k.cbx = ceil_div(a->w, 32);
  k.nunits = (long long)a->n * a->h * k.cbx;
  const int grid = thin_wgrad_grid();
  if (g_describe) { snprintf(g_describe, g_describe_len, "thin_wgrad_kernel<%s>", a->dtype == SRGANFD_F16 ? "f16" : "bf16"); return SRGANFD_OK; }
  if (a->dtype == SRGANFD_F16) SRGANFD_LAUNCH((thin_wgrad_kernel<f16_t>), dim3((unsigned)grid), dim3(64 * kTwWaves), kTwLds, s, k);
  else SRGANFD_LAUNCH((thin_wgrad_kernel<bf16_t>), dim3((unsigned)grid), dim3(64 * kTwWaves), kTwLds, s, k);
  SRGANFD_LAUNCH(thin_wgrad_reduce_kernel, dim3((unsigned)ceil_div(kTwSlab, 64)), dim3(256), 0, s, (const float*)ws, grid, dw, db, a->cs, a->w_big_is_cout ? 1 : 0);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

}  // namespace srganfd
