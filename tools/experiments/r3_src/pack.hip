// pack.hip -- NCHW fp32 parameters -> MFMA B-fragment order (see srganfd.h, srganfd_pack_weights).
// Layout of a packed operand W[tap][k][n] (k, n multiples of 32):
//   [n/32][k/32][tap][kstep][lane 0..63][frag]   with
//   bf16 / f16: kstep in 0..1, frag = 8 elements, k = 32*chunk + 16*kstep + 8*(lane>>5) + j, n = 32*ntile + (lane&31)
//   f32 : kstep in 0..15, frag = 1 float, k = 32*chunk + 2*kstep + (lane>>5)
// i.e. exactly the order in which conv_igemm.hip's lanes consume B fragments, so staging a
// (chunk, n-tile) slab into LDS is a linear 16-byte copy and the fragment read is conflict free.
#include "common.hpp"

namespace srganfd {

__global__ __launch_bounds__(256) void pack_kernel(const srganfd_pack_job* __restrict__ jobs, const float* __restrict__ params,
                                                   const float* __restrict__ scalars, char* __restrict__ packed) {
  const srganfd_pack_job& J = jobs[blockIdx.y];
  const int KT = J.ksize * J.ksize;
  const long long total = (long long)KT * J.k * J.n;
  const int nChunks = J.k >> 5;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    int lane, s, k_in, within = (int)(e & 1023);
    long long blk = e >> 10;  // (ntile, chunk, tap)
    const int tap = (int)(blk % KT); blk /= KT;
    const int chunk = (int)(blk % nChunks);
    const int ntile = (int)(blk / nChunks);
    int n_in = -1;
    if (J.dtype != SRGANFD_F32 && J.layout == 1) {   // 16x16x32 B fragments: [channel half s][lane][8], k = 8*(lane>>4) + j
      const int j = within & 7; lane = (within >> 3) & 63; s = within >> 9;
      k_in = 8 * (lane >> 4) + j;
      n_in = 16 * s + (lane & 15);
    } else if (J.dtype != SRGANFD_F32) {   // bf16 / f16, 32x32x16: [k-step s][lane][8]
      const int j = within & 7; lane = (within >> 3) & 63; s = within >> 9;
      k_in = 16 * s + 8 * (lane >> 5) + j;
    } else {
      lane = within & 63; s = within >> 6;
      k_in = 2 * s + (lane >> 5);
    }
    const int k = chunk * 32 + k_in, n = ntile * 32 + (n_in >= 0 ? n_in : (lane & 31));
    float v = 0.f;
    for (int g = 0; g < J.nseg; ++g) {
      const srganfd_pack_seg& S = J.seg[g];
      if (k >= S.k_lo && k < S.k_lo + S.k_len) {
        const int kk = k - S.k_lo;
        int co, ci, t;
        int KTs = KT;
        if (!S.transposed) { co = n + S.co_off; ci = kk + S.ci_off; t = tap; }
        else if (S.transposed == 1) { co = kk + S.co_off; ci = n + S.ci_off; t = KT - 1 - tap; }
        else if (S.transposed < 6) {  // parity class of the 4x4 stride-2 data gradient: this operand has 2x2 taps, the source 4x4
          const int py = (S.transposed - 2) >> 1, px = (S.transposed - 2) & 1, ta = tap >> 1, tb = tap & 1;
          const int ty = py ? 2 - 2 * ta : 3 - 2 * ta, tx = px ? 2 - 2 * tb : 3 - 2 * tb;
          co = kk + S.co_off; ci = n + S.ci_off; t = ty * 4 + tx; KTs = 16;
        } else if (S.transposed < 10) {  // parity class of a 3x3 stride-2 pad-1 data gradient as a 2x2-tap operand (pad 0):
          // even output rows see only kernel row 1 (tap a=0), odd rows see kernel rows 2 (a=0) and 0 (a=1)
          const int py = (S.transposed - 6) >> 1, px = (S.transposed - 6) & 1, ta = tap >> 1, tb = tap & 1;
          const int ty = py ? (ta ? 0 : 2) : (ta ? -1 : 1), tx = px ? (tb ? 0 : 2) : (tb ? -1 : 1);
          co = kk + S.co_off; ci = n + S.ci_off; KTs = 9;
          t = ty * 3 + tx;
          if (ty < 0 || tx < 0) co = S.co_src;   // unused tap -> zero
        } else {  // 10 + 2a + b: tap (a,b) of a 2x2 stride-2 conv as a 1x1 data-gradient operand
          const int ab = S.transposed - 10;
          co = kk + S.co_off; ci = n + S.ci_off; t = ab; KTs = 4;
        }
        if (co < S.co_src && ci < S.ci_src) {
          v = params[S.src_off + ((long long)co * S.ci_src + ci) * KTs + t] * S.scale;
          if (S.scale_off >= 0) v *= scalars[S.scale_off];
        }
        break;
      }
    }
    if (J.dtype == SRGANFD_BF16) ((bf16_t*)(packed + J.dst_off))[e] = f2bf(v);
    else if (J.dtype == SRGANFD_F16) ((f16_t*)(packed + J.dst_off))[e] = (f16_t)v;
    else ((float*)(packed + J.dst_off))[e] = v;
  }
}

int pack_weights_impl(const srganfd_pack_job* jobs_dev, int njobs, long long max_elems, const float* params,
                      const float* scalars, void* packed, hipStream_t stream) {
  if (!jobs_dev || njobs <= 0 || max_elems <= 0 || !params || !packed) return set_err(SRGANFD_EINVAL, "pack_weights: bad args");
  long long gx = (max_elems + 255) / 256;
  if (gx > 4096) gx = 4096;
  SRGANFD_LAUNCH(pack_kernel, dim3((unsigned)gx, (unsigned)njobs), dim3(256), 0, stream, jobs_dev, params, scalars, (char*)packed);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

}  // namespace srganfd
