// conv_stream.hip -- the 3x3 stride-1 convolutions of the hot path (dense blocks and their data gradients, BSRGAN/model.py:42-58; the
// generator's trunk / tail convs :330-355; the U-Net discriminator's decoder :116-135; VGG-19's convs) as ONE persistent workgroup per CU
// through which haloed input patches -- and, where it does not fit LDS whole, the weight slab of each 32-channel chunk -- stream by
// LDS-DMA into two stage buffers.
//
// Why a second kernel beside conv_igemm.hip: there a tile is global -> register -> LDS staging with two barriers per chunk, and the
// ablations of three rounds add up (copy + MFMA phase + epilogue = kernel time; 46 % MFMA utilisation on the 64-channel class).  Here
//   * a stage = the 18 x 34-pixel patch of one 32-channel chunk (39 KiB) [+ the chunk's weight slab, 18 KiB per 32 output channels],
//     written to LDS by global_load_lds_dwordx4 issued by EVERY wave right after the barrier that frees the buffer
//     (tools/probes/overlap_probe.hip: copy and MFMA phase overlap fully when the same waves issue the pieces and then compute; with
//     dedicated loader waves they add up) -- no staging registers, no ds_write, ONE barrier per chunk;
//   * stages run across tile borders: the first chunk of the next tile is in flight during the last MFMA phase and the epilogue;
//   * a weight block of <= 72 KiB (<= 128 input x 32 output channels, 64 x 64) is fetched once per workgroup and stays;
//   * a wave owns 2 rows x 32 pixels x ALL output channels of the workgroup (32 or 64): 60 fragment reads per 144 MFMAs on 64-channel
//     tiles (conv_igemm: 84), 256 registers per lane available (two waves per SIMD);
//   * MFMA operands are swapped (A = weights, B = pixels), so a lane's four accumulator registers are four consecutive output channels
//     of one pixel: the epilogue works from registers with 8-byte loads / stores -- no LDS tile, no barrier.
// Same contract as conv_igemm.hip (srganfd_conv2d, include/srganfd.h); weights in the 16x16x32 B-fragment order of pack.hip.
#ifdef SRGANFD_EXPERIMENT   // streaming conv: measured and rejected (profiles/r03_conv_experiments.txt 12); kept for the A/B tools
#include "conv_common.hpp"
#include <stdlib.h>

#ifndef SRGANFD_STREAM_PIPE
#define SRGANFD_STREAM_PIPE 4      // fragment reads issued this many ahead of their first use
#endif

namespace srganfd {

namespace {
__device__ __forceinline__ unsigned stream_lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to LDS [dst, dst + 1 KiB), dst wave-uniform.  Inline asm as in
// wgrad.hip: the builtin form makes hipcc order every later LDS read behind vmcnt(0); the kernel counts vmcnt itself.
__device__ __forceinline__ void stream_glds16(const void* gsrc, unsigned lds_dst) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(dst) : "m0");
#pragma clang diagnostic pop
}

// Issue order of one kernel column's fragment reads and MFMAs for NH 16-channel halves (NH = 2: conv_igemm.hip's m16 order up to a
// permutation).  Reads: A(r0,p0) A(r0,p1) A(r1,p0) A(r1,p1) | B(ky0, 0..NH-1) | A(r2,p*) | B(ky1, *) | A(r3,p*) | B(ky2, *);
// MFMAs: for ky, nh, m, ph: acc[m][ph][nh] += B(ky, nh) x A(m + ky, ph) -- 4 NH independent accumulators between two uses of one.
constexpr int sr_reads(int NH) { return 8 + 3 * NH; }
constexpr int sr_idxA(int rr, int ph, int NH) { return (rr < 2 ? 2 * rr : (rr == 2 ? 4 + NH : 6 + 2 * NH)) + ph; }
constexpr int sr_idxB(int ky, int nh, int NH) { return (ky == 0 ? 4 : (ky == 1 ? 6 + NH : 8 + 2 * NH)) + nh; }
constexpr bool sr_isB(int l, int NH) { return (l >= 4 && l < 4 + NH) || (l >= 6 + NH && l < 6 + 2 * NH) || l >= 8 + 2 * NH; }
constexpr int sr_ky(int l, int NH) { return l < 4 + NH ? 0 : (l < 6 + 2 * NH ? 1 : 2); }
constexpr int sr_nh(int l, int NH) { return l < 4 + NH ? l - 4 : (l < 6 + 2 * NH ? l - (6 + NH) : l - (8 + 2 * NH)); }
constexpr int sr_row(int l, int NH) { return l < 4 ? l / 2 : (l < 6 + NH ? 2 : 3); }
constexpr int sr_ph(int l, int NH) { return l < 4 ? l % 2 : (l < 6 + NH ? l - (4 + NH) : l - (6 + 2 * NH)); }
// last read issued before MFMA i of the chunk (3 columns x 12 NH MFMAs): what it needs plus d of read-ahead, never decreasing
constexpr int sr_hi(int i, int d, int NH) {
  const int per = 12 * NH, nl = 3 * sr_reads(NH);
  int best = 0;
  for (int q = 0; q <= i; ++q) {
    const int kx = q / per, j = q % per;
    const int ky = j / (4 * NH), nh = (j / 4) % NH, m = (j >> 1) & 1, ph = j & 1;
    const int a = sr_idxA(m + ky, ph, NH), b = sr_idxB(ky, nh, NH);
    const int need = kx * sr_reads(NH) + (a > b ? a : b) + d;
    if (need > best) best = need;
  }
  return best < nl - 1 ? best : nl - 1;
}
}  // namespace

template <int NT, int WS = 1> struct StreamCfg {
  static constexpr int TH = 16, TW = 32, PR = 18, PC = 34, NPIX = PR * PC;
  static constexpr int NWAVES = 8 * WS, NTHR = 64 * NWAVES, NH = 2 * NT, NB = 32 * NT;
  static constexpr int NHW = NH / WS;                            // 16-channel halves per wave (WS = 2: sixteen waves, each half the channels)
  static_assert(NH % WS == 0, "wave split");
  static constexpr int XPIECES = (NPIX * 64 + 1023) / 1024;     // 39 pieces of 1 KiB per patch
  static constexpr int XBYTES = XPIECES * 1024;
  static constexpr int NSLOT = (XPIECES + NWAVES - 1) / NWAVES; // patch pieces per wave and stage
  static constexpr int WT = 18 * 1024;                          // one 32-channel n-tile's slab of one chunk: 9 taps x 2 halves x 1 KiB
  static constexpr int WCH = NT * WT;
  static constexpr int STREAMED_BYTES = 2 * (XBYTES + WCH);     // weights streamed with every stage
  static constexpr int MAX_RES_CHUNKS = (160 * 1024 - 2 * XBYTES) / WCH;
  static constexpr int resident_bytes(int chunks) { return chunks * WCH + 2 * XBYTES; }
  static_assert(STREAMED_BYTES <= 160 * 1024, "LDS");
};

// EK: epilogue operands fixed at compile time -- bit 1 residual r1, 2 residual r2, 4 LeakyReLU'(mask) -- for dense 16-bit outputs; -1 =
// every operand decided at run time with per-element guards (fp32 output, partial channel blocks, y2).  WRES: the weight block stays in
// LDS (one output-channel block per launch), else each stage carries its chunk's slab.
template <typename T, int NT, int EK, bool WRES, int WS = 1>
__global__ __launch_bounds__(512 * WS, 2 * WS) void conv_stream_kernel(const ConvK a) {
  using C = StreamCfg<NT, WS>;
  using Frag = typename FragAB<T>::type;
  constexpr int NH = C::NHW;        // halves this wave owns; its first one is h0
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrow = wave / WS, h0 = (wave % WS) * NH;
  constexpr int STAGE = WRES ? C::XBYTES : C::XBYTES + C::WCH;
  char* const ldsW = smem;                                          // WRES: [n-tile][chunk][18 KiB]
  char* const ldsS = smem + (WRES ? a.nChunks * C::WCH : 0);        // stage b at + b * STAGE: patch [, the chunk's slabs [n-tile][18 KiB]]

  auto decode = [&](int vb, int& n_, int& oy_, int& ox_, int& nb_) {
    const int bid = xcd_remap(vb, a.nblocks);
    const int t0 = (int)fast_div((unsigned)bid, (unsigned)a.nNb, a.m_nNb);
    nb_ = __builtin_amdgcn_readfirstlane(bid - t0 * a.nNb);
    const int t1 = (int)fast_div((unsigned)t0, (unsigned)a.tiles_x, a.m_tx);
    const int tx = __builtin_amdgcn_readfirstlane(t0 - t1 * a.tiles_x);
    const int t2 = (int)fast_div((unsigned)t1, (unsigned)a.tiles_y, a.m_ty);
    const int ty = __builtin_amdgcn_readfirstlane(t1 - t2 * a.tiles_y);
    n_ = __builtin_amdgcn_readfirstlane(t2);
    oy_ = ty * C::TH; ox_ = tx * C::TW;
  };

  // this wave's patch pieces of a stage: piece = wave + 8 k.  Lane l of a piece fills LDS slot (pixel 16 piece + l / 4, 16-byte position
  // l & 3); the position holds source chunk (l & 3) ^ 2 ((px >> 2) & 1) of the pixel -- the column-keyed swizzle the fragment reads undo.
  int ppos[C::NSLOT];     // py | px << 8; -1 = beyond the patch
  int poff[C::NSLOT];     // element offset from the patch's first pixel
#pragma unroll
  for (int k = 0; k < C::NSLOT; ++k) {
    const int piece = wave + C::NWAVES * k;
    const int item = piece * 64 + lane, pix = item >> 2;
    ppos[k] = -1; poff[k] = 0;
    if (piece < C::XPIECES && pix < C::NPIX) {
      const int py = pix / C::PC, px = pix - py * C::PC;
      const int c16 = (item & 3) ^ (((px >> 2) & 1) << 1);
      ppos[k] = py | (px << 8);
      poff[k] = (py * a.Win + px) * a.x_ps + c16 * 8;
    }
  }
  auto issue_stage = [&](int n_, int oy_, int ox_, int nb_, int chunk, int buf) __attribute__((always_inline)) {
    const int gy0 = oy_ - 1, gx0 = ox_ - 1;
    const T* xt = (const T*)a.x + ((long long)n_ * a.Hin * a.Win * a.xC + a.x_base + (long long)chunk * a.x_cs + ((long long)gy0 * a.Win + gx0) * a.x_ps);
    char* sb = ldsS + buf * STAGE;
    if (!SRGANFD_DBG(a.dbg, 1)) {
#pragma unroll
      for (int k = 0; k < C::NSLOT; ++k) {
        const int piece = wave + C::NWAVES * k;          // wave-uniform
        if (piece >= C::XPIECES) continue;
        const int q = ppos[k];
        if (q < 0) continue;
        const int gy = gy0 + (q & 255), gx = gx0 + (q >> 8);
        if (gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) stream_glds16(xt + poff[k], stream_lds_addr(sb) + (unsigned)(piece * 1024));
        else *(u32x4*)(sb + piece * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
      }
    }
    if constexpr (!WRES) {
      if (!SRGANFD_DBG(a.dbg, 16)) {
        // the chunk's slabs: n-tile j of this block at [(nb NT + j) chunks + chunk] x 18 KiB of the packed operand
        const char* wsrc = (const char*)a.w + ((size_t)(nb_ * NT) * a.nChunks + chunk) * C::WT + lane * 16;
        const unsigned wdst = stream_lds_addr(sb) + C::XBYTES;
#pragma unroll
        for (int k = 0; k < (18 * NT + C::NWAVES - 1) / C::NWAVES; ++k) {
          const int q = wave + C::NWAVES * k;              // wave-uniform
          if (q < 18 * NT) {
            const int j = q / 18, f = q - j * 18;
            stream_glds16(wsrc + ((size_t)j * a.nChunks * 18 + f) * 1024, wdst + (unsigned)(q * 1024));
          }
        }
      }
    }
  };

  // fragment address terms: lane term of kernel column kx (cf. conv_igemm.hip, kColSwz), this wave's two output rows
  int colt[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) colt[kx] = ((lane & 15) + kx) * 64 + (((lane >> 4) ^ (((((lane & 15) + kx) >> 2) & 1) << 1)) << 4);
  const int rowoff = wrow * 2 * C::PC * 64;

  float alpha = a.alpha;
  if (a.alpha_dev) alpha *= *a.alpha_dev;
  const int g4 = lane >> 4, l15 = lane & 15;
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

  int vt = blockIdx.x;
  int n, oy0, ox0, nb;
  decode(vt, n, oy0, ox0, nb);
  issue_stage(n, oy0, ox0, nb, 0, 0);
  if constexpr (WRES) {
    const char* wsrc = (const char*)a.w + lane * 16;
    const int npieces = a.nChunks * 18 * NT;
    for (int i = wave; i < npieces; i += C::NWAVES) stream_glds16(wsrc + (size_t)i * 1024, stream_lds_addr(ldsW) + (unsigned)(i * 1024));
  }
  int buf = 0;
  bool full_prev = false;       // the previous tile of this workgroup issued all of its stores (vmcnt bookkeeping)

  for (;;) {
    const bool more = vt + (int)gridDim.x < a.nblocks;
    int n2 = 0, oy2 = 0, ox2 = 0, nb2 = 0;
    if (more) decode(vt + gridDim.x, n2, oy2, ox2, nb2);
    f32x4_t acc[2][2][NH];     // [row][pixel half][16-channel half]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int nh = 0; nh < NH; ++nh)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[m][ph][nh][i] = 0.f;

    // epilogue operands of this lane's outputs (8 bytes = its 4 channels of one pixel), requested in front of the last chunk's stage
    // issue so that they are OLDER than it: the epilogue's wait for them does not drain the next tile's first stage
    constexpr bool kR1 = EK >= 0 && (EK & 1), kR2 = EK >= 0 && (EK & 2), kMk = EK >= 0 && (EK & 4);
    constexpr bool kPreR2 = kR2 && NT == 1;      // 64-channel tiles: r2 is read in the epilogue itself (its 32 prefetch registers spilled)
    u32x2 o_r1[kR1 ? 4 * NH : 1], o_r2[kPreR2 ? 4 * NH : 1], o_mk[kMk ? 4 * NH : 1];
    f32x4_t bv[NH];          // bias of this lane's channels (requested with the operands above)
    const size_t img = (size_t)n * a.HoutF * a.WoutF;
    const int cob = nb * C::NB;              // first output channel of this workgroup

    for (int chunk = 0; chunk < a.nChunks; ++chunk) {
      // the stage's pieces are this wave's oldest outstanding vector-memory operations; behind them at most the previous tile's stores
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
      if (chunk == 0 && full_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NH) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma clang diagnostic pop
      __syncthreads();          // stage `buf` is complete; every wave has left the MFMA phase that read the other buffer
      if (chunk + 1 == a.nChunks) {
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) {
          bv[nh] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          if (a.bias) {     // (an offset into a flat parameter buffer: 4-byte alignment only)
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[nh][i] = a.bias[cob + 16 * (h0 + nh) + 4 * g4 + i];
          }
        }
        if constexpr (kR1 || kR2 || kMk) {
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
              const int oy = oy0 + 2 * wrow + m, ox = ox0 + 16 * ph + l15;
              const bool ok = oy < a.Hout && ox < a.Wout;
              const int p = oy * a.WoutF + ox;
#pragma unroll
              for (int nh = 0; nh < NH; ++nh) {
                const int e = (m * 2 + ph) * NH + nh, co = cob + 16 * (h0 + nh) + 4 * g4;
                auto ld = [&](const void* base, int Cs, int c0, int ps, int gs) -> u32x2 {
                  const int cc = c0 + co;
                  return ok ? *(const u32x2*)((const T*)base + img * Cs + (p * ps + (cc >> 5) * gs + (cc & 31))) : u32x2{0u, 0u};
                };
                if constexpr (kR1) o_r1[e] = ld(a.r1, a.r1C, a.r1_c0, a.r1_ps, a.r1_gs);
                if constexpr (kPreR2) o_r2[e] = ld(a.r2, a.r2C, a.r2_c0, a.r2_ps, a.r2_gs);
                if constexpr (kMk) o_mk[e] = ld(a.mask, a.mC, a.m_c0, a.m_ps, a.m_gs);
              }
            }
        }
      }
      if (chunk + 1 < a.nChunks) issue_stage(n, oy0, ox0, nb, chunk + 1, buf ^ 1);
      else if (more) issue_stage(n2, oy2, ox2, nb2, 0, buf ^ 1);

      // MFMA phase: the chunk's fragment reads and MFMAs in a fixed issue order, every read kPipe fragments ahead of its first use
      const char* ldsXw = ldsS + buf * STAGE + rowoff;
      const int wtile = WRES ? a.nChunks * C::WT : C::WT;           // distance of the n-tiles' slabs
      const char* ldsWn = (WRES ? ldsW + chunk * C::WT : ldsS + buf * STAGE + C::XBYTES) + lane * 16 + (h0 >> 1) * wtile + (NH == 1 ? (h0 & 1) * 1024 : 0);
      __builtin_amdgcn_s_setprio(1);
#ifdef SRGANFD_STREAM_PLAIN
      if (!SRGANFD_DBG(a.dbg, 2)) {
        // the compiler's own schedule (as tools/probes/overlap_probe.hip): per kernel column all fragments, then the MFMAs
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          Frag av[4][2], bw[3][NH];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) av[rr][ph] = *(const Frag*)(ldsXw + colt[kx] + (rr * C::PC + 16 * ph) * 64);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) bw[ky][nh] = *(const Frag*)(ldsWn + (nh >> 1) * wtile + ((ky * 3 + kx) * 2 + (nh & 1)) * 1024);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int nh = 0; nh < NH; ++nh)
#pragma unroll
              for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) acc[m][ph][nh] = mfma16<T>(bw[ky][nh], av[m + ky][ph], acc[m][ph][nh]);
        }
      }
#else
      if (!SRGANFD_DBG(a.dbg, 2)) {
        constexpr int RD = sr_reads(NH), NL = 3 * RD, PER = 12 * NH, NM = 3 * PER, kPipe = SRGANFD_STREAM_PIPE;
        Frag F[NL];
        static_for<NM>([&](auto ic) {
          constexpr int i = decltype(ic)::v;
          constexpr int hi = sr_hi(i, kPipe, NH), lo = i == 0 ? 0 : sr_hi(i - 1, kPipe, NH) + 1;
          static_for<hi - lo + 1>([&](auto jc) {
            constexpr int nn = lo + decltype(jc)::v;
            constexpr int kx = nn / RD, l = nn % RD;
            if constexpr (sr_isB(l, NH)) {
              constexpr int ky = sr_ky(l, NH), nh = sr_nh(l, NH);
              F[nn] = *(const Frag*)(ldsWn + (nh >> 1) * wtile + ((ky * 3 + kx) * 2 + (nh & 1)) * 1024);
            } else {
              constexpr int rr = sr_row(l, NH), ph = sr_ph(l, NH);
              F[nn] = *(const Frag*)(ldsXw + colt[kx] + (rr * C::PC + 16 * ph) * 64);
            }
          });
          constexpr int c = i / PER, j = i % PER;
          constexpr int ky = j / (4 * NH), nh = (j / 4) % NH, m = (j >> 1) & 1, ph = j & 1;
          acc[m][ph][nh] = mfma16<T>(F[RD * c + sr_idxB(ky, nh, NH)], F[RD * c + sr_idxA(m + ky, ph, NH)], acc[m][ph][nh]);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
#endif
      __builtin_amdgcn_s_setprio(0);
      buf ^= 1;
    }

    // ---- epilogue: lane = pixel 16 ph + l15 of row 2 wave + m, channels 16 nh + 4 g4 .. + 3 (see srganfd.h for the formula) ----
    const float neg = a.act == SRGANFD_ACT_LRELU ? a.slope : (a.act == SRGANFD_ACT_RELU ? 0.f : 1.f);
    const float ps_pos = a.post_scale, ps_neg = neg * a.post_scale;
    auto widen4 = [](const u32x2 q, float* f) {
      if constexpr (Elem<T>::kDtype == SRGANFD_F16) {
        typedef __attribute__((ext_vector_type(4))) _Float16 h4;
        const h4 hv = __builtin_bit_cast(h4, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = (float)hv[i];
      } else {
        f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xffff0000u);
        f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xffff0000u);
      }
    };
    bool full = true;
#pragma unroll
    for (int nh = 0; nh < NH; ++nh) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          const int oy = oy0 + 2 * wrow + m, ox = ox0 + 16 * ph + l15;
          const bool ok = oy < a.Hout && ox < a.Wout;
          if (__builtin_amdgcn_ballot_w64(!ok) != 0) full = false;
          const int p = oy * a.WoutF + ox;
          const int e = (m * 2 + ph) * NH + nh;
          float v4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = __builtin_fmaf(alpha, acc[m][ph][nh][i], bv[nh][i]);     // explicit fma chain: every epilogue kind rounds alike
            v4[i] = v * (v > 0.f ? ps_pos : ps_neg);
          }
          if constexpr (EK >= 0) {
            float t4[4];
            if constexpr (kR1) { widen4(o_r1[e], t4);
#pragma unroll
              for (int i = 0; i < 4; ++i) v4[i] = __builtin_fmaf(a.r1s, t4[i], v4[i]); }
            if constexpr (kR2) {
              if constexpr (kPreR2) widen4(o_r2[e], t4);
              else {
                const int cc = a.r2_c0 + cob + 16 * (h0 + nh) + 4 * g4;
                widen4(ok ? *(const u32x2*)((const T*)a.r2 + img * a.r2C + (p * a.r2_ps + (cc >> 5) * a.r2_gs + (cc & 31))) : u32x2{0u, 0u}, t4);
              }
#pragma unroll
              for (int i = 0; i < 4; ++i) v4[i] = __builtin_fmaf(a.r2s, t4[i], v4[i]); }
            if constexpr (kMk) { widen4(o_mk[e], t4);
#pragma unroll
              for (int i = 0; i < 4; ++i) v4[i] *= t4[i] > 0.f ? 1.f : a.mask_slope; }
            u32x2 pk;
            if constexpr (Elem<T>::kDtype == SRGANFD_F16) {
              typedef __attribute__((ext_vector_type(4))) _Float16 h4;
              const h4 hv = {(_Float16)v4[0], (_Float16)v4[1], (_Float16)v4[2], (_Float16)v4[3]};
              pk = __builtin_bit_cast(u32x2, hv);
            } else {
              pk = u32x2{(unsigned)f2bf(v4[0]) | ((unsigned)f2bf(v4[1]) << 16), (unsigned)f2bf(v4[2]) | ((unsigned)f2bf(v4[3]) << 16)};
            }
            if (ok && !SRGANFD_DBG(a.dbg, 4)) {
              const int cc = a.y_c0 + cob + 16 * (h0 + nh) + 4 * g4;
              *(u32x2*)((T*)a.y + img * a.yC + (p * a.y_ps + (cc >> 5) * a.y_gs + (cc & 31))) = pk;
            }
          } else {
            // run-time operands, one element at a time (the 3-channel SR output in fp32, the discriminator's 1-channel logits, y2 launches)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int co = cob + 16 * (h0 + nh) + 4 * g4 + i;
              if (ok && co < a.cout_store) {
                auto at = [&](int Cs, int c0, int ps, int gs) -> size_t { const int cc = c0 + co; return img * Cs + (size_t)(p * ps + (cc >> 5) * gs + (cc & 31)); };
                float v = v4[i];
                if (a.y2) ((T*)a.y2)[at(a.y2C, a.y2_c0, a.y2_ps, a.y2_gs)] = Elem<T>::from_f(v);
                if (a.r1) v = __builtin_fmaf(a.r1s, Elem<T>::to_f(((const T*)a.r1)[at(a.r1C, a.r1_c0, a.r1_ps, a.r1_gs)]), v);
                if (a.r2) v = __builtin_fmaf(a.r2s, Elem<T>::to_f(((const T*)a.r2)[at(a.r2C, a.r2_c0, a.r2_ps, a.r2_gs)]), v);
                if (a.mask) v *= (Elem<T>::to_f(((const T*)a.mask)[at(a.mC, a.m_c0, a.m_ps, a.m_gs)]) > 0.f) ? 1.f : a.mask_slope;
                if (a.y_f32) ((float*)a.y)[at(a.yC, a.y_c0, a.y_ps, a.y_gs)] = v;
                else ((T*)a.y)[at(a.yC, a.y_c0, a.y_ps, a.y_gs)] = Elem<T>::from_f(v);
              }
            }
          }
        }
    }
    full_prev = EK >= 0 && full && !SRGANFD_DBG(a.dbg, 4);
    if (!more) break;
    vt += gridDim.x; n = n2; oy0 = oy2; ox0 = ox2; nb = nb2;
  }
}

int g_use_stream = [] { const char* e = getenv("SRGANFD_USE_STREAM"); return e ? atoi(e) : 0; }();   // A/B: srganfd_set_igemm_variant bit 11

int g_stream_ws = [] { const char* e = getenv("SRGANFD_STREAM_WS"); return e ? atoi(e) : 1; }();   // 2: sixteen waves per workgroup, each wave half the output channels (f16, fixed kinds)

template <typename T, int NT, int EK, bool WRES, int WS = 1>
static int launch_stream(const ConvK& k, int cout, hipStream_t stream) {
  using C = StreamCfg<NT, WS>;
  auto kern = conv_stream_kernel<T, NT, EK, WRES, WS>;
  if (g_describe) {
    char ek[8] = "";
    if (EK >= 0) snprintf(ek, sizeof(ek), ",E%d", EK);
    snprintf(g_describe, g_describe_len, "conv_stream_kernel<%s,N=%d%s%s%s>", dtype_name<T>(), C::NB, WRES ? ",WR" : "", WS == 2 ? ",W16" : "", ek);
    return SRGANFD_OK;
  }
  const int lds = WRES ? C::resident_bytes(k.nChunks) : C::STREAMED_BYTES;
  static unsigned long long attr_done = 0;
  if (!g_dry_run) {
    int dev = 0;
    SRGANFD_HIP_CHECK(hipGetDevice(&dev));
    if (!(attr_done >> (dev & 63) & 1ULL)) {
      SRGANFD_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, WRES ? C::resident_bytes(C::MAX_RES_CHUNKS) : C::STREAMED_BYTES));
      attr_done |= 1ULL << (dev & 63);
    }
  }
  ConvK kk = k;
  kk.nNb = cout / C::NB;
  kk.tiles_x = ceil_div(k.Wout, C::TW);
  kk.tiles_y = ceil_div(k.Hout, C::TH);
  const long long nblk = (long long)k.N * kk.tiles_x * kk.tiles_y * kk.nNb;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return set_err(SRGANFD_EINVAL, "conv2d: bad grid %lld", nblk);
  kk.nblocks = (int)nblk;
  kk.m_nNb = div_magic((unsigned)kk.nNb, (unsigned long long)nblk);
  kk.m_tx = div_magic((unsigned)kk.tiles_x, (unsigned long long)nblk);
  kk.m_ty = div_magic((unsigned)kk.tiles_y, (unsigned long long)nblk);
  const long long slots = conv_device_cus() / 8 * 8;      // one workgroup per CU (a multiple of 8: a workgroup's tiles keep their XCD class)
  const long long grid = nblk > slots && slots >= 8 ? slots : nblk;
  SRGANFD_LAUNCH(kern, dim3((unsigned)grid), dim3(C::NTHR), (size_t)lds, stream, kk);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

template <typename T, int NT, bool WRES>
static int launch_stream_kind(const ConvK& k, int cout, int ek, hipStream_t stream) {
  if constexpr (sizeof(T) == 2 && Elem<T>::kDtype == SRGANFD_F16) {
    if (g_stream_ws == 2) {
      switch (ek) {
        case 0: return launch_stream<T, NT, 0, WRES, 2>(k, cout, stream);
        case 1: return launch_stream<T, NT, 1, WRES, 2>(k, cout, stream);
        case 3: return launch_stream<T, NT, 3, WRES, 2>(k, cout, stream);
        case 4: return launch_stream<T, NT, 4, WRES, 2>(k, cout, stream);
        default: break;
      }
    }
  }
  switch (ek) {
    case 0: return launch_stream<T, NT, 0, WRES>(k, cout, stream);
    case 1: return launch_stream<T, NT, 1, WRES>(k, cout, stream);
    case 3: return launch_stream<T, NT, 3, WRES>(k, cout, stream);
    case 4: return launch_stream<T, NT, 4, WRES>(k, cout, stream);
    default: return launch_stream<T, NT, -1, WRES>(k, cout, stream);
  }
}

template <typename T>
static int dispatch_stream(const srganfd_conv_args* a, const ConvK& k, hipStream_t stream) {
  int ek = -1;
  if (k.fast_epi && !k.y2) {
    ek = (k.r1 ? 1 : 0) | (k.r2 ? 2 : 0) | (k.mask ? 4 : 0);
    if (ek != 0 && ek != 1 && ek != 3 && ek != 4) ek = -1;
  }
  if (a->cout % 64 == 0) {
    const bool res = a->cout == 64 && k.nChunks <= StreamCfg<2>::MAX_RES_CHUNKS;
    return res ? launch_stream_kind<T, 2, true>(k, a->cout, ek, stream) : launch_stream_kind<T, 2, false>(k, a->cout, ek, stream);
  }
  const bool res = a->cout == 32 && k.nChunks <= StreamCfg<1>::MAX_RES_CHUNKS;
  return res ? launch_stream_kind<T, 1, true>(k, a->cout, ek, stream) : launch_stream_kind<T, 1, false>(k, a->cout, ek, stream);
}

// srganfd_conv2d launches this kernel takes (the caller has validated the arguments and filled k): 16-bit, 3x3 stride 1 pad 1, no
// nearest-x2 gather, dense output
int conv_stream_try(const srganfd_conv_args* a, const ConvK& k, hipStream_t stream, bool* handled) {
  *handled = false;
  if (!g_use_stream || a->dtype == SRGANFD_F32 || !conv_uses_m16(a->dtype, a->ksize, a->cout)) return SRGANFD_OK;
  if (a->ksize != 3 || a->stride != 1 || a->pad != 1 || a->up || a->out_sy > 1 || a->out_sx > 1) return SRGANFD_OK;
  *handled = true;
  return a->dtype == SRGANFD_F16 ? dispatch_stream<f16_t>(a, k, stream) : dispatch_stream<bf16_t>(a, k, stream);
}

}  // namespace srganfd
#endif
