// conv3x3_ring.hip -- the 3x3 stride-1 convolutions of the hot path (every conv of a dense block, BSRGAN/model.py:42-46,
// the generator tail :340-355, the U-Net decoder :116-135 and VGG-19) as ONE resident workgroup per CU fed by an LDS-DMA ring.
//
// Why a second kernel (measured on conv_igemm.hip's 3x3 shape, profiles/r01_*): there every wave owns 2 rows x 32 channels, so
// each v_mfma_f32_32x32x16 needs 1.17 ds_read_b128 of operands and every staged byte crosses the VGPRs twice (global -> reg ->
// ds_write): at 100 % MFMA rate the LDS would be ~90 % busy, so the MFMA phase, the LDS commit and the loads cannot overlap and
// the kernel sits at 27-35 % of the MFMA roof whatever the pipeline around that tile looks like.  This kernel changes the
// quantities that bound it:
//   * tile = (WR*MR) rows x 32 pixels x (32*NR) output channels, wave = MR rows x 32 pixels x 32*NR channels: the A fragments of
//     MR+2 patch rows serve 3 kernel rows and NR channel tiles, the B fragments MR rows -> (MR+2+3*NR) / (3*MR*NR) reads per MFMA
//     (0.75 for 4 rows x 32 channels, 0.5 for 4 rows x 64 channels);
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR round trip, no ds_write);
//     the LDS image is lane-linear, so the bank swizzle is applied to the per-lane SOURCE address; zero padding comes from a
//     16-byte zero page (padding lanes point at it), so every lane of every piece is active;
//   * K is walked in stages of SCH (16 or 32) input channels through a ring of NBUF LDS buffers: stage c+NBUF-1 is issued while
//     stage c is in its MFMA phase, retired by a counted s_waitcnt vmcnt (never 0 in the steady state) and published by ONE
//     barrier per stage;
//   * one workgroup of 8 waves per CU (2 per SIMD, up to 256 VGPRs each) or two of 4.
// Everything else (weights packed by pack.hip, epilogue formula, planar / NHWC operand addressing, nearest x2 gather) is the
// contract of srganfd_conv2d (include/srganfd.h); conv_igemm.hip keeps every other kernel shape and the f32 parity mode.
#ifdef SRGANFD_EXPERIMENT   // LDS-DMA ring / persistent stream kernels: measured and rejected; kept for the A/B tools
#include "conv_common.hpp"

namespace srganfd {

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0u, 0u, 0u, 0u};

template <typename T, int MR_, int WR_, int NR_, int SCH_, int NBUF_>
struct RingCfg {
  static constexpr int MR = MR_, WR = WR_, NR = NR_, SCH = SCH_, NBUF = NBUF_;
  static constexpr int WAVES = WR, NTHR = 64 * WR;
  static constexpr int TH = WR * MR, TW = 32;
  static constexpr int PR = TH + 2, PC = TW + 2;
  static constexpr int KSTEPS = SCH / 16;                  // 32x32x16 k-steps per stage
  static constexpr int SLOTS = SCH / 8;                    // 16-byte slots per pixel and stage (2 or 4)
  static constexpr int PIXB = SCH * 2;
  static constexpr int NPIX = PR * PC;
  static constexpr int XP = (NPIX * PIXB + 1023) / 1024;   // patch pieces (1 KiB = one wave instruction)
  static constexpr int WP = NR * 9 * KSTEPS;               // weight pieces: one (channel tile, tap, k-step) fragment set each
  static constexpr int NPW = (XP + WP + WAVES - 1) / WAVES; // pieces per wave and stage (the last few are dummies)
  static constexpr int BUFB = NPW * WAVES * 1024;
  static constexpr int SPC = 32 / SCH;                     // stages per 32-channel group
  static constexpr int NB = 32 * NR;
  static constexpr int EPI_BYTES = TH * TW * 32 * 4;       // fp32 [pixel][32 channels] tile of one epilogue pass
  static constexpr int LDS_BYTES = NBUF * BUFB > EPI_BYTES ? NBUF * BUFB : EPI_BYTES;
  static constexpr int WG_PER_CU = (2 * LDS_BYTES <= 160 * 1024 && 2 * NTHR <= 1024) ? 2 : 1;
  static constexpr int MIN_WAVES_PER_SIMD = WG_PER_CU * WAVES / 4;
  static_assert(SCH == 16 || SCH == 32, "stage = 16 or 32 input channels");
  static_assert(NBUF >= 2 && NBUF <= 4, "ring depth");
  static_assert((NBUF - 2) * NPW <= 63, "vmcnt is 6 bits");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ unsigned ring_lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to LDS [dst, dst + 1 KiB), dst wave-uniform.
// Inline asm on purpose (cf. wgrad.hip): hipcc orders every later LDS read behind the builtin form with vmcnt(0), which would
// serialise the ring; an asm statement is outside its wait-count bookkeeping and the kernel counts vmcnt itself.
__device__ __forceinline__ void ring_glds16(const void* gsrc, unsigned dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst_uniform));
}
template <int N> __device__ __forceinline__ void ring_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// byte offset of (pixel, k-step ks, lane half h) inside a staged patch.  4 slots / pixel: slot XOR (pix>>2)&3 (conv_igemm.hip's
// image); 2 slots / pixel: slot XOR (pix>>3)&1 -- both make the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots.
template <int SLOTS> __device__ __forceinline__ int ring_xoff(int pix, int ks, int h) {
  if constexpr (SLOTS == 4) return pix * 64 + (((2 * ks + h) ^ ((pix >> 2) & 3)) << 4);
  else return pix * 32 + ((h ^ ((pix >> 3) & 1)) << 4);
}

template <typename T, int MR, int WR, int NR, int SCH, int NBUF>
__global__ __launch_bounds__(64 * WR, (RingCfg<T, MR, WR, NR, SCH, NBUF>::MIN_WAVES_PER_SIMD)) void conv3x3_ring_kernel(const ConvK a) {
  using C = RingCfg<T, MR, WR, NR, SCH, NBUF>;
  using Frag = typename FragAB<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int nb = bid % a.nNb;
  int t = bid / a.nNb;
  const int tx = t % a.tiles_x; t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int n = t / a.tiles_y;
  const int oy0 = ty * C::TH, ox0 = tx * C::TW;
  const int Hl = a.Hin << a.up, Wl = a.Win << a.up;
  const T* __restrict__ xg = (const T*)a.x + (size_t)n * a.Hin * a.Win * a.xC;                       // 64-bit image base (block-uniform)
  const char* __restrict__ wgp = (const char*)a.w + (size_t)nb * NR * a.nChunks * 18432;           // this workgroup's channel tiles
  const char* zero = (const char*)g_zero_page;

  // per-lane source of this wave's pieces: piece p = wave + k * WAVES.  Patch pieces: element offset inside the image (-1 =
  // padding -> zero page); weight pieces: byte offset of the (tile, tap, k-step) fragment set; the stage part is added at issue.
  int soff[C::NPW];
#pragma unroll
  for (int k = 0; k < C::NPW; ++k) {
    const int p = wave + k * C::WAVES;
    int v = -1;
    if (p < C::XP) {
      const int item = p * 64 + lane;
      const int pix = item / C::SLOTS, sl = item % C::SLOTS;
      const int c16 = C::SLOTS == 4 ? (sl ^ ((pix >> 2) & 3)) : (sl ^ ((pix >> 3) & 1));
      const int py = pix / C::PC, px = pix - py * C::PC;
      const int gy = oy0 - a.pad_y + py, gx = ox0 - a.pad_x + px;
      const bool ok = pix < C::NPIX && gy >= 0 && gy < Hl && gx >= 0 && gx < Wl && !SRGANFD_DBG(a.dbg, 1);
      if (ok) v = ((gy >> a.up) * a.Win + (gx >> a.up)) * a.x_ps + a.x_base + c16 * 8;
    } else if (p < C::XP + C::WP) {
      const int q = p - C::XP;
      const int nr = q / (9 * C::KSTEPS), rem = q % (9 * C::KSTEPS);   // rem = tap * KSTEPS + ks
      const int tap = rem / C::KSTEPS, ks = rem % C::KSTEPS;
      v = ((nr * a.nChunks * 9 + tap) * 2 + ks) * 1024 + lane * 16;
      if (SRGANFD_DBG(a.dbg, 2)) v = -1;
    }
    soff[k] = v;
  }
  const unsigned lds0 = ring_lds_addr(smem);
  auto issue = [&](int c, int b) __attribute__((always_inline)) {
    if (SRGANFD_DBG(a.dbg, 8)) return;          // experiment: no LDS-DMA at all
    const int g = c / C::SPC, hs = c % C::SPC;
    const int xs = g * a.x_cs + hs * SCH;                       // elements
    const int ws = g * 18432 + hs * C::KSTEPS * 1024;           // bytes
    const unsigned lb = lds0 + (unsigned)(b * C::BUFB);
#pragma unroll
    for (int k = 0; k < C::NPW; ++k) {
      const int p = wave + k * C::WAVES;
      const char* src = zero;
      if (p < C::XP) { if (soff[k] >= 0) src = (const char*)(xg + (soff[k] + xs)); }
      else if (p < C::XP + C::WP) { if (soff[k] >= 0) src = wgp + (soff[k] + ws); }
      ring_glds16(src, lb + (unsigned)(p * 1024));
    }
  };

  f32x16 acc[MR][NR];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;

  const int nS = a.nChunks * C::SPC;
  const int pix00 = (wave * MR) * C::PC + r;
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s)
    if (s < nS) issue(s, s);

  int b = 0;
  for (int c = 0; c < nS; ++c) {
    // stage c landed: leave the younger stages (at most NBUF-2 of them) in flight
    {
      const int younger = (nS - 1 - c) < (NBUF - 2) ? (nS - 1 - c) : (NBUF - 2);
      if constexpr (NBUF >= 4) { if (younger == 2) ring_wait_vmcnt<2 * C::NPW>(); }
      if constexpr (NBUF >= 3) { if (younger == 1) ring_wait_vmcnt<C::NPW>(); }
      if (younger == 0) ring_wait_vmcnt<0>();
    }
    __syncthreads();     // every wave's pieces of stage c are in LDS; every wave is done reading the buffer stage c+NBUF-1 overwrites
    if (c + NBUF - 1 < nS) {
      int bn = b + NBUF - 1; if (bn >= NBUF) bn -= NBUF;
      issue(c + NBUF - 1, bn);
    }
    const char* bx = smem + b * C::BUFB;
    const char* bw = bx + C::XP * 1024 + lane * 16;
    __builtin_amdgcn_s_setprio(1);
    if (!SRGANFD_DBG(a.dbg, 16))                // experiment: copy only
#pragma unroll
    for (int ks = 0; ks < C::KSTEPS; ++ks) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        Frag av[MR + 2];
#pragma unroll
        for (int rr = 0; rr < MR + 2; ++rr) av[rr] = *(const Frag*)(bx + ring_xoff<C::SLOTS>(pix00 + rr * C::PC + kx, ks, h));
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
          for (int q = 0; q < NR; ++q) {
            const Frag bq = *(const Frag*)(bw + ((q * 9 + ky * 3 + kx) * C::KSTEPS + ks) * 1024);
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[m][q] = mfma32<T>(av[m + ky], bq, acc[m][q]);
          }
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (++b == NBUF) b = 0;
  }

  // ---- epilogue (formula: srganfd.h).  Per 32-channel pass: accumulators -> fp32 LDS tile [pixel][32] -> 16 output bytes per
  // lane with the residual / mask tensors read by 16-byte loads (one pixel's 32 channels are contiguous in both layouts). ----
  if (SRGANFD_DBG(a.dbg, 4)) { if (acc[0][0][0] == 123.456f) ((float*)a.y)[0] = 1.f; return; }
  float alpha = a.alpha;
  if (a.alpha_dev) alpha *= *a.alpha_dev;
  float* tile = (float*)smem;
  const size_t img = (size_t)n * a.HoutF * a.WoutF;
  constexpr int ITEMS = C::TH * 32 * 4, EI = ITEMS / C::NTHR;
  static_assert(ITEMS % C::NTHR == 0, "epilogue items");
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int cbase = nb * C::NB + q * 32;
    __syncthreads();
    {
      const float bv = a.bias ? a.bias[cbase + r] : 0.f;
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float v = alpha * acc[m][q][i] + bv;
          if (a.act == SRGANFD_ACT_LRELU) v = v > 0.f ? v : v * a.slope;
          else if (a.act == SRGANFD_ACT_RELU) v = fmaxf(v, 0.f);
          tile[((wave * MR + m) * 32 + mfma32_row(i, lane)) * 32 + r] = v * a.post_scale;
        }
    }
    __syncthreads();
#pragma unroll 2
    for (int e = 0; e < EI; ++e) {
      const int item = tid + e * C::NTHR;
      const int pix = item >> 2, ck = item & 3;
      const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
      if (oy < a.Hout && ox < a.Wout) {
        const int p = oy * a.WoutF + ox;
        float v[8], tt[8];
        {
          const f32x4 t0 = *(const f32x4*)(tile + pix * 32 + ck * 8), t1 = *(const f32x4*)(tile + pix * 32 + ck * 8 + 4);
          v[0] = t0[0]; v[1] = t0[1]; v[2] = t0[2]; v[3] = t0[3]; v[4] = t1[0]; v[5] = t1[1]; v[6] = t1[2]; v[7] = t1[3];
        }
        const int cch = cbase + ck * 8;
        auto addr = [&](const void* base, int Cs, int c0, int ps, int gs) -> T* {
          const int cc = c0 + cch;
          return (T*)base + img * Cs + (p * ps + (cc >> 5) * gs + (cc & 31));
        };
        if (a.y2) *(u32x4*)addr(a.y2, a.y2C, a.y2_c0, a.y2_ps, a.y2_gs) = pack8<T>(v);
        if (a.r1) { unpack8<T>(*(const u32x4*)addr(a.r1, a.r1C, a.r1_c0, a.r1_ps, a.r1_gs), tt);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += a.r1s * tt[j]; }
        if (a.r2) { unpack8<T>(*(const u32x4*)addr(a.r2, a.r2C, a.r2_c0, a.r2_ps, a.r2_gs), tt);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += a.r2s * tt[j]; }
        if (a.mask) { unpack8<T>(*(const u32x4*)addr(a.mask, a.mC, a.m_c0, a.m_ps, a.m_gs), tt);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= tt[j] > 0.f ? 1.f : a.mask_slope; }
        *(u32x4*)addr(a.y, a.yC, a.y_c0, a.y_ps, a.y_gs) = pack8<T>(v);
      }
    }
  }
}

template <typename T, int MR, int WR, int NR, int SCH, int NBUF>
static int launch_ring(const ConvK& k, int cout, hipStream_t stream) {
  using C = RingCfg<T, MR, WR, NR, SCH, NBUF>;
  auto kern = conv3x3_ring_kernel<T, MR, WR, NR, SCH, NBUF>;
  if (g_describe) { snprintf(g_describe, g_describe_len, "conv3x3_ring_kernel<%s,MR=%d,WR=%d,NR=%d,SCH=%d,NBUF=%d>", dtype_name<T>(), MR, WR, NR, SCH, NBUF); return SRGANFD_OK; }
  static unsigned long long attr_done = 0;                     // per-device bit: the attribute belongs to the device's code object
  if (!g_dry_run) {
    int dev = 0;
    SRGANFD_HIP_CHECK(hipGetDevice(&dev));
    if (!(attr_done >> (dev & 63) & 1ULL)) {
      SRGANFD_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
      attr_done |= 1ULL << (dev & 63);
    }
  }
  ConvK kk = k;
  kk.nNb = cout / C::NB;
  kk.tiles_x = ceil_div(k.Wout, C::TW);
  kk.tiles_y = ceil_div(k.Hout, C::TH);
  const long long nblk = (long long)k.N * kk.tiles_x * kk.tiles_y * kk.nNb;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return set_err(SRGANFD_EINVAL, "conv2d: bad grid %lld", nblk);
  SRGANFD_LAUNCH(kern, dim3((unsigned)nblk), dim3(C::NTHR), C::LDS_BYTES, stream, kk);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Persistent form ("stream"): measured on the kernel above (profiles/r02_conv_ablation.txt), its three parts ADD -- MFMA phase
// 18.5 us, LDS-DMA 18.4 us, epilogue + tile turn-around 19.6 us of a 55.9 us launch (128 -> 32 channels, B = 32) -- because
// (a) a wave issues its DMA pieces back to back right after the barrier, in front of its own MFMA phase, at ~100-180 cycles of
// issue time per piece, (b) with one workgroup per CU nothing runs during a tile's prologue (first stage in flight), its LDS
// transposition epilogue and the drain of its stores.  Here
//   * a workgroup is persistent and walks tiles b, b + G, ... (same XCD: G is a multiple of 8); the ring does not stop at a tile
//     boundary -- the stages of the next tile are issued while the current tile's last stages are in their MFMA phase;
//   * the pieces of a stage are issued BETWEEN the MFMA groups of the stage that is being computed (VMEM issue in the shadow of
//     the matrix pipe), not in front of them;
//   * the MFMA operands are swapped (A = weights, B = pixels), so a lane's accumulator registers hold consecutive output channels
//     of ONE pixel: four v_permlane32_swap per eight registers give every lane 8 consecutive channels = one 16-byte store, with
//     no LDS tile and no barrier; the epilogue of tile t runs after the barrier that opens tile t+1, so its stores are a full
//     stage old when the next counted vmcnt has to look past them.
struct StreamCfgBase { static constexpr int BIAS_BYTES = 2048; };   // fp32 bias of up to 512 output channels, staged once

template <typename T, int MR, int WR, int NR, int SCH, int NBUF>
__global__ __launch_bounds__(64 * WR, (WR / 4)) void conv3x3_stream_kernel(const ConvK a) {
  using C = RingCfg<T, MR, WR, NR, SCH, NBUF>;
  using Frag = typename FragAB<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lbias = (float*)smem;
  char* ring = smem + StreamCfgBase::BIAS_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nT = a.N * a.tiles_x * a.tiles_y * a.nNb;
  const int G = gridDim.x;
  const int Hl = a.Hin << a.up, Wl = a.Win << a.up;
  const char* zero = (const char*)g_zero_page;
  const int nS = a.nChunks * C::SPC;

  for (int i = tid; i < a.nNb * C::NB; i += C::NTHR) lbias[i] = a.bias ? a.bias[i] : 0.f;   // published by the first stage barrier

  // ---- issue stream: (tile, stage) pairs, NBUF-1 stages ahead of the compute stream ----
  int vi = blockIdx.x, ci = 0, bi = 0, issued = 0;
  const T* xg_i = nullptr;
  const char* wg_i = nullptr;
  int soff[C::NPW];
  auto set_tile = [&](int v) __attribute__((always_inline)) {
    int t = xcd_remap(v, nT);
    const int nb = t % a.nNb; t /= a.nNb;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int oy0 = ty * C::TH, ox0 = tx * C::TW;
    xg_i = (const T*)a.x + (size_t)n * a.Hin * a.Win * a.xC;
    wg_i = (const char*)a.w + (size_t)nb * NR * a.nChunks * 18432;
#pragma unroll
    for (int k = 0; k < C::NPW; ++k) {
      const int p = wave + k * C::WAVES;
      int v2 = -1;
      if (p < C::XP) {
        const int item = p * 64 + lane;
        const int pix = item / C::SLOTS, sl = item % C::SLOTS;
        const int c16 = C::SLOTS == 4 ? (sl ^ ((pix >> 2) & 3)) : (sl ^ ((pix >> 3) & 1));
        const int py = pix / C::PC, px = pix - py * C::PC;
        const int gy = oy0 - a.pad_y + py, gx = ox0 - a.pad_x + px;
        const bool ok = pix < C::NPIX && gy >= 0 && gy < Hl && gx >= 0 && gx < Wl && !SRGANFD_DBG(a.dbg, 1);
        if (ok) v2 = ((gy >> a.up) * a.Win + (gx >> a.up)) * a.x_ps + a.x_base + c16 * 8;
      } else if (p < C::XP + C::WP) {
        const int q = p - C::XP;
        const int nr = q / (9 * C::KSTEPS), rem = q % (9 * C::KSTEPS);
        const int tap = rem / C::KSTEPS, ks = rem % C::KSTEPS;
        v2 = ((nr * a.nChunks * 9 + tap) * 2 + ks) * 1024 + lane * 16;
        if (SRGANFD_DBG(a.dbg, 2)) v2 = -1;
      }
      soff[k] = v2;
    }
  };
  const unsigned lds0 = ring_lds_addr(ring);
  auto issue_piece = [&](int k) __attribute__((always_inline)) {
    if (vi >= nT || SRGANFD_DBG(a.dbg, 8)) return;
    const int g = ci / C::SPC, hs = ci % C::SPC;
    const int p = wave + k * C::WAVES;
    const char* src = zero;
    if (p < C::XP) { if (soff[k] >= 0) src = (const char*)(xg_i + (soff[k] + g * a.x_cs + hs * SCH)); }
    else if (p < C::XP + C::WP) { if (soff[k] >= 0) src = wg_i + (soff[k] + g * 18432 + hs * C::KSTEPS * 1024); }
    ring_glds16(src, lds0 + (unsigned)(bi * C::BUFB + p * 1024));
  };
  auto issue_advance = [&]() __attribute__((always_inline)) {
    if (vi >= nT) return;
    ++issued;
    if (++bi == NBUF) bi = 0;
    if (++ci == nS) { ci = 0; vi += G; if (vi < nT) set_tile(vi); }
  };
  if (vi < nT) set_tile(vi);
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s) {
#pragma unroll
    for (int k = 0; k < C::NPW; ++k) issue_piece(k);
    issue_advance();
  }

  f32x16 acc[MR][NR];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;

  // ---- epilogue of one finished tile: registers -> global, no LDS, no barrier ----
  float alpha = a.alpha;
  if (a.alpha_dev) alpha *= *a.alpha_dev;
  auto epilogue = [&](int v) __attribute__((always_inline)) {
    if (SRGANFD_DBG(a.dbg, 4)) { if (acc[0][0][0] == 123.456f) ((float*)a.y)[0] = 1.f; return; }
    int t = xcd_remap(v, nT);
    const int nb = t % a.nNb; t /= a.nNb;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const size_t img = (size_t)n * a.HoutF * a.WoutF;
    const int ox = tx * C::TW + r;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int oy = ty * C::TH + wave * MR + m;
      const bool ok = oy < a.Hout && ox < a.Wout;
      const int p = oy * a.WoutF + ox;
#pragma unroll
      for (int q = 0; q < NR; ++q) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          // registers 8kk..8kk+3 (channels 16kk + 4h + j) and 8kk+4..8kk+7 (16kk + 8 + 4h + j) -> 8 consecutive channels 16kk + 8h + j
          float v8[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[m][q][8 * kk + j]), __float_as_uint(acc[m][q][8 * kk + 4 + j]), false, false);
            v8[j] = __uint_as_float(sw[0]);
            v8[4 + j] = __uint_as_float(sw[1]);
          }
          const int cch = nb * C::NB + q * 32 + 16 * kk + 8 * h;
          const f32x4 b0 = *(const f32x4*)(lbias + cch), b1 = *(const f32x4*)(lbias + cch + 4);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float x = alpha * v8[j] + (j < 4 ? b0[j & 3] : b1[j & 3]);
            if (a.act == SRGANFD_ACT_LRELU) x = x > 0.f ? x : x * a.slope;
            else if (a.act == SRGANFD_ACT_RELU) x = fmaxf(x, 0.f);
            v8[j] = x * a.post_scale;
          }
          if (ok) {
            float tt[8];
            auto addr = [&](const void* base, int Cs, int c0, int ps, int gs) -> T* {
              const int cc = c0 + cch;
              return (T*)base + img * Cs + (p * ps + (cc >> 5) * gs + (cc & 31));
            };
            if (a.y2) *(u32x4*)addr(a.y2, a.y2C, a.y2_c0, a.y2_ps, a.y2_gs) = pack8<T>(v8);
            if (a.r1) { unpack8<T>(*(const u32x4*)addr(a.r1, a.r1C, a.r1_c0, a.r1_ps, a.r1_gs), tt);
#pragma unroll
              for (int j = 0; j < 8; ++j) v8[j] += a.r1s * tt[j]; }
            if (a.r2) { unpack8<T>(*(const u32x4*)addr(a.r2, a.r2C, a.r2_c0, a.r2_ps, a.r2_gs), tt);
#pragma unroll
              for (int j = 0; j < 8; ++j) v8[j] += a.r2s * tt[j]; }
            if (a.mask) { unpack8<T>(*(const u32x4*)addr(a.mask, a.mC, a.m_c0, a.m_ps, a.m_gs), tt);
#pragma unroll
              for (int j = 0; j < 8; ++j) v8[j] *= tt[j] > 0.f ? 1.f : a.mask_slope; }
            if (SRGANFD_DBG(a.dbg, 64)) __builtin_nontemporal_store(pack8<T>(v8), (u32x4*)addr(a.y, a.yC, a.y_c0, a.y_ps, a.y_gs));
            else *(u32x4*)addr(a.y, a.yC, a.y_c0, a.y_ps, a.y_gs) = pack8<T>(v8);
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int q = 0; q < NR; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;
  };

  // ---- compute stream ----
  const int pix00 = (wave * MR) * C::PC + r;
  constexpr int STEPS = 3 * C::KSTEPS;                  // (k-step, kernel column) groups of a stage
  int bc = 0, consumed = 0, vprev = -1;
#ifdef SRGANFD_EXPERIMENT
  unsigned long long* stamp = (a.stamps && blockIdx.x < 8) ? a.stamps + ((size_t)blockIdx.x * 8 + wave) * 64 * 4 : nullptr;
  int stamp_i = 0;
#define RING_STAMP(j) do { if (stamp && stamp_i < 64 && lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamp[stamp_i * 4 + (j)] = t_; } } while (0)
#else
#define RING_STAMP(j) do {} while (0)
#endif
  for (int vc = blockIdx.x; vc < nT; vc += G) {
    for (int c = 0; c < nS; ++c) {
      RING_STAMP(0);
      {
        const int younger = issued - consumed - 1;       // stages issued after the one that must have landed now (0 .. NBUF-2)
        if constexpr (NBUF >= 4) { if (younger >= 2) ring_wait_vmcnt<2 * C::NPW>(); }
        if constexpr (NBUF >= 3) { if (younger == 1) ring_wait_vmcnt<C::NPW>(); }
        if (younger <= 0) ring_wait_vmcnt<0>();
      }
      RING_STAMP(1);
      __syncthreads();   // the stage is in LDS for every wave; every wave has left the buffer the next issue overwrites
      RING_STAMP(2);
      if (c == 0 && vprev >= 0) epilogue(vprev);
      const char* bx = ring + bc * C::BUFB;
      const char* bw = bx + C::XP * 1024 + lane * 16;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const int ks = st / 3, kx = st % 3;
        if (!SRGANFD_DBG(a.dbg, 16)) {
          Frag av[MR + 2];
#pragma unroll
          for (int rr = 0; rr < MR + 2; ++rr) av[rr] = *(const Frag*)(bx + ring_xoff<C::SLOTS>(pix00 + rr * C::PC + kx, ks, h));
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {
              const Frag bq = *(const Frag*)(bw + ((q * 9 + ky * 3 + kx) * C::KSTEPS + ks) * 1024);
#pragma unroll
              for (int m = 0; m < MR; ++m) acc[m][q] = mfma32<T>(bq, av[m + ky], acc[m][q]);   // swapped: rows = channels, columns = pixels
            }
          }
        }
        // this group's share of the next stage's pieces (VMEM issue behind the MFMAs just queued)
#pragma unroll
        for (int k = st * C::NPW / STEPS; k < (st + 1) * C::NPW / STEPS; ++k) issue_piece(k);
      }
      __builtin_amdgcn_s_setprio(0);
      RING_STAMP(3);
#ifdef SRGANFD_EXPERIMENT
      ++stamp_i;
#endif
      issue_advance();
      ++consumed;
      if (++bc == NBUF) bc = 0;
    }
    vprev = vc;
  }
  if (vprev >= 0) epilogue(vprev);
}

template <typename T, int MR, int WR, int NR, int SCH, int NBUF>
static int launch_stream(const ConvK& k, int cout, hipStream_t stream) {
  using C = RingCfg<T, MR, WR, NR, SCH, NBUF>;
  constexpr int LDS = StreamCfgBase::BIAS_BYTES + NBUF * C::BUFB;
  static_assert(LDS <= 160 * 1024, "LDS");
  static_assert(WR % 4 == 0, "whole waves per SIMD");
  auto kern = conv3x3_stream_kernel<T, MR, WR, NR, SCH, NBUF>;
  if (g_describe) { snprintf(g_describe, g_describe_len, "conv3x3_stream_kernel<%s,MR=%d,WR=%d,NR=%d,SCH=%d,NBUF=%d>", dtype_name<T>(), MR, WR, NR, SCH, NBUF); return SRGANFD_OK; }
  static unsigned long long attr_done = 0;
  static int n_cu[64] = {0};
  int dev = 0;
  if (!g_dry_run) {
    SRGANFD_HIP_CHECK(hipGetDevice(&dev));
    if (!(attr_done >> (dev & 63) & 1ULL)) {
      SRGANFD_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
      SRGANFD_HIP_CHECK(hipDeviceGetAttribute(&n_cu[dev & 63], hipDeviceAttributeMultiprocessorCount, dev));
      attr_done |= 1ULL << (dev & 63);
    }
  }
  ConvK kk = k;
  kk.nNb = cout / C::NB;
  kk.tiles_x = ceil_div(k.Wout, C::TW);
  kk.tiles_y = ceil_div(k.Hout, C::TH);
  const long long nT = (long long)k.N * kk.tiles_x * kk.tiles_y * kk.nNb;
  if (nT <= 0 || nT > 0x7fffffffLL) return set_err(SRGANFD_EINVAL, "conv2d: bad grid %lld", nT);
  if (cout * 4 > StreamCfgBase::BIAS_BYTES) return set_err(SRGANFD_EINVAL, "conv2d: stream kernel takes at most 512 output channels");
  const int per_cu = (2 * LDS <= 160 * 1024 && 2 * C::NTHR <= 1024) ? 2 : 1;
  int cus = n_cu[dev & 63] > 0 ? n_cu[dev & 63] : 256;
  long long grid = (long long)cus * per_cu;
  grid -= grid % 8;                                   // tiles of one workgroup stay on one XCD (b and b + G share b & 7)
  if (grid > nT) grid = nT;
  SRGANFD_LAUNCH(kern, dim3((unsigned)grid), dim3(C::NTHR), LDS, stream, kk);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

// SRGANFD_RING (environment, read once): 0 = never use this kernel; unset / 1 = default configuration per shape; other values
// select an experiment configuration (tools/kbench.py A/B runs).
static int ring_mode() {
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("SRGANFD_RING"); mode = e ? atoi(e) : 1; }
  return mode;
}
extern "C" void srganfd_set_ring_mode(int mode);   // kbench: switch configurations inside one process

template <typename T>
static int ring_dispatch(const srganfd_conv_args* a, const ConvK& k, hipStream_t s, int mode, bool* handled) {
  const bool wide = (a->cout % 64) == 0;
  *handled = true;
  switch (mode) {
    case 1:   // library default (profiles/r02_kbench_*.txt): 64-channel tiles on the ring kernel -- 16 x 32 pixels, 4 waves of
              // 4 rows x 64 channels (0.5 fragment reads per MFMA), two workgroups per CU: 5-10 % under conv_igemm's 8-row tiles on the
              // tail / discriminator / VGG shapes; the 32-channel dense-block convs stay on conv_igemm (every ring / stream
              // configuration measured 8-15 % slower there)
      if (wide) return launch_ring<T, 4, 4, 2, 16, 2>(k, a->cout, s);
      break;
#ifdef SRGANFD_EXPERIMENT
    case 2:   // 16 x 32 pixel tiles, 4 waves, two workgroups per CU, both channel widths
      return wide ? launch_ring<T, 4, 4, 2, 16, 2>(k, a->cout, s) : launch_ring<T, 4, 4, 1, 16, 2>(k, a->cout, s);
    case 3:   // 32 x 32 pixel tiles, 8 waves, one workgroup per CU
      return wide ? launch_ring<T, 4, 8, 2, 16, 2>(k, a->cout, s) : launch_ring<T, 4, 8, 1, 16, 3>(k, a->cout, s);
    case 4:   // persistent stream, 32 x 32 pixel tiles
      return wide ? launch_stream<T, 2, 8, 2, 16, 3>(k, a->cout, s) : launch_stream<T, 4, 8, 1, 16, 3>(k, a->cout, s);
    case 5:   // persistent stream, 16 x 32 pixel tiles, deeper ring
      return wide ? launch_stream<T, 2, 8, 2, 16, 3>(k, a->cout, s) : launch_stream<T, 2, 8, 1, 16, 4>(k, a->cout, s);
#endif
    default: break;
  }
  *handled = false;
  return SRGANFD_OK;
}

static int g_ring_mode_override = -1;
extern int g_igemm_variant;
extern int g_wgrad_variant;
extern "C" void srganfd_set_ring_mode(int mode) {
  if (mode >= 0 && (mode & 0x1000)) { g_wgrad_variant = (mode >> 9) & 3; return; }      // A/B of the weight-gradient loop variants
  g_ring_mode_override = mode; g_igemm_variant = (mode & 0xff) == 7 ? 7 : 0;
}

int conv3x3_ring_try(const srganfd_conv_args* a, const ConvK& k, hipStream_t stream, bool* handled) {
  *handled = false;
  int mode = g_ring_mode_override >= 0 ? g_ring_mode_override : ring_mode();
  const bool force = (mode & 0x100) != 0;      // tests: small shapes through this kernel too
  mode &= 0xff;
  if (mode == 0 || a->ksize != 3 || a->stride != 1 || !k.fast_epi || k.osy != 1 || k.osx != 1) return SRGANFD_OK;
  if (a->dtype != SRGANFD_BF16 && a->dtype != SRGANFD_F16) return SRGANFD_OK;
  if (mode >= 4 && a->cout > 512) return SRGANFD_OK;
  // The choice depends on the IMAGE only, never on the batch: an image's pixels are then summed in the same order whatever batch it
  // sits in (tests/test_fullsize_gpu.py checks that bitwise).  Small images keep conv_igemm's 8-row tiles.
  if ((a->h_out < 64 || a->w_out < 64) && !force) return SRGANFD_OK;
  if (a->dtype == SRGANFD_BF16) return ring_dispatch<bf16_t>(a, k, stream, mode, handled);
  return ring_dispatch<f16_t>(a, k, stream, mode, handled);
}

}  // namespace srganfd

#endif  // SRGANFD_EXPERIMENT
