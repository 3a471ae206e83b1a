// wgrad.hip -- weight/bias gradients of the NHWC convolutions on the CDNA4 matrix cores.
//
// Replaces the weight/bias outputs of ATen convolution_backward for the reference's convs
// (autograd of BSRGAN/model.py:42-46,102-135,325-355 as driven by train_bsrgan.py:420,430,463).
//
//   dW[tap][ci][co] = sum_p X[p (+) tap][ci] * dY[p][co]          (contraction over PIXELS)
//
// MI355X mapping: the contraction index is the pixel, but activations are stored NHWC (channel
// fastest) for the forward/dgrad kernels.  Instead of keeping a second, transposed copy of every
// activation, the tiles are staged NHWC into LDS and the MFMA operands are read with gfx950's
// transposing LDS read (ds_read_b64_tr_b16: a 4-pixel x 16-channel block delivered channel-major),
// so A = X^T and B = dY^T fragments cost one LDS instruction pair each and no extra HBM traffic.
// One wavefront owns one (32 ci x 32 co) block for all taps (9 x 16 accumulator registers); four
// wavefronts of a workgroup share the staged X / dY tiles.  Pixel tiles are split over workgroups;
// each wave writes an fp32 partial slab and a second kernel reduces the slabs deterministically
// (no float atomics: results are bitwise reproducible) straight into the NCHW fp32 gradient.
// f32 mode (parity) uses v_mfma_f32_32x32x2_f32 with plain ds_read_b32 (channels on the lanes).
#include "conv_common.hpp"
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace srganfd {

static constexpr int kWgMagic = 0x57475244;  // 'WGRD'

static constexpr int kMaxWaves = 12;
struct WgWave { int active, ci_rel, co_rel, ks_idx, ks_n, tap0, slab_base, bias_slab; };
struct WgGroup { int x_c0, x_units, dy_c0, dy_units;
                 int x_u1, dy_u1;      // the second staged unit sits this many 32-channel groups behind the first (1 = the 64-channel range)
                 WgWave w[kMaxWaves]; };
struct WgTask {
  long long dw_off, db_off, alpha_off;
  int co_dst, ci_dst, co_base, ci_base, tap0, ntap, ksize, slab_base, nslabs, bias_slab;
  float alpha, beta;
};
struct WgHeader {
  int magic, dtype, N, Hin, Win, up, ks, stride, pad, Hout, Wout;
  int ngroups, ntasks, S, x_upad, dy_upad, lds_bytes, ntiles, tiles_x, tiles_y, ntap_wave;
  int dma_ok, pad_;   // LDS-DMA double buffering: bf16, every group stages the full pitch, two buffers fit
  long long nslabs_total, bias_slab_off /* floats */, nbias_slabs;
  long long groups_off, tasks_off, total_bytes;
};

struct WgK {
  const void* x; const void* dy; float* slabs; float* bslabs;
  const WgGroup* groups;
  int xC, x_c0v, dyC, dy_c0v;
  int x_ps, x_gs, dy_ps, dy_gs;   // pixel / 32-channel-group strides in elements: NHWC (C, 32) or planar groups (32, H*W*32), see srganfd_view
  int N, Hin, Win, up, pad, Hout, Wout, S, x_upad, dy_upad, ntiles, tiles_x, tiles_y, ngroups;
};


typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// two transposed 4-element reads -> one 8-element MFMA fragment: pure register concatenation (no VALU)
template <typename T> __device__ __forceinline__ typename FragAB<T>::type cat_frag(s16x4 lo, s16x4 hi) {
  const u32x2 l = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  const u32x4 v = {l.x, l.y, h2.x, h2.y};
  return __builtin_bit_cast(typename FragAB<T>::type, v);
}

// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to LDS [dst, dst + 1 KiB) (dst wave-uniform).
// Inline asm on purpose: hipcc orders every later LDS read behind a __builtin_amdgcn_global_load_lds with a
// vmcnt(0), which serialises the copy with the MFMA phase it is meant to overlap; an asm statement is outside its
// wait-count bookkeeping, so the kernel waits explicitly (s_waitcnt vmcnt(0) before the barrier that publishes the buffer).
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
  // m0 is declared clobbered instead of saved and restored around every piece (nothing else in these kernels lives in m0: LDS
  // instructions do not need it on gfx9+)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               : : "v"(gsrc), "s"(dst) : "m0");   // no "memory" clobber: it pins every by-reference lambda capture to scratch,
                                                   // and scratch loads share vmcnt with the copies; ordering comes from the
                                                   // explicit vmcnt(0) + barrier that publish a buffer
#pragma clang diagnostic pop
}

template <int KS, int STRIDE> struct WgTile { static constexpr int TH = (STRIDE == 1) ? 8 : 4; };

template <int KS> struct WgWaves { static constexpr int NW = (KS == 3) ? 12 : (KS == 4 ? 8 : 4); };
// DMA builds add loader wavefronts (one per SIMD) that only issue the LDS-DMA pieces of the next tile.  Measured on the
// dense-block launch (B=32, 128x128): register staging 357 us; every wave issuing its share right after the barrier
// 355 us (the copy's issue time and the LDS-bound MFMA phase add up: all waves sit in the same phase between two
// barriers); the shares spread over the MFMA rows 396 us; four specialised loaders beside the compute waves 314 us.
static constexpr int kLoaderWaves = 4;

// One wave = one (32 ci x 32 co) block x ONE kernel row (KS taps, KS*16 accumulator registers), so a
// 3x3 workgroup runs 12 waves (3 per SIMD, <=168 VGPRs each) over the same staged tiles.
// XP / YP = LDS row pitch of the x / dy tiles in 32-channel units (compile-time so that every LDS address in
// the MFMA loop is table + wave-uniform row offset + immediate).
// DMA = 1 (bf16, every group stages exactly XP / YP units): the tiles are double-buffered in LDS and filled by LDS-DMA
// (global_load_lds_dwordx4: wave-uniform LDS base + lane*16, so the LDS image is lane-linear and the 64-byte-unit swizzle
// is applied to the per-lane SOURCE address); tile t+1 lands while tile t is in the MFMA phase, one barrier per tile,
// no staging registers.  DMA = 0: global -> register -> LDS staging through one buffer.
// VAR: 0 = v_mfma_f32_32x32x16, one pair of transposed x reads per kernel column (every shape but the next); 3 = the 3x3 stride-1 16-bit
// DMA kernel: v_mfma_f32_16x16x32 on three transposed x reads per half row whose register shifts give the three kernel columns (the
// intermediate variants 1 / 2 of profiles/r02_wgrad_variants.txt live in tools/experiments/r3_src).
template <typename T, int KS, int STRIDE, int XP, int YP, bool DMA, int VAR>
__global__ __launch_bounds__(64 * (WgWaves<KS>::NW + (DMA ? kLoaderWaves : 0))) void wgrad_kernel(const WgK a) {
  constexpr int TH = WgTile<KS, STRIDE>::TH;
  constexpr int NTHR = 64 * WgWaves<KS>::NW;
  constexpr int NT = KS;
  constexpr int PR = (TH - 1) * STRIDE + KS, PC = 31 * STRIDE + KS;
  constexpr int UB = 32 * (int)sizeof(T);   // bytes of one 32-channel unit
  constexpr int CPU = UB / 16;              // 16-byte chunks per unit (4 bf16 / 8 f32)
  constexpr int CPU_SH = (CPU == 4) ? 2 : 3;
  constexpr int E16 = 16 / (int)sizeof(T);
  // staging items per thread for the largest group (2 units each side)
  constexpr int XI = (PR * PC * 2 * CPU + NTHR - 1) / NTHR;
  constexpr int YI = (TH * 32 * 2 * CPU + NTHR - 1) / NTHR;
  constexpr bool kPrefetch = sizeof(T) == 2 && !DMA;   // bf16 register staging: next tile's loads are issued before the MFMA phase
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // 1-D grid.  Blocks b and b+8 share an XCD (private L2): remap so that each XCD owns a contiguous range of
  // work ids, with the channel group as the FAST index -- the workgroups that stream the same pixel tiles
  // (different channel groups of one split) then run together on one XCD and re-read x / dy from its L2.
  int wgid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = wgid & 7, q = nwg >> 3, rr = nwg & 7;
    wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (wgid >> 3);
  }
  const int grp = wgid % a.ngroups, split = wgid / a.ngroups;
  const WgGroup& G = a.groups[grp];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WgWave W = G.w[wave < WgWaves<KS>::NW ? wave : 0];
  const int r = lane & 31, h = lane >> 5;
  const int xu_sh = G.x_units == 2 ? 1 : 0, yu_sh = G.dy_units == 2 ? 1 : 0;   // units staged by this group (1 or 2)
  constexpr int xp_sh = XP == 2 ? 1 : 0, yp_sh = YP == 2 ? 1 : 0;                // pitch (swizzle follows the pitch)
  constexpr int xRowB = UB * XP, dyRowB = UB * YP;
  constexpr int kBufBytes = PR * PC * xRowB + TH * 32 * dyRowB;   // one (x tile, dy tile) buffer
  char* ldsX = smem;
  char* ldsY = smem + PR * PC * xRowB;
  const int Hl = a.Hin << a.up, Wl = a.Win << a.up;
  const int xcb = a.x_c0v + G.x_c0, ycb = a.dy_c0v + G.dy_c0;     // first channel this group stages
  const T* __restrict__ xg = (const T*)a.x + ((xcb >> 5) * (size_t)a.x_gs + (xcb & 31));
  const T* __restrict__ dyg = (const T*)a.dy + ((ycb >> 5) * (size_t)a.dy_gs + (ycb & 31));
  const int x_gs2 = a.x_gs * G.x_u1, dy_gs2 = a.dy_gs * G.dy_u1;   // distance of the group's two staged units

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float bsum = 0.f;
  // VAR 3 (v_mfma_f32_16x16x32): [kernel column][input-channel half][output-channel half], 48 registers like acc[3] above
  f32x4_t acc16[3][2][2];
  float bsum16[2] = {0.f, 0.f};
  f32x4_t bacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // VAR 3: bias sums as MFMA results (all rows equal)
  if constexpr (VAR == 3) {
#pragma unroll
    for (int q = 0; q < 12; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc16[q / 4][(q >> 1) & 1][q & 1][i] = 0.f;
  }

  const int xItems = (PR * PC * CPU) << xu_sh;
  const int yItems = (TH * 32 * CPU) << yu_sh;
  const int rows_per = TH / W.ks_n;
  auto swz = [](int ush, int pix) { return ush ? ((pix >> 1) & 1) : 0; };

  auto tile_origin = [&](int tile, int& n, int& oy0, int& ox0) {
    const int tx = tile % a.tiles_x; tile /= a.tiles_x;
    const int ty = tile % a.tiles_y;
    n = tile / a.tiles_y; oy0 = ty * TH; ox0 = tx * 32;
  };
  auto load_x = [&](int i, int n, int oy0, int ox0) -> u32x4 {
    const int item = tid + i * NTHR;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (item < xItems) {
      const int pix = item >> (CPU_SH + xu_sh), c16 = item & ((CPU << xu_sh) - 1);
      const int py = pix / PC, px = pix - py * PC;
      const int gy = oy0 * STRIDE - a.pad + py, gx = ox0 * STRIDE - a.pad + px;
      if (gy >= 0 && gy < Hl && gx >= 0 && gx < Wl)
        v = *(const u32x4*)(xg + (size_t)n * a.Hin * a.Win * a.xC + (((gy >> a.up) * a.Win + (gx >> a.up)) * a.x_ps + ((c16 * E16) >> 5) * x_gs2 + ((c16 * E16) & 31)));   // 64-bit image base + 32-bit offset (host-checked)
    }
    return v;
  };
  auto store_x = [&](int i, u32x4 v) {
    const int item = tid + i * NTHR;
    if (item < xItems) {
      const int pix = item >> (CPU_SH + xu_sh), c16 = item & ((CPU << xu_sh) - 1);
      const int unit = c16 >> CPU_SH, w16 = c16 & (CPU - 1);
      const int f = sizeof(T) == 2 ? swz(xp_sh, pix) : 0;
      *(u32x4*)(ldsX + pix * xRowB + ((unit ^ f) * UB) + w16 * 16) = v;
    }
  };
  auto load_y = [&](int i, int n, int oy0, int ox0) -> u32x4 {
    const int item = tid + i * NTHR;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (item < yItems) {
      const int pix = item >> (CPU_SH + yu_sh), c16 = item & ((CPU << yu_sh) - 1);
      const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
      if (oy < a.Hout && ox < a.Wout) v = *(const u32x4*)(dyg + (size_t)n * a.Hout * a.Wout * a.dyC + ((oy * a.Wout + ox) * a.dy_ps + ((c16 * E16) >> 5) * dy_gs2 + ((c16 * E16) & 31)));
    }
    return v;
  };
  auto store_y = [&](int i, u32x4 v) {
    const int item = tid + i * NTHR;
    if (item < yItems) {
      const int pix = item >> (CPU_SH + yu_sh), c16 = item & ((CPU << yu_sh) - 1);
      const int unit = c16 >> CPU_SH, w16 = c16 & (CPU - 1);
      const int f = sizeof(T) == 2 ? swz(yp_sh, pix) : 0;
      *(u32x4*)(ldsY + pix * dyRowB + ((unit ^ f) * UB) + w16 * 16) = v;
    }
  };
  constexpr int XR = kPrefetch ? XI : 1, YR = kPrefetch ? YI : 1;
  u32x4 xr[XR], yr[YR];
  auto prefetch = [&](int tile) {
    if constexpr (kPrefetch) {
      int n, oy0, ox0;
      tile_origin(tile, n, oy0, ox0);
#pragma unroll
      for (int i = 0; i < XI; ++i) xr[i] = load_x(i, n, oy0, ox0);
#pragma unroll
      for (int i = 0; i < YI; ++i) yr[i] = load_y(i, n, oy0, ox0);
    }
  };
  auto commit = [&](int tile) {
    if constexpr (kPrefetch) {
#pragma unroll
      for (int i = 0; i < XI; ++i) store_x(i, xr[i]);
#pragma unroll
      for (int i = 0; i < YI; ++i) store_y(i, yr[i]);
    } else {
      int n, oy0, ox0;
      tile_origin(tile, n, oy0, ox0);
#pragma unroll 4
      for (int i = 0; i < XI; ++i) store_x(i, load_x(i, n, oy0, ox0));
#pragma unroll 4
      for (int i = 0; i < YI; ++i) store_y(i, load_y(i, n, oy0, ox0));
    }
  };

  // LDS-DMA fill (DMA builds only).  Item = one 16-byte chunk; a piece = one wave instruction = 64 consecutive items =
  // 1 KiB of the lane-linear image.  Chunk (pixel, unit', w16) of the image holds source unit (unit' ^ swizzle(pixel));
  // padding chunks are zero-filled with ds_write.  Pieces [0, XPIECES) fill the x tile, the rest the dy tile.
  constexpr int XITEMS = PR * PC * (CPU << xp_sh), YITEMS = TH * 32 * (CPU << yp_sh);
  constexpr int XPIECES = (XITEMS + 63) / 64, NPIECE = XPIECES + (YITEMS + 63) / 64;
  constexpr int NSLOT = (NPIECE + kLoaderWaves - 1) / kLoaderWaves;   // pieces per loader wave and tile
  // Per piece and tile the loader needs: patch pixel (py, px) of the lane's chunk, its (swizzled) source unit and 16-byte slot.  All
  // of it is tile-invariant, so it is decoded ONCE per kernel into one packed register per slot (the full decode -- pixel / unit,
  // swizzle, division by the patch width -- is ~100 instructions per piece; 19 pieces per tile and loader wave, issued at raised
  // priority on the SIMDs of the MFMA waves).  Per tile a piece is then: unpack, two coordinate adds, bounds, one address.
  int spk[DMA ? NSLOT : 1];      // py | px << 8 | unit << 16 | w16 << 20 ; -1 = idle lane
  int soff[DMA ? NSLOT : 1];     // the lane's element offset from the tile's first patch / gradient pixel (valid for up == 0)
  auto slots_setup = [&]() __attribute__((always_inline)) {
    constexpr int XSH = CPU_SH + xp_sh, YSH = CPU_SH + yp_sh;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int piece = (wave - WgWaves<KS>::NW) + kLoaderWaves * k;
      spk[k] = -1; soff[k] = 0;
      if (piece < XPIECES) {
        const int item = piece * 64 + lane;
        if (item < XITEMS) {
          const int pix = item >> XSH, c16 = item & ((CPU << xp_sh) - 1);
          const int unit = (c16 >> CPU_SH) ^ swz(xp_sh, pix), w16 = c16 & (CPU - 1);
          const int py = pix / PC, px = pix - py * PC;
          spk[k] = py | (px << 8) | (unit << 16) | (w16 << 20);
          soff[k] = (py * a.Win + px) * a.x_ps + unit * x_gs2 + w16 * E16;
        }
      } else if (piece < NPIECE) {
        const int item = (piece - XPIECES) * 64 + lane;
        if (item < YITEMS) {
          const int pix = item >> YSH, c16 = item & ((CPU << yp_sh) - 1);
          const int unit = (c16 >> CPU_SH) ^ swz(yp_sh, pix), w16 = c16 & (CPU - 1);
          spk[k] = (pix >> 5) | ((pix & 31) << 8) | (unit << 16) | (w16 << 20);
          soff[k] = ((pix >> 5) * a.Wout + (pix & 31)) * a.dy_ps + unit * dy_gs2 + w16 * E16;
        }
      }
    }
  };
  auto dma_slots = [&](int n, int oy0, int ox0, int buf) __attribute__((always_inline)) {
    char* bx = smem + buf * kBufBytes;
    char* by = bx + PR * PC * xRowB;
    const int gy0 = oy0 * STRIDE - a.pad, gx0 = ox0 * STRIDE - a.pad;
    const T* xi = xg + (size_t)n * a.Hin * a.Win * a.xC;            // image bases (wave-uniform)
    const T* yi = dyg + (size_t)n * a.Hout * a.Wout * a.dyC;
    // tile bases for the precomputed offsets (may point before the image for border tiles: only in-range lanes dereference)
    const T* xt = xi + ((long long)gy0 * a.Win + gx0) * a.x_ps;
    const T* yt = yi + ((long long)oy0 * a.Wout + ox0) * a.dy_ps;
    const bool lin = a.up == 0;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int piece = (wave - WgWaves<KS>::NW) + kLoaderWaves * k;       // wave-uniform
      const int q = spk[k];
      if (piece >= NPIECE || q < 0) continue;
      const int py = q & 255, px = (q >> 8) & 255, unit = (q >> 16) & 15, w16 = q >> 20;
      if (piece < XPIECES) {
        const int gy = gy0 + py, gx = gx0 + px;
        if (gy >= 0 && gy < Hl && gx >= 0 && gx < Wl)
          glds16(lin ? xt + soff[k] : xi + (((gy >> a.up) * a.Win + (gx >> a.up)) * a.x_ps + unit * x_gs2 + w16 * E16), lds_addr(bx) + (unsigned)(piece * 1024));
        else *(u32x4*)(bx + (piece * 64 + lane) * 16) = u32x4{0u, 0u, 0u, 0u};
      } else {
        const int oy = oy0 + py, ox = ox0 + px;
        if (oy < a.Hout && ox < a.Wout)
          glds16(yt + soff[k], lds_addr(by) + (unsigned)((piece - XPIECES) * 1024));
        else *(u32x4*)(by + ((piece - XPIECES) * 64 + lane) * 16) = u32x4{0u, 0u, 0u, 0u};
      }
    }
  };

  // ---- lane-constant LDS address tables (keeps the MFMA loop almost free of address VALU work) ----
  // ds_read_b64_tr_b16 lane roles inside a 16-lane group: lane 4q+p supplies row q (pixel), columns
  // 4p..4p+3 (channels); lane i receives channel i of the 4 pixels.  This lane's first block row is pixel
  // L = 8*(lane>>5) + q of the 16-pixel k-step, its channels start at byte chb inside the 32-channel unit.
  // The 64-byte-unit swizzle ((pixel>>1)&1) only depends on (row start mod 4, dx, L): row starts are
  // multiples of PC (= 2 mod 4 for the 3x3/4x4 patches), so two tables (row parity P) cover every case and
  // the per-step address is table + wave-uniform row offset (+ compile-time offsets for the second half).
  const int g16 = lane >> 4, lq = (lane >> 2) & 3, lp = lane & 3;
  const int Lpix = 8 * (g16 >> 1) + lq;
  const int chb = (16 * (g16 & 1) + 4 * lp) * 2;
  int tabX[2][KS];
  int tabY = 0;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int P = 0; P < 2; ++P)
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) {
        const int x = 2 * P + dx + Lpix * STRIDE;   // pixel index mod 4 is what the swizzle needs
        tabX[P][dx] = (Lpix * STRIDE + dx) * xRowB + ((W.ci_rel ^ swz(xp_sh, x)) * UB) + chb;
      }
    tabY = Lpix * dyRowB + ((W.co_rel ^ swz(yp_sh, Lpix)) * UB) + chb;
  }
  const int ky = W.tap0 / KS;   // this wave's kernel row

  int tile = split;
  int cur = 0;
  if constexpr (DMA) {
    if (wave >= WgWaves<KS>::NW) {
      // ---- loader wavefronts: tile t+1 -> buffer cur^1 while the compute waves run tile t out of buffer cur ----
      int n, oy0, ox0;
      slots_setup();
      auto fill = [&](int t, int buf) __attribute__((always_inline)) {
        tile_origin(t, n, oy0, ox0);
        dma_slots(n, oy0, ox0, buf);
      };
      __builtin_amdgcn_s_setprio(3);   // the copy must not wait behind the compute waves' issue slots (334 -> 317 us)
      if (tile < a.ntiles) fill(tile, 0);
      for (; tile < a.ntiles; tile += a.S) {
        // own DMAs landed (vmcnt) and zero fills written (lgkmcnt in the barrier's fence); past the barrier the buffer
        // is published and nobody reads buffer cur^1 any more (its MFMA phase precedes this barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile + a.S < a.ntiles) fill(tile + a.S, cur ^ 1);
        cur ^= 1;
      }
      return;
    }
  } else if (tile < a.ntiles) prefetch(tile);
  for (; tile < a.ntiles; tile += a.S) {
    if constexpr (DMA) {
      __syncthreads();
      ldsX = smem + cur * kBufBytes;
      ldsY = ldsX + PR * PC * xRowB;
      cur ^= 1;
    } else {
      __syncthreads();  // previous tile's LDS reads done
      commit(tile);
      __syncthreads();
      if (tile + a.S < a.ntiles) prefetch(tile + a.S);
    }
    if (W.active) {
      if constexpr (sizeof(T) == 2 && KS == 3 && STRIDE == 1 && VAR == 3) {
        // v_mfma_f32_16x16x32: K = the 32 pixels of one tile row.  A (16 input channels x 32 pixels) and B (32 pixels x 16 output
        // channels): lane group g = lane>>4 supplies / receives pixels 8g..8g+7, so each 16-lane group of a transposed read takes its
        // own 4-pixel x 16-channel block.  Per row: 2 channel halves x 3 reads of x (pixels p0..p0+11 -> the three kernel columns by
        // register shifts, as in variant 1), 2 x 2 reads of dy, 12 MFMAs of 16 cycles (= the 6 of 32 cycles of the other variants).
        using Fr = typename FragAB<T>::type;
        const int Lp = 8 * g16 + lq;                                   // this lane's address row inside the 32-pixel run
        const int tx0 = Lp * xRowB + ((W.ci_rel ^ swz(xp_sh, Lp)) * UB) + lp * 8;        // patch-row parity 0 (row start = 0 mod 4)
        const int tx1 = Lp * xRowB + ((W.ci_rel ^ swz(xp_sh, 2 + Lp)) * UB) + lp * 8;    // parity 1 (row start = 2 mod 4)
        const int ty0 = Lp * dyRowB + ((W.co_rel ^ swz(yp_sh, Lp)) * UB) + lp * 8;
        const unsigned one2 = sizeof(T) == 2 && Elem<T>::kDtype == SRGANFD_F16 ? 0x3C003C00u : 0x3F803F80u;      // two 1.0 in f16 / bf16
        const Fr ones = __builtin_bit_cast(Fr, u32x4{one2, one2, one2, one2});
        for (int rr = 0; rr < rows_per; ++rr) {
          const int ro = W.ks_idx * rows_per + rr, prow = ro + ky;
          const char* xb = ldsX + ((prow & 1) ? tx1 : tx0) + prow * PC * xRowB;
          const char* yb = ldsY + ty0 + ro * 32 * dyRowB;
          Fr bq[2];
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) {
            const u32x2 blo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(yb + nh * 32)));
            const u32x2 bhi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(yb + nh * 32 + 4 * dyRowB)));
            const u32x4 b4 = {blo.x, blo.y, bhi.x, bhi.y};
            bq[nh] = __builtin_bit_cast(Fr, b4);
            // bias gradient = sum over the row's 32 pixels of dy: one MFMA against a fragment of ones (every row of the result is
            // that sum) instead of 8 conversions + 7 adds per lane -- 16 matrix-pipe cycles for ~90 VALU cycles in the one wave of
            // three that carries it
            if (W.bias_slab >= 0) bacc[nh] = mfma16<T>(ones, bq[nh], bacc[nh]);
          }
#pragma unroll
          for (int ch = 0; ch < 2; ++ch) {
            const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xb + ch * 32)));
            const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xb + ch * 32 + 4 * xRowB)));
            const u32x2 nx = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xb + ch * 32 + 8 * xRowB)));
            const u32x4 f0 = {lo.x, lo.y, hi.x, hi.y};
            const u32x4 f1 = {__builtin_amdgcn_alignbit(lo.y, lo.x, 16), __builtin_amdgcn_alignbit(hi.x, lo.y, 16),
                              __builtin_amdgcn_alignbit(hi.y, hi.x, 16), __builtin_amdgcn_alignbit(nx.x, hi.y, 16)};
            const u32x4 f2 = {lo.y, hi.x, hi.y, nx.x};
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
              acc16[0][ch][nh] = mfma16<T>(__builtin_bit_cast(Fr, f0), bq[nh], acc16[0][ch][nh]);
              acc16[1][ch][nh] = mfma16<T>(__builtin_bit_cast(Fr, f1), bq[nh], acc16[1][ch][nh]);
              acc16[2][ch][nh] = mfma16<T>(__builtin_bit_cast(Fr, f2), bq[nh], acc16[2][ch][nh]);
            }
          }
        }
      } else
      for (int rr = 0; rr < rows_per; ++rr) {
        const int ro = W.ks_idx * rows_per + rr;
        if constexpr (sizeof(T) == 2) {
          const int prow = ro * STRIDE + ky;                       // patch row
          const int P = ((PC & 3) == 2) ? (prow & 1) : 0;          // (prow*PC) mod 4 == 2*P
          const int xrow_off = prow * PC * xRowB;                  // wave-uniform
          const int yrow_off = ro * 32 * dyRowB;
          int ax[KS];
#pragma unroll
          for (int dx = 0; dx < KS; ++dx) ax[dx] = (P ? tabX[1][dx] : tabX[0][dx]) + xrow_off;
          const int ay = tabY + yrow_off;
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const char* yb = ldsY + ay + hh * 16 * dyRowB;
            const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(yb));
            const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(yb + 4 * dyRowB));
            const auto bfrag = cat_frag<T>(blo, bhi);
            if (W.bias_slab >= 0) {
              const u32x2 l = __builtin_bit_cast(u32x2, blo), h2 = __builtin_bit_cast(u32x2, bhi);
              float f8[8];
              unpack8<T>(u32x4{l.x, l.y, h2.x, h2.y}, f8);
              bsum += ((f8[0] + f8[1]) + (f8[2] + f8[3])) + ((f8[4] + f8[5]) + (f8[6] + f8[7]));
            }
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
              const char* xb = ldsX + ax[dx] + hh * 16 * STRIDE * xRowB;
              const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xb));
              const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xb + 4 * STRIDE * xRowB));
              acc[dx] = mfma32<T>(cat_frag<T>(lo, hi), bfrag, acc[dx]);
            }
          }
        } else {
#pragma unroll 1
          for (int kk = 0; kk < 16; ++kk) {
            const int oc = 2 * kk + h;
            const float bv = *(const float*)(ldsY + (ro * 32 + oc) * dyRowB + (W.co_rel * 32 + r) * 4);
            if (W.bias_slab >= 0) bsum += bv;
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
              const int p = (ro * STRIDE + ky) * PC + oc * STRIDE + dx;
              const float av = *(const float*)(ldsX + p * xRowB + (W.ci_rel * 32 + r) * 4);
              acc[dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[dx], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  if (W.active) {
    float* slab = a.slabs + (size_t)(W.slab_base + split * W.ks_n + W.ks_idx) * (KS * KS * 1024) + W.tap0 * 1024;
    if constexpr (VAR == 3) {
      // D of 16x16x32: column = lane & 15 (output channel inside its half), row = 4 * (lane >> 4) + register (input channel)
#pragma unroll
      for (int tl = 0; tl < 3; ++tl)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int i = 0; i < 4; ++i) slab[(tl * 32 + 16 * ch + 4 * (lane >> 4) + i) * 32 + 16 * nh + (lane & 15)] = acc16[tl][ch][nh][i];
      if (W.bias_slab >= 0) {
        // every row of bacc holds the 32-pixel sums of channels 16 * nh + (lane & 15): take row 0 (lane group 0), slots 0..31 = channels
        float t0 = (lane >> 4) == 0 ? bacc[0][0] : 0.f, t1 = (lane >> 4) == 0 ? bacc[1][0] : 0.f;
        (void)bsum16;
        t0 += __shfl_xor(t0, 16, 64); t0 += __shfl_xor(t0, 32, 64);
        t1 += __shfl_xor(t1, 16, 64); t1 += __shfl_xor(t1, 32, 64);
        a.bslabs[(size_t)(W.bias_slab + split * W.ks_n + W.ks_idx) * 64 + lane] = lane < 16 ? t0 : (lane < 32 ? t1 : 0.f);
      }
    } else {
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
      for (int i = 0; i < 16; ++i) slab[(tl * 32 + mfma32_row(i, lane)) * 32 + r] = acc[tl][i];
    if (W.bias_slab >= 0) a.bslabs[(size_t)(W.bias_slab + split * W.ks_n + W.ks_idx) * 64 + lane] = bsum;
    }
  }
}

// Deterministic slab reduction + layout change to the NCHW fp32 parameter gradient.
// One block = one (task, tap); one thread = one element of its 32x32 fp32 tile (1024 threads: a launch has only
// taps x tasks ~ 230 blocks, so the memory parallelism has to come from threads), eight slabs in flight per thread
// (independent accumulators, fixed summation order -> bitwise reproducible).
struct WgRedJob { const WgTask* tasks; const float* slabs; const float* bslabs; float* grads; const float* scalars; int ntasks; int pad_; };
static constexpr int kRedBatch = 8;
struct WgRedJobs { WgRedJob j[kRedBatch]; };
// blockIdx.z selects the job: several weight-gradient launches (dense blocks) reduced by ONE launch -- the kernel is bound by load round
// trips at ~17-23 us whatever it reduces, so four blocks' slabs cost little more than one's
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const WgRedJobs jobs, int ntap_wave) {
  const WgRedJob& J = jobs.j[blockIdx.z];
  if ((int)blockIdx.y >= J.ntasks) return;
  const WgTask* __restrict__ tasks = J.tasks;
  const float* __restrict__ slabs = J.slabs;
  const float* __restrict__ bslabs = J.bslabs;
  float* __restrict__ grads = J.grads;
  const float* __restrict__ scalars = J.scalars;
  const WgTask T = tasks[blockIdx.y];
  const int KT = T.ksize * T.ksize;
  const int tl = blockIdx.x;              // tap
  float alpha = T.alpha;
  if (T.alpha_off >= 0) alpha *= scalars[T.alpha_off];
  const size_t slab_stride = (size_t)ntap_wave * 1024;
  const int row = threadIdx.x >> 5, col = threadIdx.x & 31;
  const float* p = slabs + (size_t)T.slab_base * slab_stride + tl * 1024 + threadIdx.x;
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.f;
  int k = 0;
  // 16 loads in flight per thread (a dense block's 36 pixel splits: two round trips + a short tail instead of four + four dependent ones)
  for (; k + 16 <= T.nslabs; k += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(size_t)(k + u) * slab_stride];
#pragma unroll
    for (int u = 0; u < 16; ++u) s[u & 7] += v[u];
  }
  if (k < T.nslabs) {
    float v[15];
#pragma unroll
    for (int u = 0; u < 15; ++u) v[u] = k + u < T.nslabs ? p[(size_t)(k + u) * slab_stride] : 0.f;
#pragma unroll
    for (int u = 0; u < 15; ++u) s[u & 7] += v[u];
  }
  const float tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  const int ci = T.ci_base + row, co = T.co_base + col;
  if (ci < T.ci_dst && co < T.co_dst) {
    float* d = grads + T.dw_off + ((size_t)co * T.ci_dst + ci) * KT + T.tap0 + tl;
    *d = alpha * tot + (T.beta != 0.f ? T.beta * *d : 0.f);
  }
  if (T.bias_slab >= 0 && T.db_off >= 0 && tl == 0 && threadIdx.x < 32) {
    const int cob = T.co_base + threadIdx.x;
    if (cob < T.co_dst) {
      float sb = 0.f;
      const float* pb = bslabs + (size_t)T.bias_slab * 64;
      for (int kk = 0; kk < T.nslabs; ++kk) sb += pb[kk * 64 + threadIdx.x] + pb[kk * 64 + 32 + threadIdx.x];
      float* d = grads + T.db_off + cob;
      *d = alpha * sb + (T.beta != 0.f ? T.beta * *d : 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side: plan construction
// ------------------------------------------------------------------------------------------------
static int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

struct PlanBuild {
  std::vector<WgGroup> groups;
  std::vector<WgTask> tasks;
  WgHeader hdr;
};

static int build_plan(const srganfd_wgrad_shape* s, const srganfd_wgrad_conv* convs, PlanBuild& pb) {
  if (!s || !convs || s->nconv <= 0) return set_err(SRGANFD_EINVAL, "wgrad: null shape/convs");
  if (!((s->ksize == 3 && s->stride == 1) || (s->ksize == 4 && s->stride == 2) || (s->ksize == 1 && s->stride == 1) ||
        (s->ksize == 3 && s->stride == 2) || (s->ksize == 2 && s->stride == 2)))
    return set_err(SRGANFD_EINVAL, "wgrad: unsupported ksize=%d stride=%d", s->ksize, s->stride);
  const int hl = s->h_in << (s->up ? 1 : 0), wl = s->w_in << (s->up ? 1 : 0);
  if ((hl + 2 * s->pad - s->ksize) / s->stride + 1 != s->h_out || (wl + 2 * s->pad - s->ksize) / s->stride + 1 != s->w_out)
    return set_err(SRGANFD_EINVAL, "wgrad: output dims inconsistent");
  const int KT = s->ksize * s->ksize;
  const int NT = KT;                                   // taps per slab (all of them; one wave covers one kernel row)
  const int NW = s->ksize == 3 ? 12 : (s->ksize == 4 ? 8 : 4);
  const int cap = NW / s->ksize;                       // (ci,co) blocks per workgroup
  WgHeader& H = pb.hdr;
  memset(&H, 0, sizeof(H));
  H.magic = kWgMagic; H.dtype = s->dtype; H.N = s->n; H.Hin = s->h_in; H.Win = s->w_in; H.up = s->up ? 1 : 0;
  H.ks = s->ksize; H.stride = s->stride; H.pad = s->pad; H.Hout = s->h_out; H.Wout = s->w_out; H.ntap_wave = NT;
  H.tiles_x = ceil_div(s->w_out, 32);

  // Wave tasks = (32-channel x block, 32-channel dy block, tap half).  Tasks of ALL convs of the launch
  // are bucketed by the 64-channel x range and 64-channel dy range they read (32-channel x range for the
  // 4x4 stride-2 case, whose patch is 4x larger): one workgroup stages those ranges once and its four
  // waves run the bucket's tasks (split over tile rows when a bucket holds fewer than four).
  const int kTH = (s->stride == 1) ? 8 : 4;
  H.tiles_y = ceil_div(s->h_out, kTH);
  H.ntiles = s->n * H.tiles_x * H.tiles_y;
  struct Bucket { int xb, yb; std::vector<int> tasks; std::vector<int> ci_abs, co_abs; };
  std::vector<Bucket> buckets;
  const int xdiv = (s->stride == 2) ? 1 : 2;   // stride-2 patches are 4x larger: one 32-channel x unit per workgroup
  for (int c = 0; c < s->nconv; ++c) {
    const srganfd_wgrad_conv& cv = convs[c];
    if (cv.cin <= 0 || cv.cin % 32 || cv.cout <= 0 || cv.cout % 32 || cv.ci_lo % 32 || cv.co_lo % 32 || cv.ci_lo < 0 || cv.co_lo < 0 ||
        cv.ci_lo + cv.cin > s->x_channels || cv.co_lo + cv.cout > s->dy_channels)
      return set_err(SRGANFD_EINVAL, "wgrad: conv %d channel ranges invalid", c);
    const int cib = cv.cin / 32, cob = cv.cout / 32;
    for (int ob = 0; ob < cob; ++ob)
      for (int cb = 0; cb < cib; ++cb)
        {
          WgTask t; memset(&t, 0, sizeof(t));
          t.dw_off = cv.dw_off; t.db_off = cv.db_off; t.alpha_off = cv.alpha_off;
          t.co_dst = cv.co_dst; t.ci_dst = cv.ci_dst; t.co_base = ob * 32; t.ci_base = cb * 32;
          t.tap0 = 0; t.ntap = NT; t.ksize = s->ksize; t.alpha = cv.alpha; t.beta = cv.beta;
          t.bias_slab = (cb == 0 && cv.db_off >= 0) ? 0 : -1;  // resolved below
          const int id = (int)pb.tasks.size();
          pb.tasks.push_back(t);
          const int ca = cv.ci_lo / 32 + cb, oa = cv.co_lo / 32 + ob;
          Bucket* bk = nullptr;
          for (auto& q : buckets)
            if (q.xb == ca / xdiv && q.yb == oa / 2 && (int)q.tasks.size() < cap) { bk = &q; break; }
          if (!bk) { buckets.push_back(Bucket{ca / xdiv, oa / 2, {}, {}, {}}); bk = &buckets.back(); }
          bk->tasks.push_back(id); bk->ci_abs.push_back(ca); bk->co_abs.push_back(oa);
        }
  }
  // Two buckets that hold ONE task each (a dense block's launch leaves two: conv4's fifth input block and conv2's third) become one
  // workgroup staging the two tasks' own units -- x units {a, b}, dy units {c, d}, not a 64-channel range -- instead of two workgroups
  // whose twelve waves share one task: 7 channel groups instead of 8 for the dense block, so 36 pixel splits fit the chip instead of 32
  // (3x3 stride 1, where a workgroup stages two units a side).
  struct Merged { int a, b; };
  std::vector<Merged> merged;
  std::vector<char> gone(buckets.size(), 0);
  if (xdiv == 2 && s->ksize == 3) {
    int prev = -1;
    for (int i = 0; i < (int)buckets.size(); ++i) {
      if (buckets[i].tasks.size() != 1) continue;
      if (prev < 0) { prev = i; continue; }
      const int ca0 = buckets[prev].ci_abs[0], ca1 = buckets[i].ci_abs[0], oa0 = buckets[prev].co_abs[0], oa1 = buckets[i].co_abs[0];
      if (ca0 != ca1 && oa0 != oa1) { merged.push_back(Merged{prev, i}); gone[prev] = gone[i] = 1; prev = -1; }
    }
  }
  for (const Merged& mg : merged) {
    const Bucket& p = buckets[mg.a]; const Bucket& q = buckets[mg.b];
    WgGroup g; memset(&g, 0, sizeof(g));
    const int ca[2] = {p.ci_abs[0], q.ci_abs[0]}, oa[2] = {p.co_abs[0], q.co_abs[0]}, tk[2] = {p.tasks[0], q.tasks[0]};
    const int cmin = ca[0] < ca[1] ? ca[0] : ca[1], omin = oa[0] < oa[1] ? oa[0] : oa[1];
    g.x_c0 = cmin * 32; g.dy_c0 = omin * 32; g.x_units = 2; g.dy_units = 2;
    g.x_u1 = (ca[0] > ca[1] ? ca[0] - ca[1] : ca[1] - ca[0]); g.dy_u1 = (oa[0] > oa[1] ? oa[0] - oa[1] : oa[1] - oa[0]);
    int k = 0;
    for (int t = 0; t < 2; ++t)
      for (int ky = 0; ky < s->ksize; ++ky)
        for (int qq = 0; qq < cap / 2; ++qq) {
          WgWave& w = g.w[k++];
          w.active = 1; w.ci_rel = ca[t] == cmin ? 0 : 1; w.co_rel = oa[t] == omin ? 0 : 1;
          w.ks_idx = qq; w.ks_n = cap / 2; w.tap0 = ky * s->ksize; w.slab_base = tk[t];
          w.bias_slab = ky == 0 ? 0 : -1;
        }
    for (; k < kMaxWaves; ++k) { g.w[k].active = 0; g.w[k].ks_n = 1; g.w[k].bias_slab = -1; }
    pb.groups.push_back(g);
  }
  for (size_t bi = 0; bi < buckets.size(); ++bi) {
    if (gone[bi]) continue;
    auto& bk = buckets[bi];
    WgGroup g; memset(&g, 0, sizeof(g));
    g.x_c0 = bk.xb * xdiv * 32; g.dy_c0 = bk.yb * 64;
    g.x_u1 = g.dy_u1 = 1;
    g.x_units = (xdiv == 2 && s->x_channels - g.x_c0 >= 64) ? 2 : 1;
    g.dy_units = (s->dy_channels - g.dy_c0 >= 64) ? 2 : 1;
    const int nt = (int)bk.tasks.size();
    const int ks_n = (cap % nt == 0) ? cap / nt : 1;
    int k = 0;
    for (int t = 0; t < nt; ++t)
      for (int ky = 0; ky < s->ksize; ++ky)
        for (int q = 0; q < ks_n; ++q) {
          WgWave& w = g.w[k++];
          w.active = 1; w.ci_rel = bk.ci_abs[t] - g.x_c0 / 32; w.co_rel = bk.co_abs[t] - g.dy_c0 / 32;
          w.ks_idx = q; w.ks_n = ks_n; w.tap0 = ky * s->ksize; w.slab_base = bk.tasks[t];  // task id for now
          w.bias_slab = ky == 0 ? 0 : -1;                                                    // resolved below
        }
    for (; k < kMaxWaves; ++k) { g.w[k].active = 0; g.w[k].ks_n = 1; g.w[k].bias_slab = -1; }
    pb.groups.push_back(g);
  }
  H.ngroups = (int)pb.groups.size();
  H.ntasks = (int)pb.tasks.size();
  // pixel-tile splits: aim at ~2 workgroups per CU over the whole launch
  int S = s->splits;
  if (S <= 0) { S = conv_device_cus() / H.ngroups; if (S < 1) S = 1; }   // one workgroup (12 waves) per CU, ONE round: never more workgroups than the device has CUs (256 in dry runs)
  if (S > H.ntiles) S = H.ntiles;
  if (S > 4096) S = 4096;
  H.S = S;
  // slab allocation: per task S * ks_n slabs
  std::vector<int> ksn(pb.tasks.size(), 1);
  for (auto& g : pb.groups)
    for (int k = 0; k < kMaxWaves; ++k)
      if (g.w[k].active) ksn[g.w[k].slab_base] = g.w[k].ks_n;
  long long slab = 0, bslab = 0;
  for (size_t t = 0; t < pb.tasks.size(); ++t) {
    pb.tasks[t].slab_base = (int)slab; pb.tasks[t].nslabs = S * ksn[t];
    slab += pb.tasks[t].nslabs;
    if (pb.tasks[t].bias_slab == 0) { pb.tasks[t].bias_slab = (int)bslab; bslab += pb.tasks[t].nslabs; }
  }
  if (slab > 0x7fffffffLL) return set_err(SRGANFD_EINVAL, "wgrad: too many slabs");
  for (auto& g : pb.groups)
    for (int k = 0; k < kMaxWaves; ++k)
      if (g.w[k].active) {
        const WgTask& t = pb.tasks[g.w[k].slab_base];
        g.w[k].bias_slab = (g.w[k].bias_slab == 0) ? t.bias_slab : -1;   // only the ky == 0 wave sums the bias
        g.w[k].slab_base = t.slab_base;
      }
  H.nslabs_total = slab; H.nbias_slabs = bslab;
  H.bias_slab_off = slab * NT * 1024;
  H.x_upad = 1; H.dy_upad = 1;
  for (auto& g : pb.groups) { if (g.x_units > H.x_upad) H.x_upad = g.x_units; if (g.dy_units > H.dy_upad) H.dy_upad = g.dy_units; }
  const int UB = 32 * (s->dtype == SRGANFD_F32 ? 4 : 2);
  const int PR = (kTH - 1) * s->stride + s->ksize, PC = 31 * s->stride + s->ksize;
  H.lds_bytes = PR * PC * H.x_upad * UB + kTH * 32 * H.dy_upad * UB;
  if (H.lds_bytes > 160 * 1024) return set_err(SRGANFD_EINVAL, "wgrad: LDS tile %d B too large", H.lds_bytes);
  H.dma_ok = (s->dtype != SRGANFD_F32 && 2 * H.lds_bytes <= 160 * 1024) ? 1 : 0;
  // 3x3 stride 2 (A-ESRGAN's encoder): with the four loader waves the workgroup has 1024 threads = 128 VGPRs per lane, and this shape's
  // addressing does not fit them (10-13 spilled registers, scratch inside the loop); 12 waves with register staging have 168
  if (s->ksize == 3 && s->stride == 2) H.dma_ok = 0;
  for (auto& g : pb.groups) if (g.x_units != H.x_upad || g.dy_units != H.dy_upad) H.dma_ok = 0;
  H.groups_off = (sizeof(WgHeader) + 15) & ~15LL;
  H.tasks_off = (H.groups_off + (long long)sizeof(WgGroup) * H.ngroups + 15) & ~15LL;
  H.total_bytes = (H.tasks_off + (long long)sizeof(WgTask) * H.ntasks + 15) & ~15LL;
  return SRGANFD_OK;
}

size_t wgrad_plan_bytes_impl(const srganfd_wgrad_shape* s, const srganfd_wgrad_conv* convs) {
  PlanBuild pb;
  if (build_plan(s, convs, pb) != SRGANFD_OK) return 0;
  return (size_t)pb.hdr.total_bytes;
}

int wgrad_plan_build_impl(const srganfd_wgrad_shape* s, const srganfd_wgrad_conv* convs, void* plan_host, size_t plan_bytes,
                          size_t* workspace_bytes) {
  PlanBuild pb;
  int rc = build_plan(s, convs, pb);
  if (rc != SRGANFD_OK) return rc;
  if (!plan_host || plan_bytes < (size_t)pb.hdr.total_bytes) return set_err(SRGANFD_ENOSPC, "wgrad: plan buffer too small");
  memset(plan_host, 0, pb.hdr.total_bytes);
  memcpy(plan_host, &pb.hdr, sizeof(WgHeader));
  memcpy((char*)plan_host + pb.hdr.groups_off, pb.groups.data(), sizeof(WgGroup) * pb.groups.size());
  memcpy((char*)plan_host + pb.hdr.tasks_off, pb.tasks.data(), sizeof(WgTask) * pb.tasks.size());
  if (workspace_bytes) *workspace_bytes = (size_t)(pb.hdr.bias_slab_off + pb.hdr.nbias_slabs * 64) * sizeof(float);
  return SRGANFD_OK;
}


template <typename T, int KS, int STRIDE, int XP, int YP, bool DMA, int VAR>
static int launch_wgrad4(const WgHeader& H, const WgK& k, hipStream_t stream) {
  auto kern = wgrad_kernel<T, KS, STRIDE, XP, YP, DMA, VAR>;
  static int attr_lds[64] = {0};   // per device: the attribute belongs to the device's code object
  const int lds = DMA ? 2 * H.lds_bytes : H.lds_bytes;
  if (!g_dry_run) {
    int dev = 0;
    SRGANFD_HIP_CHECK(hipGetDevice(&dev));
    if (lds > attr_lds[dev & 63]) {
      SRGANFD_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr_lds[dev & 63] = lds;
    }
  }
  SRGANFD_LAUNCH(kern, dim3(H.S * H.ngroups), dim3(64 * (WgWaves<KS>::NW + (DMA ? kLoaderWaves : 0))), lds, stream, k);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
template <typename T, int KS, int STRIDE, int XP, int YP, bool DMA>
static int launch_wgrad3(const WgHeader& H, const WgK& k, hipStream_t stream) {
  if constexpr (sizeof(T) == 2 && KS == 3 && STRIDE == 1 && DMA) {
    return launch_wgrad4<T, KS, STRIDE, XP, YP, DMA, 3>(H, k, stream);
  } else {
    return launch_wgrad4<T, KS, STRIDE, XP, YP, DMA, 0>(H, k, stream);
  }
}
template <typename T, int KS, int STRIDE, int XP, int YP>
static int launch_wgrad2(const WgHeader& H, const WgK& k, hipStream_t stream) {
  if constexpr (sizeof(T) == 2) {
    if constexpr (!(KS == 3 && STRIDE == 2))
      if (H.dma_ok) return launch_wgrad3<T, KS, STRIDE, XP, YP, true>(H, k, stream);
  }
  return launch_wgrad3<T, KS, STRIDE, XP, YP, false>(H, k, stream);
}
template <typename T, int KS, int STRIDE>
static int launch_wgrad(const WgHeader& H, const WgK& k, hipStream_t stream) {
  if (H.x_upad == 2) return H.dy_upad == 2 ? launch_wgrad2<T, KS, STRIDE, 2, 2>(H, k, stream) : launch_wgrad2<T, KS, STRIDE, 2, 1>(H, k, stream);
  return H.dy_upad == 2 ? launch_wgrad2<T, KS, STRIDE, 1, 2>(H, k, stream) : launch_wgrad2<T, KS, STRIDE, 1, 1>(H, k, stream);
}

int wgrad_reduce_batch_impl(const srganfd_wgrad_reduce_job* jobs, int njobs, hipStream_t stream);
int wgrad_impl(const void* plan_host, const void* plan_dev, srganfd_view x, srganfd_view dy, float* grads, const float* scalars,
               void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!plan_host || !plan_dev || !x.ptr || !dy.ptr || !workspace) return set_err(SRGANFD_EINVAL, "wgrad: null pointer");
  const WgHeader& H = *(const WgHeader*)plan_host;
  if (H.magic != kWgMagic) return set_err(SRGANFD_EINVAL, "wgrad: bad plan");
  if ((size_t)H.Hin * H.Win * (size_t)x.cstride >= 0x7fffffffULL || (size_t)H.Hout * H.Wout * (size_t)dy.cstride >= 0x7fffffffULL)
    return set_err(SRGANFD_EINVAL, "wgrad: one image is too large for 32-bit element offsets");
  const size_t need = (size_t)(H.bias_slab_off + H.nbias_slabs * 64) * sizeof(float);
  if (workspace_bytes < need) return set_err(SRGANFD_ENOSPC, "wgrad: workspace %zu < %zu", workspace_bytes, need);
  WgK k;
  k.x = x.ptr; k.dy = dy.ptr; k.slabs = (float*)workspace; k.bslabs = (float*)workspace + H.bias_slab_off;
  k.groups = (const WgGroup*)((const char*)plan_dev + H.groups_off);
  k.xC = x.cstride; k.x_c0v = x.c0; k.dyC = dy.cstride; k.dy_c0v = dy.c0;
  if ((x.planar && (x.c0 % 32 || x.cstride % 32)) || (dy.planar && (dy.c0 % 32 || dy.cstride % 32)))
    return set_err(SRGANFD_EINVAL, "wgrad: a planar view needs c0 and cstride multiples of 32");
  k.x_ps = x.planar ? 32 : x.cstride; k.x_gs = x.planar ? H.Hin * H.Win * 32 : 32;
  k.dy_ps = dy.planar ? 32 : dy.cstride; k.dy_gs = dy.planar ? H.Hout * H.Wout * 32 : 32;
  k.N = H.N; k.Hin = H.Hin; k.Win = H.Win; k.up = H.up; k.pad = H.pad; k.Hout = H.Hout; k.Wout = H.Wout; k.S = H.S;
  k.ngroups = H.ngroups; k.x_upad = H.x_upad; k.dy_upad = H.dy_upad; k.ntiles = H.ntiles; k.tiles_x = H.tiles_x; k.tiles_y = H.tiles_y;
  int rc;
#define WG_BY_TYPE(KS_, S_) (H.dtype == SRGANFD_BF16 ? launch_wgrad<bf16_t, KS_, S_>(H, k, stream) : H.dtype == SRGANFD_F16 ? launch_wgrad<f16_t, KS_, S_>(H, k, stream) : launch_wgrad<float, KS_, S_>(H, k, stream))
  if (H.ks == 3 && H.stride == 1) rc = WG_BY_TYPE(3, 1);
  else if (H.ks == 3) rc = WG_BY_TYPE(3, 2);
  else if (H.ks == 4) rc = WG_BY_TYPE(4, 2);
  else if (H.ks == 2) rc = WG_BY_TYPE(2, 2);
  else rc = WG_BY_TYPE(1, 1);
#undef WG_BY_TYPE
  if (rc != SRGANFD_OK) return rc;
  if (!grads) return SRGANFD_OK;          // srganfd_conv2d_wgrad_partial: the slabs stay in the workspace for srganfd_wgrad_reduce_batch
  srganfd_wgrad_reduce_job job = {plan_host, plan_dev, grads, scalars, workspace};
  return wgrad_reduce_batch_impl(&job, 1, stream);
}

int wgrad_reduce_batch_impl(const srganfd_wgrad_reduce_job* jobs, int njobs, hipStream_t stream) {
  if (!jobs || njobs <= 0 || njobs > kRedBatch) return set_err(SRGANFD_EINVAL, "wgrad_reduce_batch: 1..%d jobs", kRedBatch);
  WgRedJobs J;
  memset(&J, 0, sizeof(J));
  int ntap = 0, maxtasks = 0;
  for (int i = 0; i < njobs; ++i) {
    const srganfd_wgrad_reduce_job& q = jobs[i];
    if (!q.plan_host || !q.plan_dev || !q.grads || !q.workspace) return set_err(SRGANFD_EINVAL, "wgrad_reduce_batch: null pointer in job %d", i);
    const WgHeader& H = *(const WgHeader*)q.plan_host;
    if (H.magic != kWgMagic) return set_err(SRGANFD_EINVAL, "wgrad_reduce_batch: bad plan in job %d", i);
    if (i && H.ntap_wave != ntap) return set_err(SRGANFD_EINVAL, "wgrad_reduce_batch: jobs of one batch share the kernel size");
    ntap = H.ntap_wave;
    if (H.ntasks > maxtasks) maxtasks = H.ntasks;
    J.j[i].tasks = (const WgTask*)((const char*)q.plan_dev + H.tasks_off);
    J.j[i].slabs = (const float*)q.workspace;
    J.j[i].bslabs = (const float*)q.workspace + H.bias_slab_off;
    J.j[i].grads = q.grads; J.j[i].scalars = q.scalars; J.j[i].ntasks = H.ntasks;
  }
  SRGANFD_LAUNCH(wgrad_reduce_kernel, dim3(ntap, maxtasks, njobs), dim3(1024), 0, stream, J, ntap);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

}  // namespace srganfd
