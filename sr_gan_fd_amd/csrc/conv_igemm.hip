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
//   * one workgroup = WR x WN wavefronts (8 for the 3x3 convs = 2 per SIMD; two workgroups per CU give
//     4 waves per SIMD, which is what hides the LDS/HBM latency chain of a single wave).  Wave (wr, wn)
//     owns MR image rows x 32 output channels; each 32-pixel row is the M side of a 32x32 MFMA tile.
//     Tile = (WR*MR) rows x 32 pixels x (32*WN) channels.
//   * K = taps x input channels is walked in chunks of 32 channels: the haloed input patch and the
//     KSxKS x 32 x (32*WN) weight slab are staged in LDS once per chunk and every tap re-reads the SAME
//     patch at a shifted address (implicit im2col, no materialised columns).
//   * A fragments: ds_read_b128 (bf16) of 8 consecutive channels of one pixel; the 16-byte chunk index is
//     XOR-swizzled with (pixel>>2)&3 so the 16-lane read groups of ds_read_b128 are bank-conflict
//     free.  For a fixed kernel column the MR rows x KS kernel rows touch only (MR-1)*S+KS patch rows,
//     so each A fragment is read once and reused by every (row, kernel row) pair.
//     B fragments are pre-packed in lane order (pack.hip) -> linear ds_read_b128.
//   * global -> register -> LDS staging, next chunk's loads issued before the MFMA phase
//     (issue-early / write-late); all staging addresses are base + immediate (no per-chunk VALU).
//   * epilogue: accumulators -> fp32 LDS tile -> 16-byte vector loads of residual / mask tensors and
//     16-byte stores (one pixel's channels are contiguous in NHWC).
//   * bf16: v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  f32: v_mfma_f32_32x32x2_f32, an exact
//     fp32 fma chain, used as the parity mode against the CPU oracle.
#include "conv_common.hpp"
#include <stdlib.h>
#include <string.h>
#include <utility>
// epilogue kinds with any of these operand bits keep one tile per workgroup (0: every fixed kind runs persistent, 7: only kind 0)
#ifndef SRGANFD_CROSS_MASK
#define SRGANFD_CROSS_MASK 7
#endif
// fragment read-ahead of the 3x3 stride-1 16x16x32 loop (64-channel tiles; 32-channel tiles one less).  4 since the compile-time epilogue
// kinds freed the registers: no spills, G-only step 59.07 vs 59.34 ms (3), 5: equal with 3 spilled registers in one kind, 6: spills, +3 %; one or two more for the kinds with epilogue operands only: equal
#ifndef SRGANFD_M16_PIPE
#define SRGANFD_M16_PIPE 4
#endif

namespace srganfd {
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

// TS = tap split: the weight slab of a 32-channel chunk is staged in TS pieces of KS/TS kernel rows (the 4x4 stride-2 kernel: 64 KiB of
// weights per chunk next to a 42 KiB patch would leave room for ONE workgroup per CU; in halves two fit)
template <typename T, int KS, int STRIDE, int MR, int WR, int WN, int TS = 1>
struct ConvCfg {
  static constexpr int NWAVES = WR * WN, NTHR = 64 * NWAVES;
  static constexpr int KT = KS * KS;
  static constexpr int TW = 32, TH = WR * MR;
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
  static constexpr int XI = (NX + NTHR - 1) / NTHR;
  static_assert(KS % TS == 0, "tap split by kernel rows");
  static constexpr int WS_BYTES = WN_BYTES / TS;              // one n-tile's slab piece of one stage
  static constexpr int NW16 = WN * WS_BYTES / 16;
  static constexpr int WI = (NW16 + NTHR - 1) / NTHR;
  static constexpr int STAGE_BYTES = XBYTES + WN * WS_BYTES;
  static constexpr int NB = 32 * WN;                           // output channels per workgroup
  static constexpr int EPI_BYTES = TH * TW * NB * 4;           // fp32 tile for the vectorised epilogue
  static constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  static constexpr int NROWS = (MR - 1) * STRIDE + KS;         // patch rows one wave touches
  static constexpr int PIX_PER_I = NTHR / CPP;                 // pixels advanced per staging item index
  static_assert(PIX_PER_I % 32 == 0, "swizzle term must not depend on the staging item index");
  // workgroups per CU allowed by LDS (160 KiB) -> minimum waves per SIMD to ask the register allocator for
  // (at most 16 waves per CU: the kernels are written for 128 registers per lane or more)
  static constexpr int WG_BY_LDS = 160 * 1024 / LDS_BYTES, WG_BY_WAVES = 16 / NWAVES;
  static constexpr int WG_PER_CU = WG_BY_LDS < 1 ? 1 : (WG_BY_LDS < WG_BY_WAVES ? WG_BY_LDS : WG_BY_WAVES);
  static constexpr int MIN_WAVES_PER_SIMD = WG_PER_CU * NTHR / 256 < 1 ? 1 : WG_PER_CU * NTHR / 256;
};

// LDS patch layout.  bf16: 4 chunks of 16 B per pixel, chunk index XOR (pix>>2)&3.  f32: 32 dwords per
// pixel, dword index XOR (pix & 31) so 32 lanes reading one channel of 32 consecutive pixels hit 32 banks.
__device__ __forceinline__ int lds_x_bf16_off(int pix, int c16) { return pix * 64 + ((c16 ^ ((pix >> 2) & 3)) << 4); }
__device__ __forceinline__ int lds_x_f32_off(int pix, int k) { return pix * 128 + ((k ^ (pix & 31)) << 2); }
// 16x16x32 form: lane l reads pixel (l & 15), 16-byte slot (l >> 4); slot XOR 2*((pix>>2)&1) gives every 16-lane group of
// ds_read_b128 sixteen distinct slots of the 256-byte bank row for any patch alignment
__device__ __forceinline__ int lds_x16_m16_off(int pix, int c16) { return pix * 64 + ((c16 ^ (((pix >> 2) & 1) << 1)) << 4); }

// accumulators of one wave: MR rows x 32 pixels x 32 channels = 16 floats per lane and row in both MFMA forms
template <bool M16, int MR> struct AccSet;
template <int MR> struct AccSet<false, MR> {
  f32x16 a[MR];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) a[m][i] = 0.f;
  }
  __device__ __forceinline__ float get(int m, int e) const { return a[m][e]; }
  static __device__ __forceinline__ int pixel(int e, int lane) { return mfma32_row(e, lane); }
  static __device__ __forceinline__ int chan(int e, int lane) { return lane & 31; }
};
template <int MR> struct AccSet<true, MR> {
  f32x4_t a[MR][2][2];     // [row][pixel half][channel half]
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[m][q >> 1][q & 1][i] = 0.f;
  }
  __device__ __forceinline__ float get(int m, int e) const { return a[m][(e >> 3) & 1][(e >> 2) & 1][e & 3]; }
  static __device__ __forceinline__ int pixel(int e, int lane) { return 16 * ((e >> 3) & 1) + 4 * (lane >> 4) + (e & 3); }
  static __device__ __forceinline__ int chan(int e, int lane) { return 16 * ((e >> 2) & 1) + (lane & 15); }
};

// EK = epilogue kind, fixed at compile time for the hot 3x3 stride-1 16-bit launches: -1 = every operand decided at run time (any
// launch); >= 0 = vectorised epilogue with exactly the operands of the bit set (1: residual r1, 2: residual r2, 4: LeakyReLU' mask), no
// y2, no fp32 / partial-channel output.  A fixed kind carries no loads, address arithmetic, prefetch registers or branches for tensors
// the launch does not have: the four growth convs of a dense block (kind 0) and their data-gradient twins (kind 4) are 80 % of a
// generator step's launches.
template <typename T, int KS, int STRIDE, int MR, int WR, int WN, bool M16 = false, int TS = 1, int EK = -1, bool NT = false>
__device__ __forceinline__ void conv_igemm_body(const ConvK& a, char* smem) {
  static_assert(!M16 || sizeof(T) == 2, "16x16x32 is a 16-bit form");
  static_assert(TS == 1 || M16, "the tap split is built for the 16x16x32 loop");
  using C = ConvCfg<T, KS, STRIDE, MR, WR, WN, TS>;
  using Frag = typename FragAB<T>::type;
  constexpr int NTHR = C::NTHR;
  char* ldsX = smem;
  char* ldsW = smem + C::XBYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  // Persistent tiles: workgroup b runs the virtual blocks b, b + gridDim.x, ... (the host sizes the grid to the workgroups the chip
  // holds at once for the kernels that prefetch across tiles, and to one block per workgroup otherwise).  A tile's first chunk is
  // requested while the previous tile of the workgroup is still in its last MFMA phase and epilogue: measured per tile (s_memrealtime
  // stamps, profiles/r03_conv_timeline.txt) the head of a tile -- address set-up, the first loads' round trip, the first commit -- was
  // 4-6 us of 14 (64 -> 32 channels) to 27 us (192 -> 64), none of it overlapped with anything of the same workgroup.
  // Virtual block -> tile: XCD-aware bijective remap (blocks v and v+8 share an XCD and its L2; each XCD gets a contiguous range of
  // tiles), block-uniform by construction; the readfirstlane tells the compiler so (otherwise every address product downstream
  // stays in quarter-rate vector multiplies).
  constexpr bool kCls = (KS == 2 || KS == 1) && STRIDE == 1 && sizeof(T) == 2;
  int cls = 0;                                                            // output-parity class of this workgroup (kCls launches)
  auto decode = [&](int vb, int& n_, int& oy_, int& ox_, int& nb_) {
    const int bid = xcd_remap(vb, a.nblocks);
    const int t0 = (int)fast_div((unsigned)bid, (unsigned)a.nNb, a.m_nNb);
    nb_ = __builtin_amdgcn_readfirstlane(bid - t0 * a.nNb);
    if constexpr (kCls) {
      // all four output-parity classes of a stride-2 data gradient in one launch (srganfd_conv_args.out_classes): the class is the
      // slow half of the tile's channel-block index, so the 4 x nNb workgroups that read one dy patch are consecutive blocks of one
      // XCD and three of the four reads hit its L2
      if (a.cls_sh >= 0) { cls = nb_ >> a.cls_sh; nb_ &= (1 << a.cls_sh) - 1; }
    }
    const int t1 = (int)fast_div((unsigned)t0, (unsigned)a.tiles_x, a.m_tx);
    const int tx = __builtin_amdgcn_readfirstlane(t0 - t1 * a.tiles_x);
    const int t2 = (int)fast_div((unsigned)t1, (unsigned)a.tiles_y, a.m_ty);
    const int ty = __builtin_amdgcn_readfirstlane(t1 - t2 * a.tiles_y);
    n_ = __builtin_amdgcn_readfirstlane(t2);
    oy_ = ty * C::TH; ox_ = tx * C::TW;
  };

  const int Hl = a.Hin << a.up, Wl = a.Win << a.up;
  // a class launch derives the class's padding and output offset from its index: class (py, px) produces the output pixels
  // (2 oy + py, 2 ox + px); 4x4 stride-2 gradient (cls_pad 1): from the 2 x 2 window of dy that starts at (oy + py - 1, ox + px - 1)
  auto pad_y = [&]() { return kCls && a.cls_sh >= 0 ? a.pad_y - (cls >> 1) * a.cls_pad : a.pad_y; };
  auto pad_x = [&]() { return kCls && a.cls_sh >= 0 ? a.pad_x - (cls & 1) * a.cls_pad : a.pad_x; };
  auto ooy = [&]() { return kCls && a.cls_sh >= 0 ? (cls >> 1) : a.ooy; };
  auto oox = [&]() { return kCls && a.cls_sh >= 0 ? (cls & 1) : a.oox; };
  // load side of the tile being staged (may run one tile ahead of the tile being computed): 64-bit per-image base (block-uniform,
  // scalar registers) + 32-bit offsets inside the image (host-checked)
  const T* __restrict__ xg = nullptr;

  // 16x16x32 form, stride 1: the slot swizzle is keyed on the patch COLUMN, 2*((px>>2)&1).  What keeps a 16-lane read group
  // conflict free is that the four lanes whose pixels are 4 apart (same 64-byte position of the 256-byte bank row) alternate
  // their slot bit, and 4 columns apart they do whatever the row start is.  The fragment address is then lane term (3 kernel
  // columns) + wave-uniform row offset + immediate, instead of one precomputed VGPR per (row, column, pixel half) -- those 24
  // registers pushed the 64-channel kernel over its 128 and spilled a prefetch pointer (scratch reload + vmcnt(0) inside the loop).
  constexpr bool kColSwz = M16 && STRIDE == 1;
  constexpr bool kDeint = M16 && STRIDE == 2;
  constexpr int kHalf = (C::PC + 1) / 2;                  // even columns of a patch row
  // storage pixel (within a patch row) of column 2*l + kx + 32*ph, before the lane term l: the A-fragment reads of the stride-2 forms
  auto deint_col = [](int kx, int ph) { return (kx & 1) * kHalf + (kx >> 1) + 16 * ph; };
  constexpr int kM16Pipe = WN == 2 ? SRGANFD_M16_PIPE : SRGANFD_M16_PIPE - 1;     // fragment read-ahead (what fits 128 VGPRs) of the 3x3 stride-1 16x16x32 loop (0 = the compiler's own schedule)
  int ldsxo[kColSwz ? C::XI : 1];
  // per-thread source offsets (elements) of the X staging items; -1 = zero padding
  int xoff[C::XI];
  // patch position of staging item i (tile-independent).  stride 2 (16x16x32 form): a patch row is stored even columns first, then
  // the odd ones -- the 16 lanes of a fragment read want columns 2*l + kx, which interleaved sit 128 bytes apart (four 16-byte slots
  // of the 256-byte bank row for 16 lanes: 4-way conflicts on every A read); de-interleaved they are 16 consecutive storage pixels,
  // the stride-1 pattern the slot swizzle spreads
  auto item_pos = [&](int tid_, int i, int& py, int& px, int& c16) {
    const int item = tid_ + i * NTHR;
    const int pix = item / C::CPP;
    c16 = item % C::CPP;
    py = pix / C::PC;
    const int pq = pix % C::PC;
    px = kDeint ? (pq < kHalf ? 2 * pq : 2 * (pq - kHalf) + 1) : pq;
  };
  if constexpr (kColSwz) {
#pragma unroll
    for (int i = 0; i < C::XI; ++i) {
      int py, px, c16;
      item_pos(tid, i, py, px, c16);
      ldsxo[i] = ((tid + i * NTHR) / C::CPP) * 64 + ((c16 ^ (((px >> 2) & 1) << 1)) << 4);
    }
  }
  auto tile_offsets = [&](int oy_, int ox_) {
    // (the thread id goes through an opaque copy: otherwise the patch positions of all staging items -- tile-independent -- are hoisted
    // out of the tile loop and held in ~15 registers across it, which this kernel does not have)
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
#pragma unroll
    for (int i = 0; i < C::XI; ++i) {
      int py, px, c16;
      item_pos(tid_o, i, py, px, c16);
      const int gy = oy_ * STRIDE - pad_y() + py, gx = ox_ * STRIDE - pad_x() + px;
      const bool ok = tid_o + i * NTHR < C::NX && gy >= 0 && gy < Hl && gx >= 0 && gx < Wl;
      xoff[i] = ok ? ((gy >> a.up) * a.Win + (gx >> a.up)) * a.x_ps + a.x_base + c16 * C::E16 : -1;
    }
  };
  // LDS destination of staging item i = ldsx0 + i * (PIX_PER_I * PIXB): the swizzle term is i-invariant
  int ldsx0;
  {
    const int pix = tid / C::CPP, c16 = tid % C::CPP;
    if constexpr (M16) ldsx0 = lds_x16_m16_off(pix, c16);
    else if constexpr (sizeof(T) == 2) ldsx0 = lds_x_bf16_off(pix, c16);
    else ldsx0 = pix * 128;   // f32: per-dword XOR below
  }
  const u32x4* __restrict__ wgp = nullptr;

  // bf16: register prefetch of the next chunk (issue-early / write-late).  f32 (parity mode) stages
  // synchronously: its 2x larger tiles would not fit the register budget next to the accumulators.
  constexpr bool kPrefetch = sizeof(T) == 2;
  constexpr int XR = kPrefetch ? C::XI : 1, WRG = kPrefetch ? C::WI : 1;
  u32x4 xr[XR];
  u32x4 wrg[WRG];
  // 16-bit staging loads go through buffer descriptors: padding pixels / idle lanes carry an offset beyond the descriptor's range and
  // the hardware range check returns zeros -- no zero-initialised destination registers (32 v_mov per chunk), no exec-mask branches
  // around the loads.  Descriptors are built from wave-uniform values (image base, weight base of this output-channel block).
  constexpr bool kBuf = sizeof(T) == 2;
  constexpr int kOob = 0x7fffffff;
  // (the base pointers are block-uniform, but 64-bit products are computed in VGPRs: without the explicit readfirstlane the backend
  // wraps every buffer load in a waterfall loop over "divergent" descriptors)
  auto uniform_ptr = [](const void* p) -> void* {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
  };
  auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.x), (short)0, 0, 0x00020000);      // re-pointed per tile by setup_loads
  auto wrsrc = xrsrc;
  auto load_x = [&](int i, int chunk) -> u32x4 {
    if constexpr (kBuf) {
      typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned v4u;
      const v4u r = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xoff[i] >= 0 ? xoff[i] * (int)sizeof(T) : kOob, chunk * a.x_cs * (int)sizeof(T), 0);
      return __builtin_bit_cast(u32x4, r);
    } else {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (xoff[i] >= 0) v = *(const u32x4*)(xg + xoff[i] + chunk * a.x_cs);
      return v;
    }
  };
  auto load_w = [&](int i, int chunk, int th = 0) -> u32x4 {
    const int item = tid + i * NTHR;
    // LDS slab order [n-tile][tap][kstep][lane]; global order [n-tile][chunk][tap][kstep][lane]; piece th = taps [th, th+1) * KT/TS
    const int nn = item / (C::WS_BYTES / 16), rem = item % (C::WS_BYTES / 16);
    if constexpr (kBuf) {
      typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned v4u;
      const bool ok = item < C::NW16;
      const v4u r = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, ok ? (nn * a.nChunks * (C::WN_BYTES / 16) + rem) * 16 : kOob,
                                                          (chunk * (C::WN_BYTES / 16) + th * (C::WS_BYTES / 16)) * 16, 0);
      return __builtin_bit_cast(u32x4, r);
    } else {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (item < C::NW16) v = wgp[(nn * a.nChunks + chunk) * (C::WN_BYTES / 16) + th * (C::WS_BYTES / 16) + rem];
      return v;
    }
  };
  auto store_x = [&](int i, u32x4 v) {
    const int item = tid + i * NTHR;
    if (item < C::NX) {
      if constexpr (kColSwz) {
        *(u32x4*)(ldsX + ldsxo[i]) = v;
      } else if constexpr (sizeof(T) == 2) {
        *(u32x4*)(ldsX + ldsx0 + i * (C::PIX_PER_I * C::PIXB)) = v;
      } else {
        const int pix = item / C::CPP, c16 = item % C::CPP;
        *(unsigned int*)(ldsX + lds_x_f32_off(pix, c16 * 4 + 0)) = v[0];
        *(unsigned int*)(ldsX + lds_x_f32_off(pix, c16 * 4 + 1)) = v[1];
        *(unsigned int*)(ldsX + lds_x_f32_off(pix, c16 * 4 + 2)) = v[2];
        *(unsigned int*)(ldsX + lds_x_f32_off(pix, c16 * 4 + 3)) = v[3];
      }
    }
  };
  auto store_w = [&](int i, u32x4 v) {
    const int item = tid + i * NTHR;
    if (item < C::NW16) *(u32x4*)(ldsW + item * 16) = v;
  };
  auto prefetch = [&](int chunk) {
    if constexpr (kPrefetch) {
#pragma unroll
      for (int i = 0; i < C::XI; ++i) xr[i] = load_x(i, chunk);
#pragma unroll
      for (int i = 0; i < C::WI; ++i) wrg[i] = load_w(i, chunk);
    }
  };
  auto commit = [&](int chunk) {
    if constexpr (kPrefetch) {
#pragma unroll
      for (int i = 0; i < C::XI; ++i) store_x(i, xr[i]);
#pragma unroll
      for (int i = 0; i < C::WI; ++i) store_w(i, wrg[i]);
    } else {
#pragma unroll 4
      for (int i = 0; i < C::XI; ++i) store_x(i, load_x(i, chunk));
#pragma unroll 4
      for (int i = 0; i < C::WI; ++i) store_w(i, load_w(i, chunk));
    }
  };

  float bvn0 = 0.f, bvn1 = 0.f;
  // point the load side at virtual block vb: image / weight-block descriptors and the staging offsets of its tile
  auto setup_loads = [&](int vb) {
    int n_, oy_, ox_, nb_;
    decode(vb, n_, oy_, ox_, nb_);
    xg = (const T*)a.x + (size_t)n_ * a.Hin * a.Win * a.xC;
    const int wb_ = kCls && a.cls_sh >= 0 ? nb_ + (cls << a.cls_sh) : nb_;      // the classes' packs follow each other (host-checked)
    wgp = (const u32x4*)a.w + (size_t)wb_ * WN * a.nChunks * (C::WN_BYTES / 16);
    if constexpr (kBuf) {
      xrsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(xg), (short)0, (int)((unsigned)a.Hin * (unsigned)a.Win * (unsigned)a.xC * (unsigned)sizeof(T)), 0x00020000);
      wrsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(wgp), (short)0, (int)((unsigned)WN * (unsigned)a.nChunks * (unsigned)C::WN_BYTES), 0x00020000);
    }
    // the two bias values this lane's epilogue adds, requested here so that they are OLDER than the tile's staging loads: the copy at
    // the top of the tile then waits for them alone.  (Loaded at the top of the tile they were younger than the next tile's first
    // loads, and the epilogue's use of them drained those loads -- s_waitcnt vmcnt(0) -- before the tile could finish.)
    if constexpr (sizeof(T) == 2) {
      const int cow_ = (nb_ * WN + wn) * 32;
      bvn0 = bvn1 = 0.f;
      if (a.bias && (EK >= 0 || a.fast_epi)) { bvn0 = a.bias[cow_ + AccSet<M16, MR>::chan(0, lane)]; bvn1 = a.bias[cow_ + AccSet<M16, MR>::chan(4, lane)]; }
    }
    tile_offsets(oy_, ox_);
  };
  // the first stage's loads of the tile the load side points at
  auto first_loads = [&]() {
    if constexpr (kPrefetch) {
#pragma unroll
      for (int i = 0; i < C::XI; ++i) xr[i] = load_x(i, 0);
#pragma unroll
      for (int i = 0; i < C::WI; ++i) wrg[i] = load_w(i, 0, 0);
    }
  };
  // tiles after the first: requested during the previous tile's last MFMA phase (the kinds whose epilogue leaves room for the 32
  // staging registers), else after its epilogue
  constexpr bool kCross = kPrefetch && EK >= 0 && (EK & SRGANFD_CROSS_MASK) == 0;

  AccSet<M16, MR> A_;
  auto& acc = A_.a;

  // this lane's A-fragment base: patch pixel (wr*MR*S, r*S), B-fragment base: n-tile wn
  const int pix00 = (wr * MR * STRIDE) * C::PC + r * STRIDE;
  const char* ldsWn = ldsW + wn * C::WS_BYTES + lane * C::FRAGB;

  // 16x16x32 form, stride 1: lane term of the fragment address per kernel column (chunk-invariant)
  int colt[KS];
#pragma unroll
  for (int kx = 0; kx < KS; ++kx) colt[kx] = ((lane & 15) + kx) * 64 + (((lane >> 4) ^ (((((lane & 15) + kx) >> 2) & 1) << 1)) << 4);

  // epilogue operands requested before the main loop (their load latency used to sit at the head of every tile's epilogue): the
  // device-side scale is a scalar load, the two bias values of this lane's channels cost two registers through the loop
  float alpha = a.alpha;
  if (a.alpha_dev) alpha *= *a.alpha_dev;
  int vt = blockIdx.x;
  setup_loads(vt);
  first_loads();
  for (;;) {
  // (cin >= 32 is host-checked; without the hint the compiler sees a path from the bias loads below to the epilogue that skips the
  // chunk loop's waits, and guards the epilogue with s_waitcnt vmcnt(0) -- which would also wait for the next tile's staging loads)
  __builtin_assume(a.nChunks >= 1);
  int vt_next = vt + (int)gridDim.x;
  bool more = kCross && vt_next < a.nblocks;      // this workgroup has another tile after this one (the host gives the other kinds one block per workgroup)
  int n, oy0, ox0, nb;
  decode(vt, n, oy0, ox0, nb);
  A_.zero();
  const int cow = (nb * WN + wn) * 32;      // first output channel of this wave; element e of a lane sits at channel cow + chan(e, lane)
  const float bvh0 = bvn0, bvh1 = bvn1;     // this tile's bias values (requested with its first loads)
  if constexpr (TS > 1) {
    // Tap-split main loop (16x16x32 form): stage (chunk, th) holds the chunk's patch and kernel rows [th*KYS, (th+1)*KYS) of its
    // weights; the patch is committed with th == 0 and stays for the chunk's TS weight pieces.  Same issue-early / write-late
    // staging as below, one barrier pair per stage.
    constexpr int KYS = KS / TS, NR_ = (MR - 1) * STRIDE + KYS;
    const int l15 = lane & 15, sl = lane >> 4;
    const int pixb = (wr * MR * STRIDE) * C::PC + l15 * STRIDE;
    for (int chunk = 0; chunk < a.nChunks; ++chunk) {
      static_for<TS>([&](auto thc) {
        constexpr int th = decltype(thc)::v;
        __syncthreads();
        if constexpr (th == 0) {
#pragma unroll
          for (int i = 0; i < C::XI; ++i) store_x(i, xr[i]);
        }
#pragma unroll
        for (int i = 0; i < C::WI; ++i) store_w(i, wrg[i]);
        __syncthreads();
        if constexpr (th + 1 < TS) {
#pragma unroll
          for (int i = 0; i < C::WI; ++i) wrg[i] = load_w(i, chunk, th + 1);
        } else if (chunk + 1 < a.nChunks) {
#pragma unroll
          for (int i = 0; i < C::XI; ++i) xr[i] = load_x(i, chunk + 1);
#pragma unroll
          for (int i = 0; i < C::WI; ++i) wrg[i] = load_w(i, chunk + 1, 0);
        } else if (kCross && more) {
          setup_loads(vt_next);
          first_loads();
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          Frag av[NR_][2];
#pragma unroll
          for (int rr = 0; rr < NR_; ++rr)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
              if constexpr (kDeint) av[rr][ph] = *(const Frag*)(ldsX + lds_x16_m16_off((wr * MR * STRIDE + th * KYS + rr) * C::PC + l15 + deint_col(kx, ph), sl));
              else av[rr][ph] = *(const Frag*)(ldsX + lds_x16_m16_off(pixb + (th * KYS + rr) * C::PC + kx + 16 * ph * STRIDE, sl));
#pragma unroll
          for (int kyl = 0; kyl < KYS; ++kyl) {
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
              const Frag bq = *(const Frag*)(ldsWn + ((kyl * KS + kx) * 2 + nh) * 64 * C::FRAGB);
#pragma unroll
              for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) acc[m][ph][nh] = mfma16<T>(av[m * STRIDE + kyl][ph], bq, acc[m][ph][nh]);
            }
          }
        }
        __builtin_amdgcn_s_setprio(0);
      });
    }
  } else {
  for (int chunk = 0; chunk < a.nChunks; ++chunk) {
    __syncthreads();
    commit(chunk);
    __syncthreads();
    if (chunk + 1 < a.nChunks) prefetch(chunk + 1);
    else if (kCross && more) {
      setup_loads(vt_next);
      first_loads();
    }

    auto col_body = [&](int kx, int s) {
      if constexpr (!M16) {
      Frag av[C::NROWS];
#pragma unroll
      for (int rr = 0; rr < C::NROWS; ++rr) {
        const int pix = pix00 + rr * C::PC + kx;
        if constexpr (sizeof(T) == 2) av[rr] = *(const Frag*)(ldsX + lds_x_bf16_off(pix, 2 * s + h));
        else av[rr] = *(const Frag*)(ldsX + lds_x_f32_off(pix, 2 * s + h));
      }
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
        const Frag bq = *(const Frag*)(ldsWn + ((ky * KS + kx) * C::KSTEPS + s) * 64 * C::FRAGB);
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[m] = mfma32<T>(av[m * STRIDE + ky], bq, acc[m]);
      }
      }
    };
    if constexpr (M16) {
      // one K step = the whole 32-channel chunk; per kernel column: (NROWS x 2 pixel halves) A fragments, (KS x 2 channel halves) B
      const int l15 = lane & 15, sl = lane >> 4;
      const int pixb = (wr * MR * STRIDE) * C::PC + l15 * STRIDE;
      const char* ldsXw = ldsX + (wr * MR * STRIDE) * C::PC * 64;        // this wave's first patch row (wave-uniform)
      __builtin_amdgcn_s_setprio(1);
      if constexpr (kColSwz && KS == 3 && MR == 2 && kM16Pipe > 0) {
        // software-pipelined fragment reads (see m16_need): every ds_read_b128 is issued kM16Pipe fragments ahead of the MFMA that
        // needs it, across the three kernel columns of the chunk; same accumulation order per accumulator as the plain loop below
        constexpr int NL = 14 * KS, NM = 24 * KS;
        Frag F[NL];
        static_for<NM>([&](auto ic) {
          constexpr int i = decltype(ic)::v;
          constexpr int c = i / 24, j = i % 24, gq = j / 4, t = j % 4;
          constexpr int hi = m16_pipe_hi(i, kM16Pipe, NL), lo = i == 0 ? 0 : m16_pipe_hi(i - 1, kM16Pipe, NL) + 1;
          static_for<hi - lo + 1>([&](auto jc) {
            constexpr int n = lo + decltype(jc)::v;
            constexpr int kx = n / 14, l = n % 14;
            constexpr bool isB = l == 0 || l == 5 || l == 6 || l == 9 || l == 10 || l == 13;
            if constexpr (isB) {
              constexpr int ky = l < 6 ? 0 : (l < 10 ? 1 : 2), nh = (l == 5 || l == 9 || l == 13) ? 1 : 0;
              F[n] = *(const Frag*)(ldsWn + ((ky * KS + kx) * 2 + nh) * 64 * C::FRAGB);
            } else {
              constexpr int q = l < 5 ? l - 1 : (l < 9 ? l - 3 : l - 5), rr = q >> 1, ph = q & 1;
              F[n] = *(const Frag*)(ldsXw + colt[kx] + (rr * C::PC + 16 * ph) * 64);
            }
          });
          constexpr int ky = gq >> 1, nh = gq & 1, m = t >> 1, ph = t & 1;
          acc[m][ph][nh] = mfma16<T>(F[14 * c + m16_aidx(m + ky, ph)], F[14 * c + m16_bidx(ky, nh)], acc[m][ph][nh]);
          __builtin_amdgcn_sched_barrier(0);
        });
      } else
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        Frag av[C::NROWS][2];
        const int colterm = (l15 + kx) * 64 + ((sl ^ ((((l15 + kx) >> 2) & 1) << 1)) << 4);     // stride 1: lane term of column kx
#pragma unroll
        for (int rr = 0; rr < C::NROWS; ++rr)
#pragma unroll
          for (int ph = 0; ph < 2; ++ph) {
            if constexpr (kColSwz) av[rr][ph] = *(const Frag*)(ldsXw + colterm + (rr * C::PC + 16 * ph) * 64);
            else if constexpr (kDeint) av[rr][ph] = *(const Frag*)(ldsX + lds_x16_m16_off((wr * MR * STRIDE + rr) * C::PC + l15 + deint_col(kx, ph), sl));
            else av[rr][ph] = *(const Frag*)(ldsX + lds_x16_m16_off(pixb + rr * C::PC + kx + 16 * ph * STRIDE, sl));
          }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) {
            const Frag bq = *(const Frag*)(ldsWn + ((ky * KS + kx) * 2 + nh) * 64 * C::FRAGB);
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
              for (int ph = 0; ph < 2; ++ph) acc[m][ph][nh] = mfma16<T>(av[m * STRIDE + ky][ph], bq, acc[m][ph][nh]);
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
    } else if constexpr (sizeof(T) == 2 && KS == 3 && STRIDE == 1 && MR == 2) {
      // Software-pipelined fragment reads: the chunk's 42 ds_read_b128 and 36 MFMAs in one fixed issue order, every read kD
      // fragments ahead of the MFMA that consumes it (the compiler's own order is read -> s_waitcnt lgkmcnt(0) -> MFMA on two
      // fragment registers: each MFMA eats a full LDS round trip).  Same accumulation order as the plain loop below, so the
      // results are bitwise equal; measured -4 % on the 64-channel conv, -1 % on the 32-channel one.
      // col-body c = kx * KSTEPS + s: loads A0 B0 A1 B1 A2 B2 A3 (rows rr = j/2, taps ky = j/2), MFMAs (A0,B0)->acc0 (A1,B0)->acc1
      // (A1,B1)->acc0 (A2,B1)->acc1 (A2,B2)->acc0 (A3,B2)->acc1.
      constexpr int kD = WN == 2 ? 3 : 2;        // read-ahead that fits 128 VGPRs: 64-channel tiles 3 fragments, 32-channel tiles 2
      constexpr int NL = 7 * KS * C::KSTEPS, NM = 6 * KS * C::KSTEPS;
      Frag F[NL];
      __builtin_amdgcn_s_setprio(1);
      static_for<NM>([&](auto ic) {
        constexpr int i = decltype(ic)::v;
        constexpr int c = i / 6, j = i % 6;
        constexpr int hi = pipe_hi(i, kD, NL), lo = i == 0 ? 0 : pipe_hi(i - 1, kD, NL) + 1;
        static_for<hi - lo + 1>([&](auto jc) {
          constexpr int n = lo + decltype(jc)::v;
          constexpr int cc = n / 7, jj = n % 7, kx = cc / C::KSTEPS, ss = cc % C::KSTEPS;
          if constexpr ((jj & 1) == 0) F[n] = *(const Frag*)(ldsX + lds_x_bf16_off(pix00 + (jj / 2) * C::PC + kx, 2 * ss + h));
          else F[n] = *(const Frag*)(ldsWn + (((jj / 2) * KS + kx) * C::KSTEPS + ss) * 64 * C::FRAGB);
        });
        acc[j & 1] = mfma32<T>(F[7 * c + 2 * ((j + 1) / 2)], F[7 * c + 2 * (j / 2) + 1], acc[j & 1]);
        __builtin_amdgcn_sched_barrier(0);
      });
      __builtin_amdgcn_s_setprio(0);
    } else if constexpr (sizeof(T) == 2) {
      __builtin_amdgcn_s_setprio(1);   // waves in their MFMA phase win issue arbitration over waves that are staging (+1-3 %)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
        for (int s2 = 0; s2 < C::KSTEPS; ++s2) col_body(kx, s2);
      }
      __builtin_amdgcn_s_setprio(0);
    } else {
#pragma unroll 1
      for (int kx = 0; kx < KS; ++kx) {
#pragma unroll 2
        for (int s2 = 0; s2 < C::KSTEPS; ++s2) col_body(kx, s2);
      }
    }
  }

  }

  // ---- epilogue (see srganfd.h for the formula) ----
  [&]() __attribute__((always_inline)) {
  if constexpr (EK == 0 && M16 && WN == 1 && sizeof(T) == 2) {
    // Kind 0, 32-channel tiles (the growth convs of a dense block): bias + activation in registers, rounded to T there (the value
    // the fp32 tile path would round after its LDS round trip: same bits), and a 16-bit CHANNEL-major LDS tile per image row --
    // a lane's four accumulator registers of one (row, pixel half, channel half) are four consecutive pixels of one channel, so
    // they go out as one ds_write_b64; ds_read_b64_tr_b16 hands them back pixel-major (lane i of a 16-lane group: pixel i, four
    // channels), two reads = the 16 bytes of (pixel, 8 channels) one lane stores.  Every wave reads only the rows it wrote: one
    // barrier (staging buffers free) instead of two, 8 LDS writes + 8 reads per lane instead of 16 + 8 wider ones, no fp32 tile
    // unpacking / conversion after the reads.  Row tile = 32 channels x 64 bytes, dword index XOR (channel & 14): conflict-free
    // for the ds_write_b64 (16 channels of one 8-byte column) and for the transposed reads (4 channels x 16 pixels per group).
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));      // keeps this block's lane-only address terms out of the registers carried around the tile loop
    const int cl = lane_o & 15, g4 = lane_o >> 4;
    const float neg = a.act == SRGANFD_ACT_LRELU ? a.slope : (a.act == SRGANFD_ACT_RELU ? 0.f : 1.f);
    const float ps_pos = a.post_scale, ps_neg = neg * a.post_scale;
    char* rowt = smem + (wr * MR) * 2048;                                   // this wave's first row tile
    const int w0 = cl * 64 + (((2 * g4) ^ (cl & 14)) << 2);                 // write: channel cl (+16 nh), pixels 4 g4 .. 4 g4 + 3 (+16 ph)
    const int rq = 8 * g4 + ((lane_o >> 2) & 3), rp = lane_o & 3;               // read: this lane addresses channel rq (+4), pixels 4 rp .. (+16 pb)
    const int r0 = rq * 64 + (((2 * rp) ^ (rq & 14)) << 2), r1 = (rq + 4) * 64 + (((2 * rp) ^ ((rq + 4) & 14)) << 2);
    __syncthreads();   // all waves are done with the staging buffers
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          float v4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = alpha * acc[m][ph][nh][i] + (nh ? bvh1 : bvh0);
            v4[i] = v * (v > 0.f ? ps_pos : ps_neg);
          }
          u32x2_t pk;
          if constexpr (Elem<T>::kDtype == SRGANFD_F16) {
            typedef __attribute__((ext_vector_type(4))) _Float16 h4;
            const h4 hv = {(_Float16)v4[0], (_Float16)v4[1], (_Float16)v4[2], (_Float16)v4[3]};
            pk = __builtin_bit_cast(u32x2_t, hv);
          } else {
            pk = u32x2_t{(unsigned)f2bf(v4[0]) | ((unsigned)f2bf(v4[1]) << 16), (unsigned)f2bf(v4[2]) | ((unsigned)f2bf(v4[3]) << 16)};
          }
          *(u32x2_t*)(rowt + m * 2048 + nh * 1024 + (w0 ^ (ph << 5))) = pk;
        }
    const size_t img = (size_t)n * a.HoutF * a.WoutF;
    const int cch = nb * C::NB + 8 * g4;                                    // this lane's 8 output channels
    const int cc = a.y_c0 + cch;
    T* ybase = (T*)a.y + img * a.yC + ((cc >> 5) * a.y_gs + (cc & 31));
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const u32x2_t lo = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(rowt + m * 2048 + (r0 ^ (pb << 5)))));
        const u32x2_t hi = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(rowt + m * 2048 + (r1 ^ (pb << 5)))));
        const int oy = oy0 + wr * MR + m, ox = ox0 + 16 * pb + cl;
        if (oy < a.Hout && ox < a.Wout) {
          const int pp = (oy * a.osy + ooy()) * a.WoutF + ox * a.osx + oox();
          *(u32x4*)(ybase + pp * a.y_ps) = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
      }
    return;
  }
  constexpr bool kR1 = EK < 0 || (EK & 1), kR2 = EK < 0 || (EK & 2), kMk = EK < 0 || (EK & 4);   // operands this instantiation can have
  const bool has_r1 = EK < 0 ? a.r1 != nullptr : kR1, has_r2 = EK < 0 ? a.r2 != nullptr : kR2, has_mk = EK < 0 ? a.mask != nullptr : kMk;
  if (EK >= 0 || a.fast_epi) {
    // (1) per-channel part (alpha, bias, activation, scale) on the accumulators -> fp32 LDS tile
    // [pixel][channel]; (2) 16 output bytes per lane: residuals / LeakyReLU' mask via 16-byte global
    // loads, 16-byte stores.
    float* tile = (float*)smem;
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));     // opaque copy: this block's thread-only address terms stay out of the tile loop's carried registers
    // The residual / mask tensors of the WHOLE tile are requested here, before the accumulators go through LDS: inside the store loop
    // each item's 16-byte load was followed by its s_waitcnt vmcnt(0) -- one exposed load latency per item, four per tile.  (16-bit
    // kernels with four items per thread: every 3x3 / 4x4 / 2x2 tile shape; up to 48 registers, free at this point: the staging ones.)
    constexpr int CPq = C::NB / C::E16, ITEMSq = C::TH * 32 * CPq, EIq = (ITEMSq + NTHR - 1) / NTHR;
    // (not in the kinds that run persistent: there the staging registers hold the next tile's first chunk through the epilogue)
    // (kinds that run persistent request them LATE, after the accumulators have gone to the LDS tile: the staging registers hold the
    // next tile's first chunk through the epilogue, the accumulators' 32 registers are what is free)
    constexpr bool kEpiPre = sizeof(T) == 2 && EIq <= 4;
    constexpr bool kEpiLate = kEpiPre && kCross;
    u32x4 pre_r1[kEpiPre && kR1 ? EIq : 1], pre_r2[kEpiPre && kR2 ? EIq : 1], pre_m[kEpiPre && kMk ? EIq : 1];
    auto request_operands = [&]() __attribute__((always_inline)) {
    if constexpr (kEpiPre && (kR1 || kR2 || kMk)) {
      const size_t imgq = (size_t)n * a.HoutF * a.WoutF;
#pragma unroll
      for (int e = 0; e < EIq; ++e) {
        const int item = tid_e + e * NTHR;
        const int pix = item / CPq, ck = item % CPq;
        const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
        if constexpr (kR1) pre_r1[e] = u32x4{0u, 0u, 0u, 0u};
        if constexpr (kR2) pre_r2[e] = u32x4{0u, 0u, 0u, 0u};
        if constexpr (kMk) pre_m[e] = u32x4{0u, 0u, 0u, 0u};
        if (item < ITEMSq && oy < a.Hout && ox < a.Wout) {
          const int p = (oy * a.osy + ooy()) * a.WoutF + ox * a.osx + oox();
          const int cch = nb * C::NB + ck * C::E16;
          auto ld = [&](const void* base, int Cs, int c0, int ps, int gs) -> u32x4 {
            const int cc = c0 + cch;
            return *(const u32x4*)((const T*)base + imgq * Cs + (p * ps + (cc >> 5) * gs + (cc & 31)));
          };
          if constexpr (kR1) { if (has_r1) pre_r1[e] = ld(a.r1, a.r1C, a.r1_c0, a.r1_ps, a.r1_gs); }
          if constexpr (kR2) { if (has_r2) pre_r2[e] = ld(a.r2, a.r2C, a.r2_c0, a.r2_ps, a.r2_gs); }
          if constexpr (kMk) { if (has_mk) pre_m[e] = ld(a.mask, a.mC, a.m_c0, a.m_ps, a.m_gs); }
        }
      }
    }
    };
    if constexpr (!kEpiLate) request_operands();
    __syncthreads();   // all waves are done with the staging buffers
    {
      // a lane's 16 values per row cover one channel (32x32 form) or two (16x16 form: elements 0-3 / 8-11 vs 4-7 / 12-15)
      float bv0 = bvh0, bv1 = bvh1;
      if constexpr (sizeof(T) != 2) { bv0 = a.bias ? a.bias[cow + A_.chan(0, lane)] : 0.f; bv1 = a.bias ? a.bias[cow + A_.chan(4, lane)] : 0.f; }
      // the activation as ONE select per element: factor of the negative side = slope (LeakyReLU), 0 (ReLU), 1 (none).  Written
      // as `if (act == ...)` inside the loop the compiler emitted two scalar compares and branches per element (630 branches in this
      // epilogue's ISA), and every wave walked them.
      const float neg = a.act == SRGANFD_ACT_LRELU ? a.slope : (a.act == SRGANFD_ACT_RELU ? 0.f : 1.f);
      // post_scale * act(v) = v * (v > 0 ? post_scale : neg * post_scale): one multiply per element instead of two
      const float ps_pos = a.post_scale, ps_neg = neg * a.post_scale;
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = alpha * A_.get(m, i) + ((i >> 2) & 1 ? bv1 : bv0);
          tile[((wr * MR + m) * 32 + A_.pixel(i, lane)) * C::NB + wn * 32 + A_.chan(i, lane)] = v * (v > 0.f ? ps_pos : ps_neg);
        }
    }
    if constexpr (kEpiLate) request_operands();
    __syncthreads();
    constexpr int CP = C::NB / C::E16;                 // 16-byte output chunks per pixel
    constexpr int ITEMS = C::TH * 32 * CP;
    constexpr int EI = (ITEMS + NTHR - 1) / NTHR;
    const int cbase = nb * C::NB;
    const size_t img = (size_t)n * a.HoutF * a.WoutF;   // pixels before this image (block-uniform)
#pragma unroll kEpiPre ? 4 : 2
    for (int e = 0; e < EI; ++e) {
      const int item = tid_e + e * NTHR;
      const int pix = item / CP, ck = item % CP;
      const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
      if (item < ITEMS && oy < a.Hout && ox < a.Wout) {
        const int p = (oy * a.osy + ooy()) * a.WoutF + ox * a.osx + oox();   // pixel inside the image: 32-bit offsets (host-checked)
        float v[C::E16];
        const f32x4* tp = (const f32x4*)(tile + pix * C::NB + ck * C::E16);
        {
          const f32x4 t4 = tp[0];
          v[0] = t4[0]; v[1] = t4[1]; v[2] = t4[2]; v[3] = t4[3];
        }
        if constexpr (C::E16 == 8) {
          const f32x4 t4 = tp[1];
          v[4] = t4[0]; v[5] = t4[1]; v[6] = t4[2]; v[7] = t4[3];
        }
        const int cch = cbase + ck * C::E16;
        auto load16 = [&](const void* base, int Cs, int c0, int ps, int gs, float* out) {
          const int cc = c0 + cch;
          const T* src = (const T*)base + img * Cs + (p * ps + (cc >> 5) * gs + (cc & 31));
          if constexpr (sizeof(T) == 2) {
            unpack8<T>(*(const u32x4*)src, out);
          } else {
            const f32x4 rf = *(const f32x4*)src;
            out[0] = rf[0]; out[1] = rf[1]; out[2] = rf[2]; out[3] = rf[3];
          }
        };
        auto store16 = [&](void* base, int Cs, int c0, int ps, int gs, const float* vv) {
          const int cc = c0 + cch;
          T* dstp = (T*)base + img * Cs + (p * ps + (cc >> 5) * gs + (cc & 31));
          if constexpr (sizeof(T) == 2) {
            // NT: outputs one pass cannot keep in the 256 MiB Infinity Cache (the 256^2 / 512^2 layers at batch 32) leave with non-temporal
            // stores; a template parameter, because a run-time branch around the builtin is folded into a plain store
            if constexpr (NT) __builtin_nontemporal_store(pack8<T>(vv), (u32x4*)dstp);
            else *(u32x4*)dstp = pack8<T>(vv);
          } else {
            f32x4 o = {vv[0], vv[1], vv[2], vv[3]};
            *(f32x4*)dstp = o;
          }
        };
        if constexpr (EK < 0) { if (a.y2) store16(a.y2, a.y2C, a.y2_c0, a.y2_ps, a.y2_gs, v); }   // activation before the skip add (exact LeakyReLU' sign for backward)
        float tt[C::E16];
        auto get16 = [&](const u32x4* pre, const void* base, int Cs, int c0, int ps, int gs, float* out) {
          if constexpr (kEpiPre) unpack8<T>(pre[e], out);
          else load16(base, Cs, c0, ps, gs, out);
        };
        if constexpr (kR1) { if (has_r1) { get16(pre_r1, a.r1, a.r1C, a.r1_c0, a.r1_ps, a.r1_gs, tt);
#pragma unroll
          for (int q = 0; q < C::E16; ++q) v[q] += a.r1s * tt[q]; } }
        if constexpr (kR2) { if (has_r2) { get16(pre_r2, a.r2, a.r2C, a.r2_c0, a.r2_ps, a.r2_gs, tt);
#pragma unroll
          for (int q = 0; q < C::E16; ++q) v[q] += a.r2s * tt[q]; } }
        if constexpr (kMk) { if (has_mk) { get16(pre_m, a.mask, a.mC, a.m_c0, a.m_ps, a.m_gs, tt);
#pragma unroll
          for (int q = 0; q < C::E16; ++q) v[q] *= tt[q] > 0.f ? 1.f : a.mask_slope; } }
        store16(a.y, a.yC, a.y_c0, a.y_ps, a.y_gs, v);
      }
    }
    return;
  }
  // generic epilogue (padded channel counts, fp32 output): scalar stores
  if constexpr (EK < 0) {
  T* __restrict__ yg = (T*)a.y;
  const T* __restrict__ r1g = (const T*)a.r1;
  const T* __restrict__ r2g = (const T*)a.r2;
  const T* __restrict__ mg = (const T*)a.mask;
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const int oy = oy0 + wr * MR + m;
    if (oy >= a.Hout) continue;
    const size_t imgp = (size_t)n * a.HoutF * a.WoutF;                       // pixels before this image
    const int prow = (oy * a.osy + ooy()) * a.WoutF + oox();                   // pixel inside the image
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ox = ox0 + A_.pixel(i, lane);
      const int co = cow + A_.chan(i, lane);
      if (ox < a.Wout && co < a.cout_store) {
        const int p = prow + ox * a.osx;
        auto at = [&](int Cs, int c0, int ps, int gs) -> size_t { const int cc = c0 + co; return imgp * Cs + (size_t)(p * ps + (cc >> 5) * gs + (cc & 31)); };
        float v = alpha * A_.get(m, i) + (a.bias ? a.bias[co] : 0.f);
        if (a.act == SRGANFD_ACT_LRELU) v = v > 0.f ? v : v * a.slope;
        else if (a.act == SRGANFD_ACT_RELU) v = v > 0.f ? v : 0.f;
        v *= a.post_scale;
        if (a.y2) ((T*)a.y2)[at(a.y2C, a.y2_c0, a.y2_ps, a.y2_gs)] = Elem<T>::from_f(v);
        if (r1g) v += a.r1s * Elem<T>::to_f(r1g[at(a.r1C, a.r1_c0, a.r1_ps, a.r1_gs)]);
        if (r2g) v += a.r2s * Elem<T>::to_f(r2g[at(a.r2C, a.r2_c0, a.r2_ps, a.r2_gs)]);
        if (mg) v *= (Elem<T>::to_f(mg[at(a.mC, a.m_c0, a.m_ps, a.m_gs)]) > 0.f) ? 1.f : a.mask_slope;
        if (a.y_f32) ((float*)a.y)[at(a.yC, a.y_c0, a.y_ps, a.y_gs)] = v;
        else yg[at(a.yC, a.y_c0, a.y_ps, a.y_gs)] = Elem<T>::from_f(v);
      }
    }
  }
  }
  }();
  if (!more) break;
  if constexpr (!kCross) {
    setup_loads(vt_next);
    first_loads();
  }
  vt = vt_next;
  }   // tiles of this workgroup
}

template <typename T, int KS, int STRIDE, int MR, int WR, int WN, bool M16 = false, int TS = 1, int EK = -1, bool NT = false>
__global__ __launch_bounds__(64 * WR * WN, (ConvCfg<T, KS, STRIDE, MR, WR, WN, TS>::MIN_WAVES_PER_SIMD)) void conv_igemm_kernel(const ConvK a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  conv_igemm_body<T, KS, STRIDE, MR, WR, WN, M16, TS, EK, NT>(a, smem);
}

// compute units of the current device (cached per device id); 256 when nothing can be asked (dry runs on the CPU)
int conv_device_cus() {
  static int cus[64] = {0};
  if (g_dry_run) return 256;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int& c = cus[dev & 63];
  if (!c) {
    int v = 0;
    c = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return c;
}

template <typename T, int KS, int STRIDE, int MR, int WR, int WN, bool M16 = false, int TS = 1, int EK = -1, bool NT = false>
static int launch_conv(const ConvK& k, int cout, hipStream_t stream) {
  using C = ConvCfg<T, KS, STRIDE, MR, WR, WN, TS>;
  auto kern = conv_igemm_kernel<T, KS, STRIDE, MR, WR, WN, M16, TS, EK, NT>;
  if (g_describe) {
    char ek[8] = "";
    if (EK >= 0) snprintf(ek, sizeof(ek), ",E%d", EK);
    snprintf(g_describe, g_describe_len, "conv_igemm_kernel<%s,KS=%d,S=%d,MR=%d,WR=%d,WN=%d%s%s%s>%s", dtype_name<T>(), KS, STRIDE, MR, WR, WN, M16 ? ",M16" : "", TS > 1 ? ",TS=2" : "", ek,
             k.cls_sh >= 0 ? "[4 classes]" : "");
    return SRGANFD_OK;
  }
  static unsigned long long attr_done = 0;   // one bit per device: the attribute belongs to the device's code object
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
  if (k.cls_sh >= 0) {
    // class launch: 4 * nNb blocks per tile, class-major (see the kernel's decode); the shift needs a power-of-two block count
    if ((KS != 2 && KS != 1) || STRIDE != 1 || sizeof(T) != 2 || (kk.nNb & (kk.nNb - 1)))
      return set_err(SRGANFD_EINVAL, "conv2d: out_classes needs ksize 2 or 1, stride 1 and cout / %d a power of two", C::NB);
    kk.cls_sh = __builtin_ctz((unsigned)kk.nNb);
    kk.nNb *= 4;
  }
  kk.tiles_x = ceil_div(k.Wout, C::TW);
  kk.tiles_y = ceil_div(k.Hout, C::TH);
  const long long nblk = (long long)k.N * kk.tiles_x * kk.tiles_y * kk.nNb;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return set_err(SRGANFD_EINVAL, "conv2d: bad grid %lld", nblk);
  kk.nblocks = (int)nblk;
  kk.m_nNb = div_magic((unsigned)kk.nNb, (unsigned long long)nblk);
  kk.m_tx = div_magic((unsigned)kk.tiles_x, (unsigned long long)nblk);
  kk.m_ty = div_magic((unsigned)kk.tiles_y, (unsigned long long)nblk);
  // kinds that prefetch across tiles run as persistent workgroups: as many as the chip holds at once (a multiple of 8, so that the
  // virtual blocks v, v + grid, ... of one workgroup keep their XCD class in xcd_remap)
  long long grid = nblk;
  if (EK >= 0 && (EK & SRGANFD_CROSS_MASK) == 0 && sizeof(T) == 2) {
    const long long slots = (long long)conv_device_cus() * C::WG_PER_CU;
    if (slots >= 8 && nblk > slots) grid = slots / 8 * 8;
  }
  SRGANFD_LAUNCH(kern, dim3((unsigned)grid), dim3(C::NTHR), C::LDS_BYTES, stream, kk);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}



// MFMA form of the 16-bit kernels: v_mfma_f32_16x16x32 for every kernel shape (srganfd_get_mfma16() == 3).  Same-box A/B of the training
// steps against the 32x32x16 form (profiles/r02_mfma16_ab.txt): generator-only 78.3 -> 69.4 ms, GAN 179.6 -> 173.1 ms; the chip runs these
// steps power-limited (1.2 kW, clock at 2.06 of 2.4 GHz) and holds a higher clock on this form.  The 32x32x16 instantiations of the
// 16-bit kernels and the level switch live in the experiment sources (tools/experiments/r3_src).
int g_mfma16 = 3;
// non-temporal stores for outputs > 192 MB (same-box A/B switch: SRGANFD_CONV_NT=0 in the environment of the process that loads the library)
static const bool g_conv_nt = [] { const char* e = getenv("SRGANFD_CONV_NT"); return !(e && e[0] == '0'); }();

// does the kernel that consumes a packed operand of this kernel size / output width read 16x16x32 B fragments?  (pack.hip asks too)
bool conv_uses_m16(int dtype, int ksize, int cout) {
  (void)ksize; (void)cout;
  return dtype != SRGANFD_F32;
}

template <typename T>
static int dispatch_conv(const srganfd_conv_args* a, const ConvK& k, hipStream_t s) {
  const bool wide = (a->cout % 64) == 0;
  constexpr bool bf = sizeof(T) == 2;
  if constexpr (bf) {
    if (conv_uses_m16(a->dtype, a->ksize, a->cout)) {
      if (a->ksize == 3 && a->stride == 1) {
        // epilogue kind fixed at compile time where the launch has the vectorised epilogue and no y2 (see the kernel's EK)
        const int ek = (k.fast_epi && !k.y2) ? ((k.r1 ? 1 : 0) | (k.r2 ? 2 : 0) | (k.mask ? 4 : 0)) : -1;
        if (wide) {
          // outputs beyond the Infinity Cache: non-temporal stores (the discriminator / VGG / tail layers at 256^2 and 512^2; never the trunk)
          const size_t opix_ = (size_t)a->n * (size_t)k.HoutF * (size_t)k.WoutF;
          if (g_conv_nt && opix_ * (size_t)a->cout_store * 2 >= ((size_t)192 << 20)) {
            if (ek == 0) return launch_conv<T, 3, 1, 2, 4, 2, true, 1, 0, true>(k, a->cout, s);
            if (ek == 4) return launch_conv<T, 3, 1, 2, 4, 2, true, 1, 4, true>(k, a->cout, s);
            if (ek < 0) return launch_conv<T, 3, 1, 2, 4, 2, true, 1, -1, true>(k, a->cout, s);
          }
          if (ek == 0) return launch_conv<T, 3, 1, 2, 4, 2, true, 1, 0>(k, a->cout, s);
          if (ek == 1) return launch_conv<T, 3, 1, 2, 4, 2, true, 1, 1>(k, a->cout, s);
          if (ek == 3) return launch_conv<T, 3, 1, 2, 4, 2, true, 1, 3>(k, a->cout, s);
          if (ek == 4) return launch_conv<T, 3, 1, 2, 4, 2, true, 1, 4>(k, a->cout, s);
          return launch_conv<T, 3, 1, 2, 4, 2, true>(k, a->cout, s);
        }
        if (ek == 0) return launch_conv<T, 3, 1, 2, 8, 1, true, 1, 0>(k, a->cout, s);
        if (ek == 4) return launch_conv<T, 3, 1, 2, 8, 1, true, 1, 4>(k, a->cout, s);
        return launch_conv<T, 3, 1, 2, 8, 1, true>(k, a->cout, s);
      }
      if (a->ksize == 3 && a->stride == 2) return wide ? launch_conv<T, 3, 2, 1, 4, 2, true>(k, a->cout, s) : launch_conv<T, 3, 2, 1, 4, 1, true>(k, a->cout, s);
      if (a->ksize == 2 && a->stride == 1) return wide ? launch_conv<T, 2, 1, 2, 4, 2, true>(k, a->cout, s) : launch_conv<T, 2, 1, 2, 8, 1, true>(k, a->cout, s);
      // 4x4 stride 2, 64-channel tiles: weights staged in two halves of two kernel rows -> 73 KiB of LDS, two workgroups per CU
      if (a->ksize == 4 && a->stride == 2) return wide ? launch_conv<T, 4, 2, 1, 4, 2, true, 2>(k, a->cout, s) : launch_conv<T, 4, 2, 1, 4, 1, true>(k, a->cout, s);
      if (a->ksize == 2 && a->stride == 2) return wide ? launch_conv<T, 2, 2, 1, 4, 2, true>(k, a->cout, s) : launch_conv<T, 2, 2, 1, 4, 1, true>(k, a->cout, s);
      if (a->ksize == 1 && a->stride == 1) return launch_conv<T, 1, 1, 2, 4, 1, true>(k, a->cout, s);
      return set_err(SRGANFD_EINVAL, "conv2d: unsupported ksize=%d stride=%d", a->ksize, a->stride);
    }
  }
  if constexpr (bf) {
    return set_err(SRGANFD_EINVAL, "conv2d: the 16-bit kernels run on v_mfma_f32_16x16x32 (ksize %d stride %d has no kernel)", a->ksize, a->stride);
  } else {
    // f32 (parity mode): v_mfma_f32_32x32x2_f32, an exact fp32 fma chain
    if (a->ksize == 3 && a->stride == 1) return wide ? launch_conv<T, 3, 1, 2, 4, 2>(k, a->cout, s) : launch_conv<T, 3, 1, 2, 8, 1>(k, a->cout, s);
    if (a->ksize == 2 && a->stride == 1) return wide ? launch_conv<T, 2, 1, 2, 4, 2>(k, a->cout, s) : launch_conv<T, 2, 1, 2, 8, 1>(k, a->cout, s);
    // the stride-2 patch is 4x larger per output row: f32 fits a 4-row x 32-channel tile
    if (a->ksize == 4 && a->stride == 2) return launch_conv<T, 4, 2, 1, 4, 1>(k, a->cout, s);
    if (a->ksize == 3 && a->stride == 2) return wide ? launch_conv<T, 3, 2, 1, 4, 2>(k, a->cout, s) : launch_conv<T, 3, 2, 1, 4, 1>(k, a->cout, s);
    if (a->ksize == 2 && a->stride == 2) return wide ? launch_conv<T, 2, 2, 1, 4, 2>(k, a->cout, s) : launch_conv<T, 2, 2, 1, 4, 1>(k, a->cout, s);
    if (a->ksize == 1 && a->stride == 1) return launch_conv<T, 1, 1, 2, 4, 1>(k, a->cout, s);
    return set_err(SRGANFD_EINVAL, "conv2d: unsupported ksize=%d stride=%d", a->ksize, a->stride);
  }
}

int conv_fill_k(const srganfd_conv_args* a, ConvK& k) {
  if (!a || !a->x.ptr || !a->y.ptr || !a->w_packed) return set_err(SRGANFD_EINVAL, "conv2d: null pointer");
  if (a->cin <= 0 || a->cin % 32 || a->cout <= 0 || a->cout % 32 || a->cout_store <= 0 || a->cout_store > a->cout)
    return set_err(SRGANFD_EINVAL, "conv2d: cin=%d cout=%d cout_store=%d (need multiples of 32)", a->cin, a->cout, a->cout_store);
  if (a->n <= 0 || a->h_in <= 0 || a->w_in <= 0 || a->h_out <= 0 || a->w_out <= 0)
    return set_err(SRGANFD_EINVAL, "conv2d: bad dims");
  const int hl = a->h_in << (a->up ? 1 : 0), wl = a->w_in << (a->up ? 1 : 0);
  const bool sub = a->out_sy > 1 || a->out_sx > 1;  // parity-class launch of a stride-2 transposed conv
  const bool allcls = a->out_classes == 4;          // ... all four classes in this one launch
  if (a->out_classes != 0 && a->out_classes != 1 && !allcls) return set_err(SRGANFD_EINVAL, "conv2d: out_classes is 0, 1 or 4");
  if (allcls && (a->out_sy != 2 || a->out_sx != 2 || (a->ksize != 2 && a->ksize != 1) || a->stride != 1 || a->dtype == SRGANFD_F32 || a->up ||
                 (a->class_pad_step != 0 && a->class_pad_step != 1)))
    return set_err(SRGANFD_EINVAL, "conv2d: out_classes = 4 is the 16-bit data gradient of a stride-2 conv (ksize 2 or 1, stride 1, out_sy = out_sx = 2, class_pad_step 0 / 1)");
  if (!sub) {
    const int ho = (hl + 2 * a->pad - a->ksize) / a->stride + 1, wo = (wl + 2 * a->pad - a->ksize) / a->stride + 1;
    if (ho != a->h_out || wo != a->w_out)
      return set_err(SRGANFD_EINVAL, "conv2d: h_out/w_out %dx%d inconsistent with input (expect %dx%d)", a->h_out, a->w_out, ho, wo);
  } else if (a->out_h_full < (a->h_out - 1) * a->out_sy + (allcls ? 1 : a->out_oy) + 1 || a->out_w_full < (a->w_out - 1) * a->out_sx + (allcls ? 1 : a->out_ox) + 1) {
    return set_err(SRGANFD_EINVAL, "conv2d: strided output does not fit the full image");
  }
  const int align = a->dtype == SRGANFD_F32 ? 4 : 8;
  if (a->x.cstride % align || a->x.c0 % align) return set_err(SRGANFD_EINVAL, "conv2d: x view not 16-byte aligned");
  if (a->x.c0 + a->cin > a->x.cstride) return set_err(SRGANFD_EINVAL, "conv2d: x view exceeds buffer channels");
  if (a->y.c0 + a->cout_store > a->y.cstride) return set_err(SRGANFD_EINVAL, "conv2d: y view exceeds buffer channels");
  if ((size_t)a->h_in * a->w_in * (size_t)a->x.cstride >= 0x7fffffffULL)
    return set_err(SRGANFD_EINVAL, "conv2d: one input image is too large for 32-bit element offsets");
  // 16-bit kernels stage through a buffer descriptor per image; padding lanes carry the offset 0x7fffffff, which must lie beyond it
  if (a->dtype != SRGANFD_F32 && (size_t)a->h_in * a->w_in * (size_t)a->x.cstride * 2 >= 0x7fffffffULL)
    return set_err(SRGANFD_EINVAL, "conv2d: one input image exceeds the 2 GiB a buffer descriptor's range check can separate from padding");
  const size_t opix = sub ? (size_t)a->out_h_full * a->out_w_full : (size_t)a->h_out * a->w_out;   // per image
  auto fits = [&](const srganfd_view& v) { return !v.ptr || opix * (size_t)v.cstride < 0x7fffffffULL; };
  if (!fits(a->y) || !fits(a->y2) || !fits(a->r1) || !fits(a->r2) || !fits(a->mask))
    return set_err(SRGANFD_EINVAL, "conv2d: one output-side image is too large for 32-bit element offsets");
  k.x = a->x.ptr; k.y = a->y.ptr; k.y2 = a->y2.ptr; k.y2C = a->y2.cstride; k.y2_c0 = a->y2.c0;
  k.r1 = a->r1.ptr; k.r2 = a->r2.ptr; k.mask = a->mask.ptr; k.w = a->w_packed;
  k.bias = a->bias; k.alpha_dev = a->alpha_dev;
  k.xC = a->x.cstride; k.x_c0 = a->x.c0; k.yC = a->y.cstride; k.y_c0 = a->y.c0;
  k.r1C = a->r1.cstride; k.r1_c0 = a->r1.c0; k.r2C = a->r2.cstride; k.r2_c0 = a->r2.c0;
  k.mC = a->mask.cstride; k.m_c0 = a->mask.c0;
  {
    // addressing of every operand: NHWC or planar 32-channel groups (see ConvK)
    const long long ipix = (long long)a->h_in * a->w_in;
    for (const srganfd_view* v : {&a->x, &a->y, &a->y2, &a->r1, &a->r2, &a->mask})
      if (v->ptr && v->planar && (v->c0 % 32 || v->cstride % 32)) return set_err(SRGANFD_EINVAL, "conv2d: a planar view needs c0 and cstride multiples of 32");
    if (a->y.planar && a->y_f32) return set_err(SRGANFD_EINVAL, "conv2d: fp32 output views are NHWC only");
    k.x_ps = a->x.planar ? 32 : a->x.cstride;
    k.x_cs = a->x.planar ? (int)(ipix * 32) : 32;
    k.x_base = a->x.planar ? (a->x.c0 / 32) * k.x_cs : a->x.c0;
    auto out_strides = [&](const srganfd_view& v, int& ps, int& gs) { ps = v.planar ? 32 : v.cstride; gs = v.planar ? (int)(opix * 32) : 32; };
    out_strides(a->y, k.y_ps, k.y_gs); out_strides(a->y2, k.y2_ps, k.y2_gs); out_strides(a->r1, k.r1_ps, k.r1_gs);
    out_strides(a->r2, k.r2_ps, k.r2_gs); out_strides(a->mask, k.m_ps, k.m_gs);
  }
  k.N = a->n; k.Hin = a->h_in; k.Win = a->w_in; k.up = a->up ? 1 : 0; k.pad_y = sub ? a->pad_y : a->pad; k.pad_x = sub ? a->pad_x : a->pad;
  k.osy = sub ? a->out_sy : 1; k.osx = sub ? a->out_sx : 1; k.ooy = sub ? a->out_oy : 0; k.oox = sub ? a->out_ox : 0;
  k.HoutF = sub ? a->out_h_full : a->h_out; k.WoutF = sub ? a->out_w_full : a->w_out;
  k.Hout = a->h_out; k.Wout = a->w_out; k.nChunks = a->cin / 32; k.nNb = 0; k.cls_sh = allcls ? 0 : -1; k.cls_pad = allcls ? a->class_pad_step : 0; k.cout_store = a->cout_store;
  k.tiles_x = k.tiles_y = 0;
  k.alpha = a->alpha; k.slope = a->slope; k.post_scale = a->post_scale; k.r1s = a->r1_scale; k.r2s = a->r2_scale;
  k.mask_slope = a->mask_slope; k.act = a->act; k.y_f32 = a->y_f32 ? 1 : 0;
  auto aligned = [&](const srganfd_view& v) { return !v.ptr || (v.cstride % align == 0 && v.c0 % align == 0 && ((uintptr_t)v.ptr & 15) == 0); };
  k.fast_epi = (!a->y_f32 && a->cout_store == a->cout && aligned(a->y) && aligned(a->y2) && aligned(a->r1) && aligned(a->r2) && aligned(a->mask)) ? 1 : 0;
  return SRGANFD_OK;
}

int conv2d_impl(const srganfd_conv_args* a, hipStream_t stream) {
  ConvK k;
  {
    const int rc = conv_fill_k(a, k);
    if (rc != SRGANFD_OK) return rc;
  }
  if (a->dtype == SRGANFD_BF16) return dispatch_conv<bf16_t>(a, k, stream);
  if (a->dtype == SRGANFD_F16) return dispatch_conv<f16_t>(a, k, stream);
  if (a->dtype == SRGANFD_F32) return dispatch_conv<float>(a, k, stream);
  return set_err(SRGANFD_EINVAL, "conv2d: bad dtype %d", a->dtype);
}

}  // namespace srganfd
