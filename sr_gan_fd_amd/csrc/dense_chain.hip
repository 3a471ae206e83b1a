// dense_chain.hip -- a whole dense block as ONE launch whose activations never leave LDS.
//
// Replaces the five srganfd_conv2d launches of _ResidualDenseBlock.forward (BSRGAN/model.py:51-62) -- and, with the data-gradient
// operands, the five launches of its backward pass (the same dense structure over the stacked output gradients, engine.py) -- by one
// persistent launch, for every batch whose images are at most one 16 x 16-pixel tile per compute unit (the reference's crop sizes:
// esrgan_config.py:73-74 32 x 32, rrdbnet_config.py:51-52 48 x 48, aesrgan_config.py:102-103 60 x 60 at batch 8-16 are ONE pass;
// bsrgan_config.py:101-102 72 x 72 at batch 16 is two, BASELINE's 128 x 128 at batch 32 eight -- correct there, not faster).
//
// Why: a per-layer conv launch pays, per tile and layer, the kernel boundary, the address set-up, the round trip of the tile's own data
// through L2 / HBM and the LDS commit of every chunk (13-15 us per launch whatever it computes at the reference's sizes,
// profiles/r05_reference_shapes_before.txt; 14 us + 7.3 us per 32 input channels at batch 32, 128 x 128); neither removing the boundaries
// (layer-persistent launch) nor the commits (LDS-DMA streaming conv) changed that (profiles/r05_chain_at_small_shapes.txt,
// r05_stream_conv_at_small_shapes.txt).  Only a kernel that keeps the tile ACROSS layers removes them.
//
// MI355X mapping (this is what 160 KB of LDS per CU and 256 CUs are for -- 40 MB of on-chip activation store):
//   * one workgroup per CU owns ONE tile of 16 x 16 pixels of one image through all layers, then walks on to its tile of the next group
//     of images ("pass"; the whole batch is one launch).  The tile's 192 channels with a 1-pixel halo (18 x 18 pixels x 6 groups of 32
//     channels x 2 B = 124,416 B) are LDS-resident: groups 0-1 (the block input) are loaded once per pass, group 2 + k is written by
//     layer k's epilogue straight from the accumulators.
//   * four COMPUTE waves (one per SIMD): wave w computes rows 4w .. 4w+3 x 16 pixels x 32 output channels -- per kernel-column step 6
//     pixel fragments (patch rows 4w .. 4w+5, reused by the three kernel rows) and 6 weight fragments for 24 v_mfma_f32_16x16x32
//     (operands swapped: A = weights, B = pixels, so a lane ends up with 4 consecutive channels of one pixel).  48 KB of LDS reads per
//     step and CU = 192 LDS cycles against 384 MFMA cycles per SIMD; the fragments of step t+1 are read behind step t's first 16 MFMAs
//     (sched_group_barrier pins that order), two fragment register sets, the step written out twice.
//   * two LOADER waves own the weight stream: per step three 1 KB LDS-DMA pieces each (pack.hip's fragment order: a piece is one
//     ds_read_b128 per lane) into a ring of five 6 KB slots, four steps ahead, counted vmcnt; ONE barrier per step publishes slot t+1 and
//     frees slot t.  No compute wave ever issues or waits for a weight piece.
//   * everything that is not "barrier, 12 fragment reads, 24 MFMAs" is out of the step's straight line: per-layer scalars come from an
//     LDS table copied once from the kernel arguments, the rare steps sit behind ONE unlikely test, the epilogue's operand requests are
//     straight-line code behind steps 1 and 3 (one definition per register: nothing for the compiler to settle with a wait).
//   * epilogue by compile-time kind (operands none / mask / r1 / r1 + r2, growth or closing layer, max-form activation); a lane's two
//     channel quads become ONE 16-byte slot through v_permlane16_swap: one 16-byte LDS write and one 16-byte store per pixel.
//   * only the halo of a NEW group crosses workgroups: the epilogue also stores the tile to the block's HBM buffer (the weight gradient and
//     the next launch need it there anyway) with write-through (sc1) stores and goes on; behind step 1 of the next layer every compute
//     wave drains its stores, behind barrier 2 ONE lane publishes one flag per (layer, tile) -- row 1 of the micro-architecture guide's
//     hand-off table.  A consumer fetches its <= 8 neighbours' flags and then its share of the 68 halo pixels by LDS-DMA (sc1) into a
//     staging area -- nothing of the hand-off lives in registers between steps -- on a per-layer schedule (s_poll / s_chk / s_wr) that
//     puts the three dependent round trips (~1.5-2.5k cycles each) beside MFMA steps: hidden from conv4 on, a 4-5k-cycle wait in conv2.
//   * flags are epoch-valued: the launch reads its epoch from a device counter that the last workgroup to finish advances, so nothing is
//     zeroed per launch and a captured hipGraph replays correctly.
// Measured (profiles/r05_dense_chain_v2_*): 36.5-38.2 us (forward) / 41.1-43.4 us (data gradient) per pass of up to 256 tiles against
// 55.5-58.3 us for the five launches; a pass costs the same whatever it holds, so the engines take the launch for one-pass batches only.
// Same arithmetic contract as conv_igemm.hip (include/srganfd.h, srganfd_conv2d): fp32 accumulation in chunk, kernel-column,
// kernel-row order, v = post_scale * act(alpha * acc + bias) + r1s * r1 + r2s * r2, masked, rounded once to the 16-bit type.
#include "conv_common.hpp"
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

namespace srganfd {

int conv_fill_k(const srganfd_conv_args* a, ConvK& k);   // conv_igemm.hip: validation + operand strides of one conv launch

namespace {
constexpr int kDcMaxLayers = 6;      // four growth convs + the two 32-channel halves of the closing conv
constexpr int kDcTW = 16;            // tile width (one MFMA row of pixels)
constexpr int kDcPC = 18;            // patch columns
constexpr int kDcGroups = 6;
constexpr int kDcSlot = 6144;        // one kernel-column step of weights: 3 kernel rows x 2 channel halves x 1 KB fragments
// Tile geometry by rows per compute wave.  RPW = 4: 16 x 16-pixel tiles (256 pixels per CU and pass: the densest form the LDS holds).
// RPW = 2 (3): 8 x 16- (12 x 16-) pixel tiles for batches that fit one pass that way too: more workgroups, fewer MFMAs, a shorter epilogue and a smaller patch per
// workgroup -- a single-wave-per-SIMD step is bound by its instruction count and the barrier, so the shorter step is not half as long,
// but the pass is shorter: 29.6 against 37.1 us (batch 16, 32 x 32).  (4 x 16 tiles, RPW = 1, were built and measured: 29.0 us alone,
// but SLOWER inside the training step -- every workgroup streams the block's whole 0.96 MB of weights, and four times the workgroups
// on cold weights is four times that stream: profiles/r05_dense_chain_tile_rows_ab.txt.)
template <int RPW> struct DcGeo {
  static constexpr int TR = 4 * RPW;                      // tile rows
  static constexpr int PR = TR + 2;                       // patch rows
  static constexpr int NPIX = PR * kDcPC;
  static constexpr int GroupBytes = NPIX * 64;            // one 32-channel group of the resident patch: 20,736 / 11,520 B
  static constexpr int HaloPix = 2 * kDcPC + 2 * TR;      // 68 / 52
  static constexpr int HaloItems = HaloPix * 4;           // 16-byte items of one group's halo ring
  static constexpr int WaveItems = HaloItems / 4;         // ... of one compute wave: 68 / 52
  static constexpr int Slots = RPW == 4 ? 5 : 8;          // weight ring: the step being read + 4 / 7 in flight (16-row tiles: what the LDS left over holds)
  static constexpr int RingOff = kDcGroups * GroupBytes;  // 124,416 / 69,120
  static constexpr int StageOff = RingOff + Slots * kDcSlot;        // hand-off staging of the four compute waves (LDS-DMA destinations)
  static constexpr int StageHalo = (WaveItems > 64 ? WaveItems : 64) * 16;       // per wave: its halo items (a full 64-lane piece first), ...
  static constexpr int StageWave = StageHalo + 256;                 // ... then one flag word per lane
  static constexpr int CtlOff = StageOff + 4 * StageWave;           // float alpha[8]; float bias[6][32]; int tab[6][32]
  static constexpr int Lds = CtlOff + 32 + kDcMaxLayers * 32 * 4 + kDcMaxLayers * 128;      // 162,080 / 124,000 B
};
constexpr int kDcThreads = 384;      // waves 0-3 compute, 4-5 weight loaders
constexpr int kDcMaxTiles = 16384;   // tiles of one call (flags: 4 growth layers x tiles)

struct DcLayer {
  const char* w;             // this 32-channel n-tile of the packed operand: [chunk][tap][channel half][64 lanes x 16 B]
  const float* bias;         // its 32 bias values, or null
  const float* alpha_dev;
  char* y; const char* r1; const char* r2; const char* mask;       // image-0 bases
  int yC, y_c0, y_ps, y_gs, r1C, r1_c0, r1_ps, r1_gs, r2C, r2_c0, r2_ps, r2_gs, mC, m_c0, m_ps, m_gs;    // elements (see ConvK)
  int nChunks;
  int dst_group;             // >= 0: growth layer -- the output is also group dst_group of the resident patch and its halo is exchanged
  float alpha, neg, post_scale, r1s, r2s, mask_slope;     // neg: factor of the negative side of the activation (slope / 0 / 1)
};
struct DcK {
  DcLayer L[kDcMaxLayers];
  const char* x;             // image-0 base of the block input (groups 0, 1)
  int xC, x_ps, x_base, x_cs;
  int nLayers, stepsPerPass;
  int N, H, W, tiles_x, tiles_y, tpi;       // tpi: tiles per image
  int rpw;                   // rows per compute wave: 4 (16 x 16 tiles) or 2 (8 x 16 tiles), see DcGeo
  int ipl;                   // images per pass (the grid is ipl * tpi workgroups)
  int totalTiles;            // N * tpi
  int tab[kDcMaxLayers][32]; // per layer, copied to LDS once (kernel-argument loads in the layer loop measured ~0.7k cycles per dependent round):
                             // 0 nChunks, 1 dst_group, 2 epilogue kind, 3 post_scale (+), 4 post_scale * neg (-), 5 r1s, 6 r2s, 7 mask_slope (floats as bits),
                             // 8-9 y, 10 yC, 11 y_ps, 12 y_gs, 13 y_c0, 14-15 operand A (r1 or mask), 16 its C, 17 ps, 18 gs, 19 c0, 20-21 r2, 22 r2C, 23 r2_ps, 24 r2_gs, 25 r2_c0
  int* hdr;                  // [0] hand-off waits that gave up, [1] epoch of the last finished launch, [2] workgroups finished in this launch
  int* flags;                // [growth layer][tile of the batch], epoch-valued
};

__device__ __forceinline__ unsigned dc_lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
// One LDS-DMA piece (see wgrad.hip glds16): 64 lanes x 16 bytes, per-lane source, wave-uniform LDS destination; outside the
// compiler's wait-count bookkeeping, the loader waves count vmcnt themselves.
__device__ __forceinline__ void dc_glds16(const void* gsrc, unsigned lds_dst) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(dst) : "m0", "memory");
#pragma clang diagnostic pop
}
// the hand-off's loads: the same LDS-DMA with the sc1 policy (served by memory, never a stale L1 / L2 line); 16 bytes or one dword per lane
__device__ __forceinline__ void dc_glds16_sc1(const void* gsrc, unsigned lds_dst) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" : : "v"(gsrc), "s"(dst) : "m0", "memory");
#pragma clang diagnostic pop
}
__device__ __forceinline__ void dc_glds4_sc1(const void* gsrc, unsigned lds_dst) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off sc1" : : "v"(gsrc), "s"(dst) : "m0", "memory");
#pragma clang diagnostic pop
}
// 16-byte write-through store (no 16-byte atomic store exists; inline assembly, so the compiler's vmcnt bookkeeping does not see it:
// the hand-off drains with an explicit vmcnt(0) before the flag is published, and a kernel's stores are complete when it ends)
__device__ __forceinline__ void dc_store16_sc1(void* dst, const u32x4 v) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  // the s_nop is the gfx9 wait state between a store of more than 8 bytes and a VALU write of its data registers (the hazard
  // recognizer inserts it for its own stores; it cannot see into this one, and the epilogue reuses the registers at once)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
#pragma clang diagnostic pop
}
template <int N> __device__ __forceinline__ void dc_wait_vm() {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#pragma clang diagnostic pop
}
// at most 3 * younger_steps of this loader's pieces may still be in flight (the ring holds the step being read and SLOTS - 1 more:
// when step t + 1 must have landed, steps t + 2 .. t + SLOTS - 1 are younger)
template <int SLOTS> __device__ __forceinline__ void dc_wait_pieces(int younger_steps) {
  constexpr int kMax = SLOTS - 2;
  if (younger_steps >= kMax) dc_wait_vm<3 * kMax>();
  else if (younger_steps == 5) dc_wait_vm<15>(); else if (younger_steps == 4) dc_wait_vm<12>(); else if (younger_steps == 3) dc_wait_vm<9>();
  else if (younger_steps == 2) dc_wait_vm<6>(); else if (younger_steps == 1) dc_wait_vm<3>(); else dc_wait_vm<0>();
}
// workgroup barrier that leaves vector-memory operations in flight (LDS operations of this wave are complete when it is passed)
__device__ __forceinline__ void dc_barrier() {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma clang diagnostic pop
}
// byte position of (patch row, patch column, 16-byte slot) inside one group: pixel-major, slot XOR 2 * ((column >> 2) & 1) -- the
// column-keyed swizzle of conv_igemm.hip's 16x16x32 fragment reads (conflict-free ds_read_b128 for every kernel column)
__device__ __forceinline__ int dc_pos(int prow, int pcol, int slot) { return (prow * kDcPC + pcol) * 64 + ((slot ^ (((pcol >> 2) & 1) << 1)) << 4); }
// halo ring pixel hp of the (TR + 2) x 18 patch: top row (18), bottom row (18), left column (TR), right column (TR)
template <int TR> __device__ __forceinline__ void dc_halo_rc(int hp, int& prow, int& pcol) {
  prow = hp < 18 ? 0 : (hp < 36 ? TR + 1 : (hp < 36 + TR ? 1 + hp - 36 : 1 + hp - 36 - TR));
  pcol = hp < 18 ? hp : (hp < 36 ? hp - 18 : (hp < 36 + TR ? 0 : 17));
}

typedef unsigned long long dc_u64;
typedef __attribute__((ext_vector_type(2))) unsigned int dc_u32x2;

template <typename T> __device__ __forceinline__ void dc_widen4(const dc_u32x2 q, float* f) {
  if constexpr (Elem<T>::kDtype == SRGANFD_F16) {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    const h4 hv = __builtin_bit_cast(h4, q);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = (float)hv[i];
  } else {
    f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xffff0000u);
    f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xffff0000u);
  }
}
template <typename T> __device__ __forceinline__ dc_u32x2 dc_narrow4(const float* v4) {
  if constexpr (Elem<T>::kDtype == SRGANFD_F16) {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    const h4 hv = {(_Float16)v4[0], (_Float16)v4[1], (_Float16)v4[2], (_Float16)v4[3]};
    return __builtin_bit_cast(dc_u32x2, hv);
  } else {
    return dc_u32x2{(unsigned)f2bf(v4[0]) | ((unsigned)f2bf(v4[1]) << 16), (unsigned)f2bf(v4[2]) | ((unsigned)f2bf(v4[3]) << 16)};
  }
}
}  // namespace

template <typename T, int RPW>
__global__ __launch_bounds__(kDcThreads) void dense_chain_kernel(const DcK a) {
  using Frag = typename FragAB<T>::type;
  using Geo = DcGeo<RPW>;
  constexpr int kDcGroupBytes = Geo::GroupBytes, kDcSlots = Geo::Slots, kDcStageHalo = Geo::StageHalo;
  typedef __attribute__((address_space(1))) T GT;      // an element in global memory
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;
  char* const ring = smem + Geo::RingOff;
  float* const ctl_alpha = (float*)(smem + Geo::CtlOff);
  float* const ctl_bias = (float*)(smem + Geo::CtlOff + 32);
  int* const ctl_tab = (int*)(smem + Geo::CtlOff + 32 + kDcMaxLayers * 32 * 4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;

  const int li = blockIdx.x / a.tpi, tin = blockIdx.x - li * a.tpi;       // image of the pass, tile of the image
  const int ty = tin / a.tiles_x, tx = tin - ty * a.tiles_x;
  const int oy0 = ty * Geo::TR, ox0 = tx * kDcTW;
  const size_t ipix = (size_t)a.H * a.W;
  const int npass = li < a.N ? (a.N - li + a.ipl - 1) / a.ipl : 0;
  const int totalSteps = npass * a.stepsPerPass;

  // resident patch of a pass: groups 0, 1 from the block input (zeros outside the image), zero halo ring for the groups to come.  Nobody
  // reads the patch any more when this runs: the last step of the previous pass has had its fragments in registers since its barrier.
  auto load_patch = [&](size_t img) {
    const T* xi = (const T*)a.x + img * a.xC + a.x_base;
    constexpr int kItems = Geo::NPIX * 8, kRounds = (kItems + kDcThreads - 1) / kDcThreads;
    u32x4 v[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int item = tid + r * kDcThreads;
      const int pix = item >> 3, g = (item >> 2) & 1, slot = item & 3;
      const int prow = pix / kDcPC, pcol = pix - prow * kDcPC;
      const int gy = oy0 - 1 + prow, gx = ox0 - 1 + pcol;
      v[r] = u32x4{0u, 0u, 0u, 0u};
      if (item < kItems && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v[r] = *(const u32x4*)(xi + ((size_t)(gy * a.W + gx) * a.x_ps + (size_t)g * a.x_cs + slot * 8));
    }
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int item = tid + r * kDcThreads;
      const int pix = item >> 3, g = (item >> 2) & 1, slot = item & 3;
      const int prow = pix / kDcPC, pcol = pix - prow * kDcPC;
      if (item < kItems) *(u32x4*)(patch + g * kDcGroupBytes + dc_pos(prow, pcol, slot)) = v[r];
    }
    for (int item = tid; item < Geo::HaloItems * 4; item += kDcThreads) {
      const int gg = item / Geo::HaloItems, hi = item - gg * Geo::HaloItems;
      int prow, pcol; dc_halo_rc<Geo::TR>(hi >> 2, prow, pcol);
      *(u32x4*)(patch + (2 + gg) * kDcGroupBytes + dc_pos(prow, pcol, hi & 3)) = u32x4{0u, 0u, 0u, 0u};
    }
  };

  if (wave >= 4) {
    // =====================================================  LOADER WAVES  =====================================================
    // the weight stream: per kernel-column step this wave's three 1 KB pieces of the 6 KB slot, five steps ahead of the step that reads them
    const int ld = wave - 4;               // pieces {0, 1, 2} / {3, 4, 5}: piece q = (kernel row q >> 1, channel half q & 1)
    int pl = 0, pc = 0, pk = 0, pt = 0;    // producer cursor = the next step to request
    auto issue_next = [&]() {
      if (pt < totalSteps) {
        const char* wl = a.L[pl].w + (size_t)((pc * 9 + pk) * 2048) + lane * 16;
        const unsigned dst = dc_lds_addr(ring) + (unsigned)((pt % kDcSlots) * kDcSlot);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int q = 3 * ld + j;
          dc_glds16(wl + ((q >> 1) * 6144 + (q & 1) * 1024), dst + q * 1024);
        }
        ++pt;
        if (++pk == 3) { pk = 0; if (++pc == a.L[pl].nChunks) { pc = 0; if (++pl == a.nLayers) pl = 0; } }
      }
    };
#pragma unroll 1
    for (int i = 0; i < kDcSlots; ++i) issue_next();        // steps 0 .. SLOTS - 1
    int t = 0;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
      load_patch((size_t)(pass * a.ipl + li) * ipix);
      if (pass == 0) dc_wait_pieces<kDcSlots>(kDcSlots - 2);       // slots 0 and 1 landed; later passes: the step barriers keep the ring two slots ahead
      dc_barrier();
#pragma unroll 1
      for (int i = 0; i < a.stepsPerPass; ++i) {
        dc_wait_pieces<kDcSlots>(totalSteps - 2 - t);   // this wave's pieces of step t + 1 have landed: only those of steps t + 2 .. t + 4 are younger
        dc_barrier();                         // slot t + 1 is complete for everybody; slot t is free (its fragments are in registers)
        issue_next();                         // step t + SLOTS into it
        ++t;
      }
    }
  } else {
    // =====================================================  COMPUTE WAVES  =====================================================
    const int cw = wave;                   // rows RPW cw .. RPW cw + RPW - 1 of the tile
    // launch constants: epoch (polls and the publish use it), bias / alpha table
    const int epoch = __hip_atomic_load(a.hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (cw == 3) {      // the per-layer table: one dword per lane and round straight from the kernel-argument segment
#pragma unroll
      for (int r = 0; r < kDcMaxLayers * 32 / 64; ++r)
        ctl_tab[lane + 64 * r] = ((const __attribute__((address_space(4))) int*)__builtin_amdgcn_kernarg_segment_ptr())[offsetof(DcK, tab) / 4 + lane + 64 * r];
    }
    {
      // all six layers' loads first, then the LDS writes: one memory round trip instead of one per layer
      float bv[kDcMaxLayers], av[kDcMaxLayers];
#pragma unroll
      for (int l = 0; l < kDcMaxLayers; ++l) {
        const bool mine = l < a.nLayers && cw == (l & 3);
        bv[l] = mine && lane < 32 && a.L[l].bias ? a.L[l].bias[lane] : 0.f;
        av[l] = mine && lane == 32 && a.L[l].alpha_dev ? *a.L[l].alpha_dev : 1.f;
      }
#pragma unroll
      for (int l = 0; l < kDcMaxLayers; ++l) {
        if (l < a.nLayers && cw == (l & 3)) {
          if (lane < 32) ctl_bias[l * 32 + lane] = bv[l];
          if (lane == 32) ctl_alpha[l] = a.L[l].alpha * av[l];
        }
      }
    }
    const int rowoff = (RPW * cw) * kDcPC * 64;
    // this wave's share of a group's halo ring: items 68 cw .. 68 cw + 67 (item = 4 * halo pixel + 16-byte slot), two rounds of lanes
    int h_off[2], h_gp[2], h_slot[2]; bool h_in[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int it = lane + 64 * r;
      const int item = Geo::WaveItems * cw + (it < Geo::WaveItems ? it : 0);
      int prow, pcol; dc_halo_rc<Geo::TR>(item >> 2, prow, pcol);
      const int gy = oy0 - 1 + prow, gx = ox0 - 1 + pcol;
      h_in[r] = it < Geo::WaveItems && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      h_off[r] = dc_pos(prow, pcol, item & 3);
      h_slot[r] = item & 3;
      h_gp[r] = h_in[r] ? gy * a.W + gx : 0;          // lanes without an item read pixel 0 of the image (a valid address) and drop it
    }
    char* const stage = smem + Geo::StageOff + cw * Geo::StageWave;        // this wave's hand-off staging: 68 halo items, then 64 flag words
    const unsigned stage_lds = dc_lds_addr(stage);
    // the eight neighbours (3 x 3 without the centre) of this tile, one per lane 0 .. 7
    const int nq = (lane & 7) < 4 ? (lane & 7) : (lane & 7) + 1;
    const int ndy = nq / 3 - 1, ndx = nq % 3 - 1;
    const bool nvalid = lane < 8 && ty + ndy >= 0 && ty + ndy < a.tiles_y && tx + ndx >= 0 && tx + ndx < a.tiles_x;
    const int noff = nvalid ? ndy * a.tiles_x + ndx : 0;

    auto load_frags = [&](Frag* fw, Frag* fp, int c, int kx, int tt) {
      // pixel l15 + kx of the patch row, channel slot g4 under the column-keyed swizzle (dc_pos), as arithmetic: no selects, no branches
      const int col = l15 + kx;
      const char* pa = patch + c * kDcGroupBytes + rowoff + (col << 6) + ((g4 ^ ((col >> 1) & 2)) << 4);
#pragma unroll
      for (int rr = 0; rr < RPW + 2; ++rr) fp[rr] = *(const Frag*)(pa + rr * (kDcPC * 64));
      const char* rs = ring + (tt % kDcSlots) * kDcSlot + lane * 16;
#pragma unroll
      for (int q = 0; q < 6; ++q) fw[q] = *(const Frag*)(rs + q * 1024);
    };

    __builtin_amdgcn_s_setprio(2);
    int t = 0;        // kernel-column steps consumed so far (all passes): ring slot t % 6
    int giveups = 0;
    const int orow0 = oy0 + RPW * cw, ocol = ox0 + l15;
    const bool col_ok = ocol < a.W;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
      const int n = pass * a.ipl + li;
      const size_t img = (size_t)n * ipix;
      const int gtile = n * a.tpi + tin;
      load_patch(img);
      dc_barrier();
      Frag Aw[6], Ap[RPW + 2], Bw[6], Bp[RPW + 2];       // two fragment sets: a step computes on one while the next step's are read into the other
      load_frags(Aw, Ap, 0, 0, t);

#pragma unroll 1
      for (int l = 0; l < a.nLayers; ++l) {
        const int* const tl = ctl_tab + 32 * l;
        const int4 tb0 = *(const int4*)(tl), tb1 = *(const int4*)(tl + 4);
        const int nCh = __builtin_amdgcn_readfirstlane(tb0.x), ns = 3 * nCh;
        const int dst_group = __builtin_amdgcn_readfirstlane(tb0.y), epi_kind = __builtin_amdgcn_readfirstlane(tb0.z);
        const float ps_pos = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(tb0.w)), ps_neg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(tb1.x));
        const float r1s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(tb1.y)), r2s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(tb1.z));
        const float mslope = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(tb1.w));
        const bool prev_growth = l > 0 && __builtin_amdgcn_readfirstlane(ctl_tab[32 * (l > 0 ? l - 1 : 0) + 1]) >= 0;
        // hand-off of the previous layer's output (group nCh - 1, first read by the fragments requested behind barrier 3 (nCh - 1) - 1): three
        // dependent round trips of 1.5-2.5k cycles (store acknowledgement, flag visible + poll, halo read) beside steps of ~650 cycles.
        // Stores drained behind step 1, flag published behind barrier 2, neighbours' flags requested behind barrier s_poll (a poll right
        // after the publish always reads the old value: everybody publishes at the same time), looked at and halo requested behind barrier
        // s_chk, halo moved from the staging area into the patch behind barrier s_wr (visible behind barrier s_wr + 1 <= 3 (nCh - 1) - 1).
        // conv2 (3 chunks) has no room for that: it waits at step 4; from conv4 on everything is hidden.
        // (8-row tiles: the steps are shorter, the round trips are not: everything one step later where the layer has room)
        const int s_poll = nCh <= 3 ? 3 : (RPW == 4 ? (nCh == 4 ? 3 : nCh - 1) : nCh), s_chk = nCh == 3 ? 4 : (RPW == 4 ? 2 * nCh - 3 : 2 * nCh - 2), s_wr = nCh == 3 ? 4 : 3 * nCh - 5;
        const bool growth = dst_group >= 0;
        f32x4_t acc[RPW][2];     // [row][16-channel half]
#pragma unroll
        for (int m = 0; m < RPW; ++m)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) acc[m][nh] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        // the rare steps: everything that is not "barrier, 12 fragment reads, 24 MFMAs" lives here, out of the step's straight line.
        // The hand-off's loads are LDS-DMA into this wave's staging area: nothing of them lives in registers between steps.
        auto event = [&](int s) {
          if (s == 2) {
            // every compute wave's stores of layer l - 1 are acknowledged (vmcnt(0) behind barrier 1, then barrier 2): ONE lane publishes
            // the tile.  Behind barrier s_poll every wave asks for its neighbours' flags (lane j < 8: neighbour j; the other lanes read this tile's own word)
            if (cw == 0 && lane == 0) __hip_atomic_store(a.flags + (size_t)(l - 1) * a.totalTiles + gtile, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else if (s == s_poll) {
            dc_glds4_sc1(a.flags + (size_t)(l - 1) * a.totalTiles + gtile + noff, stage_lds + kDcStageHalo);
          } else if (s == s_chk) {
            const int4 q2 = *(const int4*)(tl - 32 + 8), q3 = *(const int4*)(tl - 32 + 12);      // the previous layer's output view
            const int PyC = __builtin_amdgcn_readfirstlane(q2.z), Py_ps = __builtin_amdgcn_readfirstlane(q2.w), Py_gs = __builtin_amdgcn_readfirstlane(q3.x), Py_c0 = __builtin_amdgcn_readfirstlane(q3.y);
            const T* const Py = (const T*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(q2.y) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(q2.x));
            // all neighbours have published?  (normally yes at the first look: they run the same schedule)
            for (int spins = 0;; ++spins) {
              dc_wait_vm<0>();
              const int pv = *(const volatile __attribute__((address_space(3))) int*)(stage_lds + kDcStageHalo + 4 * lane);
              if (__builtin_amdgcn_ballot_w64(nvalid && pv != epoch) == 0ull) break;
              if (spins > (1 << 21)) { ++giveups; break; }      // seconds: never in a correct run; wrong results, no hang
              __builtin_amdgcn_s_sleep(2);
              dc_glds4_sc1(a.flags + (size_t)(l - 1) * a.totalTiles + gtile + noff, stage_lds + kDcStageHalo);
            }
            // this wave's 68 halo items of the previous layer's output: 16 bytes per lane, lanes 0 .. 63 and 0 .. 3
            const T* yi = Py + img * PyC;
            {
              const int cc = Py_c0 + 8 * h_slot[0];
              dc_glds16_sc1(yi + ((size_t)h_gp[0] * Py_ps + (size_t)(cc >> 5) * Py_gs + (cc & 31)), stage_lds);
            }
            if (Geo::WaveItems > 64 && lane < Geo::WaveItems - 64) {
              const int cc = Py_c0 + 8 * h_slot[1];
              dc_glds16_sc1(yi + ((size_t)h_gp[1] * Py_ps + (size_t)(cc >> 5) * Py_gs + (cc & 31)), stage_lds + 1024);
            }
          }
          if (prev_growth && s == s_wr) {
            dc_wait_vm<0>();       // the halo items have landed in the staging area
            char* pg = patch + __builtin_amdgcn_readfirstlane(tl[1 - 32]) * kDcGroupBytes;
            if (h_in[0]) *(u32x4*)(pg + h_off[0]) = *(const u32x4*)(stage + 16 * lane);
            if (Geo::WaveItems > 64 && h_in[1]) *(u32x4*)(pg + h_off[1]) = *(const u32x4*)(stage + 1024 + 16 * lane);
          }
        };

        int c = 0, kx = 0;       // chunk and kernel column of the current step
        int ev = prev_growth ? 2 : -1;       // the next step with an event
        // one kernel-column step: barrier (slot t + 1 is complete, everybody is past step t - 1), the next step's fragments into N* (the
        // next layer's first step behind this layer's last: its chunk 0 is the block input), this step's 24 MFMAs on C*
#define DC_STEP(S, CW, CP, NW, NP)                                                                                  \
        {                                                                                                           \
          dc_barrier();                                                                                             \
          if (__builtin_expect((S) == ev, 0)) {                                                                     \
            event(S);                                                                                               \
            ev = (S) == 2 ? s_poll : ((S) == s_poll ? s_chk : ((S) == s_chk && s_wr > s_chk ? s_wr : -1));          \
          }                                                                                                         \
          const bool wrap_ = (S) + 1 == ns;                                                                         \
          const int kxn_ = wrap_ || kx == 2 ? 0 : kx + 1, cn_ = wrap_ ? 0 : (kx == 2 ? c + 1 : c);                  \
          load_frags(NW, NP, cn_, kxn_, t + 1);                                                                     \
          _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)                                                          \
            _Pragma("unroll") for (int nh = 0; nh < 2; ++nh)                                                        \
              _Pragma("unroll") for (int m = 0; m < RPW; ++m) acc[m][nh] = mfma16<T>(CW[ky * 2 + nh], CP[m + ky], acc[m][nh]); \
          /* issue order: the NEXT step's fragment reads (RPW + 2 pixel rows + 6 weight pieces) spread over this step's first MFMAs   \
             (the scheduler would otherwise sink the reads to the end of the step to shorten their live ranges, and the next step     \
             would start by waiting for them): 16 rows: 12 reads behind MFMAs 0-7, 9, 11, 13, 15 of 24; 12 / 8 rows: 11 / 10 reads behind the first 11 / 10 of 18 / 12 */   \
          _Pragma("unroll") for (int i_ = 0; i_ < (RPW == 4 ? 8 : RPW + 8); ++i_) {                                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                      \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                      \
          }                                                                                                         \
          _Pragma("unroll") for (int i_ = 0; i_ < (RPW == 4 ? 4 : 0); ++i_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                      \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                      \
          }                                                                                                         \
          __builtin_amdgcn_sched_group_barrier(0x008, RPW == 4 ? 8 : 5 * RPW - 8, 0);                               \
          c = cn_; kx = kxn_;                                                                                       \
          ++t;                                                                                                      \
        }
        // steps 0 and 1, then -- in straight-line code, so that nothing it loads is a loop-carried value the compiler would have to settle
        // (wait for) at a join -- the drain of the previous layer's stores and the request of this layer's epilogue operands
        DC_STEP(0, Aw, Ap, Bw, Bp)
        DC_STEP(1, Bw, Bp, Aw, Ap)
        if (prev_growth) dc_wait_vm<0>();      // the previous layer's write-through stores are acknowledged (issued two steps ago: free); barrier 2 follows
        // epilogue operands of this lane's 4 pixels: slot A = r1 or mask, slot B = r2.  Loaded as ONE 16-byte slot per pixel (slot
        // {0, 2, 1, 3}[g4], the layout the epilogue stores in) and turned into this lane's two channel quads by v_permlane16_swap in the
        // epilogue (the swap is its own inverse): 4 load instructions per operand instead of 8 (16 requests of 8 bytes per wave took ~2.7k cycles to issue)
        u32x4 rA[RPW], rB[RPW];
        T* ydst;                 // this lane's first output element (row 4 cw, 16-byte slot {0, 2, 1, 3}[g4] of its pixel)
        int yrow;                // elements per image row of the output
        {
          // output address and epilogue operands (residuals, mask) of this lane: in registers long before the epilogue
          // (channel offsets are multiples of 32, checked on the host: the second channel half is 16 elements on)
          auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
          // (global address space spelled out: a pointer assembled from two table words is generic to the compiler, and a FLAT load also counts
          // on lgkmcnt -- the next barrier's lgkmcnt(0) would wait for the operands' whole memory latency)
          auto ptr_of = [&](int lo, int hi) { return (const GT*)(((unsigned long long)(unsigned)rfl(hi) << 32) | (unsigned)rfl(lo)); };
          const int4 t2 = *(const int4*)(tl + 8), t3 = *(const int4*)(tl + 12), t4 = *(const int4*)(tl + 16);
          const int p0 = col_ok && orow0 < a.H ? orow0 * a.W + ocol : 0;       // a lane without an output pixel reads pixel 0 (a valid address) and drops it
          const int sl16 = ((g4 & 1) << 1) | (g4 >> 1);
          ydst = (T*)(unsigned long long)ptr_of(t2.x, t2.y) + img * rfl(t2.z) + ((size_t)p0 * rfl(t2.w) + (size_t)(rfl(t3.y) >> 5) * rfl(t3.x) + 8 * sl16);
          yrow = a.W * rfl(t2.w);
          // (the operand registers have ONE definition: a zero-initialised array assigned under an `if` is a merge of two, and the compiler
          // settles such a merge by WAITING for the loads -- measured as a whole memory latency per layer)
          const GT* opA = ptr_of(t3.z, t3.w);
          if ((epi_kind & 3) != 0) {      // (rA stays undefined otherwise -- never read: the merge has one definition, nothing to settle)
            const int a_ps = rfl(t4.y);
            const GT* pa = opA + img * rfl(t4.x) + ((size_t)p0 * a_ps + (size_t)(rfl(t4.w) >> 5) * rfl(t4.z) + 8 * sl16);
#pragma unroll
            for (int m = 0; m < RPW; ++m) {
              const GT* pm = orow0 + m < a.H ? pa + (size_t)m * a.W * a_ps : pa;
              rA[m] = *(const __attribute__((address_space(1))) u32x4*)pm;
            }
          }
        }
        // steps 2 and 3, then the second operand (one burst of requests per CU overruns the L1's miss queue and stalls the issuing wave
        // for a memory latency: two smaller bursts two steps apart)
        DC_STEP(2, Aw, Ap, Bw, Bp)
        DC_STEP(3, Bw, Bp, Aw, Ap)
        {
          auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
          // (global address space spelled out: a pointer assembled from two table words is generic to the compiler, and a FLAT load also counts
          // on lgkmcnt -- the next barrier's lgkmcnt(0) would wait for the operands' whole memory latency)
          auto ptr_of = [&](int lo, int hi) { return (const GT*)(((unsigned long long)(unsigned)rfl(hi) << 32) | (unsigned)rfl(lo)); };
          const int4 t5 = *(const int4*)(tl + 20), t6 = *(const int4*)(tl + 24);
          const int p0 = col_ok && orow0 < a.H ? orow0 * a.W + ocol : 0;
          const int sl16 = ((g4 & 1) << 1) | (g4 >> 1);
          const GT* opB = ptr_of(t5.x, t5.y);
          if ((epi_kind & 3) == 3) {
            const int b_ps = rfl(t5.w);
            const GT* pb = opB + img * rfl(t5.z) + ((size_t)p0 * b_ps + (size_t)(rfl(t6.y) >> 5) * rfl(t6.x) + 8 * sl16);
#pragma unroll
            for (int m = 0; m < RPW; ++m) {
              const GT* pm = orow0 + m < a.H ? pb + (size_t)m * a.W * b_ps : pb;
              rB[m] = *(const __attribute__((address_space(1))) u32x4*)pm;
            }
          }
                }
        int s = 4;
#pragma unroll 1
        for (; s + 1 < ns; s += 2) {
          DC_STEP(s, Aw, Ap, Bw, Bp)
          DC_STEP(s + 1, Bw, Bp, Aw, Ap)
        }
        if (s < ns) {      // 3 or 5 chunks: an odd number of steps -- the next layer starts on set A like every layer
          DC_STEP(s, Aw, Ap, Bw, Bp)
#pragma unroll
          for (int q = 0; q < 6; ++q) Aw[q] = Bw[q];
#pragma unroll
          for (int q = 0; q < RPW + 2; ++q) Ap[q] = Bp[q];
        }
#undef DC_STEP

        // ---- epilogue: lane = pixel l15 of rows 4 cw + m, channels 16 nh + 4 g4 .. + 3 per accumulator.  The operand combination is a
        // compile-time kind (one switch per layer): OPS 0 none, 1 mask, 2 r1, 3 r1 + r2; GROWTH: also group dst_group of the patch and a
        // write-through store; MAXACT: activation as max(v * pos, v * neg).  The two 8-byte channel quads of a lane (nh = 0 / 1) are turned
        // into ONE 16-byte slot by v_permlane16_swap (lane row g4 ends up with slot {0, 2, 1, 3}[g4] of its pixel): one 16-byte LDS write and
        // one 16-byte store per pixel instead of two 8-byte ones each ----
        const float alpha = ctl_alpha[l];
        const f32x4_t b0 = *(const f32x4_t*)(ctl_bias + l * 32 + 4 * g4), b1 = *(const f32x4_t*)(ctl_bias + l * 32 + 16 + 4 * g4);
        const int sl16 = ((g4 & 1) << 1) | (g4 >> 1);
        char* const lds_out = patch + (growth ? dst_group : 0) * kDcGroupBytes + dc_pos(RPW * cw + 1, l15 + 1, sl16);
        T* const y16 = ydst;
        auto epilogue = [&](auto growth_c, auto ops_c, auto max_c) {
          constexpr bool GROWTH = decltype(growth_c)::v != 0, MAXACT = decltype(max_c)::v != 0;
          constexpr int OPS = decltype(ops_c)::v;
#pragma unroll
          for (int m = 0; m < RPW; ++m) {
            const bool ok = col_ok && orow0 + m < a.H;
            dc_u32x2 pk[2], eA[2], eB[2];
            if constexpr (OPS >= 1) {
              const dc_u32x2 ux = __builtin_amdgcn_permlane16_swap(rA[m].x, rA[m].z, false, false), uy = __builtin_amdgcn_permlane16_swap(rA[m].y, rA[m].w, false, false);
              eA[0] = dc_u32x2{ux.x, uy.x}; eA[1] = dc_u32x2{ux.y, uy.y};
            }
            if constexpr (OPS == 3) {
              const dc_u32x2 ux = __builtin_amdgcn_permlane16_swap(rB[m].x, rB[m].z, false, false), uy = __builtin_amdgcn_permlane16_swap(rB[m].y, rB[m].w, false, false);
              eB[0] = dc_u32x2{ux.x, uy.x}; eB[1] = dc_u32x2{ux.y, uy.y};
            }
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
              float v4[4], t4[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float v = __builtin_fmaf(alpha, acc[m][nh][i], nh ? b1[i] : b0[i]);
                // v * (v > 0 ? pos : neg); with pos >= neg >= 0 that is max(v * pos, v * neg): two multiplies and a max, no compare / select pair
                if constexpr (MAXACT) v4[i] = __builtin_fmaxf(v * ps_pos, v * ps_neg);
                else v4[i] = v * (v > 0.f ? ps_pos : ps_neg);
              }
              if constexpr (OPS >= 2) { dc_widen4<T>(eA[nh], t4);
#pragma unroll
                for (int i = 0; i < 4; ++i) v4[i] = __builtin_fmaf(r1s, t4[i], v4[i]); }
              if constexpr (OPS == 3) { dc_widen4<T>(eB[nh], t4);
#pragma unroll
                for (int i = 0; i < 4; ++i) v4[i] = __builtin_fmaf(r2s, t4[i], v4[i]); }
              if constexpr (OPS == 1) { dc_widen4<T>(eA[nh], t4);
#pragma unroll
                for (int i = 0; i < 4; ++i) v4[i] *= t4[i] > 0.f ? 1.f : mslope; }
              pk[nh] = dc_narrow4<T>(v4);
              if (!ok) pk[nh] = dc_u32x2{0u, 0u};      // pixels beyond the image are zero padding for the layers that follow
            }
            const dc_u32x2 sx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
            const dc_u32x2 sy = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
            const u32x4 q = {sx.x, sy.x, sx.y, sy.y};      // lower quad (both dwords), then upper quad of this lane's 16-byte slot
            if constexpr (GROWTH) *(u32x4*)(lds_out + m * (kDcPC * 64)) = q;
            if (ok) {
              T* dst = y16 + (size_t)m * yrow;
              // growth layers: write-through (sc1), the neighbours read the halo from memory inside this launch
              if constexpr (GROWTH) dc_store16_sc1(dst, q);
              else *(u32x4*)dst = q;
            }
          }
        };
        switch (epi_kind) {
          case 0: epilogue(IC<0>{}, IC<0>{}, IC<0>{}); break;
          case 1: epilogue(IC<0>{}, IC<1>{}, IC<0>{}); break;
          case 2: epilogue(IC<0>{}, IC<2>{}, IC<0>{}); break;
          case 3: epilogue(IC<0>{}, IC<3>{}, IC<0>{}); break;
          case 4: epilogue(IC<1>{}, IC<0>{}, IC<0>{}); break;
          case 5: epilogue(IC<1>{}, IC<1>{}, IC<0>{}); break;
          case 6: epilogue(IC<1>{}, IC<2>{}, IC<0>{}); break;
          case 7: epilogue(IC<1>{}, IC<3>{}, IC<0>{}); break;
          case 8: epilogue(IC<0>{}, IC<0>{}, IC<1>{}); break;
          case 9: epilogue(IC<0>{}, IC<1>{}, IC<1>{}); break;
          case 10: epilogue(IC<0>{}, IC<2>{}, IC<1>{}); break;
          case 11: epilogue(IC<0>{}, IC<3>{}, IC<1>{}); break;
          case 12: epilogue(IC<1>{}, IC<0>{}, IC<1>{}); break;
          case 13: epilogue(IC<1>{}, IC<1>{}, IC<1>{}); break;
          case 14: epilogue(IC<1>{}, IC<2>{}, IC<1>{}); break;
          default: epilogue(IC<1>{}, IC<3>{}, IC<1>{}); break;
        }
      }
    }
    if (giveups && lane == 0) atomicAdd(a.hdr, giveups);
  }
  // ---- the last workgroup to finish advances the epoch for the next launch on the stream ----
  if (tid == 0) {
    const int done = atomicAdd(a.hdr + 2, 1);
    if (done == (int)gridDim.x - 1) { atomicExch(a.hdr + 2, 0); atomicAdd(a.hdr + 1, 1); }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
size_t dense_chain_workspace_bytes_impl() { return 64 + sizeof(int) * 4 * kDcMaxTiles; }      // header + flags of 4 growth layers x tiles of one call

// Validates that `layers` are the convs of one dense chain and fills the kernel arguments at image 0's bases.
static int dense_chain_fill(const srganfd_conv_args* layers, int n, DcK& K) {
  if (!layers || n < 2 || n > 5) return set_err(SRGANFD_EINVAL, "dense_chain: 2..5 layers");
  memset(&K, 0, sizeof(K));
  const srganfd_conv_args& a0 = layers[0];
  if (a0.dtype != SRGANFD_F16 && a0.dtype != SRGANFD_BF16) return set_err(SRGANFD_EINVAL, "dense_chain: 16-bit dtypes only");
  int nl = 0, steps = 0;
  for (int i = 0; i < n; ++i) {
    const srganfd_conv_args& a = layers[i];
    ConvK k;
    const int rc = conv_fill_k(&a, k);
    if (rc != SRGANFD_OK) return rc;
    const bool last = i == n - 1;
    if (a.dtype != a0.dtype || a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.up || a.out_sy > 1 || a.out_sx > 1 || a.out_classes == 4 || a.y2.ptr || a.y_f32 ||
        a.n != a0.n || a.h_in != a0.h_in || a.w_in != a0.w_in || a.x.ptr != a0.x.ptr || a.x.c0 != a0.x.c0 || a.x.cstride != a0.x.cstride || a.x.planar != a0.x.planar ||
        a.cin != 64 + 32 * i || a.cout != (last ? 64 : 32) || a.cout_store != a.cout || !k.fast_epi)
      return set_err(SRGANFD_EINVAL, "dense_chain: layer %d is not conv %d of a dense block (3x3 stride 1, %d -> %d channels over one buffer)", i, i + 1, 64 + 32 * i, last ? 64 : 32);
    if ((k.y_c0 | k.r1_c0 | k.r2_c0 | k.m_c0) & 31) return set_err(SRGANFD_EINVAL, "dense_chain: layer %d: channel offsets must be multiples of 32", i);
    if (a.r2.ptr && !a.r1.ptr) return set_err(SRGANFD_EINVAL, "dense_chain: layer %d has r2 without r1", i);
    if (a.mask.ptr && (a.r1.ptr || a.r2.ptr)) return set_err(SRGANFD_EINVAL, "dense_chain: layer %d has a mask and residuals (the epilogue keeps two operands)", i);
    if (!last && (a.y.ptr != a0.x.ptr || a.y.c0 != a0.x.c0 + a.cin || a.y.cstride != a0.x.cstride || a.y.planar != a0.x.planar))
      return set_err(SRGANFD_EINVAL, "dense_chain: layer %d must write channels [%d, %d) of the buffer it reads", i, a.cin, a.cin + 32);
    if (i == 0) { K.x = (const char*)a.x.ptr; K.xC = k.xC; K.x_ps = k.x_ps; K.x_base = k.x_base; K.x_cs = k.x_cs; }
    for (int h = 0; h < (last ? 2 : 1); ++h) {
      DcLayer& L = K.L[nl++];
      L.w = (const char*)a.w_packed + (size_t)h * k.nChunks * 18432;
      L.bias = a.bias ? a.bias + 32 * h : nullptr;
      L.alpha_dev = a.alpha_dev;
      L.y = (char*)a.y.ptr; L.r1 = (const char*)a.r1.ptr; L.r2 = (const char*)a.r2.ptr; L.mask = (const char*)a.mask.ptr;
      L.yC = k.yC; L.y_c0 = k.y_c0 + 32 * h; L.y_ps = k.y_ps; L.y_gs = k.y_gs;
      L.r1C = k.r1C; L.r1_c0 = k.r1_c0 + 32 * h; L.r1_ps = k.r1_ps; L.r1_gs = k.r1_gs;
      L.r2C = k.r2C; L.r2_c0 = k.r2_c0 + 32 * h; L.r2_ps = k.r2_ps; L.r2_gs = k.r2_gs;
      L.mC = k.mC; L.m_c0 = k.m_c0 + 32 * h; L.m_ps = k.m_ps; L.m_gs = k.m_gs;
      L.nChunks = k.nChunks;
      L.dst_group = last ? -1 : a.cin / 32;
      L.alpha = a.alpha; L.neg = a.act == SRGANFD_ACT_LRELU ? a.slope : (a.act == SRGANFD_ACT_RELU ? 0.f : 1.f);
      L.post_scale = a.post_scale; L.r1s = a.r1_scale; L.r2s = a.r2_scale; L.mask_slope = a.mask_slope;
      {
        const float ps_pos = L.post_scale, ps_neg = L.neg * L.post_scale;
        const int ops = L.r1 ? (L.r2 ? 3 : 2) : (L.mask ? 1 : 0);
        const int kind = (L.dst_group >= 0 ? 4 : 0) + ops + (ps_neg >= 0.f && ps_pos >= ps_neg ? 8 : 0);
        int* t = K.tab[nl - 1];
        t[0] = L.nChunks; t[1] = L.dst_group; t[2] = kind;
        memcpy(&t[3], &ps_pos, 4); memcpy(&t[4], &ps_neg, 4); memcpy(&t[5], &L.r1s, 4); memcpy(&t[6], &L.r2s, 4); memcpy(&t[7], &L.mask_slope, 4);
        memcpy(&t[8], &L.y, 8); t[10] = L.yC; t[11] = L.y_ps; t[12] = L.y_gs; t[13] = L.y_c0;
        if (L.r1) { memcpy(&t[14], &L.r1, 8); t[16] = L.r1C; t[17] = L.r1_ps; t[18] = L.r1_gs; t[19] = L.r1_c0; }
        else if (L.mask) { memcpy(&t[14], &L.mask, 8); t[16] = L.mC; t[17] = L.m_ps; t[18] = L.m_gs; t[19] = L.m_c0; }
        else { memcpy(&t[14], &L.y, 8); t[16] = L.yC; t[17] = L.y_ps; t[18] = L.y_gs; t[19] = L.y_c0; }      // absent: any valid view (the requests are unconditional)
        if (L.r2) { memcpy(&t[20], &L.r2, 8); t[22] = L.r2C; t[23] = L.r2_ps; t[24] = L.r2_gs; t[25] = L.r2_c0; }
        else { memcpy(&t[20], &L.y, 8); t[22] = L.yC; t[23] = L.y_ps; t[24] = L.y_gs; t[25] = L.y_c0; }
      }
      steps += 3 * k.nChunks;
    }
  }
  K.nLayers = nl; K.stepsPerPass = steps;
  K.N = a0.n; K.H = a0.h_in; K.W = a0.w_in;
  const int cus = conv_device_cus();
  // 8 x 16 (else 12 x 16) tiles when the whole batch is one pass that way too (more workgroups, a shorter pass); else 16 x 16 (the densest form)
  K.tiles_x = ceil_div(K.W, kDcTW);
  K.rpw = (long long)K.N * K.tiles_x * ceil_div(K.H, 8) <= cus ? 2 : ((long long)K.N * K.tiles_x * ceil_div(K.H, 12) <= cus ? 3 : 4);
  if (const char* e = getenv("SRGANFD_DC_RPW")) { const int v = atoi(e); if (v >= 2 && v <= 4) K.rpw = v; }      // same-box A/B switch
  K.tiles_y = ceil_div(K.H, 4 * K.rpw);
  K.tpi = K.tiles_x * K.tiles_y;
  K.totalTiles = K.N * K.tpi;
  K.ipl = K.tpi > cus ? 0 : (cus / K.tpi < K.N ? cus / K.tpi : K.N);      // every tile of a pass must be resident at once (one workgroup per CU)
  return SRGANFD_OK;
}

static int dense_chain_limits(const DcK& K) {
  if (K.ipl < 1) return set_err(SRGANFD_EINVAL, "dense_chain: one image is %d tiles of %d x 16, more than the device has CUs", K.tpi, 4 * K.rpw);
  if (K.totalTiles > kDcMaxTiles) return set_err(SRGANFD_EINVAL, "dense_chain: %d tiles in one call (limit %d)", K.totalTiles, kDcMaxTiles);
  return SRGANFD_OK;
}

int dense_chain_check_impl(const srganfd_conv_args* layers, int n) {
  DcK K;
  const int rc = dense_chain_fill(layers, n, K);
  return rc != SRGANFD_OK ? rc : dense_chain_limits(K);
}

int dense_chain_impl(const srganfd_conv_args* layers, int n, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  DcK K;
  int rc = dense_chain_fill(layers, n, K);
  if (rc != SRGANFD_OK) return rc;
  if ((rc = dense_chain_limits(K)) != SRGANFD_OK) return rc;
  if (!workspace || workspace_bytes < dense_chain_workspace_bytes_impl()) return set_err(SRGANFD_ENOSPC, "dense_chain: workspace too small");
  if (g_describe) { snprintf(g_describe, g_describe_len, "dense_chain_kernel<%s,%d layers>", layers[0].dtype == SRGANFD_F16 ? "f16" : "bf16", n); return SRGANFD_OK; }
  K.hdr = (int*)workspace;
  K.flags = (int*)((char*)workspace + 64);
  const bool f16 = layers[0].dtype == SRGANFD_F16;
  const int variant = (f16 ? 1 : 0) + 2 * (4 - K.rpw);       // rows per wave 4, 3, 2 x {bf16, f16}
  const void* const kerns[6] = {(const void*)dense_chain_kernel<bf16_t, 4>, (const void*)dense_chain_kernel<f16_t, 4>, (const void*)dense_chain_kernel<bf16_t, 3>,
                                (const void*)dense_chain_kernel<f16_t, 3>, (const void*)dense_chain_kernel<bf16_t, 2>, (const void*)dense_chain_kernel<f16_t, 2>};
  const int lds = K.rpw == 2 ? DcGeo<2>::Lds : (K.rpw == 3 ? DcGeo<3>::Lds : DcGeo<4>::Lds);
  static unsigned long long attr_done[6] = {0, 0, 0, 0, 0, 0};
  if (!g_dry_run) {
    int dev = 0;
    SRGANFD_HIP_CHECK(hipGetDevice(&dev));
    if (!(attr_done[variant] >> (dev & 63) & 1ULL)) {
      SRGANFD_HIP_CHECK(hipFuncSetAttribute(kerns[variant], hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr_done[variant] |= 1ULL << (dev & 63);
    }
  }
  const unsigned grid = (unsigned)(K.ipl * K.tpi);
  switch (variant) {
    case 0: SRGANFD_LAUNCH((dense_chain_kernel<bf16_t, 4>), dim3(grid), dim3(kDcThreads), lds, stream, K); break;
    case 1: SRGANFD_LAUNCH((dense_chain_kernel<f16_t, 4>), dim3(grid), dim3(kDcThreads), lds, stream, K); break;
    case 2: SRGANFD_LAUNCH((dense_chain_kernel<bf16_t, 3>), dim3(grid), dim3(kDcThreads), lds, stream, K); break;
    case 3: SRGANFD_LAUNCH((dense_chain_kernel<f16_t, 3>), dim3(grid), dim3(kDcThreads), lds, stream, K); break;
    case 4: SRGANFD_LAUNCH((dense_chain_kernel<bf16_t, 2>), dim3(grid), dim3(kDcThreads), lds, stream, K); break;
    default: SRGANFD_LAUNCH((dense_chain_kernel<f16_t, 2>), dim3(grid), dim3(kDcThreads), lds, stream, K); break;
  }
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

}  // namespace srganfd
