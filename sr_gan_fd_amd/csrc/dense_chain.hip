// dense_chain.hip -- a whole dense block as ONE launch whose activations never leave LDS.
//
// Replaces, for launches of at most one 8 x 32-pixel tile per compute unit (the reference's own crop sizes: bsrgan_config.py:101-102
// 72 x 72, esrgan_config.py:73-74 32 x 32, rrdbnet_config.py:51-52 48 x 48, aesrgan_config.py:102-103 60 x 60 at batch 8-16), the five
// srganfd_conv2d launches of _ResidualDenseBlock.forward (BSRGAN/model.py:51-62) -- and, with the data-gradient operands, the five
// launches of its backward pass (the same dense structure over the stacked output gradients, engine.py) -- by one persistent launch.
//
// Why: at those sizes a conv launch is a single tile per workgroup and costs 13-15 us whatever it computes (profiles/
// r05_reference_shapes_before.txt): kernel boundary, argument loads, address set-up, the round trip of the tile's own data through
// HBM/L2 and the LDS commit of every chunk are paid per layer; neither removing the boundaries (layer-persistent launch) nor the
// commits (LDS-DMA streaming conv) changed that (profiles/r05_chain_at_small_shapes.txt, r05_stream_conv_at_small_shapes.txt).
//
// MI355X mapping (this is what 160 KB of LDS per CU and 256 CUs are for -- 40 MB of on-chip activation store):
//   * one workgroup (8 waves) per CU owns ONE tile of 8 rows x 32 pixels of one image through all layers.  The tile's 192 channels
//     with a 1-pixel halo (10 x 34 pixels x 6 groups of 32 channels x 2 B = 130,560 B) are LDS-resident: groups 0-1 (the block
//     input) are loaded once, group 2 + k is written by layer k's epilogue straight from the accumulators.
//   * only the halo of a NEW group crosses workgroups: the epilogue also stores the tile to the block's HBM buffer (the weight
//     gradient and the next launch need it there anyway) with write-through (sc1) stores, drains them, and publishes one flag per
//     (layer, tile); a consumer polls the flags of its <= 8 neighbours with relaxed agent-scope loads and reads the 84 halo pixels
//     with sc1 loads -- row 1 of the micro-architecture guide's hand-off table; no fences, no placement assumption.  The halo of layer
//     k is first needed by the LAST chunk of layer k + 1, so the hand-off hides behind that layer's older chunks.
//   * weights stream through a 5-slot LDS ring of 6 KB kernel-column pieces (3 kernel rows x 2 channel halves x 1 KB B fragments in
//     pack.hip's order) by LDS-DMA, four pieces ahead of the MFMA step that consumes them, one barrier per step.
//   * every layer is "32 * n input channels -> 32 output channels" (the 64-channel conv5 runs as its two 32-channel n-tiles); wave
//     (rp, ph) computes rows 2 rp, 2 rp + 1 x pixels 16 ph .. 16 ph + 15 x 32 channels: 12 v_mfma_f32_16x16x32 per step with the
//     operands swapped (A = weights, B = pixels), so a lane ends up with 4 consecutive channels of one pixel: 8-byte LDS / global writes.
// Same arithmetic contract as conv_igemm.hip (include/srganfd.h, srganfd_conv2d): fp32 accumulation in chunk, kernel-column,
// kernel-row order, v = post_scale * act(alpha * acc + bias) + r1s * r1 + r2s * r2, masked, rounded once to the 16-bit type.
#include "conv_common.hpp"
#include <stdlib.h>
#include <string.h>

namespace srganfd {

int conv_fill_k(const srganfd_conv_args* a, ConvK& k);   // conv_igemm.hip: validation + operand strides of one conv launch

namespace {
constexpr int kDcMaxLayers = 6;      // four growth convs + the two 32-channel halves of the closing conv
constexpr int kDcPR = 10, kDcPC = 34, kDcNPIX = kDcPR * kDcPC;
constexpr int kDcGroupBytes = kDcNPIX * 64;          // one 32-channel group of the resident patch
constexpr int kDcGroups = 6;
constexpr int kDcSlot = 6144, kDcSlots = 5, kDcAhead = 4;
constexpr int kDcLds = kDcGroups * kDcGroupBytes + kDcSlots * kDcSlot + 64;     // 161,344 B

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
  int nLayers, totalSteps;
  int N, H, W, tiles_x, tiles_y, ntiles;
  int* flags;                // [growth layer][tile], zeroed before the launch
  int* err;                  // [0] += 1 for every hand-off wait that gave up
  int dbg;                   // TIMING EXPERIMENT (SRGANFD_DC_DBG): 1 no hand-off, 2 no epilogue memory traffic, 4 no memset, 8 no MFMA steps
};

__device__ __forceinline__ unsigned dc_lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
// One LDS-DMA piece (see wgrad.hip glds16): 64 lanes x 16 bytes, per-lane source, wave-uniform LDS destination; outside the
// compiler's wait-count bookkeeping, the kernel counts vmcnt itself.
__device__ __forceinline__ void dc_glds16(const void* gsrc, unsigned lds_dst) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(dst) : "m0");
#pragma clang diagnostic pop
}
template <int N> __device__ __forceinline__ void dc_wait_vm() {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#pragma clang diagnostic pop
}
// byte position of (patch row, patch column, 16-byte slot) inside one group: pixel-major, slot XOR 2 * ((column >> 2) & 1) -- the
// column-keyed swizzle of conv_igemm.hip's 16x16x32 fragment reads (conflict-free ds_read_b128 for every kernel column)
__device__ __forceinline__ int dc_pos(int prow, int pcol, int slot) { return (prow * kDcPC + pcol) * 64 + ((slot ^ (((pcol >> 2) & 1) << 1)) << 4); }

typedef unsigned long long dc_u64;
typedef __attribute__((ext_vector_type(2))) unsigned int dc_u32x2;
}  // namespace

template <typename T>
__global__ __launch_bounds__(512, 2) void dense_chain_kernel(const DcK a) {
  using Frag = typename FragAB<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;
  char* const ring = smem + kDcGroups * kDcGroupBytes;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rp = wave >> 1, ph = wave & 1;          // this wave's row pair and pixel half
  const int l15 = lane & 15, g4 = lane >> 4;

  const int tile = blockIdx.x;
  const int tx = tile % a.tiles_x, t1 = tile / a.tiles_x;
  const int ty = t1 % a.tiles_y, n = t1 / a.tiles_y;
  const int oy0 = ty * 8, ox0 = tx * 32;
  const size_t ipix = (size_t)a.H * a.W;

  // ---- weight stream: producer cursor = the next kernel-column step to request ----
  int pl = 0, pc = 0, pk = 0, pt = 0;
  auto issue_next = [&]() {
    if (pl < a.nLayers) {
      if (wave < 6) {
        const int ky = wave >> 1, nh = wave & 1;
        const char* src = a.L[pl].w + (size_t)(((pc * 9 + ky * 3 + pk) * 2 + nh) * 1024) + lane * 16;
        dc_glds16(src, dc_lds_addr(ring) + (unsigned)((pt % kDcSlots) * kDcSlot + wave * 1024));
      }
      ++pt;
      if (++pk == 3) { pk = 0; if (++pc == a.L[pl].nChunks) { pc = 0; ++pl; } }
    }
  };
#pragma unroll
  for (int i = 0; i < kDcAhead; ++i) issue_next();

  // ---- resident patch: groups 0, 1 from the block input (zeros outside the image), zero halo ring for the groups to come ----
  {
    const T* xi = (const T*)a.x + (size_t)n * ipix * a.xC + a.x_base;
    for (int item = tid; item < kDcNPIX * 8; item += 512) {
      const int pix = item >> 3, g = (item >> 2) & 1, slot = item & 3;
      const int prow = pix / kDcPC, pcol = pix - prow * kDcPC;
      const int gy = oy0 - 1 + prow, gx = ox0 - 1 + pcol;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = *(const u32x4*)(xi + ((size_t)(gy * a.W + gx) * a.x_ps + (size_t)g * a.x_cs + slot * 8));
      *(u32x4*)(patch + g * kDcGroupBytes + dc_pos(prow, pcol, slot)) = v;
    }
    for (int item = tid; item < 84 * 16; item += 512) {
      const int hp = item >> 4, g = 2 + ((item >> 2) & 3), slot = item & 3;
      const int prow = hp < 34 ? 0 : (hp < 68 ? 9 : (hp < 76 ? 1 + hp - 68 : 1 + hp - 76));
      const int pcol = hp < 34 ? hp : (hp < 68 ? hp - 34 : (hp < 76 ? 0 : 33));
      *(u32x4*)(patch + g * kDcGroupBytes + dc_pos(prow, pcol, slot)) = u32x4{0u, 0u, 0u, 0u};
    }
  }

  // fragment address terms of this lane: the three kernel columns (pixel 16 ph + l15 + kx of the patch row), channel slot g4
  int colt[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) { const int col = 16 * ph + l15 + kx; colt[kx] = col * 64 + ((g4 ^ (((col >> 2) & 1) << 1)) << 4); }
  const int rowoff = (2 * rp) * kDcPC * 64;

  // hand-off of growth layer `gl`'s output (group gd, tensor y of that layer): wait for the neighbours, read the 84 halo pixels
  auto halo_in = [&](int gl) {
    const DcLayer& P = a.L[gl];
    if (wave == 0) {
      const int j = lane < 8 ? lane : 0, q = j < 4 ? j : j + 1;         // the eight neighbours (3 x 3 without the centre)
      const int dy = q / 3 - 1, dx = q % 3 - 1;
      const bool valid = lane < 8 && ty + dy >= 0 && ty + dy < a.tiles_y && tx + dx >= 0 && tx + dx < a.tiles_x;
      const int* f = a.flags + (size_t)gl * a.ntiles + (valid ? tile + dy * a.tiles_x + dx : tile);
      for (int spins = 0;; ++spins) {
        const int v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_ballot_w64(valid && v == 0) == 0ull) break;
        if (spins > (1 << 22)) { if (lane == 0) atomicAdd(a.err, 1); break; }     // seconds: never in a correct run; wrong results, no hang
        __builtin_amdgcn_s_sleep(4);
      }
    }
    __syncthreads();
    const T* yi = (const T*)P.y + (size_t)n * ipix * P.yC;
    char* pg = patch + P.dst_group * kDcGroupBytes;
    for (int item = tid; item < 84 * 8; item += 512) {
      const int hp = item >> 3, piece = item & 7;
      const int prow = hp < 34 ? 0 : (hp < 68 ? 9 : (hp < 76 ? 1 + hp - 68 : 1 + hp - 76));
      const int pcol = hp < 34 ? hp : (hp < 68 ? hp - 34 : (hp < 76 ? 0 : 33));
      const int gy = oy0 - 1 + prow, gx = ox0 - 1 + pcol;
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        const int cc = P.y_c0 + 4 * piece;
        const dc_u64* src = (const dc_u64*)(yi + ((size_t)(gy * a.W + gx) * P.y_ps + (size_t)(cc >> 5) * P.y_gs + (cc & 31)));
        const dc_u64 v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1: served by L2 / memory, never a stale L1 line
        *(dc_u64*)(pg + dc_pos(prow, pcol, piece >> 1) + 8 * (piece & 1)) = v;
      }
    }
    // (the next step's barrier publishes these LDS writes; nobody has read this group yet)
  };

  int t = 0;       // kernel-column steps consumed so far (all layers)
  for (int l = 0; l < a.nLayers; ++l) {
    const DcLayer& Ld = a.L[l];
    const int nCh = Ld.nChunks;
    f32x4_t acc[2][2];       // [row of the pair][16-channel half]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) acc[m][nh] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < nCh; ++c) {
      // the newest group (written by the previous growth layer) is read by this layer's last chunk only: take its halo in now
      if (c == nCh - 1 && l > 0 && a.L[l - 1].dst_group == c && c >= 2 && !(a.dbg & 1)) halo_in(l - 1);
      const char* pg = patch + c * kDcGroupBytes + rowoff;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        // this wave's piece of step t landed (at most three younger pieces of its own are in flight) ...
        const int rem = a.totalSteps - 1 - t;
        if (rem >= 3) dc_wait_vm<3>(); else if (rem == 2) dc_wait_vm<2>(); else if (rem == 1) dc_wait_vm<1>(); else dc_wait_vm<0>();
        __syncthreads();      // ... and everybody's: slot t % 5 is complete, slot (t - 1) % 5 is free
        issue_next();         // step t + 4 into it
        const char* rs = ring + (t % kDcSlots) * kDcSlot + lane * 16;
        Frag ap[4], bw[3][2];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) ap[rr] = *(const Frag*)(pg + rr * kDcPC * 64 + colt[kx]);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) bw[ky][nh] = *(const Frag*)(rs + (ky * 2 + nh) * 1024);
        __builtin_amdgcn_s_setprio(1);
        if (!(a.dbg & 8))
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m][nh] = mfma16<T>(bw[ky][nh], ap[m + ky], acc[m][nh]);
        __builtin_amdgcn_s_setprio(0);
        ++t;
      }
    }

    // ---- epilogue: lane = pixel 16 ph + l15 of rows 2 rp + m, channels 16 nh + 4 g4 .. + 3 ----
    float alpha = Ld.alpha;
    if (Ld.alpha_dev) alpha *= *Ld.alpha_dev;
    const float ps_pos = Ld.post_scale, ps_neg = Ld.neg * Ld.post_scale;
    const bool growth = Ld.dst_group >= 0;
    const size_t img = (size_t)n * ipix;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int oy = oy0 + 2 * rp + m, ox = ox0 + 16 * ph + l15;
      const bool ok = oy < a.H && ox < a.W;
      const int p = oy * a.W + ox;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        const int co = 16 * nh + 4 * g4;
        float v4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = __builtin_fmaf(alpha, acc[m][nh][i], Ld.bias ? Ld.bias[co + i] : 0.f);
          v4[i] = v * (v > 0.f ? ps_pos : ps_neg);
        }
        auto widen4 = [](const dc_u32x2 q, float* f) {
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
        auto ld = [&](const char* base, int Cs, int c0, int ps, int gs) -> dc_u32x2 {
          const int cc = c0 + co;
          return ok && !(a.dbg & 2) ? *(const dc_u32x2*)((const T*)base + img * Cs + ((size_t)p * ps + (size_t)(cc >> 5) * gs + (cc & 31))) : dc_u32x2{0u, 0u};
        };
        float t4[4];
        if (Ld.r1) { widen4(ld(Ld.r1, Ld.r1C, Ld.r1_c0, Ld.r1_ps, Ld.r1_gs), t4);
#pragma unroll
          for (int i = 0; i < 4; ++i) v4[i] = __builtin_fmaf(Ld.r1s, t4[i], v4[i]); }
        if (Ld.r2) { widen4(ld(Ld.r2, Ld.r2C, Ld.r2_c0, Ld.r2_ps, Ld.r2_gs), t4);
#pragma unroll
          for (int i = 0; i < 4; ++i) v4[i] = __builtin_fmaf(Ld.r2s, t4[i], v4[i]); }
        if (Ld.mask) { widen4(ld(Ld.mask, Ld.mC, Ld.m_c0, Ld.m_ps, Ld.m_gs), t4);
#pragma unroll
          for (int i = 0; i < 4; ++i) v4[i] *= t4[i] > 0.f ? 1.f : Ld.mask_slope; }
        dc_u32x2 pk;
        if constexpr (Elem<T>::kDtype == SRGANFD_F16) {
          typedef __attribute__((ext_vector_type(4))) _Float16 h4;
          const h4 hv = {(_Float16)v4[0], (_Float16)v4[1], (_Float16)v4[2], (_Float16)v4[3]};
          pk = __builtin_bit_cast(dc_u32x2, hv);
        } else {
          pk = dc_u32x2{(unsigned)f2bf(v4[0]) | ((unsigned)f2bf(v4[1]) << 16), (unsigned)f2bf(v4[2]) | ((unsigned)f2bf(v4[3]) << 16)};
        }
        if (!ok) pk = dc_u32x2{0u, 0u};      // pixels beyond the image are zero padding for the layers that follow
        if (growth) *(dc_u32x2*)(patch + Ld.dst_group * kDcGroupBytes + dc_pos(2 * rp + m + 1, 16 * ph + l15 + 1, 2 * nh + (g4 >> 1)) + 8 * (g4 & 1)) = pk;
        if (ok && !(a.dbg & 2)) {
          const int cc = Ld.y_c0 + co;
          T* dst = (T*)Ld.y + img * Ld.yC + ((size_t)p * Ld.y_ps + (size_t)(cc >> 5) * Ld.y_gs + (cc & 31));
          // growth layers: write-through (sc1), the neighbours read the halo from L2 / memory inside this launch
          if (growth) __hip_atomic_store((dc_u64*)dst, __builtin_bit_cast(dc_u64, pk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else *(dc_u32x2*)dst = pk;
        }
      }
    }
    if (growth) {
      // every wave's stores are acknowledged, then ONE lane publishes the tile (this drain also retires the wave's weight pieces in
      // flight, which keeps the counted waits above exact: nothing but weight pieces is ever outstanding inside a layer)
      dc_wait_vm<0>();
      __syncthreads();
      if (tid == 0) __hip_atomic_store(a.flags + (size_t)l * a.ntiles + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      dc_wait_vm<0>();      // the epilogue's operand loads and stores are out of the vmcnt queue before the next layer counts it
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
size_t dense_chain_workspace_bytes_impl() { return 64 + sizeof(int) * 4 * 1024; }      // [err, pad] + flags of 4 growth layers x <= 1024 tiles

// Validates that `layers` are the convs of one dense chain and fills the kernel arguments for images [0, n) at image 0's bases.
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
      steps += 3 * k.nChunks;
    }
  }
  K.nLayers = nl; K.totalSteps = steps;
  K.N = a0.n; K.H = a0.h_in; K.W = a0.w_in;
  K.tiles_x = ceil_div(K.W, 32); K.tiles_y = ceil_div(K.H, 8);
  return SRGANFD_OK;
}

// images per launch: every tile of a launch must be resident at once (one workgroup per CU)
static int dense_chain_images_per_launch(const DcK& K) {
  const int per = K.tiles_x * K.tiles_y, cus = conv_device_cus();
  return per > cus || per > 1024 ? 0 : (cus / per < K.N ? cus / per : K.N);
}

int dense_chain_check_impl(const srganfd_conv_args* layers, int n) {
  DcK K;
  const int rc = dense_chain_fill(layers, n, K);
  if (rc != SRGANFD_OK) return rc;
  if (dense_chain_images_per_launch(K) < 1) return set_err(SRGANFD_EINVAL, "dense_chain: one image is %d tiles of 8 x 32, more than the device has CUs", K.tiles_x * K.tiles_y);
  return SRGANFD_OK;
}

int dense_chain_impl(const srganfd_conv_args* layers, int n, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  DcK K;
  int rc = dense_chain_fill(layers, n, K);
  if (rc != SRGANFD_OK) return rc;
  const int ipl = dense_chain_images_per_launch(K);
  if (ipl < 1) return set_err(SRGANFD_EINVAL, "dense_chain: one image is %d tiles of 8 x 32, more than the device has CUs", K.tiles_x * K.tiles_y);
  if (!workspace || workspace_bytes < dense_chain_workspace_bytes_impl()) return set_err(SRGANFD_ENOSPC, "dense_chain: workspace too small");
  if (g_describe) { snprintf(g_describe, g_describe_len, "dense_chain_kernel<%s,%d layers>", layers[0].dtype == SRGANFD_F16 ? "f16" : "bf16", n); return SRGANFD_OK; }
  K.err = (int*)workspace;
  { const char* e = getenv("SRGANFD_DC_DBG"); K.dbg = e ? atoi(e) : 0; }
  K.flags = (int*)((char*)workspace + 64);
  const bool f16 = layers[0].dtype == SRGANFD_F16;
  static unsigned long long attr_done[2] = {0, 0};
  if (!g_dry_run) {
    int dev = 0;
    SRGANFD_HIP_CHECK(hipGetDevice(&dev));
    if (!(attr_done[f16] >> (dev & 63) & 1ULL)) {
      if (f16) SRGANFD_HIP_CHECK(hipFuncSetAttribute((const void*)dense_chain_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, kDcLds));
      else SRGANFD_HIP_CHECK(hipFuncSetAttribute((const void*)dense_chain_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, kDcLds));
      attr_done[f16] |= 1ULL << (dev & 63);
    }
  }
  const size_t ipix = (size_t)K.H * K.W, esz = 2;
  const int Ntot = K.N;
  for (int n0 = 0; n0 < Ntot; n0 += ipl) {
    DcK S = K;
    S.N = Ntot - n0 < ipl ? Ntot - n0 : ipl;
    S.ntiles = S.N * K.tiles_x * K.tiles_y;
    S.x = K.x + (size_t)n0 * ipix * K.xC * esz;
    for (int i = 0; i < K.nLayers; ++i) {
      DcLayer& L = S.L[i];
      L.y += (size_t)n0 * ipix * L.yC * esz;
      if (L.r1) L.r1 += (size_t)n0 * ipix * L.r1C * esz;
      if (L.r2) L.r2 += (size_t)n0 * ipix * L.r2C * esz;
      if (L.mask) L.mask += (size_t)n0 * ipix * L.mC * esz;
    }
    if (!g_dry_run && !(S.dbg & 4)) SRGANFD_HIP_CHECK(hipMemsetAsync(S.flags, 0, sizeof(int) * 4 * (size_t)S.ntiles, stream));
    if (f16) SRGANFD_LAUNCH(dense_chain_kernel<f16_t>, dim3((unsigned)S.ntiles), dim3(512), kDcLds, stream, S);
    else SRGANFD_LAUNCH(dense_chain_kernel<bf16_t>, dim3((unsigned)S.ntiles), dim3(512), kDcLds, stream, S);
    SRGANFD_HIP_CHECK(hipGetLastError());
  }
  return SRGANFD_OK;
}

}  // namespace srganfd
