// conv_pair.hip -- EXPERIMENT (SRGANFD_EXPERIMENT builds only): conv1 -> conv2 of a dense block in one launch.
//
// BSRGAN/model.py:54-57: y1 = lrelu(conv1(x)), y2 = lrelu(conv2(cat(x, y1))) with x = 64 channels, y1 = y2 = 32.  One workgroup computes
// an 8 x 32 tile of y2: it stages the 12 x 36 patch of x once (both 32-channel chunks), computes y1 on the 10 x 34 halo'd tile into LDS
// (and stores its central 8 x 32 to memory: the later convs and the backward pass need y1), then runs conv2 over the x patch it already
// holds plus the y1 tile.  HBM traffic per pixel of the pair: 64 (x) + 32 + 32 written, instead of 64 + 32 and 96 + 32 of the two launches
// (-43 %), for +41 % conv1 MACs (10 x 36 positions computed per 8 x 32 wanted) and +12.5 % conv2 MACs (36-wide rows).
//
// Layout: the pixel tiles are FLATTENED with the x patch's row pitch (36): output position q of a conv reads, for tap (ky, kx), the 16
// consecutive staged pixels starting at q + ky * 36 + kx (+ 37 when conv2 reads the x patch: one halo ring in), so an A fragment of the
// 16x16x32 MFMA is one ds_read_b128 at lane term + wave-uniform offset + immediate.  LDS pixels are 80 bytes apart (64 of data): the 16
// lanes of a read group then hit 16 distinct 16-byte slots of the 256-byte bank row without any swizzle.  Weights (the product's packed
// B-fragment order) are streamed one 32-channel chunk at a time with a register prefetch; five chunks per tile (2 of conv1, 3 of conv2).
// Accumulation order per output element = the product kernel's (chunk, kernel column, kernel row), so y1 and y2 must equal the two
// launches' outputs bit for bit (tools/r3/pair_bench.py checks that).  118 KB of LDS: one workgroup per CU.
#include "conv_common.hpp"
#include <type_traits>

namespace srganfd {

namespace pair {
constexpr int TH = 8, TW = 32, PITCH = 36;
constexpr int XROWS = TH + 4, Y1ROWS = TH + 2;
constexpr int XPIX = 448;                       // 12 * 36 = 432 staged pixels + read-ahead of the junk columns
constexpr int Y1PIX = 368;                      // 23 groups of 16 positions >= 10 * 36
constexpr int PB = 80;                          // LDS bytes per pixel (64 of data)
constexpr int XBYTES = XPIX * PB;               // one 32-channel chunk of the x patch
constexpr int Y1BYTES = Y1PIX * PB;
constexpr int WBYTES = 9 * 2 * 64 * 16;         // one chunk's weight slab (32 output channels)
constexpr int LDS = 2 * XBYTES + Y1BYTES + WBYTES;
constexpr int G1 = 23, G2 = 18;                 // position groups of conv1 (10 x 36) and conv2 (8 x 36)
constexpr int NTHR = 512;
}

struct PairK {
  const void* buf; const void* w1; const void* w2; const float* b1; const float* b2;
  int N, H, W; float slope;
};

template <typename T>
__global__ __launch_bounds__(pair::NTHR, 2) void conv_pair_kernel(const PairK a) {
  using namespace pair;
  using Frag = typename FragAB<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsX = smem;
  char* ldsY1 = smem + 2 * XBYTES;
  char* ldsW = ldsY1 + Y1BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, sl = lane >> 4;
  const int tiles_x = a.W / TW, tiles_y = a.H / TH;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int oy0 = ty * TH, ox0 = tx * TW;
  const size_t plane = (size_t)a.H * a.W * 32;                 // one 32-channel group of one image
  const T* img = (const T*)a.buf + (size_t)n * 6 * plane;      // 192-channel planar buffer
  T* imgw = (T*)a.buf + (size_t)n * 6 * plane;

  // ---- stage the x patch (both chunks) and the first weight chunk ----
  {
    constexpr int ITEMS = 2 * XPIX * 4, PER = (ITEMS + NTHR - 1) / NTHR;
    u32x4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int item = tid + i * NTHR;
      const int ch = item / (XPIX * 4), rem = item % (XPIX * 4), p = rem >> 2, s = rem & 3;
      const int r = p / PITCH, c = p % PITCH;
      const int gy = oy0 - 2 + r, gx = ox0 - 2 + c;
      v[i] = u32x4{0u, 0u, 0u, 0u};
      if (item < ITEMS && p < XROWS * PITCH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
        v[i] = *(const u32x4*)(img + ch * plane + ((size_t)gy * a.W + gx) * 32 + s * 8);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int item = tid + i * NTHR;
      const int ch = item / (XPIX * 4), rem = item % (XPIX * 4), p = rem >> 2, s = rem & 3;
      if (item < ITEMS) *(u32x4*)(ldsX + ch * XBYTES + p * PB + s * 16) = v[i];
    }
  }
  constexpr int WI = (WBYTES / 16 + NTHR - 1) / NTHR;            // 3 (the third partly)
  u32x4 wr[WI];
  auto wsrc = [&](int c) -> const u32x4* { return c < 2 ? (const u32x4*)a.w1 + c * (WBYTES / 16) : (const u32x4*)a.w2 + (c - 2) * (WBYTES / 16); };
  auto wload = [&](int c) {
    const u32x4* src = wsrc(c);
#pragma unroll
    for (int i = 0; i < WI; ++i) { const int item = tid + i * NTHR; wr[i] = item < WBYTES / 16 ? src[item] : u32x4{0u, 0u, 0u, 0u}; }
  };
  auto wcommit = [&]() {
#pragma unroll
    for (int i = 0; i < WI; ++i) { const int item = tid + i * NTHR; if (item < WBYTES / 16) *(u32x4*)(ldsW + item * 16) = wr[i]; }
  };
  wload(0);
  wcommit();
  __syncthreads();

  const int lane_a = l15 * PB + sl * 16;               // A-fragment lane term: pixel l15 of the run, 16-byte slot sl
  const char* bq0 = ldsW + lane * 16;
  f32x4_t acc[3][2];
  auto zero = [&]() {
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) acc[g][nh] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  // one 32-channel chunk: A from `src` (an x-patch chunk or the y1 tile) at flattened offset `shift`, NG groups of this wave (wave, wave + 8, ..)
  auto chunk_body = [&](const char* src, int shift, auto ngc) __attribute__((always_inline)) {
    constexpr int NG = decltype(ngc)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      Frag av[NG][3];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) av[g][ky] = *(const Frag*)(src + lane_a + (16 * (wave + 8 * g) + shift + ky * PITCH + kx) * PB);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          const Frag bq = *(const Frag*)(bq0 + ((ky * 3 + kx) * 2 + nh) * 1024);
#pragma unroll
          for (int g = 0; g < NG; ++g) acc[g][nh] = mfma16<T>(av[g][ky], bq, acc[g][nh]);
        }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto chunk_mfma = [&](const char* src, int shift, int ng) __attribute__((always_inline)) {
    if (wave + 16 < ng) chunk_body(src, shift, std::integral_constant<int, 3>{});      // wave-uniform
    else chunk_body(src, shift, std::integral_constant<int, 2>{});
  };
  // next weight chunk: loads before the MFMA phase, commit after every wave has left it
  auto next_w = [&](int c) {
    __syncthreads();
    wcommit();
    __syncthreads();
    (void)c;
  };

  // ---- conv1 on the 10 x 36 positions ----
  zero();
  wload(1);
  chunk_mfma(ldsX, 0, G1);
  next_w(1);
  wload(2);
  chunk_mfma(ldsX + XBYTES, 0, G1);
  // epilogue of conv1: bias, LeakyReLU, round to T, y1 tile in LDS (zero outside the image: conv2's padding)
  {
    const float b_lo = a.b1 ? a.b1[l15] : 0.f, b_hi = a.b1 ? a.b1[16 + l15] : 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int gi = wave + 8 * g;
      if (gi >= G1) continue;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = 16 * gi + 4 * sl + i;
          const int r = q / PITCH, c = q % PITCH;
          const int gy = oy0 - 1 + r, gx = ox0 - 1 + c;
          float v = acc[g][nh][i] + (nh ? b_hi : b_lo);
          v = v * (v > 0.f ? 1.f : a.slope);
          const bool inside = c < TW + 2 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
          *(T*)(ldsY1 + q * PB + (16 * nh + l15) * 2) = inside ? Elem<T>::from_f(v) : Elem<T>::from_f(0.f);
        }
    }
  }
  next_w(2);                                      // (its first barrier also publishes the y1 tile)
  // y1's central 8 x 32 pixels -> channels [64, 96) of the buffer
  {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int item = tid + i * NTHR;            // 256 pixels x 4 slots
      const int p = item >> 2, s = item & 3, r = p >> 5, c = p & 31;
      const u32x4 v = *(const u32x4*)(ldsY1 + ((r + 1) * PITCH + c + 1) * PB + s * 16);
      *(u32x4*)(imgw + 2 * plane + ((size_t)(oy0 + r) * a.W + ox0 + c) * 32 + s * 8) = v;
    }
  }
  // ---- conv2 on the 8 x 36 positions: x chunk 0, x chunk 1 (one halo ring in: +37), y1 ----
  zero();
  wload(3);
  chunk_mfma(ldsX, PITCH + 1, G2);
  next_w(3);
  wload(4);
  chunk_mfma(ldsX + XBYTES, PITCH + 1, G2);
  next_w(4);
  chunk_mfma(ldsY1, 0, G2);
  __syncthreads();                                // every wave has read the y1 tile: it becomes the output tile
  {
    const float b_lo = a.b2 ? a.b2[l15] : 0.f, b_hi = a.b2 ? a.b2[16 + l15] : 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int gi = wave + 8 * g;
      if (gi >= G2) continue;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = 16 * gi + 4 * sl + i;
          float v = acc[g][nh][i] + (nh ? b_hi : b_lo);
          v = v * (v > 0.f ? 1.f : a.slope);
          *(T*)(ldsY1 + q * PB + (16 * nh + l15) * 2) = Elem<T>::from_f(v);
        }
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int item = tid + i * NTHR;
    const int p = item >> 2, s = item & 3, r = p >> 5, c = p & 31;
    const u32x4 v = *(const u32x4*)(ldsY1 + (r * PITCH + c) * PB + s * 16);
    *(u32x4*)(imgw + 3 * plane + ((size_t)(oy0 + r) * a.W + ox0 + c) * 32 + s * 8) = v;
  }
}

}  // namespace srganfd

extern "C" int srganfd_exp_conv_pair(void* buf, int32_t n, int32_t h, int32_t w, const void* w1_packed, const void* w2_packed, const float* b1,
                                     const float* b2, float slope, int32_t dtype, void* stream) {
  using namespace srganfd;
  if (!buf || !w1_packed || !w2_packed || n <= 0 || h % pair::TH || w % pair::TW) return set_err(SRGANFD_EINVAL, "exp_conv_pair: bad args");
  PairK k{buf, w1_packed, w2_packed, b1, b2, n, h, w, slope};
  const unsigned grid = (unsigned)(n * (h / pair::TH) * (w / pair::TW));
  static bool attr = false;
  if (dtype == SRGANFD_F16) {
    if (!attr) { SRGANFD_HIP_CHECK(hipFuncSetAttribute((const void*)conv_pair_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, pair::LDS)); attr = true; }
    SRGANFD_LAUNCH(conv_pair_kernel<f16_t>, dim3(grid), dim3(pair::NTHR), pair::LDS, (hipStream_t)stream, k);
  } else {
    return set_err(SRGANFD_EINVAL, "exp_conv_pair: f16 only");
  }
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
