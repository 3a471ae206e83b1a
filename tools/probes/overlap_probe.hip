// overlap_probe.hip -- does a staged-tile MFMA loop overlap its own global->LDS staging on gfx950, and at what depth?
//
// One 8-wave workgroup per CU (or two), every iteration ("chunk") each wave
//   (a) issues PIECES 1-KiB staging transfers of fresh global data (never re-read: HBM-side stream) into an LDS ring slot, either as
//       LDS-DMA (global_load_lds_dwordx4, no registers) or as global_load_dwordx4 -> registers -> ds_write_b128 after the MFMA phase,
//   (b) runs the fragment-read + MFMA body of one 32-channel chunk of the 3x3 convolution kernels (72 v_mfma_f32_16x16x32_f16 on
//       42 ds_read_b128 from a resident LDS image),
//   (c) waits until all but the youngest (DEPTH-1) chunks of its transfers have landed (counted vmcnt) and joins a raw s_barrier.
// Timed with HIP events per mode: copy only, MFMA only, both.  If "both" ~ max(copy, mfma) the hardware overlaps them and the
// convolution kernels' additive behaviour is a property of their structure; if "both" ~ copy + mfma no restructuring will help.
//   hipcc --offload-arch=gfx950 -O3 overlap_probe.hip -o overlap_probe && ./overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst_uniform) : "m0");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

constexpr int kFragBytes = 32 * 1024;     // resident image the fragment reads walk

// MODE bit 0: staging transfers, bit 1: fragment reads + MFMA, bit 2: register staging instead of LDS-DMA
// WPS = waves per SIMD the register allocator must leave room for (2: one workgroup per CU, 4: two); ring_bytes: the landing zone
// (nobody reads it, so slots may alias when LDS is short)
template <int MODE, int PIECES, int DEPTH, int WPS, int NW = 8>
__global__ __launch_bounds__(64 * NW, WPS) void probe(const char* __restrict__ src, float* __restrict__ out, int iters, size_t wg_stride, int ring_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // resident fragment image: deterministic non-trivial f16 data
  for (int i = tid; i < kFragBytes / 4; i += 64 * NW) ((unsigned*)smem)[i] = 0x3c003800u ^ ((unsigned)(i * 2654435761u) & 0x03ff03ffu);
  __syncthreads();
  char* ring = smem + kFragBytes;                       // NW waves x DEPTH slots
  const unsigned ring_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
  const char* g = src + (size_t)blockIdx.x * wg_stride + (size_t)wave * PIECES * 1024 + lane * 16;
  const size_t chunk_stride = (size_t)NW * PIECES * 1024;
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* fa = smem + lane * 16;                    // A-like reads: lane-linear 1 KiB fragments (conflict-free)
  u32x4 regs[(MODE & 4) ? PIECES : 1];
  auto issue = [&](int c) {
    if constexpr ((MODE & 1) && !(MODE & 4)) {
      const unsigned slot = ring_base + (unsigned)(((wave * DEPTH + (c % DEPTH)) * PIECES * 1024) % ring_bytes);
#pragma unroll
      for (int p = 0; p < PIECES; ++p) glds16(g + (size_t)c * chunk_stride + p * 1024, slot + p * 1024);
    }
    if constexpr ((MODE & 1) && (MODE & 4)) {
#pragma unroll
      for (int p = 0; p < PIECES; ++p) regs[p] = *(const u32x4*)(g + (size_t)c * chunk_stride + p * 1024);
    }
  };
  if constexpr (MODE & 1) {
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d);
  }
  for (int c = 0; c < iters; ++c) {
    if constexpr (MODE & 1) issue(c + DEPTH - 1);      // (reads past `iters` chunks stay inside the buffer: the host sizes it)
    if constexpr (MODE & 2) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        f16x8 a[8], b[6];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *(const f16x8*)(fa + ((kx * 14 + i) % 32) * 1024);
#pragma unroll
        for (int i = 0; i < 6; ++i) b[i] = *(const f16x8*)(fa + ((kx * 14 + 8 + i) % 32) * 1024);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[(t << 1) | nh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(t >> 1) * 2 + ky * 2 % 6 + (t & 1)], b[ky * 2 + nh], acc[(t << 1) | nh], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    if constexpr ((MODE & 1) && !(MODE & 4)) wait_vmcnt<(DEPTH - 1) * PIECES>();
    if constexpr ((MODE & 1) && (MODE & 4)) {
      char* slot = ring + ((wave * DEPTH + (c % DEPTH)) * PIECES * 1024) % ring_bytes + lane * 16;
#pragma unroll
      for (int p = 0; p < PIECES; ++p) *(u32x4*)(slot + p * 1024) = regs[p];
    }
    __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if constexpr (MODE & 1) { wait_vmcnt<0>(); __syncthreads(); s += ((float*)ring)[tid]; }
  if (s == 12345.678f) out[(blockIdx.x & 511) * 512 + (tid & 511)] = s;
}

template <int MODE, int PIECES, int DEPTH>
static double run16(const char* src, float* out, int iters) {
  auto k = probe<MODE, PIECES, DEPTH, 4, 16>;
  const int want = 16 * DEPTH * PIECES * 1024, cap = 158 * 1024 - kFragBytes;
  const int ring_bytes = (want < cap ? want : cap) / (PIECES * 1024) * (PIECES * 1024);
  const int lds = kFragBytes + ring_bytes;
  const size_t wg_stride = (size_t)(iters + DEPTH) * 16 * PIECES * 1024;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int r = 0; r < 6; ++r) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), lds, 0, src, out, iters, wg_stride, ring_bytes);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2] * 1e3;
}
template <int PIECES, int DEPTH>
static void sweep16(const char* src, float* out, int iters) {
  const double gb = 256.0 * iters * 16 * PIECES * 1024 / 1e9, tf = 256.0 * iters * 16 * 72 * 2.0 * 16 * 16 * 32 / 1e12;
  const double c = run16<1, PIECES, DEPTH>(src, out, iters), m = run16<2, PIECES, DEPTH>(src, out, iters), b = run16<3, PIECES, DEPTH>(src, out, iters);
  printf("16 waves, each issues %d pieces (%3d KB per tile) then computes, depth %d: copy %7.1f us (%5.2f TB/s)  mfma %7.1f us (%6.1f TF)  both %7.1f us  -> both/max %.2f  both/sum %.2f\n",
         PIECES, 16 * PIECES, DEPTH, c, gb / c * 1e3, m, tf / m * 1e6, b, b / std::max(c, m), b / (c + m));
}

// The weight-gradient kernel's structure: 12 compute waves (fragment reads + MFMA, never touch global memory) and 4 loader waves that
// issue the NEXT tile's LPIECES LDS-DMA pieces each right after the barrier, wait for all of them (vmcnt(0)) and join the barrier
// that publishes the buffer -- one tile of look-ahead, one barrier per tile.  MODE bit 0: loaders run, bit 1: compute waves run.
// PRIO: s_setprio of the loader waves.
template <int MODE, int LPIECES, int PRIO>
__global__ __launch_bounds__(1024) void probe_loaders(const char* __restrict__ src, float* __restrict__ out, int iters, size_t wg_stride, int ring_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < kFragBytes / 4; i += 1024) ((unsigned*)smem)[i] = 0x3c003800u ^ ((unsigned)(i * 2654435761u) & 0x03ff03ffu);
  __syncthreads();
  char* ring = smem + kFragBytes;
  const unsigned ring_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (wave >= 12) {
    const int lw = wave - 12;
    const char* g = src + (size_t)blockIdx.x * wg_stride + (size_t)lw * LPIECES * 1024 + lane * 16;
    const size_t chunk_stride = (size_t)4 * LPIECES * 1024;
    if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
    auto fill = [&](int c) {
      if constexpr (MODE & 1) {
        const unsigned slot = ring_base + (unsigned)((((c & 1) * 4 + lw) * LPIECES * 1024) % ring_bytes);
#pragma unroll
        for (int p = 0; p < LPIECES; ++p) glds16(g + (size_t)c * chunk_stride + p * 1024, slot + p * 1024);
      }
    };
    fill(0);
    for (int c = 0; c < iters; ++c) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      fill(c + 1);
    }
    wait_vmcnt<0>();
    return;
  }
  const char* fa = smem + lane * 16;
  for (int c = 0; c < iters; ++c) {
    __builtin_amdgcn_s_barrier();
    if constexpr (MODE & 2) {
      // 96 MFMAs on 80 transposed-size reads per tile and wave in the real kernel; here 72 + 42 b128 reads x 4/3 iterations ~ the same pipe time
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        f16x8 a[8], b[6];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *(const f16x8*)(fa + ((kx * 14 + i) % 32) * 1024);
#pragma unroll
        for (int i = 0; i < 6; ++i) b[i] = *(const f16x8*)(fa + ((kx * 14 + 8 + i) % 32) * 1024);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[(t << 1) | nh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(t >> 1) * 2 + ky * 2 % 6 + (t & 1)], b[ky * 2 + nh], acc[(t << 1) | nh], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[(blockIdx.x & 511) * 512 + (tid & 511)] = s;
}

template <int MODE, int LPIECES, int PRIO>
static double run_loaders(const char* src, float* out, int iters) {
  auto k = probe_loaders<MODE, LPIECES, PRIO>;
  const int want = 2 * 4 * LPIECES * 1024, cap = 158 * 1024 - kFragBytes;
  const int ring_bytes = (want < cap ? want : cap) / (LPIECES * 1024) * (LPIECES * 1024);     // (nobody reads it: slots may alias)
  const int lds = kFragBytes + ring_bytes;
  const size_t wg_stride = (size_t)(iters + 2) * 4 * LPIECES * 1024;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int r = 0; r < 6; ++r) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), lds, 0, src, out, iters, wg_stride, ring_bytes);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2] * 1e3;
}
template <int LPIECES, int PRIO>
static void sweep_loaders(const char* src, float* out, int iters) {
  const double gb = 256.0 * iters * 4 * LPIECES * 1024 / 1e9, tf = 256.0 * iters * 12 * 96 * 2.0 * 16 * 16 * 32 / 1e12;
  const double c = run_loaders<1, LPIECES, PRIO>(src, out, iters), m = run_loaders<2, LPIECES, PRIO>(src, out, iters), b = run_loaders<3, LPIECES, PRIO>(src, out, iters);
  printf("12 compute + 4 loader waves, %2d pieces per loader (%3d KB per tile), loader prio %d: copy %7.1f us (%5.2f TB/s)  mfma %7.1f us (%6.1f TF)  both %7.1f us  -> both/max %.2f  both/sum %.2f\n",
         LPIECES, 4 * LPIECES, PRIO, c, gb / c * 1e3, m, tf / m * 1e6, b, b / std::max(c, m), b / (c + m));
}

template <int MODE, int PIECES, int DEPTH, int WPS>
static double run(const char* src, float* out, int nwg, int iters, size_t wg_stride) {
  auto k = probe<MODE, PIECES, DEPTH, WPS>;
  const int want = 8 * DEPTH * PIECES * 1024, cap = (WPS == 4 ? 78 : 158) * 1024 - kFragBytes;
  const int ring_bytes = (want < cap ? want : cap) / (PIECES * 1024) * (PIECES * 1024);
  const int lds = kFragBytes + ring_bytes;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int r = 0; r < 6; ++r) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(nwg), dim3(512), lds, 0, src, out, iters, wg_stride, ring_bytes);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2] * 1e3;
}

template <int PIECES, int DEPTH, int WPS>
static void sweep(const char* src, float* out, int iters) {
  const int nwg = WPS == 4 ? 512 : 256;
  const size_t wg_stride = (size_t)(iters + DEPTH) * 8 * PIECES * 1024;
  const double gb = (double)nwg * iters * 8 * PIECES * 1024 / 1e9;
  const double tf = (double)nwg * iters * 8 * 72 * 2.0 * 16 * 16 * 32 / 1e12;
  const double c = run<1, PIECES, DEPTH, WPS>(src, out, nwg, iters, wg_stride), m = run<2, PIECES, DEPTH, WPS>(src, out, nwg, iters, wg_stride),
               b = run<3, PIECES, DEPTH, WPS>(src, out, nwg, iters, wg_stride);
  printf("wg %4d pieces %2d depth %d  LDS-DMA : copy %7.1f us (%5.2f TB/s)  mfma %7.1f us (%6.1f TF)  both %7.1f us  -> both/max %.2f  both/sum %.2f\n",
         nwg, PIECES, DEPTH, c, gb / c * 1e3, m, tf / m * 1e6, b, b / std::max(c, m), b / (c + m));
  if (DEPTH == 2) {
    const double cr = run<5, PIECES, DEPTH, WPS>(src, out, nwg, iters, wg_stride), br = run<7, PIECES, DEPTH, WPS>(src, out, nwg, iters, wg_stride);
    printf("wg %4d pieces %2d depth 1  register: copy %7.1f us (%5.2f TB/s)  mfma %7.1f us             both %7.1f us  -> both/max %.2f  both/sum %.2f\n",
           nwg, PIECES, cr, gb / cr * 1e3, m, br, br / std::max(cr, m), br / (cr + m));
  }
}

int main(int argc, char** argv) {
  const int iters = 48;
  const size_t bytes = (size_t)512 * (iters + 4) * 8 * 10 * 1024 + (1 << 20);
  char* src; float* out;
  CHECK(hipMalloc(&src, bytes)); CHECK(hipMalloc(&out, 512 * 512 * 4));
  CHECK(hipMemset(src, 0x3c, bytes));
  sweep16<5, 2>(src, out, iters);
  sweep16<4, 2>(src, out, iters);
  sweep16<3, 2>(src, out, iters);
  sweep_loaders<19, 3>(src, out, iters);
  sweep_loaders<19, 0>(src, out, iters);
  sweep_loaders<10, 3>(src, out, iters);
  sweep_loaders<10, 0>(src, out, iters);
  sweep<7, 2, 2>(src, out, iters);
  sweep<7, 3, 2>(src, out, iters);
  sweep<10, 2, 2>(src, out, iters);
  sweep<10, 3, 2>(src, out, iters);
  sweep<5, 4, 2>(src, out, iters);
  sweep<7, 2, 4>(src, out, iters);
  sweep<7, 3, 4>(src, out, iters);
  sweep<5, 2, 4>(src, out, iters);
  return 0;
}
