// Probe of gfx950 LDS-DMA semantics used by the kernels: destination = wave-uniform base + lane*16, inactive lanes
// write nothing, completion is tracked by vmcnt.  Build: hipcc --offload-arch=gfx950 -O3 glds_probe.hip -o glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__global__ __launch_bounds__(256) void probe(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int skip_mod) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // fill with a sentinel
  for (int i = tid; i < 4096; i += 256) ((unsigned*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  // each wave copies 2 pieces of 1 KiB; lane l fetches source chunk (l ^ 3) -> permuted source, linear destination
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = wave * 2 + i;
    char* ldsbase = smem + __builtin_amdgcn_readfirstlane(piece * 1024);
    const unsigned* g = src + (size_t)piece * 256 + (lane ^ 3) * 4;
    if (skip_mod == 0 || (lane % skip_mod) != 0)
      __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)ldsbase, 16, 0, 0);
  }
  __syncthreads();   // hipcc drains vmcnt(0) before the barrier
  for (int i = tid; i < 2048; i += 256) dst[i] = ((unsigned*)smem)[i];
}

int main() {
  std::vector<unsigned> h(2048), out(2048);
  for (int i = 0; i < 2048; ++i) h[i] = i;
  unsigned *d_src, *d_dst;
  hipMalloc(&d_src, 8192); hipMalloc(&d_dst, 8192);
  hipMemcpy(d_src, h.data(), 8192, hipMemcpyHostToDevice);
  int bad = 0;
  for (int skip = 0; skip <= 4; skip += 4) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 16384, 0, d_src, d_dst, skip);
    hipMemcpy(out.data(), d_dst, 8192, hipMemcpyDeviceToHost);
    for (int piece = 0; piece < 8; ++piece)
      for (int l = 0; l < 64; ++l)
        for (int q = 0; q < 4; ++q) {
          unsigned want = piece * 256 + (l ^ 3) * 4 + q;
          if (skip && (l % skip) == 0) want = 0xdeadbeefu;
          if (out[piece * 256 + l * 4 + q] != want) { if (bad < 5) printf("skip=%d piece %d lane %d q %d: got %u want %u\n", skip, piece, l, q, out[piece * 256 + l * 4 + q], want); ++bad; }
        }
  }
  printf("glds probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
