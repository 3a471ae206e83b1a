// bw_probe.hip -- practical HBM streaming rates on this box (the roofline the conv kernels are priced against).
// hipcc --offload-arch=gfx950 -O3 tools/probes/bw_probe.hip -o tools/probes/bw_probe && tools/probes/bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int NT>   // 0 read-only (sum), 1 copy, 2 write-only;  NT: nontemporal accesses
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n, unsigned* sink) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * 4) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t j = i + u * stride;
      if (MODE != 2 && j < n) v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j]; else v[u] = acc;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t j = i + u * stride;
      if (MODE == 0) { acc[0] ^= v[u][0]; acc[1] += v[u][1]; acc[2] ^= v[u][2]; acc[3] += v[u][3]; }
      else if (j < n) { if (NT) __builtin_nontemporal_store(v[u], dst + j); else dst[j] = v[u]; }
    }
  }
  if (MODE == 0 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}

template <int MODE, int NT>
static void run(const char* name, const u32x4* a, u32x4* b, size_t n, unsigned* sink, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream_kernel<MODE, NT>), dim3(blocks), dim3(256), 0, 0, a, b, n, sink);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((stream_kernel<MODE, NT>), dim3(blocks), dim3(256), 0, 0, a, b, n, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)n * 16 * (MODE == 1 ? 2 : 1) * reps;
  printf("%-34s blocks %5d: %7.3f TB/s (%s bytes)\n", name, blocks, bytes / (ms * 1e-3) / 1e12, MODE == 1 ? "read + written" : (MODE == 0 ? "read" : "written"));
}

int main() {
  const size_t bytes = (size_t)2 << 30, n = bytes / 16;   // 2 GiB per buffer: far beyond the 256 MiB Infinity Cache
  u32x4 *a, *b; unsigned* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  for (int blocks : {1024, 2048, 4096, 8192}) {
    run<0, 0>("read-only", a, b, n, sink, blocks);
    run<0, 1>("read-only, nontemporal", a, b, n, sink, blocks);
    run<1, 0>("copy", a, b, n, sink, blocks);
    run<1, 1>("copy, nontemporal", a, b, n, sink, blocks);
    run<2, 0>("write-only", a, b, n, sink, blocks);
    run<2, 1>("write-only, nontemporal", a, b, n, sink, blocks);
  }
  // 200 MB working set (one dense-block buffer): what a conv re-reading its predecessor's output can get from the Infinity Cache
  const size_t n2 = ((size_t)200 << 20) / 16;
  run<0, 0>("read-only, 200 MiB (MALL-resident)", a, b, n2, sink, 2048);
  run<1, 0>("copy, 100 -> 100 MiB", a, b, n2 / 2, sink, 2048);
  return 0;
}
