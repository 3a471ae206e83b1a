// EXPERIMENT (round 5): is hipExtStreamCreateWithCUMask honoured here, and how do its bits map to XCDs / CUs?
//   part 1: per mask, a launch of 4096 short-spinning workgroups on the masked stream records (XCC_ID, HW_ID) -> distinct CUs used per XCD
//   part 2: two launches of 128 workgroups x 100 KB of LDS x ~300 us, one on a masked stream (half the CUs) and one on an unmasked stream:
//           alone / together wall time (together ~= alone when the two really run side by side)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/cu_mask_probe tools/probes/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <set>
#include <map>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where_kernel(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

__global__ void hog_kernel(unsigned* out, int spin) {
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) out[blockIdx.x] = lds[(blockIdx.x * 7) & 63];
}

static int histogram(const char* name, hipStream_t st, unsigned* d, std::vector<unsigned>& h) {
  const int n = 4096;
  hipLaunchKernelGGL(where_kernel, dim3(n), dim3(64), 0, st, d, 20000);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * n, hipMemcpyDeviceToHost));
  std::map<unsigned, std::set<unsigned>> cus;
  for (int i = 0; i < n; ++i) cus[h[2 * i + 1] & 15].insert((h[2 * i] >> 8) & 0xff);   // CU_ID 11:8, SH_ID 12, SE_ID 15:13
  int total = 0;
  printf("%-34s CUs used per XCD:", name);
  for (auto& kv : cus) { printf(" %u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
  printf("  total %d\n", total);
  return 0;
}

int main() {
  unsigned* d = nullptr;
  CK(hipMalloc(&d, sizeof(unsigned) * 2 * 4096));
  std::vector<unsigned> h(2 * 4096);
  hipStream_t plain;
  CK(hipStreamCreate(&plain));
  if (histogram("no mask", plain, d, h)) return 1;
  struct { const char* name; uint32_t w[8]; } masks[] = {
    {"bits 0..127", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}},
    {"bits 0..31", {~0u, 0, 0, 0, 0, 0, 0, 0}},
    {"even bits (0x55555555 x 8)", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
    {"low half of every word (0xffff x 8)", {0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu}},
    {"bits 0..63", {~0u, ~0u, 0, 0, 0, 0, 0, 0}},
    {"bits 128..255", {0, 0, 0, 0, ~0u, ~0u, ~0u, ~0u}},
  };
  hipStream_t half = nullptr;
  for (auto& m : masks) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, m.w);
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask -> %s\n", m.name, hipGetErrorString(e)); continue; }
    if (histogram(m.name, st, d, h)) return 1;
    if (!half) half = st;
  }
  if (!half) return 0;
  // part 2: side by side?
  CK(hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  auto run = [&](bool a, bool b) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 10; ++r) {
      if (a) hipLaunchKernelGGL(hog_kernel, dim3(128), dim3(256), 100 * 1024, half, d, 600000);
      if (b) hipLaunchKernelGGL(hog_kernel, dim3(128), dim3(256), 100 * 1024, plain, d + 1024, 600000);
    }
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10;
  };
  run(true, true);
  printf("128 workgroups x 100 KB LDS x 600k cycles, per round: masked stream alone %.1f us, plain stream alone %.1f us, both %.1f us\n",
         run(true, false), run(false, true), run(true, true));
  // ... and with 256 workgroups on the plain stream (needs every CU: must wait for the masked half)
  auto run2 = [&]() {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 10; ++r) {
      hipLaunchKernelGGL(hog_kernel, dim3(128), dim3(256), 100 * 1024, half, d, 600000);
      hipLaunchKernelGGL(hog_kernel, dim3(256), dim3(256), 100 * 1024, plain, d + 1024, 600000);
    }
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10;
  };
  printf("masked 128 + plain 256 workgroups: %.1f us per round\n", run2());
  return 0;
}
