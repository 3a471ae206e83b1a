// Probe: does hipExtLaunchKernel's hipExtAnyOrderLaunch flag (hip_runtime_api.h:949; "launch in any order with respect to prior kernels of the stream" = no
// barrier bit on the dispatch packet) do anything on gfx950?  hip_ext.h says it is not supported on GFX9xx for the module-launch form.
//   A (ordered, spins T us on one workgroup) ; B (any-order or ordered, spins T us) ; C (ordered, writes a marker after reading A's and B's)
// time(A;B) ~ T: B ran beside A; ~ 2T: the flag is ignored.  C must see both results either way.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/any_order_probe tools/probes/any_order_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(long long ticks, int* out, int val) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) *out = val;
}
__global__ void join(const int* a, const int* b, int* out) { if (threadIdx.x == 0) *out = *a + *b; }

int main() {
  int* d; CK(hipMalloc(&d, 3 * sizeof(int)));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));   // kHz
  const double us = 200.0;
  long long ticks = (long long)(us * rate / 1000.0);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemsetAsync(d, 0, 3 * sizeof(int), s));
      int va = 1, vb = 2; int* pa = d; int* pb = d + 1; int* pc = d + 2;
      void* argsA[] = {&ticks, &pa, &va};
      void* argsB[] = {&ticks, &pb, &vb};
      void* argsC[] = {&pa, &pb, &pc};
      CK(hipEventRecord(e0, s));
      CK(hipExtLaunchKernel((const void*)spin, dim3(1), dim3(64), argsA, 0, s, nullptr, nullptr, 0));
      CK(hipExtLaunchKernel((const void*)spin, dim3(1), dim3(64), argsB, 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0));
      CK(hipExtLaunchKernel((const void*)join, dim3(1), dim3(64), argsC, 0, s, nullptr, nullptr, 0));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      int h[3]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
      printf("B %s: A;B;C took %.1f us (each spin %.0f us), C saw %d (want 3)\n", mode ? "any-order" : "ordered  ", ms * 1e3, us, h[2]);
    }
  }
  return 0;
}
