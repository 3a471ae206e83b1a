// EXPERIMENT: what v_permlane16_swap returns per lane row (gfx950).  a = 100 + lane, b = 200 + lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) unsigned int u2;
__global__ void k(unsigned* p) { const unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x; const u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false); p[threadIdx.x] = r.x; p[64 + threadIdx.x] = r.y; }
int main() { unsigned* d; hipMalloc(&d, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r) printf("row %d lane %2d: result.x = %u  result.y = %u\n", r, 16 * r, h[16 * r], h[64 + 16 * r]); return 0; }
