// EXPERIMENT (round 5): cycles per v_mfma_f32_16x16x32_f16 in a register-only stream, one wave per SIMD, by accumulator placement:
//   0: the builtin (the compiler renames accumulators: vDst != SrcC)    1: inline asm, accumulate in place in VGPRs (vDst == SrcC)
//   2: inline asm, accumulate in place in AGPRs                         3: variant 1 with 16 independent accumulators (dependency distance 16)
//   4: 32x32x16 in place in AGPRs (12 per step)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_issue_probe tools/probes/mfma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int V>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 A[6], B[6];
  for (int q = 0; q < 6; ++q) for (int j = 0; j < 8; ++j) { A[q][j] = (_Float16)(0.01f * ((lane * 7 + q * 3 + j) & 31)); B[q][j] = (_Float16)(0.02f * ((lane * 5 + q + j) & 15)); }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x16 big[2] = {};
  unsigned long long t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if (V == 0) {
#pragma unroll
      for (int k = 0; k < 24; ++k) acc[k & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[k % 6], B[(k / 4) % 6], acc[k & 7], 0, 0, 0);
    } else if (V == 1) {
#pragma unroll
      for (int k = 0; k < 24; ++k) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[k & 7]) : "v"(A[k % 6]), "v"(B[(k / 4) % 6]));
    } else if (V == 2) {
#pragma unroll
      for (int k = 0; k < 24; ++k) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[k & 7]) : "v"(A[k % 6]), "v"(B[(k / 4) % 6]));
    } else if (V == 3) {
#pragma unroll
      for (int k = 0; k < 24; ++k) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[k & 15]) : "v"(A[k % 6]), "v"(B[(k / 4) % 6]));
    } else {
#pragma unroll
      for (int k = 0; k < 12; ++k) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(big[k & 1]) : "v"(A[k % 6]), "v"(B[(k / 2) % 6]));
    }
  }
  unsigned long long t1;
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  s += big[0][0] + big[1][5];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V> void run(float* out, unsigned long long* cyc, int grid) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<V>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<V>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double n = V == 4 ? 12.0 : 24.0;
  printf("variant %d, %3d workgroups: %6.2f cycles / MFMA (s_memtime), %7.1f ns / 24-MFMA-equivalent step wall -> %.2f GHz, %.0f TFLOP/s\n", V, grid, (double)h / iters / n,
         ms * 1e6 / iters, (double)h / (ms * 1e6), grid * 4.0 * 24 * 16384.0 * iters / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  for (int grid : {1, 256}) { run<0>(out, cyc, grid); run<1>(out, cyc, grid); run<2>(out, cyc, grid); run<3>(out, cyc, grid); run<4>(out, cyc, grid); }
  return 0;
}
