// EXPERIMENT (round 5): what does one "kernel-column step" of the LDS-resident dense-block launch cost in isolation?
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_step_probe tools/probes/mfma_step_probe.hip && build/mfma_step_probe
// Variants, one wave per SIMD (256-thread workgroups, one per CU), 2000 iterations each:
//   0: 24 independent-chain v_mfma_f32_16x16x32_f16 (8 accumulators x 3), operands in registers
//   1: + 12 ds_read_b128 per iteration feeding the NEXT iteration's operands (two register sets)
//   2: + one s_barrier per iteration
//   3: variant 2 with v_mfma_f32_32x32x16_f16 x 12 (same FLOPs)
// Prints s_memtime cycles per iteration (wave 0 of workgroup 0) and wall-clock ns per iteration (hipEvents).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int V>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[49152];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 49152 / 4; i += 256) ((float*)lds)[i] = 0.001f * (i & 255);
  __syncthreads();
  f16x8 A[6], B[6], A2[6], B2[6];
  for (int q = 0; q < 6; ++q) { A[q] = *(f16x8*)(lds + q * 1024 + lane * 16); B[q] = *(f16x8*)(lds + 8192 + q * 1024 + lane * 16); A2[q] = A[q]; B2[q] = B[q]; }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x16 acc32[2] = {};
  unsigned long long t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll 1
  for (int it = 0; it < iters; it += 2) {
#define STEP(CA, CB, NA, NB, OFF)                                                                                         \
    {                                                                                                                     \
      if (V >= 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                         \
      if (V >= 1) {                                                                                                       \
        const char* p = lds + (OFF) + wave * 12288 + lane * 16;                                                           \
        _Pragma("unroll") for (int q = 0; q < 6; ++q) { NA[q] = *(const f16x8*)(p + q * 1024); NB[q] = *(const f16x8*)(p + 6144 + q * 1024); } \
      }                                                                                                                   \
      if (V == 3) {                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < 6; ++k) { acc32[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CA[k], CB[k], acc32[0], 0, 0, 0); \
                                                        acc32[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CA[k], CB[5 - k], acc32[1], 0, 0, 0); } \
      } else {                                                                                                            \
        _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)                                                                  \
          _Pragma("unroll") for (int nh = 0; nh < 2; ++nh)                                                                \
            _Pragma("unroll") for (int m = 0; m < 4; ++m) acc[m * 2 + nh] = __builtin_amdgcn_mfma_f32_16x16x32_f16(CA[ky * 2 + nh], CB[m + ky], acc[m * 2 + nh], 0, 0, 0); \
      }                                                                                                                   \
      if (V >= 1) {                                                                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, V == 3 ? 1 : 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } \
      }                                                                                                                   \
    }
    STEP(A, B, A2, B2, 0)
    STEP(A2, B2, A, B, 16)
  }
  unsigned long long t1;
  asm volatile("s_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(acc[0]), "v"(acc[7]), "v"(acc32[0]), "v"(acc32[1]) : "memory");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  s += acc32[0][0] + acc32[1][5];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V> void run(float* out, unsigned long long* cyc, int grid) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<V>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<V>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("variant %d, %3d workgroups: %7.1f cycles / step (s_memtime, workgroup 0), %7.1f ns / step wall  -> %.2f GHz; %.0f TFLOP/s\n", V, grid, (double)h[0] / iters,
         ms * 1e6 / iters, (double)h[0] / (ms * 1e6), grid * 4.0 * 24 * 16384.0 * iters / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  for (int grid : {1, 256}) { run<0>(out, cyc, grid); run<1>(out, cyc, grid); run<2>(out, cyc, grid); run<3>(out, cyc, grid); }
  return 0;
}
