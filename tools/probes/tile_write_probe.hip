// tile_write_probe.hip -- write rate of the conv epilogue's store pattern: a (TH x TW)-pixel tile of an NHWC image (C channels of 2 bytes)
// per 512-thread block, 16 bytes per lane, consecutive lanes = consecutive 16-byte chunks of a pixel, then the next pixel of the tile row.
// hipcc --offload-arch=gfx950 -O3 tools/probes/tile_write_probe.hip -o /tmp/twp && /tmp/twp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int TH, int TW, int C>
__global__ __launch_bounds__(512) void tile_write(u32x4* __restrict__ dst, int n, int h, int w, int persistent) {
  constexpr int CP = C * 2 / 16;                       // 16-byte chunks per pixel
  const int tiles_x = w / TW, tiles_y = h / TH, ntiles = n * tiles_x * tiles_y;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, img = t / (tiles_x * tiles_y);
    for (int item = threadIdx.x; item < TH * TW * CP; item += 512) {
      const int pix = item / CP, ck = item % CP;
      const int oy = ty * TH + pix / TW, ox = tx * TW + pix % TW;
      const u32x4 v = {(unsigned)item, (unsigned)t, 3u, 4u};
      dst[((size_t)(img * h + oy) * w + ox) * CP + ck] = v;
    }
    if (!persistent) break;
  }
}

template <int TH, int TW, int C>
static void run(u32x4* d, int n, int h, int w, int persistent) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int ntiles = n * (h / TH) * (w / TW), grid = persistent ? 512 : ntiles;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((tile_write<TH, TW, C>), dim3(grid), dim3(512), 0, 0, d, n, h, w, persistent);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((tile_write<TH, TW, C>), dim3(grid), dim3(512), 0, 0, d, n, h, w, persistent);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)n * h * w * C * 2 * reps;
  printf("C=%3d tile %2d x %3d px (%5d B runs) %s: %6.3f TB/s\n", C, TH, TW, TW * C * 2, persistent ? "512 persistent blocks" : "one block per tile   ", bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const int n = 32, h = 512, w = 512;                    // 64 channels: 1 GiB (far beyond the Infinity Cache)
  u32x4* d; CK(hipMalloc(&d, (size_t)n * h * w * 128 * 2));
  for (int p = 0; p < 2; ++p) {
    run<8, 32, 64>(d, n, h, w, p); run<4, 64, 64>(d, n, h, w, p); run<2, 128, 64>(d, n, h, w, p); run<1, 256, 64>(d, n, h, w, p);
    run<16, 32, 64>(d, n, h, w, p); run<16, 32, 32>(d, n, h, w, p); run<8, 64, 32>(d, n, h, w, p); run<8, 32, 128>(d, n, h, w, p);
  }
  return 0;
}
