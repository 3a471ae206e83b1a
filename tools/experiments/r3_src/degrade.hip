// degrade.hip -- the device-resident stages of Real-ESRGAN's second-order degradation (SURVEY 8f N4;
// reference: Real_ESRGAN/imgproc.py:1092-1124 filter2d_torch, :1183-1497 DiffJPEG, :1517-1540 USMSharp, :2323-2462
// degradation_process).  All of them are per-image, HBM-bound fp32 NCHW passes: each kernel reads its input once into
// LDS tiles, does the whole stage on chip and writes the result once.
#include <math.h>
#include "common.hpp"

namespace srganfd {

// ---------------------------------------------------------------------------------------------------------------
// filter2d_torch (imgproc.py:1092-1124): reflect-pad by k/2, then cross-correlate every channel of image n with kernel
// n (or the one shared kernel).  A 256-thread block owns a 32-row x 64-column output tile: thread (tx, ty) -> column tx,
// rows 8*ty .. 8*ty+7, sliding an 8-deep register window down its LDS column so that one LDS read feeds 8 FMAs; the
// filter taps are wave-uniform (scalar loads).  The USM sharpener's two passes (imgproc.py:1529-1540) are epilogues:
//   mode 1: residual = x - blur -> out;  mask = (|residual| * 255 > threshold) -> out2
//   mode 2: soft = blur(mask);  out = soft * clip(x + weight * residual, 0, 1) + (1 - soft) * x
// ---------------------------------------------------------------------------------------------------------------
static constexpr int kF2dRows = 32, kF2dCols = 64, kF2dMaxK = 51;

__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return min(max(i, 0), n - 1);   // only positions no output reads get clamped
}

// SEP: the k x k filter is the outer product kcol (vertical taps) x krow (horizontal taps), handed over as kernels[0..k) and
// kernels[k..2k): a horizontal pass into a second LDS image, then the same sliding-window pass vertically -- 2k instead of
// k*k multiply-adds per pixel (the 51x51 Gaussian of the USM sharpener: 25x fewer).
template <bool SEP>
__global__ __launch_bounds__(256) void filter2d_kernel(const float* __restrict__ src, const float* __restrict__ kernels, int kernel_batch, int c, int h,
                                                       int w, int k, int tiles_x, int mode, const float* __restrict__ x_in,
                                                       const float* __restrict__ res_in, float weight, float threshold, float* __restrict__ out,
                                                       float* __restrict__ out2) {
  constexpr int kPitch = kF2dCols + kF2dMaxK - 1;               // 114
  constexpr int kTileRows = kF2dRows + kF2dMaxK - 1 + (SEP ? 0 : 8);   // the unrolled window may address (never use) 8 rows past the halo
  constexpr int kHRows = SEP ? kF2dRows + kF2dMaxK - 1 + 8 : 1;
  __shared__ float tile[kTileRows * kPitch];
  __shared__ float hbuf[kHRows * kF2dCols];
  const int plane = blockIdx.y, img = plane / c;
  const int ty_base = (blockIdx.x / tiles_x) * kF2dRows, tx_base = (blockIdx.x % tiles_x) * kF2dCols;
  const float* sp = src + (size_t)plane * h * w;
  const float* kw = kernels + (kernel_batch > 1 ? (size_t)img * (SEP ? 2 * k : k * k) : 0);
  const int r = k / 2, in_rows = kF2dRows + k - 1, in_cols = kF2dCols + k - 1;
  for (int i = threadIdx.x; i < in_rows * in_cols; i += 256) {
    const int iy = i / in_cols, ix = i % in_cols;
    tile[iy * kPitch + ix] = sp[(size_t)reflect_idx(ty_base + iy - r, h) * w + reflect_idx(tx_base + ix - r, w)];
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty0 = (threadIdx.x >> 6) * 8;
  float acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = 0.f;
  if (SEP) {
    const float* krow = kw + k;
    for (int row = threadIdx.x >> 6; row < in_rows; row += 4) {
      const float* tr = tile + row * kPitch + tx;
      float a = 0.f;
      for (int kx = 0; kx < k; ++kx) a = fmaf(krow[kx], tr[kx], a);
      hbuf[row * kF2dCols + tx] = a;
    }
    __syncthreads();
  }
  const int n_kx = SEP ? 1 : k;
  const int pitch = SEP ? kF2dCols : kPitch;
  const int tap_stride = SEP ? 1 : k;
  for (int kx = 0; kx < n_kx; ++kx) {
    const float* col = (SEP ? hbuf : tile) + ty0 * pitch + tx + kx;
    float win[8];
#pragma unroll
    for (int o = 0; o < 7; ++o) win[o] = col[o * pitch];
    for (int ky0 = 0; ky0 < k; ky0 += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ky = ky0 + j;
        if (ky < k) {                                          // wave-uniform
          win[(j + 7) & 7] = col[(ky + 7) * pitch];
          const float wv = kw[ky * tap_stride + kx];
#pragma unroll
          for (int o = 0; o < 8; ++o) acc[o] = fmaf(wv, win[(j + o) & 7], acc[o]);
        }
      }
    }
  }
  const int x = tx_base + tx;
  if (x >= w) return;
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const int y = ty_base + ty0 + o;
    if (y >= h) break;
    const size_t idx = (size_t)plane * h * w + (size_t)y * w + x;
    if (mode == 0) {
      out[idx] = acc[o];
    } else if (mode == 1) {
      const float res = sp[(size_t)y * w + x] - acc[o];
      out[idx] = res;
      out2[idx] = (fabsf(res) * 255.f > threshold) ? 1.f : 0.f;
    } else {
      const float xv = x_in[idx], soft = acc[o];
      const float sharp = fminf(fmaxf(xv + weight * res_in[idx], 0.f), 1.f);
      out[idx] = soft * sharp + (1.f - soft) * xv;
    }
  }
}

int filter2d_impl(const float* src, const float* kernels, int kernel_batch, int b, int c, int h, int w, int k, int mode, const float* x_in,
                  const float* res_in, float weight, float threshold, float* out, float* out2, hipStream_t s, bool separable = false) {
  if (!src || !kernels || !out || b <= 0 || c <= 0 || h <= 0 || w <= 0) return set_err(SRGANFD_EINVAL, "filter2d: null / empty argument");
  if (k % 2 == 0 || k < 1) return set_err(SRGANFD_EINVAL, "Wrong kernel size.");                     // the reference's ValueError text
  if (k > kF2dMaxK) return set_err(SRGANFD_EINVAL, "filter2d: kernel size %d above the LDS tile's %d", k, kF2dMaxK);
  if (k / 2 >= h || k / 2 >= w) return set_err(SRGANFD_EINVAL, "filter2d: reflect padding %d needs an image larger than that (%dx%d)", k / 2, h, w);
  if (kernel_batch != 1 && kernel_batch != b) return set_err(SRGANFD_EINVAL, "filter2d: %d kernels for %d images", kernel_batch, b);
  if ((mode == 1 && !out2) || (mode == 2 && (!x_in || !res_in)) || mode < 0 || mode > 2) return set_err(SRGANFD_EINVAL, "filter2d: bad epilogue arguments");
  if ((long long)b * c > 65535) return set_err(SRGANFD_EINVAL, "filter2d: more than 65535 planes");
  const int tiles_x = ceil_div(w, kF2dCols), tiles_y = ceil_div(h, kF2dRows);
  if (separable)
    SRGANFD_LAUNCH(filter2d_kernel<true>, dim3(tiles_x * tiles_y, b * c), dim3(256), 0, s, src, kernels, kernel_batch, c, h, w, k, tiles_x, mode, x_in, res_in,
                   weight, threshold, out, out2);
  else
    SRGANFD_LAUNCH(filter2d_kernel<false>, dim3(tiles_x * tiles_y, b * c), dim3(256), 0, s, src, kernels, kernel_batch, c, h, w, k, tiles_x, mode, x_in, res_in,
                   weight, threshold, out, out2);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// DiffJPEG (imgproc.py:1198-1497): x255, RGB -> YCbCr, 2x2 chroma average, per 8x8 block DCT -> divide by
// table * factor -> round (or the cubic "differentiable" rounding, :1183-1195) -> multiply back -> inverse DCT, chroma
// repeat, YCbCr -> RGB, clamp to [0, 255], /255; the image is zero-padded to multiples of 16 and cropped back
// (:1482-1495).  One wavefront owns one 16x16 MCU (4 luma blocks + Cb + Cr) from load to store; nothing but the RGB
// input and output touches HBM.  Lane l owns the 2x2 pixel quad (l/8, l%8) on the way in and out, and coefficient /
// pixel (l/8, l%8) of each 8x8 block in between.  The 4-D cosine tensors are the reference's fp32 products (:1250-1252,
// :1366-1368), summed over all 64 terms like its tensordot.
// tables: [dct 4096 | idct 4096 | dct scale 64 | idct alpha 64 | y_table 64 | c_table 64] floats.
// ---------------------------------------------------------------------------------------------------------------
static constexpr int kJpegTableFloats = 4096 * 2 + 64 * 4;

void diff_jpeg_tables_host(float* t) {
  static const float y_std[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                                  14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                  49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
  static const float c_small[16] = {17, 18, 24, 47, 18, 21, 26, 66, 24, 26, 56, 99, 47, 66, 99, 99};
  const double pi = 3.14159265358979323846;
  float* dct = t;
  float* idct = t + 4096;
  float* scale = t + 8192;
  float* alpha = scale + 64;
  float* ytab = alpha + 64;
  float* ctab = ytab + 64;
  for (int x = 0; x < 8; ++x)
    for (int y = 0; y < 8; ++y)
      for (int u = 0; u < 8; ++u)
        for (int v = 0; v < 8; ++v) {
          dct[((x * 8 + y) * 8 + u) * 8 + v] = (float)(cos((2 * x + 1) * u * pi / 16) * cos((2 * y + 1) * v * pi / 16));
          idct[((x * 8 + y) * 8 + u) * 8 + v] = (float)(cos((2 * u + 1) * x * pi / 16) * cos((2 * v + 1) * y * pi / 16));
        }
  for (int u = 0; u < 8; ++u)
    for (int v = 0; v < 8; ++v) {
      const double au = u == 0 ? 1.0 / sqrt(2.0) : 1.0, av = v == 0 ? 1.0 / sqrt(2.0) : 1.0;
      scale[u * 8 + v] = (float)(au * av * 0.25);
      alpha[u * 8 + v] = (float)(au * av);
      ytab[u * 8 + v] = y_std[v * 8 + u];                                    // the reference transposes the standard table (:43-48)
      ctab[u * 8 + v] = (u < 4 && v < 4) ? c_small[v * 4 + u] : 99.f;
    }
}

__global__ __launch_bounds__(256) void diff_jpeg_kernel(const float* __restrict__ src, int b, int h, int w, int mcus_x, int mcus_y,
                                                        const float* __restrict__ factor, int differentiable, const float* __restrict__ tables,
                                                        float* __restrict__ dst) {
  __shared__ float sp_all[4][384], cf_all[4][384];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long mcu = (long long)blockIdx.x * 4 + wave;
  const long long total = (long long)b * mcus_x * mcus_y;
  const bool live = mcu < total;
  float* sp = sp_all[wave];
  float* cf = cf_all[wave];
  const float* dct = tables;
  const float* idct = tables + 4096;
  const float* scale = tables + 8192;
  const float* alpha = scale + 64;
  const float* ytab = alpha + 64;
  const float* ctab = ytab + 64;
  int img = 0, my = 0, mx = 0;
  if (live) {
    img = (int)(mcu / ((long long)mcus_x * mcus_y));
    const int rem = (int)(mcu % ((long long)mcus_x * mcus_y));
    my = rem / mcus_x;
    mx = rem % mcus_x;
  }
  const size_t plane = (size_t)h * w;
  const float* ps = src + (size_t)img * 3 * plane;
  const int qy = lane >> 3, qx = lane & 7;
  // ---- load the quad, x255, RGB -> YCbCr (tensordot with the transposed matrix + shift, :1201-1212), chroma average (:1219-1226)
  {
    float cbs = 0.f, crs = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int ly = qy * 2 + dy, lx = qx * 2 + dx;
        const int y = my * 16 + ly, x = mx * 16 + lx;
        float r = 0.f, g = 0.f, bl = 0.f;
        if (live && y < h && x < w) {
          const size_t o = (size_t)y * w + x;
          r = ps[o] * 255.f; g = ps[plane + o] * 255.f; bl = ps[2 * plane + o] * 255.f;
        }
        float yy = r * 0.299f; yy = fmaf(g, 0.587f, yy); yy = fmaf(bl, 0.114f, yy);
        float cb = r * -0.168736f; cb = fmaf(g, -0.331264f, cb); cb = fmaf(bl, 0.5f, cb); cb += 128.f;
        float cr = r * 0.5f; cr = fmaf(g, -0.418688f, cr); cr = fmaf(bl, -0.081312f, cr); cr += 128.f;
        sp[ly * 16 + lx] = yy;
        cbs += cb; crs += cr;
      }
    sp[256 + qy * 8 + qx] = cbs * 0.25f;
    sp[320 + qy * 8 + qx] = crs * 0.25f;
  }
  __syncthreads();
  const float f = live ? factor[img] : 1.f;
  // ---- forward DCT of the six blocks, quantise, round, de-quantise, x alpha (:1254-1259, :1270-1278, :1183-1195, :1333-1340, :1371)
  {
    const int u = lane >> 3, v = lane & 7;
#pragma unroll 1
    for (int blk = 0; blk < 6; ++blk) {
      const float* xb = blk < 4 ? sp + ((blk >> 1) * 8) * 16 + (blk & 1) * 8 : sp + 256 + (blk - 4) * 64;
      const int pitch = blk < 4 ? 16 : 8;
      float acc = 0.f;
#pragma unroll
      for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int y = 0; y < 8; ++y) acc = fmaf(xb[x * pitch + y] - 128.f, dct[(x * 8 + y) * 64 + lane], acc);
      const float coef = scale[lane] * acc;
      const float tab = (blk < 4 ? ytab[lane] : ctab[lane]) * f;
      const float q = coef / tab;
      float rq = rintf(q);                                        // torch.round: half to even
      if (differentiable) { const float d = q - rq; rq = rq + d * d * d; }
      cf[blk * 64 + u * 8 + v] = (rq * tab) * alpha[lane];
    }
  }
  __syncthreads();
  // ---- inverse DCT (:1370-1374): pixel (a, bb) of each block = 0.25 * sum_uv X[u,v] * idct[u,v,a,bb] + 128
  {
#pragma unroll 1
    for (int blk = 0; blk < 6; ++blk) {
      const float* xb = cf + blk * 64;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 64; ++i) acc = fmaf(xb[i], idct[i * 64 + lane], acc);
      const float pv = 0.25f * acc + 128.f;
      const int a = lane >> 3, bb = lane & 7;
      if (blk < 4) sp[((blk >> 1) * 8 + a) * 16 + (blk & 1) * 8 + bb] = pv;
      else sp[256 + (blk - 4) * 64 + a * 8 + bb] = pv;
    }
  }
  __syncthreads();
  // ---- chroma repeat (:1396-1407), YCbCr -> RGB (:1414-1424), clamp and /255 (:1459-1460), crop (:1495)
  if (live) {
    const float cb = sp[256 + qy * 8 + qx] - 128.f, cr = sp[320 + qy * 8 + qx] - 128.f;
    float* pd = dst + (size_t)img * 3 * plane;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int ly = qy * 2 + dy, lx = qx * 2 + dx;
        const int y = my * 16 + ly, x = mx * 16 + lx;
        if (y < h && x < w) {
          const float yy = sp[ly * 16 + lx];
          float r = yy * 1.f; r = fmaf(cb, 0.f, r); r = fmaf(cr, 1.402f, r);
          float g = yy * 1.f; g = fmaf(cb, -0.344136f, g); g = fmaf(cr, -0.714136f, g);
          float bl = yy * 1.f; bl = fmaf(cb, 1.772f, bl); bl = fmaf(cr, 0.f, bl);
          const size_t o = (size_t)y * w + x;
          pd[o] = fminf(255.f, fmaxf(0.f, r)) / 255.f;
          pd[plane + o] = fminf(255.f, fmaxf(0.f, g)) / 255.f;
          pd[2 * plane + o] = fminf(255.f, fmaxf(0.f, bl)) / 255.f;
        }
      }
  }
}

// quality -> factor in place, as DiffJPEG.forward does on the tensor it is given (:1476-1480, :1127-1144)
__global__ void jpeg_quality_factor_kernel(float* q, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float v = q[i];
    const float t = v < 50.f ? 5000.f / v : 200.f - v * 2.f;
    q[i] = t / 100.f;
  }
}

int diff_jpeg_impl(const float* src, int b, int c, int h, int w, float* quality, int quality_is_factor, int differentiable, const float* tables,
                   float* dst, hipStream_t s) {
  if (!src || !dst || !quality || !tables || b <= 0 || h <= 0 || w <= 0) return set_err(SRGANFD_EINVAL, "diff_jpeg: null / empty argument");
  if (c != 3) return set_err(SRGANFD_EINVAL, "diff_jpeg: needs 3-channel RGB input, got %d channels", c);
  const int mcus_x = ceil_div(w, 16), mcus_y = ceil_div(h, 16);
  const long long total = (long long)b * mcus_x * mcus_y;
  if (total > (1ll << 32)) return set_err(SRGANFD_EINVAL, "diff_jpeg: too many blocks");
  if (!quality_is_factor) SRGANFD_LAUNCH(jpeg_quality_factor_kernel, dim3(ceil_div(b, 256)), dim3(256), 0, s, quality, b);
  SRGANFD_LAUNCH(diff_jpeg_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, src, b, h, w, mcus_x, mcus_y, (const float*)quality,
                 differentiable, tables, dst);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

// the last line of degradation_process (imgproc.py:2460): lr = clamp(round(x * 255), 0, 255) / 255
__global__ __launch_bounds__(256) void quantize_u8_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    dst[i] = fminf(fmaxf(rintf(src[i] * 255.f), 0.f), 255.f) / 255.f;
}
int quantize_u8_impl(const float* src, float* dst, size_t n, hipStream_t s) {
  if (!src || !dst || n == 0) return set_err(SRGANFD_EINVAL, "quantize_u8: null / empty argument");
  const size_t blocks = (n + 255) / 256;
  SRGANFD_LAUNCH(quantize_u8_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, src, dst, n);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The three F.interpolate modes degradation_process draws from (imgproc.py:2374, :2415-2418, :2440-2442, :2454-2456;
// align_corners unset, no antialias): "area" = adaptive average pooling, "bilinear", "bicubic" (A = -0.75, border
// indices clamped).  Source coordinate = rscale * (dst + 0.5) - 0.5 in fp32, rscale = 1 / scale_factor when the caller
// passed a scale factor, else in / out.  One thread per output pixel; neighbours share their taps through L1/L2.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

template <int MODE>   // 0 area, 1 bilinear, 2 bicubic
__global__ __launch_bounds__(256) void resize_kernel(const float* __restrict__ src, int planes, int h, int w, int oh, int ow, float rs_h, float rs_w,
                                                     float* __restrict__ dst) {
  const size_t total = (size_t)planes * oh * ow;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ox = (int)(i % ow);
    const size_t t = i / ow;
    const int oy = (int)(t % oh);
    const float* sp = src + (t / oh) * (size_t)h * w;
    float v;
    if (MODE == 0) {
      // adaptive_avg_pool2d: window [floor(o*in/out), ceil((o+1)*in/out))
      const int y0 = (int)(((long long)oy * h) / oh), y1 = (int)((((long long)oy + 1) * h + oh - 1) / oh);
      const int x0 = (int)(((long long)ox * w) / ow), x1 = (int)((((long long)ox + 1) * w + ow - 1) / ow);
      float sum = 0.f;
      for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) sum += sp[(size_t)y * w + x];
      v = sum / (float)((y1 - y0) * (x1 - x0));
    } else if (MODE == 1) {
      float sy = rs_h * ((float)oy + 0.5f) - 0.5f, sx = rs_w * ((float)ox + 0.5f) - 0.5f;
      sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
      const int y0 = min((int)sy, h - 1), x0 = min((int)sx, w - 1);
      const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
      const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
      v = hy * (hx * sp[(size_t)y0 * w + x0] + lx * sp[(size_t)y0 * w + x1]) + ly * (hx * sp[(size_t)y1 * w + x0] + lx * sp[(size_t)y1 * w + x1]);
    } else {
      const float A = -0.75f;
      const float sy = rs_h * ((float)oy + 0.5f) - 0.5f, sx = rs_w * ((float)ox + 0.5f) - 0.5f;
      const float fy = floorf(sy), fx = floorf(sx);
      const int iy = (int)fy, ix = (int)fx;
      const float ty = sy - fy, tx = sx - fx;
      const float cy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
      const float cx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
      v = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int y = min(max(iy - 1 + a, 0), h - 1);
        float row = 0.f;
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) row += cx[bq] * sp[(size_t)y * w + min(max(ix - 1 + bq, 0), w - 1)];
        v += cy[a] * row;
      }
    }
    dst[i] = v;
  }
}

int resize_impl(const float* src, int planes, int h, int w, int oh, int ow, int mode, float rscale_h, float rscale_w, float* dst, hipStream_t s) {
  if (!src || !dst || planes <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0) return set_err(SRGANFD_EINVAL, "resize: null / empty argument");
  if (mode < 0 || mode > 2) return set_err(SRGANFD_EINVAL, "resize: mode %d (0 area, 1 bilinear, 2 bicubic)", mode);
  const float rh = rscale_h > 0.f ? rscale_h : (float)h / (float)oh, rw = rscale_w > 0.f ? rscale_w : (float)w / (float)ow;
  const size_t total = (size_t)planes * oh * ow, blocks = (total + 255) / 256;
  const dim3 grid((unsigned)(blocks < 65536 ? blocks : 65536));
  if (mode == 0) SRGANFD_LAUNCH(resize_kernel<0>, grid, dim3(256), 0, s, src, planes, h, w, oh, ow, rh, rw, dst);
  else if (mode == 1) SRGANFD_LAUNCH(resize_kernel<1>, grid, dim3(256), 0, s, src, planes, h, w, oh, ow, rh, rw, dst);
  else SRGANFD_LAUNCH(resize_kernel<2>, grid, dim3(256), 0, s, src, planes, h, w, oh, ow, rh, rw, dst);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Noise stages (imgproc.py:832-1089).  The random draws themselves (torch.randn / torch.poisson / torch.rand) stay
// with the caller's generator -- they are handed in as tensors -- and everything deterministic around them is fused:
//   gaussian (:849-866, :1046-1060): out = clip?(image + (n_color*(1-g) + n_gray*g) * sigma/255) with per-image sigma
//     and gray flag g; n_gray is ONE (h, w) field shared by the batch, as the reference draws it (:859-860).
//   poisson (:886-919, :1077-1089): image rounded to 8 bits; vals = 2^ceil(log2(#distinct levels)) per image (a 256-bin
//     presence count, the reference's torch.unique loop); the caller draws poisson(image_q * vals); then
//     out = clip?(image + ((p/vals - image_q)*(1-g) + (pg/vals_g - gray_q)*g) * scale).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float finish_noise(float v, int clip, int rounds) {
  if (clip && rounds) return fminf(fmaxf(rintf(v * 255.f), 0.f), 255.f) / 255.f;
  if (clip) return fminf(fmaxf(v, 0.f), 1.f);
  if (rounds) return rintf(v * 255.f) / 255.f;
  return v;
}

__global__ __launch_bounds__(256) void gaussian_noise_kernel(const float* __restrict__ image, const float* __restrict__ n_color,
                                                             const float* __restrict__ n_gray, const float* __restrict__ sigma,
                                                             const float* __restrict__ gray, int c, size_t hw, size_t total, int clip, int rounds,
                                                             float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t pix = i % hw;
    const int img = (int)(i / (hw * c));
    const float sg = sigma[img];
    float noise = n_color[i] * sg / 255.f;
    if (n_gray) {
      const float g = gray[img];
      noise = noise * (1.f - g) + (n_gray[pix] * sg / 255.f) * g;
    }
    out[i] = finish_noise(image[i] + noise, clip, rounds);
  }
}

// per image: 8-bit quantised copy (colour, and optionally the grey image of torchvision's rgb_to_grayscale) + the number
// of distinct levels -> vals.  grid (blocks, b); bins are OR-ed into a 256-entry presence table per image.
__global__ __launch_bounds__(256) void poisson_prepare_kernel(const float* __restrict__ image, int c, size_t hw, int want_gray, float* __restrict__ img_q,
                                                              float* __restrict__ gray_q, unsigned int* __restrict__ presence) {
  __shared__ unsigned int seen[2][256];
  seen[0][threadIdx.x] = 0; seen[1][threadIdx.x] = 0;
  __syncthreads();
  const int img = blockIdx.y;
  const float* p = image + (size_t)img * c * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256) {
    for (int k = 0; k < c; ++k) {
      const float q = fminf(fmaxf(rintf(p[k * hw + i] * 255.f), 0.f), 255.f);
      img_q[(size_t)img * c * hw + k * hw + i] = q / 255.f;
      seen[0][(int)q] = 1;
    }
    if (want_gray) {
      const float gr = 0.2989f * p[i] + 0.587f * p[hw + i] + 0.114f * p[2 * hw + i];
      const float q = fminf(fmaxf(rintf(gr * 255.f), 0.f), 255.f);
      gray_q[(size_t)img * hw + i] = q / 255.f;
      seen[1][(int)q] = 1;
    }
  }
  __syncthreads();
  if (seen[0][threadIdx.x]) presence[(size_t)img * 512 + threadIdx.x] = 1;
  if (want_gray && seen[1][threadIdx.x]) presence[(size_t)img * 512 + 256 + threadIdx.x] = 1;
}
__global__ __launch_bounds__(256) void poisson_vals_kernel(const unsigned int* __restrict__ presence, float* __restrict__ vals, float* __restrict__ vals_gray) {
  __shared__ int cnt[2][256];
  const int img = blockIdx.x;
  cnt[0][threadIdx.x] = presence[(size_t)img * 512 + threadIdx.x] ? 1 : 0;
  cnt[1][threadIdx.x] = presence[(size_t)img * 512 + 256 + threadIdx.x] ? 1 : 0;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { cnt[0][threadIdx.x] += cnt[0][threadIdx.x + st]; cnt[1][threadIdx.x] += cnt[1][threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // 2 ** ceil(log2(n)) for 1 <= n <= 256, in integers
    for (int which = 0; which < 2; ++which) {
      const int n = cnt[which][0];
      int p2 = 1;
      while (p2 < n) p2 <<= 1;
      float* dstv = which ? vals_gray : vals;
      if (dstv) dstv[img] = (float)p2;
    }
  }
}
__global__ __launch_bounds__(256) void poisson_apply_kernel(const float* __restrict__ image, const float* __restrict__ img_q, const float* __restrict__ gray_q,
                                                            const float* __restrict__ pois, const float* __restrict__ pois_gray,
                                                            const float* __restrict__ vals, const float* __restrict__ vals_gray,
                                                            const float* __restrict__ scale, const float* __restrict__ gray, int c, size_t hw,
                                                            size_t total, int clip, int rounds, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t pix = i % hw;
    const int img = (int)(i / (hw * c));
    float noise = pois[i] / vals[img] - img_q[i];
    if (pois_gray) {
      const float g = gray[img];
      const size_t gi = (size_t)img * hw + pix;
      noise = noise * (1.f - g) + (pois_gray[gi] / vals_gray[img] - gray_q[gi]) * g;
    }
    out[i] = finish_noise(image[i] + noise * scale[img], clip, rounds);
  }
}

static inline dim3 ew_grid(size_t total) {
  const size_t blocks = (total + 255) / 256;
  return dim3((unsigned)(blocks < 32768 ? blocks : 32768));
}
int gaussian_noise_impl(const float* image, const float* n_color, const float* n_gray, const float* sigma, const float* gray, int b, int c, int h, int w,
                        int clip, int rounds, float* out, hipStream_t s) {
  if (!image || !n_color || !sigma || !out || b <= 0 || c <= 0 || h <= 0 || w <= 0 || (n_gray && !gray))
    return set_err(SRGANFD_EINVAL, "gaussian_noise: null / empty argument (a gray field needs the per-image gray flags)");
  const size_t hw = (size_t)h * w, total = hw * c * b;
  SRGANFD_LAUNCH(gaussian_noise_kernel, ew_grid(total), dim3(256), 0, s, image, n_color, n_gray, sigma, gray, c, hw, total, clip, rounds, out);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int poisson_prepare_impl(const float* image, int b, int c, int h, int w, int want_gray, float* img_q, float* gray_q, float* vals, float* vals_gray,
                         unsigned int* presence, hipStream_t s) {
  if (!image || !img_q || !vals || !presence || b <= 0 || c <= 0 || h <= 0 || w <= 0 || b > 65535)
    return set_err(SRGANFD_EINVAL, "poisson_prepare: null / empty argument");
  if (want_gray && (c != 3 || !gray_q || !vals_gray)) return set_err(SRGANFD_EINVAL, "poisson_prepare: gray noise needs 3-channel RGB and its outputs");
  const size_t hw = (size_t)h * w;
  SRGANFD_HIP_CHECK(hipMemsetAsync(presence, 0, (size_t)b * 512 * sizeof(unsigned int), s));
  const size_t blocks = (hw + 255) / 256;
  SRGANFD_LAUNCH(poisson_prepare_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256), b), dim3(256), 0, s, image, c, hw, want_gray, img_q, gray_q, presence);
  SRGANFD_LAUNCH(poisson_vals_kernel, dim3(b), dim3(256), 0, s, (const unsigned int*)presence, vals, want_gray ? vals_gray : nullptr);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}
int poisson_apply_impl(const float* image, const float* img_q, const float* gray_q, const float* pois, const float* pois_gray, const float* vals,
                       const float* vals_gray, const float* scale, const float* gray, int b, int c, int h, int w, int clip, int rounds, float* out,
                       hipStream_t s) {
  if (!image || !img_q || !pois || !vals || !scale || !out || b <= 0 || c <= 0 || h <= 0 || w <= 0)
    return set_err(SRGANFD_EINVAL, "poisson_apply: null / empty argument");
  if (pois_gray && (!gray_q || !vals_gray || !gray)) return set_err(SRGANFD_EINVAL, "poisson_apply: gray noise needs gray_q, vals_gray and the gray flags");
  const size_t hw = (size_t)h * w, total = hw * c * b;
  SRGANFD_LAUNCH(poisson_apply_kernel, ew_grid(total), dim3(256), 0, s, image, img_q, gray_q, pois, pois_gray, vals, vals_gray, scale, gray, c, hw, total, clip,
                 rounds, out);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Batch augmentation of the Real-ESRGAN loop (train_realesrgan.py:400-404 -> imgproc.py:2081-2320): one common crop window,
// one quarter-turn, one flip for every image of the GT / GT-USM / LR lists.  op: 0 copy (crop only), 1..3 = 90 / 180 / 270
// degrees counter-clockwise (torchvision's rotate about the image centre is an exact permutation for these on square planes),
// 4 horizontal flip, 5 vertical flip; the source window (top, left, ph x pw) is cropped first.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crop_rot_flip_kernel(const float* __restrict__ src, float* __restrict__ dst, int planes, int h, int w, int top,
                                                            int left, int ph, int pw, int op) {
  const size_t total = (size_t)planes * ph * pw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % pw);
    const size_t t = i / pw;
    const int y = (int)(t % ph);
    const size_t pl = t / ph;
    int sy = y, sx = x;                                  // position inside the cropped window that lands on (y, x)
    if (op == 1) { sy = x; sx = pw - 1 - y; }            // rot90 ccw: out[y][x] = in[x][W-1-y]
    else if (op == 2) { sy = ph - 1 - y; sx = pw - 1 - x; }
    else if (op == 3) { sy = ph - 1 - x; sx = y; }       // rot270 ccw (= 90 cw): out[y][x] = in[H-1-x][y]
    else if (op == 4) { sx = pw - 1 - x; }
    else if (op == 5) { sy = ph - 1 - y; }
    dst[i] = src[(pl * h + top + sy) * w + left + sx];
  }
}
int crop_rot_flip_impl(const float* src, float* dst, int planes, int h, int w, int top, int left, int ph, int pw, int op, hipStream_t s) {
  if (!src || !dst || planes <= 0 || top < 0 || left < 0 || ph <= 0 || pw <= 0 || top + ph > h || left + pw > w)
    return set_err(SRGANFD_EINVAL, "crop_rot_flip: window %dx%d at (%d,%d) outside %dx%d", ph, pw, top, left, h, w);
  if (op < 0 || op > 5) return set_err(SRGANFD_EINVAL, "crop_rot_flip: op %d", op);
  if ((op == 1 || op == 3) && ph != pw) return set_err(SRGANFD_EINVAL, "crop_rot_flip: quarter turns need a square window (%dx%d)", ph, pw);
  SRGANFD_LAUNCH(crop_rot_flip_kernel, ew_grid((size_t)planes * ph * pw), dim3(256), 0, s, src, dst, planes, h, w, top, left, ph, pw, op);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

int jpeg_table_floats() { return kJpegTableFloats; }

}  // namespace srganfd
