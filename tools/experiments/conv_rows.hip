// conv_rows.hip -- 3x3 stride-1 convolution as a ROW STREAM: activations go global memory -> registers -> MFMA and never touch LDS.
//
// Same arithmetic as conv_igemm.hip's 3x3 stride-1 launches (BSRGAN/model.py:42-46 dense-block convs, :104-132 U-Net convs at full
// resolution, VGG-19's first stage), for the launches whose weights were packed in the row-stream order (srganfd_pack_job.layout = 2,
// srganfd_conv_args.w_layout = 2).  conv_igemm's time is dominated by what happens around its MFMA phases -- a tile's cold first loads, two
// barriers and an LDS commit per 32-channel chunk, the epilogue's LDS round trip (profiles/r03_conv_timeline.txt: 2.5 us of MFMA in a
// 14 us tile) -- and every restructuring of that tile (rings, streams, chains, fusion) added up the same way.  This kernel has no tile:
//
//   * a wave owns a strip of 16 patch columns (14 output columns) and walks down the image.  For each patch row it loads its B
//     fragments -- lane (column n, group kg): 16 bytes = channels 32 ks + 8 kg .. + 7 of pixel (row, x0 - 1 + n) -- straight into
//     registers, a ring of rows ahead of the MFMAs.  In a planar 32-channel-group buffer one such load instruction is 1 KiB contiguous.
//   * the three kernel COLUMNS are folded into M: rows (kx, co) of D'[(kx, co)][n] = sum over (ky, ci) W[co][ci][ky][kx] X[row + ky][n][ci],
//     96 rows = six 16-row tiles for 32 output channels, no padding; out[co][x] = D'[0][co][x] + D'[1][co][x + 1] + D'[2][co][x + 2] is
//     two DPP row shifts per register in the epilogue.  The three kernel ROWS are three rolling accumulator sets: patch row r feeds
//     output rows r - 1, r, r + 1 and is read once.
//   * the weights of the workgroup's 32 output channels (3 x cin/32 x 6 fragments of 1 KiB) sit in LDS for the whole kernel; an A
//     fragment is one conflict-free ds_read_b128 per MFMA.  That is the kernel's LDS traffic: 1 KiB per v_mfma_f32_16x16x32 -- the LDS
//     array's 256 B/clk at the full MFMA rate of four SIMDs -- so LDS reads, not tile overheads, are what bound it.
//   * one __syncthreads after the weight image is written, none afterwards: waves drift apart freely, loads and MFMAs of different waves
//     overlap (tools/probes/overlap_probe.hip: the chip overlaps the two perfectly in one steady loop of the same waves).
//   * epilogue in registers: a lane ends up with 8 consecutive output channels of one pixel (weight rows are permuted for that):
//     bias, activation, optional pre-residual copy y2, residual r1, LeakyReLU' mask, 16-byte stores.
// Geometry cost: 14 of 16 columns and TH of TH + 2 rows of each strip are useful.
#include "conv_common.hpp"

namespace srganfd {

struct ConvRowsK {
  const void* x; void* y; void* y2; const void* r1; const void* mask; const void* w;
  const float* bias; const float* alpha_dev;
  int xC, x_ps, x_gs, x_c0;
  int yC, y_ps, y_gs, y_c0, y2C, y2_ps, y2_gs, y2_c0, r1C, r1_ps, r1_gs, r1_c0, mC, m_ps, m_gs, m_c0;
  int N, H, W, TH, nco, tiles_x, tiles_y;
  float alpha, neg, post_scale, r1s, mask_slope;
};

constexpr int kRowsOob = 0x7fffffff;

template <typename T, int NCH, int NW>
__global__ __launch_bounds__(64 * NW) void conv_rows_kernel(const ConvRowsK a) {
  using Frag = typename FragAB<T>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NFRAG = 3 * NCH * 6;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- block -> (image, row band, strip group, output-channel tile) ----
  int bid = blockIdx.x;
  const int cot = bid % a.nco; bid /= a.nco;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int img = bid / a.tiles_y;
  // ---- weight image of this tile's 32 output channels: linear 16-byte copy of [ky][ks][t][lane][8] ----
  {
    const u32x4* src = (const u32x4*)a.w + (size_t)cot * NFRAG * 64;
    for (int i = tid; i < NFRAG * 64; i += 64 * NW) ((u32x4*)smem)[i] = src[i];
  }
  __syncthreads();
  const int n16 = lane & 15, kg = lane >> 4;
  const int x0 = (tx * NW + wave) * 14;
  if (x0 >= a.W) return;
  const int y0 = ty * a.TH, y1 = min(y0 + a.TH, a.H);
  const int gx = x0 - 1 + n16;
  const bool okx = gx >= 0 && gx < a.W;
  auto uniform_ptr = [](const void* p) -> void* {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
  };
  const T* ximg = (const T*)a.x + (size_t)img * a.H * a.W * a.xC;
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ximg), (short)0, (int)((unsigned)a.H * (unsigned)a.W * (unsigned)a.xC * 2u), 0x00020000);
  int coff[NCH];
#pragma unroll
  for (int ks = 0; ks < NCH; ++ks) {
    const int c = a.x_c0 + 32 * ks + 8 * kg;
    coff[ks] = ((c >> 5) * a.x_gs + (c & 31)) * 2;
  }
  auto ldrow = [&](int gy, u32x4* out) {
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned v4u;
    const bool ok = okx && gy >= 0 && gy < a.H;
    const int base = (gy * a.W + gx) * a.x_ps * 2;
#pragma unroll
    for (int ks = 0; ks < NCH; ++ks) {
      const v4u r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? base + coff[ks] : kRowsOob, 0, 0);
      out[ks] = __builtin_bit_cast(u32x4, r);
    }
  };
  // ---- epilogue constants: this lane's 8 output channels are cot * 32 + 8 kg .. + 7 ----
  const int co8 = cot * 32 + 8 * kg;
  float bias8[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bias8[q] = a.bias ? a.bias[co8 + q] : 0.f;
  float alpha = a.alpha;
  if (a.alpha_dev) alpha *= *a.alpha_dev;
  auto voff = [&](int c0, int ps, int gs, int& chan_off) { const int c = c0 + co8; chan_off = (c >> 5) * gs + (c & 31); (void)ps; };
  int yo, y2o = 0, r1o = 0, mo = 0;
  voff(a.y_c0, a.y_ps, a.y_gs, yo);
  if (a.y2) voff(a.y2_c0, a.y2_ps, a.y2_gs, y2o);
  if (a.r1) voff(a.r1_c0, a.r1_ps, a.r1_gs, r1o);
  if (a.mask) voff(a.m_c0, a.m_ps, a.m_gs, mo);
  const size_t ipix = (size_t)img * a.H * a.W;
  const int ox = x0 + n16;
  const bool st_ok = n16 < 14 && ox < a.W;

  constexpr int D = 2;                       // patch rows of loads in flight ahead of the MFMAs
  u32x4 ring[D][NCH];
  const int pr0 = y0 - 1, pr1 = y1;          // patch rows pr0 .. pr1 inclusive
#pragma unroll
  for (int d = 0; d < D; ++d) ldrow(pr0 + d, ring[d]);
  f32x4_t acc0[6], acc1[6];                  // output rows pr - 1 (kernel row 2 still missing) and pr (rows 1, 2 missing)
#pragma unroll
  for (int t = 0; t < 6; ++t) { acc0[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc1[t] = acc0[t]; }
  const char* wl = smem + lane * 16;
  for (int prb = pr0; prb <= pr1; prb += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int pr = prb + d;
      Frag b[NCH];
#pragma unroll
      for (int ks = 0; ks < NCH; ++ks) b[ks] = __builtin_bit_cast(Frag, ring[d][ks]);
      ldrow(pr + D, ring[d]);
      if (pr > pr1) continue;
      f32x4_t acc2[6];
#pragma unroll
      for (int t = 0; t < 6; ++t) acc2[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      // patch row pr: kernel row 2 of output row pr - 1, row 1 of pr, row 0 of pr + 1; per k-step 18 independent accumulators
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < NCH; ++ks) {
#pragma unroll
        for (int t = 0; t < 6; ++t) acc0[t] = mfma16<T>(*(const Frag*)(wl + ((2 * NCH + ks) * 6 + t) * 1024), b[ks], acc0[t]);
#pragma unroll
        for (int t = 0; t < 6; ++t) acc1[t] = mfma16<T>(*(const Frag*)(wl + ((1 * NCH + ks) * 6 + t) * 1024), b[ks], acc1[t]);
#pragma unroll
        for (int t = 0; t < 6; ++t) acc2[t] = mfma16<T>(*(const Frag*)(wl + ((0 * NCH + ks) * 6 + t) * 1024), b[ks], acc2[t]);
      }
      __builtin_amdgcn_s_setprio(0);
      const int oy = pr - 1;
      if (oy >= y0) {
        // out[co][x = n16] = D'[kx 0][n16] + D'[kx 1][n16 + 1] + D'[kx 2][n16 + 2]: DPP row shifts inside the 16-lane rows (same kg)
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int s1 = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc0[2 + h][r]), 0x101, 0xf, 0xf, true);   // row_shl:1
            const int s2 = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc0[4 + h][r]), 0x102, 0xf, 0xf, true);   // row_shl:2
            v[4 * h + r] = (acc0[h][r] + __builtin_bit_cast(float, s1)) + __builtin_bit_cast(float, s2);
          }
        if (st_ok) {
          const int p = oy * a.W + ox;                 // pixel inside the image (32-bit offsets: host-checked)
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float t = alpha * v[q] + bias8[q];
            v[q] = t * (t > 0.f ? a.post_scale : a.neg * a.post_scale);
          }
          if (a.y2) *(u32x4*)((T*)a.y2 + ipix * a.y2C + (p * a.y2_ps + y2o)) = pack8<T>(v);
          if (a.r1) {
            float rv[8];
            unpack8<T>(*(const u32x4*)((const T*)a.r1 + ipix * a.r1C + (p * a.r1_ps + r1o)), rv);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += a.r1s * rv[q];
          }
          if (a.mask) {
            float mv[8];
            unpack8<T>(*(const u32x4*)((const T*)a.mask + ipix * a.mC + (p * a.m_ps + mo)), mv);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] *= mv[q] > 0.f ? 1.f : a.mask_slope;
          }
          *(u32x4*)((T*)a.y + ipix * a.yC + (p * a.y_ps + yo)) = pack8<T>(v);
        }
      }
#pragma unroll
      for (int t = 0; t < 6; ++t) { acc0[t] = acc1[t]; acc1[t] = acc2[t]; }
    }
  }
}

// ---- host side ----
template <typename T, int NCH>
static int launch_rows(const ConvRowsK& k, hipStream_t s) {
  constexpr int NW = 8;
  constexpr int lds = 3 * NCH * 6 * 1024;
  auto kern = conv_rows_kernel<T, NCH, NW>;
  if (g_describe) { snprintf(g_describe, g_describe_len, "conv_rows_kernel<%s,NCH=%d>", dtype_name<T>(), NCH); return SRGANFD_OK; }
  static unsigned long long attr_done = 0;
  if (!g_dry_run) {
    int dev = 0;
    SRGANFD_HIP_CHECK(hipGetDevice(&dev));
    if (!(attr_done >> (dev & 63) & 1ULL)) {
      SRGANFD_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr_done |= 1ULL << (dev & 63);
    }
  }
  ConvRowsK kk = k;
  kk.tiles_x = ceil_div(ceil_div(k.W, 14), NW);
  // row bands: as tall as possible (halo rows are re-read and re-multiplied) while the launch still fills the chip a few times over
  int th = 64;
  const long long want = 4LL * conv_device_cus();
  while (th > 8 && (long long)k.N * kk.tiles_x * ceil_div(k.H, th) * k.nco < want) th >>= 1;
  kk.TH = th;
  kk.tiles_y = ceil_div(k.H, th);
  const long long grid = (long long)k.N * kk.tiles_x * kk.tiles_y * k.nco;
  if (grid <= 0 || grid > 0x7fffffffLL) return set_err(SRGANFD_EINVAL, "conv2d(rows): bad grid %lld", grid);
  SRGANFD_LAUNCH(kern, dim3((unsigned)grid), dim3(64 * NW), lds, s, kk);
  SRGANFD_HIP_CHECK(hipGetLastError());
  return SRGANFD_OK;
}

int conv_rows_impl(const srganfd_conv_args* a, hipStream_t s) {
  if (!a || !a->x.ptr || !a->y.ptr || !a->w_packed) return set_err(SRGANFD_EINVAL, "conv2d(rows): null pointer");
  if (a->dtype == SRGANFD_F32 || a->ksize != 3 || a->stride != 1 || a->pad != 1 || a->up || a->out_sy > 1 || a->out_sx > 1 || a->y_f32 || a->r2.ptr ||
      a->cout_store != a->cout || a->cout % 32 || a->cin % 32 || a->cin < 64 || a->cin > 192 || a->h_out != a->h_in || a->w_out != a->w_in)
    return set_err(SRGANFD_EINVAL, "conv2d(rows): 16-bit 3x3 stride-1 pad-1 convs with 64..192 input channels only (no up / strided / fp32 output / r2)");
  const long long ipix = (long long)a->h_in * a->w_in;
  for (const srganfd_view* v : {&a->x, &a->y, &a->y2, &a->r1, &a->mask}) {
    if (!v->ptr) continue;
    if (v->c0 % 8 || v->cstride % 8 || ((uintptr_t)v->ptr & 15)) return set_err(SRGANFD_EINVAL, "conv2d(rows): views must be 16-byte aligned");
    if (v->planar && (v->c0 % 32 || v->cstride % 32)) return set_err(SRGANFD_EINVAL, "conv2d(rows): a planar view needs c0 and cstride multiples of 32");
    if ((size_t)ipix * (size_t)v->cstride * 2 >= 0x7fffffffULL) return set_err(SRGANFD_EINVAL, "conv2d(rows): one image exceeds 2 GiB");
  }
  if (a->x.c0 + a->cin > a->x.cstride || a->y.c0 + a->cout > a->y.cstride) return set_err(SRGANFD_EINVAL, "conv2d(rows): view exceeds buffer channels");
  ConvRowsK k;
  auto strides = [&](const srganfd_view& v, int& C, int& ps, int& gs, int& c0) { C = v.cstride; ps = v.planar ? 32 : v.cstride; gs = v.planar ? (int)(ipix * 32) : 32; c0 = v.c0; };
  k.x = a->x.ptr; k.y = a->y.ptr; k.y2 = a->y2.ptr; k.r1 = a->r1.ptr; k.mask = a->mask.ptr; k.w = a->w_packed; k.bias = a->bias; k.alpha_dev = a->alpha_dev;
  strides(a->x, k.xC, k.x_ps, k.x_gs, k.x_c0);
  strides(a->y, k.yC, k.y_ps, k.y_gs, k.y_c0);
  strides(a->y2, k.y2C, k.y2_ps, k.y2_gs, k.y2_c0);
  strides(a->r1, k.r1C, k.r1_ps, k.r1_gs, k.r1_c0);
  strides(a->mask, k.mC, k.m_ps, k.m_gs, k.m_c0);
  k.N = a->n; k.H = a->h_in; k.W = a->w_in; k.nco = a->cout / 32; k.TH = 0; k.tiles_x = k.tiles_y = 0;
  k.alpha = a->alpha; k.neg = a->act == SRGANFD_ACT_LRELU ? a->slope : (a->act == SRGANFD_ACT_RELU ? 0.f : 1.f);
  k.post_scale = a->post_scale; k.r1s = a->r1_scale; k.mask_slope = a->mask_slope;
#define ROWS_NCH(TT) \
  switch (a->cin / 32) { \
    case 2: return launch_rows<TT, 2>(k, s); case 3: return launch_rows<TT, 3>(k, s); case 4: return launch_rows<TT, 4>(k, s); \
    case 5: return launch_rows<TT, 5>(k, s); default: return launch_rows<TT, 6>(k, s); }
  if (a->dtype == SRGANFD_F16) { ROWS_NCH(f16_t) }
  ROWS_NCH(bf16_t)
#undef ROWS_NCH
}

}  // namespace srganfd
