// Weight-streaming layers of a handful of pixels — the U-Net's two innermost levels (ganslate/nn/generators/unet/unet2d.py:36-66:
// Conv2d / ConvTranspose2d(k4, s2) over 8 x 16 .. 2 x 4 maps with 1024 - 2048 channels): 33 - 67 MB of weights per launch for at
// most 32 output pixels per parity class. On the im2col kernel they are split-K launches of 256 workgroups that each walk 8 - 16
// K-steps through a 4-stage LDS ring whose pixel tile is 128 rows of mostly zero page: 24 - 37 us per launch, 0.9 - 2 TB/s of
// weights (profiles/r06_conv_table_pix2pix.txt). Here nothing goes through LDS: a lane's 16 bytes of a weight row ARE the A
// operand of v_mfma_f32_16x16x32_bf16 (row = output channel, k octet = lane >> 4), its 16 bytes of an input pixel the B operand
// (column = pixel); a wave takes a few K-steps of one 16-channel tile with all its loads issued up front, the 16 waves of a
// workgroup add their tiles in wave order through LDS, and the workgroup writes one split of the partial sums the split-K
// finalize pass (gconv.hip) already consumes — bias, activation, statistics and the parity classes stay where they were.
#include "common.hpp"

namespace {
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct SkCls {
  const char* w;               // [w_rows][Kp] bf16, k = t * Ci + ci
  int T, Kp, py, px;
  signed char dh[16], dw[16];
};
struct SkK {
  const char* in;
  float* partial;
  long long split_stride;      // floats per split = N * Ho * Wo * Co
  int N, Hi, Wi, Ci, c_shift, in_cs, in_co, Hc, Wc, Ho, Wo, Co, so, si, w_rows, M, co_tiles, S;
  SkCls cls[8];
};

// PT: 16-pixel tiles (M <= 16 PT), KPW: 32-wide K-steps per wave
template <int PT, int KPW>
__global__ __launch_bounds__(1024) void skinny_kernel(const SkK p) {
  __shared__ f32x4 red[16][PT][64];
  const SkCls& c = p.cls[blockIdx.y];
  const int lane = threadIdx.x & 63, col = lane & 15, ko = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ct = blockIdx.x % p.co_tiles, s = blockIdx.x / p.co_tiles;
  const int ksteps = (c.T * p.Ci) >> 5;
  const int ks0 = (s * 16 + wave) * KPW;
  // this lane's pixels (one per pixel tile): image, class row / column
  int pn[PT], pi[PT], pj[PT];
  const int pixc = p.Hc * p.Wc;
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int m = t * 16 + col;
    const int mm = m < p.M ? m : 0;
    pn[t] = mm / pixc;
    const int r = mm - pn[t] * pixc;
    pi[t] = r / p.Wc;
    pj[t] = r - pi[t] * p.Wc;
    if (m >= p.M) pn[t] = -1;
  }
  const int co = ct * 16 + col;
  const bool w_ok = co < p.Co && co < p.w_rows;
  bf16x8 wv[KPW], xv[PT][KPW];
#pragma unroll
  for (int j = 0; j < KPW; ++j) {
    const int ks = ks0 + j;                            // (wave-uniform)
    uint4 w4{0u, 0u, 0u, 0u};
    if (ks < ksteps && w_ok) w4 = *reinterpret_cast<const uint4*>(c.w + ((size_t)co * c.Kp + ks * 32 + ko * 8) * 2);
    wv[j] = __builtin_bit_cast(bf16x8, w4);
    const int k = ks * 32;                             // a K-step stays inside one tap: 32 divides Ci
    const int t = ks < ksteps ? k >> p.c_shift : 0;
    const int ci = (k & (p.Ci - 1)) + ko * 8;
    const int dh = c.dh[t], dw = c.dw[t];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
      uint4 x4{0u, 0u, 0u, 0u};
      const int iy = pi[q] * p.si + dh, ix = pj[q] * p.si + dw;
      if (ks < ksteps && pn[q] >= 0 && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)
        x4 = *reinterpret_cast<const uint4*>(p.in + ((((size_t)pn[q] * p.Hi + iy) * p.Wi + ix) * p.in_cs + p.in_co + ci) * 2);
      xv[q][j] = __builtin_bit_cast(bf16x8, x4);
    }
  }
  f32x4 acc[PT];
#pragma unroll
  for (int q = 0; q < PT; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < KPW; ++j)
#pragma unroll
    for (int q = 0; q < PT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[j], xv[q][j], acc[q], 0, 0, 0);
#pragma unroll
  for (int q = 0; q < PT; ++q) red[wave][q][lane] = acc[q];
  __syncthreads();
  if (threadIdx.x < PT * 64) {
    const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
    f32x4 r = red[0][q][l];
#pragma unroll
    for (int w = 1; w < 16; ++w) r += red[w][q][l];    // (wave order: the K order of this split)
    const int m = q * 16 + (l & 15), oc = ct * 16 + (l >> 4) * 4;
    if (m < p.M && oc < p.Co) {
      const int n = m / pixc, rr = m - n * pixc, i = rr / p.Wc, jx = rr - i * p.Wc;
      const size_t opix = ((size_t)n * p.Ho + (i * p.so + c.py)) * p.Wo + (jx * p.so + c.px);
      *reinterpret_cast<f32x4*>(p.partial + (size_t)s * p.split_stride + opix * p.Co + oc) = r;
    }
  }
}
}  // namespace

// gconv.hip (the split-K branches of gs_gconv_forward_ws / gs_gconv_forward_multi_ws): *splits_used = number of splits written to
// `ws` in the layout gconv_splitk_finalize_kernel reads, 0 = not taken (the im2col launch runs)
int gs_skinny_try(const gs_gconv_desc* const* descs, int count, const void* in, const void* const* w_packs, float* ws,
                  int64_t ws_floats, void* stream, int* splits_used) {
  *splits_used = 0;
  const gs_gconv_desc* d = descs[0];
  if (!gs_opt(GS_OPT_SKINNY) || !ws || count < 1 || count > 8) return 0;
  const long long M = (long long)d->N * d->Hc * d->Wc;
  if (d->Dc != 1 || d->Di != 1 || d->Do != 1 || d->border != GS_BORDER_ZERO || M < 1 || M > 32 || d->Ci < 32 ||
      (d->Ci & (d->Ci - 1)) || (d->Co & 15) || d->Co < 16 || d->w_rows < d->Co || (d->so != 1 && d->so != 2))
    return 0;
  int ksteps = 0;
  for (int c = 0; c < count; ++c) {
    const gs_gconv_desc* dc = descs[c];
    if (dc->T < 1 || dc->T > 16 || dc->Kp < dc->T * dc->Ci || dc->Ci != d->Ci || dc->Co != d->Co || dc->Hc != d->Hc ||
        dc->Wc != d->Wc || dc->Hi != d->Hi || dc->Wi != d->Wi || dc->N != d->N || dc->so != d->so || dc->si != d->si ||
        dc->in_cs != d->in_cs || dc->in_co != d->in_co || dc->w_rows != d->w_rows || dc->Ho != d->Ho || dc->Wo != d->Wo ||
        dc->border != d->border || dc->Dc != 1 || dc->pz != 0)
      return 0;
    for (int t = 0; t < dc->T; ++t)
      if (dc->dd[t]) return 0;
    const int ks = (dc->T * dc->Ci) >> 5;
    if (ks > ksteps) ksteps = ks;
  }
  // the launch is worth it where the weights are what is streamed: >= 4 MB of them per class
  if ((long long)ksteps * 32 * d->Co * 2 < (4LL << 20)) return 0;
  const int kpw = ksteps <= 512 ? 4 : 8;
  const int S = (ksteps + 16 * kpw - 1) / (16 * kpw);
  const long long stride = (long long)d->N * d->Ho * d->Wo * d->Co;
  if (S < 1 || (int64_t)S * stride > ws_floats) return 0;
  SkK k;
  k.in = static_cast<const char*>(in); k.partial = ws; k.split_stride = stride;
  k.N = d->N; k.Hi = d->Hi; k.Wi = d->Wi; k.Ci = d->Ci; k.in_cs = d->in_cs; k.in_co = d->in_co; k.Hc = d->Hc; k.Wc = d->Wc;
  k.Ho = d->Ho; k.Wo = d->Wo; k.Co = d->Co; k.so = d->so; k.si = d->si; k.w_rows = d->w_rows; k.M = (int)M;
  k.c_shift = 0;
  while ((1 << k.c_shift) < d->Ci) ++k.c_shift;
  k.co_tiles = d->Co / 16; k.S = S;
  for (int c = 0; c < count; ++c) {
    const gs_gconv_desc* dc = descs[c];
    SkCls& sc = k.cls[c];
    sc.w = static_cast<const char*>(w_packs[c]); sc.T = dc->T; sc.Kp = dc->Kp; sc.py = dc->py; sc.px = dc->px;
    for (int t = 0; t < 16; ++t) { sc.dh[t] = t < dc->T ? dc->dh[t] : 0; sc.dw[t] = t < dc->T ? dc->dw[t] : 0; }
  }
  const dim3 grid((unsigned)(k.co_tiles * S), (unsigned)count);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int pt = M <= 16 ? 1 : 2;
  if (pt == 1 && kpw == 4) hipLaunchKernelGGL((skinny_kernel<1, 4>), grid, dim3(1024), 0, st, k);
  else if (pt == 1) hipLaunchKernelGGL((skinny_kernel<1, 8>), grid, dim3(1024), 0, st, k);
  else if (kpw == 4) hipLaunchKernelGGL((skinny_kernel<2, 4>), grid, dim3(1024), 0, st, k);
  else hipLaunchKernelGGL((skinny_kernel<2, 8>), grid, dim3(1024), 0, st, k);
  GS_CHECK_HIP(hipGetLastError());
  *splits_used = S;
  return 0;
}
