// V-Net elementwise family (ganslate/nn/generators/vnet/vnet3d.py:155-267, ganslate/nn/invertible.py:8-48):
// InstanceNorm3d(affine=False) -> [+ residual] -> nn.PReLU(C) (learnable per-channel slope) -> [+ residual] on channel
// slices of NDHWC bf16 tensors, and the autograd backward of that chain including the slope gradient.
//   forward : u = norm(y) (or y);  mode 1: u += res;  v = u > 0 ? u : slope[c]*u;  mode 2: v += res;  out = v
//   backward: gt = g (+ g2);  gu = gt * (u > 0 ? 1 : slope[c]);  dslope[c] += sum gt*min(u,0);
//             dy = rstd*(gu - mean gu - yhat*mean(gu*yhat))  (or gu without norm);  gres = gu (mode 1, optional)
// The residual of mode 1 may be a channel-repeated tensor (InputBlock's x.repeat, vnet3d.py:162-167): channel c reads
// res channel c % res_mod. Uses: InputBlock (norm, mode 1, repeat), down/up convs (norm, no residual), additive
// coupling y1 = x1 + PReLU(IN(conv(x2))) (norm, mode 2), block tails PReLU(core(x) + x) (no norm, mode 1).
// HBM-bound streaming kernels: 16 B per lane, fp32 math, deterministic two-level reductions (atomics only for the
// final per-image accumulation into the parameter-gradient buffer, as in wgrad).
#include "common.hpp"

// norm.hip: parameter gradients summed over the images in a fixed order
int gs_launch_norm_param_grads(const float* sums, int R, const float* mean_rstd, float* db, float* dslope, int N, int C,
                               float inv_hw, hipStream_t st);

struct PNormK {
  gs_pnorm_desc d;
  int C8;
  unsigned HW;
  int c8_shift;               // log2(C8) when C8 is a power of two (the element loops' grid strides are multiples of 256), else -1
};

__device__ __forceinline__ void pn_load8(float* f, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void pn_unpack8(float* f, const uint4 v) {
  f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 pn_pack8(const float* f) {
  uint4 o;
  o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
  return o;
}
__device__ __forceinline__ void pn_view8(float* f, const unsigned short* t, size_t pix, int cs, int co, int c8) {
  pn_unpack8(f, *reinterpret_cast<const uint4*>(t + pix * cs + co + c8 * 8));
}
// residual of 8 consecutive channels starting at c8*8 of pixel `pix`
__device__ __forceinline__ void pn_res8(float* f, const gs_pnorm_desc& d, const unsigned short* res, size_t pix,
                                        int c8) {
  if (d.res_mod > 0) {
    const unsigned short* r = res + pix * d.res_cs + d.res_co;
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = bf2f(r[(c8 * 8 + k) % d.res_mod]);
  } else {
    pn_view8(f, res, pix, d.res_cs, d.res_co, c8);
  }
}

// The per-channel numbers of one 8-channel group. A thread whose channel group does not change over its loop (every power-of-two
// channel count: the grid stride is a multiple of C8) loads them ONCE: per 16-byte element the kernels fetched three to six
// 32-byte vectors of statistics / slopes / sums and ran an integer division for the pixel index.
struct PnCh {
  float mu[8], rs[8], sl[8];
};
__device__ __forceinline__ void pn_ch_load(PnCh& c, const gs_pnorm_desc& d, const float* mr, const float* slope, int c8) {
  if (mr) { pn_load8(c.mu, mr + c8 * 8); pn_load8(c.rs, mr + d.C + c8 * 8); }
  if (slope) pn_load8(c.sl, slope + c8 * 8);
}

// pre-activation u of one 8-channel group
__device__ __forceinline__ void pn_preact_c(float* u, float* yh, const PNormK& p, const unsigned short* y, bool has_mr,
                                            const PnCh& c, const unsigned short* res, size_t pix, int c8) {
  const gs_pnorm_desc& d = p.d;
  pn_view8(yh, y, pix, d.y_cs, d.y_co, c8);
  if (has_mr) {
#pragma unroll
    for (int k = 0; k < 8; ++k) yh[k] = (yh[k] - c.mu[k]) * c.rs[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) u[k] = yh[k];
  if (d.res_mode == 1) {
    float r[8];
    pn_res8(r, d, res, pix, c8);
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] += r[k];
  }
}
__device__ __forceinline__ void pn_preact(float* u, float* yh, const PNormK& p, const unsigned short* y,
                                          const float* mr, const unsigned short* res, size_t pix, int c8) {
  const gs_pnorm_desc& d = p.d;
  pn_view8(yh, y, pix, d.y_cs, d.y_co, c8);
  if (mr) {
    float mu[8], rs[8];
    pn_load8(mu, mr + c8 * 8);
    pn_load8(rs, mr + d.C + c8 * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) yh[k] = (yh[k] - mu[k]) * rs[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) u[k] = yh[k];
  if (d.res_mode == 1) {
    float r[8];
    pn_res8(r, d, res, pix, c8);
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] += r[k];
  }
}

__global__ __launch_bounds__(256) void pnorm_fwd_kernel(const PNormK p, const unsigned short* y, const float* mean_rstd,
                                                        const unsigned short* res, const float* slope,
                                                        unsigned short* out) {
  const gs_pnorm_desc& d = p.d;
  const int n = blockIdx.y;
  const unsigned per_img = p.HW * (unsigned)p.C8;
  const float* mr = mean_rstd ? mean_rstd + (size_t)n * 2 * d.C : nullptr;
  const bool fixed = p.c8_shift >= 0;                  // (uniform) power-of-two channel groups: this thread keeps its group
  PnCh ch;
  if (fixed) pn_ch_load(ch, d, mr, slope, (int)((blockIdx.x * 256u + threadIdx.x) & (unsigned)(p.C8 - 1)));
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < per_img; e += gridDim.x * 256u) {
    const unsigned px = fixed ? e >> p.c8_shift : e / (unsigned)p.C8;
    const int c8 = (int)(e - px * (unsigned)p.C8);
    const size_t pix = (size_t)n * p.HW + px;
    float u[8], yh[8];
    if (!fixed) pn_ch_load(ch, d, mr, slope, c8);
    pn_preact_c(u, yh, p, y, mr != nullptr, ch, res, pix, c8);
    if (slope) {
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = u[k] > 0.f ? u[k] : ch.sl[k] * u[k];
    }
    if (d.res_mode == 2) {
      float r[8];
      pn_res8(r, d, res, pix, c8);
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] += r[k];
    } else if (d.res_mode == 3) {        // inverse of an additive coupling: x = y - F(.)  (memcnn AdditiveCoupling.inverse)
      float r[8];
      pn_res8(r, d, res, pix, c8);
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = r[k] - u[k];
    }
    *reinterpret_cast<uint4*>(out + pix * d.out_cs + d.out_co + c8 * 8) = pn_pack8(u);
  }
}

// gu and the slope-gradient integrand of one 8-channel group
__device__ __forceinline__ void pn_gu_c(float* gu, float* gs, const PNormK& p, const unsigned short* g,
                                        const unsigned short* g2, const float* u, bool has_slope, const PnCh& c, size_t pix,
                                        int c8) {
  const gs_pnorm_desc& d = p.d;
  float a[8], b[8];
  pn_view8(a, g, pix, d.g_cs, d.g_co, c8);
  if (g2) {
    pn_view8(b, g2, pix, d.g2_cs, d.g2_co, c8);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += b[k];
  }
  if (d.res_mode == 3) {                 // out = res - v: the branch sees the negated gradient
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = -a[k];
  }
  if (has_slope) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gu[k] = u[k] > 0.f ? a[k] : a[k] * c.sl[k];
      gs[k] = u[k] > 0.f ? 0.f : a[k] * u[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) { gu[k] = a[k]; gs[k] = 0.f; }
  }
}
__device__ __forceinline__ void pn_gu(float* gu, float* gs, const PNormK& p, const unsigned short* g,
                                      const unsigned short* g2, const float* u, const float* slope, size_t pix,
                                      int c8) {
  const gs_pnorm_desc& d = p.d;
  float a[8], b[8], sl[8];
  pn_view8(a, g, pix, d.g_cs, d.g_co, c8);
  if (g2) {
    pn_view8(b, g2, pix, d.g2_cs, d.g2_co, c8);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += b[k];
  }
  if (d.res_mode == 3) {                 // out = res - v: the branch sees the negated gradient
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = -a[k];
  }
  if (slope) {
    pn_load8(sl, slope + c8 * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gu[k] = u[k] > 0.f ? a[k] : a[k] * sl[k];
      gs[k] = u[k] > 0.f ? 0.f : a[k] * u[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) { gu[k] = a[k]; gs[k] = 0.f; }
  }
}

// pass 1: per (n, pixel chunk) partial sums of (gu, gu*yhat, yhat, g*min(u,0)) -> partial [N][chunks][4][C]
template <int COLS>
__global__ __launch_bounds__(256) void pnorm_bwd_reduce_kernel(const PNormK p, const unsigned short* g,
                                                               const unsigned short* g2, const unsigned short* y,
                                                               const float* mean_rstd, const unsigned short* res,
                                                               const float* slope, float* partial, int pix_per_block,
                                                               int chunks) {
  constexpr int ROWS = 256 / COLS;
  __shared__ float red[ROWS][COLS][33];
  const gs_pnorm_desc& d = p.d;
  const int n = blockIdx.y, tid = threadIdx.x;
  const int col = tid % COLS, row = tid / COLS;
  const int c8 = blockIdx.z * COLS + col;
  const unsigned p0 = blockIdx.x * pix_per_block;
  const unsigned p1 = min(p.HW, p0 + pix_per_block);
  const float* mr = mean_rstd ? mean_rstd + (size_t)n * 2 * d.C : nullptr;
  float acc[4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[r][k] = 0.f;
  if (c8 < p.C8) {
    PnCh ch;                                         // (the thread's channel group is fixed: its numbers once)
    pn_ch_load(ch, d, mr, slope, c8);
    for (unsigned px = p0 + row; px < p1; px += ROWS) {
      const size_t pix = (size_t)n * p.HW + px;
      float u[8], yh[8], gu[8], gs[8];
      pn_preact_c(u, yh, p, y, mr != nullptr, ch, res, pix, c8);
      pn_gu_c(gu, gs, p, g, g2, u, slope != nullptr, ch, pix, c8);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        acc[0][k] += gu[k]; acc[1][k] += gu[k] * yh[k]; acc[2][k] += yh[k]; acc[3][k] += gs[k];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < 8; ++k) red[row][col][r * 8 + k] = acc[r][k];
  __syncthreads();
  for (int o = tid; o < COLS * 32; o += 256) {
    const int cc = o / 32, k = o - cc * 32;
    const int ch8 = blockIdx.z * COLS + cc;
    if (ch8 < p.C8) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) sum += red[r][cc][k];
      float* out = partial + ((size_t)n * chunks + blockIdx.x) * 4 * d.C;
      out[(k >> 3) * d.C + ch8 * 8 + (k & 7)] = sum;
    }
  }
}

// pass 2: per (n, c) totals over the chunks -> sums [N][4][C]; parameter gradients:
//   dslope[c] += S4;  bias_grad[c] += -rstd*S2*S3/hw (sum over pixels of dy, the bias of the conv in front of the norm)
template <int CH>
__global__ __launch_bounds__(256) void pnorm_bwd_finalize_kernel(const float* partial, float* sums, int chunks, int C,
                                                                 float inv_hw, const float* mean_rstd, float* dslope,
                                                                 float* db) {
  constexpr int LANES = 256 / CH;
  __shared__ double red[4][LANES][CH + 1];
  __shared__ float tot[4][CH];
  const int n = blockIdx.y, tid = threadIdx.x;
  const int col = tid % CH, lane = tid / CH;
  const int c = blockIdx.x * CH + col;
  const float* src = partial + (size_t)n * chunks * 4 * C;
  // all four sums of a channel in ONE sweep (the four rows of a chunk are neighbours: 4 x 4 loads in flight per thread; additions
  // per sum in chunk order, exactly the order of the four serial sweeps this replaced — they cost 16 dependent memory round
  // trips and three workgroup barriers more, 8-12 us per launch, 148 launches per V-Net step)
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < C) {
    int sl = lane;
    for (; sl + 3 * LANES < chunks; sl += 4 * LANES) {
      float v[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[k][r] = src[((size_t)(sl + k * LANES) * 4 + r) * C + c];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += (double)v[k][r];
    }
    for (; sl < chunks; sl += LANES)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] += (double)src[((size_t)sl * 4 + r) * C + c];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[r][lane][col] = s[r];
  __syncthreads();
  if (lane < 4) {                                  // sum r = lane over the lanes, in lane order
    double t = 0.0;
#pragma unroll
    for (int l = 0; l < LANES; ++l) t += red[lane][l][col];
    tot[lane][col] = (float)t;
  }
  __syncthreads();
  if (lane == 0 && c < C) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sums[((size_t)n * 4 + r) * C + c] = tot[r][col];
    if (gridDim.y == 1) {
      // one image: its totals ARE the parameter gradients — norm_param_grads_kernel's arithmetic (norm.hip) with its sum over
      // one image, here instead of a launch of one workgroup behind this one (184 such launches per V-Net step at batch 1)
      if (db && mean_rstd)
        db[c] = __fadd_rn(db[c], __fmul_rn(__fmul_rn(__fmul_rn(-mean_rstd[C + c], tot[1][col]), tot[2][col]), inv_hw));
      if (dslope) dslope[c] = __fadd_rn(dslope[c], tot[3][col]);
    }
  }
}

// pass 3: dy (and optionally gres = gu)
__global__ __launch_bounds__(256) void pnorm_bwd_apply_kernel(const PNormK p, const unsigned short* g,
                                                              const unsigned short* g2, const unsigned short* y,
                                                              const float* mean_rstd, const unsigned short* res,
                                                              const float* slope, const float* sums,
                                                              unsigned short* dy, unsigned short* gres) {
  const gs_pnorm_desc& d = p.d;
  const int n = blockIdx.y;
  const unsigned per_img = p.HW * (unsigned)p.C8;
  const float inv_hw = 1.0f / (float)p.HW;
  const float* mr = mean_rstd ? mean_rstd + (size_t)n * 2 * d.C : nullptr;
  const float* sm = sums ? sums + (size_t)n * 4 * d.C : nullptr;
  const bool fixed = p.c8_shift >= 0;                  // (uniform) see pnorm_fwd_kernel
  PnCh ch;
  float s1[8], s2[8];
  auto load_ch = [&](int c8) {
    pn_ch_load(ch, d, mr, slope, c8);
    if (mr) { pn_load8(s1, sm + c8 * 8); pn_load8(s2, sm + d.C + c8 * 8); }
  };
  if (fixed) load_ch((int)((blockIdx.x * 256u + threadIdx.x) & (unsigned)(p.C8 - 1)));
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < per_img; e += gridDim.x * 256u) {
    const unsigned px = fixed ? e >> p.c8_shift : e / (unsigned)p.C8;
    const int c8 = (int)(e - px * (unsigned)p.C8);
    const size_t pix = (size_t)n * p.HW + px;
    float u[8], yh[8], gu[8], gs[8], o[8];
    if (!fixed) load_ch(c8);
    pn_preact_c(u, yh, p, y, mr != nullptr, ch, res, pix, c8);
    pn_gu_c(gu, gs, p, g, g2, u, slope != nullptr, ch, pix, c8);
    if (gres) *reinterpret_cast<uint4*>(gres + pix * d.gres_cs + d.gres_co + c8 * 8) = pn_pack8(gu);
    if (mr) {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = ch.rs[k] * (gu[k] - s1[k] * inv_hw - yh[k] * s2[k] * inv_hw);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = gu[k];
    }
    *reinterpret_cast<uint4*>(dy + pix * d.dy_cs + d.dy_co + c8 * 8) = pn_pack8(o);
  }
}

static int pn_c8_shift(int C8) {
  if (C8 <= 0 || C8 > 256 || (C8 & (C8 - 1))) return -1;
  int sh = 0;
  while ((1 << sh) < C8) ++sh;
  return sh;
}

static int pn_pix_per_block(long long pixels) {
  // (1024 chunks per image: 2048 made the second pass walk twice the rows for nothing — brats 54.48 -> 54.15 ms, 512: 55.08)
  long long ppb = (pixels + 1023) / 1024;
  return ppb < 64 ? 64 : (int)((ppb + 63) / 64 * 64);
}

static int pn_check(const gs_pnorm_desc* d, const char* who) {
  GS_REQUIRE(d && d->N > 0 && d->pixels > 0 && d->C > 0 && (d->C & 7) == 0, "%s: bad shape", who);
  GS_REQUIRE(d->pixels * (d->C / 8) < (1LL << 31), "%s: image too large", who);
  GS_REQUIRE(d->res_mode >= 0 && d->res_mode <= 3, "%s: res_mode must be 0 .. 3", who);
  GS_REQUIRE((d->y_cs & 7) == 0 && (d->y_co & 7) == 0, "%s: y view must be 8-channel aligned", who);
  GS_REQUIRE(d->res_mode == 0 || d->res_mod > 0 || ((d->res_cs & 7) == 0 && (d->res_co & 7) == 0),
             "%s: residual view must be 8-channel aligned", who);
  return 0;
}

extern "C" int gs_pnorm_forward(const gs_pnorm_desc* d, const void* y, const float* mean_rstd, const void* res,
                                const float* slope, void* out, void* stream) {
  if (int rc = pn_check(d, "gs_pnorm_forward")) return rc;
  GS_REQUIRE(y && out && (d->res_mode == 0 || res), "gs_pnorm_forward: null tensor");
  GS_REQUIRE((d->out_cs & 7) == 0 && (d->out_co & 7) == 0, "gs_pnorm_forward: output view must be 8-channel aligned");
  PNormK k;
  k.d = *d; k.C8 = d->C / 8; k.HW = (unsigned)d->pixels;
  k.c8_shift = pn_c8_shift(k.C8);
  long long bx = ((long long)k.HW * k.C8 + 255) / 256;
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(pnorm_fwd_kernel, dim3((unsigned)bx, d->N), dim3(256), 0, static_cast<hipStream_t>(stream), k,
                     static_cast<const unsigned short*>(y), mean_rstd, static_cast<const unsigned short*>(res), slope,
                     static_cast<unsigned short*>(out));
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int64_t gs_pnorm_backward_scratch_floats(const gs_pnorm_desc* d) {
  const int64_t ppb = pn_pix_per_block(d->pixels);
  const int64_t chunks = (d->pixels + ppb - 1) / ppb;
  return (int64_t)d->N * (chunks + 1) * 4 * d->C;
}

extern "C" int gs_pnorm_backward(const gs_pnorm_desc* d, const void* g, const void* g2, const void* y,
                                 const float* mean_rstd, const void* res, const float* slope, void* dy, void* gres,
                                 float* dslope, float* bias_grad, float* scratch, void* stream) {
  if (int rc = pn_check(d, "gs_pnorm_backward")) return rc;
  GS_REQUIRE(g && y && dy && (d->res_mode != 1 || res), "gs_pnorm_backward: null tensor");
  GS_REQUIRE((d->g_cs & 7) == 0 && (d->g_co & 7) == 0 && (!g2 || ((d->g2_cs & 7) == 0 && (d->g2_co & 7) == 0)) &&
                 (d->dy_cs & 7) == 0 && (d->dy_co & 7) == 0 && (!gres || ((d->gres_cs & 7) == 0 && (d->gres_co & 7) == 0)),
             "gs_pnorm_backward: views must be 8-channel aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  PNormK k;
  k.d = *d; k.C8 = d->C / 8; k.HW = (unsigned)d->pixels;
  k.c8_shift = pn_c8_shift(k.C8);
  const unsigned short* gp = static_cast<const unsigned short*>(g);
  const unsigned short* g2p = static_cast<const unsigned short*>(g2);
  const unsigned short* yp = static_cast<const unsigned short*>(y);
  const unsigned short* rp = static_cast<const unsigned short*>(res);
  float* sums = nullptr;
  if (mean_rstd || (slope && dslope)) {
    GS_REQUIRE(scratch, "gs_pnorm_backward: scratch required (normalisation or slope gradient)");
    const int ppb = pn_pix_per_block(d->pixels);
    const int chunks = (int)((d->pixels + ppb - 1) / ppb);
    sums = scratch + (size_t)d->N * chunks * 4 * d->C;
#define GS_LAUNCH_REDUCE(COLS)                                                                                      \
  hipLaunchKernelGGL((pnorm_bwd_reduce_kernel<COLS>), dim3(chunks, d->N, (k.C8 + COLS - 1) / COLS), dim3(256), 0, \
                     st, k, gp, g2p, yp, mean_rstd, rp, slope, scratch, ppb, chunks)
    // (COLS columns of one pixel are neighbouring 16-B loads: with one column per workgroup a 16- or 32-channel slice was
    // read as every second / fourth 16 bytes of a line by two / four different workgroups)
    if (k.C8 >= 32) GS_LAUNCH_REDUCE(32);
    else if (k.C8 >= 8) GS_LAUNCH_REDUCE(8);
    else if (k.C8 >= 4) GS_LAUNCH_REDUCE(4);
    else if (k.C8 >= 2) GS_LAUNCH_REDUCE(2);
    else GS_LAUNCH_REDUCE(1);
#undef GS_LAUNCH_REDUCE
    GS_CHECK_HIP(hipGetLastError());
    if (d->C < 128 && chunks > 256)
      hipLaunchKernelGGL((pnorm_bwd_finalize_kernel<4>), dim3((d->C + 3) / 4, d->N), dim3(256), 0, st, scratch, sums,
                         chunks, d->C, 1.0f / (float)d->pixels, mean_rstd, slope ? dslope : nullptr, bias_grad);
    else
      hipLaunchKernelGGL((pnorm_bwd_finalize_kernel<16>), dim3((d->C + 15) / 16, d->N), dim3(256), 0, st, scratch, sums,
                         chunks, d->C, 1.0f / (float)d->pixels, mean_rstd, slope ? dslope : nullptr, bias_grad);
    GS_CHECK_HIP(hipGetLastError());
    // slope and bias gradients: per-image totals added in image order (norm.hip), not by atomics
    if (d->N > 1)     // (one image: done by the finalize pass)
      if (int rc = gs_launch_norm_param_grads(sums, 4, mean_rstd, bias_grad, slope ? dslope : nullptr, d->N, d->C,
                                              1.0f / (float)d->pixels, st)) return rc;
  }
  long long bx = ((long long)k.HW * k.C8 + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(pnorm_bwd_apply_kernel, dim3((unsigned)bx, d->N), dim3(256), 0, st, k, gp, g2p, yp, mean_rstd, rp,
                     slope, mean_rstd ? sums : nullptr, static_cast<unsigned short*>(dy),
                     static_cast<unsigned short*>(gres));
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- dst[view] (+)= src[view]: gradient joins of the coupling / skip graph --------------------------------------------
__global__ __launch_bounds__(256) void add_views_kernel(unsigned short* dst, int dcs, int dco, const unsigned short* src,
                                                        int scs, int sco, long long pixels, int C8, int accumulate) {
  const long long total = pixels * C8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long pix = e / C8;
    const int c8 = (int)(e - pix * C8);
    float a[8], b[8];
    pn_unpack8(b, *reinterpret_cast<const uint4*>(src + pix * scs + sco + c8 * 8));
    uint4* o = reinterpret_cast<uint4*>(dst + pix * dcs + dco + c8 * 8);
    if (accumulate) {
      pn_unpack8(a, *o);
#pragma unroll
      for (int k = 0; k < 8; ++k) b[k] += a[k];
    }
    *o = pn_pack8(b);
  }
}

extern "C" int gs_add_views(void* dst, int32_t dst_cs, int32_t dst_co, const void* src, int32_t src_cs, int32_t src_co,
                            int64_t pixels, int32_t C, int32_t accumulate, void* stream) {
  GS_REQUIRE(dst && src && pixels > 0 && C > 0 && ((C | dst_cs | dst_co | src_cs | src_co) & 7) == 0,
             "gs_add_views: bad argument (channel counts / strides / offsets must be multiples of 8)");
  long long bx = (pixels * (C / 8) + 255) / 256;
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(add_views_kernel, dim3((unsigned)bx), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<unsigned short*>(dst), dst_cs, dst_co, static_cast<const unsigned short*>(src), src_cs,
                     src_co, (long long)pixels, C / 8, accumulate);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- adjoint of x.repeat(1, C/Cin, 1, 1, 1) at the image boundary: g_img[n][c0][pix] += sum_{c = c0 mod Cin} g[pix][c] ---
__global__ __launch_bounds__(256) void repeat_bwd_kernel(const unsigned short* g, int cs, int co, float* g_img, int Cin,
                                                         int C, long long pixels) {
  const int n = blockIdx.y;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long long)gridDim.x * 256) {
    const unsigned short* row = g + ((size_t)n * pixels + p) * cs + co;
    for (int c0 = 0; c0 < Cin; ++c0) {
      float s = 0.f;
      for (int c = c0; c < C; c += Cin) s += bf2f(row[c]);
      g_img[((size_t)n * Cin + c0) * pixels + p] += s;
    }
  }
}

extern "C" int gs_repeat_backward(const void* g, int32_t g_cs, int32_t g_co, float* g_img, int32_t N, int32_t Cin,
                                  int32_t C, int64_t pixels, void* stream) {
  GS_REQUIRE(g && g_img && N > 0 && Cin > 0 && C >= Cin && C % Cin == 0 && pixels > 0, "gs_repeat_backward: bad argument");
  long long bx = (pixels + 255) / 256;
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(repeat_bwd_kernel, dim3((unsigned)bx, N), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(g), g_cs, g_co, g_img, Cin, C, (long long)pixels);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- statistics of a channel slice of an activation tensor -----------------------------------------------------------
// Piresnet3D's coupling function starts with an InstanceNorm of its INPUT half (piresnet3d.py:104-108), a tensor that no
// conv epilogue has seen: per (n, pixel chunk) partial sums [N][slots][2][C] in the layout gs_inorm_finalize reads.
// 256 threads = COLS 8-channel columns x 256/COLS pixel rows, 16-B loads.
template <int COLS>
__global__ __launch_bounds__(256) void slice_stats_kernel(const unsigned short* x, unsigned pixels, int cs, int co, int C8,
                                                          int C, float* partial, int pix_per_block, int slots) {
  constexpr int ROWS = 256 / COLS;
  __shared__ float red[ROWS][COLS][17];
  const int n = blockIdx.y, tid = threadIdx.x;
  const int col = tid % COLS, row = tid / COLS;
  const int c8 = blockIdx.z * COLS + col;
  const unsigned p0 = blockIdx.x * pix_per_block;
  const unsigned p1 = min(pixels, p0 + (unsigned)pix_per_block);
  float a[8], q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = q[k] = 0.f;
  if (c8 < C8) {
    for (unsigned px = p0 + row; px < p1; px += ROWS) {
      float v[8];
      pn_view8(v, x, (size_t)n * pixels + px, cs, co, c8);
#pragma unroll
      for (int k = 0; k < 8; ++k) { a[k] += v[k]; q[k] += v[k] * v[k]; }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[row][col][k] = a[k]; red[row][col][8 + k] = q[k]; }
  __syncthreads();
  for (int o = tid; o < COLS * 16; o += 256) {
    const int cc = o / 16, k = o - cc * 16;
    const int ch8 = blockIdx.z * COLS + cc;
    if (ch8 < C8) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) sum += red[r][cc][k];
      partial[(((size_t)n * slots + blockIdx.x) * 2 + (k >> 3)) * C + ch8 * 8 + (k & 7)] = sum;
    }
  }
}

extern "C" int32_t gs_slice_stats_slots(int64_t pixels) {
  long long ppb = (pixels + 255) / 256;
  if (ppb < 256) ppb = 256;
  return (int32_t)((pixels + ppb - 1) / ppb);
}
extern "C" int gs_slice_stats(const void* x, int32_t N, int64_t pixels, int32_t cs, int32_t co, int32_t C, float* partial,
                              void* stream) {
  GS_REQUIRE(x && partial && N > 0 && pixels > 0 && pixels < (1LL << 31) && C > 0 && (C & 7) == 0 && (cs & 7) == 0 &&
                 (co & 7) == 0,
             "gs_slice_stats: bad argument (C, cs, co multiples of 8)");
  const int slots = gs_slice_stats_slots(pixels);
  const int ppb = (int)((pixels + slots - 1) / slots);
  const int C8 = C / 8;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned short* xs = static_cast<const unsigned short*>(x);
  if (C8 >= 8)
    hipLaunchKernelGGL((slice_stats_kernel<8>), dim3(slots, N, (C8 + 7) / 8), dim3(256), 0, st, xs, (unsigned)pixels, cs, co,
                       C8, C, partial, ppb, slots);
  else if (C8 >= 4)
    hipLaunchKernelGGL((slice_stats_kernel<4>), dim3(slots, N, (C8 + 3) / 4), dim3(256), 0, st, xs, (unsigned)pixels, cs, co,
                       C8, C, partial, ppb, slots);
  else if (C8 >= 2)
    hipLaunchKernelGGL((slice_stats_kernel<2>), dim3(slots, N, (C8 + 1) / 2), dim3(256), 0, st, xs, (unsigned)pixels, cs, co,
                       C8, C, partial, ppb, slots);
  else
    hipLaunchKernelGGL((slice_stats_kernel<1>), dim3(slots, N, C8), dim3(256), 0, st, xs, (unsigned)pixels, cs, co, C8, C,
                       partial, ppb, slots);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
