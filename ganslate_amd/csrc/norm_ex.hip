// Generalised InstanceNorm / activation kernels for skip-connection graphs (U-Net, ganslate/nn/generators/unet/
// unet2d.py:110-157): one normalised tensor is consumed through TWO activations (LeakyReLU(0.2) by the next
// down-conv, ReLU by the up-conv that reads the skip half of torch.cat([x, y], 1)), tensors live in channel slices of
// wider concat buffers, and nn.Dropout(0.5) sits between the norm and the consumer's ReLU (unet2d.py:146-147).
//   forward : v = drop(norm(y));  x1 = act1(v) -> slice of buffer 1;  x2 = act2(v) -> slice of buffer 2 (optional)
//   backward: ghat = mask*scale*(g1*act1'(yhat) + g2*act2'(yhat));  dy = rstd*(ghat - mean ghat - yhat*mean(ghat*yhat))
// With mean_rstd == NULL there is no norm and `y` holds a sign-preserving activation output (conv epilogue LeakyReLU).
// HBM-bound streaming kernels: 16 B per lane, fp32 math. Dropout masks come from a counter-based hash of
// (seed, image, element) so the backward pass regenerates them instead of storing them.
#include "common.hpp"

struct NormExK {
  gs_norm_ex_desc d;
  int C8;
  unsigned HW;
};

__device__ __forceinline__ unsigned gs_hash32(unsigned x) {   // murmur3 finaliser
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
// keep-mask * 1/(1-p) for element (n, idx) — idx counts scalars of one image in NHWC order
__device__ __forceinline__ float drop_scale(const gs_norm_ex_desc& d, int n, unsigned idx) {
  if (d.drop_p <= 0.f) return 1.f;
  // 64-bit seed = host part (+ device part: a captured step replays this launch with the same arguments, so what
  // changes per iteration has to be read from memory — unet2d.py:146 draws a new mask every forward)
  unsigned long long seed = ((unsigned long long)d.seed_hi << 32) | d.seed_lo;
  if (d.seed_dev) seed += ((unsigned long long)d.seed_dev[1] << 32) | d.seed_dev[0];
  const unsigned lo = (unsigned)seed, hi = (unsigned)(seed >> 32);
  const unsigned h = gs_hash32(idx ^ gs_hash32(lo + 0x9e3779b9u * (unsigned)(n + 1)) ^ hi);
  const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  return u >= d.drop_p ? 1.0f / (1.0f - d.drop_p) : 0.f;
}

__device__ __forceinline__ void ex_load8(float* f, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void ex_unpack8(float* f, const uint4 v) {
  f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 ex_pack8(const float* f) {
  uint4 o;
  o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
  return o;
}

__global__ __launch_bounds__(256) void norm_ex_fwd_kernel(const NormExK p, const uint4* y, const float* mean_rstd,
                                                          unsigned short* x1, unsigned short* x2) {
  const gs_norm_ex_desc& d = p.d;
  const int n = blockIdx.y;
  const unsigned per_img = p.HW * (unsigned)p.C8;
  const uint4* yn = y + (size_t)n * per_img;
  const float* mr = mean_rstd ? mean_rstd + (size_t)n * 2 * d.C : nullptr;
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < per_img; e += gridDim.x * 256u) {
    const unsigned px = e / (unsigned)p.C8;
    const int c8 = (int)(e - px * (unsigned)p.C8);
    float v[8], a[8];
    ex_unpack8(v, yn[e]);
    if (mr) {
      float mu[8], rs[8];
      ex_load8(mu, mr + c8 * 8);
      ex_load8(rs, mr + d.C + c8 * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (v[k] - mu[k]) * rs[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= drop_scale(d, n, e * 8u + k);
    const size_t pix = (size_t)n * p.HW + px;
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = apply_act(v[k], d.act1, d.slope);
    *reinterpret_cast<uint4*>(x1 + pix * d.x1_cs + d.x1_co + c8 * 8) = ex_pack8(a);
    if (x2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = apply_act(v[k], d.act2, d.slope);
      *reinterpret_cast<uint4*>(x2 + pix * d.x2_cs + d.x2_co + c8 * 8) = ex_pack8(a);
    }
  }
}

// ghat for one 8-channel group of one pixel. The activations of these graphs are none / ReLU / LeakyReLU: act'(yhat) is then a
// select between 1 and a per-launch constant (0, the slope, 1) — as a run-time switch per element the compiler turned the two
// activations of eight channels into a chain of scalar branches that nothing could be scheduled across (the reduction pass
// took its pixels one memory round trip at a time). tanh keeps the general form.
__device__ __forceinline__ float ex_neg(int act, float slope) {
  return act == GS_ACT_RELU ? 0.f : (act == GS_ACT_LRELU ? slope : 1.f);
}
__device__ __forceinline__ void ex_ghat_load(const NormExK& p, int n, unsigned px, int c8, const unsigned short* g1,
                                             const unsigned short* g2, uint4& av, uint4& bv) {
  const gs_norm_ex_desc& d = p.d;
  const size_t pix = (size_t)n * p.HW + px;
  av = *reinterpret_cast<const uint4*>(g1 + pix * d.g1_cs + d.g1_co + c8 * 8);
  if (g2) bv = *reinterpret_cast<const uint4*>(g2 + pix * d.g2_cs + d.g2_co + c8 * 8);
}
__device__ __forceinline__ void ex_ghat_from(const NormExK& p, int n, unsigned px, int c8, const uint4& av, const uint4& bv,
                                             bool g2, const float* yh, float* gh) {
  const gs_norm_ex_desc& d = p.d;
  float a[8], b[8];
  ex_unpack8(a, av);
  if (g2) ex_unpack8(b, bv);
  const unsigned e8 = (px * (unsigned)p.C8 + (unsigned)c8) * 8u;
  float t[8];
  if (d.act1 != GS_ACT_TANH && d.act2 != GS_ACT_TANH) {      // (uniform)
    const float n1 = ex_neg(d.act1, d.slope), n2 = ex_neg(d.act2, d.slope);
    if (g2) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        t[k] = __fadd_rn(__fmul_rn(a[k], yh[k] > 0.f ? 1.f : n1), __fmul_rn(b[k], yh[k] > 0.f ? 1.f : n2));
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = __fmul_rn(a[k], yh[k] > 0.f ? 1.f : n1);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      t[k] = __fmul_rn(a[k], act_grad_from_out(yh[k], d.act1, d.slope));
      if (g2) t[k] = __fadd_rn(t[k], __fmul_rn(b[k], act_grad_from_out(yh[k], d.act2, d.slope)));
    }
  }
  if (d.drop_p <= 0.f) {                                      // (uniform)
#pragma unroll
    for (int k = 0; k < 8; ++k) gh[k] = t[k];
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) gh[k] = t[k] * drop_scale(d, n, e8 + k);
  }
}
__device__ __forceinline__ void ex_ghat(const NormExK& p, int n, unsigned px, int c8, const unsigned short* g1,
                                        const unsigned short* g2, const float* yh, float* gh) {
  uint4 av, bv = {0u, 0u, 0u, 0u};
  ex_ghat_load(p, n, px, c8, g1, g2, av, bv);
  ex_ghat_from(p, n, px, c8, av, bv, g2 != nullptr, yh, gh);
}

template <int COLS>
__global__ __launch_bounds__(256) void norm_ex_bwd_reduce_kernel(const NormExK p, const unsigned short* g1,
                                                                 const unsigned short* g2, const uint4* y,
                                                                 const float* mean_rstd, float* partial,
                                                                 int pix_per_block, int chunks) {
  constexpr int ROWS = 256 / COLS;
  __shared__ float red[ROWS][COLS][25];
  const gs_norm_ex_desc& d = p.d;
  const int n = blockIdx.y, tid = threadIdx.x;
  const int col = tid % COLS, row = tid / COLS;
  const int c8 = blockIdx.z * COLS + col;
  const unsigned p0 = blockIdx.x * pix_per_block;
  const unsigned p1 = min(p.HW, p0 + pix_per_block);
  const uint4* y_n = y + (size_t)n * p.HW * p.C8;
  const float* mr = mean_rstd + (size_t)n * 2 * d.C;
  float a1[8], a2[8], a3[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a1[k] = a2[k] = a3[k] = 0.f;
  if (c8 < p.C8) {
    float mu[8], rs[8];
    ex_load8(mu, mr + c8 * 8);
    ex_load8(rs, mr + d.C + c8 * 8);
    // Four pixels' operands are requested before the first is used: taken one at a time, the 16-workgroup launches of the inner
    // U-Net levels spent 13-17 us on eight serial trips to memory. (Sums in pixel order, as before.)
    constexpr int U = 4;
    for (unsigned pb = p0 + row; pb < p1; pb += ROWS * U) {
      uint4 yv[U], av[U], bv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned px = pb + u * ROWS;
        const unsigned pc = px < p1 ? px : pb;               // (a valid pixel: loaded, not used)
        yv[u] = y_n[(size_t)pc * p.C8 + c8];
        bv[u] = uint4{0u, 0u, 0u, 0u};
        ex_ghat_load(p, n, pc, c8, g1, g2, av[u], bv[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned px = pb + u * ROWS;
        if (px < p1) {
          float yh[8], gh[8];
          ex_unpack8(yh, yv[u]);
#pragma unroll
          for (int k = 0; k < 8; ++k) yh[k] = (yh[k] - mu[k]) * rs[k];
          ex_ghat_from(p, n, px, c8, av[u], bv[u], g2 != nullptr, yh, gh);
#pragma unroll
          for (int k = 0; k < 8; ++k) { a1[k] += gh[k]; a2[k] += gh[k] * yh[k]; a3[k] += yh[k]; }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[row][col][k] = a1[k]; red[row][col][8 + k] = a2[k]; red[row][col][16 + k] = a3[k]; }
  __syncthreads();
  for (int o = tid; o < COLS * 24; o += 256) {
    const int cc = o / 24, k = o - cc * 24;
    const int ch8 = blockIdx.z * COLS + cc;
    if (ch8 < p.C8) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) sum += red[r][cc][k];
      float* out = partial + ((size_t)n * chunks + blockIdx.x) * 3 * d.C;
      out[(k >> 3) * d.C + ch8 * 8 + (k & 7)] = sum;
    }
  }
}

__global__ __launch_bounds__(256) void norm_ex_bwd_apply_kernel(const NormExK p, const unsigned short* g1,
                                                                const unsigned short* g2, const uint4* y,
                                                                const float* mean_rstd, const float* sums, uint4* dy) {
  const gs_norm_ex_desc& d = p.d;
  const int n = blockIdx.y;
  const unsigned per_img = p.HW * (unsigned)p.C8;
  const uint4* y_n = y + (size_t)n * per_img;
  uint4* dy_n = dy + (size_t)n * per_img;
  const float inv_hw = 1.0f / (float)p.HW;
  const float* mr = mean_rstd ? mean_rstd + (size_t)n * 2 * d.C : nullptr;
  const float* sm = sums ? sums + (size_t)n * 3 * d.C : nullptr;
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < per_img; e += gridDim.x * 256u) {
    const unsigned px = e / (unsigned)p.C8;
    const int c8 = (int)(e - px * (unsigned)p.C8);
    float yh[8], gh[8], o[8];
    ex_unpack8(yh, y_n[e]);
    if (mr) {
      float mu[8], rs[8], s1[8], s2[8];
      ex_load8(mu, mr + c8 * 8);
      ex_load8(rs, mr + d.C + c8 * 8);
      ex_load8(s1, sm + c8 * 8);
      ex_load8(s2, sm + d.C + c8 * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) yh[k] = (yh[k] - mu[k]) * rs[k];
      ex_ghat(p, n, px, c8, g1, g2, yh, gh);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = rs[k] * (gh[k] - s1[k] * inv_hw - yh[k] * s2[k] * inv_hw);
    } else {
      ex_ghat(p, n, px, c8, g1, g2, yh, o);
    }
    dy_n[e] = ex_pack8(o);
  }
}

// defined in norm.hip
int gs_launch_slot_sum3(const float* in, float* out, int N, int slots, int C, float inv_hw, const float* mean_rstd,
                        float* db, hipStream_t st);

static const int kExPixPerBlock = 64;

static int check_desc(const gs_norm_ex_desc* d, const char* who) {
  GS_REQUIRE(d && d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && (d->C & 7) == 0, "%s: bad shape", who);
  GS_REQUIRE((long long)d->H * d->W * (d->C / 8) < (1LL << 28), "%s: image too large", who);
  GS_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "%s: dropout probability out of range", who);
  return 0;
}

extern "C" int gs_norm_act_forward_ex(const gs_norm_ex_desc* d, const void* y, const float* mean_rstd, void* x1,
                                      void* x2, void* stream) {
  if (int rc = check_desc(d, "gs_norm_act_forward_ex")) return rc;
  GS_REQUIRE(y && x1, "gs_norm_act_forward_ex: null tensor");
  GS_REQUIRE((d->x1_cs & 7) == 0 && (d->x1_co & 7) == 0 && (!x2 || ((d->x2_cs & 7) == 0 && (d->x2_co & 7) == 0)),
             "gs_norm_act_forward_ex: channel strides/offsets must be multiples of 8");
  NormExK k;
  k.d = *d; k.C8 = d->C / 8; k.HW = (unsigned)(d->H * d->W);
  long long bx = ((long long)k.HW * k.C8 + 255) / 256;
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(norm_ex_fwd_kernel, dim3((unsigned)bx, d->N), dim3(256), 0, static_cast<hipStream_t>(stream), k,
                     static_cast<const uint4*>(y), mean_rstd, static_cast<unsigned short*>(x1),
                     static_cast<unsigned short*>(x2));
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int64_t gs_norm_backward_ex_scratch_floats(const gs_norm_ex_desc* d) {
  const int64_t chunks = ((int64_t)d->H * d->W + kExPixPerBlock - 1) / kExPixPerBlock;
  return (int64_t)d->N * (chunks + 1) * 3 * d->C;
}

extern "C" int gs_norm_act_backward_ex(const gs_norm_ex_desc* d, const void* g1, const void* g2, const void* y,
                                       const float* mean_rstd, void* dy, float* scratch, float* bias_grad,
                                       void* stream) {
  if (int rc = check_desc(d, "gs_norm_act_backward_ex")) return rc;
  GS_REQUIRE(g1 && y && dy, "gs_norm_act_backward_ex: null tensor");
  GS_REQUIRE((d->g1_cs & 7) == 0 && (d->g1_co & 7) == 0 && (!g2 || ((d->g2_cs & 7) == 0 && (d->g2_co & 7) == 0)),
             "gs_norm_act_backward_ex: channel strides/offsets must be multiples of 8");
  hipStream_t st = static_cast<hipStream_t>(stream);
  NormExK k;
  k.d = *d; k.C8 = d->C / 8; k.HW = (unsigned)(d->H * d->W);
  const unsigned short* a = static_cast<const unsigned short*>(g1);
  const unsigned short* b = static_cast<const unsigned short*>(g2);
  float* sums = nullptr;
  if (mean_rstd) {
    GS_REQUIRE(scratch, "gs_norm_act_backward_ex: scratch required with normalisation");
    const int chunks = (int)((k.HW + kExPixPerBlock - 1) / kExPixPerBlock);
    sums = scratch + (size_t)d->N * chunks * 3 * d->C;
#define GS_LAUNCH_REDUCE(COLS)                                                                                       \
  hipLaunchKernelGGL((norm_ex_bwd_reduce_kernel<COLS>), dim3(chunks, d->N, (k.C8 + COLS - 1) / COLS), dim3(256), 0, \
                     st, k, a, b, static_cast<const uint4*>(y), mean_rstd, scratch, kExPixPerBlock, chunks)
    if (k.C8 >= 32) GS_LAUNCH_REDUCE(32);
    else if (k.C8 >= 8) GS_LAUNCH_REDUCE(8);
    else GS_LAUNCH_REDUCE(1);
#undef GS_LAUNCH_REDUCE
    GS_CHECK_HIP(hipGetLastError());
    if (int rc = gs_launch_slot_sum3(scratch, sums, d->N, chunks, d->C, 1.0f / (float)k.HW, mean_rstd, bias_grad, st))
      return rc;
  }
  long long bx = ((long long)k.HW * k.C8 + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(norm_ex_bwd_apply_kernel, dim3((unsigned)bx, d->N), dim3(256), 0, st, k, a, b,
                     static_cast<const uint4*>(y), mean_rstd, sums, static_cast<uint4*>(dy));
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
