// InstanceNorm (+activation, +residual) forward/backward on NHWC bf16 activations; HBM-bound streaming
// kernels, 16 B (8 channels) per lane access, fp32 statistics.
// Replaces nn.InstanceNorm2d(eps=1e-5, affine=False, track_running_stats=False) + nn.ReLU / nn.LeakyReLU
// (ganslate/nn/utils.py:53-59; resnet2d.py:26-27,36-37,83-87,93; patchgan2d.py:45-46,58-59) and their
// autograd backward, including the adjoint of nn.ReflectionPad2d (the `fold`).
#include "common.hpp"
#include <cstdlib>

// ---- slot reduction: in [N][slots][R][C] -> per (n, c) totals over the slots ---------------------------------------
// grid (ceil(C/CH), N), 256 threads = CH channels x 256/CH slot lanes (CH = 16, or 4 for narrow layers whose volumes
// produce thousands of slots: more workgroups and shorter serial chains).
//  R == 2 (forward): totals are (sum y, sum y^2)  -> out [N][2][C] = (mean, rstd)
//  R == 3 (backward): totals are (sum ghat, sum ghat*yhat, sum yhat) -> out [N][3][C]; optionally the bias gradient of
//           the conv in front of the norm, db[c] += sum_n -rstd * S2 * S3 / hw  (= sum_pixels dy, see norm backward)
template <int R, int CH = 16>
__global__ __launch_bounds__(256) void slot_sum_kernel(const float* in, float* out, int slots, int C, float inv_hw,
                                                       float eps, const float* mean_rstd, float* db) {
  constexpr int LANES = 256 / CH;
  __shared__ double red[R][LANES][CH + 1];
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  const int col = tid % CH, lane = tid / CH;
  const int c = blockIdx.x * CH + col;
  const float* src = in + (size_t)n * slots * R * C;
  // all R sums of a channel in one sweep (R x 4 loads in flight per thread, additions in slot order per lane), then a
  // fixed-shape tree over the lanes: the narrow layers of the V-Nets / Piresnet (8 workgroups of this kernel per launch)
  // spent 8-16 us here walking R serial phases with a 64-term sum on one thread each
  double s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) s[r] = 0.0;
  if (c < C) {
    int sl = lane;
    for (; sl + 3 * LANES < slots; sl += 4 * LANES) {
      float v[4][R];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < R; ++r) v[k][r] = src[((size_t)(sl + k * LANES) * R + r) * C + c];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] += (double)v[k][r];
    }
    for (; sl < slots; sl += LANES)
#pragma unroll
      for (int r = 0; r < R; ++r) s[r] += (double)src[((size_t)sl * R + r) * C + c];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) red[r][lane][col] = s[r];
  __syncthreads();
  for (int w = LANES / 2; w > 0; w >>= 1) {
    if (lane < w) {
#pragma unroll
      for (int r = 0; r < R; ++r) red[r][lane][col] += red[r][lane + w][col];
    }
    __syncthreads();
  }
  if (lane == 0 && c < C) {
    if (R == 2) {
      const double mean = red[0][0][col] * (double)inv_hw;
      double var = red[1][0][col] * (double)inv_hw - mean * mean;
      if (var < 0.0) var = 0.0;
      out[(size_t)n * 2 * C + c] = (float)mean;
      out[(size_t)n * 2 * C + C + c] = (float)(1.0 / sqrt(var + (double)eps));
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) out[((size_t)n * R + r) * C + c] = (float)red[r][0][col];
      if (R == 3 && gridDim.y == 1 && db && mean_rstd) {
        // one image: its totals give the bias gradient right here — norm_param_grads_kernel's arithmetic with its sum over one
        // image (gs_launch_slot_sum3 then skips that launch)
        const float t1 = (float)red[1][0][col], t2 = (float)red[2][0][col];
        db[c] = __fadd_rn(db[c], __fmul_rn(__fmul_rn(__fmul_rn(-mean_rstd[C + c], t1), t2), inv_hw));
      }
    }
  }
}

// Parameter gradients that are sums over the images of per-image totals, added in IMAGE ORDER by one thread per channel
// (they used to be fp32 atomics from the per-image workgroups: order, and with it the last bit, changed from run to run):
//   db[c]     += sum_n -rstd[n][c] * S2[n][c] * S3[n][c] / hw    (bias of the conv in front of a norm: = sum of dy, which is
//                                                                  zero up to rounding — the reference's is rounding noise too)
//   dslope[c] += sum_n S4[n][c]                                   (nn.PReLU slope, R == 4)
__global__ __launch_bounds__(256) void norm_param_grads_kernel(const float* sums, int R, const float* mean_rstd, float* db,
                                                               float* dslope, int N, int C, float inv_hw) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  if (db && mean_rstd) {
    float acc = 0.f;
    for (int n = 0; n < N; ++n)
      acc += -mean_rstd[(size_t)n * 2 * C + C + c] * sums[((size_t)n * R + 1) * C + c] * sums[((size_t)n * R + 2) * C + c] * inv_hw;
    db[c] += acc;
  }
  if (dslope) {
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += sums[((size_t)n * R + 3) * C + c];
    dslope[c] += acc;
  }
}
int gs_launch_norm_param_grads(const float* sums, int R, const float* mean_rstd, float* db, float* dslope, int N, int C,
                               float inv_hw, hipStream_t st) {
  if (!db && !dslope) return 0;
  hipLaunchKernelGGL(norm_param_grads_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, R, mean_rstd, db, dslope, N, C,
                     inv_hw);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_inorm_finalize(const float* partial, int32_t N, int32_t slots, int32_t C, int64_t hw, float eps,
                                 float* mean_rstd, void* stream) {
  GS_REQUIRE(partial && mean_rstd && N > 0 && slots > 0 && C > 0 && hw > 0, "gs_inorm_finalize: bad argument");
  if (slots > 256 && (long long)N * ((C + 15) / 16) < 128)      // few workgroups, long slot loops: 4x the workgroups
    hipLaunchKernelGGL((slot_sum_kernel<2, 4>), dim3((C + 3) / 4, N), dim3(256), 0, static_cast<hipStream_t>(stream),
                       partial, mean_rstd, slots, C, 1.0f / (float)hw, eps, (const float*)nullptr, (float*)nullptr);
  else
    hipLaunchKernelGGL((slot_sum_kernel<2>), dim3((C + 15) / 16, N), dim3(256), 0, static_cast<hipStream_t>(stream),
                       partial, mean_rstd, slots, C, 1.0f / (float)hw, eps, (const float*)nullptr, (float*)nullptr);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- forward apply -----------------------------------------------------------------------------------
__device__ __forceinline__ void load8(float* f, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
// A thread keeps its 8-channel group across iterations (the grid stride is a multiple of C8 whenever C8 divides 256: every
// power-of-two channel count), so mean / rstd are fetched once per thread, and a thread handles several 16-byte elements
// (one element per thread with four 16-byte statistic loads next to it ran the 256^2 x 64 layers at 2.6 TB/s).
template <bool FIXED_C8>
__global__ __launch_bounds__(256) void inorm_act_fwd_kernel(const uint4* y, const float* mean_rstd, const uint4* res,
                                                            uint4* x, long long hw, int C8, int act, float slope) {
  const int n = blockIdx.y;
  const long long per_img = hw * C8;
  const float* mr = mean_rstd + (size_t)n * 2 * C8 * 8;
  const uint4* yn = y + (size_t)n * per_img;
  const uint4* rn = res ? res + (size_t)n * per_img : nullptr;
  uint4* xn = x + (size_t)n * per_img;
  float mu[8], rs[8];
  const long long e0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
  if constexpr (FIXED_C8) {
    const int c8 = (int)(e0 % C8);
    load8(mu, mr + c8 * 8);
    load8(rs, mr + C8 * 8 + c8 * 8);
  }
  auto one = [&](long long e, const uint4 v, const uint4 r) {
    if constexpr (!FIXED_C8) {
      const int c8 = (int)(e % C8);
      load8(mu, mr + c8 * 8);
      load8(rs, mr + C8 * 8 + c8 * 8);
    }
    float f[8] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y), bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w)};
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = apply_act_small((f[k] - mu[k]) * rs[k], act, slope);
    if (rn) {
      f[0] += bf_lo(r.x); f[1] += bf_hi(r.x); f[2] += bf_lo(r.y); f[3] += bf_hi(r.y);
      f[4] += bf_lo(r.z); f[5] += bf_hi(r.z); f[6] += bf_lo(r.w); f[7] += bf_hi(r.w);
    }
    uint4 o;
    o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
    xn[e] = o;
  };
  long long e = e0;
  for (; e + stride < per_img; e += 2 * stride) {          // two elements in flight per thread
    const uint4 v0 = yn[e], v1 = yn[e + stride];
    const uint4 r0 = rn ? rn[e] : uint4{0u, 0u, 0u, 0u}, r1 = rn ? rn[e + stride] : uint4{0u, 0u, 0u, 0u};
    one(e, v0, r0);
    one(e + stride, v1, r1);
  }
  if (e < per_img) one(e, yn[e], rn ? rn[e] : uint4{0u, 0u, 0u, 0u});
}

extern "C" int gs_inorm_act_forward(const void* y, const float* mean_rstd, const void* res, void* x, int32_t N,
                                    int64_t hw, int32_t C, int32_t act, float slope, void* stream) {
  GS_REQUIRE(y && mean_rstd && x && N > 0 && hw > 0 && C > 0 && (C & 7) == 0, "gs_inorm_act_forward: bad argument");
  const long long per_img = hw * (C / 8);
  // ~8 elements of 16 B per thread, at least 2048 workgroups over the batch (8 per CU)
  long long bx = (per_img + 2047) / 2048;
  const long long floor_bx = (2048 + N - 1) / N;
  if (bx < floor_bx) bx = floor_bx;
  if (bx > (per_img + 255) / 256) bx = (per_img + 255) / 256;
  if (bx > 2048) bx = 2048;
  const dim3 grid((unsigned)bx, N);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (256 % (C / 8) == 0)
    hipLaunchKernelGGL(inorm_act_fwd_kernel<true>, grid, dim3(256), 0, st, static_cast<const uint4*>(y), mean_rstd,
                       static_cast<const uint4*>(res), static_cast<uint4*>(x), (long long)hw, C / 8, act, slope);
  else
    hipLaunchKernelGGL(inorm_act_fwd_kernel<false>, grid, dim3(256), 0, st, static_cast<const uint4*>(y), mean_rstd,
                       static_cast<const uint4*>(res), static_cast<uint4*>(x), (long long)hw, C / 8, act, slope);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}


// sum of `slots` partial values in slot order (double accumulation): the loads of eight slots are issued together, the
// additions keep the order — a plain loop waits out one L2 round trip per slot (16 slots: ~6 us of a 21 us launch)
__device__ __forceinline__ double slot_sum_ordered(const float* __restrict__ src, size_t stride, int slots) {
  double s = 0.0;
  int sl = 0;
  for (; sl + 8 <= slots; sl += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(sl + k) * stride];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += (double)v[k];
  }
  for (; sl < slots; ++sl) s += (double)src[(size_t)sl * stride];
  return s;
}

// Workgroup -> (channel group, pixel chunk, image) of the channel-group norm kernels from a 1-D grid. Default: channel group
// fastest, then pixel chunk, then image. xcd_affine (option norm_xcd, N a multiple of 8): workgroup L runs on XCD L % 8
// (observed dispatch order; speed only) and takes an image n with n % 8 == L % 8 — the XCD whose workgroups wrote that image in
// the persistent conv launch in front of this kernel and will read it in the one behind (hconvw.hip: XCD x owns the tiles of
// images x, x + 8, ...), so its lines are served by that XCD's L2 instead of the fabric — the image written LAST first.
__device__ __forceinline__ void norm_block_decode(int& cg, int& chunk, int& n, int CG, int nchunks, int N, int xcd_affine) {
  const int L = blockIdx.x;
  const int per_img = CG * nchunks;
  if (xcd_affine && (N & 7) == 0) {
    const int x = L & 7, j = L >> 3;
    const int k = j / per_img, r = j - k * per_img;
    n = x + 8 * (xcd_affine == 2 ? k : (N / 8 - 1 - k));
    cg = r % CG;
    chunk = r / CG;
  } else {
    cg = L % CG;
    const int t = L / CG;
    chunk = t % nchunks;
    n = t / nchunks;
  }
}

// ---- forward apply with the statistics finalised in the prologue ---------------------------------------------------
// The producing conv leaves per-tile partial sums [N][slots][2][C]; the old path ran slot_sum_kernel (a ~5 us launch of
// 128 tiny workgroups, ~200 of them per training step) and then the apply kernel. Here the apply kernel is laid out by
// channel group — grid (pixel chunks, N, C/64), 256 threads = 8 x 8-channel columns x 32 pixel lanes — so a workgroup
// only needs the totals of ITS 64 channels: it adds up their slots itself (double accumulation, fixed order), and the
// workgroups of the first pixel chunk also publish mean / rstd for the backward pass.
__global__ __launch_bounds__(256) void inorm_stats_act_fwd_kernel(const uint4* y, const float* partial, int slots,
                                                                  float eps, float* mean_rstd, const uint4* res,
                                                                  uint4* x, int hw, int C8, int act, float slope,
                                                                  int pix_per_block, int nchunks, int N, int xcd_affine) {
  __shared__ double tot[2][64];
  __shared__ float mrs[2][64];
  // channel group fastest in dispatch order: the C/64 workgroups that together cover whole 2*C-byte pixel rows of one
  // pixel chunk run side by side (DRAM pages are walked once, not C/64 times at different moments)
  int cg, chunk, n;
  norm_block_decode(cg, chunk, n, C8 / 8, nchunks, N, xcd_affine);
  const int tid = threadIdx.x;
  const int C = C8 * 8;
  // the first two pixels of this thread do not depend on the statistics: their loads go out before the slot sums
  const int cl = tid & 7, lane = tid >> 3;
  const size_t img = (size_t)n * hw * C8;
  const int p0 = chunk * pix_per_block, p1 = min(hw, p0 + pix_per_block);
  const int pxa = p0 + lane, pxb = pxa + 32;
  const size_t ea = img + (size_t)pxa * C8 + cg * 8 + cl, eb = img + (size_t)pxb * C8 + cg * 8 + cl;
  uint4 va = {0, 0, 0, 0}, vb = {0, 0, 0, 0}, ra = {0, 0, 0, 0}, rb = {0, 0, 0, 0};
  if (pxa < p1) { va = y[ea]; if (res) ra = res[ea]; }
  if (pxb < p1) { vb = y[eb]; if (res) rb = res[eb]; }
  if (tid < 128) {
    const int r = tid >> 6, ch = tid & 63;
    tot[r][ch] = slot_sum_ordered(partial + ((size_t)n * slots * 2 + r) * C + cg * 64 + ch, (size_t)2 * C, slots);
  }
  __syncthreads();
  if (tid < 64) {
    const double inv_hw = 1.0 / (double)hw;
    const double mean = tot[0][tid] * inv_hw;
    double var = tot[1][tid] * inv_hw - mean * mean;
    if (var < 0.0) var = 0.0;
    const float m = (float)mean, rs = (float)(1.0 / sqrt(var + (double)eps));
    mrs[0][tid] = m;
    mrs[1][tid] = rs;
    if (chunk == 0) {
      mean_rstd[(size_t)n * 2 * C + cg * 64 + tid] = m;
      mean_rstd[(size_t)n * 2 * C + C + cg * 64 + tid] = rs;
    }
  }
  __syncthreads();
  float mu[8], rs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { mu[k] = mrs[0][cl * 8 + k]; rs[k] = mrs[1][cl * 8 + k]; }
  auto finish = [&](const uint4& v, const uint4& r, size_t e) {
    float f[8] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y), bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w)};
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = apply_act((f[k] - mu[k]) * rs[k], act, slope);
    if (res) {
      f[0] += bf_lo(r.x); f[1] += bf_hi(r.x); f[2] += bf_lo(r.y); f[3] += bf_hi(r.y);
      f[4] += bf_lo(r.z); f[5] += bf_hi(r.z); f[6] += bf_lo(r.w); f[7] += bf_hi(r.w);
    }
    uint4 o;
    o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
    x[e] = o;
  };
  if (pxa < p1) finish(va, ra, ea);
  if (pxb < p1) finish(vb, rb, eb);
  for (int px = pxb + 32; px < p1; px += 32) {
    const size_t e = img + (size_t)px * C8 + cg * 8 + cl;
    const uint4 v = y[e];
    uint4 r = {0, 0, 0, 0};
    if (res) r = res[e];
    finish(v, r, e);
  }
}

extern "C" int gs_inorm_stats_act_forward(const void* y, const float* partial, int32_t slots, float eps,
                                          float* mean_rstd, const void* res, void* x, int32_t N, int64_t hw, int32_t C,
                                          int32_t act, float slope, void* stream) {
  GS_REQUIRE(y && partial && mean_rstd && x && N > 0 && slots > 0 && hw > 0 && C > 0 && (C & 7) == 0,
             "gs_inorm_stats_act_forward: bad argument");
  // every workgroup repeats the slot sum of its 64 channels: only worth it while that prologue is short (the 64 x 64 maps
  // of the residual blocks have 16 slots; a 256 x 256 map has 256-512 and keeps the two-launch form)
  if ((C & 63) != 0 || hw >= (1LL << 31) || slots > 64) {
    if (int rc = gs_inorm_finalize(partial, N, slots, C, hw, eps, mean_rstd, stream)) return rc;
    return gs_inorm_act_forward(y, mean_rstd, res, x, N, hw, C, act, slope, stream);
  }
  // 64 pixels x 64 channels per workgroup: thousands of workgroups for the 64 x 64 maps (a streaming kernel wants the
  // occupancy), while the slot sums (<= 64 slots x 128 floats) stay a short prologue
  int ppb = 64;
  while ((hw + ppb - 1) / ppb > 1024) ppb *= 2;
  const int nchunks = (int)((hw + ppb - 1) / ppb);
  hipLaunchKernelGGL(inorm_stats_act_fwd_kernel, dim3((unsigned)(C / 64) * nchunks * N), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const uint4*>(y), partial, slots, eps, mean_rstd,
                     static_cast<const uint4*>(res), static_cast<uint4*>(x), (int)hw, C / 8, act, slope, ppb, nchunks, N,
                     gs_opt(GS_OPT_NORM_XCD));
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- backward ------------------------------------------------------------------------------------------
// folded gradient: g(n, ih, iw, c8) = sum over the padded-domain positions that the padding maps to (ih, iw)
// reflect: up to 3 source positions per axis; replicate: one contiguous range per axis (the border cell collects its
// `p` pad cells). Kept as two code paths so the index lists stay in registers.
// (scalars + selects, not an indexed array: a dynamically indexed idx[3] went to scratch memory and made the fold
// variants of both backward kernels 2.5x slower than the plain ones)
struct FoldIdx {
  int i0, i1, i2, cnt;
  __device__ __forceinline__ int at(int a) const { return a == 0 ? i0 : (a == 1 ? i1 : i2); }
};
__device__ __forceinline__ FoldIdx fold_sources(int x, int n, int p) {
  FoldIdx f;
  f.i0 = x + p;
  f.i1 = f.i2 = 0;
  f.cnt = 1;
  if (p > 0) {
    const bool lo = x >= 1 && x <= p, hi = x >= n - 1 - p && x <= n - 2;
    const int e1 = p - x, e2 = p + 2 * (n - 1) - x;
    f.i1 = lo ? e1 : e2;
    f.i2 = e2;
    f.cnt = 1 + (lo ? 1 : 0) + (hi ? 1 : 0);
  }
  return f;
}
__device__ __forceinline__ void fold_range(int x, int n, int p, int& lo, int& hi) {   // replicate
  lo = x == 0 ? 0 : x + p;
  hi = x == n - 1 ? n - 1 + 2 * p : x + p;
}

__device__ __forceinline__ void add_bf8(float* f, const uint4 v) {
  f[0] += bf_lo(v.x); f[1] += bf_hi(v.x); f[2] += bf_lo(v.y); f[3] += bf_hi(v.y);
  f[4] += bf_lo(v.z); f[5] += bf_hi(v.z); f[6] += bf_lo(v.w); f[7] += bf_hi(v.w);
}

// px = (iz*H + ih)*W + iw is the unpadded pixel index; the depth axis is only padded when D > 1
// FM (compile-time fold mode): 0 = no fold, 1 = reflect 2-D, 2 = reflect 3-D, 3 = replicate
template <int FM>
__device__ __forceinline__ void load_folded(float* f, const uint4* gpad_n, const uint4* g2_n, int px, int D, int H,
                                            int W, int C8, int c8, int fold) {
#pragma unroll
  for (int k = 0; k < 8; ++k) f[k] = 0.f;
  if constexpr (FM == 0) {
    add_bf8(f, gpad_n[(size_t)px * C8 + c8]);
  } else {
    const int zi = div_small(px, W, 1.0f / (float)W), iw = px - zi * W;
    const int Wp = W + 2 * fold, Hp = H + 2 * fold;
    if constexpr (FM == 1) {                            // the 2-D nets (nn.ReflectionPad2d)
      // only rows / columns 1..fold and n-1-fold..n-2 collect reflected copies: everything else is one load
      const bool plain = (zi < 1 || zi > fold) && (zi < H - 1 - fold || zi > H - 2) &&
                         (iw < 1 || iw > fold) && (iw < W - 1 - fold || iw > W - 2);
      if (plain) {
        add_bf8(f, gpad_n[((size_t)(zi + fold) * Wp + (iw + fold)) * C8 + c8]);
      } else {
        const FoldIdx fh = fold_sources(zi, H, fold);
        const FoldIdx fw = fold_sources(iw, W, fold);
        for (int a = 0; a < fh.cnt; ++a)
          for (int b = 0; b < fw.cnt; ++b) add_bf8(f, gpad_n[((size_t)fh.at(a) * Wp + fw.at(b)) * C8 + c8]);
      }
    } else if constexpr (FM == 2) {
      const int iz = div_small(zi, H, 1.0f / (float)H), ih = zi - iz * H;
      const FoldIdx fd = fold_sources(iz, D, fold);
      const FoldIdx fh = fold_sources(ih, H, fold);
      const FoldIdx fw = fold_sources(iw, W, fold);
      for (int c = 0; c < fd.cnt; ++c)
        for (int a = 0; a < fh.cnt; ++a)
          for (int b = 0; b < fw.cnt; ++b)
            add_bf8(f, gpad_n[(((size_t)fd.at(c) * Hp + fh.at(a)) * Wp + fw.at(b)) * C8 + c8]);
    } else {                                            // replicate (nn.ReplicationPad3d)
      const int iz = div_small(zi, H, 1.0f / (float)H), ih = zi - iz * H;
      const int fd = D > 1 ? fold : 0;
      // only the first and last voxel of an axis collect the replicated border: everything else is one load
      const bool plain = (D == 1 || (iz > 0 && iz < D - 1)) && ih > 0 && ih < H - 1 && iw > 0 && iw < W - 1;
      if (plain) {
        add_bf8(f, gpad_n[(((size_t)(iz + fd) * Hp + (ih + fold)) * Wp + (iw + fold)) * C8 + c8]);
      } else {
        int d0, d1, h0, h1, w0, w1;
        fold_range(iz, D, fd, d0, d1);
        fold_range(ih, H, fold, h0, h1);
        fold_range(iw, W, fold, w0, w1);
        for (int c = d0; c <= d1; ++c)
          for (int a = h0; a <= h1; ++a)
            for (int b = w0; b <= w1; ++b) add_bf8(f, gpad_n[(((size_t)c * Hp + a) * Wp + b) * C8 + c8]);
      }
    }
  }
  if (g2_n) add_bf8(f, g2_n[(size_t)px * C8 + c8]);
}
__device__ __forceinline__ size_t padded_pixels(int D, int H, int W, int fold) {
  return (size_t)(D > 1 ? D + 2 * fold : D) * (H + 2 * fold) * (W + 2 * fold);
}

__device__ __forceinline__ void unpack8(float* f, const uint4 v) {
  f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}

// pass 1: per (n, pixel-chunk, column group) partial sums of ghat and ghat*yhat  -> scratch [N][chunks][2][C]
// 256 threads = COLS 8-channel columns x (256/COLS) pixel lanes; grid (chunks, N, ceil(C8/COLS))
template <int COLS, int FM>
__global__ __launch_bounds__(256) void inorm_bwd_reduce_kernel(const uint4* gpad, const uint4* g2, const uint4* y,
                                                               const float* mean_rstd, float* partial, int D, int H,
                                                               int W, int C8, int fold, int mode, int act,
                                                               float slope, int pix_per_block, int chunks) {
  constexpr int ROWS = 256 / COLS;
  __shared__ float red[ROWS][COLS][25];
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  const int col = tid % COLS, row = tid / COLS;
  const int c8 = blockIdx.z * COLS + col;
  const int HW = D * H * W;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(HW, p0 + pix_per_block);
  const size_t pad_img = padded_pixels(D, H, W, fold) * C8;
  const uint4* gpad_n = gpad + (size_t)n * pad_img;
  const uint4* g2_n = g2 ? g2 + (size_t)n * HW * C8 : nullptr;
  const uint4* y_n = y + (size_t)n * HW * C8;
  const float* mr = mean_rstd + (size_t)n * 2 * C8 * 8;
  float a1[8], a2[8], a3[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a1[k] = a2[k] = a3[k] = 0.f;
  if (c8 < C8) {
    float mu[8], rs[8];
    load8(mu, mr + c8 * 8);
    load8(rs, mr + C8 * 8 + c8 * 8);
    // two pixels per iteration: twice the loads in flight per lane (the kernel is latency-bound at 8 waves per CU)
    int px = p0 + row;
    for (; px + ROWS < p1; px += 2 * ROWS) {
      float g[8], yy[8], g_b[8], yy_b[8];
      const uint4 ya = y_n[(size_t)px * C8 + c8];
      const uint4 yb = y_n[(size_t)(px + ROWS) * C8 + c8];
      load_folded<FM>(g, gpad_n, g2_n, px, D, H, W, C8, c8, fold);
      load_folded<FM>(g_b, gpad_n, g2_n, px + ROWS, D, H, W, C8, c8, fold);
      unpack8(yy, ya);
      unpack8(yy_b, yb);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float yh = (yy[k] - mu[k]) * rs[k];
        const float gh = g[k] * act_grad_from_out(yh, act, slope);
        const float yh2 = (yy_b[k] - mu[k]) * rs[k];
        const float gh2 = g_b[k] * act_grad_from_out(yh2, act, slope);
        a1[k] += gh + gh2;
        a2[k] += gh * yh + gh2 * yh2;
        a3[k] += yh + yh2;
      }
    }
    for (; px < p1; px += ROWS) {
      float g[8], yy[8];
      load_folded<FM>(g, gpad_n, g2_n, px, D, H, W, C8, c8, fold);
      unpack8(yy, y_n[(size_t)px * C8 + c8]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float yh = (yy[k] - mu[k]) * rs[k];
        const float gh = g[k] * act_grad_from_out(yh, act, slope);
        a1[k] += gh;
        a2[k] += gh * yh;
        a3[k] += yh;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[row][col][k] = a1[k]; red[row][col][8 + k] = a2[k]; red[row][col][16 + k] = a3[k]; }
  __syncthreads();
  // COLS*24 outputs, summed over ROWS pixel lanes
  for (int o = tid; o < COLS * 24; o += 256) {
    const int cc = o / 24, k = o - cc * 24;
    const int ch8 = blockIdx.z * COLS + cc;
    if (ch8 < C8) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) sum += red[r][cc][k];
      float* out = partial + ((size_t)n * chunks + blockIdx.x) * 3 * C8 * 8;
      out[(k >> 3) * C8 * 8 + ch8 * 8 + (k & 7)] = sum;
    }
  }
}

// pass 2: dy = rstd * (ghat - S1/hw - yhat*S2/hw) ; optional gsum = folded gradient (before act')
template <int FM>
__global__ __launch_bounds__(256) void inorm_bwd_apply_kernel(const uint4* gpad, const uint4* g2, const uint4* y,
                                                              const float* mean_rstd, const float* sums, uint4* dy,
                                                              uint4* gsum, int D, int H, int W, int C8,
                                                              int c8_shift, int fold, int mode, int act,
                                                              float slope) {
  const int n = blockIdx.y;
  const unsigned HW = (unsigned)(D * H * W);
  const unsigned per_img = HW * (unsigned)C8;
  const size_t pad_img = padded_pixels(D, H, W, fold) * C8;
  const uint4* gpad_n = gpad + (size_t)n * pad_img;
  const uint4* g2_n = g2 ? g2 + (size_t)n * per_img : nullptr;
  const uint4* y_n = y + (size_t)n * per_img;
  uint4* dy_n = dy + (size_t)n * per_img;
  uint4* gs_n = gsum ? gsum + (size_t)n * per_img : nullptr;
  const float inv_hw = 1.0f / (float)HW;
  const float* mr = mean_rstd ? mean_rstd + (size_t)n * 2 * C8 * 8 : nullptr;
  const float* sm = sums ? sums + (size_t)n * 3 * C8 * 8 : nullptr;
  // a thread keeps its channel group across iterations whenever the grid stride is a multiple of C8 (every power-of-two
  // channel count): the four per-channel vectors are then fetched once, not once per element
  int c8h = -1;
  float mu[8], rs[8], s1[8], s2[8];
#pragma unroll 2
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < per_img; e += gridDim.x * 256u) {
    const unsigned px = c8_shift >= 0 ? (e >> c8_shift) : e / (unsigned)C8;
    const int c8 = (int)(e - px * (unsigned)C8);
    float g[8], yy[8], d[8];
    if (mr && c8 != c8h) {
      load8(mu, mr + c8 * 8);
      load8(rs, mr + C8 * 8 + c8 * 8);
      load8(s1, sm + c8 * 8);
      load8(s2, sm + C8 * 8 + c8 * 8);
      c8h = c8;
    }
    if constexpr (FM == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) g[k] = 0.f;
      add_bf8(g, gpad_n[e]);
      if (g2_n) add_bf8(g, g2_n[e]);
    } else {
      load_folded<FM>(g, gpad_n, g2_n, (int)px, D, H, W, C8, c8, fold);
    }
    if (gs_n) {
      uint4 o;
      o.x = pack_bf2(g[0], g[1]); o.y = pack_bf2(g[2], g[3]); o.z = pack_bf2(g[4], g[5]); o.w = pack_bf2(g[6], g[7]);
      gs_n[e] = o;
    }
    unpack8(yy, y_n[e]);
    if (mr) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float yh = (yy[k] - mu[k]) * rs[k];
        const float gh = g[k] * act_grad_from_out(yh, act, slope);
        d[k] = rs[k] * (gh - s1[k] * inv_hw - yh * s2[k] * inv_hw);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = g[k] * act_grad_from_out(yy[k], act, slope);
    }
    uint4 o;
    o.x = pack_bf2(d[0], d[1]); o.y = pack_bf2(d[2], d[3]); o.z = pack_bf2(d[4], d[5]); o.w = pack_bf2(d[6], d[7]);
    dy_n[e] = o;
  }
}


// pass 2 by channel group (C % 64 == 0): grid (pixel chunks, N, C/64), 256 threads = 8 x 8-channel columns x 32 pixel
// lanes. The totals S1 = sum ghat, S2 = sum ghat*yhat (and S3 = sum yhat for the bias gradient) of the workgroup's 64
// channels are summed from the partial slots in the prologue (fixed order): no slot_sum launch between the passes.
template <int FM>
__global__ __launch_bounds__(256) void inorm_bwd_apply_cg_kernel(const uint4* gpad, const uint4* g2, const uint4* y,
                                                                 const float* mean_rstd, const float* partial,
                                                                 int slots, uint4* dy, uint4* gsum, float* sums, int D,
                                                                 int H, int W, int C8, int fold, int act, float slope,
                                                                 int pix_per_block, int nchunks, int N, int xcd_affine) {
  __shared__ float tot[3][64];
  int cg, chunk, n;
  norm_block_decode(cg, chunk, n, C8 / 8, nchunks, N, xcd_affine);     // channel group fastest, see the forward kernel
  const int tid = threadIdx.x;
  const int C = C8 * 8;
  const int HW = D * H * W;
  const float inv_hw = 1.0f / (float)HW;
  const int cl = tid & 7, lane = tid >> 3;
  const int c8 = cg * 8 + cl;
  const size_t per_img = (size_t)HW * C8;
  const size_t pad_img = padded_pixels(D, H, W, fold) * C8;
  const uint4* gpad_n = gpad + (size_t)n * pad_img;
  const uint4* g2_n = g2 ? g2 + (size_t)n * per_img : nullptr;
  const uint4* y_n = y + (size_t)n * per_img;
  uint4* dy_n = dy + (size_t)n * per_img;
  uint4* gs_n = gsum ? gsum + (size_t)n * per_img : nullptr;
  const int p0 = chunk * pix_per_block, p1 = min(HW, p0 + pix_per_block);
  // the first two pixels of this thread do not depend on the totals: their loads go out before the slot sums
  const int pxa = p0 + lane, pxb = pxa + 32;
  float ga[8], gb[8];
  uint4 ya = {0, 0, 0, 0}, yb = {0, 0, 0, 0};
  if (pxa < p1) { load_folded<FM>(ga, gpad_n, g2_n, pxa, D, H, W, C8, c8, fold); ya = y_n[(size_t)pxa * C8 + c8]; }
  if (pxb < p1) { load_folded<FM>(gb, gpad_n, g2_n, pxb, D, H, W, C8, c8, fold); yb = y_n[(size_t)pxb * C8 + c8]; }
  const float* mr = mean_rstd + (size_t)n * 2 * C;
  float mu[8], rs[8], s1[8], s2[8];
  load8(mu, mr + c8 * 8);
  load8(rs, mr + C + c8 * 8);
  if (tid < 192) {
    const int r = tid >> 6, ch = tid & 63;
    tot[r][ch] = (float)slot_sum_ordered(partial + ((size_t)n * slots * 3 + r) * C + cg * 64 + ch, (size_t)3 * C, slots);
  }
  __syncthreads();
  if (sums && chunk == 0 && tid < 192) {
    // per-image totals for the bias gradient of the conv in front of the norm (gs_norm_bias_grads adds them up over the
    // images in order; they used to be fp32 atomics)
    const int r = tid >> 6, ch = tid & 63;
    sums[((size_t)n * 3 + r) * C + cg * 64 + ch] = tot[r][ch];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = tot[0][cl * 8 + k] * inv_hw; s2[k] = tot[1][cl * 8 + k] * inv_hw; }
  auto finish = [&](const float* g, const uint4& yv, int px) {
    const size_t e = (size_t)px * C8 + c8;
    float yy[8], d[8];
    if (gs_n) {
      uint4 o;
      o.x = pack_bf2(g[0], g[1]); o.y = pack_bf2(g[2], g[3]); o.z = pack_bf2(g[4], g[5]); o.w = pack_bf2(g[6], g[7]);
      gs_n[e] = o;
    }
    unpack8(yy, yv);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float yh = (yy[k] - mu[k]) * rs[k];
      d[k] = inorm_dy(g[k], yh, act_grad_from_out(yh, act, slope), s1[k], s2[k], rs[k]);
    }
    uint4 o;
    o.x = pack_bf2(d[0], d[1]); o.y = pack_bf2(d[2], d[3]); o.z = pack_bf2(d[4], d[5]); o.w = pack_bf2(d[6], d[7]);
    dy_n[e] = o;
  };
  if (pxa < p1) finish(ga, ya, pxa);
  if (pxb < p1) finish(gb, yb, pxb);
  for (int px = pxb + 32; px < p1; px += 32) {
    float g[8];
    load_folded<FM>(g, gpad_n, g2_n, px, D, H, W, C8, c8, fold);
    finish(g, y_n[(size_t)px * C8 + c8], px);
  }
}

// shared with norm_ex.hip
int gs_launch_slot_sum3(const float* in, float* out, int N, int slots, int C, float inv_hw, const float* mean_rstd,
                        float* db, hipStream_t st) {
  if (slots > 256 && (long long)N * ((C + 15) / 16) < 128)      // few workgroups, long slot loops: 4x the workgroups
    hipLaunchKernelGGL((slot_sum_kernel<3, 4>), dim3((C + 3) / 4, N), dim3(256), 0, st, in, out, slots, C, inv_hw, 0.f,
                       mean_rstd, db);
  else
    hipLaunchKernelGGL((slot_sum_kernel<3>), dim3((C + 15) / 16, N), dim3(256), 0, st, in, out, slots, C, inv_hw, 0.f,
                       mean_rstd, db);
  GS_CHECK_HIP(hipGetLastError());
  if (N == 1) return 0;      // (one image: the bias gradient was added by the kernel above)
  return gs_launch_norm_param_grads(out, 3, mean_rstd, db, nullptr, N, C, inv_hw, st);
}

// Bias gradients of the convs in front of InstanceNorms for MANY layers in one launch: item i adds
// sum_n -rstd[n][c] * S2[n][c] * S3[n][c] / hw to db_i[c] from the per-image totals gs_inorm_act_backward left at
// scratch + N * slots * 3 * C (block = (256 channels, item)); one thread per channel walks the images in order.
struct NormDbBatch {
  gs_norm_db_item it[GS_NORM_DB_MAX];
};
__global__ __launch_bounds__(256) void norm_bias_grads_kernel(const NormDbBatch b) {
  const gs_norm_db_item& it = b.it[blockIdx.y];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= it.C) return;
  float acc = 0.f;
  for (int n = 0; n < it.N; ++n)
    acc += -it.mean_rstd[(size_t)n * 2 * it.C + it.C + c] * it.sums[((size_t)n * 3 + 1) * it.C + c] *
           it.sums[((size_t)n * 3 + 2) * it.C + c] * it.inv_hw;
  it.db[c] += acc;
}
extern "C" int gs_norm_bias_grads(const gs_norm_db_item* items, int32_t count, void* stream) {
  GS_REQUIRE(items && count > 0, "gs_norm_bias_grads: bad argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int i0 = 0; i0 < count; i0 += GS_NORM_DB_MAX) {
    NormDbBatch b;
    const int n = count - i0 < GS_NORM_DB_MAX ? count - i0 : GS_NORM_DB_MAX;
    int cmax = 0;
    for (int i = 0; i < n; ++i) {
      b.it[i] = items[i0 + i];
      GS_REQUIRE(b.it[i].sums && b.it[i].mean_rstd && b.it[i].db && b.it[i].N > 0 && b.it[i].C > 0,
                 "gs_norm_bias_grads: bad item %d", i0 + i);
      if (b.it[i].C > cmax) cmax = b.it[i].C;
    }
    hipLaunchKernelGGL(norm_bias_grads_kernel, dim3((cmax + 255) / 256, n), dim3(256), 0, st, b);
    GS_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

// pixels per block of the reduction pass: 64 for 2-D sized maps, more for volumes so that the second-level sum
// stays at <= 4096 slots per image
// pixels per workgroup of the reduction pass: about 4096 workgroups over the whole batch, at least 64 pixels each (per image
// it ran the 256^2 x 64 layers of a 16-image twin batch as 16384 workgroups of two pixels per thread: +0.35 % on the step)
static int bwd_pix_per_block(long long pixels, int N) {
  const int forced = gs_opt(GS_OPT_NORM_BWD_PPB);   // tuning aid
  if (forced > 0) return forced;
  long long ppb = (pixels * (N > 0 ? N : 1) + 4095) / 4096;
  if (ppb > pixels) ppb = pixels;
  return ppb < 64 ? 64 : (int)((ppb + 63) / 64 * 64);
}

extern "C" int64_t gs_inorm_backward_scratch_floats(int32_t N, int32_t D, int32_t H, int32_t W, int32_t C) {
  const int64_t pixels = (int64_t)D * H * W;
  const int64_t ppb = bwd_pix_per_block(pixels, N);
  const int64_t chunks = (pixels + ppb - 1) / ppb;
  return (int64_t)N * (chunks + 1) * 3 * C;
}

extern "C" int gs_inorm_act_backward(const void* g_pad, const void* g2, const void* y, const float* mean_rstd,
                                     void* dy, void* gsum, float* scratch, float* bias_grad, int32_t N, int32_t D,
                                     int32_t H, int32_t W, int32_t C, int32_t fold, int32_t fold_mode, int32_t act,
                                     float slope, int32_t pre_slots, void* stream) {
  GS_REQUIRE(g_pad && y && dy && N > 0 && D > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0,
             "gs_inorm_act_backward: bad argument");
  GS_REQUIRE(fold == 0 || fold_mode == GS_BORDER_REFLECT || (fold_mode == GS_BORDER_REPLICATE && fold <= 3),
             "gs_inorm_act_backward: fold must be reflect, or replicate with fold <= 3");
  GS_REQUIRE(fold == 0 || fold_mode != GS_BORDER_REFLECT || (H > 2 * fold && W > 2 * fold && (D == 1 || D > 2 * fold)),
             "gs_inorm_act_backward: fold larger than image");
  GS_REQUIRE((long long)D * H * W < (1LL << 31), "gs_inorm_act_backward: image too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int C8 = C / 8;
  const int HW = D * H * W;
  const int kBwdPixPerBlock = bwd_pix_per_block(HW, N);
  // compile-time fold mode of the kernels: 0 none, 1 reflect 2-D, 2 reflect 3-D, 3 replicate
  const int fm = fold == 0 ? 0 : (fold_mode == GS_BORDER_REFLECT ? (D == 1 ? 1 : 2) : 3);
  float* sums = nullptr;
  if (mean_rstd) {
    GS_REQUIRE(scratch, "gs_inorm_act_backward: scratch required with normalisation");
    const int chunks = pre_slots > 0 ? pre_slots : (HW + kBwdPixPerBlock - 1) / kBwdPixPerBlock;
    sums = scratch + (size_t)N * chunks * 3 * C;
    if (pre_slots <= 0) {
#define GS_LAUNCH_REDUCE2(COLS, FM)                                                                                   \
  hipLaunchKernelGGL((inorm_bwd_reduce_kernel<COLS, FM>), dim3(chunks, N, (C8 + COLS - 1) / COLS), dim3(256), 0, st, \
                     static_cast<const uint4*>(g_pad), static_cast<const uint4*>(g2), static_cast<const uint4*>(y),     \
                     mean_rstd, scratch, D, H, W, C8, fold, fold_mode, act, slope, kBwdPixPerBlock, chunks)
#define GS_LAUNCH_REDUCE(COLS)                                        \
  do {                                                                \
    if (fm == 0) GS_LAUNCH_REDUCE2(COLS, 0);                          \
    else if (fm == 1) GS_LAUNCH_REDUCE2(COLS, 1);                     \
    else if (fm == 2) GS_LAUNCH_REDUCE2(COLS, 2);                     \
    else GS_LAUNCH_REDUCE2(COLS, 3);                                  \
  } while (0)
    if (C8 >= 32) GS_LAUNCH_REDUCE(32);
    else if (C8 >= 8) GS_LAUNCH_REDUCE(8);
    else if (C8 >= 4) GS_LAUNCH_REDUCE(4);      // 32- / 16-channel layers: the columns of a pixel in one workgroup
    else if (C8 >= 2) GS_LAUNCH_REDUCE(2);
    else GS_LAUNCH_REDUCE(1);
#undef GS_LAUNCH_REDUCE
#undef GS_LAUNCH_REDUCE2
    GS_CHECK_HIP(hipGetLastError());
    }
    if ((C & 63) == 0 && chunks <= 64) {
      // wide layers with few slots: the apply pass sums the slots of its own 64 channels in its prologue (no slot_sum
      // launch); with hundreds of slots (256 x 256 maps) that prologue would dominate, so those keep the slot_sum launch
      int ppb = 64;
      while ((HW + ppb - 1) / ppb > 1024) ppb *= 2;
      {
        // one round of workgroups: the kernel keeps 6 workgroups per CU resident (74 registers), so the 2048 workgroups of a
        // residual-block layer at batch 8 ran as 1536 + a third-full second round; 96 pixels per workgroup make it 1376
        // (17.82 -> 17.77 ms on the step)
        static int slots = 0;
        if (!slots) {
          int dev = 0;
          hipDeviceProp_t prop;
          if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            slots = prop.multiProcessorCount * 6;
          else
            slots = 1536;
        }
        const long long per = (long long)(C / 64) * N;
        while (ppb < 256 && per * ((HW + ppb - 1) / ppb) > slots && per * ((HW + ppb + 31) / (ppb + 32)) >= slots / 2) ppb += 32;
      }
#define GS_LAUNCH_APPLY_CG(FM)                                                                                      \
  hipLaunchKernelGGL((inorm_bwd_apply_cg_kernel<FM>), dim3((unsigned)(C / 64) * (unsigned)((HW + ppb - 1) / ppb) * N), dim3(256), 0, \
                     st, static_cast<const uint4*>(g_pad), static_cast<const uint4*>(g2),                            \
                     static_cast<const uint4*>(y), mean_rstd, scratch, chunks, static_cast<uint4*>(dy),              \
                     static_cast<uint4*>(gsum), sums, D, H, W, C8, fold, act, slope, ppb, (int)((HW + ppb - 1) / ppb), N, \
                     gs_opt(GS_OPT_NORM_XCD))
      if (fm == 0) GS_LAUNCH_APPLY_CG(0);
      else if (fm == 1) GS_LAUNCH_APPLY_CG(1);
      else if (fm == 2) GS_LAUNCH_APPLY_CG(2);
      else GS_LAUNCH_APPLY_CG(3);
#undef GS_LAUNCH_APPLY_CG
      GS_CHECK_HIP(hipGetLastError());
      // per-image totals are at `sums` now; the bias gradient, if asked for here, is one more tiny launch (executors
      // pass bias_grad = NULL and batch all their norm layers into one gs_norm_bias_grads call)
      return gs_launch_norm_param_grads(sums, 3, mean_rstd, bias_grad, nullptr, N, C, 1.0f / (float)HW, st);
    }
    if (int rc = gs_launch_slot_sum3(scratch, sums, N, chunks, C, 1.0f / (float)HW, mean_rstd, bias_grad, st)) return rc;
  }
  const long long per_img = (long long)HW * C8;
  GS_REQUIRE(per_img < (1LL << 31), "gs_inorm_act_backward: image too large");
  const int apply_u = gs_opt(GS_OPT_NORM_APPLY_UNROLL);   // elements per thread
  long long bx = (per_img + 256LL * apply_u - 1) / (256LL * apply_u);
  if (bx > 1024) bx = 1024;
  if (bx < 1) bx = 1;
  int c8_shift = -1;
  if ((C8 & (C8 - 1)) == 0) { c8_shift = 0; while ((1 << c8_shift) < C8) ++c8_shift; }
#define GS_LAUNCH_APPLY(FM)                                                                                        \
  hipLaunchKernelGGL((inorm_bwd_apply_kernel<FM>), dim3((unsigned)bx, N), dim3(256), 0, st,                         \
                     static_cast<const uint4*>(g_pad), static_cast<const uint4*>(g2), static_cast<const uint4*>(y), \
                     mean_rstd, sums, static_cast<uint4*>(dy), static_cast<uint4*>(gsum), D, H, W, C8, c8_shift,    \
                     fold, fold_mode, act, slope)
  if (fm == 0) GS_LAUNCH_APPLY(0);
  else if (fm == 1) GS_LAUNCH_APPLY(1);
  else if (fm == 2) GS_LAUNCH_APPLY(2);
  else GS_LAUNCH_APPLY(3);
#undef GS_LAUNCH_APPLY
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
