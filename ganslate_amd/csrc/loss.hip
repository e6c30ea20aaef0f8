// Loss / metric reductions on the boundary tensors (fp32): wavefront shuffle reduction -> workgroup -> a
// fixed-order final sum by the last-arriving workgroup (deterministic, single launch).
// Replaces nn.MSELoss vs an expanded constant target (ganslate/nn/losses/adversarial_loss.py:28-29,60-62),
// nn.L1Loss (cyclegan_losses.py:64,75,97-101; pix2pix_losses.py:15-19), tensor.mean() of
// utils/metrics/train_metrics.py:27-33 and SSIMLoss (nn/losses/utils/ssim.py:65-99).
#include "common.hpp"

float* gs_reduce_workspace(void* stream);   // 1024 floats + 1 counter, one per launching stream (api.hip)

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// publish this workgroup's partial; the last workgroup to arrive sums all partials in index order
__device__ __forceinline__ void finish_reduction(float partial, float* ws, float scale, float* out) {
  __shared__ int last;
  unsigned* counter = reinterpret_cast<unsigned*>(ws + 1024);
  if (threadIdx.x == 0) {
    __hip_atomic_store(ws + blockIdx.x, partial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (t == gridDim.x - 1);
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (last) {
    // all 256 threads of the last workgroup: thread t adds partials t, t+256, ... in index order, then a fixed-shape
    // tree over the threads — the same association every run (one thread walking up to 1024 partials cost 90 us)
    __shared__ double tree[256];
    double s = 0.0;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 256)
      s += (double)__hip_atomic_load(ws + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tree[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) tree[threadIdx.x] += tree[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      out[0] = (float)(tree[0] * (double)scale);
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(256) void mse_const_kernel(const float* x, long long n, float target, float* ws,
                                                        float* loss) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float d = x[i] - target;
    s += d * d;
  }
  s = block_sum(s, sh);
  finish_reduction(s, ws, 1.0f / (float)n, loss);
}
__global__ __launch_bounds__(256) void mse_const_grad_kernel(const float* x, long long n, float target, float* grad,
                                                             const float* gscale) {
  const float k = (gscale ? gscale[0] : 1.f) * 2.0f / (float)n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    grad[i] = k * (x[i] - target);
}

// The other GAN objectives of AdversarialLoss (adversarial_loss.py:26-34,60-73) as one pointwise function per mode:
//   vanilla       nn.BCEWithLogitsLoss vs the expanded label t: f = max(x,0) - x t + log1p(exp(-|x|)), f' = sigmoid(x) - t
//   wgangp        -mean(x) for real, +mean(x) for fake
//   nonsaturating softplus(-x) for real / softplus(x) for fake, averaged PER SAMPLE (a vector of `rows` losses); the
//                 reference's branch raises NameError (F is never imported, :68-73) — F.softplus' definition (threshold 20)
enum { ADV_LSGAN = 0, ADV_VANILLA = 1, ADV_WGANGP = 2, ADV_NONSAT = 3 };
template <int MODE>
__device__ __forceinline__ float adv_point(float x, float label, float sgn) {
  if constexpr (MODE == ADV_LSGAN) { const float d = x - label; return d * d; }
  if constexpr (MODE == ADV_VANILLA) return fmaxf(x, 0.f) - x * label + log1pf(expf(-fabsf(x)));
  if constexpr (MODE == ADV_WGANGP) return sgn * x;
  const float z = sgn * x;
  return z > 20.f ? z : log1pf(expf(z));
}
template <int MODE>
__device__ __forceinline__ float adv_point_grad(float x, float label, float sgn) {
  if constexpr (MODE == ADV_LSGAN) return 2.f * (x - label);
  if constexpr (MODE == ADV_VANILLA) return 1.f / (1.f + expf(-x)) - label;
  if constexpr (MODE == ADV_WGANGP) return sgn;
  const float z = sgn * x;
  return z > 20.f ? sgn : sgn / (1.f + expf(-z));
}
template <int MODE>
__global__ __launch_bounds__(256) void adv_loss_kernel(const float* x, long long n, float label, float sgn, float* ws,
                                                       float* loss) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    s += adv_point<MODE>(x[i], label, sgn);
  s = block_sum(s, sh);
  finish_reduction(s, ws, 1.0f / (float)n, loss);
}
template <int MODE>
__global__ __launch_bounds__(256) void adv_loss_grad_kernel(const float* x, long long n, float label, float sgn,
                                                            float* grad, const float* gscale) {
  const float k = (gscale ? gscale[0] : 1.f) / (float)n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    grad[i] = k * adv_point_grad<MODE>(x[i], label, sgn);
}
// nonsaturating: one workgroup per sample, fixed-shape sum
__global__ __launch_bounds__(256) void adv_rows_kernel(const float* x, long long per, float sgn, float* loss) {
  __shared__ float sh[4];
  const float* xr = x + (long long)blockIdx.x * per;
  float s = 0.f;
  for (long long i = threadIdx.x; i < per; i += blockDim.x) s += adv_point<ADV_NONSAT>(xr[i], 0.f, sgn);
  s = block_sum(s, sh);
  if (threadIdx.x == 0) loss[blockIdx.x] = s / (float)per;
}
__global__ __launch_bounds__(256) void adv_rows_grad_kernel(const float* x, long long per, float sgn, float* grad,
                                                            const float* gscale) {
  const long long base = (long long)blockIdx.x * per;
  const float k = (gscale ? gscale[blockIdx.x] : 1.f) / (float)per;
  for (long long i = threadIdx.x; i < per; i += blockDim.x)
    grad[base + i] = k * adv_point_grad<ADV_NONSAT>(x[base + i], 0.f, sgn);
}

__global__ __launch_bounds__(256) void l1_kernel(const float* a, const float* b, long long n, float* ws, float* loss) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    s += fabsf(a[i] - b[i]);
  s = block_sum(s, sh);
  finish_reduction(s, ws, 1.0f / (float)n, loss);
}
__global__ __launch_bounds__(256) void l1_grad_kernel(const float* a, const float* b, long long n, float* grad,
                                                      const float* gscale) {
  const float k = (gscale ? gscale[0] : 1.f) / (float)n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    grad[i] = d > 0.f ? k : (d < 0.f ? -k : 0.f);
  }
}
__global__ __launch_bounds__(256) void mean_kernel(const float* x, long long n, float* ws, float* out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    s += x[i];
  s = block_sum(s, sh);
  finish_reduction(s, ws, 1.0f / (float)n, out);
}

static inline unsigned red_blocks(long long n) {
  long long b = (n + 256 * 8 - 1) / (256 * 8);
  if (b < 1) b = 1;
  if (b > 1024) b = 1024;
  return (unsigned)b;
}

extern "C" int gs_mse_const(const float* x, int64_t n, float target, float* loss, float* grad,
                            const float* grad_scale, void* stream) {
  GS_REQUIRE(x && n > 0 && (loss || grad), "gs_mse_const: bad argument");
  float* ws = gs_reduce_workspace(stream);
  if (!ws) { if (!gs_zero_page()) gs_set_error("gs_mse_const: library not initialised (call gs_init)"); return 2; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (loss) hipLaunchKernelGGL(mse_const_kernel, dim3(red_blocks(n)), dim3(256), 0, st, x, (long long)n, target, ws, loss);
  if (grad) hipLaunchKernelGGL(mse_const_grad_kernel, dim3(red_blocks(n)), dim3(256), 0, st, x, (long long)n, target, grad, grad_scale);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
template <int MODE>
static void adv_launch(const float* x, long long n, float label, float sgn, float* ws, float* loss, float* grad,
                       const float* gscale, hipStream_t st) {
  if (loss) hipLaunchKernelGGL(adv_loss_kernel<MODE>, dim3(red_blocks(n)), dim3(256), 0, st, x, n, label, sgn, ws, loss);
  if (grad) hipLaunchKernelGGL(adv_loss_grad_kernel<MODE>, dim3(red_blocks(n)), dim3(256), 0, st, x, n, label, sgn, grad, gscale);
}
extern "C" int gs_adv_loss(const float* x, int64_t n, int32_t rows, int32_t mode, int32_t target_is_real, float label,
                           float* loss, float* grad, const float* grad_scale, void* stream) {
  GS_REQUIRE(x && n > 0 && (loss || grad) && mode >= ADV_LSGAN && mode <= ADV_NONSAT, "gs_adv_loss: bad argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float sgn = target_is_real ? -1.f : 1.f;
  if (mode == ADV_NONSAT) {
    GS_REQUIRE(rows > 0 && n % rows == 0, "gs_adv_loss: nonsaturating needs n divisible by rows (the batch)");
    if (loss) hipLaunchKernelGGL(adv_rows_kernel, dim3(rows), dim3(256), 0, st, x, (long long)(n / rows), sgn, loss);
    if (grad) hipLaunchKernelGGL(adv_rows_grad_kernel, dim3(rows), dim3(256), 0, st, x, (long long)(n / rows), sgn, grad, grad_scale);
    GS_CHECK_HIP(hipGetLastError());
    return 0;
  }
  float* ws = gs_reduce_workspace(stream);
  if (!ws) { if (!gs_zero_page()) gs_set_error("gs_adv_loss: library not initialised (call gs_init)"); return 2; }
  if (mode == ADV_LSGAN) adv_launch<ADV_LSGAN>(x, n, label, sgn, ws, loss, grad, grad_scale, st);
  else if (mode == ADV_VANILLA) adv_launch<ADV_VANILLA>(x, n, label, sgn, ws, loss, grad, grad_scale, st);
  else adv_launch<ADV_WGANGP>(x, n, label, sgn, ws, loss, grad, grad_scale, st);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int gs_l1(const float* a, const float* b, int64_t n, float* loss, float* grad_a, const float* grad_scale,
                     void* stream) {
  GS_REQUIRE(a && b && n > 0 && (loss || grad_a), "gs_l1: bad argument");
  float* ws = gs_reduce_workspace(stream);
  if (!ws) { if (!gs_zero_page()) gs_set_error("gs_l1: library not initialised (call gs_init)"); return 2; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (loss) hipLaunchKernelGGL(l1_kernel, dim3(red_blocks(n)), dim3(256), 0, st, a, b, (long long)n, ws, loss);
  if (grad_a) hipLaunchKernelGGL(l1_grad_kernel, dim3(red_blocks(n)), dim3(256), 0, st, a, b, (long long)n, grad_a, grad_scale);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int gs_mean(const float* x, int64_t n, float* out, void* stream) {
  GS_REQUIRE(x && out && n > 0, "gs_mean: bad argument");
  float* ws = gs_reduce_workspace(stream);
  if (!ws) { if (!gs_zero_page()) gs_set_error("gs_mean: library not initialised (call gs_init)"); return 2; }
  hipLaunchKernelGGL(mean_kernel, dim3(red_blocks(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long long)n, ws, out);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- SSIM distance -----------------------------------------------------------------------------------------
// One workgroup = one 16x32 tile of the valid (H-10)x(W-10) output of one plane. Horizontal 11-tap Gaussian of
// the five maps X, Y, XX, YY, XY into LDS, then the vertical pass, S1/S2, sqrt(relu(2-S1-S2)).
#define SSIM_TH 16
#define SSIM_TW 32
__constant__ float c_gauss[11];

__global__ __launch_bounds__(256) void ssim_kernel(const float* x, const float* y, int H, int W, int tiles_w,
                                                   int tiles_h, float* partial) {
  __shared__ float sx[SSIM_TH + 10][SSIM_TW + 10];
  __shared__ float sy[SSIM_TH + 10][SSIM_TW + 10];
  __shared__ float hz[5][SSIM_TH + 10][SSIM_TW];
  __shared__ float sh[4];
  int b = blockIdx.x;
  const int tw = b % tiles_w; b /= tiles_w;
  const int th = b % tiles_h; const int plane = b / tiles_h;
  const int Ho = H - 10, Wo = W - 10;
  const int oh0 = th * SSIM_TH, ow0 = tw * SSIM_TW;
  const float* xp = x + (size_t)plane * H * W;
  const float* yp = y + (size_t)plane * H * W;
  for (int e = threadIdx.x; e < (SSIM_TH + 10) * (SSIM_TW + 10); e += 256) {
    const int r = e / (SSIM_TW + 10), c = e % (SSIM_TW + 10);
    const int ih = oh0 + r, iw = ow0 + c;
    float a = 0.f, bb = 0.f;
    if (ih < H && iw < W) { a = (xp[(size_t)ih * W + iw] + 1.f) * 0.5f; bb = (yp[(size_t)ih * W + iw] + 1.f) * 0.5f; }
    sx[r][c] = a; sy[r][c] = bb;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < (SSIM_TH + 10) * SSIM_TW; e += 256) {
    const int r = e / SSIM_TW, c = e % SSIM_TW;
    float m1 = 0.f, m2 = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float g = c_gauss[k], a = sx[r][c + k], bb = sy[r][c + k];
      m1 += g * a; m2 += g * bb; xx += g * a * a; yy += g * bb * bb; xy += g * a * bb;
    }
    hz[0][r][c] = m1; hz[1][r][c] = m2; hz[2][r][c] = xx; hz[3][r][c] = yy; hz[4][r][c] = xy;
  }
  __syncthreads();
  float acc = 0.f;
  for (int e = threadIdx.x; e < SSIM_TH * SSIM_TW; e += 256) {
    const int r = e / SSIM_TW, c = e % SSIM_TW;
    if (oh0 + r < Ho && ow0 + c < Wo) {
      float m1 = 0.f, m2 = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float g = c_gauss[k];
        m1 += g * hz[0][r + k][c]; m2 += g * hz[1][r + k][c]; xx += g * hz[2][r + k][c];
        yy += g * hz[3][r + k][c]; xy += g * hz[4][r + k][c];
      }
      const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
      const float s1sq = xx - m1 * m1, s2sq = yy - m2 * m2, s12 = xy - m1 * m2;
      const float S1 = (2.f * m1 * m2 + C1) / (m1 * m1 + m2 * m2 + C1);
      const float S2 = (2.f * s12 + C2) / (s1sq + s2sq + C2);
      const float S = fmaxf(2.f - (S1 + S2), 0.f);
      acc += sqrtf(S);
    }
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ void ssim_final_kernel(const float* partial, int n, float scale, float* out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = (float)(sh[0] * (double)scale);
}

static int ssim_init_gauss() {
  static bool init = false;
  if (!init) {
    // fp32 restatement of _fspecial_gauss_1d(11, 1.5) (ssim.py:22-40)
    float g[11], s = 0.f;
    for (int i = 0; i < 11; ++i) { const float c = (float)(i - 5); g[i] = expf(-(c * c) / (2.f * 1.5f * 1.5f)); s += g[i]; }
    for (int i = 0; i < 11; ++i) g[i] /= s;
    GS_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_gauss), g, sizeof(g)));
    init = true;
  }
  return 0;
}

extern "C" int64_t gs_ssim_scratch_floats(int32_t NC, int32_t H, int32_t W) {
  const int64_t th = (H - 10 + SSIM_TH - 1) / SSIM_TH, tw = (W - 10 + SSIM_TW - 1) / SSIM_TW;
  return (int64_t)NC * th * tw;
}
extern "C" int gs_ssim_distance(const float* x, const float* y, int32_t NC, int32_t H, int32_t W, float* out,
                                float* scratch, void* stream) {
  GS_REQUIRE(x && y && out && scratch && NC > 0 && H > 10 && W > 10, "gs_ssim_distance: bad argument");
  if (int rc = ssim_init_gauss()) return rc;
  const int th = (H - 10 + SSIM_TH - 1) / SSIM_TH, tw = (W - 10 + SSIM_TW - 1) / SSIM_TW;
  const int blocks = NC * th * tw;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(ssim_kernel, dim3(blocks), dim3(256), 0, st, x, y, H, W, tw, th, scratch);
  const double cnt = (double)NC * (H - 10) * (W - 10);
  hipLaunchKernelGGL(ssim_final_kernel, dim3(1), dim3(256), 0, st, scratch, blocks, (float)(1.0 / cnt), out);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}


// ---- SSIM distance: gradient w.r.t. the second image (the distance is symmetric, swap the arguments for the first) ----
// L = mean(D), D = sqrt(relu(2 - S1 - S2)) with mu1 = b(X), mu2 = b(Y), e11 = b(XX), e22 = b(YY), e12 = b(XY) (b = the
// separable 11-tap Gaussian, valid region). Per output pixel o, with gD = -dL/(2 D n):
//   G0 = gD * (dS1/dmu2 - mu1 * dS2/ds12 - 2 mu2 * dS2/ds2),  G1 = gD * dS2/ds12,  G2 = gD * dS2/ds2
//   dS1/dmu2 = (2 mu1 B1 - 2 mu2 A1) / B1^2,  dS2/ds12 = 2 / B2,  dS2/ds2 = -A2 / B2^2   (S1 = A1/B1, S2 = A2/B2)
// and dL/dY(p) = 1/2 * ( bT(G0)(p) + X(p) bT(G1)(p) + 2 Y(p) bT(G2)(p) ) with bT the transposed (full) Gaussian; the 1/2 is
// the (y + 1)/2 input mapping (nn/losses/utils/ssim.py:65-99 differentiated; cyclegan_losses.py:78-90 uses it as a loss).
__global__ __launch_bounds__(256) void ssim_grad_maps_kernel(const float* x, const float* y, int H, int W, int tiles_w,
                                                             int tiles_h, const float* grad_scale, float inv_cnt,
                                                             float* maps, size_t map_stride) {
  __shared__ float sx[SSIM_TH + 10][SSIM_TW + 10];
  __shared__ float sy[SSIM_TH + 10][SSIM_TW + 10];
  __shared__ float hz[5][SSIM_TH + 10][SSIM_TW];
  int b = blockIdx.x;
  const int tw = b % tiles_w; b /= tiles_w;
  const int th = b % tiles_h; const int plane = b / tiles_h;
  const int Ho = H - 10, Wo = W - 10;
  const int oh0 = th * SSIM_TH, ow0 = tw * SSIM_TW;
  const float* xp = x + (size_t)plane * H * W;
  const float* yp = y + (size_t)plane * H * W;
  for (int e = threadIdx.x; e < (SSIM_TH + 10) * (SSIM_TW + 10); e += 256) {
    const int r = e / (SSIM_TW + 10), c = e % (SSIM_TW + 10);
    const int ih = oh0 + r, iw = ow0 + c;
    float a = 0.f, bb = 0.f;
    if (ih < H && iw < W) { a = (xp[(size_t)ih * W + iw] + 1.f) * 0.5f; bb = (yp[(size_t)ih * W + iw] + 1.f) * 0.5f; }
    sx[r][c] = a; sy[r][c] = bb;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < (SSIM_TH + 10) * SSIM_TW; e += 256) {
    const int r = e / SSIM_TW, c = e % SSIM_TW;
    float m1 = 0.f, m2 = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float g = c_gauss[k], a = sx[r][c + k], bb = sy[r][c + k];
      m1 += g * a; m2 += g * bb; xx += g * a * a; yy += g * bb * bb; xy += g * a * bb;
    }
    hz[0][r][c] = m1; hz[1][r][c] = m2; hz[2][r][c] = xx; hz[3][r][c] = yy; hz[4][r][c] = xy;
  }
  __syncthreads();
  const float up = (grad_scale ? grad_scale[0] : 1.f) * inv_cnt;
  float* mp = maps + (size_t)plane * Ho * Wo;
  for (int e = threadIdx.x; e < SSIM_TH * SSIM_TW; e += 256) {
    const int r = e / SSIM_TW, c = e % SSIM_TW;
    if (oh0 + r < Ho && ow0 + c < Wo) {
      float m1 = 0.f, m2 = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float g = c_gauss[k];
        m1 += g * hz[0][r + k][c]; m2 += g * hz[1][r + k][c]; xx += g * hz[2][r + k][c];
        yy += g * hz[3][r + k][c]; xy += g * hz[4][r + k][c];
      }
      const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
      const float s1sq = xx - m1 * m1, s2sq = yy - m2 * m2, s12 = xy - m1 * m2;
      const float A1 = 2.f * m1 * m2 + C1, B1 = m1 * m1 + m2 * m2 + C1;
      const float A2 = 2.f * s12 + C2, B2 = s1sq + s2sq + C2;
      const float S = 2.f - (A1 / B1 + A2 / B2);
      float g0 = 0.f, g1 = 0.f, g2 = 0.f;
      if (S > 0.f) {
        const float gD = -up / (2.f * sqrtf(S));
        const float dS1 = (2.f * m1 * B1 - 2.f * m2 * A1) / (B1 * B1);
        const float dS2_12 = 2.f / B2, dS2_2 = -A2 / (B2 * B2);
        g0 = gD * (dS1 - m1 * dS2_12 - 2.f * m2 * dS2_2);
        g1 = gD * dS2_12;
        g2 = gD * dS2_2;
      }
      const size_t o = (size_t)(oh0 + r) * Wo + (ow0 + c);
      mp[o] = g0; mp[map_stride + o] = g1; mp[2 * map_stride + o] = g2;
    }
  }
}

// transposed separable Gaussian of the three maps + combination with X, Y: one workgroup = 16x32 input pixels
__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* x, const float* y, const float* maps,
                                                       size_t map_stride, int H, int W, int tiles_w, int tiles_h,
                                                       float* grad_y) {
  __shared__ float sm[3][SSIM_TH + 10][SSIM_TW + 10];
  __shared__ float hz[3][SSIM_TH + 10][SSIM_TW];
  int b = blockIdx.x;
  const int tw = b % tiles_w; b /= tiles_w;
  const int th = b % tiles_h; const int plane = b / tiles_h;
  const int Ho = H - 10, Wo = W - 10;
  const int ph0 = th * SSIM_TH, pw0 = tw * SSIM_TW;
  const float* mp = maps + (size_t)plane * Ho * Wo;
  // output-pixel window [ph0 - 10, ph0 + TH) x [pw0 - 10, pw0 + TW), zero outside the valid map
  for (int e = threadIdx.x; e < (SSIM_TH + 10) * (SSIM_TW + 10); e += 256) {
    const int r = e / (SSIM_TW + 10), c = e % (SSIM_TW + 10);
    const int oh = ph0 - 10 + r, ow = pw0 - 10 + c;
    const bool ok = oh >= 0 && oh < Ho && ow >= 0 && ow < Wo;
    const size_t o = ok ? (size_t)oh * Wo + ow : 0;
#pragma unroll
    for (int m = 0; m < 3; ++m) sm[m][r][c] = ok ? mp[m * map_stride + o] : 0.f;
  }
  __syncthreads();
  // horizontal: t[r][c] = sum_k w[k] * G[.., pw - k]  ->  window column (c + 10 - k)
  for (int e = threadIdx.x; e < (SSIM_TH + 10) * SSIM_TW; e += 256) {
    const int r = e / SSIM_TW, c = e % SSIM_TW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float g = c_gauss[k];
      a0 += g * sm[0][r][c + 10 - k]; a1 += g * sm[1][r][c + 10 - k]; a2 += g * sm[2][r][c + 10 - k];
    }
    hz[0][r][c] = a0; hz[1][r][c] = a1; hz[2][r][c] = a2;
  }
  __syncthreads();
  const float* xp = x + (size_t)plane * H * W;
  const float* yp = y + (size_t)plane * H * W;
  float* gp = grad_y + (size_t)plane * H * W;
  for (int e = threadIdx.x; e < SSIM_TH * SSIM_TW; e += 256) {
    const int r = e / SSIM_TW, c = e % SSIM_TW;
    const int ph = ph0 + r, pw = pw0 + c;
    if (ph < H && pw < W) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float g = c_gauss[k];
        a0 += g * hz[0][r + 10 - k][c]; a1 += g * hz[1][r + 10 - k][c]; a2 += g * hz[2][r + 10 - k][c];
      }
      const size_t i = (size_t)ph * W + pw;
      const float X = (xp[i] + 1.f) * 0.5f, Y = (yp[i] + 1.f) * 0.5f;
      gp[i] = 0.5f * (a0 + X * a1 + 2.f * Y * a2);
    }
  }
}

extern "C" int64_t gs_ssim_backward_scratch_floats(int32_t NC, int32_t H, int32_t W) {
  return 3 * (int64_t)NC * (H - 10) * (W - 10);
}

extern "C" int gs_ssim_distance_backward(const float* x, const float* y, int32_t NC, int32_t H, int32_t W,
                                         const float* grad_scale, float* grad_y, float* scratch, void* stream) {
  GS_REQUIRE(x && y && grad_y && scratch && NC > 0 && H > 10 && W > 10, "gs_ssim_distance_backward: bad argument");
  if (int rc = ssim_init_gauss()) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int Ho = H - 10, Wo = W - 10;
  const size_t map_stride = (size_t)NC * Ho * Wo;
  const int th = (Ho + SSIM_TH - 1) / SSIM_TH, tw = (Wo + SSIM_TW - 1) / SSIM_TW;
  hipLaunchKernelGGL(ssim_grad_maps_kernel, dim3(NC * th * tw), dim3(256), 0, st, x, y, H, W, tw, th, grad_scale,
                     (float)(1.0 / ((double)NC * Ho * Wo)), scratch, map_stride);
  const int th2 = (H + SSIM_TH - 1) / SSIM_TH, tw2 = (W + SSIM_TW - 1) / SSIM_TW;
  hipLaunchKernelGGL(ssim_bwd_kernel, dim3(NC * th2 * tw2), dim3(256), 0, st, x, y, scratch, map_stride, H, W, tw2, th2,
                     grad_y);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- scalar algebra of a recipe's loss assembly ----------------------------------------------------------------------------------
// out[r] = c[r] + sum_k m[r][k] * x_k[0] for R <= 8 rows over K <= 16 device scalars (a null x_k counts as 0): the
// "lambda_AB * (alpha * ssim + beta * l1)", "loss_real + loss_fake", "sum of the G losses" lines of the recipes
// (cyclegan_losses.py:21-32,70-90, cyclegan.py:150,182) as ONE launch instead of one torch elementwise kernel per operator;
// the backward of such a combination is the same launch with the transposed matrix over the rows' upstream gradients.
struct ScalarAffineK {
  const float* x[16];
  float m[8][16];
  float c[8];
  float* out;
  int K, R;
};
__global__ __launch_bounds__(64) void scalar_affine_kernel(const ScalarAffineK p) {
  const int r = threadIdx.x;
  if (r >= p.R) return;
  float acc = p.c[r];
  for (int k = 0; k < p.K; ++k)
    if (p.x[k]) acc += p.m[r][k] * p.x[k][0];
  p.out[r] = acc;
}
extern "C" int gs_scalar_affine(const float* const* x, int32_t K, const float* m, const float* c, int32_t R, float* out,
                                void* stream) {
  GS_REQUIRE(x && m && out && K > 0 && K <= 16 && R > 0 && R <= 8, "gs_scalar_affine: bad argument (K <= 16, R <= 8)");
  ScalarAffineK p;
  for (int k = 0; k < 16; ++k) p.x[k] = k < K ? x[k] : nullptr;
  for (int r = 0; r < 8; ++r) {
    p.c[r] = (c && r < R) ? c[r] : 0.f;
    for (int k = 0; k < 16; ++k) p.m[r][k] = (r < R && k < K) ? m[r * K + k] : 0.f;
  }
  p.out = out; p.K = K; p.R = R;
  hipLaunchKernelGGL(scalar_affine_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), p);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// out = a + b over n floats: the join of two gradients of one image (a generated image feeds a discriminator AND the
// other generator, cyclegan.py:131-141 — autograd's own accumulation would be a torch kernel)
__global__ __launch_bounds__(256) void sum2_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                   float4* __restrict__ out, long long n4, const float* as, const float* bs,
                                                   float* os, int tail) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 u = a[i], v = b[i];
    out[i] = float4{u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w};
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) os[threadIdx.x] = as[threadIdx.x] + bs[threadIdx.x];
}
extern "C" int gs_sum2_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
  GS_REQUIRE(a && b && out && n > 0, "gs_sum2_f32: bad argument");
  GS_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
             "gs_sum2_f32: buffers must be 16-byte aligned");
  const long long n4 = n / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sum2_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(out), n4,
                     a + n4 * 4, b + n4 * 4, out + n4 * 4, (int)(n - n4 * 4));
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

