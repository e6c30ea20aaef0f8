// Feature taps of CUT's PatchNCE loss (ganslate/nn/gans/unpaired/cut.py:229-312, FeaturePatchMLP.forward :262-277): the
// features of P sampled pixels per image are read out of NHWC bf16 activations (forward), their gradients are added back at
// those pixels (backward). The reference does both with torch indexing on [N, HW, C] views; here they are three small
// kernels on the executor's own buffers, so neither a dense zero tensor per tapped level nor a dense add is needed.
// Layer 0 of nce_layers is the ReflectionPad2d(3) output of the IMAGE (fp32 NCHW): its gather resolves the reflection
// per sample and its backward scatters through the same map.
#include "common.hpp"

// out[n][p][ch] = src[n][ids[p]][ch], ch < c           grid (P, n)
__global__ __launch_bounds__(256) void tap_gather_kernel(const unsigned short* __restrict__ src, long long img_stride, int cs,
                                                         const long long* __restrict__ ids, int P, int c,
                                                         float* __restrict__ out) {
  const int p = blockIdx.x, n = blockIdx.y;
  const unsigned short* s = src + (size_t)n * img_stride + (size_t)ids[p] * cs;
  float* o = out + ((size_t)n * P + p) * c;
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) o[ch] = bf2f(s[ch]);
}

// dst[n][(y + f0) * Wp + x + f0][ch] += g[n][p][ch] with (y, x) = divmod(ids[p], W): ids are distinct per image (a random
// permutation's head), so no two threads meet
__global__ __launch_bounds__(256) void tap_scatter_add_kernel(unsigned short* __restrict__ dst, long long img_stride, int cs,
                                                              const long long* __restrict__ ids, int P, int c, int W, int Wp,
                                                              int f0, const float* __restrict__ g) {
  const int p = blockIdx.x, n = blockIdx.y;
  const long long id = ids[p];
  const long long y = id / W, x = id - y * W;
  unsigned short* d = dst + (size_t)n * img_stride + (size_t)((y + f0) * Wp + x + f0) * cs;
  const float* s = g + ((size_t)n * P + p) * c;
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) d[ch] = f2bf(bf2f(d[ch]) + s[ch]);
}

__global__ __launch_bounds__(256) void zero16_kernel(uint4* p, long long n16) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x)
    p[i] = uint4{0u, 0u, 0u, 0u};
}

// db[ch] += sum over rows of g[row][ch] (fixed order): the bias gradient a tap of a RAW conv output carries. 32 row lanes x 32
// channels per workgroup, four rows in flight per lane (8 row lanes walking 2048 rows one load at a time took 66 us)
__global__ __launch_bounds__(1024) void tap_rows_sum_kernel(const float* __restrict__ g, long long rows, int c, float* db) {
  __shared__ float red[32][33];
  const int lane = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int ch = blockIdx.x * 32 + lane;
  float s = 0.f;
  if (ch < c) {
    long long r = part;
    for (; r + 96 < rows; r += 128) {
      const float a0 = g[r * c + ch], a1 = g[(r + 32) * c + ch], a2 = g[(r + 64) * c + ch], a3 = g[(r + 96) * c + ch];
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; r < rows; r += 32) s += g[r * c + ch];
  }
  red[part][lane] = s;
  __syncthreads();
  if (part == 0 && ch < c) {
    float t = 0.f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) t += red[k][lane];
    db[ch] += t;
  }
}

extern "C" int gs_tap_gather(const void* src, int32_t n, int64_t pixels, int32_t cs, const int64_t* ids_dev, int32_t P, int32_t c,
                             float* out, void* stream) {
  GS_REQUIRE(src && ids_dev && out && n > 0 && pixels > 0 && P > 0 && c > 0 && c <= cs, "gs_tap_gather: bad argument");
  hipLaunchKernelGGL(tap_gather_kernel, dim3((unsigned)P, (unsigned)n), dim3(c >= 256 ? 256 : (c > 64 ? 128 : 64)), 0,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned short*>(src), (long long)pixels * cs, cs,
                     reinterpret_cast<const long long*>(ids_dev), P, c, out);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_tap_scatter_add(void* dst, int32_t n, int64_t pixels, int32_t cs, const int64_t* ids_dev, int32_t P, int32_t c,
                                  int32_t W, int32_t Wp, int32_t f0, const float* g, void* stream) {
  GS_REQUIRE(dst && ids_dev && g && n > 0 && pixels > 0 && P > 0 && c > 0 && c <= cs && W > 0 && Wp >= W + 2 * f0 && f0 >= 0,
             "gs_tap_scatter_add: bad argument");
  hipLaunchKernelGGL(tap_scatter_add_kernel, dim3((unsigned)P, (unsigned)n), dim3(c >= 256 ? 256 : (c > 64 ? 128 : 64)), 0,
                     static_cast<hipStream_t>(stream), static_cast<unsigned short*>(dst), (long long)pixels * cs, cs,
                     reinterpret_cast<const long long*>(ids_dev), P, c, W, Wp, f0, g);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_zero_bytes(void* p, int64_t bytes, void* stream) {
  GS_REQUIRE(p && bytes > 0 && bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0,
             "gs_zero_bytes: 16-byte aligned buffer of a multiple of 16 bytes expected");
  long long blocks = (bytes / 16 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(zero16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<uint4*>(p), (long long)(bytes / 16));
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_tap_rows_sum(const float* g, int64_t rows, int32_t c, float* db, void* stream) {
  GS_REQUIRE(g && db && rows > 0 && c > 0, "gs_tap_rows_sum: bad argument");
  hipLaunchKernelGGL(tap_rows_sum_kernel, dim3((unsigned)((c + 31) / 32)), dim3(1024), 0, static_cast<hipStream_t>(stream), g,
                     (long long)rows, c, db);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- taps of the reflection-padded image (nce layer 0) --------------------------------------------------------------------------
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// out[n][p][ch] = x[n][ch][reflect(yp - pad)][reflect(xp - pad)], (yp, xp) = divmod(ids[p], W + 2 pad)      grid (P, N)
__global__ __launch_bounds__(64) void image_tap_gather_kernel(const float* __restrict__ x, int C, int H, int W, int pad,
                                                              const long long* __restrict__ ids, int P, float* __restrict__ out) {
  const int p = blockIdx.x, n = blockIdx.y, Wp = W + 2 * pad;
  const long long id = ids[p];
  const int yp = (int)(id / Wp), xp = (int)(id - (long long)yp * Wp);
  const int y = reflect_idx(yp - pad, H), xx = reflect_idx(xp - pad, W);
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x)
    out[((size_t)n * P + p) * C + ch] = x[(((size_t)n * C + ch) * H + y) * W + xx];
}
// gx[n][ch][y][x] += g[n][p][ch] through the same map (gx zeroed by the caller). Padded positions that reflect onto one
// pixel (up to 4) add up in SAMPLE ORDER: the first sample of a pixel owns it and adds its later colliders itself, the others
// leave — no atomics, so two runs (and the two reduction forms of the data-parallel self-check) agree bit for bit.
__global__ __launch_bounds__(64) void image_tap_scatter_kernel(float* gx, int C, int H, int W, int pad,
                                                               const long long* __restrict__ ids, int P,
                                                               const float* __restrict__ g) {
  extern __shared__ int tap_tgt[];          // [P] target pixel of every sample
  const int p = blockIdx.x, n = blockIdx.y, Wp = W + 2 * pad;
  for (int q = threadIdx.x; q < P; q += blockDim.x) {
    const long long id = ids[q];
    const int yp = (int)(id / Wp), xp = (int)(id - (long long)yp * Wp);
    tap_tgt[q] = reflect_idx(yp - pad, H) * W + reflect_idx(xp - pad, W);
  }
  __syncthreads();
  const int mine = tap_tgt[p];
  for (int q = 0; q < p; ++q)
    if (tap_tgt[q] == mine) return;          // (block-uniform) an earlier sample owns this pixel
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    float acc = g[((size_t)n * P + p) * C + ch];
    for (int q = p + 1; q < P; ++q)
      if (tap_tgt[q] == mine) acc += g[((size_t)n * P + q) * C + ch];
    gx[((size_t)n * C + ch) * H * W + mine] = acc;
  }
}

extern "C" int gs_image_tap_gather(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, int32_t pad,
                                   const int64_t* ids_dev, int32_t P, float* out, void* stream) {
  GS_REQUIRE(x && ids_dev && out && N > 0 && C > 0 && H > pad && W > pad && pad >= 0 && P > 0, "gs_image_tap_gather: bad argument");
  hipLaunchKernelGGL(image_tap_gather_kernel, dim3((unsigned)P, (unsigned)N), dim3(64), 0, static_cast<hipStream_t>(stream), x, C,
                     H, W, pad, reinterpret_cast<const long long*>(ids_dev), P, out);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int gs_image_tap_scatter(const float* g, int32_t N, int32_t C, int32_t H, int32_t W, int32_t pad,
                                    const int64_t* ids_dev, int32_t P, float* gx, void* stream) {
  GS_REQUIRE(g && ids_dev && gx && N > 0 && C > 0 && H > pad && W > pad && pad >= 0 && P > 0 && P <= 16384,
             "gs_image_tap_scatter: bad argument");
  const long long bytes = (long long)N * C * H * W * 4;
  if (bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(gx) & 15) == 0) {
    if (int rc = gs_zero_bytes(gx, bytes, stream)) return rc;
  } else {
    GS_CHECK_HIP(hipMemsetAsync(gx, 0, (size_t)bytes, static_cast<hipStream_t>(stream)));
  }
  hipLaunchKernelGGL(image_tap_scatter_kernel, dim3((unsigned)P, (unsigned)N), dim3(64), (size_t)P * 4, static_cast<hipStream_t>(stream), gx,
                     C, H, W, pad, reinterpret_cast<const long long*>(ids_dev), P, g);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- FastCUT's flip-equivariance coin (cut.py:146-152): x.flip(-1) or x, decided by a flag in DEVICE memory so that a captured
// step replays either way ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flip_w_if_kernel(const float* __restrict__ x, float* __restrict__ out, long long rows, int W,
                                                        const int* __restrict__ flag) {
  const bool f = flag[0] != 0;
  const long long n = rows * W;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / W;
    const int w = (int)(e - r * W);
    out[e] = x[r * W + (f ? W - 1 - w : w)];
  }
}
extern "C" int gs_flip_w_if(const float* x, float* out, int64_t rows, int32_t W, const int32_t* flag_dev, void* stream) {
  GS_REQUIRE(x && out && flag_dev && rows > 0 && W > 0 && x != out, "gs_flip_w_if: bad argument (not in place)");
  long long blocks = (rows * W + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(flip_w_if_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, out,
                     (long long)rows, W, flag_dev);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

