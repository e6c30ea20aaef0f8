// Fused flat Adam over a network's contiguous fp32 parameter / gradient / moment buffers, and the refresh of
// the bf16 weight packs the conv kernels read. Replaces torch.optim.Adam(lr, betas=(0.5, 0.999)).step()
// (ganslate/nn/gans/unpaired/cyclegan.py:81-82,107,123); arithmetic order follows torch's single-tensor Adam:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.hpp"

// (adam_one — one element of the update — lives in common.hpp: the fused weight-gradient epilogue of wgrad.hip runs it too)

// 28-32 bytes of HBM traffic per parameter and nothing else: 16-byte accesses, two of them per array in flight per
// thread (dword accesses with one element per iteration ran at 1.5 TB/s: 469 us for the 57 M parameters of pix2pix's
// U-Net, a sixth of that step). Buffers are 16-byte aligned (torch allocations); a tail of n % 4 elements and
// unaligned views take the scalar loop.
__device__ __forceinline__ void adam_body(float* p, float* g, float* m, float* v, long long n, float lr, float b1,
                                          float b2, float eps, float bc1, float bc2_sqrt, float gscale, int zero_grad) {
  const float step_size = lr / bc1;
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
  const bool aligned = ((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) |
                         reinterpret_cast<size_t>(v)) & 15) == 0;
  const long long n4 = aligned ? n >> 2 : 0;
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  auto upd = [&](float4& P, float4& G, float4& M, float4& V) {
    adam_one(P.x, G.x, M.x, V.x, b1, b2, eps, bc2_sqrt, step_size, gscale, zero_grad);
    adam_one(P.y, G.y, M.y, V.y, b1, b2, eps, bc2_sqrt, step_size, gscale, zero_grad);
    adam_one(P.z, G.z, M.z, V.z, b1, b2, eps, bc2_sqrt, step_size, gscale, zero_grad);
    adam_one(P.w, G.w, M.w, V.w, b1, b2, eps, bc2_sqrt, step_size, gscale, zero_grad);
  };
  long long i = tid;
  for (; i + nth < n4; i += 2 * nth) {
    const long long j = i + nth;
    float4 P0 = p4[i], G0 = g4[i], M0 = m4[i], V0 = v4[i];
    float4 P1 = p4[j], G1 = g4[j], M1 = m4[j], V1 = v4[j];
    upd(P0, G0, M0, V0);
    upd(P1, G1, M1, V1);
    p4[i] = P0; m4[i] = M0; v4[i] = V0;
    p4[j] = P1; m4[j] = M1; v4[j] = V1;
    if (zero_grad) { g4[i] = G0; g4[j] = G1; }
  }
  for (; i < n4; i += nth) {
    float4 P0 = p4[i], G0 = g4[i], M0 = m4[i], V0 = v4[i];
    upd(P0, G0, M0, V0);
    p4[i] = P0; m4[i] = M0; v4[i] = V0;
    if (zero_grad) g4[i] = G0;
  }
  for (long long e = n4 * 4 + tid; e < n; e += nth) adam_one(p[e], g[e], m[e], v[e], b1, b2, eps, bc2_sqrt, step_size, gscale, zero_grad);
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, float* g, float* m, float* v, long long n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float gscale, int zero_grad) {
  adam_body(p, g, m, v, n, lr, b1, b2, eps, bc1, bc2_sqrt, gscale, zero_grad);
}

// Same update with the step-dependent scalars read from device memory: a captured hipGraph of the training step
// replays this launch unchanged while the host refreshes hyper_dev (lr schedule, bias corrections) between replays.
__global__ __launch_bounds__(256) void adam_dev_kernel(float* p, float* g, float* m, float* v, long long n,
                                                       const float* __restrict__ hyper, float gscale, int zero_grad) {
  adam_body(p, g, m, v, n, hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], gscale, zero_grad);
}

// The same update that also writes the bf16 weight packs where a pack group of 8 elements IS a group of 8 consecutive
// master elements (every row-major pack: a conv's forward pack, a transposed conv's data-gradient pack): inv_x[i] = pack
// group of master elements 8 i .. 8 i + 7, or -1. The lane that updated elements 8 i + 4 h .. + 3 stores half h of the
// group (8 bytes): the separate refresh launch read every such master element a second time (4 of its 6.5 bytes per weight;
// 219 of 1630 us of Adam + refresh on the 167 M-parameter U-Net). Transposed packs stay with gs_repack_bf16_tiled_groups.
__device__ __forceinline__ void adam_packs_body(float* p, float* g, float* m, float* v, long long n,
                                                const float* __restrict__ hyper, float gscale, int zero_grad,
                                                const int* __restrict__ inv_f, uint2* __restrict__ fpack,
                                                const int* __restrict__ inv_d, uint2* __restrict__ dpack) {
  const float b1 = hyper[1], b2 = hyper[2], eps = hyper[3], bc2_sqrt = hyper[5];
  const float step_size = hyper[0] / hyper[4];
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
  const long long n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  auto upd = [&](float4& P, float4& G, float4& M, float4& V) {
    adam_one(P.x, G.x, M.x, V.x, b1, b2, eps, bc2_sqrt, step_size, gscale, zero_grad);
    adam_one(P.y, G.y, M.y, V.y, b1, b2, eps, bc2_sqrt, step_size, gscale, zero_grad);
    adam_one(P.z, G.z, M.z, V.z, b1, b2, eps, bc2_sqrt, step_size, gscale, zero_grad);
    adam_one(P.w, G.w, M.w, V.w, b1, b2, eps, bc2_sqrt, step_size, gscale, zero_grad);
  };
  auto packs = [&](long long i, const float4& P) {
    const uint2 o = {pack_bf2(P.x, P.y), pack_bf2(P.z, P.w)};
    const long long grp = i >> 1;
    const int jf = inv_f ? inv_f[grp] : -1, jd = inv_d ? inv_d[grp] : -1;
    if (jf >= 0) fpack[(long long)jf * 2 + (i & 1)] = o;
    if (jd >= 0) dpack[(long long)jd * 2 + (i & 1)] = o;
  };
  long long i = tid;
  for (; i + nth < n4; i += 2 * nth) {
    const long long j = i + nth;
    float4 P0 = p4[i], G0 = g4[i], M0 = m4[i], V0 = v4[i];
    float4 P1 = p4[j], G1 = g4[j], M1 = m4[j], V1 = v4[j];
    upd(P0, G0, M0, V0);
    upd(P1, G1, M1, V1);
    p4[i] = P0; m4[i] = M0; v4[i] = V0;
    p4[j] = P1; m4[j] = M1; v4[j] = V1;
    if (zero_grad) { g4[i] = G0; g4[j] = G1; }
    packs(i, P0);
    packs(j, P1);
  }
  for (; i < n4; i += nth) {
    float4 P0 = p4[i], G0 = g4[i], M0 = m4[i], V0 = v4[i];
    upd(P0, G0, M0, V0);
    p4[i] = P0; m4[i] = M0; v4[i] = V0;
    if (zero_grad) g4[i] = G0;
    packs(i, P0);
  }
  // (a tail of n % 4 elements belongs to no complete group: plain update)
  for (long long e = n4 * 4 + tid; e < n; e += nth) adam_one(p[e], g[e], m[e], v[e], b1, b2, eps, bc2_sqrt, step_size, gscale, zero_grad);
}
__global__ __launch_bounds__(256) void adam_dev_packs_kernel(float* p, float* g, float* m, float* v, long long n,
                                                             const float* __restrict__ hyper, float gscale, int zero_grad,
                                                             const int* __restrict__ inv_f, uint2* __restrict__ fpack,
                                                             const int* __restrict__ inv_d, uint2* __restrict__ dpack) {
  adam_packs_body(p, g, m, v, n, hyper, gscale, zero_grad, inv_f, fpack, inv_d, dpack);
}
// the same update over several ranges of the flat buffers in one launch (blockIdx.y = range; [start, end) in elements, start a
// multiple of 8): what is left of a network after gs_wgrad_adam took its large layers — biases and small layers between them
__global__ __launch_bounds__(256) void adam_dev_packs_ranges_kernel(float* p, float* g, float* m, float* v,
                                                                    const long long* __restrict__ ranges,
                                                                    const float* __restrict__ hyper, float gscale, int zero_grad,
                                                                    const int* __restrict__ inv_f, uint2* __restrict__ fpack,
                                                                    const int* __restrict__ inv_d, uint2* __restrict__ dpack) {
  const long long a = ranges[2 * blockIdx.y], b = ranges[2 * blockIdx.y + 1];
  if ((long long)blockIdx.x * blockDim.x * 4 >= b - a) return;          // (uniform: this range is shorter than the grid)
  adam_packs_body(p + a, g + a, m + a, v + a, b - a, hyper, gscale, zero_grad, inv_f ? inv_f + (a >> 3) : nullptr, fpack,
                  inv_d ? inv_d + (a >> 3) : nullptr, dpack);
}

static long long adam_blocks(int64_t n) {
  long long blocks = (n / 4 + 511) / 512;         // two 16-byte vectors per thread per pass
  if (blocks < 1) blocks = 1;
  const long long cap = gs_opt(GS_OPT_ADAM_BLOCKS) > 0 ? gs_opt(GS_OPT_ADAM_BLOCKS) : 8192;
  if (blocks > cap) blocks = cap;
  return blocks;
}

extern "C" int gs_adam_step_dev(float* p, float* g, float* m, float* v, int64_t n, const float* hyper_dev,
                                float grad_scale, int32_t zero_grad, void* stream) {
  GS_REQUIRE(p && g && m && v && hyper_dev && n > 0, "gs_adam_step_dev: bad argument");
  hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)adam_blocks(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g,
                     m, v, (long long)n, hyper_dev, grad_scale, zero_grad);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_adam_step_dev_packs(float* p, float* g, float* m, float* v, int64_t n, const float* hyper_dev,
                                      float grad_scale, int32_t zero_grad, const int32_t* inv_f, void* fpack,
                                      const int32_t* inv_d, void* dpack, void* stream) {
  GS_REQUIRE(p && g && m && v && hyper_dev && n > 0, "gs_adam_step_dev_packs: bad argument");
  GS_REQUIRE((inv_f == nullptr) == (fpack == nullptr) && (inv_d == nullptr) == (dpack == nullptr),
             "gs_adam_step_dev_packs: an inverse table comes with its pack");
  GS_REQUIRE(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
               reinterpret_cast<uintptr_t>(v)) & 15) == 0 &&
                 ((reinterpret_cast<uintptr_t>(fpack) | reinterpret_cast<uintptr_t>(dpack)) & 15) == 0,
             "gs_adam_step_dev_packs: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_dev_packs_kernel, dim3((unsigned)adam_blocks(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g,
                     m, v, (long long)n, hyper_dev, grad_scale, zero_grad, inv_f, static_cast<uint2*>(fpack), inv_d,
                     static_cast<uint2*>(dpack));
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_adam_step_dev_packs_ranges(float* p, float* g, float* m, float* v, const int64_t* ranges_dev, int32_t n_ranges,
                                             int64_t max_len, const float* hyper_dev, float grad_scale, int32_t zero_grad,
                                             const int32_t* inv_f, void* fpack, const int32_t* inv_d, void* dpack, void* stream) {
  GS_REQUIRE(p && g && m && v && ranges_dev && hyper_dev && n_ranges > 0 && n_ranges <= 65535 && max_len > 0,
             "gs_adam_step_dev_packs_ranges: bad argument");
  GS_REQUIRE((inv_f == nullptr) == (fpack == nullptr) && (inv_d == nullptr) == (dpack == nullptr),
             "gs_adam_step_dev_packs_ranges: an index table and its pack go together");
  GS_REQUIRE(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
               reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(fpack) | reinterpret_cast<uintptr_t>(dpack)) & 15) == 0,
             "gs_adam_step_dev_packs_ranges: buffers must be 16-byte aligned");
  long long bx = adam_blocks(max_len);
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(adam_dev_packs_ranges_kernel, dim3((unsigned)bx, (unsigned)n_ranges), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, g, m, v, reinterpret_cast<const long long*>(ranges_dev), hyper_dev,
                     grad_scale, zero_grad, inv_f, static_cast<uint2*>(fpack), inv_d, static_cast<uint2*>(dpack));
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// History buffer of generated images (ganslate/data/utils/image_pool.py:31-60) with the coin flips in device memory:
// code[b] < 0 passes image b through; code[b] = slot stores it in `slot` and returns it (pool still filling);
// code[b] = slot | 0x40000000 returns the stored image of `slot` and stores image b there. The images of a batch are
// processed in order by every thread for its own elements, so two images drawing the same slot see each other exactly
// like the reference's sequential loop.
__global__ __launch_bounds__(256) void pool_query_kernel(uint4* pool, const uint4* __restrict__ images, uint4* out,
                                                         const int* __restrict__ code, int B, long long vecs) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < vecs; i += (long long)gridDim.x * blockDim.x) {
    for (int b = 0; b < B; ++b) {
      const int c = code[b];
      uint4 v = images[(long long)b * vecs + i];
      if (c >= 0) {
        uint4* slot = pool + (long long)(c & 0x3fffffff) * vecs + i;
        if (c & 0x40000000) { const uint4 old = *slot; *slot = v; v = old; }
        else *slot = v;
      }
      out[(long long)b * vecs + i] = v;
    }
  }
}

extern "C" int gs_pool_query(void* pool, const void* images, void* out, const int32_t* code_dev, int32_t B,
                             int64_t image_bytes, void* stream) {
  GS_REQUIRE(pool && images && out && code_dev && B > 0 && image_bytes > 0 && image_bytes % 16 == 0,
             "gs_pool_query: bad argument (images must be a multiple of 16 bytes)");
  const long long vecs = image_bytes / 16;
  long long blocks = (vecs + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pool_query_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<uint4*>(pool), static_cast<const uint4*>(images), static_cast<uint4*>(out), code_dev, B,
                     vecs);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_adam_step(float* p, float* g, float* m, float* v, int64_t n, const float* hyper_host,
                            float grad_scale, int32_t zero_grad, void* stream) {
  GS_REQUIRE(p && g && m && v && hyper_host && n > 0, "gs_adam_step: bad argument");
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)adam_blocks(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v,
                     (long long)n, hyper_host[0], hyper_host[1], hyper_host[2], hyper_host[3], hyper_host[4],
                     hyper_host[5], grad_scale, zero_grad);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void repack_kernel(const float* master, const int* index, unsigned short* pack,
                                                     long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int s = index[i];
    pack[i] = s >= 0 ? f2bf(master[s]) : (unsigned short)0;
  }
}

// 8 pack elements per thread for 16-byte aligned runs: inside a (tap, channel-run) of a row-major pack the indices are
// consecutive, so the common case is two 16-byte master loads and one 16-byte pack store (the large transposed segments
// go to repack_tiled_kernel, so the scattered fallback below is rare)
__global__ __launch_bounds__(256) void repack_vec_kernel(const float* __restrict__ master, const int* __restrict__ index,
                                                         unsigned short* __restrict__ pack, long long n8) {
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n8; v += (long long)gridDim.x * blockDim.x) {
    const int4 ia = reinterpret_cast<const int4*>(index)[2 * v], ib = reinterpret_cast<const int4*>(index)[2 * v + 1];
    const int s0 = ia.x;
    float f[8];
    if (s0 >= 0 && (s0 & 3) == 0 && ia.y == s0 + 1 && ia.z == s0 + 2 && ia.w == s0 + 3 && ib.x == s0 + 4 &&
        ib.y == s0 + 5 && ib.z == s0 + 6 && ib.w == s0 + 7) {
      const float4 a = *reinterpret_cast<const float4*>(master + s0), b = *reinterpret_cast<const float4*>(master + s0 + 4);
      f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
      f[0] = ia.x >= 0 ? master[ia.x] : 0.f; f[1] = ia.y >= 0 ? master[ia.y] : 0.f;
      f[2] = ia.z >= 0 ? master[ia.z] : 0.f; f[3] = ia.w >= 0 ? master[ia.w] : 0.f;
      f[4] = ib.x >= 0 ? master[ib.x] : 0.f; f[5] = ib.y >= 0 ? master[ib.y] : 0.f;
      f[6] = ib.z >= 0 ? master[ib.z] : 0.f; f[7] = ib.w >= 0 ? master[ib.w] : 0.f;
    }
    uint4 o;
    o.x = (unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16);
    o.y = (unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16);
    o.z = (unsigned)f2bf(f[4]) | ((unsigned)f2bf(f[5]) << 16);
    o.w = (unsigned)f2bf(f[6]) | ((unsigned)f2bf(f[7]) << 16);
    reinterpret_cast<uint4*>(pack)[v] = o;
  }
}

// Pack segments whose gather runs along the ROWS of the pack (the transposed packs: a conv's data-gradient pack, a
// transposed conv's forward pack — master[ch][t][row]): consecutive pack elements are T*Q floats apart in the master, so
// the element-per-thread kernel above uses 4 bytes of every 64-byte line it touches (1.7 TB/s of nominal traffic on the
// 167 M-parameter U-Net). Here a workgroup owns a 64 x 64 tile [rows][k]: the index tile is read along k, the master
// along the rows (64 consecutive floats), and the bf16 tile is written along k again, transposed through LDS.
__global__ __launch_bounds__(256) void repack_tiled_kernel(const float* __restrict__ master, const int* __restrict__ index,
                                                           unsigned short* __restrict__ pack, int rows, int kp) {
  __shared__ int sidx[64][65];
  __shared__ unsigned short sval[64][68];
  const int r0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  for (int e = threadIdx.x; e < 4096; e += 256) {
    const int r = e >> 6, k = e & 63;
    sidx[r][k] = (r0 + r < rows) ? index[(size_t)(r0 + r) * kp + k0 + k] : -1;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 4096; e += 256) {
    const int r = e & 63, k = e >> 6;
    const int id = sidx[r][k];
    sval[r][k] = id >= 0 ? f2bf(master[id]) : (unsigned short)0;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += 256) {      // 4 elements (8 bytes) per thread
    const int r = e >> 4, k4 = (e & 15) * 4;
    if (r0 + r < rows) {
      uint2 o;
      o.x = (unsigned)sval[r][k4] | ((unsigned)sval[r][k4 + 1] << 16);
      o.y = (unsigned)sval[r][k4 + 2] | ((unsigned)sval[r][k4 + 3] << 16);
      *reinterpret_cast<uint2*>(pack + (size_t)(r0 + r) * kp + k0 + k4) = o;
    }
  }
}

extern "C" int gs_repack_bf16_tiled(const float* master, const int32_t* index, void* pack, int32_t rows, int32_t kp,
                                    void* stream) {
  GS_REQUIRE(master && index && pack && rows > 0 && kp > 0 && kp % 64 == 0, "gs_repack_bf16_tiled: bad argument");
  GS_REQUIRE((reinterpret_cast<uintptr_t>(pack) & 7) == 0, "gs_repack_bf16_tiled: pack must be 8-byte aligned");
  hipLaunchKernelGGL(repack_tiled_kernel, dim3((unsigned)(kp / 64), (unsigned)((rows + 63) / 64)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), master, index, static_cast<unsigned short*>(pack), rows, kp);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- group-indexed refresh: one index per 8 pack elements ------------------------------------------------------------------
// The element-wise kernels above read a 4-byte index per 2-byte pack element: 40 % of their HBM traffic (and 4 bytes of HBM per
// weight, twice, for every network). In every pack the conv kernels use, 8 consecutive pack elements come from 8 consecutive
// master elements — along k for the row-major packs (8 channels of one tap), along the ROWS for the transposed ones — or are
// padding, so the executor (nn/native/net.py) stores one base index per group and falls back to the element-wise kernels
// where that does not hold: a group of the row-major launch marked -2 reads its eight own indices from the element-wise
// table (`index`, may be null when no group is marked), a group marked -3 belongs to a transposed segment and is skipped.
__global__ __launch_bounds__(256) void repack_groups_kernel(const float* __restrict__ master, const int* __restrict__ gindex,
                                                            const int* __restrict__ index, uint4* __restrict__ pack,
                                                            long long n8) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const int b = gindex[i];
    if (b == -3) continue;         // belongs to a transposed segment: written by the tiled launch
    uint4 o = {0u, 0u, 0u, 0u};
    if (b == -2) {                 // an irregular group (small transposed segments merged into a run): its 8 own indices
      float f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int id = index[i * 8 + k]; f[k] = id >= 0 ? master[id] : 0.f; }
      o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
    } else if (b >= 0) {
      float f[8];
      if ((b & 3) == 0) {
        const float4 lo = *reinterpret_cast<const float4*>(master + b), hi = *reinterpret_cast<const float4*>(master + b + 4);
        f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w; f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = master[b + k];
      }
      o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
    }
    pack[i] = o;
  }
}

extern "C" int gs_repack_bf16_groups(const float* master, const int32_t* gindex, const int32_t* index, void* pack,
                                     int64_t n8, void* stream) {
  GS_REQUIRE(master && gindex && pack && n8 > 0, "gs_repack_bf16_groups: bad argument");
  GS_REQUIRE(((reinterpret_cast<uintptr_t>(pack) | reinterpret_cast<uintptr_t>(master)) & 15) == 0,
             "gs_repack_bf16_groups: master and pack must be 16-byte aligned");
  long long blocks = (n8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(repack_groups_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), master,
                     gindex, index, static_cast<uint4*>(pack), (long long)n8);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// transposed segments [rows][kp], each with a group index [rows / 8][kp]: pack[8 G + j][k] = master[gindex[G][k] + j]. ALL such
// segments of a pack in one launch (a network has dozens of them, most far too small for a launch of their own): seg[i] =
// {pack offset (elements), gindex offset, rows, kp, first tile}; a workgroup finds its segment by its tile number and owns
// 64 rows x 64 k of it: 8 x 64 group indices, 4096 master elements read as 8-element runs (two 16-byte loads), transposed
// through LDS.
__global__ __launch_bounds__(256) void repack_tiled_groups_kernel(const float* __restrict__ master,
                                                                  const int* __restrict__ gindex,
                                                                  unsigned short* __restrict__ pack_base,
                                                                  const long long* __restrict__ seg, int nseg) {
  __shared__ unsigned short sval[64][68];
  int lo = 0, hi = nseg - 1;                             // last segment whose first tile is <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (seg[mid * 5 + 4] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* sg = seg + lo * 5;
  const int rows = (int)sg[2], kp = (int)sg[3];
  const int tile = (int)((long long)blockIdx.x - sg[4]), tx = kp >> 6;
  const int r0 = (tile / tx) * 64, k0 = (tile % tx) * 64;
  const int* gi = gindex + sg[1];
  unsigned short* pack = pack_base + sg[0];
  for (int e = threadIdx.x; e < 512; e += 256) {       // (row group, k): 8 x 64
    const int G = e >> 6, k = e & 63;
    const int rg = (r0 >> 3) + G;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = 0.f;
    if (rg * 8 < rows) {
      const int b = gi[(size_t)rg * kp + k0 + k];
      if (b >= 0) {
        if ((b & 3) == 0) {
          const float4 lo4 = *reinterpret_cast<const float4*>(master + b), hi4 = *reinterpret_cast<const float4*>(master + b + 4);
          f[0] = lo4.x; f[1] = lo4.y; f[2] = lo4.z; f[3] = lo4.w; f[4] = hi4.x; f[5] = hi4.y; f[6] = hi4.z; f[7] = hi4.w;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = master[b + j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sval[G * 8 + j][k] = f2bf(f[j]);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += 256) {      // 4 elements (8 bytes) per thread
    const int r = e >> 4, k4 = (e & 15) * 4;
    if (r0 + r < rows) {
      uint2 o;
      o.x = (unsigned)sval[r][k4] | ((unsigned)sval[r][k4 + 1] << 16);
      o.y = (unsigned)sval[r][k4 + 2] | ((unsigned)sval[r][k4 + 3] << 16);
      *reinterpret_cast<uint2*>(pack + (size_t)(r0 + r) * kp + k0 + k4) = o;
    }
  }
}

extern "C" int gs_repack_bf16_tiled_groups(const float* master, const int32_t* gindex, void* pack, const int64_t* seg_dev,
                                           int32_t nseg, int64_t tiles, void* stream) {
  GS_REQUIRE(master && gindex && pack && seg_dev && nseg > 0 && tiles > 0 && tiles < (1LL << 31),
             "gs_repack_bf16_tiled_groups: bad argument");
  GS_REQUIRE(((reinterpret_cast<uintptr_t>(pack) & 7) | (reinterpret_cast<uintptr_t>(master) & 15)) == 0,
             "gs_repack_bf16_tiled_groups: pack must be 8-byte, master 16-byte aligned");
  hipLaunchKernelGGL(repack_tiled_groups_kernel, dim3((unsigned)tiles), dim3(256), 0, static_cast<hipStream_t>(stream), master,
                     gindex, static_cast<unsigned short*>(pack), reinterpret_cast<const long long*>(seg_dev), nseg);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_repack_bf16(const float* master, const int32_t* index, void* pack, int64_t n, void* stream) {
  GS_REQUIRE(master && index && pack && n > 0, "gs_repack_bf16: bad argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool aligned = ((reinterpret_cast<uintptr_t>(index) | reinterpret_cast<uintptr_t>(pack)) & 15) == 0;
  const long long n8 = aligned ? n >> 3 : 0;
  if (n8 > 0) {
    long long blocks = (n8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(repack_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, st, master, index,
                       static_cast<unsigned short*>(pack), n8);
  }
  const long long rest = n - (n8 << 3);
  if (rest > 0) {
    long long blocks = (rest + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, master, index + (n8 << 3),
                       static_cast<unsigned short*>(pack) + (n8 << 3), rest);
  }
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
