// Fused flat Adam over a network's contiguous fp32 parameter / gradient / moment buffers, and the refresh of
// the bf16 weight packs the conv kernels read. Replaces torch.optim.Adam(lr, betas=(0.5, 0.999)).step()
// (ganslate/nn/gans/unpaired/cyclegan.py:81-82,107,123); arithmetic order follows torch's single-tensor Adam:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.hpp"

__global__ __launch_bounds__(256) void adam_kernel(float* p, float* g, float* m, float* v, long long n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float gscale, int zero_grad) {
  const float step_size = lr / bc1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);          // torch: exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;         // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
    if (zero_grad) g[i] = 0.f;
  }
}

// Same update with the step-dependent scalars read from device memory: a captured hipGraph of the training step
// replays this launch unchanged while the host refreshes hyper_dev (lr schedule, bias corrections) between replays.
__global__ __launch_bounds__(256) void adam_dev_kernel(float* p, float* g, float* m, float* v, long long n,
                                                       const float* __restrict__ hyper, float gscale, int zero_grad) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], bc1 = hyper[4], bc2_sqrt = hyper[5];
  const float step_size = lr / bc1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
    if (zero_grad) g[i] = 0.f;
  }
}

extern "C" int gs_adam_step_dev(float* p, float* g, float* m, float* v, int64_t n, const float* hyper_dev,
                                float grad_scale, int32_t zero_grad, void* stream) {
  GS_REQUIRE(p && g && m && v && hyper_dev && n > 0, "gs_adam_step_dev: bad argument");
  long long blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m,
                     v, (long long)n, hyper_dev, grad_scale, zero_grad);
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
  long long blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v,
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

extern "C" int gs_repack_bf16(const float* master, const int32_t* index, void* pack, int64_t n, void* stream) {
  GS_REQUIRE(master && index && pack && n > 0, "gs_repack_bf16: bad argument");
  long long blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(repack_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), master,
                     index, static_cast<unsigned short*>(pack), (long long)n);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
