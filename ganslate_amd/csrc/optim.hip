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
