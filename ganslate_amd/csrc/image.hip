// Network boundary: NCHW fp32 images (the reference's `visuals`, cyclegan.py:39,84-90) <-> NHWC bf16
// activations with the channel count padded to a multiple of 8, plus the gradients of both conversions
// (tanh' of resnet2d.py:65 and the ReflectionPad2d adjoint of resnet2d.py:24 folded in).
#include "common.hpp"

__global__ __launch_bounds__(256) void image_to_act_kernel(const float* img, unsigned short* act, int C, long long hw,
                                                           int Cp) {
  const int n = blockIdx.y;
  const float* in = img + (size_t)n * C * hw;
  unsigned short* out = act + (size_t)n * hw * Cp;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (long long)gridDim.x * blockDim.x) {
    for (int c0 = 0; c0 < Cp; c0 += 8) {
      float f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = (c0 + k < C) ? in[(size_t)(c0 + k) * hw + p] : 0.f;
      uint4 o;
      o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
      *reinterpret_cast<uint4*>(out + (size_t)p * Cp + c0) = o;
    }
  }
}

__global__ __launch_bounds__(256) void act_to_image_kernel(const unsigned short* act, float* img, int C, long long hw,
                                                           int Cp, int act_kind) {
  const int n = blockIdx.y;
  const unsigned short* in = act + (size_t)n * hw * Cp;
  float* out = img + (size_t)n * C * hw;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (long long)gridDim.x * blockDim.x)
    for (int c = 0; c < C; ++c) out[(size_t)c * hw + p] = apply_act(bf2f(in[(size_t)p * Cp + c]), act_kind, 0.f);
}

__global__ __launch_bounds__(256) void act_to_image_bwd_kernel(const float* g_img, const float* out_img,
                                                               unsigned short* g_act, int C, long long hw, int Cp,
                                                               int act_kind) {
  const int n = blockIdx.y;
  const float* gi = g_img + (size_t)n * C * hw;
  const float* oi = out_img ? out_img + (size_t)n * C * hw : nullptr;
  unsigned short* out = g_act + (size_t)n * hw * Cp;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (long long)gridDim.x * blockDim.x) {
    for (int c0 = 0; c0 < Cp; c0 += 8) {
      float f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = c0 + k;
        float v = 0.f;
        if (c < C) {
          v = gi[(size_t)c * hw + p];
          if (oi) v *= act_grad_from_out(oi[(size_t)c * hw + p], act_kind, 0.f);
        }
        f[k] = v;
      }
      uint4 o;
      o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
      *reinterpret_cast<uint4*>(out + (size_t)p * Cp + c0) = o;
    }
  }
}

// sources of the pad adjoint along one axis (reflect: <= 3, replicate: border cell + its p pad cells, p <= 3)
__device__ __forceinline__ int img_fold_sources(int* idx, int x, int n, int p, int mode) {
  int cnt = 1;
  idx[0] = x + p;
  if (p > 0) {
    if (mode == GS_BORDER_REFLECT) {
      if (x >= 1 && x <= p) idx[cnt++] = p - x;
      if (x >= n - 1 - p && x <= n - 2) idx[cnt++] = p + 2 * (n - 1) - x;
    } else {
      if (x == 0) for (int k = 0; k < p; ++k) idx[cnt++] = k;
      if (x == n - 1) for (int k = 1; k <= p; ++k) idx[cnt++] = n - 1 + p + k;
    }
  }
  return cnt;
}

__global__ __launch_bounds__(256) void image_to_act_bwd_kernel(const unsigned short* g_pad, float* g_img, int C, int D,
                                                               int H, int W, int Cp, int fold, int mode,
                                                               int accumulate) {
  const int n = blockIdx.y;
  const int fd = D > 1 ? fold : 0;
  const int Dp = D + 2 * fd, Hp = H + 2 * fold, Wp = W + 2 * fold;
  const unsigned short* gp = g_pad + (size_t)n * Dp * Hp * Wp * Cp;
  const long long hw = (long long)D * H * W;
  float* out = g_img + (size_t)n * C * hw;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (long long)gridDim.x * blockDim.x) {
    const long long zi = p / W;
    const int iw = (int)(p - zi * W);
    const int iz = (int)(zi / H), ih = (int)(zi - (long long)iz * H);
    int ds[8], hs[8], ws[8];
    const int nd = img_fold_sources(ds, iz, D, fd, mode);
    const int nh = img_fold_sources(hs, ih, H, fold, mode);
    const int nw = img_fold_sources(ws, iw, W, fold, mode);
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
      for (int e = 0; e < nd; ++e)
        for (int a = 0; a < nh; ++a)
          for (int b = 0; b < nw; ++b) s += bf2f(gp[(((size_t)ds[e] * Hp + hs[a]) * Wp + ws[b]) * Cp + c]);
      if (accumulate) out[(size_t)c * hw + p] += s; else out[(size_t)c * hw + p] = s;
    }
  }
}

static inline dim3 img_grid(long long hw, int N) {
  long long bx = (hw + 255) / 256;
  if (bx > 1024) bx = 1024;
  return dim3((unsigned)bx, N);
}

extern "C" int gs_image_to_act(const float* img, void* act, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cp,
                               void* stream) {
  GS_REQUIRE(img && act && N > 0 && C > 0 && Cp >= C && (Cp & 7) == 0, "gs_image_to_act: bad argument");
  const long long hw = (long long)H * W;
  hipLaunchKernelGGL(image_to_act_kernel, img_grid(hw, N), dim3(256), 0, static_cast<hipStream_t>(stream), img,
                     static_cast<unsigned short*>(act), C, hw, Cp);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// Two images side by side along the channel axis (the conditional discriminator's input torch.cat([real_A, fake_B], dim=1),
// ganslate/nn/gans/paired/pix2pix.py:70,80) converted in one pass: channel c < Ca comes from a, the others from b.
__global__ __launch_bounds__(256) void image_pair_to_act_kernel(const float* a, int Ca, const float* b, int Cb,
                                                                unsigned short* act, long long hw, int Cp) {
  const int n = blockIdx.y;
  const float* ia = a + (size_t)n * Ca * hw;
  const float* ib = b + (size_t)n * Cb * hw;
  unsigned short* out = act + (size_t)n * hw * Cp;
  const int C = Ca + Cb;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (long long)gridDim.x * blockDim.x) {
    for (int c0 = 0; c0 < Cp; c0 += 8) {
      float f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = c0 + k;
        f[k] = c < Ca ? ia[(size_t)c * hw + p] : (c < C ? ib[(size_t)(c - Ca) * hw + p] : 0.f);
      }
      uint4 o;
      o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
      *reinterpret_cast<uint4*>(out + (size_t)p * Cp + c0) = o;
    }
  }
}
// its gradient: channels [0, Ca) of g to ga, [Ca, Ca + Cb) to gb (either may be null: that part needs no gradient)
__global__ __launch_bounds__(256) void image_pair_to_act_bwd_kernel(const unsigned short* g, float* ga, int Ca, float* gb,
                                                                    int Cb, long long hw, int Cp) {
  const int n = blockIdx.y;
  const unsigned short* gp = g + (size_t)n * hw * Cp;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (long long)gridDim.x * blockDim.x) {
    if (ga)
      for (int c = 0; c < Ca; ++c) ga[((size_t)n * Ca + c) * hw + p] = bf2f(gp[(size_t)p * Cp + c]);
    if (gb)
      for (int c = 0; c < Cb; ++c) gb[((size_t)n * Cb + c) * hw + p] = bf2f(gp[(size_t)p * Cp + Ca + c]);
  }
}

extern "C" int gs_image_pair_to_act(const float* a, int32_t Ca, const float* b, int32_t Cb, void* act, int32_t N, int32_t H,
                                    int32_t W, int32_t Cp, void* stream) {
  GS_REQUIRE(a && b && act && N > 0 && Ca > 0 && Cb > 0 && Cp >= Ca + Cb && (Cp & 7) == 0, "gs_image_pair_to_act: bad argument");
  const long long hw = (long long)H * W;
  hipLaunchKernelGGL(image_pair_to_act_kernel, img_grid(hw, N), dim3(256), 0, static_cast<hipStream_t>(stream), a, Ca, b, Cb,
                     static_cast<unsigned short*>(act), hw, Cp);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_image_pair_to_act_backward(const void* g, float* ga, int32_t Ca, float* gb, int32_t Cb, int32_t N, int32_t H,
                                             int32_t W, int32_t Cp, void* stream) {
  GS_REQUIRE(g && (ga || gb) && N > 0 && Ca > 0 && Cb > 0 && Cp >= Ca + Cb, "gs_image_pair_to_act_backward: bad argument");
  const long long hw = (long long)H * W;
  hipLaunchKernelGGL(image_pair_to_act_bwd_kernel, img_grid(hw, N), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(g), ga, Ca, gb, Cb, hw, Cp);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_act_to_image(const void* act, float* img, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cp,
                               int32_t act_kind, void* stream) {
  GS_REQUIRE(img && act && N > 0 && C > 0 && Cp >= C, "gs_act_to_image: bad argument");
  const long long hw = (long long)H * W;
  hipLaunchKernelGGL(act_to_image_kernel, img_grid(hw, N), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(act), img, C, hw, Cp, act_kind);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_act_to_image_backward(const float* g_img, const float* out_img, void* g_act, int32_t N, int32_t C,
                                        int32_t H, int32_t W, int32_t Cp, int32_t act_kind, void* stream) {
  GS_REQUIRE(g_img && g_act && N > 0 && C > 0 && Cp >= C && (Cp & 7) == 0, "gs_act_to_image_backward: bad argument");
  GS_REQUIRE(act_kind == GS_ACT_NONE || out_img, "gs_act_to_image_backward: activation needs the forward output");
  const long long hw = (long long)H * W;
  hipLaunchKernelGGL(act_to_image_bwd_kernel, img_grid(hw, N), dim3(256), 0, static_cast<hipStream_t>(stream), g_img,
                     act_kind == GS_ACT_NONE ? nullptr : out_img, static_cast<unsigned short*>(g_act), C, hw, Cp,
                     act_kind);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_image_to_act_backward(const void* g_pad, float* g_img, int32_t N, int32_t C, int32_t D, int32_t H,
                                        int32_t W, int32_t Cp, int32_t fold, int32_t fold_mode, int32_t accumulate,
                                        void* stream) {
  GS_REQUIRE(g_pad && g_img && N > 0 && C > 0 && Cp >= C && D > 0, "gs_image_to_act_backward: bad argument");
  GS_REQUIRE(fold == 0 || fold_mode == GS_BORDER_REFLECT || (fold_mode == GS_BORDER_REPLICATE && fold <= 3),
             "gs_image_to_act_backward: fold must be reflect, or replicate with fold <= 3");
  const long long hw = (long long)D * H * W;
  hipLaunchKernelGGL(image_to_act_bwd_kernel, img_grid(hw, N), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(g_pad), g_img, C, D, H, W, Cp, fold, fold_mode, accumulate);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
