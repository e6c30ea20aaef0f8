// Device-side image preprocessing for the image-folder datasets: what
// ganslate/data/utils/transforms.py:9-61 composes from torchvision / PIL on the host — Resize(load_size, BICUBIC),
// RandomCrop(final_size), RandomHorizontalFlip, ToTensor, Normalize(0.5, 0.5) — on decoded 8-bit images already in HBM.
//
// The resize is Pillow's, bit for bit (Pillow src/libImaging/Resample.c, ImagingResampleHorizontal_8bpc /
// ImagingResampleVertical_8bpc; 12.2.0 is the pinned release here): two separable passes, horizontal first, each
//     out = clip8((2^21 + sum_k in[xmin + k] * kk[k]) >> 22)
// with 22-bit fixed-point coefficients (built on the host in double exactly as precompute_coeffs / normalize_coeffs_8bpc do,
// ganslate_amd/data/device_transforms.py) and an 8-bit intermediate image. Byte work, HBM-bound: one thread per output
// pixel, all channels; the second pass only computes the pixels inside the crop window and writes them flipped /
// normalised straight into the fp32 NCHW batch tensor the networks' boundary kernels read.
#include "common.hpp"

namespace {
constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ unsigned clip8(int v) {
  v >>= PRECISION_BITS;                        // arithmetic shift, like Pillow's clip8 lookup index
  return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: out[y][xx][c], y over ALL input rows (the vertical pass picks the ones it needs)
template <int C>
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* in, unsigned char* out, int in_h, int in_w,
                                                         int out_w, const int* bounds, const int* kk, int ksize) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (xx >= out_w) return;
  const int xmin = bounds[xx * 2], xn = bounds[xx * 2 + 1];
  const int* k = kk + (size_t)xx * ksize;
  int acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
  const unsigned char* row = in + ((size_t)y * in_w + xmin) * C;
  for (int x = 0; x < xn; ++x) {
    const int w = k[x];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += (int)row[x * C + c] * w;
  }
  unsigned char* o = out + ((size_t)y * out_w + xx) * C;
#pragma unroll
  for (int c = 0; c < C; ++c) o[c] = (unsigned char)clip8(acc[c]);
}

// plain vertical pass, 8-bit out (the intermediate image between two resizes: scale_width / resize followed by random_zoom)
template <int C>
__global__ __launch_bounds__(256) void resample_v_kernel(const unsigned char* tmp, unsigned char* out, int w,
                                                         const int* bounds, const int* kk, int ksize) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int yy = blockIdx.y;
  if (x >= w) return;
  const int ymin = bounds[yy * 2], yn = bounds[yy * 2 + 1];
  const int* k = kk + (size_t)yy * ksize;
  int acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
  for (int y = 0; y < yn; ++y) {
    const unsigned char* p = tmp + ((size_t)(ymin + y) * w + x) * C;
    const int wgt = k[y];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += (int)p[c] * wgt;
  }
  unsigned char* o = out + ((size_t)yy * w + x) * C;
#pragma unroll
  for (int c = 0; c < C; ++c) o[c] = (unsigned char)clip8(acc[c]);
}

// vertical pass restricted to the crop window [top, top+fh) x [left, left+fw) of the resized image, then
// RandomHorizontalFlip, ToTensor (x / 255) and Normalize ((x - 0.5) / 0.5) in torchvision's fp32 operation order:
// out[c][i][j] fp32, planes fh*fw apart
template <int C>
__global__ __launch_bounds__(256) void resample_v_crop_norm_kernel(const unsigned char* tmp, float* out, int tmp_w,
                                                                   const int* bounds, const int* kk, int ksize, int top,
                                                                   int left, int fh, int fw, int flip) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= fw) return;
  const int yy = top + i;
  const int xs = left + (flip ? fw - 1 - j : j);
  const int ymin = bounds[yy * 2], yn = bounds[yy * 2 + 1];
  const int* k = kk + (size_t)yy * ksize;
  int acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
  for (int y = 0; y < yn; ++y) {
    const unsigned char* p = tmp + ((size_t)(ymin + y) * tmp_w + xs) * C;
    const int w = k[y];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += (int)p[c] * w;
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float v = (float)clip8(acc[c]) / 255.0f;
    out[((size_t)c * fh + i) * fw + j] = (v - 0.5f) / 0.5f;
  }
}
}  // namespace

extern "C" int gs_u8_resample_h(const void* in, void* out, int32_t in_h, int32_t in_w, int32_t out_w, int32_t C,
                                const int32_t* bounds, const int32_t* kk, int32_t ksize, void* stream) {
  GS_REQUIRE(in && out && bounds && kk && in_h > 0 && in_w > 0 && out_w > 0 && ksize > 0 && (C == 1 || C == 3),
             "gs_u8_resample_h: bad argument (C must be 1 or 3)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((out_w + 255) / 256, in_h);
  const unsigned char* i8 = static_cast<const unsigned char*>(in);
  unsigned char* o8 = static_cast<unsigned char*>(out);
  if (C == 3) hipLaunchKernelGGL(resample_h_kernel<3>, grid, dim3(256), 0, st, i8, o8, in_h, in_w, out_w, bounds, kk, ksize);
  else hipLaunchKernelGGL(resample_h_kernel<1>, grid, dim3(256), 0, st, i8, o8, in_h, in_w, out_w, bounds, kk, ksize);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_u8_resample_v(const void* tmp, void* out, int32_t tmp_h, int32_t w, int32_t out_h, int32_t C,
                                const int32_t* bounds, const int32_t* kk, int32_t ksize, void* stream) {
  GS_REQUIRE(tmp && out && bounds && kk && tmp_h > 0 && w > 0 && out_h > 0 && ksize > 0 && (C == 1 || C == 3),
             "gs_u8_resample_v: bad argument (C must be 1 or 3)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((w + 255) / 256, out_h);
  const unsigned char* t8 = static_cast<const unsigned char*>(tmp);
  unsigned char* o8 = static_cast<unsigned char*>(out);
  if (C == 3) hipLaunchKernelGGL(resample_v_kernel<3>, grid, dim3(256), 0, st, t8, o8, w, bounds, kk, ksize);
  else hipLaunchKernelGGL(resample_v_kernel<1>, grid, dim3(256), 0, st, t8, o8, w, bounds, kk, ksize);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_u8_resample_v_crop_normalize(const void* tmp, float* out, int32_t tmp_h, int32_t tmp_w, int32_t out_h,
                                               int32_t C, const int32_t* bounds, const int32_t* kk, int32_t ksize,
                                               int32_t top, int32_t left, int32_t fh, int32_t fw, int32_t flip,
                                               void* stream) {
  GS_REQUIRE(tmp && out && bounds && kk && tmp_h > 0 && tmp_w > 0 && out_h > 0 && ksize > 0 && (C == 1 || C == 3),
             "gs_u8_resample_v_crop_normalize: bad argument (C must be 1 or 3)");
  GS_REQUIRE(top >= 0 && left >= 0 && fh > 0 && fw > 0 && top + fh <= out_h && left + fw <= tmp_w,
             "gs_u8_resample_v_crop_normalize: crop window [%d+%d, %d+%d] outside the %d x %d resized image", top, fh,
             left, fw, out_h, tmp_w);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((fw + 255) / 256, fh);
  const unsigned char* t8 = static_cast<const unsigned char*>(tmp);
  if (C == 3)
    hipLaunchKernelGGL(resample_v_crop_norm_kernel<3>, grid, dim3(256), 0, st, t8, out, tmp_w, bounds, kk, ksize, top, left,
                       fh, fw, flip);
  else
    hipLaunchKernelGGL(resample_v_crop_norm_kernel<1>, grid, dim3(256), 0, st, t8, out, tmp_w, bounds, kk, ksize, top, left,
                       fh, fw, flip);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
