// "W-fold" of the k7 boundary convolutions of the ResNet generators (resnet2d.py:24-25,64-65; resnet3d.py:24-25,64):
// a conv with very few input (stem) or output (last layer) channels wastes the 16-wide MFMA tile, so the taps of the
// W axis are moved into the channel axis and the conv runs with a k x k x 1 kernel:
//   stem:  x'[n,z,i,j][dw*C + c] = x[n][c][z][i][B(j + dw - p)]                      ("unfold", here fused with the
//          NCHW fp32 -> NHWC bf16 conversion), conv' has k*C input channels and no W taps;
//   last:  z'[n,z,i,j'][dw*Co + co] = sum over (dd,dh,ci) taps, j' in [0, W + 2p)     (conv' with k*Co output channels),
//          out[n][co][z][i][j] = act(bias[co] + sum_dw z'[n,z,i,j+dw][dw*Co + co])   ("shift-add", fused with the
//          NHWC bf16 -> NCHW fp32 conversion and tanh).
// The four kernels here are those two boundary transforms and their adjoints; the convolutions in between are ordinary
// gs_gconv_forward / gs_wgrad launches on the transformed layer. HBM-bound, one thread per pixel.
#include "common.hpp"

// source positions (padded coordinates) that the padding maps onto x: reflect <= 3, replicate <= p+1 (p <= 3)
__device__ __forceinline__ int wf_sources(int* idx, int x, int n, int p, int mode) {
  int cnt = 1;
  idx[0] = x + p;
  if (p > 0) {
    if (mode == GS_BORDER_REFLECT) {
      if (x >= 1 && x <= p) idx[cnt++] = p - x;
      if (x >= n - 1 - p && x <= n - 2) idx[cnt++] = p + 2 * (n - 1) - x;
    } else if (mode == GS_BORDER_REPLICATE) {
      if (x == 0) for (int k = 0; k < p; ++k) idx[cnt++] = k;
      if (x == n - 1) for (int k = 1; k <= p; ++k) idx[cnt++] = n - 1 + p + k;
    }
  }
  return cnt;
}

__global__ __launch_bounds__(256) void image_unfold_kernel(const float* img, unsigned short* act, int C, long long rows,
                                                           int W, int Qp, int k, int p, int mode) {
  const int n = blockIdx.y;
  const long long hw = rows * W;
  const float* in = img + (size_t)n * C * hw;
  unsigned short* out = act + (size_t)n * hw * Qp;
  for (long long px = (long long)blockIdx.x * 256 + threadIdx.x; px < hw; px += (long long)gridDim.x * 256) {
    const long long r = px / W;
    const int j = (int)(px - r * W);
    for (int q0 = 0; q0 < Qp; q0 += 8) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = q0 + e;
        float v = 0.f;
        if (q < k * C) {
          const int dw = q / C, c = q - dw * C;
          bool ok = true;
          const int jj = border_index(j + dw - p, W, mode, ok);
          if (ok) v = in[(size_t)c * hw + r * W + jj];
        }
        f[e] = v;
      }
      uint4 o;
      o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
      *reinterpret_cast<uint4*>(out + (size_t)px * Qp + q0) = o;
    }
  }
}

// g_img[n][c][z][i][j] (+)= sum over the (depth,row) fold sources and over dw, j' with B(j'+dw-p) = j of
// g[n, zs, is, j'][dw*C + c]; g lives on the domain padded by `fold` along depth (if D > 1) and rows, not along W
__global__ __launch_bounds__(256) void image_unfold_bwd_kernel(const unsigned short* g, float* g_img, int C, int D,
                                                               int H, int W, int Qp, int k, int p, int fold, int mode,
                                                               int accumulate) {
  const int n = blockIdx.y;
  const int fd = D > 1 ? fold : 0;
  const int Dp = D + 2 * fd, Hp = H + 2 * fold;
  const unsigned short* gp = g + (size_t)n * Dp * Hp * W * Qp;
  const long long hw = (long long)D * H * W;
  float* out = g_img + (size_t)n * C * hw;
  for (long long px = (long long)blockIdx.x * 256 + threadIdx.x; px < hw; px += (long long)gridDim.x * 256) {
    const long long zi = px / W;
    const int j = (int)(px - zi * W);
    const int iz = (int)(zi / H), ih = (int)(zi - (long long)iz * H);
    int ds[8], hs[8], ws[8];
    const int nd = wf_sources(ds, iz, D, fd, mode);
    const int nh = wf_sources(hs, ih, H, fold, mode);
    const int nw = wf_sources(ws, j, W, p, mode);       // W positions in coordinates padded by p
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
      for (int a = 0; a < nd; ++a)
        for (int b = 0; b < nh; ++b) {
          const unsigned short* row = gp + ((size_t)ds[a] * Hp + hs[b]) * W * Qp;
          for (int e = 0; e < nw; ++e)
            for (int dw = 0; dw < k; ++dw) {
              const int jp = ws[e] - dw;
              if (jp >= 0 && jp < W) s += bf2f(row[(size_t)jp * Qp + dw * C + c]);
            }
        }
      if (accumulate) out[(size_t)c * hw + px] += s; else out[(size_t)c * hw + px] = s;
    }
  }
}

__global__ __launch_bounds__(256) void shiftadd_to_image_kernel(const unsigned short* z, const float* bias, float* img,
                                                                int Co, long long rows, int W, int Pp, int k,
                                                                int act_kind) {
  const int n = blockIdx.y;
  const int Wx = W + k - 1;
  const unsigned short* in = z + (size_t)n * rows * Wx * Pp;
  const long long hw = rows * W;
  float* out = img + (size_t)n * Co * hw;
  for (long long px = (long long)blockIdx.x * 256 + threadIdx.x; px < hw; px += (long long)gridDim.x * 256) {
    const long long r = px / W;
    const int j = (int)(px - r * W);
    const unsigned short* row = in + ((size_t)r * Wx + j) * Pp;
    for (int co = 0; co < Co; ++co) {
      float s = bias ? bias[co] : 0.f;
      for (int dw = 0; dw < k; ++dw) s += bf2f(row[(size_t)dw * Pp + dw * Co + co]);
      out[(size_t)co * hw + px] = apply_act(s, act_kind, 0.f);
    }
  }
}

__global__ __launch_bounds__(256) void shiftadd_bwd_kernel(const float* g_img, const float* out_img, unsigned short* gz,
                                                           int Co, long long rows, int W, int Pp, int k, int act_kind) {
  const int n = blockIdx.y;
  const int Wx = W + k - 1;
  const long long hw = rows * W, hwx = rows * Wx;
  const float* gi = g_img + (size_t)n * Co * hw;
  const float* oi = out_img ? out_img + (size_t)n * Co * hw : nullptr;
  unsigned short* out = gz + (size_t)n * hwx * Pp;
  for (long long px = (long long)blockIdx.x * 256 + threadIdx.x; px < hwx; px += (long long)gridDim.x * 256) {
    const long long r = px / Wx;
    const int jp = (int)(px - r * Wx);
    for (int q0 = 0; q0 < Pp; q0 += 8) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = q0 + e;
        float v = 0.f;
        if (q < k * Co) {
          const int dw = q / Co, co = q - dw * Co;
          const int j = jp - dw;
          if (j >= 0 && j < W) {
            const size_t o = (size_t)co * hw + r * W + j;
            v = gi[o];
            if (oi) v *= act_grad_from_out(oi[o], act_kind, 0.f);
          }
        }
        f[e] = v;
      }
      uint4 o;
      o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
      *reinterpret_cast<uint4*>(out + (size_t)px * Pp + q0) = o;
    }
  }
}


// ---- row-staged forms (round 3) -------------------------------------------------------------------------------------------
// The one-thread-per-pixel kernels above read their operand as scalar bf16 / fp32 gathers: 21 two-byte loads per pixel, 48 B
// apart between neighbouring threads, for the output conv's shift-add — 52 us for a 26 MB tensor whose stream time is 6 us
// (profiles/r02_step_by_grid_v7.txt: 0.54 ms per training step in the four boundary transforms). Here a workgroup stages the
// row segment it needs ONCE with 16-byte coalesced loads (or coalesced fp32 loads for the NCHW side) and every thread reads
// its taps out of LDS. Same arithmetic, same summation order per output element -> bit-identical results.
constexpr int WF_SEG = 256;

__global__ __launch_bounds__(256) void shiftadd_to_image_row_kernel(const unsigned short* z, const float* bias, float* img,
                                                                    int Co, long long rows, int W, int Pp, int k,
                                                                    int act_kind, int segs) {
  extern __shared__ __attribute__((aligned(16))) char wf_smem[];
  unsigned short* lds = reinterpret_cast<unsigned short*>(wf_smem);
  const int n = blockIdx.y;
  const long long r = blockIdx.x / segs;
  const int j0 = (int)(blockIdx.x - r * segs) * WF_SEG;
  const int Wx = W + k - 1;
  const int npx = min(WF_SEG, W - j0);
  const int pieces = (npx + k - 1) * Pp / 8;
  const uint4* src = reinterpret_cast<const uint4*>(z + (((size_t)n * rows + r) * Wx + j0) * Pp);
  for (int q = threadIdx.x; q < pieces; q += 256) reinterpret_cast<uint4*>(lds)[q] = src[q];
  __syncthreads();
  const int j = threadIdx.x;
  if (j < npx) {
    const long long hw = rows * W;
    float* out = img + (size_t)n * Co * hw + r * W + j0 + j;
    const unsigned short* row = lds + (size_t)j * Pp;
    for (int co = 0; co < Co; ++co) {
      float s = bias ? bias[co] : 0.f;
      for (int dw = 0; dw < k; ++dw) s += bf2f(row[dw * Pp + dw * Co + co]);
      out[(size_t)co * hw] = apply_act(s, act_kind, 0.f);
    }
  }
}

__global__ __launch_bounds__(256) void shiftadd_bwd_row_kernel(const float* g_img, const float* out_img, unsigned short* gz,
                                                               int Co, long long rows, int W, int Pp, int k, int act_kind,
                                                               int segs) {
  extern __shared__ __attribute__((aligned(16))) char wf_smem[];
  float* lds = reinterpret_cast<float*>(wf_smem);          // [Co][WF_SEG + k - 1]: g * act'(out) of image pixels jp0-(k-1) ..
  const int n = blockIdx.y;
  const long long r = blockIdx.x / segs;
  const int jp0 = (int)(blockIdx.x - r * segs) * WF_SEG;
  const int Wx = W + k - 1;
  const int nvx = min(WF_SEG, Wx - jp0);
  const int L = WF_SEG + k - 1;
  const long long hw = rows * W;
  const float* gi = g_img + (size_t)n * Co * hw + r * W;
  const float* oi = out_img ? out_img + (size_t)n * Co * hw + r * W : nullptr;
  for (int idx = threadIdx.x; idx < Co * L; idx += 256) {
    const int co = idx / L, i = idx - co * L;
    const int j = jp0 - (k - 1) + i;
    float v = 0.f;
    if (j >= 0 && j < W) {
      v = gi[(size_t)co * hw + j];
      if (oi) v *= act_grad_from_out(oi[(size_t)co * hw + j], act_kind, 0.f);
    }
    lds[idx] = v;
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < nvx) {
    unsigned short* out = gz + (((size_t)n * rows + r) * Wx + jp0 + t) * Pp;
    for (int q0 = 0; q0 < Pp; q0 += 8) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = q0 + e;
        float v = 0.f;
        if (q < k * Co) {
          const int dw = q / Co, co = q - dw * Co;
          v = lds[co * L + t + (k - 1) - dw];               // image pixel jp - dw (zero outside the image)
        }
        f[e] = v;
      }
      uint4 o;
      o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
      *reinterpret_cast<uint4*>(out + q0) = o;
    }
  }
}

__global__ __launch_bounds__(256) void image_unfold_row_kernel(const float* img, unsigned short* act, int C, long long rows,
                                                               int W, int Qp, int k, int p, int mode, int segs) {
  extern __shared__ __attribute__((aligned(16))) char wf_smem[];
  float* lds = reinterpret_cast<float*>(wf_smem);          // [C][WF_SEG + k - 1]: pixels B(j0 - p + i) of the row
  const int n = blockIdx.y;
  const long long r = blockIdx.x / segs;
  const int j0 = (int)(blockIdx.x - r * segs) * WF_SEG;
  const int npx = min(WF_SEG, W - j0);
  const int L = WF_SEG + k - 1;
  const long long hw = rows * W;
  const float* in = img + (size_t)n * C * hw + r * W;
  for (int idx = threadIdx.x; idx < C * L; idx += 256) {
    const int c = idx / L, i = idx - c * L;
    bool ok = true;
    const int jj = border_index(j0 + i - p, W, mode, ok);
    lds[idx] = (ok && i < npx + k - 1) ? in[(size_t)c * hw + jj] : 0.f;
  }
  __syncthreads();
  const int j = threadIdx.x;
  if (j < npx) {
    unsigned short* out = act + (((size_t)n * rows + r) * W + j0 + j) * Qp;
    for (int q0 = 0; q0 < Qp; q0 += 8) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = q0 + e;
        float v = 0.f;
        if (q < k * C) {
          const int dw = q / C, c = q - dw * C;
          v = lds[c * L + j + dw];
        }
        f[e] = v;
      }
      uint4 o;
      o.x = pack_bf2(f[0], f[1]); o.y = pack_bf2(f[2], f[3]); o.z = pack_bf2(f[4], f[5]); o.w = pack_bf2(f[6], f[7]);
      *reinterpret_cast<uint4*>(out + q0) = o;
    }
  }
}

// one workgroup per output row (iz, ih); every source row (fold sources of depth x rows) is staged whole
__global__ __launch_bounds__(256) void image_unfold_bwd_row_kernel(const unsigned short* g, float* g_img, int C, int D,
                                                                   int H, int W, int Qp, int k, int p, int fold, int mode,
                                                                   int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char wf_smem[];
  unsigned short* lds = reinterpret_cast<unsigned short*>(wf_smem);      // [W][Qp]
  const int n = blockIdx.y;
  const int fd = D > 1 ? fold : 0;
  const int Dp = D + 2 * fd, Hp = H + 2 * fold;
  const unsigned short* gp = g + (size_t)n * Dp * Hp * W * Qp;
  const long long hw = (long long)D * H * W;
  const int iz = blockIdx.x / H, ih = blockIdx.x - iz * H;
  float* out = g_img + (size_t)n * C * hw + ((size_t)iz * H + ih) * W;
  int ds[8], hs[8];
  const int nd = wf_sources(ds, iz, D, fd, mode);
  const int nh = wf_sources(hs, ih, H, fold, mode);
  constexpr int MAXJ = 4, MAXC = 4;                      // pixels per thread (W <= 1024), channels
  float acc[MAXJ][MAXC];
#pragma unroll
  for (int u = 0; u < MAXJ; ++u)
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[u][c] = 0.f;
  const int pieces = W * Qp / 8;
  for (int a = 0; a < nd; ++a)
    for (int b = 0; b < nh; ++b) {
      const uint4* src = reinterpret_cast<const uint4*>(gp + ((size_t)ds[a] * Hp + hs[b]) * W * Qp);
      __syncthreads();
      for (int q = threadIdx.x; q < pieces; q += 256) reinterpret_cast<uint4*>(lds)[q] = src[q];
      __syncthreads();
#pragma unroll
      for (int u = 0; u < MAXJ; ++u) {
        const int j = threadIdx.x + u * 256;
        if (j < W) {
          int ws[8];
          const int nw = wf_sources(ws, j, W, p, mode);
          for (int c = 0; c < C; ++c) {
            float s = 0.f;
            for (int e = 0; e < nw; ++e)
              for (int dw = 0; dw < k; ++dw) {
                const int jp = ws[e] - dw;
                if (jp >= 0 && jp < W) s += bf2f(lds[(size_t)jp * Qp + dw * C + c]);
              }
            acc[u][c] += s;
          }
        }
      }
    }
#pragma unroll
  for (int u = 0; u < MAXJ; ++u) {
    const int j = threadIdx.x + u * 256;
    if (j < W)
      for (int c = 0; c < C; ++c) {
        if (accumulate) out[(size_t)c * hw + j] += acc[u][c]; else out[(size_t)c * hw + j] = acc[u][c];
      }
  }
}

static inline dim3 wf_grid(long long pixels, int N) {
  long long bx = (pixels + 255) / 256;
  if (bx > 2048) bx = 2048;
  return dim3((unsigned)bx, N);
}

extern "C" int gs_image_unfold(const float* img, void* act, int32_t N, int32_t C, int64_t rows, int32_t W, int32_t Qp,
                               int32_t k, int32_t p, int32_t border, void* stream) {
  GS_REQUIRE(img && act && N > 0 && C > 0 && rows > 0 && W > 0 && k > 0 && Qp >= k * C && (Qp & 7) == 0,
             "gs_image_unfold: bad argument (Qp must be a multiple of 8 and >= k*C)");
  const long long segs = (W + WF_SEG - 1) / WF_SEG;
  if (gs_opt(GS_OPT_WFOLD_ROWS) && rows * segs < (1LL << 31) && N <= 65535) {
    const int lds = C * (WF_SEG + k - 1) * 4;
    hipLaunchKernelGGL(image_unfold_row_kernel, dim3((unsigned)(rows * segs), N), dim3(256), lds,
                       static_cast<hipStream_t>(stream), img, static_cast<unsigned short*>(act), C, (long long)rows, W, Qp, k,
                       p, border, (int)segs);
    GS_CHECK_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(image_unfold_kernel, wf_grid(rows * W, N), dim3(256), 0, static_cast<hipStream_t>(stream), img,
                     static_cast<unsigned short*>(act), C, (long long)rows, W, Qp, k, p, border);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_image_unfold_backward(const void* g, float* g_img, int32_t N, int32_t C, int32_t D, int32_t H,
                                        int32_t W, int32_t Qp, int32_t k, int32_t p, int32_t fold, int32_t border,
                                        int32_t accumulate, void* stream) {
  GS_REQUIRE(g && g_img && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && k > 0 && Qp >= k * C,
             "gs_image_unfold_backward: bad argument");
  GS_REQUIRE(border != GS_BORDER_REPLICATE || (fold <= 3 && p <= 3), "gs_image_unfold_backward: replicate needs pad <= 3");
  // the sum over fold sources is taken source row by source row here, tap by tap inside a row, as in the per-pixel kernel
  // when there is one source row; border rows (several fold sources) add the rows' partial sums in the same (a, b) order
  if (gs_opt(GS_OPT_WFOLD_ROWS) && C <= 4 && W <= 1024 && (Qp & 7) == 0 && (long long)W * Qp * 2 <= 48 * 1024 && (long long)D * H < (1LL << 31) && N <= 65535) {
    hipLaunchKernelGGL(image_unfold_bwd_row_kernel, dim3((unsigned)((long long)D * H), N), dim3(256), W * Qp * 2,
                       static_cast<hipStream_t>(stream), static_cast<const unsigned short*>(g), g_img, C, D, H, W, Qp, k, p,
                       fold, border, accumulate);
    GS_CHECK_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(image_unfold_bwd_kernel, wf_grid((long long)D * H * W, N), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned short*>(g), g_img, C, D, H, W, Qp, k, p,
                     fold, border, accumulate);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_shiftadd_to_image(const void* z, const float* bias, float* img, int32_t N, int32_t Co, int64_t rows,
                                    int32_t W, int32_t Pp, int32_t k, int32_t act_kind, void* stream) {
  GS_REQUIRE(z && img && N > 0 && Co > 0 && rows > 0 && W > 0 && k > 0 && Pp >= k * Co, "gs_shiftadd_to_image: bad argument");
  const long long segs = (W + WF_SEG - 1) / WF_SEG;
  if (gs_opt(GS_OPT_WFOLD_ROWS) && (Pp & 7) == 0 && Pp <= 64 && rows * segs < (1LL << 31) && N <= 65535) {
    const int lds = (WF_SEG + k - 1) * Pp * 2;
    hipLaunchKernelGGL(shiftadd_to_image_row_kernel, dim3((unsigned)(rows * segs), N), dim3(256), lds,
                       static_cast<hipStream_t>(stream), static_cast<const unsigned short*>(z), bias, img, Co,
                       (long long)rows, W, Pp, k, act_kind, (int)segs);
    GS_CHECK_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(shiftadd_to_image_kernel, wf_grid(rows * W, N), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(z), bias, img, Co, (long long)rows, W, Pp, k, act_kind);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_shiftadd_to_image_backward(const float* g_img, const float* out_img, void* gz, int32_t N, int32_t Co,
                                             int64_t rows, int32_t W, int32_t Pp, int32_t k, int32_t act_kind,
                                             void* stream) {
  GS_REQUIRE(g_img && gz && N > 0 && Co > 0 && rows > 0 && W > 0 && k > 0 && Pp >= k * Co && (Pp & 7) == 0,
             "gs_shiftadd_to_image_backward: bad argument");
  GS_REQUIRE(act_kind == GS_ACT_NONE || out_img, "gs_shiftadd_to_image_backward: activation needs the forward output");
  const long long segs = (W + k - 1 + WF_SEG - 1) / WF_SEG;
  if (gs_opt(GS_OPT_WFOLD_ROWS) && Co <= 16 && rows * segs < (1LL << 31) && N <= 65535) {
    const int lds = Co * (WF_SEG + k - 1) * 4;
    hipLaunchKernelGGL(shiftadd_bwd_row_kernel, dim3((unsigned)(rows * segs), N), dim3(256), lds,
                       static_cast<hipStream_t>(stream), g_img, act_kind == GS_ACT_NONE ? nullptr : out_img,
                       static_cast<unsigned short*>(gz), Co, (long long)rows, W, Pp, k, act_kind, (int)segs);
    GS_CHECK_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(shiftadd_bwd_kernel, wf_grid(rows * (W + k - 1), N), dim3(256), 0, static_cast<hipStream_t>(stream),
                     g_img, act_kind == GS_ACT_NONE ? nullptr : out_img, static_cast<unsigned short*>(gz), Co,
                     (long long)rows, W, Pp, k, act_kind);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
