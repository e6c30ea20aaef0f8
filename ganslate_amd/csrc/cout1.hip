// The PatchGAN's last layer (ganslate/nn/discriminators/patchgan/patchgan2d.py:62: Conv2d(8 ndf, 1, k4, s1, p1)) has ONE output
// channel: 1.2 GFLOP over 30 MB of input at batch 32. On the matrix-core kernels (a 256 x 16 im2col tile, a 16 x 256 weight-
// gradient tile) it is a 16-fold L2 gather for an eighth of an MFMA's rows — 0.13 + 0.09 ms per step at 10-30 TFLOP/s
// (profiles/r04_conv_table_v1.txt). It is a dot product, so here it runs on the vector ALUs: a wave holds the filter's 4 x 4 x Ci
// weights in registers (8 channels per lane, Ci <= 512), slides a 4 x 4 window of packed bf16 input pixels along an output
// row (each input pixel is loaded once per kernel row) and reduces over its lanes: v_dot2c_f32_bf16, fp32 accumulation.
// The weight gradient walks the INPUT pixels instead: x[q] (8 channels per lane) times the 4 x 4 patch of dy around q into
// 16 x 8 accumulators per lane, per-workgroup slabs added in a fixed order (no atomics).
#include "common.hpp"

typedef __attribute__((ext_vector_type(2))) __bf16 gs_bf2;
__device__ __forceinline__ float dot8(const uint4 a, const uint4 b, float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gs_bf2, a.x), __builtin_bit_cast(gs_bf2, b.x), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gs_bf2, a.y), __builtin_bit_cast(gs_bf2, b.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gs_bf2, a.z), __builtin_bit_cast(gs_bf2, b.z), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gs_bf2, a.w), __builtin_bit_cast(gs_bf2, b.w), acc, false);
  return acc;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {      // (lane 0 holds the total)
  v = row16_sum(v);                                   // lanes 0, 16, 32, 48 hold their rows' sums
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

struct Cout1K {
  const char* in;
  const char* w;              // row 0 of the [w_rows][Kp] pack: [16 taps][Ci]
  const float* bias;
  char* out;
  int N, Hi, Wi, Ci, in_cs, in_co, Ho, Wo, out_cs, out_co, h0, w0, act;
  float slope;
  int nsplit;
  long long w_delta, bias_delta;
};

// one workgroup per output row (n, y); its 4 waves split the columns
__global__ __launch_bounds__(256) void cout1_fwd_kernel(const Cout1K p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x / p.Ho, y = blockIdx.x - n * p.Ho;
  const bool live = lane * 8 < p.Ci;
  const char* wn = p.w + (n >= p.nsplit ? p.w_delta : 0);
  uint4 wr[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
    wr[t] = live ? *reinterpret_cast<const uint4*>(wn + ((size_t)t * p.Ci + lane * 8) * 2) : uint4{0u, 0u, 0u, 0u};
  const float b0 = p.bias ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.bias) + (n >= p.nsplit ? p.bias_delta : 0))
                          : 0.f;
  const int per = ((p.Wo + 3) / 4 + 3) & ~3;          // a multiple of 4: the window slots of a wave's first column are static
  const int xs = wave * per, xe = min(p.Wo, xs + per);
  if (xs >= xe) return;
  const char* in_n = p.in + ((size_t)n * p.Hi * p.Wi * p.in_cs + p.in_co + lane * 8) * 2;
  auto load = [&](int r, int ix) -> uint4 {          // input pixel (y + h0 + r, ix) of this lane's 8 channels, zero border
    const int iy = y + p.h0 + r;
    if (!live || iy < 0 || iy >= p.Hi || ix < 0 || ix >= p.Wi) return uint4{0u, 0u, 0u, 0u};
    return *reinterpret_cast<const uint4*>(in_n + ((size_t)iy * p.Wi + ix) * p.in_cs * 2);
  };
  uint4 win[4][4];                                    // [kernel row][column slot]; slot (x + s) & 3 holds column x + w0 + s
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) win[r][s] = load(r, xs + p.w0 + s);      // (xs & 3 == 0)
  char* out_row = p.out + (((size_t)n * p.Ho + y) * p.Wo * p.out_cs + p.out_co) * 2;
  // unrolled by 4 so that the window slots are compile-time registers (x & 3 = rot)
  for (int x0 = xs; x0 < xe; x0 += 4) {
#pragma unroll
    for (int rot = 0; rot < 4; ++rot) {
      const int x = x0 + rot;
      if (x >= xe) break;
#pragma unroll
      for (int r = 0; r < 4; ++r) win[r][(rot + 3) & 3] = load(r, x + p.w0 + 3);
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = dot8(win[r][(rot + s) & 3], wr[r * 4 + s], acc);
      acc = wave_sum_dpp(acc);
      if (lane == 0) {
        const float v = apply_act_small(acc + b0, p.act, p.slope);
        *reinterpret_cast<uint4*>(out_row + (size_t)x * p.out_cs * 2) = uint4{pack_bf2(v, 0.f), 0u, 0u, 0u};
      }
    }
  }
}

static bool cout1_grid(const signed char* dh, const signed char* dw, int T, int* h0, int* w0) {
  if (T != 16) return false;
  *h0 = dh[0]; *w0 = dw[0];
  for (int t = 0; t < 16; ++t)
    if (dh[t] != *h0 + t / 4 || dw[t] != *w0 + t % 4) return false;
  return true;
}

extern "C" int gs_conv_cout1_eligible(const gs_gconv_desc* d) {
  if (!d) return 0;
  int h0, w0;
  return d->so == 1 && d->si == 1 && d->Di == 1 && d->Do == 1 && d->Dc == 1 && !d->py && !d->px && !d->pz && d->Co == 8 &&
         d->out_cs >= 8 && d->Ci % 8 == 0 && d->Ci <= 512 && d->Hc == d->Ho && d->Wc == d->Wo && d->stats_slots == 0 &&
         !d->accumulate && d->border == GS_BORDER_ZERO && cout1_grid(d->dh, d->dw, d->T, &h0, &w0) &&
         (long long)d->N * d->Ho < (1LL << 31);
}

extern "C" int gs_conv_cout1_forward(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out,
                                     const gs_twin* tw, void* stream) {
  GS_REQUIRE(d && in && w_pack && out, "gs_conv_cout1_forward: null argument");
  GS_REQUIRE(gs_conv_cout1_eligible(d), "gs_conv_cout1_forward: not a 4 x 4 stride-1 layer with one output channel (gs_conv_cout1_eligible)");
  Cout1K k;
  k.in = static_cast<const char*>(in); k.w = static_cast<const char*>(w_pack); k.bias = bias; k.out = static_cast<char*>(out);
  k.N = d->N; k.Hi = d->Hi; k.Wi = d->Wi; k.Ci = d->Ci; k.in_cs = d->in_cs; k.in_co = d->in_co;
  k.Ho = d->Ho; k.Wo = d->Wo; k.out_cs = d->out_cs; k.out_co = d->out_co;
  cout1_grid(d->dh, d->dw, d->T, &k.h0, &k.w0);
  k.act = d->act; k.slope = d->slope;
  k.nsplit = tw ? tw->n_split : 0x7fffffff;
  k.w_delta = tw ? tw->w_delta : 0;
  k.bias_delta = tw ? tw->bias_delta : 0;
  hipLaunchKernelGGL(cout1_fwd_kernel, dim3((unsigned)(d->N * d->Ho)), dim3(256), 0, static_cast<hipStream_t>(stream), k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- weight gradient: dw[0][t][c] += sum_{n, q} dy[n][q - (dh_t, dw_t)][0] * x[n][q][c] --------------------------------------
struct Cout1WK {
  const char* a;              // dy [N][Ha][Wa][a_cs], channel a_co is the real one
  const char* g;              // x  [N][Hg][Wg][g_cs]
  float* ws;                  // [workgroups][16][Q] partial sums
  int N, Ha, Wa, a_cs, a_co, Hg, Wg, Q, g_cs, g_co, h0, w0, rows_per_wg, wgs_per_img;
};

// workgroup = (image, block of input rows); wave w takes the rows w, w + 4, ... of the block
__global__ __launch_bounds__(256) void cout1_wgrad_kernel(const Cout1WK p) {
  __shared__ float dyl[40][40];                       // dy of this image with a zero frame of 4 (Ha, Wa <= 32)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x / p.wgs_per_img, rb = blockIdx.x - n * p.wgs_per_img;
  for (int e = threadIdx.x; e < 40 * 40; e += 256) {
    const int yy = e / 40 - 4, xx = e % 40 - 4;
    float v = 0.f;
    if (yy >= 0 && yy < p.Ha && xx >= 0 && xx < p.Wa)
      v = bf2f(*reinterpret_cast<const unsigned short*>(p.a + ((((size_t)n * p.Ha + yy) * p.Wa + xx) * p.a_cs + p.a_co) * 2));
    dyl[e / 40][e % 40] = v;
  }
  __syncthreads();
  const bool live = lane * 8 < p.Q;
  float acc[16][8];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[t][k] = 0.f;
  const char* g_n = p.g + ((size_t)n * p.Hg * p.Wg * p.g_cs + p.g_co + lane * 8) * 2;
  const int y0 = rb * p.rows_per_wg, y1 = min(p.Hg, y0 + p.rows_per_wg);
  for (int qy = y0 + wave; qy < y1; qy += 4) {
    auto ldx = [&](int qx) -> uint4 {
      return (live && qx < p.Wg) ? *reinterpret_cast<const uint4*>(g_n + ((size_t)qy * p.Wg + qx) * p.g_cs * 2) : uint4{0u, 0u, 0u, 0u};
    };
    uint4 nx0 = ldx(0), nx1 = ldx(1), nx2 = ldx(2);      // three pixels in flight (a load per iteration waited a round trip)
    for (int qx = 0; qx < p.Wg; ++qx) {
      const uint4 xv = nx0;
      nx0 = nx1; nx1 = nx2; nx2 = ldx(qx + 3);
      const float xf[8] = {bf_lo(xv.x), bf_hi(xv.x), bf_lo(xv.y), bf_hi(xv.y), bf_lo(xv.z), bf_hi(xv.z), bf_lo(xv.w), bf_hi(xv.w)};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // output pixel that reads q through tap (h0 + r, w0 + s): (qy - h0 - r, qx - w0 - s); +4: the zero frame
        const float* drow = &dyl[qy - p.h0 - r + 4][qx - p.w0 - 3 + 4];       // s = 3 .. 0 -> ascending addresses
        const float d3 = drow[0], d2 = drow[1], d1 = drow[2], d0 = drow[3];
        const float ds[4] = {d0, d1, d2, d3};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[r * 4 + s][k] += ds[s] * xf[k];
      }
    }
  }
  // the four waves' sums in wave order (fixed), then the workgroup's slab
  float* slab = p.ws + (size_t)blockIdx.x * 16 * p.Q;
  __shared__ float wsum[16][512];
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (wave == w && live) {
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (w == 0) wsum[t][lane * 8 + k] = acc[t][k];
          else wsum[t][lane * 8 + k] += acc[t][k];
        }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < 16 * p.Q; e += 256) slab[e] = wsum[e / p.Q][e % p.Q];
}

extern "C" int gs_wgrad_cout1_eligible(const gs_wgrad_desc* d) {
  if (!d) return 0;
  int hh, ww;
  return d->si == 1 && d->Da == 1 && d->Dg == 1 && d->P == 8 && d->Q % 8 == 0 && d->Q <= 512 && d->Ha <= 32 && d->Wa <= 32 &&
         d->border == GS_BORDER_ZERO && d->dw_ld == d->T * d->Q && cout1_grid(d->dh, d->dw_, d->T, &hh, &ww) &&
         hh >= -3 && hh <= 0 && ww >= -3 && ww <= 0 &&
         // a wave per input row: below half a chip of workgroups (batch 1-2) the matrix-core kernel's pixel split is ahead
         (long long)d->N * ((d->Hg + 3) / 4) >= 128;
}
// 4 input rows per workgroup = one per wave (8 rows left half of the chip idle at batch 16-32: 57 -> 3x us)
static constexpr int kCout1Rows = 4;
static int cout1_wgs_per_img(const gs_wgrad_desc* d) { return (d->Hg + kCout1Rows - 1) / kCout1Rows; }
extern "C" int64_t gs_wgrad_cout1_ws_floats(const gs_wgrad_desc* d) {
  return gs_wgrad_cout1_eligible(d) ? (int64_t)d->N * cout1_wgs_per_img(d) * 16 * d->Q : -1;
}
// wgrad.hip: dst[e] += slab 0 [e] + slab 1 [e] + ... in slab order (nets = 2: the second network's slabs / buffer as well)
void gs_launch_slab_reduce(const float* ws, float* dst, long long n4, int slabs, long long stride4, hipStream_t st, int nets,
                           long long ws_y4, long long dw_y4);
extern "C" int gs_wgrad_cout1_ws(const gs_wgrad_desc* d, const void* a, const void* g, float* dw, float* ws, int64_t ws_floats,
                                 const gs_twin* tw, void* stream) {
  GS_REQUIRE(d && a && g && dw && ws, "gs_wgrad_cout1_ws: null argument");
  GS_REQUIRE(gs_wgrad_cout1_eligible(d), "gs_wgrad_cout1_ws: not the weight gradient of a 4 x 4 one-channel layer");
  const int wpi = cout1_wgs_per_img(d);
  const long long slabs = (long long)d->N * wpi;
  GS_REQUIRE(ws_floats >= slabs * 16 * d->Q, "gs_wgrad_cout1_ws: workspace too small (gs_wgrad_cout1_ws_floats)");
  GS_REQUIRE(!tw || (2 * tw->n_split == d->N && tw->dw_delta % 16 == 0),
             "gs_wgrad_cout1_ws: the two networks take the same number of images, gradient buffers 16 bytes apart");
  Cout1WK k;
  k.a = static_cast<const char*>(a); k.g = static_cast<const char*>(g); k.ws = ws;
  k.N = d->N; k.Ha = d->Ha; k.Wa = d->Wa; k.a_cs = d->a_cs; k.a_co = d->a_co;
  k.Hg = d->Hg; k.Wg = d->Wg; k.Q = d->Q; k.g_cs = d->g_cs; k.g_co = d->g_co;
  cout1_grid(d->dh, d->dw_, d->T, &k.h0, &k.w0);
  k.rows_per_wg = kCout1Rows; k.wgs_per_img = wpi;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(cout1_wgrad_kernel, dim3((unsigned)slabs), dim3(256), 0, st, k);
  GS_CHECK_HIP(hipGetLastError());
  const long long n4 = 16LL * d->Q / 4;
  if (tw) gs_launch_slab_reduce(ws, dw, n4, (int)(slabs / 2), n4, st, 2, slabs / 2 * n4, tw->dw_delta / 16);
  else gs_launch_slab_reduce(ws, dw, n4, (int)slabs, n4, st, 1, 0, 0);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
