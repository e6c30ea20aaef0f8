// SelfAttentionBlock of the self-attention discriminator / V-Net (ganslate/nn/attention.py:12-47, used by
// nn/discriminators/patchgan/selfattention_patchgan3d.py:58,73 and nn/generators/vnet/selfattention_vnet3d.py:97-104):
//
//   q = Wq x + bq  (C/8 channels),  k = Wk x + bk  (C/8),  v = Wv x + bv  (C)        three 1x1x1 convolutions
//   A = softmax_j(q_i . k_j)                                                          N x N over all voxels of the map
//   out_i = gamma * sum_j A_ij v_j + x_i
//
// on NDHWC bf16 activations x [B][N][C] (N = D*H*W <= a few thousand: the block sits on 8^3 .. 16^3 maps). Everything
// GEMM-shaped runs on ONE batched MFMA kernel, bgemm_kernel<AT, BT>: C[m][n] = sum_k A(m,k) B(n,k) with either operand stored
// k-contiguous ("N": rows of [m][k], fragments by ds_read_b128) or k-strided ("T": rows of [k][m], fragments by the LDS
// transpose read ds_read_b64_tr_b16) — that covers the projections and their three gradients, Q K^T, P V and the four products
// of the attention backward without a transposed copy of anything. The N x N logits are materialised (fp32 logits, bf16
// probabilities: 6 bytes per pair, 100 MB per image at N = 4096) with a row softmax between the two GEMMs; the FLOP count
// (2 N^2 (C/8 + C) forward) is a few GFLOP per block, so the launches are bandwidth / latency bound, not MFMA bound.
// Backward: dV = P^T dO, dP = dO V^T, dS = P o (dP - rowsum(P o dP)), dQ = dS K, dK = dS^T Q, then the projections'
// data / weight / bias gradients; dgamma = sum(dout o O) reduced in a fixed order (deterministic).
#include "common.hpp"

// wgrad.hip: dst[e] += slab 0 [e] + slab 1 [e] + ... in slab order
void gs_launch_slab_reduce(const float* ws, float* dst, long long n4, int slabs, long long stride4, hipStream_t st, int nets,
                           long long ws_y4, long long dw_y4);

namespace {
constexpr int GT = 64;                 // output tile per workgroup (GT x GT), 4 waves of 32 x 32
constexpr int GK = 32;                 // K-step
constexpr int NP = GK * 2 + 16;        // "N" tile row pitch (bytes): 32 k + pad
constexpr int TP = GT * 2 + 16;        // "T" tile row pitch: 64 m + pad

struct BGemmK {
  const void* A; const void* B; void* C;
  const float* bias;                   // per column n (nullptr: none)
  const unsigned short* res;           // bf16 residual added to the result, same layout as C (nullptr: none)
  long long sa, sb, sc;                // batch strides (elements)
  int lda, ldb, ldc;                   // row pitches (elements) of the stored matrices
  int M, N, K;
  float alpha;
  int accumulate;                      // fp32 output only: C += result
};

// stage a 64 (rows) x 32 (k) tile of an operand into LDS as bf16, zero outside [rows_valid) x [k_valid)
//   T = false: element (r, k) at src[r * ld + k]   -> LDS [64][32] (pitch NP)
//   T = true : element (r, k) at src[k * ld + r]   -> LDS [32][64] (pitch TP)
//   lo != nullptr (fp32 sources, T = false): the bf16 remainder v - bf16(v) goes to a second tile (hi / lo split operands)
template <bool T, typename S>
__device__ __forceinline__ void stage_tile(char* lds, const S* src, int ld, int r0, int rows_valid, int k0, int k_valid,
                                           char* lo = nullptr) {
  const int tid = threadIdx.x;
  if (!T) {
    // 64 rows x 4 pieces of 8 k
    const int r = tid >> 2, pc = tid & 3;
    unsigned short v[8], w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + pc * 8 + e;
      float f = 0.f;
      if (r0 + r < rows_valid && k < k_valid) {
        const S s = src[(size_t)(r0 + r) * ld + k];
        if constexpr (sizeof(S) == 2) f = bf2f(s); else f = s;
      }
      v[e] = f2bf(f);
      w[e] = f2bf(f - bf2f(v[e]));
    }
    uint4 o;
    o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
    o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
    *reinterpret_cast<uint4*>(lds + r * NP + pc * 16) = o;
    if (lo) {
      o.x = w[0] | ((unsigned)w[1] << 16); o.y = w[2] | ((unsigned)w[3] << 16);
      o.z = w[4] | ((unsigned)w[5] << 16); o.w = w[6] | ((unsigned)w[7] << 16);
      *reinterpret_cast<uint4*>(lo + r * NP + pc * 16) = o;
    }
  } else {
    // 32 k rows x 8 pieces of 8 r
    const int k = tid >> 3, pc = tid & 7;
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = r0 + pc * 8 + e;
      float f = 0.f;
      if (r < rows_valid && k0 + k < k_valid) {
        const S s = src[(size_t)(k0 + k) * ld + r];
        if constexpr (sizeof(S) == 2) f = bf2f(s); else f = s;
      }
      v[e] = f2bf(f);
    }
    uint4 o;
    o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
    o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
    *reinterpret_cast<uint4*>(lds + k * TP + pc * 16) = o;
  }
}

// MFMA operand fragment (16 rows x 32 k) of rows [rb, rb + 16) of a staged tile
template <bool T>
__device__ __forceinline__ bf16x8 tile_frag(const char* lds, int rb, int lane) {
  if (!T) {
    return *reinterpret_cast<const bf16x8*>(lds + (rb + (lane & 15)) * NP + (lane >> 4) * 16);
  } else {
    // [k][r] image: lane group fk covers k rows fk*8 .. fk*8+7; the transpose read hands lane (l & 15) its own column
    const int fk = lane >> 4, frr = (lane & 15) >> 2, fcc = lane & 3;
    const char* ap = lds + (fk * 8 + frr) * TP + (rb + fcc * 4) * 2;
    const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)GS_LDS(ap)));
    const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)GS_LDS(ap + 4 * TP)));
    return __builtin_bit_cast(bf16x8, uint4{lo.x, lo.y, hi.x, hi.y});
  }
}

// C[b][m][n] = alpha * sum_k A(m,k) B(n,k) (+ bias[n]) (+ res[m][n]); A / B element types SA / SB (bf16 = unsigned short, or
// float, converted while staging); OUT_BF16 selects the output type
// SPLIT (fp32 sources, both "N"): hi / lo split operands, three MFMA products a_hi b_hi + a_hi b_lo + a_lo b_hi — the attention
// logits feed a softmax, where a bf16 rounding of q and k (2^-9 of a logit of magnitude 10-30) would move the probabilities by
// percents (the same reason csrc/patchnce.hip splits its logit GEMM)
template <bool AT, bool BT, typename SA, typename SB, bool OUT_BF16, bool SPLIT = false>
__global__ __launch_bounds__(256) void bgemm_kernel(const BGemmK p) {
  constexpr int TILE = GT * NP > GK * TP ? GT * NP : GK * TP;
  __shared__ __attribute__((aligned(16))) char lds[(SPLIT ? 4 : 2) * TILE];
  char* la = lds;
  char* lb = lds + TILE;
  [[maybe_unused]] char* la_lo = lds + 2 * TILE;
  [[maybe_unused]] char* lb_lo = lds + 3 * TILE;
  const int b = blockIdx.z;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const SA* A = static_cast<const SA*>(p.A) + (size_t)b * p.sa;
  const SB* B = static_cast<const SB*>(p.B) + (size_t)b * p.sb;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;                 // 32 x 32 per wave
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < p.K; k0 += GK) {
    __syncthreads();
    stage_tile<AT>(la, A, p.lda, m0, p.M, k0, p.K, SPLIT ? la_lo : nullptr);
    stage_tile<BT>(lb, B, p.ldb, n0, p.N, k0, p.K, SPLIT ? lb_lo : nullptr);
    __syncthreads();
    bf16x8 af[2], bf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[i] = tile_frag<AT>(la, wm * 32 + i * 16, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[j] = tile_frag<BT>(lb, wn * 32 + j * 16, lane);
    if constexpr (SPLIT) {       // small terms first
      bf16x8 al[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) al[i] = tile_frag<AT>(la_lo, wm * 32 + i * 16, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) bl[j] = tile_frag<BT>(lb_lo, wn * 32 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bf[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bl[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
  }
  // D layout: column n = lane & 15, rows m = (lane >> 4) * 4 + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 32 + j * 16 + (lane & 15);
      if (n >= p.N) continue;
      const float bia = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
        if (m >= p.M) continue;
        const size_t o = (size_t)b * p.sc + (size_t)m * p.ldc + n;
        float v = p.alpha * acc[i][j][r] + bia;
        if (p.res) v += bf2f(p.res[o]);
        if constexpr (OUT_BF16) static_cast<unsigned short*>(p.C)[o] = f2bf(v);
        else {
          float* c = static_cast<float*>(p.C);
          c[o] = p.accumulate ? c[o] + v : v;
        }
      }
    }
}

template <bool AT, bool BT, typename SA, typename SB, bool OUT_BF16, bool SPLIT = false>
int bgemm(const BGemmK& k, int batch, hipStream_t st) {
  static_assert(!SPLIT || (!AT && !BT), "split operands: k-contiguous sources only");
  const dim3 grid((k.N + GT - 1) / GT, (k.M + GT - 1) / GT, batch);
  hipLaunchKernelGGL((bgemm_kernel<AT, BT, SA, SB, OUT_BF16, SPLIT>), grid, dim3(256), 0, st, k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  for (int o = 32; o > 0; o >>= 1) {
    const float w = __shfl_xor(v, o, 64);
    v = is_max ? fmaxf(v, w) : v + w;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];     // fixed order
  return r;
}

// one workgroup per row: P = softmax(S) (nn.Softmax(dim=-1), attention.py:24,38), bf16 out
__global__ __launch_bounds__(256) void attn_softmax_kernel(const float* S, unsigned short* P, int N) {
  __shared__ float red[4];
  const size_t row = (size_t)blockIdx.y * N + blockIdx.x;
  const float* s = S + row * N;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < N; j += 256) m = fmaxf(m, s[j]);
  m = block_reduce(m, red, true);
  float l = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) l += __expf(s[j] - m);
  l = block_reduce(l, red, false);
  const float inv = 1.f / l;
  unsigned short* pr = P + row * N;
  for (int j = threadIdx.x; j < N; j += 256) pr[j] = f2bf(__expf(s[j] - m) * inv);
}

// dS = P o (dP - sum_j P o dP) per row, bf16 out (softmax backward)
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const unsigned short* P, const float* dP, unsigned short* dS,
                                                               int N) {
  __shared__ float red[4];
  const size_t row = (size_t)blockIdx.y * N + blockIdx.x;
  const unsigned short* pr = P + row * N;
  const float* dp = dP + row * N;
  float t = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) t += bf2f(pr[j]) * dp[j];
  t = block_reduce(t, red, false);
  unsigned short* o = dS + row * N;
  for (int j = threadIdx.x; j < N; j += 256) o[j] = f2bf(bf2f(pr[j]) * (dp[j] - t));
}

// out = gamma * O + x (attention.py:46), all bf16 [rows][C]; gamma is a device scalar
__global__ __launch_bounds__(256) void attn_residual_kernel(const unsigned short* O, const unsigned short* x,
                                                            const float* gamma, unsigned short* out, long long n) {
  const float g = gamma[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = f2bf(g * bf2f(O[i]) + bf2f(x[i]));
}

// dO = gamma * dout (bf16) and per-workgroup partial sums of dout o O (the gradient of gamma), fixed order
__global__ __launch_bounds__(256) void attn_dout_kernel(const unsigned short* dout, const unsigned short* O, const float* gamma,
                                                        unsigned short* dO, float* part, long long n) {
  __shared__ float red[4];
  const float g = gamma[0];
  float t = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float d = bf2f(dout[i]);
    t += d * bf2f(O[i]);
    dO[i] = f2bf(g * d);
  }
  t = block_reduce(t, red, false);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ void attn_gamma_grad_kernel(const float* part, int n, float* dgamma) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < n; ++i) t += part[i];
    dgamma[0] += t;
  }
}

// column sums over rows of a bf16 [rows][ld] matrix, columns [c0, c0 + n): db[c] += sum (bias gradients), fixed order
__global__ __launch_bounds__(256) void attn_colsum_kernel(const unsigned short* m, long long rows, int ld, int c0, int n, float* db) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  float t = 0.f;
  for (long long r = threadIdx.x; r < rows; r += 256) t += bf2f(m[(size_t)r * ld + c0 + c]);
  red[threadIdx.x] = t;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && c < n) db[c] += red[0];
}

// zero the pad columns behind the q and the k block of dqkv [rows][ct] (widths that are not a multiple of 8)
__global__ __launch_bounds__(256) void attn_zero_cols_kernel(unsigned short* m, long long rows, int ld, int c0, int n, int block) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r < rows)
    for (int b = 0; b < 2; ++b)
      for (int c = 0; c < n; ++c) m[(size_t)r * ld + b * block + c0 + c] = 0;
}

constexpr int DOUT_BLOCKS = 512;
int rup(int v, int m) { return (v + m - 1) / m * m; }
struct Plan {
  int dp, ct;                           // padded q / k width, channels of the stacked projection [q | k | v]
  long long rows;                       // B * N
  size_t qk, qkv, O, P, S, fwd_end, dqkv, dO, dS, part, wcat, dws, end; // byte offsets into `work`
  int splits;                           // K splits of the projections' weight gradients (rows = splits * kc)
  long long kc;
};
Plan plan(const gs_attn_desc* d) {
  Plan p;
  p.dp = rup(d->C / 8, 8);
  p.ct = 2 * p.dp + d->C;
  p.rows = (long long)d->B * d->N;
  const size_t nn = (size_t)d->B * d->N * d->N;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
  p.qk = take((size_t)p.rows * 2 * p.dp * 4);    // q | k in fp32 (the logit GEMM splits them into bf16 hi + lo)
  p.qkv = take((size_t)p.rows * p.ct * 2);       // bf16 [q | k | v]: only the v columns are used
  p.O = take((size_t)p.rows * d->C * 2);
  p.P = take(nn * 2);
  p.S = take(nn * 4);                   // logits (forward), dP (backward)
  p.fwd_end = o;                        // (a pass that will never run backward needs nothing behind this point)
  p.dqkv = take((size_t)p.rows * p.ct * 2);
  p.dO = take((size_t)p.rows * d->C * 2);
  p.dS = take(nn * 2);
  p.part = take(DOUT_BLOCKS * 4);
  p.wcat = take((size_t)p.ct * d->C * 4);          // [Wq; Wk; Wv] stacked along k for the ONE launch that forms dx
  // weight gradients: K = all voxels of the batch over a handful of output tiles — split over up to 32 ranges whose partial
  // products go to slabs, added in slab order afterwards (deterministic)
  p.splits = 1;
  for (int sp = 32; sp > 1; sp >>= 1)
    if (p.rows % sp == 0 && p.rows / sp >= 128) { p.splits = sp; break; }
  p.kc = p.rows / p.splits;
  p.dws = take(p.splits > 1 ? (size_t)p.splits * d->C * d->C * 4 : 0);
  p.end = o;
  return p;
}
int check(const gs_attn_desc* d) {
  GS_REQUIRE(d && d->B >= 1 && d->N >= 1 && d->C >= 8 && d->C % 8 == 0, "gs_attn: B, N >= 1 and C a multiple of 8");
  GS_REQUIRE((long long)d->N * d->N * d->B < (1LL << 31), "gs_attn: B * N * N must stay below 2^31");
  return 0;
}
}  // namespace

extern "C" int64_t gs_attn_work_bytes(const gs_attn_desc* d) {
  if (!d || d->B < 1 || d->N < 1 || d->C < 8) return -1;
  return (int64_t)plan(d).end;
}
// the same for a forward pass whose state is never handed to gs_attn_backward (inference): q / k / v, logits, probabilities and
// the attention output only — without the backward's buffers (at N = 4096: 134 MB less per block)
extern "C" int64_t gs_attn_forward_work_bytes(const gs_attn_desc* d) {
  if (!d || d->B < 1 || d->N < 1 || d->C < 8) return -1;
  return (int64_t)plan(d).fwd_end;
}

extern "C" int gs_attn_forward(const gs_attn_desc* d, const void* x, const gs_attn_params* w, void* out, void* work,
                               void* stream) {
  if (int rc = check(d)) return rc;
  GS_REQUIRE(x && w && out && work && w->gamma && w->wq && w->wk && w->wv, "gs_attn_forward: null argument");
  const Plan p = plan(d);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* wk = static_cast<char*>(work);
  unsigned short* qkv = reinterpret_cast<unsigned short*>(wk + p.qkv);
  unsigned short* O = reinterpret_cast<unsigned short*>(wk + p.O);
  unsigned short* P = reinterpret_cast<unsigned short*>(wk + p.P);
  float* S = reinterpret_cast<float*>(wk + p.S);
  const int dq = d->C / 8, C = d->C, N = d->N;
  // projections: [q | k | v] = x W^T + b, rows = all voxels of the batch (pad columns of q / k stay untouched: never read)
  BGemmK g{};
  g.A = x; g.lda = C; g.sa = 0; g.K = C; g.M = (int)p.rows; g.alpha = 1.f; g.ldc = p.ct; g.sc = 0; g.ldb = C; g.sb = 0;
  const float* Ws[3] = {w->wq, w->wk, w->wv};
  const float* bs[3] = {w->bq, w->bk, w->bv};
  const int cols[3] = {dq, dq, C}, c0[3] = {0, p.dp, 2 * p.dp};
  float* qk = reinterpret_cast<float*>(wk + p.qk);
  for (int i = 0; i < 3; ++i) {
    g.B = Ws[i]; g.bias = bs[i]; g.N = cols[i];
    if (i < 2) {       // q, k: fp32
      g.C = qk + i * p.dp; g.ldc = 2 * p.dp;
      // (split weights: the logits are as accurate as fp32 q / k make them — x itself is exact in bf16)
      if (int rc = bgemm<false, false, unsigned short, float, false, true>(g, 1, st)) return rc;
    } else {
      g.C = qkv + c0[i]; g.ldc = p.ct;
      if (int rc = bgemm<false, false, unsigned short, float, true>(g, 1, st)) return rc;
    }
  }
  // S = q k^T per image (hi / lo split operands)
  BGemmK s{};
  s.A = qk; s.lda = 2 * p.dp; s.sa = (long long)N * 2 * p.dp; s.B = qk + p.dp; s.ldb = 2 * p.dp; s.sb = s.sa; s.C = S; s.ldc = N;
  s.sc = (long long)N * N; s.M = N; s.N = N; s.K = dq; s.alpha = 1.f;
  if (int rc = bgemm<false, false, float, float, false, true>(s, d->B, st)) return rc;
  hipLaunchKernelGGL(attn_softmax_kernel, dim3(N, d->B), dim3(256), 0, st, S, P, N);
  // O = P v  (v is [key][C]: k-strided operand)
  BGemmK o{};
  o.A = P; o.lda = N; o.sa = (long long)N * N; o.B = qkv + 2 * p.dp; o.ldb = p.ct; o.sb = (long long)N * p.ct; o.C = O; o.ldc = C;
  o.sc = (long long)N * C; o.M = N; o.N = C; o.K = N; o.alpha = 1.f;
  if (int rc = bgemm<false, true, unsigned short, unsigned short, true>(o, d->B, st)) return rc;
  const long long n = p.rows * C;
  hipLaunchKernelGGL(attn_residual_kernel, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, st,
                     O, static_cast<const unsigned short*>(x), w->gamma, static_cast<unsigned short*>(out), n);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_attn_backward(const gs_attn_desc* d, const void* x, const void* dout, const gs_attn_params* w,
                                const gs_attn_params* gw, void* work, void* dx, void* stream) {
  if (int rc = check(d)) return rc;
  GS_REQUIRE(x && dout && w && work && dx, "gs_attn_backward: null argument");
  const Plan p = plan(d);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* wk = static_cast<char*>(work);
  unsigned short* qkv = reinterpret_cast<unsigned short*>(wk + p.qkv);
  unsigned short* O = reinterpret_cast<unsigned short*>(wk + p.O);
  unsigned short* P = reinterpret_cast<unsigned short*>(wk + p.P);
  float* dP = reinterpret_cast<float*>(wk + p.S);
  unsigned short* dqkv = reinterpret_cast<unsigned short*>(wk + p.dqkv);
  unsigned short* dO = reinterpret_cast<unsigned short*>(wk + p.dO);
  unsigned short* dS = reinterpret_cast<unsigned short*>(wk + p.dS);
  float* part = reinterpret_cast<float*>(wk + p.part);
  const int dq = d->C / 8, C = d->C, N = d->N;
  const long long n = p.rows * C;
  // dO = gamma dout; dgamma += sum(dout o O)
  const int blocks = (int)((n + 255) / 256 > DOUT_BLOCKS ? DOUT_BLOCKS : (n + 255) / 256);
  hipLaunchKernelGGL(attn_dout_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const unsigned short*>(dout), O, w->gamma, dO,
                     part, n);
  if (gw && gw->gamma) hipLaunchKernelGGL(attn_gamma_grad_kernel, dim3(1), dim3(64), 0, st, part, blocks, gw->gamma);
  const long long sNN = (long long)N * N, sq = (long long)N * p.ct, sO = (long long)N * C;
  // dv = P^T dO   (both operands k-strided: k = query)
  BGemmK a{};
  a.A = P; a.lda = N; a.sa = sNN; a.B = dO; a.ldb = C; a.sb = sO; a.C = dqkv + 2 * p.dp; a.ldc = p.ct; a.sc = sq;
  a.M = N; a.N = C; a.K = N; a.alpha = 1.f;
  if (int rc = bgemm<true, true, unsigned short, unsigned short, true>(a, d->B, st)) return rc;
  // dP = dO v^T
  BGemmK b{};
  b.A = dO; b.lda = C; b.sa = sO; b.B = qkv + 2 * p.dp; b.ldb = p.ct; b.sb = sq; b.C = dP; b.ldc = N; b.sc = sNN;
  b.M = N; b.N = N; b.K = C; b.alpha = 1.f;
  if (int rc = bgemm<false, false, unsigned short, unsigned short, false>(b, d->B, st)) return rc;
  hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3(N, d->B), dim3(256), 0, st, P, dP, dS, N);
  // dq = dS k   (k is [key][dq]: k-strided)
  BGemmK c{};
  const float* qk = reinterpret_cast<const float*>(wk + p.qk);
  const long long sqk = (long long)N * 2 * p.dp;
  c.A = dS; c.lda = N; c.sa = sNN; c.B = qk + p.dp; c.ldb = 2 * p.dp; c.sb = sqk; c.C = dqkv; c.ldc = p.ct; c.sc = sq;
  c.M = N; c.N = dq; c.K = N; c.alpha = 1.f;
  if (int rc = bgemm<false, true, unsigned short, float, true>(c, d->B, st)) return rc;
  // dk = dS^T q
  BGemmK e{};
  e.A = dS; e.lda = N; e.sa = sNN; e.B = qk; e.ldb = 2 * p.dp; e.sb = sqk; e.C = dqkv + p.dp; e.ldc = p.ct; e.sc = sq;
  e.M = N; e.N = dq; e.K = N; e.alpha = 1.f;
  if (int rc = bgemm<true, true, unsigned short, float, true>(e, d->B, st)) return rc;
  // projections: dx = dout + [dq | dk | dv] [Wq; Wk; Wv] as ONE launch over the stacked k axis: one fp32 accumulation, one
  // rounding (three launches, each taking the previous bf16 result as its residual, rounded three times). The stacked fp32
  // weights are built in the scratch (pad rows between the blocks zero, like the pad columns of dqkv).
  const int cols[3] = {dq, dq, C}, c0[3] = {0, p.dp, 2 * p.dp};
  {
    const float* Ws[3] = {w->wq, w->wk, w->wv};
    float* wcat = reinterpret_cast<float*>(wk + p.wcat);
    if (p.dp != dq) {
      GS_CHECK_HIP(hipMemsetAsync(wcat, 0, (size_t)p.ct * C * 4, st));
      // (dq / dk are written column by column range: their pad columns would multiply the zero rows with whatever the scratch held)
      hipLaunchKernelGGL(attn_zero_cols_kernel, dim3((unsigned)((p.rows + 255) / 256)), dim3(256), 0, st, dqkv, p.rows, p.ct, dq,
                         p.dp - dq, p.dp);
    }
    for (int i = 0; i < 3; ++i)
      GS_CHECK_HIP(hipMemcpyAsync(wcat + (size_t)c0[i] * C, Ws[i], (size_t)cols[i] * C * 4, hipMemcpyDeviceToDevice, st));
    BGemmK f{};
    f.A = dqkv; f.lda = p.ct; f.B = wcat; f.ldb = C; f.C = dx; f.ldc = C; f.M = (int)p.rows; f.N = C; f.K = p.ct;
    f.alpha = 1.f; f.res = static_cast<const unsigned short*>(dout);
    if (int rc = bgemm<false, true, unsigned short, float, true>(f, 1, st)) return rc;
  }
  if (gw) {
    float* dWs[3] = {gw->wq, gw->wk, gw->wv};
    float* dbs[3] = {gw->bq, gw->bk, gw->bv};
    float* slabs = reinterpret_cast<float*>(wk + p.dws);
    for (int i = 0; i < 3; ++i) {
      if (dWs[i]) {       // dW[o][c] += sum_rows dproj[row][o] x[row][c]
        BGemmK h{};
        h.A = dqkv + c0[i]; h.lda = p.ct; h.B = x; h.ldb = C; h.ldc = C; h.M = cols[i]; h.N = C; h.alpha = 1.f;
        if (p.splits > 1) {       // K ranges as the batch axis: slab sp = the partial product over voxels [sp * kc, (sp + 1) * kc)
          h.K = (int)p.kc; h.sa = p.kc * p.ct; h.sb = p.kc * C; h.C = slabs; h.sc = (long long)cols[i] * C; h.accumulate = 0;
          if (int rc = bgemm<true, true, unsigned short, unsigned short, false>(h, p.splits, st)) return rc;
          gs_launch_slab_reduce(slabs, dWs[i], (long long)cols[i] * C / 4, p.splits, (long long)cols[i] * C / 4, st, 1, 0, 0);
        } else {
          h.K = (int)p.rows; h.C = dWs[i]; h.accumulate = 1;
          if (int rc = bgemm<true, true, unsigned short, unsigned short, false>(h, 1, st)) return rc;
        }
      }
      if (dbs[i])
        hipLaunchKernelGGL(attn_colsum_kernel, dim3(cols[i]), dim3(256), 0, st, dqkv, p.rows, p.ct, c0[i], cols[i], dbs[i]);
    }
  }
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
