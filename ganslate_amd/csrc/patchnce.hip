// PatchNCE (CUT's contrastive loss) and its patch MLP on the gfx950 matrix cores.
//
// Reference: ganslate/nn/gans/unpaired/cut.py:229-294 (FeaturePatchMLP: per feature level Linear(C, nc) - ReLU -
// Linear(nc, nc) on the sampled patches, then x / (||x||_2 + 1e-7)) and ganslate/nn/losses/cut_losses.py:14-43
// (PatchNCELoss: logits [q.k+ , q.k_j for the other patches j of the SAME image, the diagonal replaced by -10] / T,
// cross-entropy against class 0, feat_k detached). Round 1 ran this as library GEMMs + a few dozen elementwise launches
// per level; here all levels of one NCE term are ONE launch per stage:
//
//   nce_mlp_fwd_kernel   (row tile, level, side q|k)  H = relu(X W1^T + b1), F = H W2^T + b2, n = ||F||, Fhat = F / (n + eps)
//   nce_loss_kernel      (row tile, image, level)     S = Qhat Khat_b^T, logits, loss partials, G = dL/dS (dpos on the
//                                                     diagonal), dQhat = G Khat_b, norm backward -> dF
//   nce_mlp_bwd_kernel   (row tile, level)            dH = (dF W2) * relu'(H), dX = dH W1          (x upstream scale)
//   nce_param_grad_kernel(64-row tile of the output, which, level)   dW2 += dF^T H, db2 += sum dF, dW1 += dH^T X, db1 += sum dH
//
// Every GEMM is a 64 x 256 output tile per workgroup (8 waves, wave w owns columns 32w .. 32w+31, v_mfma_f32_16x16x32_bf16,
// fp32 accumulate) with both operands staged in LDS as bf16 [rows][k]; operands that are stored the other way round in
// memory (W^T, Khat^T, dF^T, H^T, X^T) are transposed while they are staged. The tensors are small (2048 rows x 256):
// what matters here is the launch count and that nothing leaves the device; row reductions (norms, soft-max) go through
// DPP row sums + a small LDS exchange, the loss is summed from per-workgroup partials in a fixed order.
#include "common.hpp"

namespace {
constexpr int NCE_MAX_LEVELS = 8;
constexpr int NCE_RSPLIT = 8;             // most row ranges of the parameter-gradient launch
constexpr int TM = 64;                    // rows per tile
constexpr int TN = 256;                   // columns per tile (= nc, = patches per image)
constexpr int AP = (TN + 8) * 2;          // pitch (bytes) of a [64][256] bf16 operand in LDS
constexpr int BP = (64 + 8) * 2;          // pitch of a [256][64] bf16 K-chunk

struct NceLevel {
  const float* xq;      // [R][C] sampled target patches
  const float* xk;      // [R][C] sampled source patches
  float* dxq;           // [R][C] gradient w.r.t. xq (backward)
  int C;                // channels of this level
  long long w_off;      // offset of this level's parameters in the flat buffer: W1 [nc][C], b1 [nc], W2 [nc][nc], b2 [nc]
  long long row_off;    // first row of this level in the [L][R] x nc work buffers
};
struct NceK {
  NceLevel lv[NCE_MAX_LEVELS];
  const float* params;  // flat fp32 master
  float* grads;         // flat fp32 gradient (same layout)
  float* fhat;          // [2][L*R][nc] fp32: normalised features, q then k
  float* norms;         // [L*R] ||F|| of the q side
  unsigned short* h;    // [L*R][nc] bf16 hidden activations of the q side (post ReLU)
  unsigned short* df;   // [L*R][nc] bf16 dL/dF (unscaled), later dH (scaled) is written to dh
  unsigned short* dh;   // [L*R][nc] bf16
  float* loss_part;     // [L][B][P/64] partial sums of the per-row losses
  const float* gscale;  // device scalar: upstream gradient of the summed loss
  float* wpart;         // parameter-gradient partial sums [L][2][RS][nc * TN + nc] (row splits > 1)
  int rsplit;           // row splits of the parameter-gradient launch
  int L, R, B, P, nc;
  float inv_T, coef;    // 1 / nce_T ; lambda_nce / (L * R): weight of one row's loss in the returned scalar
  long long side_stride;  // L*R*nc
};

struct Acc { f32x4 a[4][2]; };

// stage ROWS x 64 (k) elements as bf16 into dst[row][k] (pitch P bytes):
//   TRANS = false: src[(r0 + row) * ld + k0 + k]      rows along the slow axis of src
//   TRANS = true : src[(k0 + k) * ld + r0 + row]      the operand lives transposed in memory
// elements with row >= rows_valid or k >= k_valid are zero
template <int ROWS, int P, bool TRANS, typename T>
__device__ __forceinline__ void stage(char* dst, const T* src, long long ld, int r0, int rows_valid, int k0, int k_valid,
                                      float scale = 1.f) {
  // up to sixteen loads in flight per thread, then their LDS stores: written as one load - convert - store per element the
  // compiler waited for every load before issuing the next (40 serialised L2 round trips per 64-row chunk; the parameter-
  // gradient kernel, which walks 32 such chunks on 40 workgroups, took 489 us for 2.7 GFLOP)
  constexpr int PER = ROWS * 64 / 512, UN = PER < 16 ? PER : 16;
  static_assert(PER % UN == 0, "stage: rows per tile");
#pragma unroll 1
  for (int b = 0; b < PER; b += UN) {
    float v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int e = threadIdx.x + (b + u) * 512;
      int row, k;
      if (TRANS) { row = e % ROWS; k = e / ROWS; } else { row = e >> 6; k = e & 63; }
      v[u] = 0.f;
      if (r0 + row < rows_valid && k0 + k < k_valid) {
        const T s = TRANS ? src[(long long)(k0 + k) * ld + r0 + row] : src[(long long)(r0 + row) * ld + k0 + k];
        if constexpr (sizeof(T) == 2) v[u] = bf2f(s); else v[u] = s;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int e = threadIdx.x + (b + u) * 512;
      int row, k;
      if (TRANS) { row = e % ROWS; k = e / ROWS; } else { row = e >> 6; k = e & 63; }
      *reinterpret_cast<unsigned short*>(dst + row * P + k * 2) = f2bf(v[u] * scale);
    }
  }
}

// the same with every value split into bf16 hi + bf16 lo (hi + lo carries ~16 mantissa bits): the logits are divided by
// T = 0.07, so plain bf16 operands would put a 14x amplified rounding error into the soft-max
template <int ROWS, int P>
__device__ __forceinline__ void stage_split(char* dhi, char* dlo, const float* src, long long ld, int r0, int rows_valid,
                                            int k0, int k_valid) {
  for (int e = threadIdx.x; e < ROWS * 64; e += 512) {
    const int row = e >> 6, k = e & 63;
    float v = 0.f;
    if (r0 + row < rows_valid && k0 + k < k_valid) v = src[(long long)(r0 + row) * ld + k0 + k];
    const unsigned short hi = f2bf(v);
    *reinterpret_cast<unsigned short*>(dhi + row * P + k * 2) = hi;
    *reinterpret_cast<unsigned short*>(dlo + row * P + k * 2) = f2bf(v - bf2f(hi));
  }
}

// acc += A[64][k0 .. k0+63] * Bc[256][64]^T for this wave's 64 x 32 block (A with pitch AP or BP given by APITCH)
template <int APITCH>
__device__ __forceinline__ void mma_chunk(Acc& c, const char* As, int a_k0, const char* Bc, int wave, int lane) {
  const int fr = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    bf16x8 af[4], bf[2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      af[i] = *reinterpret_cast<const bf16x8*>(As + (i * 16 + fr) * APITCH + (a_k0 + kk * 32 + kg * 8) * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      bf[j] = *reinterpret_cast<const bf16x8*>(Bc + (wave * 32 + j * 16 + fr) * BP + (kk * 32 + kg * 8) * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) c.a[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], c.a[i][j], 0, 0, 0);
  }
}
__device__ __forceinline__ void zero(Acc& c) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) c.a[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// element (i, j, r) of a wave's accumulators: row = 16 i + 4 (lane >> 4) + r, column = 32 wave + 16 j + (lane & 15)
#define NCE_ROW(i, r) ((i) * 16 + 4 * (lane >> 4) + (r))
#define NCE_COL(j) (wave * 32 + (j) * 16 + (lane & 15))

// sum over the 256 columns of f(i, r) per row: DPP over the 16 columns of a lane row, then the 2 column blocks, then the
// 8 waves through red[8][64]; result for row `row` in out[row]
__device__ __forceinline__ void row_reduce_sum(const float (&v)[4][4], float* red, float* out, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = row16_sum(v[i][r]);
      if ((lane & 15) == 0) red[wave * 64 + NCE_ROW(i, r)] = s;
    }
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += red[w * 64 + threadIdx.x];
    out[threadIdx.x] = t;
  }
  __syncthreads();
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}
}  // namespace

// ---- patch MLP forward ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void nce_mlp_fwd_kernel(const NceK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                          // X tile, later H tile: [64][256] bf16
  char* Bc = smem + TM * AP;                // [256][64] bf16 K-chunk of W1 / W2
  float* red = reinterpret_cast<float*>(Bc + TN * BP);   // [8][64]
  float* rowv = red + 8 * 64;               // [64]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int level = blockIdx.y, side = blockIdx.z;
  const NceLevel& lv = p.lv[level];
  const int r0 = blockIdx.x * TM;
  const float* x = side ? lv.xk : lv.xq;
  const float* W1 = p.params + lv.w_off;
  const float* b1 = W1 + (long long)p.nc * lv.C;
  const float* W2 = b1 + p.nc;
  const float* b2 = W2 + (long long)p.nc * p.nc;
  const int Cp = (lv.C + 63) / 64 * 64;
  Acc acc;
  zero(acc);
  // ---- H = relu(X W1^T + b1)
  for (int k0 = 0; k0 < Cp; k0 += 64) {
    __syncthreads();
    stage<TM, AP, false>(As + k0 * 2, x, lv.C, r0, p.R, k0, lv.C);
    stage<TN, BP, false>(Bc, W1, lv.C, 0, p.nc, k0, lv.C);
    __syncthreads();
    mma_chunk<AP>(acc, As, k0, Bc, wave, lane);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float bb = b1[NCE_COL(j)];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = fmaxf(acc.a[i][j][r] + bb, 0.f);
        const unsigned short hb = f2bf(hv);
        *reinterpret_cast<unsigned short*>(As + NCE_ROW(i, r) * AP + NCE_COL(j) * 2) = hb;
        if (side == 0 && r0 + NCE_ROW(i, r) < p.R)
          p.h[(lv.row_off + r0 + NCE_ROW(i, r)) * p.nc + NCE_COL(j)] = hb;
      }
    }
  // ---- F = H W2^T + b2
  zero(acc);
  for (int k0 = 0; k0 < p.nc; k0 += 64) {
    __syncthreads();
    stage<TN, BP, false>(Bc, W2, p.nc, 0, p.nc, k0, p.nc);
    __syncthreads();
    mma_chunk<AP>(acc, As, k0, Bc, wave, lane);
  }
  float sq[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) sq[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float bb = b2[NCE_COL(j)];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc.a[i][j][r] += bb;
        sq[i][r] += acc.a[i][j][r] * acc.a[i][j][r];
      }
    }
  __syncthreads();
  row_reduce_sum(sq, red, rowv, wave, lane);
  float* fh = p.fhat + side * p.side_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = NCE_ROW(i, r);
      if (r0 + row >= p.R) continue;
      const float n = sqrtf(rowv[row]);
      const float inv = 1.0f / (n + 1e-7f);
      if (side == 0 && (lane & 15) == 0 && wave == 0) p.norms[lv.row_off + r0 + row] = n;
#pragma unroll
      for (int j = 0; j < 2; ++j) fh[(lv.row_off + r0 + row) * p.nc + NCE_COL(j)] = acc.a[i][j][r] * inv;
    }
}

// ---- logits, loss, gradient w.r.t. F of the q side -----------------------------------------------------------------------
__global__ __launch_bounds__(512) void nce_loss_kernel(const NceK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                          // Qhat tile (bf16 hi), later G: [64][256] bf16
  char* Bc = smem + TM * AP;
  float* red = reinterpret_cast<float*>(Bc + TN * BP);
  float* rowv = red + 8 * 64;               // [64]
  float* rowm = rowv + 64;                  // [64]
  float* posv = rowm + 64;                  // [64]
  char* Al = reinterpret_cast<char*>(posv + 64);     // bf16 lo parts of the Qhat tile / the Khat chunk
  char* Bl = Al + TM * AP;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x, b = blockIdx.y, level = blockIdx.z;
  const NceLevel& lv = p.lv[level];
  const long long base = lv.row_off + (long long)b * p.P;       // first row of image b at this level
  const int q0 = tile * TM;                                     // first patch of this tile inside the image
  const float* Q = p.fhat + base * p.nc;
  const float* K = p.fhat + p.side_stride + base * p.nc;
  // ---- S = Qhat_tile Khat_b^T  (K = nc features, columns = the P patches of the image)
  Acc acc;
  zero(acc);
  for (int k0 = 0; k0 < p.nc; k0 += 64) {
    __syncthreads();
    stage_split<TM, AP>(As + k0 * 2, Al + k0 * 2, Q, p.nc, q0, p.P, k0, p.nc);
    stage_split<TN, BP>(Bc, Bl, K, p.nc, 0, p.P, k0, p.nc);
    __syncthreads();
    mma_chunk<AP>(acc, As, k0, Bc, wave, lane);      // hi * hi
    mma_chunk<AP>(acc, As, k0, Bl, wave, lane);      // hi * lo
    mma_chunk<AP>(acc, Al, k0, Bc, wave, lane);      // lo * hi   (lo * lo is below fp32 rounding of the sum)
  }
  // positive logit = the diagonal S[i][q0 + i] (cut_losses.py:20-21); negatives = the row with the diagonal at -10
  float mx[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[i][r] = -1e30f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = NCE_ROW(i, r), col = NCE_COL(j);
        float v = acc.a[i][j][r];
        if (col == q0 + row) { posv[row] = v; v = -10.0f; }
        if (col >= p.P) v = -1e30f;
        v *= p.inv_T;
        acc.a[i][j][r] = v;
        mx[i][r] = fmaxf(mx[i][r], v);
      }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m = row16_max(mx[i][r]);
      if ((lane & 15) == 0) red[wave * 64 + NCE_ROW(i, r)] = m;
    }
  __syncthreads();
  if (threadIdx.x < 64) {
    float m = posv[threadIdx.x] * p.inv_T;
#pragma unroll
    for (int w = 0; w < 8; ++w) m = fmaxf(m, red[w * 64 + threadIdx.x]);
    rowm[threadIdx.x] = m;
  }
  __syncthreads();
  float ex[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) ex[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(acc.a[i][j][r] - rowm[NCE_ROW(i, r)]);
        acc.a[i][j][r] = e;
        ex[i][r] += e;
      }
  row_reduce_sum(ex, red, rowv, wave, lane);         // rowv = sum over the negatives of exp(l - max)
  // per-row loss = log(sum_all) + max - pos/T ; gradient of the row's logits = softmax - onehot(0)
  if (threadIdx.x < 64) {
    const int row = threadIdx.x;
    const float lp = posv[row] * p.inv_T;
    const float tot = rowv[row] + __expf(lp - rowm[row]);
    const bool valid = q0 + row < p.P;
    const float loss = valid ? __logf(tot) + rowm[row] - lp : 0.f;
    rowv[row] = 1.0f / tot;                                             // normaliser of the soft-max
    posv[row] = (__expf(lp - rowm[row]) / tot - 1.0f);                  // dL/d(pos logit)
    red[row] = loss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int r = 0; r < 64; ++r) s += red[r];
    p.loss_part[((long long)level * p.B + b) * gridDim.x + tile] = s;
  }
  // G[i][j] = dL/dS_ij = p_ij / T (negatives), dpos / T on the diagonal (the masked diagonal logit is a constant)
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = NCE_ROW(i, r), col = NCE_COL(j);
        float g = acc.a[i][j][r] * rowv[row];
        if (col == q0 + row) g = posv[row];
        if (col >= p.P || q0 + row >= p.P) g = 0.f;
        *reinterpret_cast<unsigned short*>(As + row * AP + col * 2) = f2bf(g * p.inv_T * p.coef);
      }
  // ---- dQhat = G Khat_b   (contraction over the patches: Khat_b enters transposed, [feature][patch])
  zero(acc);
  for (int k0 = 0; k0 < p.P; k0 += 64) {
    __syncthreads();
    stage<TN, BP, true>(Bc, K, p.nc, 0, p.nc, k0, p.P);
    __syncthreads();
    mma_chunk<AP>(acc, As, k0, Bc, wave, lane);
  }
  // ---- through x / (||x|| + eps): dF = (dQhat - Qhat (Qhat . dQhat) (n + eps) / n) / (n + eps)
  float dot[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) dot[i][r] = 0.f;
  float qv[4][2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = NCE_ROW(i, r);
        qv[i][j][r] = q0 + row < p.P ? Q[(long long)(q0 + row) * p.nc + NCE_COL(j)] : 0.f;
        dot[i][r] += qv[i][j][r] * acc.a[i][j][r];
      }
  __syncthreads();
  row_reduce_sum(dot, red, rowv, wave, lane);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = NCE_ROW(i, r);
      if (q0 + row >= p.P) continue;
      const float n = p.norms[base + q0 + row];
      const float ne = n + 1e-7f;
      const float k1 = rowv[row] * ne / fmaxf(n, 1e-30f);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        p.df[(base + q0 + row) * p.nc + NCE_COL(j)] = f2bf((acc.a[i][j][r] - qv[i][j][r] * k1) / ne);
    }
}

// ---- patch MLP backward (data) ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void nce_mlp_bwd_kernel(const NceK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                          // dF tile, later dH tile
  char* Bc = smem + TM * AP;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int level = blockIdx.y;
  const NceLevel& lv = p.lv[level];
  const int r0 = blockIdx.x * TM;
  const float* W1 = p.params + lv.w_off;
  const float* W2 = W1 + (long long)p.nc * lv.C + p.nc;
  const float gs = p.gscale ? p.gscale[0] : 1.0f;
  const unsigned short* dF = p.df + lv.row_off * p.nc;
  // ---- dH = (dF W2) * relu'(H): contraction over W2's OUTPUT index -> W2 enters transposed ([in][out])
  Acc acc;
  zero(acc);
  for (int k0 = 0; k0 < p.nc; k0 += 64) {
    __syncthreads();
    stage<TM, AP, false>(As + k0 * 2, dF, p.nc, r0, p.R, k0, p.nc, gs);
    stage<TN, BP, true>(Bc, W2, p.nc, 0, p.nc, k0, p.nc);
    __syncthreads();
    mma_chunk<AP>(acc, As, k0, Bc, wave, lane);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = NCE_ROW(i, r), col = NCE_COL(j);
        float v = 0.f;
        if (r0 + row < p.R) {
          const long long e = (lv.row_off + r0 + row) * p.nc + col;
          v = bf2f(p.h[e]) > 0.f ? acc.a[i][j][r] : 0.f;
          p.dh[e] = f2bf(v);
        }
        *reinterpret_cast<unsigned short*>(As + row * AP + col * 2) = f2bf(v);
      }
  // ---- dX = dH W1: contraction over W1's output index -> W1 enters transposed ([C][nc]); columns = channels (<= 256)
  zero(acc);
  for (int k0 = 0; k0 < p.nc; k0 += 64) {
    __syncthreads();
    stage<TN, BP, true>(Bc, W1, lv.C, 0, lv.C, k0, p.nc);
    __syncthreads();
    mma_chunk<AP>(acc, As, k0, Bc, wave, lane);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = NCE_ROW(i, r), col = NCE_COL(j);
        if (r0 + row < p.R && col < lv.C) lv.dxq[(long long)(r0 + row) * lv.C + col] = acc.a[i][j][r];
      }
}

// ---- parameter gradients: out[m][n] += sum_r P1[r][m] P2[r][n],  bias[m] += sum_r P1[r][m] ----------------------------------
// which = 0: (dF x gscale, H) -> dW2, db2 ; which = 1: (dH, X) -> dW1, db1. A workgroup owns 64 output rows and one of
// `rsplit` contiguous ranges of the rows r (walked in order). With one range it adds into the gradient itself; with more
// (one range = 32 chunks of 64 rows on 40 workgroups took 489 us for 2.7 GFLOP) it leaves its partial sums in wpart and
// nce_param_reduce_kernel adds the ranges in order: the sums stay order-fixed.
__global__ __launch_bounds__(512) void nce_param_grad_kernel(const NceK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                          // P1^T chunk: [64 m][64 r] (pitch BP)
  char* Bc = smem + TM * BP;                // P2^T chunk: [256 n][64 r]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int mtiles = p.nc / TM;
  const int m0 = (blockIdx.x % mtiles) * TM, split = blockIdx.x / mtiles, which = blockIdx.y, level = blockIdx.z;
  const NceLevel& lv = p.lv[level];
  const float gs = p.gscale ? p.gscale[0] : 1.0f;
  const unsigned short* P1 = (which == 0 ? p.df : p.dh) + lv.row_off * p.nc;
  const int N = which == 0 ? p.nc : lv.C;
  float* gW1 = p.grads + lv.w_off;
  float* gb1 = gW1 + (long long)p.nc * lv.C;
  float* gW2 = gb1 + p.nc;
  float* gb2 = gW2 + (long long)p.nc * p.nc;
  float* out = which == 0 ? gW2 : gW1;
  float* bias = which == 0 ? gb2 : gb1;
  int ldo = N;
  if (p.rsplit > 1) {
    out = p.wpart + ((long long)(level * 2 + which) * p.rsplit + split) * ((long long)p.nc * TN + p.nc);
    bias = out + (long long)p.nc * TN;
    ldo = TN;
  }
  const int chunks = (p.R + 63) / 64, per = (chunks + p.rsplit - 1) / p.rsplit;
  const int rbeg = split * per * 64, rend = min(p.R, (split + 1) * per * 64);
  Acc acc;
  zero(acc);
  float bacc = 0.f;
  for (int r0 = rbeg; r0 < rend; r0 += 64) {
    __syncthreads();
    stage<TM, BP, true>(As, P1, p.nc, m0, p.nc, r0, rend, which == 0 ? gs : 1.0f);
    if (which == 0) stage<TN, BP, true>(Bc, p.h + lv.row_off * p.nc, p.nc, 0, p.nc, r0, rend);
    else stage<TN, BP, true>(Bc, lv.xq, lv.C, 0, lv.C, r0, rend);
    __syncthreads();
    mma_chunk<BP>(acc, As, 0, Bc, wave, lane);
    if (threadIdx.x < 64) {
      float s = 0.f;
      for (int k = 0; k < 64; ++k) s += bf2f(*reinterpret_cast<const unsigned short*>(As + threadIdx.x * BP + k * 2));
      bacc += s;
    }
  }
  const bool add = p.rsplit <= 1;
  if (threadIdx.x < 64 && m0 + threadIdx.x < p.nc) bias[m0 + threadIdx.x] = (add ? bias[m0 + threadIdx.x] : 0.f) + bacc;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + NCE_ROW(i, r), n = NCE_COL(j);
        if (m < p.nc && n < N) {
          float* o = out + (long long)m * ldo + n;
          *o = (add ? *o : 0.f) + acc.a[i][j][r];
        }
      }
}

// grads += the row-split partial sums, ranges in order (grid: (elements / 256, 2, L))
__global__ __launch_bounds__(256) void nce_param_reduce_kernel(const NceK p) {
  const int which = blockIdx.y, level = blockIdx.z;
  const NceLevel& lv = p.lv[level];
  const int N = which == 0 ? p.nc : lv.C;
  const long long stride = (long long)p.nc * TN + p.nc;
  const float* part = p.wpart + (long long)(level * 2 + which) * p.rsplit * stride;
  float* gW1 = p.grads + lv.w_off;
  float* gb1 = gW1 + (long long)p.nc * lv.C;
  float* gW2 = gb1 + p.nc;
  float* gb2 = gW2 + (long long)p.nc * p.nc;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= stride) return;
  float* dst;
  if (e < (long long)p.nc * TN) {
    const int m = (int)(e / TN), n = (int)(e % TN);
    if (n >= N) return;
    dst = (which == 0 ? gW2 : gW1) + (long long)m * N + n;
  } else {
    dst = (which == 0 ? gb2 : gb1) + (e - (long long)p.nc * TN);
  }
  float s = *dst;
  for (int k = 0; k < p.rsplit; ++k) s += part[k * stride + e];
  *dst = s;
}

// loss[l] = coef * sum of the per-workgroup partials of level l, in order
__global__ void nce_loss_sum_kernel(const float* part, float* loss, int per_level, float coef) {
  const int l = blockIdx.x;
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < per_level; ++i) s += part[(long long)l * per_level + i];
    loss[l] = s * coef;
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
namespace {
constexpr int LDS_FWD = TM * AP + TN * BP + (8 * 64 + 4 * 64) * 4;
constexpr int LDS_LOSS = LDS_FWD + TM * AP + TN * BP;      // + the bf16 lo parts of the logit GEMM's operands
int fill(NceK& k, const gs_patchnce_desc* d, const float* const* xq, const float* const* xk, float* const* dxq,
         const float* params, float* grads, void* work, float* loss_part) {
  GS_REQUIRE(d && d->levels >= 1 && d->levels <= NCE_MAX_LEVELS, "gs_patchnce: 1..%d levels", NCE_MAX_LEVELS);
  GS_REQUIRE(d->nc == TN, "gs_patchnce: mlp_nc must be %d (cut.py:18 default)", TN);
  GS_REQUIRE(d->patches >= 1 && d->patches <= TN && d->batch >= 1, "gs_patchnce: 1..%d patches per image", TN);
  GS_REQUIRE(params && work, "gs_patchnce: null argument");
  k.L = d->levels; k.B = d->batch; k.P = d->patches; k.R = d->batch * d->patches; k.nc = d->nc;
  k.inv_T = 1.0f / d->nce_T;
  k.coef = d->lambda_nce / ((float)d->levels * (float)k.R);
  k.params = params; k.grads = grads; k.gscale = nullptr;
  const long long LR = (long long)k.L * k.R;
  k.side_stride = LR * k.nc;
  char* w = static_cast<char*>(work);
  k.fhat = reinterpret_cast<float*>(w); w += 2 * LR * k.nc * 4;
  k.norms = reinterpret_cast<float*>(w); w += LR * 4;
  k.h = reinterpret_cast<unsigned short*>(w); w += LR * k.nc * 2;
  k.df = reinterpret_cast<unsigned short*>(w); w += LR * k.nc * 2;
  k.dh = reinterpret_cast<unsigned short*>(w); w += LR * k.nc * 2;
  k.loss_part = loss_part;
  {   // behind the loss partials (gs_patchnce_work_bytes)
    const long long tiles = (k.P + TM - 1) / TM;
    w += ((long long)k.L * k.B * tiles * 4 + 255) / 256 * 256;
    k.wpart = reinterpret_cast<float*>(w);
    const int chunks = (k.R + 63) / 64;
    k.rsplit = chunks >= 16 ? NCE_RSPLIT : (chunks >= 4 ? chunks / 2 : 1);
    if (k.rsplit > NCE_RSPLIT) k.rsplit = NCE_RSPLIT;
  }
  long long off = 0;
  for (int l = 0; l < k.L; ++l) {
    GS_REQUIRE(d->channels[l] >= 1 && d->channels[l] <= TN, "gs_patchnce: level %d has %d channels (1..%d)", l,
               d->channels[l], TN);
    k.lv[l].xq = xq ? xq[l] : nullptr;
    k.lv[l].xk = xk ? xk[l] : nullptr;
    k.lv[l].dxq = dxq ? dxq[l] : nullptr;
    k.lv[l].C = d->channels[l];
    k.lv[l].w_off = off;
    k.lv[l].row_off = (long long)l * k.R;
    off += (long long)k.nc * d->channels[l] + k.nc + (long long)k.nc * k.nc + k.nc;
  }
  return 0;
}
template <typename Kern>
int set_lds(Kern kern, int bytes) {
  GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return 0;
}
}  // namespace

extern "C" int64_t gs_patchnce_param_floats(const gs_patchnce_desc* d) {
  if (!d || d->levels < 1 || d->levels > NCE_MAX_LEVELS) return -1;
  int64_t n = 0;
  for (int l = 0; l < d->levels; ++l) n += (int64_t)d->nc * d->channels[l] + d->nc + (int64_t)d->nc * d->nc + d->nc;
  return n;
}
extern "C" int64_t gs_patchnce_work_bytes(const gs_patchnce_desc* d) {
  if (!d || d->levels < 1) return -1;
  const int64_t LR = (int64_t)d->levels * d->batch * d->patches;
  const int64_t tiles = (d->patches + TM - 1) / TM;
  const int64_t part = ((int64_t)d->levels * d->batch * tiles * 4 + 255) / 256 * 256;
  const int64_t wpart = (int64_t)d->levels * 2 * NCE_RSPLIT * ((int64_t)d->nc * TN + d->nc) * 4;
  return 2 * LR * d->nc * 4 + LR * 4 + 3 * LR * d->nc * 2 + part + wpart + 256;
}

extern "C" int gs_patchnce_forward(const gs_patchnce_desc* d, const float* const* xq, const float* const* xk,
                                   const float* params, void* work, float* loss, void* stream) {
  NceK k;
  const int64_t LR = d ? (int64_t)d->levels * d->batch * d->patches : 0;
  float* part = d ? reinterpret_cast<float*>(static_cast<char*>(work) + 2 * LR * d->nc * 4 + LR * 4 + 3 * LR * d->nc * 2) : nullptr;
  if (int rc = fill(k, d, xq, xk, nullptr, params, nullptr, work, part)) return rc;
  GS_REQUIRE(xq && xk && loss, "gs_patchnce_forward: null argument");
  static bool configured = false;
  if (!configured) {
    if (int rc = set_lds(nce_mlp_fwd_kernel, LDS_FWD)) return rc;
    if (int rc = set_lds(nce_loss_kernel, LDS_LOSS)) return rc;
    if (int rc = set_lds(nce_mlp_bwd_kernel, LDS_FWD)) return rc;
    if (int rc = set_lds(nce_param_grad_kernel, LDS_FWD)) return rc;
    configured = true;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int row_tiles = (k.R + TM - 1) / TM, img_tiles = (k.P + TM - 1) / TM;
  hipLaunchKernelGGL(nce_mlp_fwd_kernel, dim3(row_tiles, k.L, 2), dim3(512), LDS_FWD, st, k);
  hipLaunchKernelGGL(nce_loss_kernel, dim3(img_tiles, k.B, k.L), dim3(512), LDS_LOSS, st, k);
  hipLaunchKernelGGL(nce_loss_sum_kernel, dim3(k.L), dim3(64), 0, st, k.loss_part, loss, k.B * img_tiles, k.coef);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int gs_patchnce_backward(const gs_patchnce_desc* d, const float* const* xq, float* const* dxq,
                                    const float* params, float* grads, void* work, const float* grad_scale,
                                    void* stream) {
  NceK k;
  if (int rc = fill(k, d, xq, nullptr, dxq, params, grads, work, nullptr)) return rc;
  GS_REQUIRE(xq && dxq && grads, "gs_patchnce_backward: null argument");
  k.gscale = grad_scale;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int row_tiles = (k.R + TM - 1) / TM;
  hipLaunchKernelGGL(nce_mlp_bwd_kernel, dim3(row_tiles, k.L), dim3(512), LDS_FWD, st, k);
  hipLaunchKernelGGL(nce_param_grad_kernel, dim3(k.nc / TM * k.rsplit, 2, k.L), dim3(512), LDS_FWD, st, k);
  if (k.rsplit > 1)
    hipLaunchKernelGGL(nce_param_reduce_kernel, dim3((unsigned)(((long long)k.nc * TN + k.nc + 255) / 256), 2, k.L), dim3(256),
                       0, st, k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
