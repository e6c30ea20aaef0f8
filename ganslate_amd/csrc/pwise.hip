// Pointwise (one tap at offset 0, stride 1) layers with few channels on one side — the V-Net's output conv (ganslate/nn/
// generators/vnet/vnet3d.py:246-268 OutBlock: Conv3d(32, 1, k1) after the k5 conv) over a 128^3 volume: 2 M voxels of 64 + 16
// bytes forward, the mirror image for the data gradient and a 32-number weight gradient. On the im2col kernels these three ran
// at 5-12 TFLOP/s (90 / 166 / 227 us per launch, profiles/r06_conv_table_brats_v0.txt) against ~30 us of HBM time each: with
// K = Ci <= 32 a 256-pixel tile's K loop is one step, so the launch is all prologue and epilogue.
//
// Forward / data gradient: no LDS, no im2col — a lane's 16 bytes of a voxel ARE the B operand of v_mfma_f32_16x16x32_bf16
// (column = voxel lane & 15, k octet lane >> 4), the weights are the A operand (rows = output channels) and stay in registers
// for the whole launch; the accumulator gives each lane 4 consecutive output channels of its voxel: one 8-byte store.
// Weight gradient: vector ALUs (the K dimension is the voxel index, which would want a transposing load): a lane holds the
// 8 x 8 outer-product sums of one (a octet, g octet) pair, a workgroup adds its lanes in lane order, per-workgroup slabs are
// added in slab order by the shared reduction (no atomics on the deterministic path).
#include "common.hpp"

namespace {
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct PwK {
  const char* in;
  const char* w;               // [w_rows][Kp] bf16 pack, k = ci (one tap)
  const float* bias;
  char* out;
  long long nv;                // voxels (N * D * H * W)
  int Ci, Co, in_cs, in_co, out_cs, out_co, Kp, w_rows, act;
  float slope;
  int ntiles;                  // ceil(nv / 16)
};

// KS: 32-channel k steps (Ci <= 32 KS), CT: 16-row output tiles (Co <= 16 CT); a wave walks 16-voxel tiles, UN in flight
template <int KS, int CT, int UN>
__global__ __launch_bounds__(256) void pwise_kernel(const PwK p) {
  const int lane = threadIdx.x & 63, col = lane & 15, ko = lane >> 4;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  bf16x8 wa[CT][KS];
  float bs[CT][4];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int row = ct * 16 + col;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int ci = (s * 4 + ko) * 8;
      uint4 v{0u, 0u, 0u, 0u};
      if (row < p.Co && row < p.w_rows && ci < p.Ci) v = *reinterpret_cast<const uint4*>(p.w + ((size_t)row * p.Kp + ci) * 2);
      wa[ct][s] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = ct * 16 + ko * 4 + i;
      bs[ct][i] = (p.bias && co < p.Co) ? p.bias[co] : 0.f;
    }
  }
  const char* in = p.in + (size_t)p.in_co * 2;
  char* out = p.out + (size_t)p.out_co * 2;
  for (int t0 = wave * UN; t0 < p.ntiles; t0 += nwaves * UN) {
    bf16x8 xb[UN][KS];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long v = (long long)(t0 + u) * 16 + col;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int ci = (s * 4 + ko) * 8;
        uint4 x{0u, 0u, 0u, 0u};
        if (v < p.nv && ci < p.Ci) x = *reinterpret_cast<const uint4*>(in + ((size_t)v * p.in_cs + ci) * 2);
        xb[u][s] = __builtin_bit_cast(bf16x8, x);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long v = (long long)(t0 + u) * 16 + col;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        f32x4 acc{bs[ct][0], bs[ct][1], bs[ct][2], bs[ct][3]};
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ct][s], xb[u][s], acc, 0, 0, 0);
        const int co = ct * 16 + ko * 4;
        if (v < p.nv && co < p.Co) {
          float r[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) r[i] = apply_act_small(acc[i], p.act, p.slope);
          *reinterpret_cast<uint2*>(out + ((size_t)v * p.out_cs + co) * 2) = uint2{pack_bf2(r[0], r[1]), pack_bf2(r[2], r[3])};
        }
      }
    }
  }
}

template <int KS, int CT>
void pwise_launch(const PwK& k, hipStream_t st) {
  constexpr int UN = KS * CT <= 2 ? 4 : 2;
  long long blocks = ((long long)k.ntiles + 4 * UN - 1) / (4 * UN);
  if (blocks > 2048) blocks = 2048;                  // 8 workgroups (32 waves) per CU, each wave walks its tiles
  hipLaunchKernelGGL((pwise_kernel<KS, CT, UN>), dim3((unsigned)blocks), dim3(256), 0, st, k);
}

struct PwWK {
  const char* a;
  const char* g;
  float* dst;                  // slabs [workgroups][P][Q] (ws) or dw itself (atomic adds)
  long long nv;
  int P, Q, a_cs, a_co, g_cs, g_co, atomic;
};

// G = max(P, Q) / 8 threads per voxel; thread j of a voxel: the 8 x 8 block (a octet, g octet) = (0, j) if P == 8 else (j, 0)
template <int G>
__global__ __launch_bounds__(256) void pwise_wgrad_kernel(const PwWK p) {
  __shared__ float red[256 * 64];                    // row t rotated by t: rows are written and columns summed bank-conflict free
  const int tid = threadIdx.x, j = tid % G, vl = tid / G;
  constexpr int VPB = 256 / G;                       // voxels per workgroup step
  const int ja = p.P == 8 ? 0 : j, jg = p.P == 8 ? j : 0;
  float acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = 0.f;
  const char* ap = p.a + (size_t)(p.a_co + ja * 8) * 2;
  const char* gp = p.g + (size_t)(p.g_co + jg * 8) * 2;
  const long long stride = (long long)gridDim.x * VPB;
  long long v = (long long)blockIdx.x * VPB + vl;
  auto lda = [&](long long vv) -> uint4 {
    return vv < p.nv ? *reinterpret_cast<const uint4*>(ap + (size_t)vv * p.a_cs * 2) : uint4{0u, 0u, 0u, 0u};
  };
  auto ldg = [&](long long vv) -> uint4 {
    return vv < p.nv ? *reinterpret_cast<const uint4*>(gp + (size_t)vv * p.g_cs * 2) : uint4{0u, 0u, 0u, 0u};
  };
  uint4 a0 = lda(v), g0 = ldg(v), a1 = lda(v + stride), g1 = ldg(v + stride);      // two voxels in flight
  for (; v < p.nv; v += stride) {
    const uint4 av = a0, gv = g0;
    a0 = a1; g0 = g1;
    a1 = lda(v + 2 * stride); g1 = ldg(v + 2 * stride);
    const float af[8] = {bf_lo(av.x), bf_hi(av.x), bf_lo(av.y), bf_hi(av.y), bf_lo(av.z), bf_hi(av.z), bf_lo(av.w), bf_hi(av.w)};
    const float gf[8] = {bf_lo(gv.x), bf_hi(gv.x), bf_lo(gv.y), bf_hi(gv.y), bf_lo(gv.z), bf_hi(gv.z), bf_lo(gv.w), bf_hi(gv.w)};
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] += af[a] * gf[b];
  }
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) red[tid * 64 + ((a * 8 + b + tid) & 63)] = acc[a][b];
  __syncthreads();
  // element e = (j, a, b): the sum over the workgroup's voxel lanes in lane order
  for (int e = tid; e < 64 * G; e += 256) {
    const int ej = e >> 6, ab = e & 63;
    float s = 0.f;
    for (int l = 0; l < VPB; ++l) s += red[(l * G + ej) * 64 + ((ab + l * G + ej) & 63)];
    const int pa = (p.P == 8 ? 0 : ej) * 8 + (ab >> 3), qb = (p.P == 8 ? ej : 0) * 8 + (ab & 7);
    if (p.atomic) atomicAdd(p.dst + (size_t)pa * p.Q + qb, s);
    else p.dst[(size_t)blockIdx.x * p.P * p.Q + (size_t)pa * p.Q + qb] = s;
  }
}
// ---- the 8 output-parity classes of a k2 stride-2 layer (V-Net up convs: ConvTranspose3d(k2, s2), vnet3d.py:190-215; the data
// gradient of its k2 s2 down convs): every class is a one-tap layer over the SAME input voxels, writing output voxel
// 2 v + (pz, py, px). A workgroup takes one pixel tile of the im2col kernel (bm class voxels of an image = one statistics slot
// per class, so the slot layout of gs_gconv_stat_slots stays what it is) for all classes; wave w computes classes w and w + 4
// from its own copy of the input fragment (the four waves' loads of a tile hit L1).
struct PwMK {
  const char* in;
  const char* w[8];
  const float* bias;
  char* out;
  float* stats;
  int N, Dc, Hc, Wc, Do, Ho, Wo, Ci, Co, in_cs, in_co, out_cs, out_co, Kp, w_rows, act;
  float slope, rcp_wc, rcp_hc;
  int bm, tiles_m, stats_slots;
  int cpz[8], cpy[8], cpx[8], cslot0[8];
};

template <int KS, int CT>
__global__ __launch_bounds__(256) void pwise_multi_kernel(const PwMK p) {
  const int lane = threadIdx.x & 63, col = lane & 15, ko = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.x / p.tiles_m, mt = blockIdx.x - n * p.tiles_m;
  const int pix = p.Dc * p.Hc * p.Wc;
  bf16x8 wa[2][CT][KS];
  float bs[CT][4];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const char* wc = p.w[wave + 4 * c];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int row = ct * 16 + col;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int ci = (s * 4 + ko) * 8;
        uint4 v{0u, 0u, 0u, 0u};
        if (row < p.Co && row < p.w_rows && ci < p.Ci) v = *reinterpret_cast<const uint4*>(wc + ((size_t)row * p.Kp + ci) * 2);
        wa[c][ct][s] = __builtin_bit_cast(bf16x8, v);
      }
    }
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = ct * 16 + ko * 4 + i;
      bs[ct][i] = (p.bias && co < p.Co) ? p.bias[co] : 0.f;
    }
  float s1[2][CT][4], s2[2][CT][4];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) { s1[c][ct][i] = 0.f; s2[c][ct][i] = 0.f; }
  const char* in_n = p.in + ((size_t)n * pix * p.in_cs + p.in_co) * 2;
  char* out_n = p.out + ((size_t)n * p.Do * p.Ho * p.Wo * p.out_cs + p.out_co) * 2;
  const int m0 = mt * p.bm;
  auto ldx = [&](int m, bf16x8 (&xb)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int ci = (s * 4 + ko) * 8;
      uint4 x{0u, 0u, 0u, 0u};
      if (m < pix && ci < p.Ci) x = *reinterpret_cast<const uint4*>(in_n + ((size_t)m * p.in_cs + ci) * 2);
      xb[s] = __builtin_bit_cast(bf16x8, x);
    }
  };
  bf16x8 xcur[KS], xnext[KS];
  ldx(m0 + col, xcur);
  const int ntile = p.bm >> 4;
#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    const int m = m0 + t * 16 + col;
    if (t + 1 < ntile) ldx(m + 16, xnext);
    const bool valid = m < pix;
    const int zy = div_small(m, p.Wc, p.rcp_wc);
    const int x = m - zy * p.Wc;
    const int z = div_small(zy, p.Hc, p.rcp_hc);
    const int y = zy - z * p.Hc;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int cls = wave + 4 * c;
      const size_t ov = ((size_t)(2 * z + p.cpz[cls]) * p.Ho + (2 * y + p.cpy[cls])) * p.Wo + (2 * x + p.cpx[cls]);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        f32x4 acc{bs[ct][0], bs[ct][1], bs[ct][2], bs[ct][3]};
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c][ct][s], xcur[s], acc, 0, 0, 0);
        const int co = ct * 16 + ko * 4;
        if (valid && co < p.Co) {
          float r[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s1[c][ct][i] += acc[i];
            s2[c][ct][i] += acc[i] * acc[i];
            r[i] = apply_act_small(acc[i], p.act, p.slope);
          }
          *reinterpret_cast<uint2*>(out_n + (ov * p.out_cs + co) * 2) = uint2{pack_bf2(r[0], r[1]), pack_bf2(r[2], r[3])};
        }
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) xcur[s] = xnext[s];
  }
  if (p.stats_slots > 0) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int cls = wave + 4 * c;
      float* spt = p.stats + (((size_t)n * p.stats_slots + p.cslot0[cls] + mt) * 2) * p.Co;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a = row16_sum(s1[c][ct][i]), q = row16_sum(s2[c][ct][i]);
          const int co = ct * 16 + ko * 4 + i;
          if (col == 0 && co < p.Co) { spt[co] = a; spt[p.Co + co] = q; }
        }
    }
  }
}
}  // namespace

static bool pwise_conv_ok(const gs_gconv_desc* d) {
  return gs_opt(GS_OPT_PWISE) && d->T == 1 && d->so == 1 && d->si == 1 && !d->dd[0] && !d->dh[0] && !d->dw[0] && !d->pz && !d->py &&
         !d->px && d->Dc == d->Do && d->Do == d->Di && d->Hc == d->Ho && d->Ho == d->Hi && d->Wc == d->Wo && d->Wo == d->Wi &&
         d->stats_slots == 0 && !d->accumulate && d->Ci <= 64 && d->Co <= 32 && d->Ci * d->Co <= 512 &&
         (long long)d->N * d->Do * d->Ho * d->Wo >= (long long)gs_opt(GS_OPT_PWISE) * 2048 &&
         (long long)d->N * d->Do * d->Ho * d->Wo < (1LL << 34);
}

// gconv.hip: *handled = 1 when the launch went out here
int gs_pwise_try(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out, void* stream,
                 int* handled) {
  *handled = 0;
  if (!pwise_conv_ok(d)) return 0;
  PwK k;
  k.in = static_cast<const char*>(in); k.w = static_cast<const char*>(w_pack); k.bias = bias; k.out = static_cast<char*>(out);
  k.nv = (long long)d->N * d->Do * d->Ho * d->Wo;
  k.Ci = d->Ci; k.Co = d->Co; k.in_cs = d->in_cs; k.in_co = d->in_co; k.out_cs = d->out_cs; k.out_co = d->out_co;
  k.Kp = d->Kp; k.w_rows = d->w_rows; k.act = d->act; k.slope = d->slope;
  k.ntiles = (int)((k.nv + 15) / 16);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ks = (d->Ci + 31) / 32, ct = (d->Co + 15) / 16;
  if (ks == 1 && ct == 1) pwise_launch<1, 1>(k, st);
  else if (ks == 1 && ct == 2) pwise_launch<1, 2>(k, st);
  else if (ks == 2 && ct == 1) pwise_launch<2, 1>(k, st);
  else return 0;
  GS_CHECK_HIP(hipGetLastError());
  *handled = 1;
  return 0;
}

// wgrad.hip: same contract as gs_hwgrad_try2 — *handled = slabs written to ws (deterministic path: the caller adds them to dw
// in slab order), or 1 after atomic adds into dw
int gs_pwise_wgrad_try(const gs_wgrad_desc* d, const void* a, const void* g, float* dw, float* ws, int plan_only, void* stream,
                       int* handled) {
  *handled = 0;
  const long long nv = (long long)d->N * d->Da * d->Ha * d->Wa;
  if (!gs_opt(GS_OPT_PWISE) || d->T != 1 || d->si != 1 || d->dd[0] || d->dh[0] || d->dw_[0] || d->Da != d->Dg || d->Ha != d->Hg ||
      d->Wa != d->Wg || (d->P != 8 && d->Q != 8) || d->P > 32 || d->Q > 32 || nv < (long long)gs_opt(GS_OPT_PWISE) * 2048)
    return 0;
  const int G = (d->P > d->Q ? d->P : d->Q) / 8;
  if (G != 1 && G != 2 && G != 4) return 0;
  const int vpb = 256 / G;
  long long groups = (nv + vpb * 8 - 1) / (vpb * 8);             // >= 8 voxels per lane
  if (groups > 512) groups = 512;
  *handled = ws || plan_only ? (int)groups : 1;
  if (plan_only) return 0;
  PwWK k;
  k.a = static_cast<const char*>(a); k.g = static_cast<const char*>(g);
  k.dst = ws ? ws : dw;
  k.atomic = ws ? 0 : 1;
  k.nv = nv; k.P = d->P; k.Q = d->Q; k.a_cs = d->a_cs; k.a_co = d->a_co; k.g_cs = d->g_cs; k.g_co = d->g_co;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (G == 1) hipLaunchKernelGGL(pwise_wgrad_kernel<1>, dim3((unsigned)groups), dim3(256), 0, st, k);
  else if (G == 2) hipLaunchKernelGGL(pwise_wgrad_kernel<2>, dim3((unsigned)groups), dim3(256), 0, st, k);
  else hipLaunchKernelGGL(pwise_wgrad_kernel<4>, dim3((unsigned)groups), dim3(256), 0, st, k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// gconv.hip (gs_gconv_forward_multi): the 8 one-tap parity classes of a k2 stride-2 volume layer in one launch
int gs_pwise_multi_try(const gs_gconv_desc* const* descs, int count, const void* in, const void* const* w_packs, const float* bias,
                       void* out, float* stats, void* stream, int* handled) {
  *handled = 0;
  const gs_gconv_desc* d = descs[0];
  if (!gs_opt(GS_OPT_PWISE) || count != 8 || d->so != 2 || d->si != 1 || d->Co > 64 || d->Ci > 128 || d->accumulate) return 0;
  const long long pix = (long long)d->Dc * d->Hc * d->Wc;
  if (d->Dc != d->Di || d->Hc != d->Hi || d->Wc != d->Wi || d->Do != 2 * d->Dc || d->Ho != 2 * d->Hc || d->Wo != 2 * d->Wc ||
      pix >= (1LL << 24) || (long long)d->N * pix < (long long)gs_opt(GS_OPT_PWISE) * 2048)
    return 0;
  for (int c = 0; c < count; ++c) {
    const gs_gconv_desc* dc = descs[c];
    if (dc->T != 1 || dc->dd[0] || dc->dh[0] || dc->dw[0] || dc->Kp != d->Kp || (unsigned)dc->pz > 1u || (unsigned)dc->py > 1u ||
        (unsigned)dc->px > 1u)
      return 0;
  }
  PwMK k;
  k.in = static_cast<const char*>(in);
  for (int c = 0; c < 8; ++c) {
    k.w[c] = static_cast<const char*>(w_packs[c]);
    k.cpz[c] = descs[c]->pz; k.cpy[c] = descs[c]->py; k.cpx[c] = descs[c]->px; k.cslot0[c] = descs[c]->stats_slot0;
  }
  k.bias = bias; k.out = static_cast<char*>(out); k.stats = stats;
  k.N = d->N; k.Dc = d->Dc; k.Hc = d->Hc; k.Wc = d->Wc; k.Do = d->Do; k.Ho = d->Ho; k.Wo = d->Wo;
  k.Ci = d->Ci; k.Co = d->Co; k.in_cs = d->in_cs; k.in_co = d->in_co; k.out_cs = d->out_cs; k.out_co = d->out_co;
  k.Kp = d->Kp; k.w_rows = d->w_rows; k.act = d->act; k.slope = d->slope;
  k.rcp_wc = 1.0f / (float)d->Wc; k.rcp_hc = 1.0f / (float)d->Hc;
  k.bm = d->Co <= 16 ? 256 : 128;                    // gconv.hip pick_tile: one statistics slot per pixel tile and class
  k.tiles_m = (int)((pix + k.bm - 1) / k.bm);
  k.stats_slots = d->stats_slots;
  const long long blocks = (long long)d->N * k.tiles_m;
  if (blocks >= (1LL << 31)) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ks = (d->Ci + 31) / 32, ct = (d->Co + 15) / 16;
#define GS_PWM(KS_, CT_)                                                                                        \
  if (ks == KS_ && ct == CT_) {                                                                                 \
    hipLaunchKernelGGL((pwise_multi_kernel<KS_, CT_>), dim3((unsigned)blocks), dim3(256), 0, st, k);            \
    GS_CHECK_HIP(hipGetLastError());                                                                            \
    *handled = 1;                                                                                               \
    return 0;                                                                                                   \
  }
  GS_PWM(1, 1) GS_PWM(2, 1) GS_PWM(4, 1) GS_PWM(1, 2) GS_PWM(2, 2) GS_PWM(4, 2) GS_PWM(2, 4) GS_PWM(4, 4)
#undef GS_PWM
  return 0;
}
