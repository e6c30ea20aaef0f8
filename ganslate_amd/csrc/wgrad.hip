// Weight gradient of the generalised convolution family on the gfx950 matrix cores:
//
//   dw[p][t*Q+q] += sum_{n,z,i,j} a[n,z,i,j,p] * g[n, B(z*si+dd[t]), B(i*si+dh[t]), B(j*si+dw[t]), q]
//
// `a` is the dense side (dY for Conv, X for ConvTranspose), `g` the gathered side. Replaces the
// autograd weight-gradient kernels behind loss.backward() (ganslate/nn/gans/base.py:170) for every conv of
// resnet2d.py / patchgan2d.py and their 3-D twins resnet3d.py / patchgan3d.py (2-D = depth 1).
//
// GEMM view: C[p][n'] = A[m][p]^T * G[m][n'], contraction over pixels m. Both operands are pixel-major in
// memory (NHWC), so MFMA fragments (8 consecutive k per lane) are read with the LDS transpose read
// ds_read_b64_tr_b16 from [64 pixels][128 channels] tiles staged by LDS-DMA. The pixel range of every image
// is split over workgroups (split-K); partial tiles are accumulated with fp32 atomics.
#include "common.hpp"
#include <cstdlib>

struct WGradK {
  const char* a;
  const char* g;
  float* dw;
  const char* zero;
  int tiles_p, tiles_q, splits, chunk, q_shift;
  float rcp_wa, rcp_hw, rcp_da;
  float* ws;                  // != nullptr: split sp writes its partial tile to slab sp of ws ([splits][P * dw_ld] floats)
  long long ws_stride;        // instead of fp32 atomics on dw (deterministic accumulation, gs_wgrad_ws)
  // twin batch (gs_twin): the pixel range of each network's images is split on its own (splits = nets * splits_net, no
  // workgroup straddles the two), the second network's sums go to dw + dw_delta floats / its own slabs
  int nets, splits_net, m_net;
  int vec, rmw;               // vec: every row of dw / ws starts on a 16-byte boundary; rmw: one split per network (see the epilogue)
  long long dw_delta;
  int adam;                   // gs_wgrad_adam: the epilogue updates the parameters instead of storing the sums
  gs_adam_fuse ad;
  gs_wgrad_desc d;
};

// 16-B slot swizzle of a 256-B row R so that the 8 rows touched by one half-wave transpose read spread
// over all 16 slots
__device__ __forceinline__ int wg_swz(int R) { return ((R & 3) << 1) | (((R >> 3) & 1) << 3); }

template <int BP, int BQ, int WP, int WQ, bool VOL = true>
__global__ __launch_bounds__(WP * WQ * 64) void wgrad_kernel(const WGradK p) {
  constexpr int NW = WP * WQ, BK = 64, NSTAGE = 3;
  constexpr int ARB = BP * 2, GRB = BQ * 2;                  // row bytes of the two LDS tiles
  constexpr int AT = BK * ARB, GT = BK * GRB, STAGE = AT + GT;
  constexpr int ARI = 1024 / ARB, GRI = 1024 / GRB;          // rows per 1-KiB DMA instruction
  constexpr int A_INSTR = BK / ARI;                          // DMA instructions of the dense tile per stage
  constexpr int NAI = (A_INSTR + NW - 1) / NW, NGI = BK / GRI / NW;   // per wave per stage
  constexpr int LOADS = NAI + NGI;                           // uniform over waves (surplus lanes hit a dummy)
  constexpr int TI = BP / WP / 16, TJ = BQ / WQ / 16;
  constexpr int ASW = ARB / 16 - 1 < 15 ? ARB / 16 - 1 : 15; // swizzle mask: stay inside the row
  static_assert((BK / GRI) % NW == 0, "gathered tile must split evenly over the waves");
  static_assert(NAI == 1 || (A_INSTR % NW == 0 && (ARI * NW) % 16 == 0), "dense tile DMA layout");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* taps = reinterpret_cast<int*>(smem + NSTAGE * STAGE);
  char* dummy = smem + NSTAGE * STAGE + GS_MAX_TAPS * 4;     // 1 KiB sink for surplus DMA instructions
  const gs_wgrad_desc& d = p.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;

  int b = blockIdx.x;
  const int tq = b % p.tiles_q;
  b /= p.tiles_q;
  const int tp = b % p.tiles_p;
  b /= p.tiles_p;
  const int sp = b;                       // pixel range [k0, k1) of the batch-flattened pixel index
  const int HW = d.Ha * d.Wa;
  const int net = p.nets > 1 ? sp / p.splits_net : 0;
  const int k0 = net * p.m_net + (sp - net * p.splits_net) * p.chunk;
  const int k1 = min((net + 1) * p.m_net, k0 + p.chunk);
  if (k0 >= k1) return;

  for (int t = tid; t < d.T; t += NW * 64)
    taps[t] = ((int)d.dh[t] & 0xff) | (((int)d.dw_[t] & 0xff) << 8) | ((int)d.dd[t] << 16);
  __syncthreads();

  // ---- per-lane DMA bookkeeping: the 16-B chunk (and with it the tap / channel group) is fixed per lane ----
  const char* a_n = p.a + (size_t)d.a_co * 2;
  const char* g_n = p.g + (size_t)d.g_co * 2;
  const unsigned g_img = (unsigned)(d.Dg * d.Hg * d.Wg);
  const int a_row0 = ARI * wave + lane / (ARB / 16);         // row of this lane in its first A instruction
  const int a_slot = lane % (ARB / 16);
  const int a_chunk = a_slot ^ (wg_swz(a_row0) & ASW);       // swz is invariant under row += ARI*NW (multiple of 16)
  const int a_pch = tp * BP + a_chunk * 8;
  const bool a_cv = a_pch < d.P;
  const int g_row0 = GRI * wave + lane / (GRB / 16);
  const int g_slot = lane % (GRB / 16);
  const int g_chunk = g_slot ^ wg_swz(g_row0);
  const int g_col = tq * BQ + g_chunk * 8;                   // n' = t*Q + q
  const int g_t = (g_col >> 3) >> p.q_shift;
  const int g_q8 = (g_col >> 3) & ((1 << p.q_shift) - 1);
  const bool g_tv = g_t < d.T;
  const int tpv = taps[g_tv ? g_t : 0];
  const int g_dh = (int)(signed char)(tpv & 0xff), g_dw = (int)(signed char)((tpv >> 8) & 0xff), g_dd = tpv >> 16;
  static_assert((GRI * NW) % 16 == 0, "row step must keep the swizzle invariant");

  // pixel coordinates of this lane's gathered rows, advanced by BK pixels per K-step (no divisions in the loop)
  int gn[NGI], gz[NGI], gi[NGI], gj[NGI];
#pragma unroll
  for (int i = 0; i < NGI; ++i) {
    const int m = k0 + g_row0 + i * GRI * NW;
    const int nz = div_small(m, HW, p.rcp_hw);          // (image, depth) combined
    const int rem = m - nz * HW;
    gi[i] = div_small(rem, d.Wa, p.rcp_wa);
    gj[i] = rem - gi[i] * d.Wa;
    gn[i] = VOL ? div_small(nz, d.Da, p.rcp_da) : nz;
    gz[i] = VOL ? nz - gn[i] * d.Da : 0;
  }

  auto issue = [&](int ks, int buf) {      // always called with ks increasing by 1
    char* sb = smem + buf * STAGE;
    const int mb = k0 + ks * BK;
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
      const int R = a_row0 + i * ARI * NW;
      const int m = mb + R;
      unsigned off = ((unsigned)m * (unsigned)d.a_cs + (unsigned)a_pch) * 2u;
      asm volatile("" : "+v"(off));
      const bool real = A_INSTR % NW == 0 || wave + i * NW < A_INSTR;   // wave-uniform
      const char* src = (real && m < k1 && a_cv) ? a_n + off : p.zero;
      glds16(src, real ? sb + (wave + i * NW) * 1024 : dummy);
    }
#pragma unroll
    for (int i = 0; i < NGI; ++i) {
      const int R = g_row0 + i * GRI * NW;
      const int m = mb + R;
      bool ok = g_tv && m < k1;
      const int nn = gn[i], zz = gz[i], ii = gi[i], jj = gj[i];
      gj[i] += BK;
      while (gj[i] >= d.Wa) { gj[i] -= d.Wa; ++gi[i]; }
      int iz = 0;
      if constexpr (VOL) {
        while (gi[i] >= d.Ha) { gi[i] -= d.Ha; ++gz[i]; }
        while (gz[i] >= d.Da) { gz[i] -= d.Da; ++gn[i]; }
        iz = border_index(zz * d.si + g_dd, d.Dg, d.border, ok);
      } else {
        while (gi[i] >= d.Ha) { gi[i] -= d.Ha; ++gn[i]; }
      }
      const int ih = border_index(ii * d.si + g_dh, d.Hg, d.border, ok);
      const int iw = border_index(jj * d.si + g_dw, d.Wg, d.border, ok);
      unsigned off = (((unsigned)nn * g_img + (unsigned)((iz * d.Hg + ih) * d.Wg + iw)) * (unsigned)d.g_cs + (unsigned)(g_q8 * 8)) * 2u;
      asm volatile("" : "+v"(off));
      const char* src = ok ? g_n + off : p.zero;
      glds16(src, sb + AT + (wave + i * NW) * 1024);
    }
  };

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wp = wave / WQ, wq = wave % WQ;
  const int nk = (k1 - k0 + BK - 1) / BK;
  // transpose-read geometry: lane supplies the address of 4 consecutive channels of one pixel row
  const int fk = lane >> 4;            // 8-pixel block inside a 32-deep MFMA step
  const int frr = (lane & 15) >> 2;    // pixel row inside a 4-row block
  const int fcc = lane & 3;            // 4-channel group inside the 16-channel tile

  // per-lane LDS byte offsets of the transpose reads, hoisted out of the K loop: the swizzle term of row
  // R = kk*32 + fk*8 + h*4 + frr only depends on (frr, fk&1), so (kk, h) enter as compile-time row offsets
  int aoff[TI], goff[TJ];
  {
    const int sw = wg_swz(fk * 8 + frr);
    const int sub = (fcc & 1) * 8;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int ch = (wp * (BP / WP) / 8) + i * 2 + (fcc >> 1);
      aoff[i] = (fk * 8 + frr) * ARB + sub + ((ch ^ (sw & ASW)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int ch = (wq * (BQ / WQ) / 8) + j * 2 + (fcc >> 1);
      goff[j] = (fk * 8 + frr) * GRB + sub + ((ch ^ sw) << 4);
    }
  }

  issue(0, 0);
  if (nk > 1) issue(1, 1);

  int cur = 0, nxt2 = 2;
  for (int ks = 0; ks < nk; ++ks) {
    if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ks + 2 < nk) issue(ks + 2, nxt2);
    const char* ab = smem + cur * STAGE;
    const char* gb = ab + AT;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint2 ah[TI][2], gh[TJ][2];   // two 64-bit transpose reads = one 8-element MFMA fragment
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
          ah[i][h] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)GS_LDS(ab + aoff[i] + (kk * 32 + h * 4) * ARB)));
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          gh[j][h] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)GS_LDS(gb + goff[j] + (kk * 32 + h * 4) * GRB)));
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8, uint4{ah[i][0].x, ah[i][0].y, ah[i][1].x, ah[i][1].y}),
              __builtin_bit_cast(bf16x8, uint4{gh[j][0].x, gh[j][0].y, gh[j][1].x, gh[j][1].y}), acc[i][j], 0, 0, 0);
    }
    cur = cur == 2 ? 0 : cur + 1;
    nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
  }

  // ---- epilogue ----------------------------------------------------------------------------------------------------------
  // C[p][n'] goes to its slab (p.ws), or — one split per network: this workgroup is the only contributor of its elements —
  // is added to dw with plain loads and stores (p.rmw), or is accumulated with fp32 atomics (several splits without slabs).
  // The first two go through LDS so that every instruction moves whole rows of the tile: 16 B per lane, BQ * 4 contiguous
  // bytes per row. (The accumulator layout gives 4-byte accesses in 64-byte runs; as fp32 atomics on a 134 MB gradient —
  // the U-Net's 2048 -> 1024 transposed convolutions, one split — that ran at 1.2 TB/s.)
  const int frow = lane & 15;
  const int TQ = d.T * d.Q;
  float* const obase = p.ws ? p.ws + (size_t)sp * p.ws_stride : p.dw + (size_t)net * p.dw_delta;
  if (p.vec && (p.ws || p.rmw)) {          // (uniform)
    constexpr int CP = BQ + 4;             // floats per LDS row: the four 4-row groups of a write land on distinct banks
    static_assert(BP * CP * 4 <= NSTAGE * STAGE, "the output tile must fit the stage buffers");
    static_assert(BQ % 4 == 0 && BQ / 4 <= 64 && 64 % (BQ / 4) == 0, "row = a whole number of 16-byte lanes");
    float* ct = reinterpret_cast<float*>(smem);
    __syncthreads();                       // every wave has read its last fragments
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          ct[(wp * (BP / WP) + i * 16 + fk * 4 + r) * CP + wq * (BQ / WQ) + j * 16 + frow] = acc[i][j][r];
    __syncthreads();
    constexpr int LPR = BQ / 4, RPI = 64 / LPR;      // lanes per row, rows per instruction
    constexpr int NR = (BP + NW * RPI - 1) / (NW * RPI);
    const int c4 = (lane % LPR) * 4, rsub = lane / LPR;
    const int col = tq * BQ + c4;
    const bool rmw = p.ws == nullptr && !d.dw_fresh;     // dw_fresh: dw holds zeros (caller's guarantee), nothing to read
    f32x4 old[NR];
#pragma unroll
    for (int it = 0; it < NR; ++it) {
      const int rr = (it * NW + wave) * RPI + rsub, pp = tp * BP + rr;
      old[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rmw && rr < BP && pp < d.P && col < TQ) old[it] = *reinterpret_cast<const f32x4*>(obase + (size_t)pp * d.dw_ld + col);
    }
#pragma unroll
    for (int it = 0; it < NR; ++it) {
      const int rr = (it * NW + wave) * RPI + rsub, pp = tp * BP + rr;
      if (rr < BP && pp < d.P && col < TQ) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ct + rr * CP + c4);
        if (p.adam) {
          // the only gradient of these four elements in this optimiser step: update them here (gs_adam_step_dev_packs' arithmetic)
          const size_t e = (size_t)pp * d.dw_ld + col;
          const float b1 = p.ad.hyper[1], b2 = p.ad.hyper[2], eps = p.ad.hyper[3], bc2_sqrt = p.ad.hyper[5];
          const float step_size = p.ad.hyper[0] / p.ad.hyper[4];
          float4 P = *reinterpret_cast<const float4*>(p.ad.p + e), M = *reinterpret_cast<const float4*>(p.ad.m + e);
          float4 V = *reinterpret_cast<const float4*>(p.ad.v + e), G = {v[0], v[1], v[2], v[3]};
          adam_one(P.x, G.x, M.x, V.x, b1, b2, eps, bc2_sqrt, step_size, 1.0f, 0);
          adam_one(P.y, G.y, M.y, V.y, b1, b2, eps, bc2_sqrt, step_size, 1.0f, 0);
          adam_one(P.z, G.z, M.z, V.z, b1, b2, eps, bc2_sqrt, step_size, 1.0f, 0);
          adam_one(P.w, G.w, M.w, V.w, b1, b2, eps, bc2_sqrt, step_size, 1.0f, 0);
          *reinterpret_cast<float4*>(p.ad.p + e) = P;
          *reinterpret_cast<float4*>(p.ad.m + e) = M;
          *reinterpret_cast<float4*>(p.ad.v + e) = V;
          const uint2 o = {pack_bf2(P.x, P.y), pack_bf2(P.z, P.w)};
          const size_t grp = e >> 3, half = (e >> 2) & 1;
          const int jf = p.ad.inv_f ? p.ad.inv_f[grp] : -1, jd = p.ad.inv_d ? p.ad.inv_d[grp] : -1;
          if (jf >= 0) static_cast<uint2*>(p.ad.fpack)[(size_t)jf * 2 + half] = o;
          if (jd >= 0) static_cast<uint2*>(p.ad.dpack)[(size_t)jd * 2 + half] = o;
          if (p.ad.tr_pack) *reinterpret_cast<float4*>(ct + rr * CP + c4) = P;      // (for the transposed groups below)
        } else {
          *reinterpret_cast<f32x4*>(obase + (size_t)pp * d.dw_ld + col) = old[it] + v;
        }
      }
    }
    if (p.adam && p.ad.tr_pack) {
      // the transposed pack: 8 consecutive rows of one column are one 16-byte group (the tile of updated parameters is in LDS).
      // Neighbouring lanes take neighbouring COLUMNS (conflict-free LDS reads; with neighbouring row groups of one column per
      // lane the 8-row stride put 16 lanes on 2 banks: +25 % on the launch); a lane's 16 row groups follow each other in its
      // pack row, and L2 merges those partial-line writes
      __syncthreads();
      constexpr int RG = BP / 8;
      for (int gi = threadIdx.x; gi < RG * BQ; gi += NW * 64) {
        int rg, cc;
        if constexpr (RG % 4 == 0 && BQ % 16 == 0) {
          // 16 columns x 4 neighbouring row groups per wave: 64-byte runs of the pack rows, 2-way LDS conflicts
          cc = ((gi >> 6) % (BQ / 16)) * 16 + (gi & 15);
          rg = (gi / (64 * (BQ / 16))) * 4 + ((gi >> 4) & 3);
        } else {
          rg = gi / BQ; cc = gi - rg * BQ;
        }
        const int colt = tq * BQ + cc, p0 = tp * BP + rg * 8;
        if (colt < TQ && p0 < d.P) {
          const int t = colt >> (3 + p.q_shift), q = colt & ((8 << p.q_shift) - 1);
          const int base = p.ad.tr_base[t];
          if (base >= 0) {
            float f[8];
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) f[k8] = ct[(rg * 8 + k8) * CP + cc];
            const uint4 o = {pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7])};
            *reinterpret_cast<uint4*>(static_cast<char*>(p.ad.tr_pack) +
                                      ((size_t)base + (size_t)q * p.ad.tr_kp[t] + p0) * 2) = o;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TI; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pp = tp * BP + wp * (BP / WP) + i * 16 + fk * 4 + r;
      if (pp < d.P) {
        float* row = obase + (size_t)pp * d.dw_ld;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const int col = tq * BQ + wq * (BQ / WQ) + j * 16 + frow;
          if (col < TQ) {
            if (p.ws) row[col] = acc[i][j][r];
            else unsafeAtomicAdd(row + col, acc[i][j][r]);
          }
        }
      }
    }
  }
}

namespace {
template <int BP, int BQ, int WP, int WQ, bool VOL = true>
int launch_wgrad_impl(WGradK& k, const gs_wgrad_desc* d, hipStream_t st, int plan_only, int* slabs) {
  k.tiles_p = (d->P + BP - 1) / BP;
  k.tiles_q = (d->T * d->Q + BQ - 1) / BQ;
  // split-K over the batch-flattened pixel index: one workgroup per CU is resident (144 KiB of LDS), so aim for
  // a grid of (almost) exactly r * 256 workgroups with >= 16 K-steps each, preferring the smallest r
  const long long M = (long long)d->N / k.nets * d->Da * d->Ha * d->Wa;      // pixels per network
  GS_REQUIRE(M * k.nets < (1 << 24) && M * k.nets * d->a_cs < (1LL << 31) &&
                 (long long)d->N * d->Dg * d->Hg * d->Wg * d->g_cs < (1LL << 31),
             "gs_wgrad: tensor too large for 32-bit offsets");
  const long long tiles = (long long)k.tiles_p * k.tiles_q * k.nets;
  const long long max_splits = (M + 1023) / 1024 > 0 ? (M + 1023) / 1024 : 1;
  long long splits = 1;
  double best = 1e30;
  for (int r = 1; r <= 4; ++r) {
    long long s = (256LL * r) / tiles;
    if (s < 1) s = 1;
    if (s > max_splits) s = max_splits;
    const long long blocks = tiles * s;
    const long long rounds = (blocks + 255) / 256;
    const double cost = (double)rounds * ((double)M / (double)s + 512.0);   // K-steps per block + fixed overhead
    if (cost < best) { best = cost; splits = s; }
  }
  int chunk = (int)((M + splits - 1) / splits);
  chunk = (chunk + 63) / 64 * 64;
  splits = (M + chunk - 1) / chunk;
  k.splits = (int)splits * k.nets;
  k.splits_net = (int)splits;
  k.m_net = (int)M;
  k.chunk = chunk;
  k.rcp_hw = 1.0f / (float)(d->Ha * d->Wa);
  const long long blocks = tiles * splits;
  GS_REQUIRE(blocks > 0 && blocks < (1LL << 31), "gs_wgrad: bad grid %lld", blocks);
  // One split: every output element has a single contributing workgroup, so it accumulates straight into dw (the atomic add
  // has nothing to race with and the result is order-independent) — no slab to write and re-read. The U-Net's bottleneck
  // layers (1-128 pixels, 16.8 M weights) paid 4 x 67 MB of traffic per weight gradient for a slab of one.
  *slabs = splits == 1 ? 0 : (int)splits;
  if (splits == 1) k.ws = nullptr;
  k.rmw = splits == 1 && gs_opt(GS_OPT_WGRAD_ROWS) != 0;
  k.vec = gs_opt(GS_OPT_WGRAD_ROWS) != 0 && (reinterpret_cast<uintptr_t>(k.dw) & 15) == 0 && (k.dw_delta & 3) == 0 &&
          (reinterpret_cast<uintptr_t>(k.ws) & 15) == 0 && (k.ws_stride & 3) == 0;
  if (plan_only) return 0;
  GS_REQUIRE(!k.adam || (splits == 1 && k.vec && k.rmw && k.nets == 1 && d->dw_ld % 8 == 0),
             "gs_wgrad_adam: not a one-split launch with 16-byte rows (gs_wgrad_adam_eligible)");
  constexpr int lds = 3 * 64 * (BP + BQ) * 2 + GS_MAX_TAPS * 4 + 1024;
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<BP, BQ, WP, WQ, VOL>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  hipLaunchKernelGGL((wgrad_kernel<BP, BQ, WP, WQ, VOL>), dim3((unsigned)blocks), dim3(WP * WQ * 64), lds, st, k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
template <int BP, int BQ, int WP, int WQ>
int launch_wgrad(WGradK& k, const gs_wgrad_desc* d, hipStream_t st, int plan_only, int* slabs) {
  // depth-1 tensors (the 2-D nets) run the instantiation without the depth bookkeeping in the K loop
  if (d->Da == 1 && d->Dg == 1) return launch_wgrad_impl<BP, BQ, WP, WQ, false>(k, d, st, plan_only, slabs);
  return launch_wgrad_impl<BP, BQ, WP, WQ, true>(k, d, st, plan_only, slabs);
}
}  // namespace

// dw[e] += ws[0][e] + ws[1][e] + ... : the fixed-order second stage of the deterministic accumulation. 16 B per lane;
// 256 threads = EL element lanes x 256/EL slab lanes (slab lane l adds slabs l, l+SL, ... in order, then the SL partial
// sums are added in lane order through LDS): the order never depends on timing, and layers with few output elements but
// hundreds of slabs (the 64-channel boundary convs) still spread over the chip. EL = 16 / 4 for outputs of <= 16 / 4
// float4 (a 64-channel bias gradient over 4096 pixel chunks took 44 us with 4 slab lanes walking 1024 slabs each).
template <int EL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float4* ws, float4* dw, long long n4, int slabs,
                                                           long long stride4, long long ws_y4, long long dw_y4) {
  ws += (long long)blockIdx.y * ws_y4;             // blockIdx.y: the network of a twin launch (its slabs, its gradient buffer)
  dw += (long long)blockIdx.y * dw_y4;
  constexpr int SL = 256 / EL;
  __shared__ float4 part[SL][EL];
  const int el = threadIdx.x % EL, sl = threadIdx.x / EL;
  for (long long e0 = (long long)blockIdx.x * EL; e0 < n4; e0 += (long long)gridDim.x * EL) {
    const long long e = e0 + el;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    if (e < n4) {      // four slabs in flight per lane, added in slab order (a plain loop waits one round trip per slab)
      int k = sl;
      for (; k + 3 * SL < slabs; k += 4 * SL) {
        float4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = ws[(long long)(k + u * SL) * stride4 + e];
#pragma unroll
        for (int u = 0; u < 4; ++u) { s.x += t[u].x; s.y += t[u].y; s.z += t[u].z; s.w += t[u].w; }
      }
      for (; k < slabs; k += SL) {
        const float4 t = ws[(long long)k * stride4 + e];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
    }
    part[sl][el] = s;
    __syncthreads();
    if (sl == 0 && e < n4) {
      float4 o = dw[e];
#pragma unroll 4
      for (int l = 0; l < SL; ++l) { o.x += part[l][el].x; o.y += part[l][el].y; o.z += part[l][el].z; o.w += part[l][el].w; }
      dw[e] = o;
    }
    __syncthreads();
  }
}
// nets = 2: the same reduction for the second network of a twin launch in the same launch (its slabs ws_y4, its gradient
// buffer dw_y4 float4s further on)
static inline void launch_slab_reduce(const float* ws, float* dst, long long n4, int slabs, long long stride4,
                                      hipStream_t st, int nets = 1, long long ws_y4 = 0, long long dw_y4 = 0) {
  const float4* w4 = reinterpret_cast<const float4*>(ws);
  float4* d4 = reinterpret_cast<float4*>(dst);
  const unsigned ny = (unsigned)nets;
  if (n4 <= 4 && slabs > 16)
    hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3(1, ny), dim3(256), 0, st, w4, d4, n4, slabs, stride4, ws_y4, dw_y4);
  else if (n4 <= 16 && slabs > 16)
    hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3(1, ny), dim3(256), 0, st, w4, d4, n4, slabs, stride4, ws_y4, dw_y4);
  else {
    long long blocks = (n4 + 63) / 64;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel<64>, dim3((unsigned)blocks, ny), dim3(256), 0, st, w4, d4, n4, slabs, stride4, ws_y4,
                       dw_y4);
  }
}

// (cout1.hip)
void gs_launch_slab_reduce(const float* ws, float* dst, long long n4, int slabs, long long stride4, hipStream_t st, int nets,
                           long long ws_y4, long long dw_y4) {
  launch_slab_reduce(ws, dst, n4, slabs, stride4, st, nets, ws_y4, dw_y4);
}

// hwgrad.hip: halo-resident kernels for narrow stride-1 layers and the wide 3x3 layers
int gs_hwgrad_try2(const gs_wgrad_desc* d, const void* a, const void* g, const void* a2, const void* g2, float* dw,
                   float* ws, int plan_only, void* stream, int* handled, const gs_twin* tw);

// pwise.hip: one-tap layers with 8 channels on one side
int gs_pwise_wgrad_try(const gs_wgrad_desc* d, const void* a, const void* g, float* dw, float* ws, int plan_only, void* stream,
                       int* handled);

namespace {
// the im2col kernel for one operand pair; ws != nullptr: partial tiles to slabs, *slabs = how many
int wgrad_generic(const gs_wgrad_desc* d, const void* a, const void* g, float* dw, float* ws, int plan_only,
                  void* stream, int* slabs, const gs_twin* tw = nullptr, const gs_adam_fuse* adam = nullptr) {
  WGradK k;
  k.nets = tw ? 2 : 1;                                       // (*slabs: per network)
  k.dw_delta = tw ? tw->dw_delta / 4 : 0;
  k.a = static_cast<const char*>(a);
  k.g = static_cast<const char*>(g);
  k.dw = dw;
  k.ws = ws;
  k.ws_stride = (long long)d->P * d->dw_ld;
  k.adam = adam ? 1 : 0;
  if (adam) k.ad = *adam; else k.ad = gs_adam_fuse{};
  k.zero = static_cast<const char*>(gs_zero_page());
  GS_REQUIRE(k.zero || plan_only, "gs_wgrad: library not initialised (call gs_init)");
  int sh = 0;
  while ((8 << sh) < d->Q) ++sh;
  k.q_shift = sh;
  k.rcp_wa = 1.0f / (float)d->Wa;
  k.rcp_da = 1.0f / (float)d->Da;
  k.d = *d;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (d->P <= 16) return launch_wgrad<16, 256, 1, 8>(k, d, st, plan_only, slabs);      // Cout 1/3 layers: skinny P
  if (d->P <= 64) return launch_wgrad<64, 256, 1, 8>(k, d, st, plan_only, slabs);
  return launch_wgrad<128, 256, 4, 4>(k, d, st, plan_only, slabs);                      // 16 waves: best measured
}

int wgrad_check(const gs_wgrad_desc* d) {
  GS_REQUIRE(d->Q >= 8 && (d->Q & 7) == 0 && ((d->Q >> 3) & ((d->Q >> 3) - 1)) == 0,
             "gs_wgrad: Q=%d must be 8*2^k", d->Q);
  GS_REQUIRE((d->P & 7) == 0, "gs_wgrad: P=%d must be a multiple of 8", d->P);
  GS_REQUIRE(d->T >= 1 && d->T <= GS_MAX_TAPS, "gs_wgrad: T=%d out of range", d->T);
  GS_REQUIRE((d->a_cs & 7) == 0 && (d->a_co & 7) == 0 && (d->g_cs & 7) == 0 && (d->g_co & 7) == 0,
             "gs_wgrad: channel strides/offsets must be multiples of 8");
  GS_REQUIRE(d->Da >= 1 && d->Dg >= 1, "gs_wgrad: depths must be >= 1 (1 for 2-D tensors)");
  GS_REQUIRE(d->dw_ld == d->T * d->Q, "gs_wgrad: dw_ld must be T * Q");
  return 0;
}

int wgrad_reduce(const gs_wgrad_desc* d, const float* ws, float* dw, int slabs, void* stream) {
  const long long n = (long long)d->P * d->dw_ld;          // multiple of 64: P and Q are multiples of 8
  launch_slab_reduce(ws, dw, n / 4, slabs, n / 4, static_cast<hipStream_t>(stream));
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// one implementation behind gs_wgrad / gs_wgrad_pair / gs_wgrad_ws / gs_wgrad_ws_floats
int wgrad_impl(const gs_wgrad_desc* d, const void* a1, const void* g1, const void* a2, const void* g2, float* dw,
               float* ws, int64_t ws_floats, int plan_only, void* stream, int64_t* need_floats, const gs_twin* tw = nullptr) {
  if (int rc = wgrad_check(d)) return rc;
  const long long slab = (long long)d->P * d->dw_ld;
  const bool det = ws != nullptr || plan_only;
  int handled = 0;
  if (!a2 && !tw)
    if (int rc = gs_pwise_wgrad_try(d, a1, g1, dw, det ? ws : nullptr, plan_only, stream, &handled)) return rc;
  if (!handled)
    if (int rc = gs_hwgrad_try2(d, a1, g1, a2, g2, dw, det ? ws : nullptr, plan_only, stream, &handled, tw)) return rc;
  if (handled) {
    const int nets = tw ? 2 : 1;                             // twin batch: `handled` slabs per network
    if (need_floats) *need_floats = (int64_t)nets * handled * slab;
    if (plan_only || !det) return 0;
    GS_REQUIRE(ws_floats >= (int64_t)nets * handled * slab, "gs_wgrad_ws: workspace of %lld floats, %lld needed",
               (long long)ws_floats, (long long)nets * handled * slab);
    if (tw && tw->dw_delta % 16 == 0 && ((size_t)handled * slab) % 4 == 0) {      // both networks in one launch
      launch_slab_reduce(ws, dw, slab / 4, handled, slab / 4, static_cast<hipStream_t>(stream), 2, (long long)handled * slab / 4,
                         tw->dw_delta / 16);
      GS_CHECK_HIP(hipGetLastError());
      return 0;
    }
    if (int rc = wgrad_reduce(d, ws, dw, handled, stream)) return rc;
    if (tw) return wgrad_reduce(d, ws + (size_t)handled * slab, dw + tw->dw_delta / 4, handled, stream);
    return 0;
  }
  if (tw) {
    // the im2col kernel splits each network's pixels on its own — unless one of the halo kernels without a twin form
    // (hwgrad_kernel, hwgrad_ft) would take the halves, which is the better deal: *need_floats = -1 says "run the halves"
    gs_wgrad_desc half = *d;
    half.N = d->N / 2;
    int hh = 0;
    static const char dummy = 0;
    if (int rc = gs_hwgrad_try2(&half, &dummy, &dummy, a2 ? &dummy : nullptr, a2 ? &dummy : nullptr, nullptr, nullptr, 1, nullptr,
                                &hh, nullptr)) return rc;
    if (hh || !gs_opt(GS_OPT_WGRAD_TWIN)) {
      if (need_floats) *need_floats = -1;
      GS_REQUIRE(plan_only, "gs_wgrad_ws_twin: this layer's kernel has no twin form (gs_wgrad_twin_native): run the two halves");
      return 0;
    }
  }
  // the im2col kernel: one launch (+ one reduction) per operand pair, the workspace is reused
  const int nets = tw ? 2 : 1;
  int64_t need = 0;
  // dw_fresh holds for ONE contributor of dw: the second operand pair adds to what the first one stored
  gs_wgrad_desc second = *d;
  second.dw_fresh = 0;
  for (int pass = 0; pass < (a2 ? 2 : 1); ++pass) {
    int slabs = 0;
    if (int rc = wgrad_generic(pass ? &second : d, pass ? a2 : a1, pass ? g2 : g1, dw, det ? ws : nullptr, plan_only, stream,
                               &slabs, tw))
      return rc;
    if ((int64_t)nets * slabs * slab > need) need = (int64_t)nets * slabs * slab;
    if (plan_only || !det) continue;
    GS_REQUIRE(ws_floats >= (int64_t)nets * slabs * slab, "gs_wgrad_ws: workspace of %lld floats, %lld needed",
               (long long)ws_floats, (long long)nets * slabs * slab);
    if (slabs > 0) {
      if (tw && tw->dw_delta % 16 == 0 && ((size_t)slabs * slab) % 4 == 0) {       // both networks in one launch
        launch_slab_reduce(ws, dw, slab / 4, slabs, slab / 4, static_cast<hipStream_t>(stream), 2, (long long)slabs * slab / 4,
                           tw->dw_delta / 16);
        GS_CHECK_HIP(hipGetLastError());
        continue;
      }
      if (int rc = wgrad_reduce(d, ws, dw, slabs, stream)) return rc;
      if (tw)
        if (int rc = wgrad_reduce(d, ws + (size_t)slabs * slab, dw + tw->dw_delta / 4, slabs, stream)) return rc;
    }
  }
  if (need_floats) *need_floats = need;
  return 0;
}
}  // namespace

extern "C" int gs_wgrad(const gs_wgrad_desc* d, const void* a, const void* g, float* dw, void* stream) {
  GS_REQUIRE(d && a && g && dw, "gs_wgrad: null argument");
  return wgrad_impl(d, a, g, nullptr, nullptr, dw, nullptr, 0, 0, stream, nullptr);
}

// dw += wgrad(a1, g1) + wgrad(a2, g2) for two operand pairs of the SAME layer and shapes (the two backward passes a
// network sees per step): one launch where the kernel can merge them (fixed costs paid once), else two
// one-split layers of the im2col kernel (what no halo-resident / pointwise kernel takes): weight gradient + Adam in one launch
extern "C" int gs_wgrad_adam_eligible(const gs_wgrad_desc* d) {
  if (!d || wgrad_check(d) || !gs_opt(GS_OPT_WGRAD_ROWS) || d->dw_ld % 8) return 0;
  static const char dummy = 0;
  int handled = 0;
  if (gs_pwise_wgrad_try(d, &dummy, &dummy, nullptr, nullptr, 1, nullptr, &handled) || handled) return 0;
  if (gs_hwgrad_try2(d, &dummy, &dummy, nullptr, nullptr, nullptr, nullptr, 1, nullptr, &handled, nullptr) || handled) return 0;
  int slabs = -1;
  if (wgrad_generic(d, &dummy, &dummy, nullptr, nullptr, 1, nullptr, &slabs) || slabs != 0) return 0;
  return 1;
}
extern "C" int gs_wgrad_adam(const gs_wgrad_desc* d, const void* a, const void* g, const gs_adam_fuse* adam, void* stream) {
  GS_REQUIRE(d && a && g && adam && adam->p && adam->m && adam->v && adam->hyper, "gs_wgrad_adam: null argument");
  GS_REQUIRE((adam->inv_f == nullptr) == (adam->fpack == nullptr) && (adam->inv_d == nullptr) == (adam->dpack == nullptr),
             "gs_wgrad_adam: an index table and its pack go together");
  GS_REQUIRE(((reinterpret_cast<uintptr_t>(adam->p) | reinterpret_cast<uintptr_t>(adam->m) | reinterpret_cast<uintptr_t>(adam->v) |
               reinterpret_cast<uintptr_t>(adam->fpack) | reinterpret_cast<uintptr_t>(adam->dpack)) & 15) == 0,
             "gs_wgrad_adam: buffers must be 16-byte aligned");
  GS_REQUIRE((adam->tr_pack == nullptr) == (adam->tr_base == nullptr) && (adam->tr_pack == nullptr) == (adam->tr_kp == nullptr) &&
                 (reinterpret_cast<uintptr_t>(adam->tr_pack) & 15) == 0,
             "gs_wgrad_adam: the transposed pack comes with both of its tables, 16-byte aligned");
  GS_REQUIRE(gs_wgrad_adam_eligible(d), "gs_wgrad_adam: this layer does not run as a one-split im2col launch (gs_wgrad_adam_eligible)");
  gs_wgrad_desc fresh = *d;
  fresh.dw_fresh = 1;                                 // (nothing of a gradient buffer is read)
  int slabs = 0;
  return wgrad_generic(&fresh, a, g, adam->p, nullptr, 0, stream, &slabs, nullptr, adam);
}

extern "C" int gs_wgrad_pair(const gs_wgrad_desc* d, const void* a1, const void* g1, const void* a2, const void* g2,
                             float* dw, void* stream) {
  GS_REQUIRE(d && a1 && g1 && a2 && g2 && dw, "gs_wgrad_pair: null argument");
  return wgrad_impl(d, a1, g1, a2, g2, dw, nullptr, 0, 0, stream, nullptr);
}

// Deterministic form of both: workgroups that share output elements write partial sums to slabs of the caller's
// workspace and a second launch adds the slabs in a fixed order (no fp32 atomics: two runs give bit-identical dw).
extern "C" int64_t gs_wgrad_ws_floats(const gs_wgrad_desc* d, int32_t pair) {
  if (!d) return 0;
  int64_t need = 0;
  static const char dummy = 0;
  if (wgrad_impl(d, &dummy, &dummy, pair ? &dummy : nullptr, pair ? &dummy : nullptr, nullptr, nullptr, 0, 1, nullptr, &need))
    return -1;
  return need;
}
extern "C" int gs_wgrad_ws(const gs_wgrad_desc* d, const void* a1, const void* g1, const void* a2, const void* g2,
                           float* dw, float* ws, int64_t ws_floats, void* stream) {
  GS_REQUIRE(d && a1 && g1 && dw && (a2 == nullptr) == (g2 == nullptr), "gs_wgrad_ws: null argument");
  GS_REQUIRE(ws || gs_wgrad_ws_floats(d, a2 != nullptr) == 0, "gs_wgrad_ws: this layer needs a workspace (gs_wgrad_ws_floats)");
  return wgrad_impl(d, a1, g1, a2, g2, dw, ws, ws_floats, 0, stream, nullptr);
}

// Twin batches (gs_twin): the wide 3x3 residual convs (hwgrad.hip) take both networks' images in one launch — half the
// pixel splits per network, so half the slab traffic per gradient, one launch and one prologue instead of two.
static gs_twin twin_probe(const gs_wgrad_desc* d) { gs_twin t{}; t.n_split = d->N / 2; return t; }
extern "C" int gs_wgrad_twin_native(const gs_wgrad_desc* d, int32_t pair) {
  if (!d || d->N < 2 || (d->N & 1)) return 0;
  int64_t need = 0;
  static const char dummy = 0;
  const gs_twin t = twin_probe(d);
  if (wgrad_impl(d, &dummy, &dummy, pair ? &dummy : nullptr, pair ? &dummy : nullptr, nullptr, nullptr, 0, 1, nullptr, &need, &t))
    return 0;
  return need >= 0;
}
extern "C" int64_t gs_wgrad_ws_floats_twin(const gs_wgrad_desc* d, int32_t pair) {
  if (!d) return -1;
  int64_t need = 0;
  static const char dummy = 0;
  const gs_twin t = twin_probe(d);
  if (wgrad_impl(d, &dummy, &dummy, pair ? &dummy : nullptr, pair ? &dummy : nullptr, nullptr, nullptr, 0, 1, nullptr, &need, &t))
    return -1;
  return need;
}
extern "C" int gs_wgrad_ws_twin(const gs_wgrad_desc* d, const void* a1, const void* g1, const void* a2, const void* g2,
                                float* dw, float* ws, int64_t ws_floats, const gs_twin* tw, void* stream) {
  GS_REQUIRE(d && a1 && g1 && dw && ws && tw && (a2 == nullptr) == (g2 == nullptr), "gs_wgrad_ws_twin: null argument");
  GS_REQUIRE(2 * tw->n_split == d->N, "gs_wgrad_ws_twin: the two networks take the same number of images");
  return wgrad_impl(d, a1, g1, a2, g2, dw, ws, ws_floats, 0, stream, nullptr, tw);
}

// ---- bias gradient: db[c] += sum_pixels dy[pix][c] ---------------------------------------------------
// 256 threads = COLS 8-channel columns x (256/COLS) pixel lanes, 16-B loads; grid (pixel chunks, column groups)
template <int COLS>
__global__ __launch_bounds__(256) void bias_grad_kernel(const unsigned short* dy, long long pixels, int C8, int cs,
                                                        int co, float* db, int pix_per_block, float* ws) {
  constexpr int ROWS = 256 / COLS;
  __shared__ float red[ROWS][COLS][9];
  const int tid = threadIdx.x;
  const int col = tid % COLS, row = tid / COLS;
  const int c8 = blockIdx.y * COLS + col;
  const long long p0 = (long long)blockIdx.x * pix_per_block;
  const long long p1 = min(pixels, p0 + pix_per_block);
  float a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = 0.f;
  if (c8 < C8)
    for (long long px = p0 + row; px < p1; px += ROWS) {
      const uint4 v = *reinterpret_cast<const uint4*>(dy + px * cs + co + c8 * 8);
      a[0] += bf_lo(v.x); a[1] += bf_hi(v.x); a[2] += bf_lo(v.y); a[3] += bf_hi(v.y);
      a[4] += bf_lo(v.z); a[5] += bf_hi(v.z); a[6] += bf_lo(v.w); a[7] += bf_hi(v.w);
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[row][col][k] = a[k];
  __syncthreads();
  for (int o = tid; o < COLS * 8; o += 256) {
    const int cc = o >> 3, k = o & 7;
    const int ch8 = blockIdx.y * COLS + cc;
    if (ch8 < C8) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) s += red[r][cc][k];
      if (ws) ws[(size_t)blockIdx.x * C8 * 8 + ch8 * 8 + k] = s;      // slab per pixel chunk, summed in order afterwards
      else unsafeAtomicAdd(db + ch8 * 8 + k, s);
    }
  }
}

// db[c] += sum of the slabs' column c, slabs in order, for the first c_valid (< 8) channels only: the bias of a layer whose
// real channel count is not a multiple of 8 sits in the flat gradient buffer with the next parameter right behind it
__global__ __launch_bounds__(256) void bias_head_reduce_kernel(const float* __restrict__ ws, float* db, int slabs, int C,
                                                               int c_valid) {
  // 8 channel lanes x 32 slab lanes (slab lane l adds slabs l, l + 32, ... in order, four loads in flight), then the 32
  // partial sums in lane order: the order never depends on timing
  __shared__ float part[32][8];
  const int c = threadIdx.x & 7, sl = threadIdx.x >> 3;
  float s = 0.f;
  int k = sl;
  for (; k + 96 < slabs; k += 128) {
    float t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = ws[(size_t)(k + 32 * u) * C + c];
#pragma unroll
    for (int u = 0; u < 4; ++u) s += t[u];
  }
  for (; k < slabs; k += 32) s += ws[(size_t)k * C + c];
  part[sl][c] = s;
  __syncthreads();
  if (threadIdx.x < c_valid) {
    float o = db[threadIdx.x];
#pragma unroll 4
    for (int l = 0; l < 32; ++l) o += part[l][threadIdx.x];
    db[threadIdx.x] = o;
  }
}

static int bias_grad_impl(const void* dy, int64_t pixels, int32_t C, int32_t cs, int32_t co, float* db, float* ws,
                          int64_t ws_floats, int plan_only, void* stream, int64_t* need, int c_valid = -1) {
  GS_REQUIRE(pixels > 0 && C > 0 && (C & 7) == 0 && (cs & 7) == 0 && (co & 7) == 0,
             "gs_bias_grad: bad argument (C, cs, co must be multiples of 8)");
  const int C8 = C / 8;
  int ppb = (int)((pixels + 1023) / 1024);
  if (ppb < 64) ppb = 64;
  const unsigned bx = (unsigned)((pixels + ppb - 1) / ppb);
  if (need) *need = (int64_t)bx * C;
  if (plan_only) return 0;
  GS_REQUIRE(dy && db, "gs_bias_grad: null argument");
  GS_REQUIRE(!ws || ws_floats >= (int64_t)bx * C, "gs_bias_grad_ws: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned short* p = static_cast<const unsigned short*>(dy);
  if (C8 >= 32) hipLaunchKernelGGL((bias_grad_kernel<32>), dim3(bx, (C8 + 31) / 32), dim3(256), 0, st, p, (long long)pixels, C8, cs, co, db, ppb, ws);
  else if (C8 >= 8) hipLaunchKernelGGL((bias_grad_kernel<8>), dim3(bx, (C8 + 7) / 8), dim3(256), 0, st, p, (long long)pixels, C8, cs, co, db, ppb, ws);
  else if (C8 >= 2) hipLaunchKernelGGL((bias_grad_kernel<2>), dim3(bx, (C8 + 1) / 2), dim3(256), 0, st, p, (long long)pixels, C8, cs, co, db, ppb, ws);
  else hipLaunchKernelGGL((bias_grad_kernel<1>), dim3(bx, C8), dim3(256), 0, st, p, (long long)pixels, C8, cs, co, db, ppb, ws);
  GS_CHECK_HIP(hipGetLastError());
  if (ws && c_valid >= 0) {
    hipLaunchKernelGGL(bias_head_reduce_kernel, dim3(1), dim3(256), 0, st, ws, db, (int)bx, C, c_valid);
    GS_CHECK_HIP(hipGetLastError());
  } else if (ws) {
    const long long n4 = C / 4;
    launch_slab_reduce(ws, db, n4, (int)bx, n4, st);
    GS_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

extern "C" int gs_bias_grad(const void* dy, int64_t pixels, int32_t C, int32_t cs, int32_t co, float* db,
                            void* stream) {
  return bias_grad_impl(dy, pixels, C, cs, co, db, nullptr, 0, 0, stream, nullptr);
}
// deterministic form: per-chunk partial sums to the caller's workspace, then added to db in chunk order
extern "C" int64_t gs_bias_grad_ws_floats(int64_t pixels, int32_t C) {
  int64_t need = 0;
  return bias_grad_impl(nullptr, pixels, C, 8, 0, nullptr, nullptr, 0, 1, nullptr, &need) ? -1 : need;
}
extern "C" int gs_bias_grad_ws(const void* dy, int64_t pixels, int32_t C, int32_t cs, int32_t co, float* db, float* ws,
                               int64_t ws_floats, void* stream) {
  GS_REQUIRE(ws, "gs_bias_grad_ws: null workspace");
  return bias_grad_impl(dy, pixels, C, cs, co, db, ws, ws_floats, 0, stream, nullptr);
}
// the same over the first 8 channels of dy's channel window, adding only db[0 .. c_valid) (c_valid <= 8): db has exactly
// c_valid floats (workspace: gs_bias_grad_ws_floats(pixels, 8))
extern "C" int gs_bias_grad_head_ws(const void* dy, int64_t pixels, int32_t cs, int32_t co, int32_t c_valid, float* db,
                                    float* ws, int64_t ws_floats, void* stream) {
  GS_REQUIRE(ws && c_valid > 0 && c_valid <= 8, "gs_bias_grad_head_ws: null workspace or c_valid outside 1..8");
  return bias_grad_impl(dy, pixels, 8, cs, co, db, ws, ws_floats, 0, stream, nullptr, c_valid);
}
