// Generalised convolution as an implicit GEMM on the gfx950 matrix cores.
//
//   out[n, z*so+pz, i*so+py, j*so+px, co] =
//       act(bias[co] + sum_t sum_ci in[n, B(z*si+dd[t]), B(i*si+dh[t]), B(j*si+dw[t]), ci] * w[co][t*Ci+ci])
//
// One kernel covers Conv2d/Conv3d forward (any stride), the data-gradient of stride-1 convs, and — one launch per
// output parity class — ConvTranspose2d/3d(stride 2) forward and the data-gradient of stride-2 convs.
// Replaces the cuDNN/MIOpen calls behind nn.Conv2d / nn.ConvTranspose2d in
// ganslate/nn/generators/resnet/resnet2d.py:25,35,52-57,65,80-87 and
// ganslate/nn/discriminators/patchgan/patchgan2d.py:29,36-62 (forward) and their autograd backward, and their
// 3-D twins (resnet3d.py:25-64,78-84, patchgan3d.py:28-60). A 2-D tensor is a volume of depth 1: the (depth, row)
// pair is one "row" of the gather tables, so the K loop is the same for both.
//
// GEMM view: D[co][pixel] = W[co][k] * X[pixel][k]^T, k = (tap, ci). Weights are the MFMA A operand so each
// lane ends up with 4 consecutive output channels of one pixel (8-byte NHWC stores).
// Both tiles are staged by LDS-DMA (global_load_lds_dwordx4): the LDS image is lane-linear, rows of 64 bf16
// (128 B), and the bank swizzle (16-B slot ^= row&7) is applied on the per-lane SOURCE address and again on
// the ds_read_b128 address. The im2col gather happens in that per-lane source address (reflect / replicate /
// zero borders; out-of-range lanes read a zero page). Two LDS stages, one barrier per K-step of 64.
#include "common.hpp"
#include <cstdlib>

#include "gconv.hpp"
#include <algorithm>

template <int BM, int BN, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(WM * WN * 64) void gconv_kernel(const GConvK p) {
  constexpr int NW = WM * WN;           // waves per workgroup
  constexpr int WT = BN * 128;          // weight tile bytes per stage
  constexpr int XT = BM * 128;          // pixel tile bytes per stage
  constexpr int STAGE = WT + XT;
  constexpr int TI = BN / WN / 16;      // 16-channel tiles per wave
  constexpr int TJ = BM / WM / 16;      // 16-pixel tiles per wave
  constexpr int NXI = BM / 8 / NW;      // pixel-tile DMA instructions per wave per stage
  constexpr int NWI = (BN / 8 + NW - 1) / NW;  // weight-tile DMA instructions per wave per stage (upper bound)
  constexpr bool W_UNIFORM = (BN / 8) % NW == 0;
  constexpr int LOADS = NXI + NWI;      // DMA instructions per stage per wave (when W_UNIFORM)
  static_assert(BM % (8 * NW) == 0, "pixel tile must split evenly over the waves");
  static_assert(NSTAGE == 2 || ((NSTAGE == 3 || NSTAGE == 4) && W_UNIFORM), "3 / 4 stages need a uniform DMA count per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  short* taps = reinterpret_cast<short*>(smem + NSTAGE * STAGE);

  const gs_gconv_desc& d = p.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;

  // XCD-aware remap (blocks are dispatched round-robin over the 8 XCDs): give every XCD a contiguous run of
  // logical tiles so the workgroups sharing a pixel tile / neighbouring halos also share an L2
  int b;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  // merged parity classes: the classes of one pixel tile are neighbours in the grid (they gather the same input rows)
  const bool multi = p.n_cls > 0;
  int b2 = b;
  const int sp = b2 % p.splits; b2 /= p.splits;
  const int nt = b2 % p.tiles_n; b2 /= p.tiles_n;
  const int ci = multi ? b2 % p.n_cls : 0;
  if (multi) b2 /= p.n_cls;
  const int mt = b2 % p.tiles_m;
  const int n = b2 / p.tiles_m;
  // (a weight-major order — the pixel tiles that read one weight tile as neighbours — measured 1 % slower on the U-Net's
  // weight-heavy split-K launches: profiles/r05_ab_splitk_wmajor.txt)
  const GConvCls& cl = p.cls[ci];
  const int cT = multi ? cl.T : d.T, cKp = multi ? cl.Kp : d.Kp;
  const int cpz = multi ? cl.pz : d.pz, cpy = multi ? cl.py : d.py, cpx = multi ? cl.px : d.px;
  const int cslot0 = multi ? cl.stats_slot0 : d.stats_slot0;
  const int cnh = multi ? cl.nh : p.nh, cnw = multi ? cl.nw : p.nw;
  const bool second = n >= p.nsplit;                   // workgroup-uniform: a tile never straddles images
  const char* cw = p.w + (multi ? cl.w_off : 0) + (second ? p.w_delta : 0);
  const float* cbias = p.bias ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.bias) + (second ? p.bias_delta : 0))
                              : nullptr;

  for (int t = tid; t < cT; t += NW * 64) {
    const int th = multi ? (int)cl.tap_h[t] : (int)p.tap_h[t], tw = multi ? (int)cl.tap_w[t] : (int)p.tap_w[t];
    taps[t] = (short)(th | (tw << 8));   // (row-table, column-table) index
  }

  // ---- per-lane DMA bookkeeping -------------------------------------------------------------------
  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;  // 16-B k-chunk fetched by this lane (swizzled source)
  const int HWc = d.Dc * d.Hc * d.Wc;
  // Gather tables (built once per workgroup): tabh[h][row] = B(z*si+ud[h])*Hi + B(i*si+uh[h]) and
  // tabw[w][row] = B(j*si+uw[w]) for the distinct tap (depth,row) / column offsets, GS_TAB_BAD for a masked row or a zero-padded tap. The K loop then needs two
  // ds_reads + a multiply-add per DMA instruction instead of ~35 VALU of border / stride arithmetic (the loop was
  // VALU-issue bound: 200 VALU per 32 MFMA).
  constexpr unsigned GS_TAB_BAD = 0x8000u;
  unsigned short* tabh = reinterpret_cast<unsigned short*>(smem + NSTAGE * STAGE + GS_MAX_TAPS * 2);
  unsigned short* tabw = tabh + cnh * BM;
  for (int e = tid; e < BM * (cnh + cnw); e += NW * 64) {
    const int k = e / BM, row = e - k * BM;
    const int m = mt * BM + row;
    bool ok = m < HWc;
    const int zi = div_small(m, d.Wc, p.rcp_wc);
    const int jj = m - zi * d.Wc;
    const int zz = div_small(zi, d.Hc, p.rcp_hc);
    const int ii = zi - zz * d.Hc;
    unsigned v;
    if (k < cnh) {
      const int od = multi ? (int)cl.ud[k] : (int)p.ud[k], oh = multi ? (int)cl.uh[k] : (int)p.uh[k];
      const int iz = border_index(zz * d.si + od, d.Di, d.border, ok);
      v = (unsigned)(iz * d.Hi + border_index(ii * d.si + oh, d.Hi, d.border, ok));
    } else {
      const int ow = multi ? (int)cl.uw[k - cnh] : (int)p.uw[k - cnh];
      v = (unsigned)border_index(jj * d.si + ow, d.Wi, d.border, ok);
    }
    tabh[e] = (unsigned short)(ok ? v : GS_TAB_BAD);
  }
  const char* in_n = p.in + ((size_t)n * d.Di * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
  const char* wsrc[NWI];
  int winc[NWI];  // 128 B per K-step for a real weight row, 0 for a masked row (stays on the zero page)
#pragma unroll
  for (int i = 0; i < NWI; ++i) {
    const int wi = wave + NW * i;
    const int co = nt * BN + wi * 8 + lrow;
    const bool wv = co < d.w_rows;
    const char* real = cw + ((size_t)co * cKp + chunk * 8) * 2;
    wsrc[i] = wv ? real : p.zero;
    winc[i] = wv ? 128 : 0;
  }
  const int cmask = (1 << p.ci_shift) - 1;

  // K order: for Ci >= 64 the K-steps walk taps fastest inside one 64-channel chunk, so the 9 (or 16, 49)
  // shifted re-reads of an input chunk are back to back and hit L2; for Ci < 64 a K-step spans several taps
  // and the natural (tap, channel) order of the pack is kept. `issue` is always called with ks increasing by 1.
  const bool chunk_major = d.Ci >= 64;
  const int nk = cKp >> 6;
  const int ks_begin = (int)((long long)sp * nk / p.splits), ks_end = (int)((long long)(sp + 1) * nk / p.splits);
  int it_t = ks_begin % cT, it_c = ks_begin / cT;     // (tap, chunk) of the first K-step (chunk-major order)
  // Staging of a K-step in two parts: `prep` is the address generation (tap / table look-ups, border select: ~50 VALU
  // per wave), `fire` the NWI + NXI LDS-DMA instructions on the prepared addresses. Issued back to back behind the barrier
  // (`issue`) they put every wave of a SIMD into a VALU phase at the same moment, followed by an MFMA phase with the VALU
  // idle (SQ_INSTS_VALU: 99 VALU per wave per K-step against 24 MFMAs in the 288-pixel tile, profiles/r02_trunk_pmc.txt);
  // the 2-stage loop below runs prep(ks + 2) between the two 32-deep halves of compute(ks) instead.
  const char* nx_src[NXI];
  int n_q0 = 0;
  auto prep = [&](int ks) {
    int q0;                                   // first 16-B k-group of this K-step inside a pack row
    if (chunk_major) {
      q0 = (it_t << p.ci_shift) + it_c * 8;
      if (++it_t == cT) { it_t = 0; ++it_c; }
    } else {
      q0 = ks * 8;
    }
    n_q0 = q0;
    const int q = q0 + chunk;
    const int t = q >> p.ci_shift;
    const int c8 = q & cmask;
    const bool tv = t < cT;
    const int tt = tv ? t : 0;
    const int tp = taps[tt];
    const unsigned short* hrow = tabh + (tp & 0xff) * BM + wave * 8 + lrow;
    const unsigned short* wrow = tabw + (tp >> 8) * BM + wave * 8 + lrow;
    const unsigned wi = (unsigned)d.Wi;
    const unsigned cs2 = (unsigned)d.in_cs * 2u;
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const unsigned a = hrow[NW * 8 * i], bq = wrow[NW * 8 * i];
      const bool ok = tv && !((a | bq) & GS_TAB_BAD);
      unsigned off = (a * wi + bq) * cs2 + (unsigned)(c8 * 16);
      asm volatile("" : "+v"(off));  // keep the address math unconditional: select, don't branch
      nx_src[i] = ok ? in_n + off : p.zero;
    }
  };
  auto fire = [&](int buf) {
    char* sb = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int wi = wave + NW * i;
      if (W_UNIFORM || wi < BN / 8) glds16(wsrc[i] + n_q0 * (winc[i] >> 3), sb + wi * 1024);
    }
#pragma unroll
    for (int i = 0; i < NXI; ++i) glds16(nx_src[i], sb + WT + (wave + NW * i) * 1024);
  };
  auto issue = [&](int ks, int buf) { prep(ks); fire(buf); };

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 15;           // row inside a 16-row fragment
  const int fk = lane >> 4;             // k-chunk (8 bf16) inside a 32-deep MFMA step

  const int swz = lane & 7;             // == row & 7 for every fragment row of this lane

  __syncthreads();  // taps visible
  if constexpr (NSTAGE == 2) {
    issue(ks_begin, 0);
    if (ks_begin + 1 < ks_end) prep(ks_begin + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else if constexpr (NSTAGE == 4) {
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2)
      if (ks_begin + s2 < ks_end) issue(ks_begin + s2, s2);
  } else {
    issue(0, 0);
    if (nk > 1) issue(1, 1);
  }

  // (Fragment reads stay compiler-managed here: issuing both 32-deep halves through lds_read128 with counted waits, as
  // hconvw / hwgrad do, measured no gain on these two-wave-per-SIMD+ tiles and cost the 320-pixel tile 28 spills.)
  auto compute_half = [&](int cur, int kk) {
    const char* wb = smem + cur * STAGE + (wn * (BN / WN) + frow) * 128;
    const char* xb = smem + cur * STAGE + WT + (wm * (BM / WM) + frow) * 128;
    {
      const int coff = ((kk * 4 + fk) ^ swz) << 4;
      bf16x8 wf[TI], xf[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(wb + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < TJ; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xb + j * 16 * 128 + coff);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  };
  auto compute = [&](int cur) { compute_half(cur, 0); compute_half(cur, 1); };

  if constexpr (NSTAGE == 2) {
    for (int ks = ks_begin; ks < ks_end; ++ks) {
      const int cur = (ks - ks_begin) & 1;
      if (ks + 1 < ks_end) fire(cur ^ 1);               // addresses of K-step ks + 1 were prepared one iteration ago
      compute_half(cur, 0);
      if (ks + 2 < ks_end) prep(ks + 2);                // the DMA of ks + 1 is in flight while ks + 2 is being addressed
      compute_half(cur, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next stage landed (this wave's share)
      __syncthreads();
    }
  } else if constexpr (NSTAGE == 4) {
    // 4-stage ring over this split's K range, DMA three K-steps ahead: the split-K launches stream COLD weights (every
    // byte once, from HBM), so a K-step lasts as long as the memory latency divided by the stages in flight — with the
    // 2-stage loop above that was one stage per workgroup, 1.5-2 us per K-step for 0.2 us of MFMA work.
    int cur = 0, nxt3 = 3;
    for (int ks = ks_begin; ks < ks_end; ++ks) {
      const int ahead = ks_end - 1 - ks;               // K-steps behind this one that have been issued: min(ahead, 2)
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (ks + 3 < ks_end) issue(ks + 3, nxt3);
      compute(cur);
      cur = cur == 3 ? 0 : cur + 1;
      nxt3 = nxt3 == 3 ? 0 : nxt3 + 1;
    }
    __syncthreads();
  } else {
    // 3-stage ring, DMA two K-steps ahead; one raw barrier per K-step; counted vmcnt keeps the younger stage in
    // flight across the barrier (a __syncthreads() here would drain it: LDS-DMA counts as a pending LDS write)
    int cur = 0, nxt2 = 2;
    for (int ks = 0; ks < nk; ++ks) {
      if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (ks + 2 < nk) issue(ks + 2, nxt2);
      compute(cur);
      cur = cur == 2 ? 0 : cur + 1;
      nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
    }
    __syncthreads();
  }

  if constexpr (NSTAGE == 2 || NSTAGE == 4) {
    if (p.splits > 1) {       // raw partial sums, dense [output pixel][Co] per split; the epilogue runs in the finalize pass
      float* part = p.partial + (size_t)sp * p.split_stride;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int m = mt * BM + wm * (BM / WM) + j * 16 + frow;
        if (m >= HWc) continue;
        const int zi = div_small(m, d.Wc, p.rcp_wc);
        const int jj = m - zi * d.Wc;
        const int zz = div_small(zi, d.Hc, p.rcp_hc);
        const int ii = zi - zz * d.Hc;
        const size_t opix = (((size_t)n * d.Do + (zz * d.so + cpz)) * d.Ho + (ii * d.so + cpy)) * d.Wo + (jj * d.so + cpx);
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const int co = nt * BN + wn * (BN / WN) + i * 16 + fk * 4;
          if (co < d.Co) *reinterpret_cast<f32x4*>(part + opix * d.Co + co) = acc[i][j];
        }
      }
      return;
    }
  }
  // this lane's bias values as TI vector loads issued back to back (one exposed latency; as 4*TI dependent scalar loads
  // inside the loops below they cost ~6 us per launch, and fetched before the K loop they cost registers in it)
  f32x4 bia[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int co = nt * BN + wn * (BN / WN) + i * 16 + fk * 4;
    bia[i] = (cbias && co < d.Co) ? *reinterpret_cast<const f32x4*>(cbias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // ---- epilogue: bias, InstanceNorm partial statistics, activation, bf16 NHWC store -----------------
  const bool want_stats = d.stats_slots > 0;
  float s1[TI][4], s2[TI][4];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;

  // The accumulators hold, per lane, 4 consecutive channels of one pixel. They are staged through a wave-private
  // LDS slab [pixels of the wave][channels of the wave] so the global stores are 16 B per lane and a pixel's
  // channel run is written as one contiguous line (the direct 8-B stores were store-issue bound).
  constexpr int CW = BN / WN;                 // channels per wave
  constexpr int PW = BM / WM;                 // pixels per wave
  constexpr int SROW = CW * 2 + 16;           // padded slab row (bytes), keeps 16-B alignment
  constexpr int RED_BYTES = WM * BN * 3 * 4;  // [WM][BN][2] statistics, or [WM][BN][3] fused norm-backward sums
  char* slab = smem + ((RED_BYTES + 255) / 256) * 256 + wave * (PW * SROW);
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int m = mt * BM + wm * PW + j * 16 + frow;
    const bool pv = m < HWc;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int co = nt * BN + wn * CW + i * 16 + fk * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] + bia[i][r];
        if (pv) { s1[i][r] += v[r]; s2[i][r] += v[r] * v[r]; }
        v[r] = apply_act_small(v[r], d.act, d.slope);
      }
      uint2 o;
      o.x = pack_bf2(v[0], v[1]);
      o.y = pack_bf2(v[2], v[3]);
      *reinterpret_cast<uint2*>(slab + (j * 16 + frow) * SROW + (i * 16 + fk * 4) * 2) = o;
    }
  }
  // per-wave statistics to LDS before the store loop, so the s1/s2 registers are dead inside it
  if (want_stats) {
    float* red = reinterpret_cast<float*>(smem);  // [WM][BN][2] in front of the store slabs
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[i][r], q = s2[i][r];
        a = row16_sum(a);
        q = row16_sum(q);
        if (frow == 0) {
          const int cl = wn * (BN / WN) + i * 16 + fk * 4 + r;
          red[(wm * BN + cl) * 2 + 0] = a;
          red[(wm * BN + cl) * 2 + 1] = q;
        }
      }
  }
  __syncthreads();
  if (want_stats) {
    float* red = reinterpret_cast<float*>(smem);
    if (tid < BN) {
      const int co = nt * BN + tid;
      if (co < d.Co) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { a += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
        float* sp = p.stats + (((size_t)n * d.stats_slots + cslot0 + mt) * 2) * d.Co;
        sp[co] = a;
        sp[d.Co + co] = q;
      }
    }
  }
  {
    constexpr int LPR = CW / 8;               // lanes per pixel row (16 B each)
    constexpr int PPI = 64 / LPR;             // pixels per store instruction
    const int sub = lane % LPR, prow = lane / LPR;
    const int co = nt * BN + wn * CW + sub * 8;
    // fused first pass of the consumer's InstanceNorm backward (gs_gconv_forward_fused): sums over this tile of
    // ghat = (g + g2) * act'(yhat), ghat * yhat, yhat, with yhat taken at the pixel the padding folds this one onto
    const bool fuse = p.f.partial != nullptr;
    float fa1[8], fa2[8], fa3[8], fmu[8], frs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) fa1[k] = fa2[k] = fa3[k] = fmu[k] = frs[k] = 0.f;
    if (fuse && co < d.Co) {
      const float* mr = p.f.mean_rstd + (size_t)n * 2 * d.Co;
#pragma unroll
      for (int k = 0; k < 8; ++k) { fmu[k] = mr[co + k]; frs[k] = mr[d.Co + co + k]; }
    }
#pragma unroll
    for (int it = 0; it < PW / PPI; ++it) {
      const int pl = it * PPI + prow;
      const int m = mt * BM + wm * PW + pl;
      if (m < HWc && co < d.Co) {
        const int zi = div_small(m, d.Wc, p.rcp_wc);
        const int jj = m - zi * d.Wc;
        const int zz = div_small(zi, d.Hc, p.rcp_hc);
        const int ii = zi - zz * d.Hc;
        const size_t opix = (((size_t)n * d.Do + (zz * d.so + cpz)) * d.Ho + (ii * d.so + cpy)) * d.Wo + (jj * d.so + cpx);
        uint4 val = *reinterpret_cast<const uint4*>(slab + pl * SROW + sub * 16);
        uint4* dst = reinterpret_cast<uint4*>(p.out + (opix * d.out_cs + d.out_co + co) * 2);
        if (fuse) {
          const int fd = p.f.Dy > 1 ? p.f.fold : 0;
          const int uz = zz - fd, uy = ii - p.f.fold, ux = jj - p.f.fold;
          const bool interior = (unsigned)uz < (unsigned)p.f.Dy && (unsigned)uy < (unsigned)p.f.Hy &&
                                (unsigned)ux < (unsigned)p.f.Wy;
          bool okd = true;
          int yz = border_index(uz, p.f.Dy, p.f.fold_mode, okd);
          int yy = border_index(uy, p.f.Hy, p.f.fold_mode, okd);
          int yx = border_index(ux, p.f.Wy, p.f.fold_mode, okd);
          yz = min(max(yz, 0), p.f.Dy - 1); yy = min(max(yy, 0), p.f.Hy - 1); yx = min(max(yx, 0), p.f.Wy - 1);
          const size_t ypix = (((size_t)n * p.f.Dy + yz) * p.f.Hy + yy) * p.f.Wy + yx;
          const uint4 yv = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.f.y) + (ypix * d.Co + co) * 2);
          float g[8] = {bf_lo(val.x), bf_hi(val.x), bf_lo(val.y), bf_hi(val.y),
                        bf_lo(val.z), bf_hi(val.z), bf_lo(val.w), bf_hi(val.w)};
          const float yr[8] = {bf_lo(yv.x), bf_hi(yv.x), bf_lo(yv.y), bf_hi(yv.y),
                               bf_lo(yv.z), bf_hi(yv.z), bf_lo(yv.w), bf_hi(yv.w)};
          if (interior && p.f.g2) {
            const uint4 gv = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.f.g2) + (ypix * d.Co + co) * 2);
            g[0] += bf_lo(gv.x); g[1] += bf_hi(gv.x); g[2] += bf_lo(gv.y); g[3] += bf_hi(gv.y);
            g[4] += bf_lo(gv.z); g[5] += bf_hi(gv.z); g[6] += bf_lo(gv.w); g[7] += bf_hi(gv.w);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float yh = (yr[k] - fmu[k]) * frs[k];
            const float gh = g[k] * act_grad_from_out(yh, p.f.act, p.f.slope);
            fa1[k] += gh;
            fa2[k] += gh * yh;
            fa3[k] += interior ? yh : 0.f;
          }
        }
        if (d.accumulate) {
          const uint4 old = *dst;
          val.x = pack_bf2(bf_lo(val.x) + bf_lo(old.x), bf_hi(val.x) + bf_hi(old.x));
          val.y = pack_bf2(bf_lo(val.y) + bf_lo(old.y), bf_hi(val.y) + bf_hi(old.y));
          val.z = pack_bf2(bf_lo(val.z) + bf_lo(old.z), bf_hi(val.z) + bf_hi(old.z));
          val.w = pack_bf2(bf_lo(val.w) + bf_lo(old.w), bf_hi(val.w) + bf_hi(old.w));
        }
        *dst = val;
      }
    }
    if (fuse) {
      // lanes sharing `sub` hold different pixels of the same 8 channels: butterfly over the pixel bits, then across
      // the pixel waves through LDS; one slot per pixel tile, no atomics (same contract as the statistics slots)
      float* red3 = reinterpret_cast<float*>(smem);   // [WM][BN][3]
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        // inside a 16-lane row through DPP rotations, across the 4 rows through the (LDS) permute
        if constexpr (LPR == 4) { fa1[k] = row_sum_stride4(fa1[k]); fa2[k] = row_sum_stride4(fa2[k]); fa3[k] = row_sum_stride4(fa3[k]); }
        else if constexpr (LPR == 8) { fa1[k] = row_sum_stride8(fa1[k]); fa2[k] = row_sum_stride8(fa2[k]); fa3[k] = row_sum_stride8(fa3[k]); }
        else {
#pragma unroll
          for (int o = LPR; o < 16; o <<= 1) {
            fa1[k] += __shfl_xor(fa1[k], o, 64); fa2[k] += __shfl_xor(fa2[k], o, 64); fa3[k] += __shfl_xor(fa3[k], o, 64);
          }
        }
#pragma unroll
        for (int o = (LPR > 16 ? LPR : 16); o < 64; o <<= 1) {
          fa1[k] += __shfl_xor(fa1[k], o, 64);
          fa2[k] += __shfl_xor(fa2[k], o, 64);
          fa3[k] += __shfl_xor(fa3[k], o, 64);
        }
      }
      if (prow == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int cl = wn * CW + sub * 8 + k;
          red3[(wm * BN + cl) * 3 + 0] = fa1[k];
          red3[(wm * BN + cl) * 3 + 1] = fa2[k];
          red3[(wm * BN + cl) * 3 + 2] = fa3[k];
        }
      }
      __syncthreads();
      if (tid < BN) {
        const int c = nt * BN + tid;
        if (c < d.Co) {
          float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int w = 0; w < WM; ++w) {
            t0 += red3[(w * BN + tid) * 3]; t1 += red3[(w * BN + tid) * 3 + 1]; t2 += red3[(w * BN + tid) * 3 + 2];
          }
          float* sp = p.f.partial + ((size_t)n * p.fuse_slots + mt) * 3 * d.Co;
          sp[c] = t0; sp[d.Co + c] = t1; sp[2 * d.Co + c] = t2;
        }
      }
    }
  }

}

// ---- split-K finalize: out = act(bias + sum_sp partial[sp]) as bf16, plus the partial statistics of the pixel tile ----
struct SplitFinK {
  const float* partial;
  const float* bias;
  char* out;
  float* stats;
  long long split_stride;
  int splits, tiles_m, bm;
  float rcp_wc, rcp_hc;
  // merged parity classes (gs_gconv_forward_multi_ws): class c's output parity and first statistics slot; n_cls = 0: d's own
  int n_cls;
  int cpz[GS_MULTI_MAX_CLS], cpy[GS_MULTI_MAX_CLS], cpx[GS_MULTI_MAX_CLS], cslot0[GS_MULTI_MAX_CLS];
  gs_gconv_desc d;
};

// Workgroup = (image, pixel tile, 16 output channels): 64 pixel lanes x 4 channel quads, so even a 2-pixel layer with 1024
// channels spreads its 32-way sum over 64 workgroups (the first version used 64-channel groups and 16 pixel lanes: 583 us
// for 128 pixels x 1024 channels x 32 splits on 32 workgroups).
__global__ __launch_bounds__(256) void gconv_splitk_finalize_kernel(const SplitFinK p) {
  const gs_gconv_desc& d = p.d;
  const int cgroups = (d.Co + 15) / 16;
  int b = blockIdx.x;
  const int cg = b % cgroups;
  b /= cgroups;
  const int ncls = p.n_cls > 0 ? p.n_cls : 1;
  const int ci = b % ncls;
  b /= ncls;
  const int cpz = p.n_cls > 0 ? p.cpz[ci] : d.pz, cpy = p.n_cls > 0 ? p.cpy[ci] : d.py, cpx = p.n_cls > 0 ? p.cpx[ci] : d.px;
  const int cslot0 = p.n_cls > 0 ? p.cslot0[ci] : d.stats_slot0;
  const int mt = b % p.tiles_m;
  const int n = b / p.tiles_m;
  const int tx = threadIdx.x & 3, py = threadIdx.x >> 2;     // channel quad, pixel lane (0..63)
  const int co = cg * 16 + tx * 4;
  const bool cv = co < d.Co;
  const int HWc = d.Dc * d.Hc * d.Wc;
  f32x4 bia = (cv && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int pl = py; pl < p.bm; pl += 64) {
    const int m = mt * p.bm + pl;
    if (m >= HWc || !cv) continue;
    const int zi = div_small(m, d.Wc, p.rcp_wc);
    const int jj = m - zi * d.Wc;
    const int zz = div_small(zi, d.Hc, p.rcp_hc);
    const int ii = zi - zz * d.Hc;
    const size_t opix = (((size_t)n * d.Do + (zz * d.so + cpz)) * d.Ho + (ii * d.so + cpy)) * d.Wo + (jj * d.so + cpx);
    f32x4 v = bia;
    const float* src = p.partial + opix * d.Co + co;
    int sp = 0;                                  // fixed order: reproducible; four splits' loads in flight
    for (; sp + 4 <= p.splits; sp += 4) {
      f32x4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(sp + u) * p.split_stride);
#pragma unroll
      for (int u = 0; u < 4; ++u) { v[0] += t[u][0]; v[1] += t[u][1]; v[2] += t[u][2]; v[3] += t[u][3]; }
    }
    for (; sp < p.splits; ++sp) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(src + (size_t)sp * p.split_stride);
      v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1[r] += v[r];
      s2[r] += v[r] * v[r];
      o[r] = apply_act(v[r], d.act, d.slope);
    }
    uint2 pk;
    pk.x = pack_bf2(o[0], o[1]);
    pk.y = pack_bf2(o[2], o[3]);
    uint2* dst = reinterpret_cast<uint2*>(p.out + (opix * d.out_cs + d.out_co + co) * 2);
    if (d.accumulate) {     // bf16 read-modify-write, the rounding points of gconv_kernel's own accumulate epilogue
      const uint2 old = *dst;
      pk.x = pack_bf2(bf_lo(pk.x) + bf_lo(old.x), bf_hi(pk.x) + bf_hi(old.x));
      pk.y = pack_bf2(bf_lo(pk.y) + bf_lo(old.y), bf_hi(pk.y) + bf_hi(old.y));
    }
    *dst = pk;
  }
  if (d.stats_slots > 0) {
    __shared__ float red[64][16][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) { red[py][tx * 4 + r][0] = s1[r]; red[py][tx * 4 + r][1] = s2[r]; }
    __syncthreads();
    if (threadIdx.x < 16) {
      const int c = cg * 16 + threadIdx.x;
      if (c < d.Co) {
        float a = 0.f, q = 0.f;
        for (int y = 0; y < 64; ++y) { a += red[y][threadIdx.x][0]; q += red[y][threadIdx.x][1]; }
        float* spt = p.stats + (((size_t)n * d.stats_slots + cslot0 + mt) * 2) * d.Co;
        spt[c] = a;
        spt[d.Co + c] = q;
      }
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------
namespace {
struct TileCfg { int bm, bn; int waves = 16; };
TileCfg pick_tile(const gs_gconv_desc* d) {
  if (d->Co <= 16) return {256, 16};
  if (d->Co <= 64) return {128, 64};
  // few K-steps: a tile's fixed cost (tables, ring fill, epilogue) is most of its time, and one 150-KB workgroup per CU pays it
  // in series. 128 x 128 tiles on 8 waves need 64 KB and <= 128 registers: two workgroups per CU, one's loop under the other's
  // prologue / epilogue (the 16-wave 128 x 128 tile holds 96 registers per lane: one workgroup per CU whatever its LDS)
  if ((d->Kp >> 6) <= gs_opt(GS_OPT_GCONV_SMALLK)) return {128, 128, 8};
  // big tile (8 waves, 3 stages, 1 workgroup per CU) once it still fills the chip; else the 4-wave 128x128 tile
  const long long pix = (long long)d->Dc * d->Hc * d->Wc;
  const long long big = (long long)d->N * ((pix + 255) / 256) * ((d->Co + 127) / 128);
  if (big >= gs_opt(GS_OPT_GCONV_BIG)) {
    // one workgroup per CU is resident: if the 256-pixel tiling needs a second, mostly empty round of workgroups
    // but 320-pixel tiles fit in one round, the larger tile wins (e.g. the 66x66 padded-domain data gradients)
    const long long big320 = (long long)d->N * ((pix + 319) / 320) * ((d->Co + 127) / 128);
    // ... and 288-pixel tiles (12 waves) when those fit too: 16 x 8 x 2 = 256 workgroups for the 66 x 66 domain at batch 8,
    // every CU busy with 10 % less work each than the 224 workgroups of the 320-pixel tiling
    const long long big288 = (long long)d->N * ((pix + 287) / 288) * ((d->Co + 127) / 128);
    if (big > 256 && big <= 512 && big288 <= 256 && gs_opt(GS_OPT_GCONV_TILE288)) return {288, 128};
    if (big > 256 && big <= 512 && big320 <= 256) return {320, 128};
    return {256, 128};
  }
  return {128, 128};
}
// Split-K plan: layers whose output tiles cannot fill the chip but whose K loop is long (U-Net bottleneck convs with
// 2..128 pixels and K = 16*1024, the PatchGAN 512->1 tail) are latency-bound on a handful of workgroups that each
// stream megabytes of weights through a 2-stage ring; splitting K spreads that stream over all CUs. Returns 1 = no split.
int splitk_plan(const gs_gconv_desc* d, const TileCfg& tc, bool fused) {
  const bool enabled = gs_opt(GS_OPT_SPLITK) != 0;
  if (!enabled || fused || (tc.bm != 128 && tc.bn != 16)) return 1;
  const long long pix = (long long)d->Dc * d->Hc * d->Wc;
  const long long blocks = (long long)d->N * ((pix + tc.bm - 1) / tc.bm) * ((d->Co + tc.bn - 1) / tc.bn);
  const int nk = d->Kp >> 6;
  const int max_blocks = gs_opt(GS_OPT_SPLITK_MAX_BLOCKS);
  if (blocks > max_blocks || nk < 16) return 1;
  const int target = gs_opt(GS_OPT_SPLITK_TARGET);
  long long splits = target / blocks;
  if (splits > nk / 4) splits = nk / 4;
  const long long out_floats = (long long)d->N * d->Do * d->Ho * d->Wo * d->Co;
  while (splits > 1 && splits * out_floats > (64LL << 20)) --splits;     // <= 256 MiB of partial sums
  return splits < 2 ? 1 : (int)splits;
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
int launch(const GConvK& k, int blocks, hipStream_t st) {
  const int lds = NSTAGE * (BM + BN) * 128 + GS_MAX_TAPS * 2 + BM * (k.nh + k.nw) * 2;
  GS_REQUIRE(lds <= 160 * 1024, "gs_gconv_forward: gather tables do not fit in LDS (T=%d)", k.d.T);
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv_kernel<BM, BN, WM, WN, NSTAGE>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    configured = true;
  }
  hipLaunchKernelGGL((gconv_kernel<BM, BN, WM, WN, NSTAGE>), dim3(blocks), dim3(WM * WN * 64), lds, st, k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
// wave counts per tile picked by measurement (tools/bench_kernels.py): more waves hide the LDS-DMA issue latency
int launch_tile(const TileCfg& tc, const GConvK& k, int blocks, hipStream_t st) {
  if (tc.bn == 16) return launch<256, 16, 8, 1, 2>(k, blocks, st);
  // A grid of at most two workgroups per CU with a long K loop is latency-bound on the 2-stage loop (one K-step in flight per
  // workgroup: the V-Net's 32^3 x 64-channel k5 convs ran 125 K-steps of 0.8 us for 0.05 us of MFMA work each): 4-stage ring
  int nk = k.d.Kp >> 6;
  for (int c = 0; c < k.n_cls; ++c) nk = std::min(nk, k.cls[c].Kp >> 6);
  const bool ring4 = gs_opt(GS_OPT_GCONV_RING4) != 0 && blocks <= 2 * 256 && nk >= gs_opt(GS_OPT_GCONV_RING4);
  if (tc.bn == 64) return ring4 ? launch<128, 64, 4, 2, 4>(k, blocks, st) : launch<128, 64, 4, 2, 2>(k, blocks, st);
  if (tc.bm == 128 && tc.waves == 8) return launch<128, 128, 2, 4, 2>(k, blocks, st);
  if (tc.bm == 128) return ring4 ? launch<128, 128, 4, 4, 4>(k, blocks, st) : launch<128, 128, 4, 4, 2>(k, blocks, st);
  if (tc.bm == 320) return launch<320, 128, 5, 2, 2>(k, blocks, st);
  if (tc.bm == 288) return launch<288, 128, 6, 2, 2>(k, blocks, st);
  if (tc.bm == 256) return launch<256, 128, 4, 4, 3>(k, blocks, st);   // 16 waves: best measured (8 and 4 lose)
  return launch<128, 128, 4, 4, 2>(k, blocks, st);
}
}  // namespace

// hconv.hip: halo-resident kernel for narrow stride-1 layers
int gs_hconv_slots(const gs_gconv_desc* d);
int gs_pwise_try(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out, void* stream,
                 int* handled);                   // (pwise.hip)
int gs_pwise_multi_try(const gs_gconv_desc* const* descs, int count, const void* in, const void* const* w_packs, const float* bias,
                       void* out, float* stats, void* stream, int* handled);
int gs_hconv_try(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out, float* stats,
                 void* stream, int* handled);
// hconvw.hip: halo-resident forward kernel for the wide 3x3 stride-1 layers
int gs_hconvw_slots(const gs_gconv_desc* d);
int gs_hconvw_try(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out, float* stats,
                  const gs_twin* tw, void* stream, int* handled);
int gs_hconvw_ring(const gs_gconv_desc* d, const void* in, const void* w_pack, void* out, const gs_gconv_fuse* fuse,
                   const gs_twin* tw, void* stream);
// hstrip.hip: W-folded k7 boundary convs (vertical taps, <= 64 channels) out of a resident input strip
int gs_hstrip_slots(const gs_gconv_desc* d);
int gs_hstrip_try(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out, float* stats,
                  void* stream, int* handled, const gs_twin* tw);
// hconvt.hip: the four parity classes of a stride-2 layer out of one halo-resident pass
int gs_hconvt_pattern(const gs_gconv_desc* const* descs, int count);
int gs_hconvt_launch(const gs_gconv_desc* const* descs, int pat, const void* in, const void* const* w_packs,
                     const float* bias, void* out, float* stats, const gs_gconv_fuse* fuse, void* stream,
                     const gs_twin* tw = nullptr);
// pconv.hip: persistent form of the 256 x 128 im2col tile for launches of several tiles per CU with a short K loop
bool gs_pconv_eligible(const GConvK& k, bool fused);
int gs_pconv_launch(const GConvK& k, bool fused, hipStream_t st);

extern "C" int gs_tile_m(const gs_gconv_desc* d) { return pick_tile(d).bm; }

extern "C" int gs_gconv_stat_slots(const gs_gconv_desc* d) {
  if (!d || d->Dc < 1 || d->Hc < 1 || d->Wc < 1) return 0;
  const int hs = gs_hconv_slots(d);
  if (hs) return hs;
  const int ws = gs_hconvw_slots(d);
  if (ws) return ws;
  const int ss = gs_hstrip_slots(d);
  if (ss) return ss;
  const long long pix = (long long)d->Dc * d->Hc * d->Wc;
  const int bm = pick_tile(d).bm;
  return (int)((pix + bm - 1) / bm);
}

static int gconv_forward_impl(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out,
                              float* stats, const gs_gconv_fuse* fuse, float* ws, int64_t ws_floats, void* stream,
                              const gs_twin* tw = nullptr);

extern "C" int gs_gconv_forward(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias,
                                void* out, float* stats, void* stream) {
  return gconv_forward_impl(d, in, w_pack, bias, out, stats, nullptr, nullptr, 0, stream);
}

// floats of workspace gs_gconv_forward_ws wants for this launch (0: the launch does not split K)
extern "C" int64_t gs_gconv_splitk_ws_floats(const gs_gconv_desc* d) {
  if (!d || d->Dc < 1 || d->Hc < 1 || d->Wc < 1 || d->Co < 1) return 0;
  if (gs_hconv_slots(d) || gs_hconvw_slots(d) || gs_hstrip_slots(d)) return 0;
  const TileCfg tc = pick_tile(d);
  const int splits = splitk_plan(d, tc, false);
  return splits > 1 ? (int64_t)splits * d->N * d->Do * d->Ho * d->Wo * d->Co : 0;
}

extern "C" int gs_gconv_forward_ws(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias,
                                   void* out, float* stats, float* ws, int64_t ws_floats, void* stream) {
  return gconv_forward_impl(d, in, w_pack, bias, out, stats, nullptr, ws, ws_floats, stream);
}

static int gconv_forward_fused_impl(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias,
                                    void* out, float* stats, const gs_gconv_fuse* fuse, void* stream, const gs_twin* tw);
extern "C" int gs_gconv_forward_fused(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias,
                                      void* out, float* stats, const gs_gconv_fuse* fuse, void* stream) {
  return gconv_forward_fused_impl(d, in, w_pack, bias, out, stats, fuse, stream, nullptr);
}

// ---- twin batches (gs_twin): which launches pick the weight set per image ------------------------------------------------
extern "C" int gs_gconv_twin_native(const gs_gconv_desc* d, const gs_gconv_fuse* fuse) {
  if (!d) return 0;
  if (fuse) {
    if (fuse->fold > 0 && d->Do == fuse->Dy && d->Ho == fuse->Hy && d->Wo == fuse->Wy) return gs_gconv_ring_slots(d) > 0;
    return gs_opt(GS_OPT_GCONV_TWIN) != 0;        // padded-domain fused launch: always the im2col kernel (no split-K there)
  }
  if (gs_hconv_slots(d)) return 0;
  if (gs_hconvw_slots(d) > 0) return 1;
  // the im2col kernel picks the weight set per tile (tiles never straddle images) — unless the halves would run split-K (few
  // tiles, long K: kept, the caller owns that workspace)
  if (d->accumulate) return 0;
  if (gs_hstrip_slots(d) > 0) return 1;           // hstrip.hip picks the weight set per tile / per persistent workgroup
  if (!gs_opt(GS_OPT_GCONV_TWIN)) return 0;
  gs_gconv_desc half = *d;
  half.N = d->N / 2;
  return splitk_plan(&half, pick_tile(&half), false) <= 1;
}
extern "C" int gs_gconv_forward_twin(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out,
                                     float* stats, const gs_gconv_fuse* fuse, const gs_twin* tw, void* stream) {
  GS_REQUIRE(d && tw && tw->n_split > 0 && tw->n_split < d->N, "gs_gconv_forward_twin: null argument / empty half");
  GS_REQUIRE(gs_gconv_twin_native(d, fuse), "gs_gconv_forward_twin: this layer's kernel has no twin form "
                                            "(gs_gconv_twin_native): run the two halves");
  if (fuse) return gconv_forward_fused_impl(d, in, w_pack, bias, out, stats, fuse, stream, tw);
  return gconv_forward_impl(d, in, w_pack, bias, out, stats, nullptr, nullptr, 0, stream, tw);
}

static int gconv_forward_fused_impl(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias,
                                    void* out, float* stats, const gs_gconv_fuse* fuse, void* stream, const gs_twin* tw) {
  GS_REQUIRE(d && fuse && fuse->y && fuse->mean_rstd && fuse->partial, "gs_gconv_forward_fused: null argument");
  // (si = 2: the data gradient of a transposed conv — a strided gather on the input side; the epilogue only sees output pixels)
  GS_REQUIRE(d->so == 1 && (d->si == 1 || d->si == 2) && !d->accumulate && d->stats_slots == 0 && d->act == GS_ACT_NONE &&
                 d->out_cs == d->Co && d->out_co == 0,
             "gs_gconv_forward_fused: only single-class data-gradient launches with a dense output can be fused");
  const int fd = fuse->Dy > 1 ? fuse->fold : 0;
  if (fuse->fold > 0 && d->Do == fuse->Dy && d->Ho == fuse->Hy && d->Wo == fuse->Wy) {
    // unpadded output domain: the launch applies the pad adjoint itself (hconvw.hip RING; gs_gconv_ring_slots says when)
    GS_REQUIRE(in && w_pack && out && !bias, "gs_gconv_forward_fused: null argument / bias on a data-gradient launch");
    return gs_hconvw_ring(d, in, w_pack, out, fuse, tw, stream);
  }
  GS_REQUIRE(d->Do == fuse->Dy + 2 * fd && d->Ho == fuse->Hy + 2 * fuse->fold && d->Wo == fuse->Wy + 2 * fuse->fold,
             "gs_gconv_forward_fused: output domain must be the norm's domain padded by `fold`");
  return gconv_forward_impl(d, in, w_pack, bias, out, stats, fuse, nullptr, 0, stream, tw);
}

static int gconv_forward_impl(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out,
                              float* stats, const gs_gconv_fuse* fuse, float* ws, int64_t ws_floats, void* stream,
                              const gs_twin* tw) {
  GS_REQUIRE(d && in && w_pack && out, "gs_gconv_forward: null argument");
  GS_REQUIRE(d->Ci >= 8 && (d->Ci & 7) == 0 && ((d->Ci >> 3) & ((d->Ci >> 3) - 1)) == 0,
             "gs_gconv_forward: Ci=%d must be 8*2^k", d->Ci);
  GS_REQUIRE((d->Co & 7) == 0 && d->Co > 0, "gs_gconv_forward: Co=%d must be a multiple of 8", d->Co);
  GS_REQUIRE(d->T >= 1 && d->T <= GS_MAX_TAPS, "gs_gconv_forward: T=%d out of range", d->T);
  GS_REQUIRE(d->Kp % 64 == 0 && d->Kp >= d->T * d->Ci, "gs_gconv_forward: bad Kp=%d", d->Kp);
  GS_REQUIRE((d->in_cs & 7) == 0 && (d->in_co & 7) == 0 && (d->out_cs & 7) == 0 && (d->out_co & 7) == 0,
             "gs_gconv_forward: channel strides/offsets must be multiples of 8 (16-B accesses)");
  GS_REQUIRE(d->Di >= 1 && d->Do >= 1 && d->Dc >= 1, "gs_gconv_forward: depths must be >= 1 (1 for 2-D tensors)");
  GS_REQUIRE((long long)d->Dc * d->Hc * d->Wc < (1 << 24) && (long long)d->Di * d->Hi < 32768 && d->Wi < 32768 &&
                 (long long)d->Di * d->Hi * d->Wi * d->in_cs * 2 < (1LL << 32),
             "gs_gconv_forward: class extent too large");
  GS_REQUIRE(d->stats_slots == 0 || stats, "gs_gconv_forward: stats requested without buffer");
  GS_REQUIRE(!d->accumulate || (d->stats_slots == 0 && d->act == GS_ACT_NONE && !bias),
             "gs_gconv_forward: accumulate excludes bias, activation and statistics");
  if (!fuse) {
    int handled = 0;
    if (!tw) {
      if (int rc = gs_pwise_try(d, in, w_pack, bias, out, stream, &handled)) return rc;
      if (handled) return 0;
      if (int rc = gs_hconv_try(d, in, w_pack, bias, out, stats, stream, &handled)) return rc;
      if (handled) return 0;
    }
    if (int rc = gs_hconvw_try(d, in, w_pack, bias, out, stats, tw, stream, &handled)) return rc;
    if (handled) return 0;
    if (int rc = gs_hstrip_try(d, in, w_pack, bias, out, stats, stream, &handled, tw)) return rc;
    if (handled) return 0;
  }
  GS_REQUIRE(!tw || !ws, "gs_gconv_forward_twin: the split-K launch has no twin form");
  const TileCfg tc = pick_tile(d);
  GConvK k;
  if (fuse) k.f = *fuse; else k.f = gs_gconv_fuse{};
  k.fuse_slots = (int)(((long long)d->Dc * d->Hc * d->Wc + tc.bm - 1) / tc.bm);
  k.in = static_cast<const char*>(in);
  k.w = static_cast<const char*>(w_pack);
  k.bias = bias;
  k.out = static_cast<char*>(out);
  k.stats = stats;
  k.zero = static_cast<const char*>(gs_zero_page());
  GS_REQUIRE(k.zero, "gs_gconv_forward: library not initialised (call gs_init)");
  k.tiles_m = (d->Dc * d->Hc * d->Wc + tc.bm - 1) / tc.bm;
  k.tiles_n = (d->Co + tc.bn - 1) / tc.bn;
  int sh = 0;
  while ((8 << sh) < d->Ci) ++sh;
  k.ci_shift = sh;
  k.rcp_wc = 1.0f / (float)d->Wc;
  k.rcp_hc = 1.0f / (float)d->Hc;
  k.d = *d;
  k.nh = k.nw = 0;
  for (int t = 0; t < d->T; ++t) {
    int h = 0, w = 0;
    while (h < k.nh && (k.uh[h] != d->dh[t] || k.ud[h] != d->dd[t])) ++h;
    if (h == k.nh) {
      GS_REQUIRE(k.nh < 64, "gs_gconv_forward: more than 64 distinct tap (depth,row) pairs");
      k.ud[k.nh] = d->dd[t];
      k.uh[k.nh++] = d->dh[t];
    }
    while (w < k.nw && k.uw[w] != d->dw[t]) ++w;
    if (w == k.nw) { GS_REQUIRE(k.nw < 16, "gs_gconv_forward: more than 16 distinct tap columns"); k.uw[k.nw++] = d->dw[t]; }
    k.tap_h[t] = (unsigned char)h;
    k.tap_w[t] = (unsigned char)w;
  }
  k.n_cls = 0;
  k.nsplit = tw ? tw->n_split : 0x7fffffff;
  k.w_delta = tw ? tw->w_delta : 0;
  k.bias_delta = tw ? tw->bias_delta : 0;
  k.splits = 1;
  k.partial = nullptr;
  k.split_stride = (long long)d->N * d->Do * d->Ho * d->Wo * d->Co;
  if (ws) {
    const int splits = splitk_plan(d, tc, fuse != nullptr);
    if (splits > 1 && ws_floats >= (int64_t)splits * k.split_stride) { k.splits = splits; k.partial = ws; }
  }
  const long long blocks = (long long)d->N * k.tiles_m * k.tiles_n * k.splits;
  GS_REQUIRE(blocks > 0 && blocks < (1LL << 31), "gs_gconv_forward: bad grid %lld", blocks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (k.splits > 1) {
    int rc = tc.bn == 16 ? launch<256, 16, 8, 1, 2>(k, (int)blocks, st)
           : tc.bn == 64 ? (gs_opt(GS_OPT_SPLITK_RING) ? launch<128, 64, 4, 2, 4>(k, (int)blocks, st) : launch<128, 64, 4, 2, 2>(k, (int)blocks, st))
           : gs_opt(GS_OPT_SPLITK_RING) ? launch<128, 128, 4, 4, 4>(k, (int)blocks, st) : launch<128, 128, 4, 4, 2>(k, (int)blocks, st);
    if (rc) return rc;
    SplitFinK f;
    f.partial = ws; f.bias = bias; f.out = static_cast<char*>(out); f.stats = stats;
    f.split_stride = k.split_stride; f.splits = k.splits; f.tiles_m = k.tiles_m; f.bm = tc.bm;
    f.rcp_wc = k.rcp_wc; f.rcp_hc = k.rcp_hc; f.d = *d; f.n_cls = 0;
    const int fblocks = d->N * k.tiles_m * ((d->Co + 15) / 16);
    hipLaunchKernelGGL(gconv_splitk_finalize_kernel, dim3(fblocks), dim3(256), 0, st, f);
    GS_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (tc.bm == 256 && tc.bn == 128 && gs_pconv_eligible(k, fuse != nullptr)) return gs_pconv_launch(k, fuse != nullptr, st);
  return launch_tile(tc, k, (int)blocks, st);
}

// ---- the fused form for the parity classes of a stride-2 data gradient (hconvt.hip) -------------------------------------------
// slots per image the fused launch writes (one per 16 x 16 box of the class grid), 0 when the layer does not run there
extern "C" int gs_gconv_multi_fused_slots(const gs_gconv_desc* const* descs, int32_t count) {
  if (!descs || count < 1 || !descs[0]) return 0;
  const gs_gconv_desc* d = descs[0];
  if (d->stats_slots != 0 || d->act != GS_ACT_NONE || d->out_cs != d->Co || d->out_co != 0) return 0;
  return gs_hconvt_pattern(descs, count) >= 0 ? (d->Hc / 16) * (d->Wc / 16) : 0;
}

extern "C" int gs_gconv_forward_multi_fused(const gs_gconv_desc* const* descs, int32_t count, const void* in,
                                            const void* const* w_packs, void* out, const gs_gconv_fuse* fuse, void* stream) {
  GS_REQUIRE(descs && w_packs && in && out && fuse && fuse->y && fuse->mean_rstd && fuse->partial,
             "gs_gconv_forward_multi_fused: null argument");
  GS_REQUIRE(gs_gconv_multi_fused_slots(descs, count) > 0,
             "gs_gconv_forward_multi_fused: the layer does not run on the halo-resident class kernel (gs_gconv_multi_fused_slots)");
  const gs_gconv_desc* d = descs[0];
  GS_REQUIRE(fuse->fold == 0 && fuse->Dy <= 1 && fuse->Hy == d->Ho && fuse->Wy == d->Wo,
             "gs_gconv_forward_multi_fused: the consumer's tensor must be the unpadded output domain");
  return gs_hconvt_launch(descs, gs_hconvt_pattern(descs, count), in, w_packs, nullptr, out, nullptr, fuse, stream);
}

// Twin batches of a multi-class layer (gs_twin): the halo-resident class kernel (hconvt.hip) picks the packs per box.
// gs_gconv_multi_twin_native: 1 when the layer — descs carry the whole batch of both networks — runs there;
// gs_gconv_forward_multi_twin is that launch (fuse == NULL: gs_gconv_forward_multi, else gs_gconv_forward_multi_fused).
extern "C" int gs_gconv_multi_twin_native(const gs_gconv_desc* const* descs, int32_t count) {
  if (!descs || count < 1 || !descs[0] || (descs[0]->N & 1)) return 0;
  return gs_hconvt_pattern(descs, count) >= 0;
}
extern "C" int gs_gconv_forward_multi_twin(const gs_gconv_desc* const* descs, int32_t count, const void* in,
                                           const void* const* w_packs, const float* bias, void* out, float* stats,
                                           const gs_gconv_fuse* fuse, const gs_twin* tw, void* stream) {
  GS_REQUIRE(descs && w_packs && in && out && tw && count >= 1 && descs[0], "gs_gconv_forward_multi_twin: null argument");
  GS_REQUIRE(2 * tw->n_split == descs[0]->N, "gs_gconv_forward_multi_twin: the two networks take the same number of images");
  const int pat = gs_hconvt_pattern(descs, count);
  GS_REQUIRE(pat >= 0, "gs_gconv_forward_multi_twin: this layer has no twin form (gs_gconv_multi_twin_native): run the halves");
  if (fuse) {
    GS_REQUIRE(fuse->y && fuse->mean_rstd && fuse->partial && !bias && !stats && gs_gconv_multi_fused_slots(descs, count) > 0,
               "gs_gconv_forward_multi_twin: bad fused launch (contract of gs_gconv_forward_multi_fused)");
    const gs_gconv_desc* d = descs[0];
    GS_REQUIRE(fuse->fold == 0 && fuse->Dy <= 1 && fuse->Hy == d->Ho && fuse->Wy == d->Wo,
               "gs_gconv_forward_multi_twin: the consumer's tensor must be the unpadded output domain");
  }
  return gs_hconvt_launch(descs, pat, in, w_packs, bias, out, stats, fuse, stream, tw);
}

// ---- merged launch over the output-parity classes of one layer ---------------------------------------------------------
// A stride-2 transposed conv (and the data gradient of a stride-2 conv) is one class per output parity: 4 launches in 2-D
// (k3: 1/2/2/4 taps, k4: 4 each), 8 in 3-D, each of them a quarter of the layer's pixels — at batch 8 the 32x32 / 64x64
// PatchGAN gradients are 64-256 workgroups of a few K-steps per launch, paid four times with a split-K pass behind each
// (profiles/r02_conv_table.txt: 100-250 TFLOP/s). Merged, the classes are one grid: fixed costs once, 4x the workgroups.
namespace {
// classes of one shape (even extents), short tap lists, at least two of them: they can share a grid
bool multi_mergeable(const gs_gconv_desc* const* descs, int count) {
  const gs_gconv_desc* d0 = descs[0];
  bool merge = count >= 2 && count <= GS_MULTI_MAX_CLS && gs_opt(GS_OPT_GCONV_MULTI) != 0;
  for (int c = 0; c < count && merge; ++c) {
    const gs_gconv_desc* d = descs[c];
    merge = d && d->T >= 1 && d->T <= GS_MULTI_MAX_TAPS && !d->accumulate && d->N == d0->N && d->Hi == d0->Hi &&
            d->Wi == d0->Wi && d->Ci == d0->Ci && d->Di == d0->Di && d->Do == d0->Do && d->Dc == d0->Dc &&
            d->in_cs == d0->in_cs && d->in_co == d0->in_co && d->Ho == d0->Ho && d->Wo == d0->Wo && d->Co == d0->Co &&
            d->out_cs == d0->out_cs && d->out_co == d0->out_co && d->Hc == d0->Hc && d->Wc == d0->Wc && d->so == d0->so &&
            d->si == d0->si && d->w_rows == d0->w_rows && d->border == d0->border && d->act == d0->act &&
            d->slope == d0->slope && d->stats_slots == d0->stats_slots && !gs_hconv_slots(d) && !gs_hconvw_slots(d);
  }
  return merge;
}
// Split-K plan of the merged launch: the U-Net's bottleneck transposed convs / data gradients are four classes of 1-128
// pixels with K = 4 x 1024..2048 each; one class at a time they are four launches of a few K-steps per workgroup plus four
// finalize passes, and each launch pays its fixed cost (tables, first DMA latency, drain) for 8 MB of weights. Merged, the
// classes fill the chip with a quarter of the splits each. Returns 1 = no split.
int splitk_plan_multi(const gs_gconv_desc* const* descs, int count, const TileCfg& tc) {
  const gs_gconv_desc* d = descs[0];
  if (!gs_opt(GS_OPT_SPLITK) || !gs_opt(GS_OPT_SPLITK_MULTI) || (tc.bm != 128 && tc.bn != 16)) return 1;
  const long long pix = (long long)d->Dc * d->Hc * d->Wc;
  const long long blocks = (long long)count * d->N * ((pix + tc.bm - 1) / tc.bm) * ((d->Co + tc.bn - 1) / tc.bn);
  int nk = descs[0]->Kp >> 6;
  for (int c = 1; c < count; ++c) nk = std::min(nk, descs[c]->Kp >> 6);
  if (blocks > gs_opt(GS_OPT_SPLITK_MAX_BLOCKS) || nk < 16) return 1;
  long long splits = gs_opt(GS_OPT_SPLITK_TARGET) / blocks;
  if (splits > nk / 4) splits = nk / 4;
  const long long out_floats = (long long)d->N * d->Do * d->Ho * d->Wo * d->Co;
  while (splits > 1 && splits * out_floats > (64LL << 20)) --splits;     // <= 256 MiB of partial sums
  return splits < 2 ? 1 : (int)splits;
}
int gconv_forward_multi_impl(const gs_gconv_desc* const* descs, int32_t count, const void* in, const void* const* w_packs,
                             const float* bias, void* out, float* stats, float* ws, int64_t ws_floats, void* stream);
}  // namespace

extern "C" int gs_gconv_forward_multi(const gs_gconv_desc* const* descs, int32_t count, const void* in,
                                      const void* const* w_packs, const float* bias, void* out, float* stats,
                                      void* stream) {
  return gconv_forward_multi_impl(descs, count, in, w_packs, bias, out, stats, nullptr, 0, stream);
}

// floats of workspace gs_gconv_forward_multi_ws wants for the merged launch of these classes (0: it would not split K —
// the classes do not merge, the halo-resident class kernel takes them, or the merged grid is large enough as it is)
extern "C" int64_t gs_gconv_multi_splitk_ws_floats(const gs_gconv_desc* const* descs, int32_t count) {
  if (!descs || count < 2 || !descs[0]) return 0;
  for (int c = 0; c < count; ++c)
    if (!descs[c] || descs[c]->Dc < 1 || descs[c]->Hc < 1 || descs[c]->Wc < 1 || descs[c]->Co < 1) return 0;
  if (!multi_mergeable(descs, count) || gs_hconvt_pattern(descs, count) >= 0) return 0;
  const gs_gconv_desc* d = descs[0];
  const int splits = splitk_plan_multi(descs, count, pick_tile(d));
  return splits > 1 ? (int64_t)splits * d->N * d->Do * d->Ho * d->Wo * d->Co : 0;
}

// gs_gconv_forward_multi with a split-K workspace: the merged grid is (classes x tiles x splits), the partial sums of all
// classes share one dense [split][output pixel][Co] buffer (the classes partition the output pixels) and ONE finalize pass
// applies bias / activation / statistics for every class. ws == NULL or too small: gs_gconv_forward_multi.
extern "C" int gs_gconv_forward_multi_ws(const gs_gconv_desc* const* descs, int32_t count, const void* in,
                                         const void* const* w_packs, const float* bias, void* out, float* stats,
                                         float* ws, int64_t ws_floats, void* stream) {
  return gconv_forward_multi_impl(descs, count, in, w_packs, bias, out, stats, ws, ws_floats, stream);
}

namespace {
int gconv_forward_multi_impl(const gs_gconv_desc* const* descs, int32_t count, const void* in, const void* const* w_packs,
                             const float* bias, void* out, float* stats, float* ws, int64_t ws_floats, void* stream) {
  GS_REQUIRE(descs && w_packs && count >= 1 && in && out, "gs_gconv_forward_multi: null argument");
  for (int c = 0; c < count; ++c) GS_REQUIRE(descs[c] && w_packs[c], "gs_gconv_forward_multi: null class %d", c);
  if (!multi_mergeable(descs, count)) {      // classes of different shapes (odd extents), long tap lists, a single class: one launch each
    for (int c = 0; c < count; ++c)
      if (int rc = gconv_forward_impl(descs[c], in, w_packs[c], bias, out, stats, nullptr, ws, ws_floats, stream)) return rc;
    return 0;
  }
  // validate through the single-class checks, then build the shared arguments from class 0
  const gs_gconv_desc* d = descs[0];
  GS_REQUIRE(d->Ci >= 8 && (d->Ci & 7) == 0 && ((d->Ci >> 3) & ((d->Ci >> 3) - 1)) == 0,
             "gs_gconv_forward_multi: Ci=%d must be 8*2^k", d->Ci);
  GS_REQUIRE((d->Co & 7) == 0 && d->Co > 0, "gs_gconv_forward_multi: Co=%d must be a multiple of 8", d->Co);
  GS_REQUIRE((d->in_cs & 7) == 0 && (d->in_co & 7) == 0 && (d->out_cs & 7) == 0 && (d->out_co & 7) == 0,
             "gs_gconv_forward_multi: channel strides/offsets must be multiples of 8 (16-B accesses)");
  GS_REQUIRE(d->Di >= 1 && d->Do >= 1 && d->Dc >= 1, "gs_gconv_forward_multi: depths must be >= 1");
  GS_REQUIRE((long long)d->Dc * d->Hc * d->Wc < (1 << 24) && (long long)d->Di * d->Hi < 32768 && d->Wi < 32768 &&
                 (long long)d->Di * d->Hi * d->Wi * d->in_cs * 2 < (1LL << 32),
             "gs_gconv_forward_multi: class extent too large");
  GS_REQUIRE(d->stats_slots == 0 || stats, "gs_gconv_forward_multi: stats requested without buffer");
  for (int c = 0; c < count; ++c)
    GS_REQUIRE(descs[c]->Kp % 64 == 0 && descs[c]->Kp >= descs[c]->T * descs[c]->Ci, "gs_gconv_forward_multi: bad Kp=%d",
               descs[c]->Kp);
  {                                                       // k2 stride-2 volume layers: 8 one-tap classes (pwise.hip)
    int handled = 0;
    if (int rc = gs_pwise_multi_try(descs, count, in, w_packs, bias, out, stats, stream, &handled)) return rc;
    if (handled) return 0;
  }
  {
    const int pat = gs_hconvt_pattern(descs, count);      // 2-D k3 / k4 stride-2 layers with 64-multiple channels: one pass
    if (pat >= 0) return gs_hconvt_launch(descs, pat, in, w_packs, bias, out, stats, nullptr, stream);
  }
  // the tile one class alone would get: the statistics slots (gs_gconv_stat_slots) are counted per class from it
  const TileCfg tc = pick_tile(d);
  static GConvK k;      // ~3 KB: filled per call, passed by value to the launch
  k.f = gs_gconv_fuse{};
  k.fuse_slots = 0;
  k.in = static_cast<const char*>(in);
  k.w = static_cast<const char*>(w_packs[0]);
  k.bias = bias;
  k.out = static_cast<char*>(out);
  k.stats = stats;
  k.zero = static_cast<const char*>(gs_zero_page());
  GS_REQUIRE(k.zero, "gs_gconv_forward_multi: library not initialised (call gs_init)");
  k.tiles_m = (d->Dc * d->Hc * d->Wc + tc.bm - 1) / tc.bm;
  k.tiles_n = (d->Co + tc.bn - 1) / tc.bn;
  int sh = 0;
  while ((8 << sh) < d->Ci) ++sh;
  k.ci_shift = sh;
  k.rcp_wc = 1.0f / (float)d->Wc;
  k.rcp_hc = 1.0f / (float)d->Hc;
  k.d = *d;
  k.nh = k.nw = 0;
  k.nsplit = 0x7fffffff;     // (no twin form of the merged class launch)
  k.w_delta = k.bias_delta = 0;
  k.n_cls = count;
  for (int c = 0; c < count; ++c) {
    const gs_gconv_desc* dc = descs[c];
    GConvCls& cl = k.cls[c];
    GS_REQUIRE(dc->Kp % 64 == 0 && dc->Kp >= dc->T * dc->Ci, "gs_gconv_forward_multi: bad Kp=%d", dc->Kp);
    cl.w_off = static_cast<const char*>(w_packs[c]) - k.w;
    cl.T = dc->T; cl.Kp = dc->Kp; cl.pz = dc->pz; cl.py = dc->py; cl.px = dc->px; cl.stats_slot0 = dc->stats_slot0;
    cl.nh = cl.nw = 0;
    for (int t = 0; t < dc->T; ++t) {
      int h = 0, w = 0;
      while (h < cl.nh && (cl.uh[h] != dc->dh[t] || cl.ud[h] != dc->dd[t])) ++h;
      if (h == cl.nh) { cl.ud[cl.nh] = dc->dd[t]; cl.uh[cl.nh++] = dc->dh[t]; }
      while (w < cl.nw && cl.uw[w] != dc->dw[t]) ++w;
      if (w == cl.nw) cl.uw[cl.nw++] = dc->dw[t];
      cl.tap_h[t] = (unsigned char)h;
      cl.tap_w[t] = (unsigned char)w;
    }
    if (cl.nh > k.nh) k.nh = cl.nh;         // LDS gather tables are sized for the largest class
    if (cl.nw > k.nw) k.nw = cl.nw;
  }
  k.splits = 1;
  k.partial = nullptr;
  k.split_stride = (long long)d->N * d->Do * d->Ho * d->Wo * d->Co;
  if (ws) {
    const int splits = splitk_plan_multi(descs, count, tc);
    if (splits > 1 && ws_floats >= (int64_t)splits * k.split_stride) { k.splits = splits; k.partial = ws; }
  }
  const long long blocks = (long long)count * d->N * k.tiles_m * k.tiles_n * k.splits;
  GS_REQUIRE(blocks > 0 && blocks < (1LL << 31), "gs_gconv_forward_multi: bad grid %lld", blocks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (k.splits > 1) {
    int rc = tc.bn == 16 ? launch<256, 16, 8, 1, 2>(k, (int)blocks, st)
           : tc.bn == 64 ? (gs_opt(GS_OPT_SPLITK_RING) ? launch<128, 64, 4, 2, 4>(k, (int)blocks, st) : launch<128, 64, 4, 2, 2>(k, (int)blocks, st))
           : gs_opt(GS_OPT_SPLITK_RING) ? launch<128, 128, 4, 4, 4>(k, (int)blocks, st) : launch<128, 128, 4, 4, 2>(k, (int)blocks, st);
    if (rc) return rc;
    SplitFinK f;
    f.partial = ws; f.bias = bias; f.out = static_cast<char*>(out); f.stats = stats;
    f.split_stride = k.split_stride; f.splits = k.splits; f.tiles_m = k.tiles_m; f.bm = tc.bm;
    f.rcp_wc = k.rcp_wc; f.rcp_hc = k.rcp_hc; f.d = *d; f.n_cls = count;
    for (int c = 0; c < count; ++c) {
      f.cpz[c] = descs[c]->pz; f.cpy[c] = descs[c]->py; f.cpx[c] = descs[c]->px; f.cslot0[c] = descs[c]->stats_slot0;
    }
    const int fblocks = count * d->N * k.tiles_m * ((d->Co + 15) / 16);
    hipLaunchKernelGGL(gconv_splitk_finalize_kernel, dim3(fblocks), dim3(256), 0, st, f);
    GS_CHECK_HIP(hipGetLastError());
    return 0;
  }
  return launch_tile(tc, k, (int)blocks, st);
}
}  // namespace
