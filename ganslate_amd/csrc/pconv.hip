// Persistent form of gconv_kernel<256, 128, 16 waves, 3 stages> for launches with SEVERAL tiles per CU and a short K loop:
// the stride-2 convs of the generators and their transposed-conv data gradients (resnet2d.py:35,52-57: 9 / 18 K-steps), the
// discriminators' k4 layers (patchgan2d.py:36-62: 16 .. 64 K-steps).
//
// Measured on the one-tile-per-workgroup kernel (tools/probe/e2_ksweep.sh, 512 tiles = two rounds of 256 workgroups, one
// 150-KB workgroup per CU): T = 21.7 us + 2.19 us per K-step — with 9 K-steps more than half of a launch is fixed cost paid
// once per ROUND by every CU at the same moment: the first stage's memory latency, the bias / statistics / slab epilogue, a
// 16.8-MB burst of output stores, the workgroup turnover. Here workgroup w walks tiles w, w + G, w + 2G, ...: the K-step
// sequence simply continues across tiles — the 3-stage ring is fed with the NEXT tile's first two K-steps while the current
// tile's last two compute (its gather tables were built a tile ahead in the other table buffer) — the epilogue works out of
// the one ring stage that is free at a tile boundary (two passes of 32 pixels per wave through 40 KiB), and a tile's output
// stores are issued and never waited for: the first two K-steps of the next tile wait with counted vmcnt (VMEM retires in
// issue order; every wave issues the SAME number of store instructions — masked lanes write to a dump line — so the counts
// are exact), and they drain under that tile's loop.
// Same arithmetic, same summation order, same statistics-slot contract as gconv_kernel<256,128,4,4,3>: results are
// bit-identical (tests/test_ops_gpu.py::test_persistent_im2col_kernel).
#include "gconv.hpp"
#include <cstdlib>

namespace {
constexpr int BM = 256, BN = 128, WM = 4, WN = 4, NW = 16;
constexpr int WT = BN * 128, XT = BM * 128, STAGE = WT + XT;      // 16 + 32 KiB per ring stage
constexpr int TI = BN / WN / 16, TJ = BM / WM / 16;               // 2 x 4 fragments per wave
constexpr int NXI = BM / 8 / NW;                                  // 2 pixel-tile DMA instructions per wave per stage
constexpr int LOADS = NXI + 1;                                    // + one weight instruction
constexpr unsigned TAB_BAD = 0x8000u;
constexpr int CW = BN / WN, PW = BM / WM;                         // 32 channels x 64 pixels per wave
constexpr int SROW = CW * 2 + 16;                                 // slab row: 32 channels + a pad piece
constexpr int PH = 32;                                            // pixels per wave per epilogue pass
constexpr int RED_BYTES = WM * BN * 3 * 4;                        // [WM][BN][2 | 3] floats in front of the slabs
static_assert(RED_BYTES + NW * PH * SROW + 2 * BN * 4 <= STAGE, "epilogue scratch must fit one ring stage");

struct PConvK {
  GConvK g;
  int ntiles;                  // tiles of the launch = N * tiles_m * tiles_n
  char* dump;                  // 256 writable bytes: destination of masked store lanes
};

__device__ __forceinline__ void lds_write64(unsigned addr, uint2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ float lds_read32(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_write32(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// four 16-byte registers by compile-time index (an indexed array of them ends up in scratch memory, i.e. VMEM)
struct Quad {
  uint4 a, b, c, d;
  template <class Tag> __device__ __forceinline__ uint4& at(Tag) {
    constexpr int u = Tag::value;
    if constexpr (u == 0) return a; else if constexpr (u == 1) return b; else if constexpr (u == 2) return c; else return d;
  }
};

template <bool FUSE>
__global__ __launch_bounds__(1024) void pconv_kernel(const PConvK pk) {
  const GConvK& p = pk.g;
  const gs_gconv_desc& d = p.d;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  short* taps = reinterpret_cast<short*>(smem + 3 * STAGE);
  unsigned short* tab0 = reinterpret_cast<unsigned short*>(smem + 3 * STAGE + GS_MAX_TAPS * 2);
  const int tabsz = BM * (p.nh + p.nw);                            // entries of one table buffer (two of them)
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;

  // workgroup -> first tile (XCD-aware: every XCD owns a contiguous run of a round's tiles), then every G-th tile
  const int G = gridDim.x;
  int w0;
  {
    const int q = G >> 3, r = G & 7, xcd = blockIdx.x & 7;
    w0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int ntl = (pk.ntiles - w0 + G - 1) / G;                    // tiles of this workgroup (>= 1: G <= ntiles)
  const int nt = w0 % p.tiles_n;                                   // the launcher makes G a multiple of tiles_n: one channel tile

  for (int t = tid; t < d.T; t += NW * 64) taps[t] = (short)((int)p.tap_h[t] | ((int)p.tap_w[t] << 8));
  // the distinct tap offsets, out of the kernel arguments into LDS once: indexed dynamically they are global loads, and a
  // global load next to the DMA stream (the tables of a later tile are built inside the loop) is waited for with vmcnt(0)
  int* const toff = reinterpret_cast<int*>(tab0 + 2 * tabsz) + 2 * BN;          // [nh] (depth << 8 | row & 0xff) | [nw] column
  if (tid < p.nh) toff[tid] = ((int)p.ud[tid] << 8) | ((int)p.uh[tid] & 0xff);
  else if (tid < p.nh + p.nw) toff[tid] = (int)p.uw[tid - p.nh];

  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;
  const int HWc = d.Dc * d.Hc * d.Wc;
  auto build_tables = [&](int L, int buf) {                        // gather tables of logical tile L (gconv.hip)
    const int mt = (L / p.tiles_n) % p.tiles_m;
    unsigned short* tab = tab0 + buf * tabsz;
    for (int e = tid; e < tabsz; e += NW * 64) {
      const int k = e / BM, row = e - k * BM;
      const int m = mt * BM + row;
      bool ok = m < HWc;
      const int zi = div_small(m, d.Wc, p.rcp_wc);
      const int jj = m - zi * d.Wc;
      const int zz = div_small(zi, d.Hc, p.rcp_hc);
      const int ii = zi - zz * d.Hc;
      unsigned v;
      const int to = toff[k];
      if (k < p.nh) {
        const int iz = border_index(zz * d.si + (to >> 8), d.Di, d.border, ok);
        v = (unsigned)(iz * d.Hi + border_index(ii * d.si + (int)(signed char)(to & 0xff), d.Hi, d.border, ok));
      } else {
        v = (unsigned)border_index(jj * d.si + to, d.Wi, d.border, ok);
      }
      tab[e] = (unsigned short)(ok ? v : TAB_BAD);
    }
  };

  // the bias of this channel tile for both networks of a twin batch, into LDS BEFORE the first LDS-DMA (an ordinary load
  // whose result is used while a DMA is in flight makes the compiler wait vmcnt(0) there; in registers it costs 16 of the 128)
  float* const bias_lds = reinterpret_cast<float*>(tab0 + 2 * tabsz);          // [2 networks][BN]
  if (!FUSE && tid < 2 * BN) {
    const int net = tid / BN, c = nt * BN + tid % BN;
    const bool bv = p.bias && c < d.Co;
    const float* bsrc = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.bias) +
                                                       ((net && p.nsplit < d.N) ? p.bias_delta : 0));
    bias_lds[tid] = bv ? bsrc[c] : 0.f;
  }
  // ---- the issue stream: K-steps of tile 0, tile 1, ... in order, two ahead of the compute stream ----------------------
  const int cmask = (1 << p.ci_shift) - 1;
  const bool chunk_major = d.Ci >= 64;
  const int nk = d.Kp >> 6;
  int is_it = 0, is_ks = 0, it_t = 0, it_c = 0;
  const char* is_in = nullptr;
  const char* is_w = nullptr;
  int is_winc = 0;
  const unsigned short* is_tab = tab0;
  auto set_issue_tile = [&](int it) {
    const int L = w0 + it * G;
    const int n = L / (p.tiles_n * p.tiles_m);
    const char* cw = p.w + (n >= p.nsplit ? p.w_delta : 0);
    is_in = p.in + ((size_t)n * d.Di * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
    const int co = nt * BN + wave * 8 + lrow;
    const bool wv = co < d.w_rows;
    is_w = wv ? cw + ((size_t)co * d.Kp + chunk * 8) * 2 : p.zero;
    is_winc = wv ? 128 : 0;
    is_tab = tab0 + (it & 1) * tabsz;
    it_t = it_c = 0;
  };
  auto issue_next = [&](int buf) {
    int q0;
    if (chunk_major) {
      q0 = (it_t << p.ci_shift) + it_c * 8;
      if (++it_t == d.T) { it_t = 0; ++it_c; }
    } else {
      q0 = is_ks * 8;
    }
    const int q = q0 + chunk;
    const int t = q >> p.ci_shift;
    const int c8 = q & cmask;
    const bool tv = t < d.T;
    const int tp = taps[tv ? t : 0];
    const unsigned short* hrow = is_tab + (tp & 0xff) * BM + wave * 8 + lrow;
    const unsigned short* wrow = is_tab + (p.nh + (tp >> 8)) * BM + wave * 8 + lrow;
    const unsigned wi = (unsigned)d.Wi, cs2 = (unsigned)d.in_cs * 2u;
    char* sb = smem + buf * STAGE;
    glds16(is_w + q0 * (is_winc >> 3), sb + wave * 1024);
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const unsigned a = hrow[NW * 8 * i], bq = wrow[NW * 8 * i];
      const bool ok = tv && !((a | bq) & TAB_BAD);
      unsigned off = (a * wi + bq) * cs2 + (unsigned)(c8 * 16);
      asm volatile("" : "+v"(off));
      glds16(ok ? is_in + off : p.zero, sb + WT + (wave + NW * i) * 1024);
    }
    if (++is_ks == nk) {
      is_ks = 0;
      if (++is_it < ntl) set_issue_tile(is_it);
    }
  };

  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 15, fk = lane >> 4, swz = lane & 7;
  f32x4 acc[TI][TJ];
  auto compute = [&](int cur) {
    const char* wb = smem + cur * STAGE + (wn * CW + frow) * 128;
    const char* xb = smem + cur * STAGE + WT + (wm * PW + frow) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((kk * 4 + fk) ^ swz) << 4;
      bf16x8 wf[TI], xf[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(wb + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < TJ; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xb + j * 16 * 128 + coff);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  };

  __syncthreads();                                                 // tap offsets
  build_tables(w0, 0);
  __syncthreads();                                                 // taps + tables of tile 0 (nothing in flight yet)
  set_issue_tile(0);
  issue_next(0);
  issue_next(1);                                                   // (nk >= 3: both belong to tile 0)

  const bool want_stats = !FUSE && d.stats_slots > 0;
  int stage = 0;
#pragma clang loop unroll(disable)
  for (int it = 0; it < ntl; ++it) {
    const int L = w0 + it * G;
    const int mt = (L / p.tiles_n) % p.tiles_m;
    const int n = L / (p.tiles_n * p.tiles_m);
    const bool last_tile = it + 1 == ntl;
    // tables of the next tile into the other buffer: its first K-step is issued at this tile's K-step nk - 2 (>= one barrier
    // away), and the buffer's previous user — the tile before this one — was last read when this tile's step 1 was issued
    if (!last_tile) build_tables(L + G, (it + 1) & 1);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma clang loop unroll(disable)
    for (int ks = 0; ks < nk; ++ks) {
      // this wave's share of the current K-step has landed; what may still fly: the next K-step (LOADS instructions) and, in
      // the first two K-steps behind a tile boundary, the previous tile's store instructions (4 outputs + 2 statistics / 3 sums)
      if (last_tile && ks + 1 == nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (it > 0 && ks < 2) {
        if (FUSE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS + 7) : "memory");
        else if (want_stats) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS + 6) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS + 4) : "memory");
      } else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
      __builtin_amdgcn_s_barrier();
      if (is_it < ntl) issue_next(stage == 0 ? 2 : stage - 1);     // K-step + 2 into the stage of K-step - 1
      compute(stage);
      stage = stage == 2 ? 0 : stage + 1;
    }
    // ---- epilogue out of the stage the last K-step just left (the other two hold the next tile's first K-steps) -------------
    lds_barrier();                                                 // every wave has read its last fragments
    char* const ebase = smem + (stage == 0 ? 2 : stage - 1) * STAGE;
    float* const red = reinterpret_cast<float*>(ebase);
    char* const slab = ebase + RED_BYTES + wave * (PH * SROW);
    [[maybe_unused]] float* const mrs = reinterpret_cast<float*>(ebase + RED_BYTES + NW * PH * SROW);      // [2][BN] mean | rstd
    if constexpr (FUSE) {
      // mean / rstd of this tile's 128 channels: 64 threads fetch 4 floats each, everybody reads them back from LDS (as 16
      // registers per lane they pushed the loop's pointers into scratch memory)
      if (tid < 2 * BN / 4) {
        const int c = nt * BN + (tid % (BN / 4)) * 4;
        const float* mr = p.f.mean_rstd + (size_t)n * 2 * d.Co + (tid >= BN / 4 ? d.Co : 0) + c;
        *reinterpret_cast<f32x4*>(mrs + tid * 4) = c < d.Co ? *reinterpret_cast<const f32x4*>(mr) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    constexpr int LPR = CW / 8, PPI = 64 / LPR;                    // 4 lanes per pixel, 16 pixels per store instruction
    const int sub = lane % LPR, prow = lane / LPR;
    const int co8 = nt * BN + wn * CW + sub * 8;
    const bool second = n >= p.nsplit;
    uint4 val0, val1, val2, val3;                  // (named: an indexed array ended up in scratch memory, i.e. VMEM)
    auto val_of = [&](auto u_tag) -> uint4& {
      constexpr int u = decltype(u_tag)::value;
      if constexpr (u == 0) return val0; else if constexpr (u == 1) return val1; else if constexpr (u == 2) return val2; else return val3;
    };
    f32x4 bia[TI];
    if constexpr (!FUSE) {
#pragma unroll
      for (int i = 0; i < TI; ++i) bia[i] = *reinterpret_cast<const f32x4*>(bias_lds + (second ? BN : 0) + wn * CW + i * 16 + fk * 4);
    }
    float s1[TI][4], s2[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
    static_for<0, 2>([&](auto ph_tag) {
      constexpr int ph = decltype(ph_tag)::value;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = ph * 2 + jj;
        const bool pv = mt * BM + wm * PW + j * 16 + frow < HWc;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (FUSE) {
              v[r] = acc[i][j][r];
            } else {
              v[r] = acc[i][j][r] + bia[i][r];
              if (pv) { s1[i][r] += v[r]; s2[i][r] += v[r] * v[r]; }
              v[r] = apply_act_small(v[r], d.act, d.slope);
            }
          }
          uint2 o;
          o.x = pack_bf2(v[0], v[1]);
          o.y = pack_bf2(v[2], v[3]);
          // (inline asm: in front of a compiler-visible LDS write the compiler waits vmcnt(0) — the next tile's first two
          // K-steps are in flight into the other stages)
          lds_write64(lds_addr(slab + (jj * 16 + frow) * SROW + (i * 16 + fk * 4) * 2), o);
        }
      }
      __builtin_amdgcn_wave_barrier();                             // wave-private slab: a wave's LDS operations complete in order
      {
        bf16x8 r0, r1;
        const unsigned ra = lds_addr(slab + prow * SROW + sub * 16);
        lds_read128<0>(r0, ra);
        lds_read128<PPI * SROW>(r1, ra);
        gs_lgkm_wait_only<0>();
        reg_fence(r0); reg_fence(r1);
        val_of(std::integral_constant<int, ph * 2>{}) = __builtin_bit_cast(uint4, r0);
        val_of(std::integral_constant<int, ph * 2 + 1>{}) = __builtin_bit_cast(uint4, r1);
      }
      __builtin_amdgcn_wave_barrier();
    });
    // output pixel of store u: pixel u * 16 + prow of the wave's 64
    bool ov[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ov[u] = mt * BM + wm * PW + u * PPI + prow < HWc && co8 < d.Co;
    if constexpr (FUSE) {
      // first pass of the consumer's InstanceNorm backward (gs_gconv_forward_fused, gconv.hip): sums over this tile of
      // ghat = (g + g2) * act'(yhat), ghat * yhat, yhat, with yhat taken at the pixel the padding folds this one onto
      float fa1[8], fa2[8], fa3[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) fa1[k] = fa2[k] = fa3[k] = 0.f;
      lds_barrier();                                               // mrs published
      // (one pixel at a time, as gconv_kernel does: two pixels' operands in flight cost 32 spilled registers under the 128 cap)
      static_for<0, 4>([&](auto u_tag) {
        constexpr int u = decltype(u_tag)::value;
        const uint4 vu = val_of(u_tag);
        if (ov[u]) {
          const int m = mt * BM + wm * PW + u * PPI + prow;
          const int zi = div_small(m, d.Wc, p.rcp_wc);
          const int jj = m - zi * d.Wc;
          const int zz = div_small(zi, d.Hc, p.rcp_hc);
          const int ii = zi - zz * d.Hc;
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
          const uint4 yv = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.f.y) + (ypix * d.Co + co8) * 2);
          float g[8] = {bf_lo(vu.x), bf_hi(vu.x), bf_lo(vu.y), bf_hi(vu.y), bf_lo(vu.z), bf_hi(vu.z), bf_lo(vu.w), bf_hi(vu.w)};
          const float yr[8] = {bf_lo(yv.x), bf_hi(yv.x), bf_lo(yv.y), bf_hi(yv.y), bf_lo(yv.z), bf_hi(yv.z), bf_lo(yv.w), bf_hi(yv.w)};
          if (interior && p.f.g2) {
            const uint4 gv = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.f.g2) + (ypix * d.Co + co8) * 2);
            g[0] += bf_lo(gv.x); g[1] += bf_hi(gv.x); g[2] += bf_lo(gv.y); g[3] += bf_hi(gv.y);
            g[4] += bf_lo(gv.z); g[5] += bf_hi(gv.z); g[6] += bf_lo(gv.w); g[7] += bf_hi(gv.w);
          }
          const int cl8 = wn * CW + sub * 8;
          const f32x4 mua = *reinterpret_cast<const f32x4*>(mrs + cl8), mub = *reinterpret_cast<const f32x4*>(mrs + cl8 + 4);
          const f32x4 rsa = *reinterpret_cast<const f32x4*>(mrs + BN + cl8), rsb = *reinterpret_cast<const f32x4*>(mrs + BN + cl8 + 4);
          const float fmu[8] = {mua[0], mua[1], mua[2], mua[3], mub[0], mub[1], mub[2], mub[3]};
          const float frs[8] = {rsa[0], rsa[1], rsa[2], rsa[3], rsb[0], rsb[1], rsb[2], rsb[3]};
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float yh = (yr[k] - fmu[k]) * frs[k];
            const float gh = g[k] * act_grad_from_out(yh, p.f.act, p.f.slope);
            fa1[k] += gh;
            fa2[k] += gh * yh;
            fa3[k] += interior ? yh : 0.f;
          }
        }
      });
      // lanes sharing `sub` hold different pixels of the same 8 channels: inside a 16-lane row through DPP rotations, across
      // the 4 rows through the permute, across the pixel waves through LDS — the order of gconv_kernel's fused epilogue
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        fa1[k] = row_sum_stride4(fa1[k]); fa2[k] = row_sum_stride4(fa2[k]); fa3[k] = row_sum_stride4(fa3[k]);
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
          fa1[k] += __shfl_xor(fa1[k], o, 64);
          fa2[k] += __shfl_xor(fa2[k], o, 64);
          fa3[k] += __shfl_xor(fa3[k], o, 64);
        }
      }
      if (prow == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int cl = wn * CW + sub * 8 + k;
          lds_write32(lds_addr(red + (wm * BN + cl) * 3 + 0), fa1[k]);
          lds_write32(lds_addr(red + (wm * BN + cl) * 3 + 1), fa2[k]);
          lds_write32(lds_addr(red + (wm * BN + cl) * 3 + 2), fa3[k]);
        }
      }
      lds_barrier();
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
      const int c = nt * BN + tid;
      const bool sv = tid < BN && c < d.Co;
      if (tid < BN) {
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          t0 += lds_read32(lds_addr(red + (w * BN + tid) * 3)); t1 += lds_read32(lds_addr(red + (w * BN + tid) * 3 + 1));
          t2 += lds_read32(lds_addr(red + (w * BN + tid) * 3 + 2));
        }
      }
      // (every wave issues the three stores: the counted waits of the next tile's first K-steps rely on it)
      float* sp = sv ? p.f.partial + ((size_t)n * p.fuse_slots + mt) * 3 * d.Co + c : reinterpret_cast<float*>(pk.dump);
      const int cs = sv ? d.Co : 0;
      asm volatile("" ::: "memory");
      sp[0] = t0; sp[cs] = t1; sp[2 * cs] = t2;
    } else if (want_stats) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row16_sum(s1[i][r]), q = row16_sum(s2[i][r]);
          if (frow == 0) {
            const int cl = wn * CW + i * 16 + fk * 4 + r;
            lds_write32(lds_addr(red + (wm * BN + cl) * 2 + 0), a);
            lds_write32(lds_addr(red + (wm * BN + cl) * 2 + 1), q);
          }
        }
      lds_barrier();
      float a = 0.f, q = 0.f;
      const int c = nt * BN + tid;
      const bool sv = tid < BN && c < d.Co;
      if (tid < BN) {
#pragma unroll
        for (int w = 0; w < WM; ++w) { a += lds_read32(lds_addr(red + (w * BN + tid) * 2)); q += lds_read32(lds_addr(red + (w * BN + tid) * 2 + 1)); }
      }
      float* sp = sv ? p.stats + (((size_t)n * d.stats_slots + d.stats_slot0 + mt) * 2) * d.Co + c : reinterpret_cast<float*>(pk.dump);
      const int cs = sv ? d.Co : 0;
      asm volatile("" ::: "memory");
      sp[0] = a; sp[cs] = q;
    }
    // the tile's four output stores: never waited for (masked lanes write the dump line, so every wave issues four)
    asm volatile("" ::: "memory");
    static_for<0, 4>([&](auto u_tag) {
      constexpr int u = decltype(u_tag)::value;
      const int m = ov[u] ? mt * BM + wm * PW + u * PPI + prow : 0;
      const int zi = div_small(m, d.Wc, p.rcp_wc);
      const int jj = m - zi * d.Wc;
      const int zz = div_small(zi, d.Hc, p.rcp_hc);
      const int ii = zi - zz * d.Hc;
      const size_t opix = (((size_t)n * d.Do + (zz * d.so + d.pz)) * d.Ho + (ii * d.so + d.py)) * d.Wo + (jj * d.so + d.px);
      char* dst = ov[u] ? p.out + (opix * d.out_cs + d.out_co + co8) * 2 : pk.dump;
      *reinterpret_cast<uint4*>(dst) = val_of(u_tag);
    });
    asm volatile("" ::: "memory");
  }
}

int g_cus = 0;
int cus() {
  if (!g_cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    g_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                ? prop.multiProcessorCount : 256;
  }
  return g_cus;
}
}  // namespace

void* gs_dump_page();

// Does this launch — a single-class 256 x 128 im2col launch without split-K — run on the persistent kernel? (k.tiles_m / tiles_n /
// nh / nw filled in by the caller)
bool gs_pconv_eligible(const GConvK& k, bool fused) {
  const gs_gconv_desc& d = k.d;
  const int opt = gs_opt(GS_OPT_GCONV_PERSIST);
  if (!opt || k.n_cls > 0 || k.splits > 1 || d.accumulate) return false;
  const int nk = d.Kp >> 6;
  if (nk < 4 || nk > opt) return false;                            // long K loops amortise their fixed cost themselves
  const long long ntiles = (long long)d.N * k.tiles_m * k.tiles_n;
  if (ntiles <= cus() || k.tiles_n > cus() || ntiles >= (1LL << 30)) return false;
  const int lds = 3 * STAGE + GS_MAX_TAPS * 2 + 2 * BM * (k.nh + k.nw) * 2 + 2 * BN * 4 + 80 * 4;
  if (lds > 160 * 1024) return false;
  if (fused && d.act != GS_ACT_NONE) return false;
  return true;
}

int gs_pconv_launch(const GConvK& k, bool fused, hipStream_t st) {
  PConvK pk;
  pk.g = k;
  pk.ntiles = k.d.N * k.tiles_m * k.tiles_n;
  pk.dump = static_cast<char*>(gs_dump_page());
  GS_REQUIRE(pk.dump, "gs_gconv_forward: library not initialised (call gs_init)");
  int G = cus() / k.tiles_n * k.tiles_n;                           // one channel tile per workgroup
  if (G > pk.ntiles) G = pk.ntiles;
  const int lds = 3 * STAGE + GS_MAX_TAPS * 2 + 2 * BM * (k.nh + k.nw) * 2 + 2 * BN * 4 + 80 * 4;
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pconv_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pconv_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    configured = true;
  }
  if (fused) hipLaunchKernelGGL((pconv_kernel<true>), dim3(G), dim3(1024), lds, st, pk);
  else hipLaunchKernelGGL((pconv_kernel<false>), dim3(G), dim3(1024), lds, st, pk);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
