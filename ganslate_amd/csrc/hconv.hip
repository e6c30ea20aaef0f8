// Halo-resident convolution for narrow layers (stride 1, <= 64 channels on both sides, many taps): the k5 convs of
// Vnet3D's additive couplings / input / output blocks (ganslate/nn/generators/vnet/vnet3d.py:161,252,262-267) and the
// W-folded k7 boundary convs of the ResNets.
//
// The im2col-style gather of gconv.hip re-reads every input voxel once per tap from L2 (125x for a 5x5x5 kernel): with
// 16-32 channels those layers are L2-bandwidth bound at a few percent of the matrix peak. Here a workgroup owns a box
// of 256 output voxels (4x8x8, or 1x16x16 for images), stages the input box + halo for 16 channels in LDS ONCE
// (LDS-DMA, border handling in the per-lane source address) and runs all taps out of it: the MFMA B operand of a tap is
// a ds_read_b128 at (voxel + tap offset); the weights stream through a double-buffered LDS stage of 8 K-steps, fetched
// one stage ahead. HBM/L2 traffic drops from taps x input to ~1.7 x input; the loop is LDS-read bound.
//   D[co][voxel] += W[co][(tap, ci)] * X[voxel + off(tap)][ci],  v_mfma_f32_16x16x32_bf16, fp32 accumulate.
// Same epilogue contract as gconv_kernel: bias, per-workgroup InstanceNorm partial sums (one slot per box), activation,
// optional accumulate-into (additive-coupling gradient joins), channel-slice output views.
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

struct HConvK {
  const char* in;
  const char* w;
  const float* bias;
  char* out;
  float* stats;
  const char* zero;
  int BD, BH, BW;        // output box
  int HD, HH, HW;        // halo box = output box + tap range
  int dmin, hmin, wmin;  // smallest tap offset per axis
  int nbd, nbh, nbw;     // boxes per axis
  int chunks;            // Ci / CC
  int cc_shift;          // log2(CC)
  gs_gconv_desc d;
};

// LDS map: [tap offsets GS_MAX_TAPS*4][stats scratch 2 KiB][2 weight stages][halo]
//   weight stage: TI*16 rows (output channels) x 8 K-steps x 32 k, row pitch 33 pieces of 16 B (528 B: the pad piece
//   spreads the 16 rows of a fragment read over all banks); filled by LDS-DMA, lane q -> (row q/33, piece q%33)
template <int TI, int CC, int NW = 4>
__global__ __launch_bounds__(NW * 64) void hconv_kernel(const HConvK p) {
  constexpr int PP = CC / 8;                      // 16-B pieces per voxel
  constexpr int SK = 8;                           // K-steps per weight stage
  constexpr int WROWS = TI * 16;
  constexpr int WPIECES = WROWS * 33;
  constexpr int WINSTR = (WPIECES + 63) / 64;     // LDS-DMA instructions per stage
  constexpr int WSTAGE = WINSTR * 1024;           // bytes per stage
  constexpr int WPW = (WINSTR + NW - 1) / NW;     // instructions per wave per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* toff = reinterpret_cast<int*>(smem);                        // [GS_MAX_TAPS] halo-linear tap offsets
  float* red = reinterpret_cast<float*>(smem + GS_MAX_TAPS * 4);   // [NW waves][64 channels][2]
  char* wst = smem + GS_MAX_TAPS * 4 + NW * 64 * 2 * 4;
  char* halo = wst + 2 * WSTAGE;
  const gs_gconv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int co0 = blockIdx.y * WROWS;             // output-channel group of this workgroup
  int b = blockIdx.x;
  const int bx = b % p.nbw; b /= p.nbw;
  const int by = b % p.nbh; b /= p.nbh;
  const int bz = b % p.nbd;
  const int n = b / p.nbd;
  const int box = (bz * p.nbh + by) * p.nbw + bx;
  const int oz0 = bz * p.BD, oy0 = by * p.BH, ox0 = bx * p.BW;

  // tap table entry: halo-linear voxel offset << 8 | column offset << 1 | parity of the row offset (the last two feed the
  // bank swizzle of the fragment reads, see swz below)
  for (int t = tid; t < d.T; t += NW * 64)
    toff[t] = ((((int)d.dd[t] - p.dmin) * p.HH + ((int)d.dh[t] - p.hmin)) * p.HW + ((int)d.dw[t] - p.wmin)) << 8 |
              ((int)d.dw[t] - p.wmin) << 1 | (((int)d.dh[t] - p.hmin) & 1);

  const int row = lane & 15, kg = lane >> 4;
  // Bank swizzle (16-channel chunks: a voxel is two 16-B halves, a lane reads one of them). The 16 lanes of a fragment read
  // are two runs of 8 consecutive voxels — two box rows (8-wide volume boxes) or one row of 16 (image boxes) — and the two
  // runs start 384 B or 256 B apart: the same 4-bank groups, a 2-way conflict on 4 of the 5 reads of every K-step (the loop
  // is LDS-bound). A voxel therefore stores its halves swapped when swz(voxel) = 1, with swz = row parity (volume boxes) or
  // bit 3 of the column (image boxes): the two runs always differ in it, so they land on complementary bank groups.
  const bool rows8 = p.BW == 8;
  const int lxl = row & (p.BW - 1), lyp = (row >> 3) & 1;
  int pbase[4];
  bool pval[4];
  size_t opix[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pl = wave * 64 + j * 16 + row;
    const int lz = pl / (p.BH * p.BW), rem = pl - lz * (p.BH * p.BW);
    const int ly = rem / p.BW, lx = rem - ly * p.BW;
    pbase[j] = (lz * p.HH + ly) * p.HW + lx;
    const int oz = oz0 + lz, oy = oy0 + ly, ox = ox0 + lx;
    pval[j] = oz < d.Do && oy < d.Ho && ox < d.Wo;
    opix[j] = (((size_t)n * d.Do + oz) * d.Ho + oy) * d.Wo + ox;
  }

  f32x4 acc[TI][4];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int HV = p.HD * p.HH * p.HW;
  const int pieces = HV * PP;
  const int hhw = p.HH * p.HW;
  const char* in_n = p.in + ((size_t)n * d.Di * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
  const int nsteps = (d.T * CC + 31) >> 5;
  const int nstages = (nsteps + SK - 1) / SK;

  // weight stage g of channel chunk `chunk` -> LDS buffer `buf` (every wave issues WPW instructions)
  auto issue_w = [&](int g, int chunk, int buf) {
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      const int inst = wave * WPW + i;               // wave-uniform
      if (inst < WINSTR) {
        const int q = inst * 64 + lane;
        const int r = q / 33, pc = q - r * 33;
        const int kk = g * (SK * 32) + pc * 8;
        const int tap = kk >> p.cc_shift, c0 = kk & (CC - 1);
        const bool ok = pc < 32 && r < WROWS && co0 + r < d.w_rows && tap < d.T;
        unsigned off = ((unsigned)(co0 + r) * (unsigned)d.Kp + (unsigned)(tap * d.Ci + chunk * CC + c0)) * 2u;
        asm volatile("" : "+v"(off));
        const char* src = ok ? p.w + off : p.zero;
        glds16(src, wst + buf * WSTAGE + inst * 1024);
      }
    }
  };

  for (int chunk = 0; chunk < p.chunks; ++chunk) {
    __syncthreads();   // tap table visible / the previous chunk's reads are done
    // ---- stage the input box + halo of this channel chunk: one 16-B piece per lane per LDS-DMA instruction ----
    for (int q0 = wave * 64; q0 < pieces; q0 += NW * 64) {
      const int q = q0 + lane;
      const int v = q / PP, part = q - v * PP;
      const int hz = v / hhw, r2 = v - hz * hhw;
      const int hy = r2 / p.HW, hx = r2 - hy * p.HW;
      bool ok = q < pieces;
      // boxes hanging over the image edge ask for positions even a reflection cannot map: clamp (those voxels only
      // feed masked output pixels)
      int iz = border_index(oz0 + hz + p.dmin, d.Di, d.border, ok);
      int iy = border_index(oy0 + hy + p.hmin, d.Hi, d.border, ok);
      int ix = border_index(ox0 + hx + p.wmin, d.Wi, d.border, ok);
      iz = min(max(iz, 0), d.Di - 1);
      iy = min(max(iy, 0), d.Hi - 1);
      ix = min(max(ix, 0), d.Wi - 1);
      const int spart = CC == 16 ? part ^ ((rows8 ? hy : (hx >> 3)) & 1) : part;     // swizzled half (see swz above)
      unsigned off = ((unsigned)((iz * d.Hi + iy) * d.Wi + ix) * (unsigned)d.in_cs + (unsigned)(chunk * CC + spart * 8)) * 2u;
      asm volatile("" : "+v"(off));
      const char* src = ok ? in_n + off : p.zero;
      glds16(src, halo + (size_t)q0 * 16);
    }
    issue_w(0, chunk, 0);

    // ---- all taps out of LDS; weights arrive one stage (8 K-steps of 32) ahead -------------------------------
    for (int g = 0; g < nstages; ++g) {
      const int buf = g & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage g (and, for g == 0, the halo) landed
      __syncthreads();                                   // ... for every wave; stage g-1 fully consumed
      if (g + 1 < nstages) issue_w(g + 1, chunk, buf ^ 1);
      const char* wb = wst + buf * WSTAGE;
      // tap offsets of the stage's 8 K-steps first, then a hand-pipelined loop: the 4 + TI fragment reads of step
      // u+1 are issued before the 4*TI MFMAs of step u (left to itself the compiler emits read-wait-MFMA per fragment)
      int vo[SK];                                        // byte offset of this lane's piece relative to its voxel's
#pragma unroll
      for (int u = 0; u < SK; ++u) {
        const int tap = ((g * SK + u) * 32 + kg * 8) >> p.cc_shift;
        const int e = toff[tap < d.T ? tap : 0];         // past the last tap the weights are zero: any finite B
        if (CC == 16) {
          const int bit = rows8 ? ((e ^ lyp) & 1) : (((lxl + ((e >> 1) & 127)) >> 3) & 1);
          vo[u] = (e >> 8) * 32 + (((kg & 1) ^ bit) << 4);
        } else {
          vo[u] = (e >> 8) * 16;
        }
      }
      // Reads through inline asm with counted waits (common.hpp): with the LDS-DMA in this loop hipcc waits lgkmcnt(0) in front
      // of every MFMA group and every read group — read, wait, MFMA — whatever the source order says.
      const unsigned wba = lds_addr(wb) + (unsigned)(row * 528 + kg * 16);
      unsigned xa[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xa[j] = lds_addr(halo) + (unsigned)(pbase[j] * (CC * 2));
      bf16x8 wA[TI], xA[4], wB[TI], xB[4];
      auto load_frags = [&](auto u_tag, bf16x8 (&wf)[TI], bf16x8 (&xf)[4]) {
        constexpr int u = decltype(u_tag)::value;
        static_for<0, TI>([&](auto i_tag) {
          constexpr int i = decltype(i_tag)::value;
          lds_read128<i * 16 * 528 + u * 64>(wf[i], wba);
        });
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_read128<0>(xf[j], xa[j] + (unsigned)vo[u]);
      };
      auto wait_mma = [&](auto pending_tag, bf16x8 (&wf)[TI], bf16x8 (&xf)[4]) {
        constexpr int N = decltype(pending_tag)::value;
        if constexpr (TI == 1) gs_lgkm_wait<N>(wf[0], xf[0], xf[1], xf[2], xf[3]);
        else gs_lgkm_wait<N>(wf[0], wf[1], xf[0], xf[1], xf[2], xf[3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      load_frags(std::integral_constant<int, 0>{}, wA, xA);
      static_for<0, SK / 2>([&](auto h_tag) {
        constexpr int u = 2 * decltype(h_tag)::value;
        load_frags(std::integral_constant<int, u + 1>{}, wB, xB);
        wait_mma(std::integral_constant<int, TI + 4>{}, wA, xA);
        if constexpr (u + 2 < SK) {
          load_frags(std::integral_constant<int, u + 2>{}, wA, xA);
          wait_mma(std::integral_constant<int, TI + 4>{}, wB, xB);
        } else {
          wait_mma(std::integral_constant<int, 0>{}, wB, xB);
        }
      });
    }
  }

  // ---- epilogue: bias, partial statistics (one slot per box), activation, [accumulate], 8-B NHWC stores ----------
  const bool want_stats = d.stats_slots > 0;
  float s1[TI][4], s2[TI][4];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int co = co0 + i * 16 + kg * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] + ((p.bias && co < d.Co) ? p.bias[co + r] : 0.f);
        if (pval[j]) { s1[i][r] += v[r]; s2[i][r] += v[r] * v[r]; }
        v[r] = apply_act_small(v[r], d.act, d.slope);
      }
      if (pval[j] && co < d.Co) {
        uint2* dst = reinterpret_cast<uint2*>(p.out + (opix[j] * d.out_cs + d.out_co + co) * 2);
        uint2 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        if (d.accumulate) {   // bf16 read-modify-write, same rounding points as gconv_kernel
          const uint2 old = *dst;
          o.x = pack_bf2(bf_lo(o.x) + bf_lo(old.x), bf_hi(o.x) + bf_hi(old.x));
          o.y = pack_bf2(bf_lo(o.y) + bf_lo(old.y), bf_hi(o.y) + bf_hi(old.y));
        }
        *dst = o;
      }
    }
  }
  if (want_stats) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[i][r], q = s2[i][r];
        a = row16_sum(a);
        q = row16_sum(q);
        if (row == 0) {
          const int c = i * 16 + kg * 4 + r;
          red[(wave * 64 + c) * 2 + 0] = a;
          red[(wave * 64 + c) * 2 + 1] = q;
        }
      }
    __syncthreads();
    if (tid < WROWS && co0 + tid < d.Co) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { a += red[(w * 64 + tid) * 2]; q += red[(w * 64 + tid) * 2 + 1]; }
      float* sp = p.stats + (((size_t)n * d.stats_slots + d.stats_slot0 + box) * 2) * d.Co;
      sp[co0 + tid] = a;
      sp[d.Co + co0 + tid] = q;
    }
  }
}


// ---- hconv_kernel's volume form rebuilt as persistent workgroups (round 6) -------------------------------------------------------
// hconv_kernel<2, 16, 4> ran the V-Net's 32 -> 32 channel k5 convs (5.1 + 4.3 ms of the brats step) one 256-voxel box per
// workgroup: every box decoded its nine pieces per thread from scratch (divisions, three border rules each), staged its halo in
// front of its own tap loop (two workgroups per CU hid part of that), and re-streamed the layer's 128 KB of weights per channel
// chunk. Here: 8 waves on 8 x 8 x 8 boxes (half the weight stream per voxel), workgroups walk boxes b, b + grid, ...; the pieces a
// thread stages are decoded once, border-resolved source planes / rows / columns come from three small LDS tables, and the halo
// of the NEXT (box, chunk) unit is staged into a second buffer under the current unit's tap loop. The weight ring and the tap
// loop (pipelined inline-asm fragment reads) are hconv_kernel's; the ring simply runs on across units. Epilogue LDS traffic goes
// through inline asm (an LDS access the compiler sees would wait for the staging in flight). Same arithmetic per output element.
template <int TI>
__global__ __launch_bounds__(512) void hconv2_kernel(const HConvK p) {
  constexpr int NW = 8, CC = 16, PP = 2, SK = 8;
  constexpr int WROWS = TI * 16;
  constexpr int WPIECES = WROWS * 33;
  constexpr int WINSTR = (WPIECES + 63) / 64;     // LDS-DMA instructions per stage
  constexpr int WSTAGE = WINSTR * 1024;           // bytes per stage
  constexpr int WPW = (WINSTR + NW - 1) / NW;     // instructions per wave per stage
  constexpr int NHP = 7;                          // halo instructions per wave per unit (uniform: surplus ones fill a sink)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* toff = reinterpret_cast<int*>(smem);                        // [GS_MAX_TAPS]
  float* red = reinterpret_cast<float*>(smem + GS_MAX_TAPS * 4);   // [NW waves][64 channels][2]
  char* wst = smem + GS_MAX_TAPS * 4 + NW * 64 * 2 * 4;            // 2 stages
  const gs_gconv_desc& d = p.d;
  const int HV = p.HD * p.HH * p.HW, hhw = p.HH * p.HW;
  const int pieces = HV * PP;
  const int hbytes = (pieces * 16 + 1023) / 1024 * 1024;
  char* halo0 = wst + 2 * WSTAGE;                 // 2 halo buffers, then a 1-KiB sink, then the border tables
  char* sink = halo0 + 2 * (size_t)hbytes;
  unsigned short* ztab = reinterpret_cast<unsigned short*>(sink + 1024);
  unsigned short* ytab = ztab + p.nbd * p.HD;
  unsigned short* xtab = ytab + p.nbh * p.HH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int co0 = blockIdx.y * WROWS;             // output-channel group of this workgroup

  for (int t = tid; t < d.T; t += NW * 64)
    toff[t] = ((((int)d.dd[t] - p.dmin) * p.HH + ((int)d.dh[t] - p.hmin)) * p.HW + ((int)d.dw[t] - p.wmin)) << 8 |
              ((int)d.dw[t] - p.wmin) << 1 | (((int)d.dh[t] - p.hmin) & 1);
  const int ntab = p.nbd * p.HD + p.nbh * p.HH + p.nbw * p.HW;
  for (int e = tid; e < ntab; e += NW * 64) {      // border-resolved source index of every halo plane / row / column of every box
    bool ok = true;                                // (boxes hanging over the volume's edge ask for positions even a reflection cannot
    int v;                                         //  map: clamped — those voxels only feed masked output voxels)
    if (e < p.nbd * p.HD) {
      const int bz = e / p.HD, hz = e - bz * p.HD;
      v = border_index(bz * p.BD + hz + p.dmin, d.Di, d.border, ok);
      v = min(max(v, 0), d.Di - 1);
    } else if (e < p.nbd * p.HD + p.nbh * p.HH) {
      const int e2 = e - p.nbd * p.HD;
      const int by = e2 / p.HH, hy = e2 - by * p.HH;
      v = border_index(by * p.BH + hy + p.hmin, d.Hi, d.border, ok);
      v = min(max(v, 0), d.Hi - 1);
    } else {
      const int e2 = e - p.nbd * p.HD - p.nbh * p.HH;
      const int bx = e2 / p.HW, hx = e2 - bx * p.HW;
      v = border_index(bx * p.BW + hx + p.wmin, d.Wi, d.border, ok);
      v = min(max(v, 0), d.Wi - 1);
    }
    ztab[e] = ok ? (unsigned short)v : (unsigned short)0x8000;
  }

  const int row = lane & 15, kg = lane >> 4;
  const int lyp = (row >> 3) & 1;                 // (8-wide box rows: see hconv_kernel's bank swizzle)
  int pbase[4], pvox[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pl = wave * 64 + j * 16 + row;
    const int lz = pl / (p.BH * p.BW), rem = pl - lz * (p.BH * p.BW);
    const int ly = rem / p.BW, lx = rem - ly * p.BW;
    pbase[j] = (lz * p.HH + ly) * p.HW + lx;
    pvox[j] = lz << 16 | ly << 8 | lx;
  }
  // the halo pieces this thread stages of every unit, decoded once: table indices + the swizzled 16-B half
  int h_zyx[NHP], h_sp[NHP];
#pragma unroll
  for (int i = 0; i < NHP; ++i) {
    const int q = (i * NW + wave) * 64 + lane;
    const int v = q >> 1, part = q & 1;
    const int hz = min(v / hhw, p.HD - 1), r2 = v % hhw;
    const int hy = r2 / p.HW, hx = r2 - hy * p.HW;
    h_zyx[i] = q < pieces ? (hz << 16 | hy << 8 | hx) : -1;
    h_sp[i] = (part ^ (hy & 1)) * 8;
  }
  // ... and its weight pieces per stage
  int w_off[WPW];                                 // (row * Kp + piece * 8) of this lane's piece, -1: pad / past the pack's rows
  int w_pc[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int inst = wave * WPW + i;
    const int q = inst * 64 + lane;
    const int r = q / 33, pc = q - r * 33;
    const bool ok = inst < WINSTR && pc < 32 && r < WROWS && co0 + r < d.w_rows;
    w_off[i] = ok ? (co0 + r) * d.Kp : -1;
    w_pc[i] = pc * 8;
  }
  __syncthreads();

  const int nsteps = (d.T * CC + 31) >> 5;
  const int nstages = (nsteps + SK - 1) / SK;
  const int boxes_per_img = p.nbd * p.nbh * p.nbw;
  const int nboxes = d.N * boxes_per_img;
  auto issue_w = [&](int g, int chunk, int buf) {   // weight stage g of channel chunk `chunk` -> LDS buffer `buf`
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      const int inst = wave * WPW + i;            // wave-uniform
      if (inst < WINSTR) {
        const int kk = g * (SK * 32) + w_pc[i];
        const int tap = kk >> 4, c0 = kk & (CC - 1);
        const bool ok = w_off[i] >= 0 && tap < d.T;
        unsigned off = ((unsigned)w_off[i] + (unsigned)(tap * d.Ci + chunk * CC + c0)) * 2u;
        asm volatile("" : "+v"(off));
        glds16(ok ? p.w + off : p.zero, wst + buf * WSTAGE + inst * 1024);
      }
    }
  };
  auto issue_halo = [&](int box, int chunk, int buf) {
    int b = box;
    const int bx = b % p.nbw; b /= p.nbw;
    const int by = b % p.nbh; b /= p.nbh;
    const int bz = b % p.nbd;
    const int n = b / p.nbd;
    const char* in_n = p.in + ((size_t)n * d.Di * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
    const unsigned short* zrow = ztab + bz * p.HD;
    const unsigned short* yrow = ytab + by * p.HH;
    const unsigned short* xrow = xtab + bx * p.HW;
    unsigned iz[NHP], iy[NHP], ix[NHP];
#pragma unroll
    for (int i = 0; i < NHP; ++i) {
      const int zyx = h_zyx[i] < 0 ? 0 : h_zyx[i];
      iz[i] = zrow[zyx >> 16]; iy[i] = yrow[(zyx >> 8) & 255]; ix[i] = xrow[zyx & 255];
    }
    char* halo = halo0 + (size_t)buf * hbytes;
#pragma unroll
    for (int i = 0; i < NHP; ++i) {
      const bool ok = h_zyx[i] >= 0 && !((iz[i] | iy[i] | ix[i]) & 0x8000u);
      unsigned off = (((iz[i] * (unsigned)d.Hi + iy[i]) * (unsigned)d.Wi + ix[i]) * (unsigned)d.in_cs +
                      (unsigned)(chunk * CC + h_sp[i])) * 2u;
      asm volatile("" : "+v"(off));
      const int inst = i * NW + wave;             // (wave-uniform) instructions past the halo fill the sink: every wave issues NHP
      glds16(ok ? in_n + off : p.zero, inst * 64 < pieces ? halo + (size_t)inst * 1024 : sink);
    }
  };

  // bias of this workgroup's channels, once (a load in the epilogue would wait for the staging in flight)
  float bv[TI][4];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + i * 16 + kg * 4 + r;
      bv[i][r] = (p.bias && co < d.Co) ? p.bias[co] : 0.f;
    }
  f32x4 acc[TI][4];
  int wbuf = 0, hb = 0;                           // ring positions of the current stage / unit
  const int chunks = p.chunks;
  int box = blockIdx.x;
  if (box < nboxes) {
    issue_halo(box, 0, 0);
    issue_w(0, 0, 0);
  }
  bool halo_young = false;                        // the next unit's halo went out behind the weights waited for next
#pragma clang loop unroll(disable)
  for (; box < nboxes; box += gridDim.x) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma clang loop unroll(disable)
    for (int chunk = 0; chunk < chunks; ++chunk) {
      const bool last_unit = chunk + 1 == chunks && box + (int)gridDim.x >= nboxes;
      const char* halo = halo0 + (size_t)hb * hbytes;
#pragma clang loop unroll(disable)
      for (int g = 0; g < nstages; ++g) {
        // stage g landed (this wave's share) — and with it everything older (the unit's halo for g == 0); a halo that went out
        // behind these weights (the next unit's, issued one stage ago) may still fly
        if (halo_young) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NHP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        halo_young = false;
        __builtin_amdgcn_s_barrier();               // ... everybody's; the previous stage (and unit) is fully consumed
        asm volatile("" ::: "memory");
        // next weights: stage g + 1 of this unit, or stage 0 of the next one
        if (g + 1 < nstages) issue_w(g + 1, chunk, wbuf ^ 1);
        else if (!last_unit) issue_w(0, chunk + 1 < chunks ? chunk + 1 : 0, wbuf ^ 1);
        if (g == 0 && !last_unit) {                 // the next unit's halo into the buffer the previous unit left
          if (chunk + 1 < chunks) issue_halo(box, chunk + 1, hb ^ 1);
          else issue_halo(box + gridDim.x, 0, hb ^ 1);
          halo_young = nstages > 1;                 // (one stage per unit: the next wait is that unit's own)
        }
        const char* wb = wst + wbuf * WSTAGE;
        int vo[SK];                                 // byte offset of this lane's piece relative to its voxel's
        {
          int tt[SK];
#pragma unroll
          for (int u = 0; u < SK; ++u) {
            const int tap = ((g * SK + u) * 32 + kg * 8) >> 4;
            tt[u] = tap < d.T ? tap : 0;            // past the last tap the weights are zero: any finite B
          }
          // (tap table reads through inline asm: a compiler-visible LDS read here would wait vmcnt(0) for the staging in flight)
          const unsigned t0 = lds_addr(toff);
#pragma unroll
          for (int u = 0; u < SK; ++u) asm volatile("ds_read_b32 %0, %1" : "=v"(vo[u]) : "v"(t0 + (unsigned)tt[u] * 4u) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int u = 0; u < SK; ++u) {
            asm volatile("" : "+v"(vo[u]));
            const int e = vo[u];
            const int bit = (e ^ lyp) & 1;
            vo[u] = (e >> 8) * 32 + (((kg & 1) ^ bit) << 4);
          }
        }
        const unsigned wba = lds_addr(wb) + (unsigned)(row * 528 + kg * 16);
        unsigned xa[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[j] = lds_addr(halo) + (unsigned)(pbase[j] * (CC * 2));
        bf16x8 wA[TI], xA[4], wB[TI], xB[4];
        auto load_frags = [&](auto u_tag, bf16x8 (&wf)[TI], bf16x8 (&xf)[4]) {
          constexpr int u = decltype(u_tag)::value;
          static_for<0, TI>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            lds_read128<i * 16 * 528 + u * 64>(wf[i], wba);
          });
#pragma unroll
          for (int j = 0; j < 4; ++j) lds_read128<0>(xf[j], xa[j] + (unsigned)vo[u]);
        };
        auto wait_mma = [&](auto pending_tag, bf16x8 (&wf)[TI], bf16x8 (&xf)[4]) {
          constexpr int N = decltype(pending_tag)::value;
          if constexpr (TI == 1) gs_lgkm_wait<N>(wf[0], xf[0], xf[1], xf[2], xf[3]);
          else gs_lgkm_wait<N>(wf[0], wf[1], xf[0], xf[1], xf[2], xf[3]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        };
        load_frags(std::integral_constant<int, 0>{}, wA, xA);
        static_for<0, SK / 2>([&](auto h_tag) {
          constexpr int u = 2 * decltype(h_tag)::value;
          load_frags(std::integral_constant<int, u + 1>{}, wB, xB);
          wait_mma(std::integral_constant<int, TI + 4>{}, wA, xA);
          if constexpr (u + 2 < SK) {
            load_frags(std::integral_constant<int, u + 2>{}, wA, xA);
            wait_mma(std::integral_constant<int, TI + 4>{}, wB, xB);
          } else {
            wait_mma(std::integral_constant<int, 0>{}, wB, xB);
          }
        });
        wbuf ^= 1;
      }
      hb ^= 1;
    }

    // ---- epilogue of the box: bias, partial statistics (one slot per box), activation, [accumulate], 8-B NDHWC stores --------
    int b = box;
    const int bx = b % p.nbw; b /= p.nbw;
    const int by = b % p.nbh; b /= p.nbh;
    const int bz = b % p.nbd;
    const int n = b / p.nbd;
    const int slot = (bz * p.nbh + by) * p.nbw + bx;
    const int oz0 = bz * p.BD, oy0 = by * p.BH, ox0 = bx * p.BW;
    const bool want_stats = d.stats_slots > 0;
    float s1[TI][4], s2[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int co = co0 + i * 16 + kg * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int oz = oz0 + (pvox[j] >> 16), oy = oy0 + ((pvox[j] >> 8) & 255), ox = ox0 + (pvox[j] & 255);
        const bool pv = oz < d.Do && oy < d.Ho && ox < d.Wo;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r] + bv[i][r];
          if (pv) { s1[i][r] += v[r]; s2[i][r] += v[r] * v[r]; }
          v[r] = apply_act_small(v[r], d.act, d.slope);
        }
        if (pv && co < d.Co) {
          const size_t opix = (((size_t)n * d.Do + oz) * d.Ho + oy) * d.Wo + ox;
          uint2* dst = reinterpret_cast<uint2*>(p.out + (opix * d.out_cs + d.out_co + co) * 2);
          uint2 o;
          o.x = pack_bf2(v[0], v[1]);
          o.y = pack_bf2(v[2], v[3]);
          if (d.accumulate) {   // bf16 read-modify-write, same rounding points as gconv_kernel
            const uint2 old = *dst;
            o.x = pack_bf2(bf_lo(o.x) + bf_lo(old.x), bf_hi(o.x) + bf_hi(old.x));
            o.y = pack_bf2(bf_lo(o.y) + bf_lo(old.y), bf_hi(o.y) + bf_hi(old.y));
          }
          *dst = o;
        }
      }
    }
    if (want_stats) {
      const unsigned red0 = lds_addr(red);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row16_sum(s1[i][r]), q = row16_sum(s2[i][r]);
          if (row == 0) {
            const unsigned ad = red0 + (unsigned)((wave * 64 + i * 16 + kg * 4 + r) * 8);
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4" ::"v"(ad), "v"(a), "v"(q) : "memory");
          }
        }
      lds_barrier();
      if (tid < WROWS && co0 + tid < d.Co) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          float ra, rq;
          const unsigned ad = red0 + (unsigned)((w * 64 + tid) * 8);
          asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(ra), "=v"(rq) : "v"(ad) : "memory");
          a += ra; q += rq;
        }
        float* sp = p.stats + (((size_t)n * d.stats_slots + d.stats_slot0 + slot) * 2) * d.Co;
        sp[co0 + tid] = a;
        sp[d.Co + co0 + tid] = q;
      }
      lds_barrier();                                // red may be rewritten by the next box
    }
    // (the epilogue's loads / stores sit between the staging in flight and the next stage's wait: that wait is a full one)
    halo_young = false;
  }
}


namespace {
struct HPlan {
  bool ok;
  int BD, BH, BW, HD, HH, HW, dmin, hmin, wmin, nbd, nbh, nbw, CC, TI, cog, lds, NW;
  bool v2;               // hconv2_kernel: persistent workgroups on 8 x 8 x 8 boxes, next unit's halo under the tap loop
};

HPlan plan(const gs_gconv_desc* d) {
  HPlan h{};
  const bool enabled = gs_opt(GS_OPT_HCONV) != 0;
  if (!enabled) return h;
  // T < 9: the W-folded boundary convs (7 taps) measured slower here than on the im2col kernel (stem fwd 93 vs 74 us)
  if (d->si != 1 || d->so != 1 || d->Co > 64 || d->Ci > 64 || d->T < 9) return h;
  // wide on both sides: the im2col kernel is the better fit (measured) — hconv2 >= 2 tries the persistent form on volumes
  const bool wide2 = gs_opt(GS_OPT_HCONV2) >= 2 && d->Do >= 8;
  if (d->Ci > 32 && d->Co > 16 && !wide2) return h;
  if (d->Dc != d->Do || d->Hc != d->Ho || d->Wc != d->Wo || d->pz || d->py || d->px) return h;
  int lo[3] = {127, 127, 127}, hi[3] = {-128, -128, -128};
  for (int t = 0; t < d->T; ++t) {
    const int o[3] = {d->dd[t], d->dh[t], d->dw[t]};
    for (int a = 0; a < 3; ++a) { if (o[a] < lo[a]) lo[a] = o[a]; if (o[a] > hi[a]) hi[a] = o[a]; }
  }
  // volumes with room for them: 8 x 8 x 8 boxes on 8 waves — the weight stream (64 KB per box for 16 -> 16 channels, k5) is
  // shared by twice the voxels and the halo box is 3.4x its output instead of 4.5x (826 -> 492 MB of L2 -> LDS traffic per
  // launch at 128^3, which is what bounds these layers)
  // (16 output channels only: with 32 the larger box leaves room for ONE workgroup per CU and measured slower, 728 vs 617 us)
  // 17..64 output channels on volumes: the persistent double-buffered form (hconv2_kernel), 8 x 8 x 8 boxes on 8 waves
  h.v2 = gs_opt(GS_OPT_HCONV2) != 0 && d->Do >= 8 && d->Co > 16 && d->Ci >= 16 && d->T <= GS_MAX_TAPS;
  h.NW = (d->Do >= 8 && (d->Co <= 16 || h.v2) && gs_opt(GS_OPT_HCONV_BOX8)) ? 8 : 4;
  if (h.NW != 8) h.v2 = false;
  if (d->Do > 1) { h.BD = h.NW == 8 ? 8 : 4; h.BH = 8; h.BW = 8; } else { h.BD = 1; h.BH = 16; h.BW = 16; }
  h.HD = h.BD + hi[0] - lo[0]; h.HH = h.BH + hi[1] - lo[1]; h.HW = h.BW + hi[2] - lo[2];
  h.dmin = lo[0]; h.hmin = lo[1]; h.wmin = lo[2];
  h.nbd = (d->Do + h.BD - 1) / h.BD; h.nbh = (d->Ho + h.BH - 1) / h.BH; h.nbw = (d->Wo + h.BW - 1) / h.BW;
  h.CC = d->Ci < 16 ? d->Ci : 16;
  h.TI = d->Co <= 16 ? 1 : 2;                      // 16 or 32 output channels per workgroup,
  // (persistent form on a small volume: 16 channels per workgroup put a 32- / 64-channel layer's 64 boxes on 128 / 256 workgroups)
  if (h.v2 && gs_opt(GS_OPT_HCONV2) >= 3 && d->Co == 32 && (long long)d->N * h.nbd * h.nbh * h.nbw * 2 <= 256) h.TI = 1;
  if (h.v2 && gs_opt(GS_OPT_HCONV2) >= 4 && d->Co == 64 && (long long)d->N * h.nbd * h.nbh * h.nbw * 4 <= 256) h.TI = 1;
  h.cog = (d->Co + h.TI * 16 - 1) / (h.TI * 16);   // wider layers split over blockIdx.y (each stages its own halo)
  const long long hv = (long long)h.HD * h.HH * h.HW;
  const long long halo_bytes = (hv * h.CC * 2 + 1023) / 1024 * 1024 + 1024;
  const int wstage = (h.TI * 16 * 33 + 63) / 64 * 1024;
  h.lds = GS_MAX_TAPS * 4 + h.NW * 64 * 2 * 4 + 2 * wstage + (int)halo_bytes;
  if (h.v2) {
    const long long tab_bytes = ((long long)h.nbd * h.HD + (long long)h.nbh * h.HH + (long long)h.nbw * h.HW) * 2;
    const long long hb2 = (hv * h.CC * 2 + 1023) / 1024 * 1024;
    const long long lds2 = GS_MAX_TAPS * 4 + 8 * 64 * 2 * 4 + 2 * wstage + 2 * hb2 + 1024 + (tab_bytes + 15) / 16 * 16;
    const long long boxes = (long long)d->N * h.nbd * h.nbh * h.nbw;
    if (hv * 2 <= 7 * 512 && lds2 <= 160 * 1024 && tab_bytes < 16384 && d->Di < 32768 && d->Hi < 32768 && d->Wi < 32768 &&
        boxes >= 64 && (long long)d->Di * d->Hi * d->Wi * d->in_cs * 2 < (1LL << 32)) {
      h.lds = (int)lds2;
      h.ok = true;
      return h;
    }
    // (does not fit: the 4-wave form below)
    if (d->Ci > 32 && d->Co > 16) return h;
    h.v2 = false;
    h.NW = 4;
    h.BD = 4;
    h.HD = h.BD + hi[0] - lo[0];
    h.nbd = (d->Do + h.BD - 1) / h.BD;
    const long long hv4 = (long long)h.HD * h.HH * h.HW;
    h.lds = GS_MAX_TAPS * 4 + h.NW * 64 * 2 * 4 + 2 * wstage + (int)((hv4 * h.CC * 2 + 1023) / 1024 * 1024 + 1024);
  }
  // two workgroups per CU must fit, so a workgroup's staging overlaps the other's tap loop
  if (h.lds > 110 * 1024) return h;
  h.ok = true;
  return h;
}

template <int TI, int CC, int NW>
int launch_h(const HConvK& k, int blocks, int cog, int lds, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconv_kernel<TI, CC, NW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024));
    configured = true;
  }
  hipLaunchKernelGGL((hconv_kernel<TI, CC, NW>), dim3(blocks, cog), dim3(NW * 64), lds, st, k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
}  // namespace

// number of partial-statistics slots per image this class writes if it runs on the halo kernel, 0 if it does not
int gs_hconv5_slots(const gs_gconv_desc* d);
int gs_hconv5_try(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out, float* stats,
                  void* stream, int* handled);

int gs_hconv_slots(const gs_gconv_desc* d) {
  if (const int s5 = gs_hconv5_slots(d)) return s5;       // the 16 -> 16 k5 volume layers: hconv5.hip
  const HPlan h = plan(d);
  return h.ok ? h.nbd * h.nbh * h.nbw : 0;
}

// returns 0 and sets *handled when the layer ran here; *handled = 0 -> the caller falls back to gconv_kernel
int gs_hconv_try(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out, float* stats,
                 void* stream, int* handled) {
  *handled = 0;
  if (int rc = gs_hconv5_try(d, in, w_pack, bias, out, stats, stream, handled)) return rc;
  if (*handled) return 0;
  const HPlan h = plan(d);
  if (!h.ok) return 0;
  HConvK k;
  k.in = static_cast<const char*>(in);
  k.w = static_cast<const char*>(w_pack);
  k.bias = bias;
  k.out = static_cast<char*>(out);
  k.stats = stats;
  k.zero = static_cast<const char*>(gs_zero_page());
  GS_REQUIRE(k.zero, "gs_gconv_forward: library not initialised (call gs_init)");
  k.BD = h.BD; k.BH = h.BH; k.BW = h.BW; k.HD = h.HD; k.HH = h.HH; k.HW = h.HW;
  k.dmin = h.dmin; k.hmin = h.hmin; k.wmin = h.wmin; k.nbd = h.nbd; k.nbh = h.nbh; k.nbw = h.nbw;
  k.chunks = d->Ci / h.CC;
  k.cc_shift = h.CC == 8 ? 3 : 4;
  k.d = *d;
  const long long blocks = (long long)d->N * h.nbd * h.nbh * h.nbw;
  GS_REQUIRE(blocks > 0 && blocks < (1LL << 31), "gs_gconv_forward: bad grid %lld", blocks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  *handled = 1;
  if (h.v2) {
    static int cus = 0;
    if (!cus) {
      int dev = 0;
      hipDeviceProp_t prop;
      cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                ? prop.multiProcessorCount : 256;
    }
    const long long gmax = cus / h.cog > 0 ? cus / h.cog : 1;  // one workgroup per CU (channel groups side by side)
    const long long per = (blocks + gmax - 1) / gmax;          // boxes per workgroup: equal shares
    const int groups = (int)((blocks + per - 1) / per);
    static bool configured2 = false;
    if (!configured2) {
      GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconv2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      configured2 = true;
    }
    if (h.TI == 1) {
      static bool configured1 = false;
      if (!configured1) {
        GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconv2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured1 = true;
      }
      hipLaunchKernelGGL((hconv2_kernel<1>), dim3(groups, h.cog), dim3(512), h.lds, st, k);
    } else {
      hipLaunchKernelGGL((hconv2_kernel<2>), dim3(groups, h.cog), dim3(512), h.lds, st, k);
    }
    GS_CHECK_HIP(hipGetLastError());
    return 0;
  }
#define GS_H(TI_, CC_)                                                                                   \
  if (h.TI == TI_ && h.CC == CC_)                                                                        \
    return h.NW == 8 ? launch_h<TI_, CC_, 8>(k, (int)blocks, h.cog, h.lds, st) : launch_h<TI_, CC_, 4>(k, (int)blocks, h.cog, h.lds, st)
  GS_H(1, 8); GS_H(1, 16);
  GS_H(2, 8); GS_H(2, 16);
#undef GS_H
  *handled = 0;
  return 0;
}
