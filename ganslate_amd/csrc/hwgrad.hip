// Halo-resident weight gradient for narrow stride-1 layers (<= 64 dense channels, <= 32 gathered channels, many taps):
// the k5 convs of Vnet3D (ganslate/nn/generators/vnet/vnet3d.py:161,252,262-267) and the W-folded k7 boundary convs.
//
//   dw[p][t][q] += sum over pixels of a[pix][p] * g[B(pix + off_t)][q]
//
// wgrad.hip gathers the shifted operand once per tap from L2 (125x for a 5x5x5 kernel) and spends one fp32 atomic per
// output element per 1-2k pixels; with 16-32 channels that is L2- and atomic-bound (85 TFLOP/s on the 16-channel
// coupling convs at 128^3). Here a workgroup walks over boxes of 256 pixels (4x8x8, or 1x16x16 for images), stages the
// dense tile [256][P] and the gathered box + halo [voxels][16] in LDS once per box, and keeps the accumulators of ALL
// its taps in registers across its boxes: 8 waves x 16 taps, C[p][q] per tap, contraction over pixels with
// v_mfma_f32_16x16x32_bf16. Both operands are pixel-major, so fragments come from ds_read_b64_tr_b16: the 8 pixels of a
// box row shifted by a tap are 8 consecutive halo voxels. The dense fragment is shared by the 16 taps of a wave, so the
// loop issues 2 transpose reads per MFMA; atomics happen once per workgroup at the end.
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

struct HWGradK {
  const char* a;
  const char* g;
  const char* a2;          // optional second (dense, gathered) pair of the same layer: boxes >= nboxes1 come from it
  const char* g2;          // (two backward passes through a network merged into one launch, gs_wgrad_pair)
  int nboxes1;
  float* dw;
  const char* zero;
  int BD, BH, BW;        // pixel box
  int HD, HH, HW;        // halo box
  int dmin, hmin, wmin;
  int nbd, nbh, nbw;
  int nboxes;            // N * nbd * nbh * nbw
  int qchunks, phalves, tgroups;
  // deterministic accumulation (gs_wgrad_ws): workgroups that share output elements (the box groups, blockIdx.x) write
  // their partial sums to slab blockIdx.x of ws ([slabs][P * dw_ld] floats, dw's own layout) instead of fp32 atomics
  // on dw; wgrad_reduce_kernel then adds the slabs in a fixed order
  float* ws;
  long long ws_stride;
  // twin batch (gs_twin; hwgrad_wide only): d.N = 2 x nimg images per operand tensor, images >= nimg belong to the second
  // network, whose gradient buffer lies dw_delta floats behind dw. Workgroups blockIdx.x < gsplit walk the first network's
  // boxes, the others the second's; slab = blockIdx.x either way. One network: nimg = d.N, gsplit = gridDim.x.
  int nimg, gsplit;
  long long dw_delta;
  gs_wgrad_desc d;
};

template <int TI, int TPW>
__global__ __launch_bounds__(512) void hwgrad_kernel(const HWGradK p) {
  constexpr int NW = 8;                            // waves per workgroup; TPW = taps per wave (16, or 8 for <= 64 taps)
  constexpr int APITCH = TI * 32;                  // bytes per pixel row of the dense tile
  constexpr int APIECES = 256 * TI * 2;            // 16-B pieces of the dense tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* toff = reinterpret_cast<int*>(smem);        // [<= 128] halo-linear tap offsets of this tap group
  char* at = smem + 512;                           // dense tile [256 pixels][TI*16 channels]
  char* halo = at + ((APIECES * 16 + 1023) / 1024 * 1024);
  const gs_wgrad_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int y = blockIdx.y;
  const int qc = y % p.qchunks; y /= p.qchunks;
  const int ph = y % p.phalves;
  const int tg = y / p.phalves;
  const int tbase = tg * (NW * TPW);
  for (int t = tid; t < NW * TPW; t += 512) {
    const int tt = tbase + t;
    toff[t] = tt < d.T ? (((int)d.dd[tt] - p.dmin) * p.HH + ((int)d.dh[tt] - p.hmin)) * p.HW + ((int)d.dw_[tt] - p.wmin)
                       : 0;
  }

  f32x4 acc[TPW][TI];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int i = 0; i < TI; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fk = lane >> 4, frr = (lane & 15) >> 2, fcc = lane & 3;
  const int HV = p.HD * p.HH * p.HW, hhw = p.HH * p.HW;
  const int hpieces = HV * 2;
  const int pch0 = ph * TI * 16;                   // first dense channel of this workgroup
  const int ntaps = min(TPW, d.T - (tbase + wave * TPW));   // taps this wave owns (may be <= 0)

  for (int box = blockIdx.x; box < p.nboxes; box += gridDim.x) {
    int b = box;
    const int bx = b % p.nbw; b /= p.nbw;
    const int by = b % p.nbh; b /= p.nbh;
    const int bz = b % p.nbd;
    const int n = b / p.nbd;
    const int oz0 = bz * p.BD, oy0 = by * p.BH, ox0 = bx * p.BW;
    __syncthreads();   // tap table visible / previous box consumed
    // ---- dense tile: pixel-major [256][TI*16], zero for pixels outside the image / channels past P ----
    for (int q0 = wave * 64; q0 < APIECES; q0 += NW * 64) {
      const int q = q0 + lane;
      const int px = q / (TI * 2), part = q - px * (TI * 2);
      const int lz = px / (p.BH * p.BW), rem = px - lz * (p.BH * p.BW);
      const int ly = rem / p.BW, lx = rem - ly * p.BW;
      const int oz = oz0 + lz, oy = oy0 + ly, ox = ox0 + lx;
      const bool ok = oz < d.Da && oy < d.Ha && ox < d.Wa && pch0 + part * 8 < d.P;
      const size_t pix = (((size_t)n * d.Da + oz) * d.Ha + oy) * d.Wa + ox;
      const char* src = ok ? p.a + (pix * d.a_cs + d.a_co + pch0 + part * 8) * 2 : p.zero;
      glds16(src, at + (size_t)q0 * 16);
    }
    // ---- gathered box + halo: [voxels][16 channels of chunk qc] ----
    const char* g_n = p.g + ((size_t)n * d.Dg * d.Hg * d.Wg * d.g_cs + d.g_co) * 2;
    for (int q0 = wave * 64; q0 < hpieces; q0 += NW * 64) {
      const int q = q0 + lane;
      const int v = q >> 1, part = q & 1;
      const int hz = v / hhw, r2 = v - hz * hhw;
      const int hy = r2 / p.HW, hx = r2 - hy * p.HW;
      bool ok = q < hpieces && qc * 16 + part * 8 < d.Q;
      int iz = border_index(oz0 + hz + p.dmin, d.Dg, d.border, ok);
      int iy = border_index(oy0 + hy + p.hmin, d.Hg, d.border, ok);
      int ix = border_index(ox0 + hx + p.wmin, d.Wg, d.border, ok);
      iz = min(max(iz, 0), d.Dg - 1);
      iy = min(max(iy, 0), d.Hg - 1);
      ix = min(max(ix, 0), d.Wg - 1);
      unsigned off = ((unsigned)((iz * d.Hg + iy) * d.Wg + ix) * (unsigned)d.g_cs + (unsigned)(qc * 16 + part * 8)) * 2u;
      asm volatile("" : "+v"(off));
      const char* src = ok ? g_n + off : p.zero;
      glds16(src, halo + (size_t)q0 * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (ntaps > 0) {
      int tbv[TPW];                                // tap offsets of this wave (tap 0 stands in for the ones past T:
#pragma unroll                                     // no branch in the loop, their accumulators are dropped at the end)
      for (int t = 0; t < TPW; ++t) tbv[t] = toff[wave * TPW + (t < ntaps ? t : 0)];
#pragma unroll 1
      for (int ks = 0; ks < 8; ++ks) {             // 8 K-steps of 32 pixels = 4 box rows of 8
        // dense fragments (shared by every tap of this wave): rows = channels, k = the 8 pixels of box row ks*4+fk
        bf16x8 af[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const char* ap = at + (size_t)(ks * 32 + fk * 8 + frr) * APITCH + (i * 16 + fcc * 4) * 2;
          const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)GS_LDS(ap)));
          const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)GS_LDS(ap + 4 * APITCH)));
          af[i] = __builtin_bit_cast(bf16x8, uint4{lo.x, lo.y, hi.x, hi.y});
        }
        const int px0 = (ks * 4 + fk) * 8;         // first pixel of this lane's 8-pixel run (BW is a multiple of 8)
        const int lz = px0 / (p.BH * p.BW), rem = px0 - lz * (p.BH * p.BW);
        const int ly = rem / p.BW, lx0 = rem - ly * p.BW;
        const int rb = (lz * p.HH + ly) * p.HW + lx0 + frr;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          {
            const char* gp = halo + ((size_t)(rb + tbv[t]) * 16 + fcc * 4) * 2;
            const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)GS_LDS(gp)));
            const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)GS_LDS(gp + 4 * 32)));
            const bf16x8 gf = __builtin_bit_cast(bf16x8, uint4{lo.x, lo.y, hi.x, hi.y});
#pragma unroll
            for (int i = 0; i < TI; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], gf, acc[t][i], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- one atomic per output element per workgroup ---------------------------------------------------------------
  const int col = lane & 15;
  const int q = qc * 16 + col;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = tbase + wave * TPW + t;
    if (t < ntaps && q < d.Q) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pp = pch0 + i * 16 + fk * 4 + r;
          if (pp < d.P) {
            const size_t e = (size_t)pp * d.dw_ld + tap * d.Q + q;
            if (p.ws) p.ws[(size_t)blockIdx.x * p.ws_stride + e] = acc[t][i][r];
            else unsafeAtomicAdd(p.dw + e, acc[t][i][r]);
          }
        }
    }
  }
}

// ---- narrow layers with FEW taps (the W-folded k7 boundary convs of 2-D nets: 7 vertical taps, 32 <-> 64 channels) ----------
// hwgrad_kernel above spreads TAPS over its 8 waves; with 7 taps one wave would work. Here the workgroup stages the dense tile
// [256 pixels][P] and the gathered box + halo with ALL Q channels once per box, and the waves split the (p, q) plane: wave w
// owns the 16 x 16 block (w / nq, w % nq) for every tap (nq = Q / 16; np * nq <= 8), T accumulators in registers across the
// workgroup's boxes. wgrad_kernel gathered the shifted operand once per tap from L2 — 470 MB of L2 -> LDS traffic for a
// 100 MB layer, 115 / 69 us per launch at 130 / 218 TFLOP/s (profiles/r02_conv_table_v5.txt) — this form reads both operands
// once per box.
// Round 4: (a) every thread stages the same pieces of every box, so their decode (five integer divisions and three border
// rules per 16-byte piece: ~900 VALU instructions per thread per box next to 56 MFMAs per wave) happens once per workgroup,
// with the border-resolved source rows / columns in two small LDS tables as in hwgrad_wide; (b) the voxel pitch of both
// tiles is padded by one 16-byte piece (80 / 144 B for 32 / 64 channels; the pad piece of the lane-linear LDS-DMA image
// fetches the zero page) and a lane's two transpose reads take pixels 2 frr / 2 frr + 1 of its 8-pixel run instead of frr /
// frr + 4: a 32-lane read then touches 2 runs x 4 rows x 8 banks = all 64 banks once (64 / 128-byte pitches with rows
// frr: 2- to 4-way conflicts, 66 % of the kernel's LDS cycles, profiles/r03_trunk_pmc.txt). Boxes are 16 x 16 (launcher).
template <int TMAX>
__global__ __launch_bounds__(512) void hwgrad_ft_kernel(const HWGradK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* toff = reinterpret_cast<int*>(smem);        // [TMAX] halo-linear tap offsets
  const gs_wgrad_desc& d = p.d;
  const int np = p.phalves, nq = p.qchunks;        // 16-channel blocks of the dense / gathered side
  const int NPP = np * 2 + 1, NQP = nq * 2 + 1;    // 16-byte pieces per pixel / voxel incl. the pad piece
  const int APITCH = NPP * 16, GPITCH = NQP * 16;  // bytes per pixel / voxel
  const int HV = p.HH * p.HW;
  const int apieces = 256 * NPP, hpieces = HV * NQP;
  const int abytes = (apieces * 16 + 1023) / 1024 * 1024, hbytes = (hpieces * 16 + 1023) / 1024 * 1024 + 1024;
  char* at = smem + 256;
  char* halo = at + abytes;
  unsigned short* ytab = reinterpret_cast<unsigned short*>(halo + hbytes);
  unsigned short* xtab = ytab + p.nbh * p.HH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < TMAX)
    toff[tid] = tid < d.T ? (((int)d.dh[tid] - p.hmin) * p.HW + ((int)d.dw_[tid] - p.wmin)) * GPITCH : 0;
  for (int e = tid; e < p.nbh * p.HH + p.nbw * p.HW; e += 512) {      // see hwgrad_wide_kernel
    bool ok = true;
    int v;
    if (e < p.nbh * p.HH) {
      const int by = e / p.HH, hy = e - by * p.HH;
      v = border_index(by * 16 + hy + p.hmin, d.Hg, d.border, ok);
      v = min(max(v, 0), d.Hg - 1);
    } else {
      const int e2 = e - p.nbh * p.HH;
      const int bx = e2 / p.HW, hx = e2 - bx * p.HW;
      v = border_index(bx * 16 + hx + p.wmin, d.Wg, d.border, ok);
      v = min(max(v, 0), d.Wg - 1);
    }
    ytab[e] = ok ? (unsigned short)v : (unsigned short)0x8000;
  }
  const bool active = wave < np * nq;
  const int pi = active ? wave / nq : 0, qi = active ? wave % nq : 0;
  f32x4 acc[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fk = lane >> 4, frr = (lane & 15) >> 2, fcc = lane & 3;
  // the pieces this thread stages of every box, decoded once
  constexpr int NA = 5, NH = 7;                    // <= 2560 dense, <= 3584 halo pieces (launcher)
  int a_rel[NA], a_yx[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int q = i * 512 + wave * 64 + lane;
    const int px = q / NPP, part = q - px * NPP;
    const int ly = px >> 4, lx = px & 15;
    a_yx[i] = (q < apieces && part < np * 2 && part * 8 < d.P) ? (ly << 8 | lx) : -1;
    a_rel[i] = (ly * d.Wa + lx) * d.a_cs + part * 8;
  }
  int h_y[NH], h_x[NH], h_c[NH];
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int q = i * 512 + wave * 64 + lane;
    const int v = q / NQP, part = q - v * NQP;
    const int hy = v / p.HW;
    h_y[i] = min(hy, p.HH - 1); h_x[i] = v - hy * p.HW;
    h_c[i] = (q < hpieces && part < nq * 2 && part * 8 < d.Q) ? part * 8 : -1;
  }
  const bool ragged = (d.Ha & 15) != 0 || (d.Wa & 15) != 0;
  bool okz = true;
  int izs = border_index(p.dmin, d.Dg, d.border, okz);      // Da == 1: one source slice for the whole launch
  izs = min(max(izs, 0), d.Dg - 1);
  int tbv[TMAX];
  // twin batch: workgroups [0, gsplit) walk the first network's boxes, the rest the second's (its images follow, its sums go
  // to its own slabs / dw + dw_delta); nboxes counts ONE network's boxes
  const int net = (int)blockIdx.x >= p.gsplit ? 1 : 0;
  const int gx = (int)blockIdx.x - net * p.gsplit, gnum = net ? (int)gridDim.x - p.gsplit : p.gsplit;
  for (int box = gx; box < p.nboxes; box += gnum) {
    int b = box;
    const int bx = b % p.nbw; b /= p.nbw;
    const int by = b % p.nbh;
    const int n = b / p.nbh + net * p.nimg;
    const int oy0 = by * 16, ox0 = bx * 16;
    __syncthreads();   // tables visible / previous box consumed
    const char* a_n = p.a + ((((size_t)n * d.Ha + oy0) * d.Wa + ox0) * d.a_cs + d.a_co) * 2;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (i * 512 + wave * 64 < apieces) {           // wave-uniform
        bool ok = a_yx[i] >= 0;
        if (ragged) ok = ok && oy0 + (a_yx[i] >> 8) < d.Ha && ox0 + (a_yx[i] & 255) < d.Wa;
        glds16(ok ? a_n + (size_t)a_rel[i] * 2 : p.zero, at + (size_t)(i * 512 + wave * 64) * 16);
      }
    const char* g_n = p.g + (((size_t)n * d.Dg + izs) * d.Hg * d.Wg * d.g_cs + d.g_co) * 2;
    const unsigned short* yrow = ytab + by * p.HH;
    const unsigned short* xrow = xtab + bx * p.HW;
#pragma unroll
    for (int i = 0; i < NH; ++i)
      if (i * 512 + wave * 64 < hpieces) {           // wave-uniform: whole 64-piece instructions inside the halo
        const unsigned iy = yrow[h_y[i]], ix = xrow[h_x[i]];
        const bool ok = okz && h_c[i] >= 0 && !((iy | ix) & 0x8000u);
        unsigned off = ((iy * (unsigned)d.Wg + ix) * (unsigned)d.g_cs + (unsigned)h_c[i]) * 2u;
        asm volatile("" : "+v"(off));
        glds16(ok ? g_n + off : p.zero, halo + (size_t)(i * 512 + wave * 64) * 16);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (active) {
#pragma unroll
      for (int t = 0; t < TMAX; ++t) tbv[t] = toff[t];
#pragma unroll 1
      for (int ks = 0; ks < 8; ++ks) {             // 8 K-steps of 32 pixels
        const int run = ks * 4 + fk;               // this lane's 8-pixel run: box row run / 2, columns 8 (run & 1) ...
        const char* ap = at + (size_t)(run * 8 + 2 * frr) * APITCH + (pi * 16 + fcc * 4) * 2;
        const uint2 alo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)GS_LDS(ap)));
        const uint2 ahi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)GS_LDS(ap + APITCH)));
        const bf16x8 af = __builtin_bit_cast(bf16x8, uint4{alo.x, alo.y, ahi.x, ahi.y});
        const char* g0 = halo + (size_t)((run >> 1) * p.HW + (run & 1) * 8 + 2 * frr) * GPITCH + (qi * 16 + fcc * 4) * 2;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
          const char* gp = g0 + tbv[t];
          const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)GS_LDS(gp)));
          const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)GS_LDS(gp + GPITCH)));
          const bf16x8 gf = __builtin_bit_cast(bf16x8, uint4{lo.x, lo.y, hi.x, hi.y});
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, gf, acc[t], 0, 0, 0);
        }
      }
    }
  }
  if (!active) return;
  const int q = qi * 16 + (lane & 15);
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    if (t < d.T && q < d.Q) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pp = pi * 16 + fk * 4 + r;
        if (pp < d.P) {
          const size_t e = (size_t)pp * d.dw_ld + t * d.Q + q;
          if (p.ws) p.ws[(size_t)blockIdx.x * p.ws_stride + e] = acc[t][r];
          else unsafeAtomicAdd(p.dw + (size_t)net * p.dw_delta + e, acc[t][r]);
        }
      }
    }
  }
}

// ---- wide layers with few taps (the 3x3 residual convs: P, Q multiples of 64, T <= 9) -------------------------------------
// Same staging, different work split: a workgroup owns a 64 x 64 block of (p, q) for ALL taps; its 8 waves each own a
// 32 x 16 sub-block (two dense fragments, one gathered fragment per tap -> 2 + 2T transpose reads per 2T MFMAs, against
// 24 reads per 16 MFMAs in wgrad_kernel) and keep T x 2 accumulators in registers across the workgroup's boxes. Boxes are
// double-buffered: the LDS-DMA of the next box (32 KiB dense tile + 41 KiB halo) runs under the MFMAs of the current one.
// LDS-DMA traffic per MAC is ~3x below wgrad_kernel's (the gathered operand is staged once for 9 taps).
// 32-B slot swizzle of a 128-B row v: spreads the 8 (row, k-block) patches a 32-lane transpose read touches over all
// 64 banks (brute-forced over every alignment: 1 access per bank for the dense tile and the 2-D halo)
__device__ __forceinline__ int hw_swz(int v) { return ((v >> 1) ^ ((v >> 3) << 1)) & 3; }

template <int TMAX>
__global__ __launch_bounds__(512) void hwgrad_wide_kernel(const HWGradK p) {
  constexpr int APITCH = 128;                      // 64 channels of the dense tile
  constexpr int ABYTES = 256 * APITCH;             // 32 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* toff = reinterpret_cast<int*>(smem);        // [TMAX]
  const gs_wgrad_desc& d = p.d;
  const int HV = p.HD * p.HH * p.HW, hhw = p.HH * p.HW;
  constexpr int GPITCH = 144;                      // halo voxel pitch: 128 B of channels + 16 B pad (9 pieces), so the
                                                   // per-tap read address is rowbase(ks) + tapbase(t): one add, no swizzle
  const int hbytes = (HV * GPITCH + 1023) / 1024 * 1024 + 1024;
  char* bufs = smem + 256;
  auto at_of = [&](int b) { return bufs + (size_t)b * (ABYTES + hbytes); };
  auto halo_of = [&](int b) { return bufs + (size_t)b * (ABYTES + hbytes) + ABYTES; };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ptiles = d.P / 64;
  const int pt = blockIdx.y % ptiles, qt = blockIdx.y / ptiles;
  const int wp = wave >> 2, wq = wave & 3;         // 32-row half of p, 16-column quarter of q
  for (int t = tid; t < d.T; t += 512)
    toff[t] = ((((int)d.dd[t] - p.dmin) * p.HH + ((int)d.dh[t] - p.hmin)) * p.HW + ((int)d.dw_[t] - p.wmin)) * GPITCH;
  // Border-resolved source rows / columns of every halo row / column of every box row / column, once per workgroup:
  // ytab[by][hy] = B(by*BH + hy + hmin) in [0, Hg), 0x8000 for a zero-padded position; xtab likewise. Staging a box then
  // costs two 2-byte LDS reads and a multiply-add per piece; resolving the borders per piece per box (reflect / clamp /
  // validity on three axes) was 440 VALU instructions per wave per box next to 144 MFMAs, and with both waves of a SIMD
  // doing it at the same moment (right behind the barrier) it stretched every box by a third (profiles/r02_trunk_pmc.txt).
  unsigned short* ytab = reinterpret_cast<unsigned short*>(bufs + 2 * (size_t)(ABYTES + hbytes));
  unsigned short* xtab = ytab + p.nbh * p.HH;
  for (int e = tid; e < p.nbh * p.HH + p.nbw * p.HW; e += 512) {
    bool ok = true;
    int v;
    if (e < p.nbh * p.HH) {
      const int by = e / p.HH, hy = e - by * p.HH;
      v = border_index(by * p.BH + hy + p.hmin, d.Hg, d.border, ok);
      v = min(max(v, 0), d.Hg - 1);
    } else {
      const int e2 = e - p.nbh * p.HH;
      const int bx = e2 / p.HW, hx = e2 - bx * p.HW;
      v = border_index(bx * p.BW + hx + p.wmin, d.Wg, d.border, ok);
      v = min(max(v, 0), d.Wg - 1);
    }
    ytab[e] = ok ? (unsigned short)v : (unsigned short)0x8000;
  }
  __syncthreads();
  int tb[TMAX];                                    // tap byte offsets inside the halo, in registers for the whole kernel
#pragma unroll
  for (int t = 0; t < TMAX; ++t) tb[t] = toff[t < d.T ? t : 0];

  f32x4 acc[TMAX][2];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int fk = lane >> 4, frr = (lane & 15) >> 2, fcc = lane & 3;
  const int hpieces = HV * 9;

  // Each thread stages the same pieces of every box: decode them once (the integer divisions of a per-box decode were
  // 3/4 of the kernel's VALU work and made it issue-bound).
  constexpr int NA = 4;                                  // dense pieces per thread: 2048 / 512
  constexpr int NHMAX = 8;                               // halo pieces per thread (<= 4096 pieces)
  int a_rel[NA], a_lz[NA], a_ly[NA], a_lx[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int q = i * 512 + wave * 64 + lane;
    const int px = q >> 3, part = q & 7;
    // LDS row px = 8 m + 4 h + j holds box pixel 8 m + 2 j + h: the MFMA's K order inside an 8-pixel run is "even pixels, then
    // odd pixels" for BOTH operands, which is what makes the halo's transpose reads conflict-free (see rbk below); the
    // dense tile keeps its row pattern (and hw_swz) and only the source of each row moves
    const int spx = (px & ~7) | ((px & 3) << 1) | ((px >> 2) & 1);
    a_lz[i] = spx / (p.BH * p.BW);
    const int rem = spx - a_lz[i] * (p.BH * p.BW);
    a_ly[i] = rem / p.BW;
    a_lx[i] = rem - a_ly[i] * p.BW;
    const int spart = part ^ (hw_swz(px) << 1);          // LDS piece `part` of row px holds source piece spart
    a_rel[i] = ((a_lz[i] * d.Ha + a_ly[i]) * d.Wa + a_lx[i]) * d.a_cs + pt * 64 + spart * 8;
  }
  int h_y[NHMAX], h_x[NHMAX], h_c[NHMAX];            // halo row / column (table indices) and channel of each piece; HD == 1
#pragma unroll
  for (int i = 0; i < NHMAX; ++i) {
    const int q = i * 512 + wave * 64 + lane;
    const int v = q / 9, part = q - v * 9;               // piece 8 of a voxel is the pad
    const int r2 = v % hhw;
    const int hy = r2 / p.HW;
    h_y[i] = min(hy, p.HH - 1); h_x[i] = r2 - hy * p.HW;
    h_c[i] = (q < hpieces && part < 8) ? qt * 64 + part * 8 : -1;
  }
  const bool ragged = (d.Ha % p.BH) != 0 || (d.Wa % p.BW) != 0;

  // this workgroup's network (twin batch) and its place among the workgroups that share the (p, q) block
  const int net = (int)blockIdx.x >= p.gsplit ? 1 : 0;
  const int gx = (int)blockIdx.x - net * p.gsplit, gnum = net ? (int)gridDim.x - p.gsplit : p.gsplit;
  auto issue_box = [&](int box, int b) {                 // box index among this network's boxes (first pair, then second)
    int bb = box;
    const int bx = bb % p.nbw; bb /= p.nbw;
    const int by = bb % p.nbh; bb /= p.nbh;
    const int bz = bb % p.nbd;
    int n = bb / p.nbd;
    const bool second = box >= p.nboxes1;                // wave-uniform: the second operand pair
    if (second) n -= p.nimg;                             // image index inside the second pair
    n += net * p.nimg;                                   // ... of this network's half of the batch
    const int oz0 = bz * p.BD, oy0 = by * p.BH, ox0 = bx * p.BW;
    char* at = at_of(b);
    char* halo = halo_of(b);
    const size_t pix0 = (((size_t)n * d.Da + oz0) * d.Ha + oy0) * d.Wa + ox0;
    const char* a_n = (second ? p.a2 : p.a) + (pix0 * d.a_cs + d.a_co) * 2;
    if (ragged) {                                        // wave-uniform: boxes that hang over the image edge
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool ok = oz0 + a_lz[i] < d.Da && oy0 + a_ly[i] < d.Ha && ox0 + a_lx[i] < d.Wa;
        const char* src = ok ? a_n + (size_t)a_rel[i] * 2 : p.zero;
        glds16(src, at + (size_t)(i * 512 + wave * 64) * 16);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) glds16(a_n + (size_t)a_rel[i] * 2, at + (size_t)(i * 512 + wave * 64) * 16);
    }
    // gathered operand: the slice (depth border rule) is the same for the whole box, rows / columns come from the tables
    bool okz = true;
    int izs = border_index(oz0 + p.dmin, d.Dg, d.border, okz);
    izs = min(max(izs, 0), d.Dg - 1);
    const char* g_n = (second ? p.g2 : p.g) + (((size_t)n * d.Dg + izs) * d.Hg * d.Wg * d.g_cs + d.g_co) * 2;
    const unsigned short* yrow = ytab + by * p.HH;
    const unsigned short* xrow = xtab + bx * p.HW;
    unsigned iy[NHMAX], ix[NHMAX];
#pragma unroll
    for (int i = 0; i < NHMAX; ++i)
      if (i * 512 + wave * 64 < hpieces) { iy[i] = yrow[h_y[i]]; ix[i] = xrow[h_x[i]]; }
#pragma unroll
    for (int i = 0; i < NHMAX; ++i) {
      if (i * 512 + wave * 64 < hpieces) {               // wave-uniform: whole 64-piece instructions inside the halo
        const bool ok = okz && h_c[i] >= 0 && !((iy[i] | ix[i]) & 0x8000u);
        unsigned off = ((iy[i] * (unsigned)d.Wg + ix[i]) * (unsigned)d.g_cs + (unsigned)h_c[i]) * 2u;
        asm volatile("" : "+v"(off));
        const char* src = ok ? g_n + off : p.zero;
        glds16(src, halo + (size_t)(i * 512 + wave * 64) * 16);
      }
    }
  };
  // dense-tile byte offsets of this lane's transpose reads inside a K-step (the swizzle only looks at row bits 1..3,
  // which a K-step offset of 32 rows does not touch)
  int aoff[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = fk * 8 + h * 4 + frr;
      aoff[i][h] = r * APITCH + (((wp * 2 + i) ^ hw_swz(r)) << 5) + fcc * 8;
    }
  // halo row base of this lane's 8-pixel run, per K-step
  int rbk[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int px0 = (ks * 4 + fk) * 8;
    const int lz = px0 / (p.BH * p.BW), rem = px0 - lz * (p.BH * p.BW);
    const int ly = rem / p.BW, lx0 = rem - ly * p.BW;
    // A 32-lane half of a transpose read covers two 8-pixel runs x 4 rows x 4 column pieces. With the rows at pixels
    // +0..3 of each run (pitch 144 B = 36 banks, runs 8 pixels = 32 banks apart) the 8-bank patches overlapped pairwise:
    // SQ_LDS_BANK_CONFLICT = 2 extra cycles on every one of these reads, 45 % of the kernel's LDS cycles
    // (profiles/r02_trunk_pmc.txt). Rows at the EVEN pixels +0,2,4,6 (the odd ones in the second read) land 8 banks apart:
    // 2 runs x 4 rows x 8 banks = all 64 banks once.
    rbk[ks] = ((lz * p.HH + ly) * p.HW + lx0 + 2 * frr) * GPITCH + wq * 32 + fcc * 8;   // byte offset incl. this lane's columns
  }

  int cur = 0;
  int box = gx;
  if (box < p.nboxes) issue_box(box, 0);
  for (; box < p.nboxes; box += gnum) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this box landed (nothing else is outstanding)
    __syncthreads();                                     // ... for every wave; the other buffer is fully consumed
    if (box + gnum < p.nboxes) issue_box(box + gnum, cur ^ 1);
    // ---- 8 K-steps x T taps as a rolling pipeline of "units" (one tap of one K-step = 2 transpose reads of the gathered
    // fragment, plus the 4 reads of the K-step's two dense fragments in front of tap 0; 2 MFMAs) ---------------------
    // The reads go through inline asm (common.hpp, lds_read128 family): with the LDS-DMA in this loop hipcc waits
    // lgkmcnt(0) before every MFMA block, and the 22 reads of a K-step do not even fit the 4-bit counter, so the former
    // "issue the next K-step's reads, then run this one's MFMAs" ran read - wait - MFMA (matrix pipe 38 % busy, LDS 23 %).
    // Here the reads of unit u + D are issued before the MFMAs of unit u and the wait in front of those MFMAs counts the
    // reads issued after unit u's own (<= 14).
    const unsigned at_a = lds_addr(at_of(cur)), ha = lds_addr(halo_of(cur));
    constexpr int D = 5, NU = 8 * TMAX;
    uint2 glo[D + 1], ghi[D + 1], alo[2][2], ahi[2][2];
    auto issue_unit = [&](auto uu) {
      constexpr int u = decltype(uu)::value, ks = u / TMAX, t = u % TMAX, slot = u % (D + 1);
      if constexpr (t == 0) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          lds_read64_tr<ks * 32 * APITCH>(alo[ks & 1][i2], at_a + (unsigned)aoff[i2][0]);
          lds_read64_tr<ks * 32 * APITCH>(ahi[ks & 1][i2], at_a + (unsigned)aoff[i2][1]);
        }
      }
      const unsigned g0 = ha + (unsigned)(rbk[ks] + tb[t]);
      lds_read64_tr<0>(glo[slot], g0);
      lds_read64_tr<GPITCH>(ghi[slot], g0);
    };
    static_for<0, D>(issue_unit);
    static_for<0, NU>([&](auto uu) {
      constexpr int u = decltype(uu)::value, ks = u / TMAX, t = u % TMAX, slot = u % (D + 1);
      if constexpr (u + D < NU) issue_unit(std::integral_constant<int, u + D>{});
      constexpr int last = u + D < NU ? u + D : NU - 1;
      constexpr int newer = 2 * (last - u) + 4 * ((last / TMAX) - ks);      // reads issued after unit u's
      if constexpr (t == 0)
        gs_lgkm_wait64<newer>(glo[slot], ghi[slot], alo[ks & 1][0], ahi[ks & 1][0], alo[ks & 1][1], ahi[ks & 1][1]);
      else
        gs_lgkm_wait64<newer>(glo[slot], ghi[slot]);
      const bf16x8 gf = __builtin_bit_cast(bf16x8, uint4{glo[slot].x, glo[slot].y, ghi[slot].x, ghi[slot].y});
      const bf16x8 a0 = __builtin_bit_cast(bf16x8, uint4{alo[ks & 1][0].x, alo[ks & 1][0].y, ahi[ks & 1][0].x, ahi[ks & 1][0].y});
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, uint4{alo[ks & 1][1].x, alo[ks & 1][1].y, ahi[ks & 1][1].x, ahi[ks & 1][1].y});
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, gf, acc[t][0], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, gf, acc[t][1], 0, 0, 0);
    });
    cur ^= 1;
  }

  const int col = lane & 15;
  const int q = qt * 64 + wq * 16 + col;
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    if (t < d.T) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pp = pt * 64 + wp * 32 + i * 16 + fk * 4 + r;
          const size_t e = (size_t)pp * d.dw_ld + t * d.Q + q;
          if (p.ws) p.ws[(size_t)blockIdx.x * p.ws_stride + e] = acc[t][i][r];
          else unsafeAtomicAdd(p.dw + net * p.dw_delta + e, acc[t][i][r]);
        }
    }
  }
}

// ---- hwgrad_kernel's 16-taps-per-wave volume form, rebuilt on hwgrad_wide's recipe (round 6) ---------------------------------
// hwgrad_kernel<*, 16> ran the V-Net's k5 weight gradients (16 -> 16 at 128^3: 176 us, 0.31 of the MFMA peak; 12.6 ms of the
// brats step with the 32-channel form) the way round 1 left it: (i) every box decoded its ~7 pieces per thread from scratch —
// five integer divisions and three border rules per 16-byte piece; (ii) ONE buffer: stage, wait, barrier, compute — and with
// 146 registers per lane only one workgroup fits a CU, so nothing ran under the staging; (iii) compiler-issued transpose reads,
// which hipcc serialises as read - wait - MFMA whenever an LDS-DMA is in the loop. Here: pieces decoded once per workgroup with
// the border-resolved source planes / rows / columns in three small LDS tables, two buffers (the next box is staged under the
// current one's MFMAs), and the rolling read pipeline of hwgrad_wide (inline-asm transpose reads D units ahead, counted lgkmcnt).
// Same LDS images, same accumulation order per workgroup: per-workgroup sums are bit-identical to hwgrad_kernel's.
template <int TI>
__global__ __launch_bounds__(512) void hwgrad2_kernel(const HWGradK p) {
  constexpr int NW = 8, TPW = 16;
  constexpr int APITCH = TI * 32;                  // bytes per pixel row of the dense tile
  constexpr int ABYTES = 256 * APITCH;             // 8 / 16 KiB
  constexpr int NA = 256 * TI * 2 / 512;           // dense pieces per thread
  constexpr int NH = 5;                            // halo pieces per thread (<= 2560 pieces, launcher)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* toff = reinterpret_cast<int*>(smem);        // [128] halo-linear tap offsets
  const gs_wgrad_desc& d = p.d;
  const int HV = p.HD * p.HH * p.HW, hhw = p.HH * p.HW;
  const int hpieces = HV * 2;
  const int hbytes = (HV * 32 + 1023) / 1024 * 1024 + 1024;
  char* bufs = smem + 512;
  auto at_of = [&](int b) { return bufs + (size_t)b * (ABYTES + hbytes); };
  auto halo_of = [&](int b) { return bufs + (size_t)b * (ABYTES + hbytes) + ABYTES; };
  unsigned short* ztab = reinterpret_cast<unsigned short*>(bufs + 2 * (size_t)(ABYTES + hbytes));
  unsigned short* ytab = ztab + p.nbd * p.HD;
  unsigned short* xtab = ytab + p.nbh * p.HH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int y = blockIdx.y;
  const int qc = y % p.qchunks; y /= p.qchunks;
  const int ph = y % p.phalves;
  const int pch0 = ph * TI * 16;                   // first dense channel of this workgroup
  for (int t = tid; t < NW * TPW; t += 512)
    toff[t] = t < d.T ? (((int)d.dd[t] - p.dmin) * p.HH + ((int)d.dh[t] - p.hmin)) * p.HW + ((int)d.dw_[t] - p.wmin) : 0;
  const int ntab = p.nbd * p.HD + p.nbh * p.HH + p.nbw * p.HW;
  for (int e = tid; e < ntab; e += 512) {          // border-resolved source index of every halo plane / row / column of every box
    bool ok = true;
    int v;
    if (e < p.nbd * p.HD) {
      const int bz = e / p.HD, hz = e - bz * p.HD;
      v = border_index(bz * p.BD + hz + p.dmin, d.Dg, d.border, ok);
      v = min(max(v, 0), d.Dg - 1);
    } else if (e < p.nbd * p.HD + p.nbh * p.HH) {
      const int e2 = e - p.nbd * p.HD;
      const int by = e2 / p.HH, hy = e2 - by * p.HH;
      v = border_index(by * p.BH + hy + p.hmin, d.Hg, d.border, ok);
      v = min(max(v, 0), d.Hg - 1);
    } else {
      const int e2 = e - p.nbd * p.HD - p.nbh * p.HH;
      const int bx = e2 / p.HW, hx = e2 - bx * p.HW;
      v = border_index(bx * p.BW + hx + p.wmin, d.Wg, d.border, ok);
      v = min(max(v, 0), d.Wg - 1);
    }
    ztab[e] = ok ? (unsigned short)v : (unsigned short)0x8000;
  }
  __syncthreads();
  const int ntaps = min(TPW, d.T - wave * TPW);    // taps this wave owns (may be <= 0)
  // tap byte offsets of this wave (tap 0 stands in for the ones past T: their accumulators are dropped at the end)
  int tb[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) tb[t] = __builtin_amdgcn_readfirstlane(toff[wave * TPW + (t < ntaps ? t : 0)] * 32);

  f32x4 acc[TPW][TI];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int i = 0; i < TI; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fk = lane >> 4, frr = (lane & 15) >> 2, fcc = lane & 3;

  // the pieces this thread stages of every box, decoded once
  int a_rel[NA], a_zyx[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int q = i * 512 + tid;
    const int px = q / (TI * 2), part = q - px * (TI * 2);
    const int lz = px / (p.BH * p.BW), rem = px - lz * (p.BH * p.BW);
    const int ly = rem / p.BW, lx = rem - ly * p.BW;
    a_zyx[i] = pch0 + part * 8 < d.P ? (lz << 16 | ly << 8 | lx) : -1;
    a_rel[i] = ((lz * d.Ha + ly) * d.Wa + lx) * d.a_cs + pch0 + part * 8;
  }
  int h_z[NH], h_y[NH], h_x[NH], h_c[NH];
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int q = i * 512 + tid;
    const int v = q >> 1, part = q & 1;
    const int hz = min(v / hhw, p.HD - 1), r2 = v % hhw;
    const int hy = r2 / p.HW;
    h_z[i] = hz; h_y[i] = hy; h_x[i] = r2 - hy * p.HW;
    h_c[i] = (q < hpieces && qc * 16 + part * 8 < d.Q) ? qc * 16 + part * 8 : -1;
  }
  const bool ragged = (d.Da % p.BD) != 0 || (d.Ha % p.BH) != 0 || (d.Wa % p.BW) != 0;

  auto issue_box = [&](int box, int b) {
    int bb = box;
    const int bx = bb % p.nbw; bb /= p.nbw;
    const int by = bb % p.nbh; bb /= p.nbh;
    const int bz = bb % p.nbd;
    const int n = bb / p.nbd;
    const int oz0 = bz * p.BD, oy0 = by * p.BH, ox0 = bx * p.BW;
    char* at = at_of(b);
    char* halo = halo_of(b);
    const char* a_n = p.a + (((((size_t)n * d.Da + oz0) * d.Ha + oy0) * d.Wa + ox0) * d.a_cs + d.a_co) * 2;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      bool ok = a_zyx[i] >= 0;
      if (ragged) ok = ok && oz0 + (a_zyx[i] >> 16) < d.Da && oy0 + ((a_zyx[i] >> 8) & 255) < d.Ha && ox0 + (a_zyx[i] & 255) < d.Wa;
      glds16(ok ? a_n + (size_t)a_rel[i] * 2 : p.zero, at + (size_t)(i * 512 + wave * 64) * 16);
    }
    const char* g_n = p.g + ((size_t)n * d.Dg * d.Hg * d.Wg * d.g_cs + d.g_co) * 2;
    const unsigned short* zrow = ztab + bz * p.HD;
    const unsigned short* yrow = ytab + by * p.HH;
    const unsigned short* xrow = xtab + bx * p.HW;
    unsigned iz[NH], iy[NH], ix[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i)
      if (i * 512 + wave * 64 < hpieces) { iz[i] = zrow[h_z[i]]; iy[i] = yrow[h_y[i]]; ix[i] = xrow[h_x[i]]; }
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      if (i * 512 + wave * 64 < hpieces) {         // wave-uniform: whole 64-piece instructions inside the halo
        const bool ok = h_c[i] >= 0 && !((iz[i] | iy[i] | ix[i]) & 0x8000u);
        unsigned off = (((iz[i] * (unsigned)d.Hg + iy[i]) * (unsigned)d.Wg + ix[i]) * (unsigned)d.g_cs + (unsigned)h_c[i]) * 2u;
        asm volatile("" : "+v"(off));
        glds16(ok ? g_n + off : p.zero, halo + (size_t)(i * 512 + wave * 64) * 16);
      }
    }
  };
  // byte offsets of this lane's transpose reads: dense fragments (rows = channels, k = the 8 pixels of box row ks*4 + fk) ...
  int aoff[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) aoff[i] = (fk * 8 + frr) * APITCH + (i * 16 + fcc * 4) * 2;
  // ... and the halo row base of this lane's 8-pixel run, per K-step
  int rbk[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int px0 = (ks * 4 + fk) * 8;             // first pixel of this lane's 8-pixel run (BW is a multiple of 8)
    const int lz = px0 / (p.BH * p.BW), rem = px0 - lz * (p.BH * p.BW);
    const int ly = rem / p.BW, lx0 = rem - ly * p.BW;
    rbk[ks] = ((lz * p.HH + ly) * p.HW + lx0 + frr) * 32 + fcc * 8;
  }

  int cur = 0;
  int box = blockIdx.x;
  if (box < p.nboxes) issue_box(box, 0);
  for (; box < p.nboxes; box += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this box landed (nothing else is outstanding)
    __syncthreads();                                     // ... for every wave; the other buffer is fully consumed
    if (box + (int)gridDim.x < p.nboxes) issue_box(box + gridDim.x, cur ^ 1);
    if (ntaps > 0) {
      // 8 K-steps x 16 taps as a rolling pipeline of units (one tap of one K-step: 2 transpose reads of the gathered fragment,
      // plus the 2 TI reads of the K-step's dense fragments in front of tap 0; TI MFMAs); reads of unit u + D go out before the
      // MFMAs of unit u, the wait in front of those counts the reads issued after unit u's own
      const unsigned at_a = lds_addr(at_of(cur)), ha = lds_addr(halo_of(cur));
      constexpr int D = TI == 1 ? 6 : 5, NU = 8 * TPW;      // (lgkmcnt counts to 15: 2 D + 2 TI reads may be younger)
      uint2 glo[D + 1], ghi[D + 1], alo[2][TI], ahi[2][TI];
      auto issue_unit = [&](auto uu) {
        constexpr int u = decltype(uu)::value, ks = u / TPW, t = u % TPW, slot = u % (D + 1);
        if constexpr (t == 0) {
#pragma unroll
          for (int i = 0; i < TI; ++i) {
            lds_read64_tr<ks * 32 * APITCH>(alo[ks & 1][i], at_a + (unsigned)aoff[i]);
            lds_read64_tr<ks * 32 * APITCH + 4 * APITCH>(ahi[ks & 1][i], at_a + (unsigned)aoff[i]);
          }
        }
        const unsigned g0 = ha + (unsigned)(rbk[ks] + tb[t]);
        lds_read64_tr<0>(glo[slot], g0);
        lds_read64_tr<4 * 32>(ghi[slot], g0);
      };
      static_for<0, D>(issue_unit);
      static_for<0, NU>([&](auto uu) {
        constexpr int u = decltype(uu)::value, ks = u / TPW, t = u % TPW, slot = u % (D + 1);
        if constexpr (u + D < NU) issue_unit(std::integral_constant<int, u + D>{});
        constexpr int last = u + D < NU ? u + D : NU - 1;
        constexpr int newer = 2 * (last - u) + 2 * TI * ((last / TPW) - ks);      // reads issued after unit u's
        gs_lgkm_wait_only<newer>();
        asm volatile("" : "+v"(glo[slot]), "+v"(ghi[slot]));
        if constexpr (t == 0) {
#pragma unroll
          for (int i = 0; i < TI; ++i) asm volatile("" : "+v"(alo[ks & 1][i]), "+v"(ahi[ks & 1][i]));
        }
        const bf16x8 gf = __builtin_bit_cast(bf16x8, uint4{glo[slot].x, glo[slot].y, ghi[slot].x, ghi[slot].y});
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const bf16x8 af = __builtin_bit_cast(bf16x8, uint4{alo[ks & 1][i].x, alo[ks & 1][i].y, ahi[ks & 1][i].x, ahi[ks & 1][i].y});
          acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, gf, acc[t][i], 0, 0, 0);
        }
      });
    }
    cur ^= 1;
  }

  // ---- one slab row (or atomic) per output element per workgroup ------------------------------------------------------------
  const int col = lane & 15;
  const int q = qc * 16 + col;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = wave * TPW + t;
    if (t < ntaps && q < d.Q) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pp = pch0 + i * 16 + fk * 4 + r;
          if (pp < d.P) {
            const size_t e = (size_t)pp * d.dw_ld + tap * d.Q + q;
            if (p.ws) p.ws[(size_t)blockIdx.x * p.ws_stride + e] = acc[t][i][r];
            else unsafeAtomicAdd(p.dw + e, acc[t][i][r]);
          }
        }
    }
  }
}

namespace {
template <int TI, int TPW>
int launch_hw(const HWGradK& k, dim3 grid, int lds, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hwgrad_kernel<TI, TPW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    configured = true;
  }
  hipLaunchKernelGGL((hwgrad_kernel<TI, TPW>), grid, dim3(512), lds, st, k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
}  // namespace

// returns 0 and sets *handled when the layer ran here; *handled = 0 -> the caller falls back to wgrad_kernel
// ws != nullptr: partial sums go to slabs of ws (see HWGradK) and *handled returns the number of slabs written (the caller
// runs the reduction); plan_only: nothing is launched, *handled is what a launch would return.
int gs_hwgrad_try2(const gs_wgrad_desc* d, const void* a, const void* g, const void* a2, const void* g2, float* dw,
                   float* ws, int plan_only, void* stream, int* handled, const gs_twin* tw);

// a2/g2 != nullptr: a second operand pair of the same layer (only the wide kernel merges; otherwise *handled stays 0).
// tw != nullptr: twin batch (only the wide kernel takes it; *handled then is the number of slabs PER NETWORK, the second
// network's slabs follow the first's)
int gs_hwgrad_try2(const gs_wgrad_desc* d, const void* a, const void* g, const void* a2, const void* g2, float* dw,
                   float* ws, int plan_only, void* stream, int* handled, const gs_twin* tw) {
  *handled = 0;
  const long long ws_stride = (long long)d->P * d->dw_ld;
  const bool enabled = gs_opt(GS_OPT_HWGRAD) != 0;
  const bool wide_enabled = gs_opt(GS_OPT_HWGRAD_WIDE) != 0;
  const bool planes_enabled = gs_opt(GS_OPT_HWGRAD_PLANES) != 0;
  // 3x3x3 layers of volumes (Resnet3D residual convs): the 27 taps are three depth planes of 9, and a plane is the 2-D
  // problem over (image, slice) pairs with the gathered operand read from slice z + dd (border rule applied in depth):
  // three launches of the wide kernel, each writing its own 9 tap rows of dw
  if (enabled && wide_enabled && planes_enabled && d->si == 1 && d->T == 27 && d->Da > 1 && d->P % 64 == 0 &&
      d->Q % 64 == 0 && d->P >= 64 && d->Q >= 64) {
    bool ok = true;
    for (int k = 0; k < 3 && ok; ++k)
      for (int t = 0; t < 9; ++t) {
        ok = ok && d->dd[9 * k + t] == d->dd[9 * k] && d->dh[9 * k + t] == d->dh[t] && d->dw_[9 * k + t] == d->dw_[t];
      }
    if (ok) {
      for (int k = 0; k < 3; ++k) {
        gs_wgrad_desc sub = *d;
        sub.T = 9;
        for (int t = 0; t < 9; ++t) { sub.dd[t] = d->dd[9 * k + t]; sub.dh[t] = d->dh[9 * k + t]; sub.dw_[t] = d->dw_[9 * k + t]; }
        int h = 0;
        // (the plane's slab columns start 9 k Q floats into every slab row, like its columns of dw)
        if (int rc = gs_hwgrad_try2(&sub, a, g, a2, g2, dw + (size_t)9 * k * d->Q, ws ? ws + (size_t)9 * k * d->Q : nullptr,
                                    plan_only, stream, &h, tw)) return rc;
        if (!h) {
          GS_REQUIRE(k == 0, "gs_wgrad: depth plane %d of a 27-tap layer was refused after plane 0 ran", k);
          return 0;                                  // not eligible after all: the caller falls back for all 27 taps
        }
        *handled = h;
      }
      return 0;
    }
  }
  int dlo = 127, dhi = -128;
  for (int t = 0; t < d->T; ++t) { if (d->dd[t] < dlo) dlo = d->dd[t]; if (d->dd[t] > dhi) dhi = d->dd[t]; }
  if (enabled && wide_enabled && d->si == 1 && d->T == 9 && d->P % 64 == 0 && d->Q % 64 == 0 && d->P >= 64 &&
      d->Q >= 64 && (d->Da == 1 || (dlo == dhi && planes_enabled))) {
    int lo[3] = {127, 127, 127}, hi[3] = {-128, -128, -128};
    for (int t = 0; t < d->T; ++t) {
      const int o[3] = {d->dd[t], d->dh[t], d->dw_[t]};
      for (int ax = 0; ax < 3; ++ax) { if (o[ax] < lo[ax]) lo[ax] = o[ax]; if (o[ax] > hi[ax]) hi[ax] = o[ax]; }
    }
    HWGradK k;
    k.BD = 1; k.BH = 16; k.BW = 16;
    k.HD = k.BD + hi[0] - lo[0]; k.HH = k.BH + hi[1] - lo[1]; k.HW = k.BW + hi[2] - lo[2];
    k.dmin = lo[0]; k.hmin = lo[1]; k.wmin = lo[2];
    k.nbd = d->Da; k.nbh = (d->Ha + k.BH - 1) / k.BH; k.nbw = (d->Wa + k.BW - 1) / k.BW;   // BD = 1: a box per slice
    const int nimg = tw ? tw->n_split : d->N;          // images per network and operand tensor
    if (tw && 2 * nimg != d->N) return 0;
    const long long nboxes1 = (long long)nimg * k.nbd * k.nbh * k.nbw;
    const long long nboxes = a2 ? 2 * nboxes1 : nboxes1;  // per network
    const long long hv = (long long)k.HD * k.HH * k.HW;
    const int hbytes = (int)((hv * 144 + 1023) / 1024 * 1024 + 1024);
    const long long tab_bytes = ((long long)k.nbh * k.HH + (long long)k.nbw * k.HW) * 2;      // border tables, see the kernel
    const int lds = (int)(256 + 2 * (256 * 128 + hbytes) + (tab_bytes + 15) / 16 * 16);
    const int tiles = (d->P / 64) * (d->Q / 64);
    if (lds <= 160 * 1024 && k.HD == 1 && d->Hg < 32768 && d->Wg < 32768 && tab_bytes < 8192 && hv * 9 <= 4096 && nboxes >= 4 && nboxes < (1LL << 31) && tiles <= 65535 &&
        (long long)d->N * d->Dg * d->Hg * d->Wg * d->g_cs < (1LL << 31)) {
      k.nboxes = (int)nboxes;
      k.nboxes1 = (int)nboxes1;
      k.a2 = static_cast<const char*>(a2);
      k.g2 = static_cast<const char*>(g2);
      k.qchunks = k.phalves = 1;
      k.tgroups = 1;
      k.a = static_cast<const char*>(a);
      k.g = static_cast<const char*>(g);
      k.dw = dw;
      k.zero = static_cast<const char*>(gs_zero_page());
      GS_REQUIRE(k.zero || plan_only, "gs_wgrad: library not initialised (call gs_init)");
      k.d = *d;
      long long groups = 256 / (tiles * (tw ? 2 : 1));   // one workgroup per CU (twin: per network)
      if (groups < 1) groups = 1;
      if (groups > nboxes) groups = nboxes;
      k.nimg = nimg;
      k.gsplit = (int)groups;
      k.dw_delta = tw ? tw->dw_delta / 4 : 0;
      static bool configured = false;
      if (!configured) {
        GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hwgrad_wide_kernel<9>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
      }
      k.ws = ws;
      k.ws_stride = ws_stride;
      *handled = ws || plan_only ? (int)groups : 1;
      if (plan_only) return 0;
      hipLaunchKernelGGL((hwgrad_wide_kernel<9>), dim3((unsigned)(groups * (tw ? 2 : 1)), (unsigned)tiles), dim3(512), lds,
                         static_cast<hipStream_t>(stream), k);
      GS_CHECK_HIP(hipGetLastError());
      return 0;
    }
  }
  if (a2) return 0;                                // only the wide kernel merges two passes
  const int nimg_ft = tw ? tw->n_split : d->N;     // images per network
  if (tw && 2 * nimg_ft != d->N) return 0;
  // few taps, narrow on both sides (2-D W-folded k7 boundary convs): the (p, q)-split form
  if (enabled && gs_opt(GS_OPT_HWGRAD_FT) && d->si == 1 && d->T >= 2 && d->T <= 8 && d->Da == 1 && d->P <= 64 && d->Q <= 64 &&
      ((d->P + 15) / 16) * ((d->Q + 15) / 16) <= 8 && (long long)nimg_ft * ((d->Ha + 15) / 16) * ((d->Wa + 15) / 16) >= 512) {
    int lo[3] = {127, 127, 127}, hi[3] = {-128, -128, -128};
    for (int t = 0; t < d->T; ++t) {
      const int o[3] = {d->dd[t], d->dh[t], d->dw_[t]};
      for (int ax = 0; ax < 3; ++ax) { if (o[ax] < lo[ax]) lo[ax] = o[ax]; if (o[ax] > hi[ax]) hi[ax] = o[ax]; }
    }
    HWGradK k;
    k.BD = 1; k.BH = 16; k.BW = 16;
    k.HD = 1 + hi[0] - lo[0]; k.HH = k.BH + hi[1] - lo[1]; k.HW = k.BW + hi[2] - lo[2];
    k.dmin = lo[0]; k.hmin = lo[1]; k.wmin = lo[2];
    k.nbd = 1; k.nbh = (d->Ha + 15) / 16; k.nbw = (d->Wa + 15) / 16;
    const long long nboxes = (long long)nimg_ft * k.nbh * k.nbw;      // per network
    k.phalves = (d->P + 15) / 16;                  // np
    k.qchunks = (d->Q + 15) / 16;                  // nq
    k.tgroups = 1;
    const long long hv = (long long)k.HD * k.HH * k.HW;
    // (pitches padded by one piece per pixel / voxel, border tables behind the tiles: see the kernel)
    const long long apieces = 256LL * (k.phalves * 2 + 1), hpieces = hv * (k.qchunks * 2 + 1);
    const long long tab_bytes = ((long long)k.nbh * k.HH + (long long)k.nbw * k.HW) * 2;
    const int lds = 256 + (int)((apieces * 16 + 1023) / 1024 * 1024) + (int)((hpieces * 16 + 1023) / 1024 * 1024 + 1024) +
                    (int)((tab_bytes + 15) / 16 * 16);
    if (k.HD == 1 && lds <= 80 * 1024 && apieces <= 5 * 512 && hpieces <= 7 * 512 && tab_bytes < 8192 && d->Hg < 32768 &&
        d->Wg < 32768 && nboxes < (1LL << 31) &&
        (long long)d->N * d->Dg * d->Hg * d->Wg * d->g_cs < (1LL << 31)) {
      k.nboxes = (int)nboxes; k.nboxes1 = (int)nboxes;
      k.a2 = k.g2 = nullptr;
      k.a = static_cast<const char*>(a);
      k.g = static_cast<const char*>(g);
      k.dw = dw;
      k.zero = static_cast<const char*>(gs_zero_page());
      GS_REQUIRE(k.zero || plan_only, "gs_wgrad: library not initialised (call gs_init)");
      k.d = *d;
      // 512 = two co-resident workgroups per CU (<= 78 KB of LDS each): 61 + 12.6 us (kernel + slab reduction) against 69 + 7.4
      // with 256 and 81.6 + 6.1 for the im2col form (rocprofv3 averages over both k7 layers, profiles/r03_hwgrad_ft.txt)
      long long gmax = gs_opt(GS_OPT_HWGRAD_FT) > 1 ? gs_opt(GS_OPT_HWGRAD_FT) : 512;
      if (tw) gmax /= 2;                                         // (per network)
      const long long groups = nboxes < gmax ? nboxes : gmax;    // workgroups walking nboxes / groups boxes each
      k.nimg = nimg_ft;
      k.gsplit = (int)groups;
      k.dw_delta = tw ? tw->dw_delta / 4 : 0;
      k.ws = ws;
      k.ws_stride = ws_stride;
      *handled = ws || plan_only ? (int)groups : 1;
      if (plan_only) return 0;
      static bool configured = false;
      if (!configured) {
        GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hwgrad_ft_kernel<8>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        configured = true;
      }
      hipLaunchKernelGGL((hwgrad_ft_kernel<8>), dim3((unsigned)(groups * (tw ? 2 : 1))), dim3(512), lds,
                         static_cast<hipStream_t>(stream), k);
      GS_CHECK_HIP(hipGetLastError());
      return 0;
    }
  }
  if (tw) return 0;                                // hwgrad_kernel below has no twin form
  // one-channel volume layers (PatchGAN3D's last conv: P = 8, Q = 256, 64 taps): 16 channel chunks of the gathered side, each
  // with its own workgroups (option hwgrad2 >= 2) — the im2col kernel gathered 64 taps x 512 bytes per voxel through L2 (715 us)
  const bool thin = gs_opt(GS_OPT_HWGRAD2) >= 2 && d->P <= 16 && d->Da > 1 && d->T >= 27 && d->Q <= 512;
  if (!enabled || d->si != 1 || d->P > 64 || (d->Q > 64 && !thin) || d->T < 9) return 0;
  // wide on both sides: the im2col kernel was the better fit for hwgrad_kernel (measured, round 1); the double-buffered form
  // takes 64 <-> 64 channel volume layers too (option hwgrad2 >= 2)
  if (d->Q > 32 && d->P > 16 && !(gs_opt(GS_OPT_HWGRAD2) >= 2 && d->Da > 1 && d->T > 64 && d->T <= 128)) return 0;
  int lo[3] = {127, 127, 127}, hi[3] = {-128, -128, -128};
  for (int t = 0; t < d->T; ++t) {
    const int o[3] = {d->dd[t], d->dh[t], d->dw_[t]};
    for (int ax = 0; ax < 3; ++ax) { if (o[ax] < lo[ax]) lo[ax] = o[ax]; if (o[ax] > hi[ax]) hi[ax] = o[ax]; }
  }
  HWGradK k;
  if (d->Da > 1) { k.BD = 4; k.BH = 8; k.BW = 8; } else { k.BD = 1; k.BH = 16; k.BW = 16; }
  k.HD = k.BD + hi[0] - lo[0]; k.HH = k.BH + hi[1] - lo[1]; k.HW = k.BW + hi[2] - lo[2];
  k.dmin = lo[0]; k.hmin = lo[1]; k.wmin = lo[2];
  k.nbd = (d->Da + k.BD - 1) / k.BD; k.nbh = (d->Ha + k.BH - 1) / k.BH; k.nbw = (d->Wa + k.BW - 1) / k.BW;
  const long long nboxes = (long long)d->N * k.nbd * k.nbh * k.nbw;
  if (nboxes <= 0 || nboxes >= (1LL << 31)) return 0;
  const int TI = d->P <= 16 ? 1 : 2;
  k.phalves = (d->P + TI * 16 - 1) / (TI * 16);
  k.qchunks = (d->Q + 15) / 16;
  const int TPW = d->T <= 64 ? 8 : 16;             // spread few taps over all 8 waves
  k.tgroups = (d->T + 8 * TPW - 1) / (8 * TPW);
  const long long hv = (long long)k.HD * k.HH * k.HW;
  const int lds = 512 + (256 * TI * 32 + 1023) / 1024 * 1024 + (int)((hv * 32 + 1023) / 1024 * 1024 + 1024);
  if (lds > 96 * 1024) return 0;
  if ((long long)d->N * d->Dg * d->Hg * d->Wg * d->g_cs >= (1LL << 31)) return 0;
  k.nboxes = (int)nboxes;
  k.nboxes1 = (int)nboxes;
  k.a2 = k.g2 = nullptr;
  k.a = static_cast<const char*>(a);
  k.g = static_cast<const char*>(g);
  k.dw = dw;
  k.zero = static_cast<const char*>(gs_zero_page());
  GS_REQUIRE(k.zero || plan_only, "gs_wgrad: library not initialised (call gs_init)");
  k.d = *d;
  // one workgroup per CU (its registers hold 16 taps per wave); ~2 rounds of box groups keep the tail short
  const int per_y = k.qchunks * k.phalves * k.tgroups;
  // volumes with 65..128 taps: the double-buffered form (hwgrad2_kernel) — one workgroup per CU walks its boxes with the next one
  // staged under the current one's MFMAs
  const long long tab_bytes2 = ((long long)k.nbd * k.HD + (long long)k.nbh * k.HH + (long long)k.nbw * k.HW) * 2;
  const int lds2 = 512 + 2 * (256 * TI * 32 + (int)((hv * 32 + 1023) / 1024 * 1024 + 1024)) + (int)((tab_bytes2 + 15) / 16 * 16);
  if (gs_opt(GS_OPT_HWGRAD2) && TPW == 16 && k.tgroups == 1 && d->Da > 1 && hv * 2 <= 5 * 512 && lds2 <= 160 * 1024 &&
      tab_bytes2 < 16384 && d->Dg < 32768 && d->Hg < 32768 && d->Wg < 32768) {
    long long groups2 = 256 / per_y;
    if (groups2 < 1) groups2 = 1;
    if (groups2 > nboxes) groups2 = nboxes;
    k.ws = ws;
    k.ws_stride = ws_stride;
    *handled = ws || plan_only ? (int)groups2 : 1;
    if (plan_only) return 0;
    const dim3 grid2((unsigned)groups2, (unsigned)per_y);
    hipStream_t st2 = static_cast<hipStream_t>(stream);
    static bool configured2[2] = {false, false};
    if (TI == 1) {
      if (!configured2[0]) {
        GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hwgrad2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured2[0] = true;
      }
      hipLaunchKernelGGL((hwgrad2_kernel<1>), grid2, dim3(512), lds2, st2, k);
    } else {
      if (!configured2[1]) {
        GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hwgrad2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured2[1] = true;
      }
      hipLaunchKernelGGL((hwgrad2_kernel<2>), grid2, dim3(512), lds2, st2, k);
    }
    GS_CHECK_HIP(hipGetLastError());
    return 0;
  }
  long long groups = 512 / per_y;
  if (groups < 1) groups = 1;
  if (groups > nboxes) groups = nboxes;
  k.ws = ws;
  k.ws_stride = ws_stride;
  *handled = ws || plan_only ? (int)groups : 1;
  if (plan_only) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)st;
  const dim3 grid((unsigned)groups, (unsigned)per_y);
  if (TI == 1) return TPW == 8 ? launch_hw<1, 8>(k, grid, lds, st) : launch_hw<1, 16>(k, grid, lds, st);
  return TPW == 8 ? launch_hw<2, 8>(k, grid, lds, st) : launch_hw<2, 16>(k, grid, lds, st);
}
