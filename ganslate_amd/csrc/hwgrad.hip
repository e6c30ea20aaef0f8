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

struct HWGradK {
  const char* a;
  const char* g;
  float* dw;
  const char* zero;
  int BD, BH, BW;        // pixel box
  int HD, HH, HW;        // halo box
  int dmin, hmin, wmin;
  int nbd, nbh, nbw;
  int nboxes;            // N * nbd * nbh * nbw
  int qchunks, phalves, tgroups;
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
          if (t < ntaps) {                         // wave-uniform
            const char* gp = halo + ((size_t)(rb + toff[wave * TPW + t]) * 16 + fcc * 4) * 2;
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
          if (pp < d.P) unsafeAtomicAdd(p.dw + (size_t)pp * d.dw_ld + tap * d.Q + q, acc[t][i][r]);
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
int gs_hwgrad_try(const gs_wgrad_desc* d, const void* a, const void* g, float* dw, void* stream, int* handled) {
  *handled = 0;
  static const bool enabled = !(getenv("GS_HWGRAD") && atoi(getenv("GS_HWGRAD")) == 0);
  if (!enabled || d->si != 1 || d->P > 64 || d->Q > 64 || d->T < 9) return 0;
  if (d->Q > 32 && d->P > 16) return 0;            // wide on both sides: the im2col kernel is the better fit (measured)
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
  k.a = static_cast<const char*>(a);
  k.g = static_cast<const char*>(g);
  k.dw = dw;
  k.zero = static_cast<const char*>(gs_zero_page());
  GS_REQUIRE(k.zero, "gs_wgrad: library not initialised (call gs_init)");
  k.d = *d;
  // one workgroup per CU (its registers hold 16 taps per wave); ~2 rounds of box groups keep the tail short
  const int per_y = k.qchunks * k.phalves * k.tgroups;
  long long groups = 512 / per_y;
  if (groups < 1) groups = 1;
  if (groups > nboxes) groups = nboxes;
  *handled = 1;
  const dim3 grid((unsigned)groups, (unsigned)per_y);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (TI == 1) return TPW == 8 ? launch_hw<1, 8>(k, grid, lds, st) : launch_hw<1, 16>(k, grid, lds, st);
  return TPW == 8 ? launch_hw<2, 8>(k, grid, lds, st) : launch_hw<2, 16>(k, grid, lds, st);
}
