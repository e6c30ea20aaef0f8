// Register-resident-weights kernel for the 16 -> 16 channel 5x5x5 stride-1 convolutions of Vnet3D's additive couplings at full
// resolution (ganslate/nn/generators/vnet/vnet3d.py:262-267 through nn/invertible.py:8-48; forward and data gradient: the same
// geometry with mirrored taps).
//
// hconv_kernel runs this layer LDS-read bound: with 16 output channels the MFMA's M side is one 16-row tile, so every voxel
// fragment (16 voxels x 32 k) read from LDS feeds exactly ONE v_mfma_f32_16x16x32_bf16 — one ds_read_b128 per MFMA, 668 TFLOP/s
// at 128^3 (profiles/r05_brats_by_grid_v1.txt: 200.9 us per launch). The whole weight set of the layer is 16 x 2000 bf16 = 64 KB:
// it FITS THE REGISTER FILE of a wave that has a SIMD to itself (512 registers: 75 A fragments = 300 registers), and then a voxel
// fragment feeds every tap that can use it:
//   * K = 32 of one MFMA = two y-adjacent taps x 16 input channels: tap pairs (dy -2,-1), (0,1) and (2, none: zero weights,
//     partner = the same row so that the padding operand is finite data) for each (dz, dx) -> 15 "units" per dx, 75 in all;
//   * a wave owns 4 x 8 rows (z, y) of 16 voxels in x: 32 accumulator tiles = 128 registers. A fragment of the input rows
//     (z', y' | y'+1) at x window dx contributes to the rows (z' - dz, y' - dy0) for up to 5 dz x 2 dy0: 3.3 MFMAs per
//     fragment read on average (2400 MFMAs against 720 ds_read_b128 per wave and tile) — the loop is MFMA-issue bound.
// Workgroup = 4 waves (one per SIMD) on one 8 x 16 x 16 box: halo box 12 x 20 x 20 voxels x 32 B = 150 KB of LDS, staged once per
// tile by LDS-DMA (border handling in the per-lane source address); persistent workgroups walk tiles b, b + grid, ... and load
// the weights once. Everything is statically unrolled (the accumulator a fragment feeds is a compile-time register).
// Epilogue contract of hconv_kernel: bias, per-box InstanceNorm partial sums (one slot per box), activation, accumulate-into,
// channel-slice views.
#include "common.hpp"
#include <type_traits>
#include <utility>

namespace {
struct HConv5K {
  const char* in;
  const char* w;
  const float* bias;
  char* out;
  float* stats;
  const char* zero;
  int nbd, nbh, nbw;     // boxes per axis
  int ntiles;            // N * nbd * nbh * nbw
  gs_gconv_desc d;
};

constexpr int BZ = 8, BY = 16, BX = 16;          // output box of a workgroup
constexpr int HZ = BZ + 4, HY = BY + 4, HX = BX + 4;
constexpr int VP = 32;                           // bytes per halo voxel (16 channels)
constexpr int ROWB = HX * VP;                    // 640
constexpr int PLANEB = HY * ROWB;                // 12800
constexpr int HALO_BYTES = HZ * PLANEB;          // 153600
constexpr int HPIECES = HZ * HY * HX * 2;        // 16-B pieces
constexpr int HINSTR = HPIECES / 64;             // 150 (exact)
static_assert(HPIECES % 64 == 0, "whole LDS-DMA instructions");
constexpr int LZ = 4, LY = 8;                    // rows of a wave
constexpr int NRING = 6;                         // voxel fragments in flight

// MFMA through inline asm with the A operand's register class spelled out: the weights must live in BOTH halves of the
// unified register file (64 of the 75 fragments in accumulation registers, the rest beside the accumulators in the
// architectural ones). Left to the compiler every A operand is an architectural register: 17 fragments went to scratch memory
// and 50 more were copied out of AGPR "spill slots" before each use. The hardware interlocks SrcC on the previous MFMA's result;
// nothing here feeds an MFMA result to an A / B operand, and the epilogue waits out the last MFMA before VALU touches acc.
constexpr int A_IN_AGPR = 64;
__device__ __forceinline__ void mfma_a(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int U>
__device__ __forceinline__ void mfma_u(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (U < A_IN_AGPR) mfma_a(acc, a, b); else mfma_v(acc, a, b);
}

template <int... I, class F>
__device__ __forceinline__ void unroll_seq(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

// fragment f of a wave's walk: x window dxi, input plane zp (0..7 = wave z base - 2 ..), kind / row
constexpr int FR_PER_PLANE = (LY + 2) + LY;      // 10 paired fragments (y' = -2 .. LY-1) + 8 single ones (y' = 2 .. LY+1)
constexpr int NFRAG = 5 * (LZ + 4) * FR_PER_PLANE;
struct FragId { int dxi, zp, single, yp; };
constexpr FragId frag_of(int f) {
  const int k = f % FR_PER_PLANE, r = f / FR_PER_PLANE;
  return FragId{r / (LZ + 4), r % (LZ + 4), k >= LY + 2 ? 1 : 0, k >= LY + 2 ? (k - (LY + 2)) + 2 : k - 2};
}

__global__ __launch_bounds__(256) void hconv5_kernel(const HConv5K p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;                                           // [HZ][HY][HX][32 B]
  int* lut = reinterpret_cast<int*>(smem + HALO_BYTES);        // [125] tap index of (dz, dy, dx), -1 = absent
  float* red = reinterpret_cast<float*>(smem + HALO_BYTES + 512);   // [4 waves][16 channels][2]
  const gs_gconv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;
  const int zq = wave >> 1, yq = wave & 1;

  for (int t = tid; t < 125; t += 256) lut[t] = -1;
  __syncthreads();
  for (int t = tid; t < d.T; t += 256)
    lut[((int)d.dd[t] + 2) * 25 + ((int)d.dh[t] + 2) * 5 + ((int)d.dw[t] + 2)] = t;
  __syncthreads();

  // ---- the layer's weights: 75 A fragments, lane (row co = col, k group kg): tap member kg >> 1 of the pair, channels
  // (kg & 1) * 8 .. + 8 of pack row co (tap-major K: t * Ci + ci) ------------------------------------------------------------
  bf16x8 A[5][3][5];                               // [dz][dy pair][dx]
  {
    const bool row_ok = col < d.w_rows;
    const char* wrow = p.w + ((size_t)col * d.Kp + (kg & 1) * 8) * 2;
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          const int dy = b * 2 + (kg >> 1);        // 0..5 (5: the missing partner of dy = +2)
          const int t = dy < 5 ? lut[a * 25 + dy * 5 + c] : -1;
          const char* src = (row_ok && t >= 0) ? wrow + (size_t)t * d.Ci * 2 : p.zero;
          A[a][b][c] = *reinterpret_cast<const bf16x8*>(src);
        }
  }

  // per-lane halo byte offsets of a fragment read: voxel column col (+ window), channel half kg & 1, partner row for kg >= 2
  const unsigned lane_b = (unsigned)(col * VP + (kg & 1) * 16);
  const unsigned halo0 = lds_addr(halo) + (unsigned)((zq * LZ) * PLANEB + (yq * LY) * ROWB);
  // (two bases per kind: planes 0-4 and 5-7 of the wave's window, so that the rest of the offset fits the ds_read immediate)
  const unsigned a_pair = halo0 + lane_b + (unsigned)((kg >> 1) * ROWB), a_pair_hi = a_pair + 5 * PLANEB;
  const unsigned a_single = halo0 + lane_b, a_single_hi = a_single + 5 * PLANEB;

  const int tiles_per_img = p.nbd * p.nbh * p.nbw;
#pragma clang loop unroll(disable)
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int n = tile / tiles_per_img;
    int b = tile - n * tiles_per_img;
    const int box = b;
    const int bx = b % p.nbw; b /= p.nbw;
    const int by = b % p.nbh;
    const int bz = b / p.nbh;
    const int oz0 = bz * BZ, oy0 = by * BY, ox0 = bx * BX;
    // ---- stage the halo box: one 16-B piece per lane per LDS-DMA instruction --------------------------------------------------
    __syncthreads();                              // the previous tile's reads are done
    {
      const char* in_n = p.in + ((size_t)n * d.Di * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
      for (int inst = wave; inst < HINSTR; inst += 4) {
        const int q = inst * 64 + lane;
        const int v = q >> 1, part = q & 1;
        const int hz = v / (HY * HX), r2 = v - hz * (HY * HX);
        const int hy = r2 / HX, hx = r2 - hy * HX;
        bool ok = true;
        int iz = border_index(oz0 + hz - 2, d.Di, d.border, ok);
        int iy = border_index(oy0 + hy - 2, d.Hi, d.border, ok);
        int ix = border_index(ox0 + hx - 2, d.Wi, d.border, ok);
        iz = min(max(iz, 0), d.Di - 1);
        iy = min(max(iy, 0), d.Hi - 1);
        ix = min(max(ix, 0), d.Wi - 1);
        unsigned off = ((unsigned)((iz * d.Hi + iy) * d.Wi + ix) * (unsigned)d.in_cs + (unsigned)(part * 8)) * 2u;
        asm volatile("" : "+v"(off));
        const char* src = ok ? in_n + off : p.zero;
        glds16(src, halo + inst * 1024);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 acc[LZ][LY];
#pragma unroll
    for (int z = 0; z < LZ; ++z)
#pragma unroll
      for (int y = 0; y < LY; ++y) acc[z][y] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- all taps out of registers: fragment f is read NRING fragments ahead of its MFMAs --------------------------------------
    bf16x8 ring[NRING];
    auto issue = [&](auto f_tag) {
      constexpr int f = decltype(f_tag)::value;
      if constexpr (f < NFRAG) {
        constexpr FragId id = frag_of(f);
        // halo row of the fragment: plane zp, row yp + 2 (halo coordinates of the wave's window), x window dxi
        constexpr int off = (id.zp % 5) * PLANEB + (id.yp + 2) * ROWB + id.dxi * VP;
        static_assert(off >= 0 && off < 65536 - 16, "ds_read immediate");
        lds_read128<off>(ring[f % NRING], id.zp < 5 ? (id.single ? a_single : a_pair) : (id.single ? a_single_hi : a_pair_hi));
      }
    };
    unroll_seq(std::make_integer_sequence<int, NRING - 1>{}, issue);
    unroll_seq(std::make_integer_sequence<int, NFRAG>{}, [&](auto f_tag) {
      constexpr int f = decltype(f_tag)::value;
      constexpr FragId id = frag_of(f);
      issue(std::integral_constant<int, f + NRING - 1>{});
      // fragments are consumed in issue order: at most NRING - 1 younger reads may still be in flight
      constexpr int younger = (NFRAG - 1 - f) < (NRING - 1) ? (NFRAG - 1 - f) : (NRING - 1);
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[f % NRING]) : "i"(younger) : "memory");
      constexpr int zi = id.zp - 2;                // input plane relative to the wave's first output plane
      unroll_seq(std::make_integer_sequence<int, 5>{}, [&](auto a_tag) {      // dz = a - 2: output plane zi - dz
        constexpr int a = decltype(a_tag)::value;
        constexpr int zr = zi - (a - 2);
        if constexpr (zr >= 0 && zr < LZ) {
          if constexpr (id.single) {               // dy0 = +2: output row yp - 2
            mfma_u<(a * 3 + 2) * 5 + id.dxi>(acc[zr][id.yp - 2], A[a][2][id.dxi], ring[f % NRING]);
          } else {
            if constexpr (id.yp + 2 >= 0 && id.yp + 2 < LY)    // dy0 = -2: output row yp + 2
              mfma_u<(a * 3 + 0) * 5 + id.dxi>(acc[zr][id.yp + 2], A[a][0][id.dxi], ring[f % NRING]);
            if constexpr (id.yp >= 0 && id.yp < LY)            // dy0 = 0: output row yp
              mfma_u<(a * 3 + 1) * 5 + id.dxi>(acc[zr][id.yp], A[a][1][id.dxi], ring[f % NRING]);
          }
        }
      });
    });
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMA's result is architecturally visible to VALU

    // ---- epilogue: bias, partial statistics (one slot per box), activation, [accumulate], 8-B NDHWC stores ------------------------
    const bool want_stats = d.stats_slots > 0;
    const int co = kg * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = co + r < d.Co ? p.bias[co + r] : 0.f;
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    char* out_n = p.out + (size_t)n * d.Do * d.Ho * d.Wo * d.out_cs * 2;
#pragma unroll
    for (int z = 0; z < LZ; ++z)
#pragma unroll
      for (int y = 0; y < LY; ++y) {
        const int oz = oz0 + zq * LZ + z, oy = oy0 + yq * LY + y, ox = ox0 + col;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[z][y][r] + bv[r];
          s1[r] += v[r];
          s2[r] += v[r] * v[r];
          v[r] = apply_act_small(v[r], d.act, d.slope);
        }
        if (co < d.Co) {
          uint2* dst = reinterpret_cast<uint2*>(out_n + ((size_t)((oz * d.Ho + oy) * d.Wo + ox) * d.out_cs + d.out_co + co) * 2);
          uint2 o;
          o.x = pack_bf2(v[0], v[1]);
          o.y = pack_bf2(v[2], v[3]);
          if (d.accumulate) {     // bf16 read-modify-write, same rounding points as gconv_kernel / hconv_kernel
            const uint2 old = *dst;
            o.x = pack_bf2(bf_lo(o.x) + bf_lo(old.x), bf_hi(o.x) + bf_hi(old.x));
            o.y = pack_bf2(bf_lo(o.y) + bf_lo(old.y), bf_hi(o.y) + bf_hi(old.y));
          }
          *dst = o;
        }
      }
    if (want_stats) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s1[r]), q = row16_sum(s2[r]);
        if (col == 0) {
          red[(wave * 16 + co + r) * 2 + 0] = a;
          red[(wave * 16 + co + r) * 2 + 1] = q;
        }
      }
      __syncthreads();
      if (tid < 16 && tid < d.Co) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += red[(w * 16 + tid) * 2]; q += red[(w * 16 + tid) * 2 + 1]; }
        float* sp = p.stats + (((size_t)n * d.stats_slots + d.stats_slot0 + box) * 2) * d.Co;
        sp[tid] = a;
        sp[d.Co + tid] = q;
      }
    }
  }
}

bool hconv5_eligible(const gs_gconv_desc* d) {
  if (!gs_opt(GS_OPT_HCONV5)) return false;
  if (d->T != 125 || d->Ci != 16 || d->Co > 16 || d->Co < 8 || d->si != 1 || d->so != 1) return false;
  if (d->Dc != d->Do || d->Hc != d->Ho || d->Wc != d->Wo || d->pz || d->py || d->px) return false;
  if (d->Do % BZ || d->Ho % BY || d->Wo % BX || d->Di != d->Do || d->Hi != d->Ho || d->Wi != d->Wo) return false;
  for (int t = 0; t < 125; ++t)
    if (d->dd[t] < -2 || d->dd[t] > 2 || d->dh[t] < -2 || d->dh[t] > 2 || d->dw[t] < -2 || d->dw[t] > 2) return false;
  const long long tiles = (long long)d->N * (d->Do / BZ) * (d->Ho / BY) * (d->Wo / BX);
  if (tiles < gs_opt(GS_OPT_HCONV5) || tiles >= (1LL << 31)) return false;      // small volumes: hconv_kernel's 512-voxel boxes fill the chip
  if ((long long)d->Di * d->Hi * d->Wi * d->in_cs * 2 >= (1LL << 32)) return false;
  if ((long long)d->Do * d->Ho * d->Wo >= (1LL << 31)) return false;
  return true;
}
}  // namespace

// partial-statistics slots per image when the layer runs here (one per 8 x 16 x 16 box), 0 when it does not
int gs_hconv5_slots(const gs_gconv_desc* d) {
  return hconv5_eligible(d) ? (d->Do / BZ) * (d->Ho / BY) * (d->Wo / BX) : 0;
}

// returns 0 and sets *handled when the layer ran here
int gs_hconv5_try(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out, float* stats,
                  void* stream, int* handled) {
  *handled = 0;
  if (!hconv5_eligible(d)) return 0;
  HConv5K k;
  k.in = static_cast<const char*>(in);
  k.w = static_cast<const char*>(w_pack);
  k.bias = bias;
  k.out = static_cast<char*>(out);
  k.stats = stats;
  k.zero = static_cast<const char*>(gs_zero_page());
  GS_REQUIRE(k.zero, "gs_gconv_forward: library not initialised (call gs_init)");
  k.nbd = d->Do / BZ; k.nbh = d->Ho / BY; k.nbw = d->Wo / BX;
  k.ntiles = d->N * k.nbd * k.nbh * k.nbw;
  k.d = *d;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  // persistent: equal shares of whole tiles (1024 tiles on 256 CUs: 4 each)
  const int per = (k.ntiles + cus - 1) / cus;
  const int grid = (k.ntiles + per - 1) / per;
  constexpr int lds = HALO_BYTES + 512 + 4 * 16 * 2 * 4;
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconv5_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  *handled = 1;
  hipLaunchKernelGGL(hconv5_kernel, dim3(grid), dim3(256), lds, static_cast<hipStream_t>(stream), k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
