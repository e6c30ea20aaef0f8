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
//   * a fragment of the input rows (z', y' | y'+1) at x window dx contributes to the output rows (z' - dz, y' - dy0) a wave
//     owns for up to 5 dz x 2 dy0: 3 MFMAs per fragment read on average — the loop is MFMA-issue bound, not LDS-read bound.
// Workgroup = 4 waves (one per SIMD) on a COLUMN of 16 x 16 voxels in (y, x) that walks along z, four output planes per step:
// a ring of 12 input planes (20 x 20 halo voxels x 32 B each, 150 KB of LDS) holds the 8 planes a step reads while the 4 planes
// the next step adds are staged by LDS-DMA under the step's MFMA stream (border handling in the source address) — nothing of the
// staging is exposed after a column's first step, and the z halo is read once per column segment instead of once per box (input
// re-read 1.7x instead of 2.3x). A wave owns the 4 planes x 4 rows (its quarter of y) of a step: 16 accumulator tiles, 400
// fragment reads for 1200 MFMAs. The volume's columns are cut into z segments so that every CU gets a workgroup; persistent
// workgroups walk segments b, b + grid, ... and load the weights once. The MFMA stream is statically unrolled (the accumulator
// a fragment feeds is a compile-time register).
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
  int nbh, nbw;          // columns per axis
  int nseg, seg_steps;   // z segments per column, steps (of 4 planes) per segment
  int nwork;             // N * nbh * nbw * nseg
  gs_gconv_desc d;
};

constexpr int SZ = 4, BY = 16, BX = 16;          // output planes per step, column footprint
constexpr int HY = BY + 4, HX = BX + 4;
constexpr int VP = 32;                           // bytes per halo voxel (16 channels)
constexpr int ROWB = HX * VP;                    // 640
constexpr int PLANEB = HY * ROWB;                // 12800
constexpr int NSLOT = 12;                        // ring of input planes: 8 in use + 4 arriving
constexpr int HALO_BYTES = NSLOT * PLANEB;       // 153600
constexpr int LZ = 4, LY = 4;                    // rows of a wave: all planes of the step x its quarter of y
constexpr int NRING = 8;                         // voxel fragments in flight

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
#ifdef HCONV5_BUILTIN
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
#else
  if constexpr (U < A_IN_AGPR) mfma_a(acc, a, b); else mfma_v(acc, a, b);
#endif
}

template <int... I, class F>
__device__ __forceinline__ void unroll_seq(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

// Fragments of a wave's walk through a step, per x window: K = 32 of an MFMA is two taps x 16 channels, and the 25 (dz, dy) taps
// of an x offset pair up into 13 units (96 % of the K slots used; 15 units with y pairs only):
//   kind P  rows (z', y') | (z', y'+1):  units (dz, dy pair) for dy in (-2,-1), (0,+1)        10 units, 48 fragments
//   kind Z  rows (z', y') | (z'+1, y'):  the dy = +2 taps paired along z, dz in (-2,-1), (0,+1)   2 units, 24 fragments
//   kind S  row  (z', y') twice:         the tap (dz, dy) = (+2, +2) (partner weights zero)        1 unit,  16 fragments
// zp = input plane (0..7 = first output plane - 2 ..), yp = input row relative to the wave's first output row. Heavy (P) and
// light (Z, S: one or two MFMAs) fragments alternate, P fragments with the plane index fastest: any 8 consecutive fragments carry
// >= 16 MFMAs, which keeps the fragment prefetch (NRING - 1 reads ahead) beyond the LDS latency.
struct FragId { int dxi, kind, zp, yp; };
constexpr int NP = (LZ + 4) * (LY + 2), NZ = (LZ + 2) * LY, NS = LZ * LY, FR_PER_DX = NP + NZ + NS;   // 48 + 24 + 16
constexpr int NFRAG = 5 * FR_PER_DX;
constexpr FragId frag_of(int f) {
  const int dxi = f / FR_PER_DX, r = f % FR_PER_DX;
  constexpr int NL = NZ + NS;                      // light fragments: after each of the first NL heavy ones
  int heavy = r - NL, light = 0;
  bool is_light = false;
  if (r < 2 * NL) { heavy = r / 2; light = r / 2; is_light = r & 1; }
  if (!is_light) return FragId{dxi, 0, heavy % (LZ + 4), heavy / (LZ + 4) - 2};
  if (light < NZ) return FragId{dxi, 1, light % (LZ + 2), light / (LZ + 2) + 2};
  return FragId{dxi, 2, (light - NZ) % LZ + 4, (light - NZ) / LZ + 2};
}
constexpr int NUNIT = 13 * 5;                      // A fragments

__global__ __launch_bounds__(256) void hconv5_kernel(const HConv5K p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;                                           // [NSLOT][HY][HX][32 B]
  int* lut = reinterpret_cast<int*>(smem + HALO_BYTES);        // [125] tap index of (dz, dy, dx), -1 = absent
  float* red = reinterpret_cast<float*>(smem + HALO_BYTES + 512);   // [4 waves][16 channels][2]
  const gs_gconv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;

  for (int t = tid; t < 125; t += 256) lut[t] = -1;
  __syncthreads();
  for (int t = tid; t < d.T; t += 256)
    lut[((int)d.dd[t] + 2) * 25 + ((int)d.dh[t] + 2) * 5 + ((int)d.dw[t] + 2)] = t;
  __syncthreads();

  // ---- the layer's weights: 65 A fragments, lane (row co = col, k group kg): tap member kg >> 1 of the pair, channels
  // (kg & 1) * 8 .. + 8 of pack row co (tap-major K: t * Ci + ci). All tap lookups first, then the loads back to back. -----------
  bf16x8 A[13][5];                                 // [unit][dx]: units 0-9 = P (dz * 2 + dy pair), 10-11 = Z (dz pair), 12 = S
  {
    const bool row_ok = col < d.w_rows;
    const char* wrow = p.w + ((size_t)col * d.Kp + (kg & 1) * 8) * 2;
    const int m = kg >> 1;                         // member of the tap pair this lane's k group belongs to
    int tt[13][5];
#pragma unroll
    for (int u = 0; u < 13; ++u)
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        int dzi, dyi;                              // tap (dz + 2, dy + 2) of this lane's member; dzi = -1: none (zero weights)
        if (u < 10) { dzi = u >> 1; dyi = (u & 1) * 2 + m; }
        else if (u < 12) { dzi = (u - 10) * 2 + m; dyi = 4; }
        else { dzi = m ? -1 : 4; dyi = 4; }
        tt[u][c] = dzi >= 0 ? lut[dzi * 25 + dyi * 5 + c] : -1;
      }
    // (two batches with a wait behind each: vmcnt counts to 63 — with all the loads and the first planes' 40 staging instructions
    // in flight the counter wrapped and the first step's vmcnt(0) let the MFMAs start on weights that had not arrived)
#pragma unroll
    for (int u = 0; u < 13; ++u) {
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const int t = tt[u][c];
        const char* src = (row_ok && t >= 0) ? wrow + (size_t)t * d.Ci * 2 : p.zero;
        A[u][c] = *reinterpret_cast<const bf16x8*>(src);
      }
      if (u == 6 || u == 12) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }

  // per-lane byte offset of a fragment read inside a plane: voxel column col (+ window), channel half kg & 1, row of the wave's
  // quarter (+ the partner row for kg >= 2 of a paired fragment)
  const unsigned halo0 = lds_addr(halo);
  const unsigned lane_single = (unsigned)((wave * LY) * ROWB + col * VP + (kg & 1) * 16);
  const unsigned lane_pair = lane_single + (unsigned)((kg >> 1) * ROWB);

  const int cols_per_img = p.nbh * p.nbw;
#pragma clang loop unroll(disable)
  for (int work = blockIdx.x; work < p.nwork; work += gridDim.x) {
    int b = work;
    const int seg = b % p.nseg; b /= p.nseg;
    const int bx = b % p.nbw; b /= p.nbw;
    const int by = b % p.nbh;
    const int n = b / p.nbh;
    const int oy0 = by * BY, ox0 = bx * BX;
    const int zs0 = seg * p.seg_steps * SZ;                              // first output plane of the segment
    const int nsteps = min(p.seg_steps, (d.Do - zs0 + SZ - 1) / SZ);
    const char* in_n = p.in + ((size_t)n * d.Di * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
    // ---- staging: a halo row = 40 lanes x 16 B (voxel hx = lane >> 1, half lane & 1) by one LDS-DMA instruction; this wave stages
    // rows wave, wave + 4, ... of every plane. Source = plane offset + row offset + lane offset, each resolved once: the lane's
    // x at the top of the column, the wave's five rows too, a plane's z when it is staged. 0xffffffff = outside a zero border.
    const bool x_lane = lane < 2 * HX;
    unsigned x_off;
    {
      bool ok = true;
      int ix = border_index(ox0 + (lane >> 1) - 2, d.Wi, d.border, ok);
      ix = min(max(ix, 0), d.Wi - 1);
      x_off = ok ? (unsigned)(ix * d.in_cs + (lane & 1) * 8) * 2u : 0xffffffffu;
    }
    unsigned row_off[HY / 4];
#pragma unroll
    for (int j = 0; j < HY / 4; ++j) {
      bool ok = true;
      int iy = border_index(oy0 + wave + 4 * j - 2, d.Hi, d.border, ok);
      iy = min(max(iy, 0), d.Hi - 1);
      row_off[j] = ok ? (unsigned)(iy * d.Wi * d.in_cs) * 2u : 0xffffffffu;
    }
    auto plane_off = [&](int pl) {                 // input plane number pl of the segment: z = zs0 - 2 + pl
      bool ok = true;
      int iz = border_index(zs0 - 2 + pl, d.Di, d.border, ok);
      iz = min(max(iz, 0), d.Di - 1);
      return ok ? (unsigned)(iz * d.Hi * d.Wi * d.in_cs) * 2u : 0xffffffffu;
    };
    auto stage_row = [&](unsigned poff, int slot, auto j_tag) {
      constexpr int j = decltype(j_tag)::value;
      const bool ok = poff != 0xffffffffu && row_off[j] != 0xffffffffu && x_off != 0xffffffffu;
      const char* src = ok ? in_n + (poff + row_off[j] + x_off) : p.zero;
      if (x_lane) glds16(src, halo + slot * PLANEB + (wave + 4 * j) * ROWB);
    };
    __syncthreads();                              // the previous segment's reads are done
    for (int pl = 0; pl < 8; ++pl) {
      const unsigned poff = plane_off(pl);
      unroll_seq(std::make_integer_sequence<int, HY / 4>{}, [&](auto j_tag) { stage_row(poff, pl % NSLOT, j_tag); });
    }

    bool counted = false;                         // the previous step issued exactly NST output stores behind its staging
#pragma clang loop unroll(disable)
    for (int step = 0; step < nsteps; ++step) {
      // this step's planes have landed (this wave's share): everything but the previous step's output stores, which are younger
      // than its staging instructions and drain under this step (VMEM retires in issue order)
      if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LZ * LY) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                          // ... everybody's; the previous step's reads are done
      asm volatile("" ::: "memory");
      // the 4 planes the next step adds go into the slots this step does not read; their 20 row instructions are spread over
      // the first half of the MFMA stream
      const bool stage_next = step + 1 < nsteps;
      unsigned npoff[4];
      int nslot[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { npoff[k] = plane_off(4 * step + 8 + k); nslot[k] = (4 * step + 8 + k) % NSLOT; }
      // plane bases of this step: input plane zp of the step sits in slot (4 * step + zp) % NSLOT
      unsigned pbase[LZ + 4], zbase[LZ + 2];        // zbase: plane zp for the pair's first member, zp + 1 for its second (k groups 2, 3)
#pragma unroll
      for (int zp = 0; zp < LZ + 4; ++zp) pbase[zp] = halo0 + (unsigned)(((4 * step + zp) % NSLOT) * PLANEB);
#pragma unroll
      for (int zp = 0; zp < LZ + 2; ++zp)           // (arithmetic select: `cond ? pbase[zp + 1] : pbase[zp]` became an indexed scratch array)
        zbase[zp] = pbase[zp] + (unsigned)(kg >> 1) * (pbase[zp + 1] - pbase[zp]);

      f32x4 acc[LZ][LY];
#pragma unroll
      for (int z = 0; z < LZ; ++z)
#pragma unroll
        for (int y = 0; y < LY; ++y) {
          acc[z][y] = f32x4{0.f, 0.f, 0.f, 0.f};
          asm volatile("" : "+v"(acc[z][y]));       // (materialised HERE: the hazard recognizer does not see the inline-asm MFMAs, and a
        }                                          // v_mov sunk in front of the first MFMA that reads it as SrcC raced with it)
      asm volatile("s_nop 7" ::: "memory");

      // ---- all taps out of registers: fragment f is read NRING - 1 fragments ahead of its MFMAs ------------------------------
      bf16x8 ring[NRING];
      auto issue = [&](auto f_tag) {
        constexpr int f = decltype(f_tag)::value;
        if constexpr (f < NFRAG) {
          constexpr FragId id = frag_of(f);
          // row yp + 2 of the wave's quarter (halo coordinates), x window dxi
          constexpr int off = (id.yp + 2) * ROWB + id.dxi * VP;
          if constexpr (id.kind == 0) lds_read128<off>(ring[f % NRING], pbase[id.zp] + lane_pair);
          else if constexpr (id.kind == 1) lds_read128<off>(ring[f % NRING], zbase[id.zp] + lane_single);
          else lds_read128<off>(ring[f % NRING], pbase[id.zp] + lane_single);
        }
      };
      unroll_seq(std::make_integer_sequence<int, NRING - 1>{}, issue);
      unroll_seq(std::make_integer_sequence<int, NFRAG>{}, [&](auto f_tag) {
        constexpr int f = decltype(f_tag)::value;
        constexpr FragId id = frag_of(f);
        issue(std::integral_constant<int, f + NRING - 1>{});
        if constexpr (f % 10 == 5 && f / 10 < 4 * (HY / 4)) {      // staging row f / 10 of the next step: plane r / 5, row r % 5
          constexpr int r = f / 10;
          if (stage_next) stage_row(npoff[r / (HY / 4)], nslot[r / (HY / 4)], std::integral_constant<int, r % (HY / 4)>{});
        }
        // fragments are consumed in issue order: at most NRING - 1 younger reads may still be in flight
        constexpr int younger = (NFRAG - 1 - f) < (NRING - 1) ? (NFRAG - 1 - f) : (NRING - 1);
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[f % NRING]) : "i"(younger) : "memory");
        if constexpr (id.kind == 0) {
          constexpr int zi = id.zp - 2;              // input plane relative to the step's first output plane
          unroll_seq(std::make_integer_sequence<int, 5>{}, [&](auto a_tag) {      // dz = a - 2: output plane zi - dz
            constexpr int a = decltype(a_tag)::value;
            constexpr int zr = zi - (a - 2);
            if constexpr (zr >= 0 && zr < LZ) {
              if constexpr (id.yp + 2 >= 0 && id.yp + 2 < LY)    // dy pair (-2, -1): output row yp + 2
                mfma_u<(a * 2 + 0) * 5 + id.dxi>(acc[zr][id.yp + 2], A[a * 2 + 0][id.dxi], ring[f % NRING]);
              if constexpr (id.yp >= 0 && id.yp < LY)            // dy pair (0, +1): output row yp
                mfma_u<(a * 2 + 1) * 5 + id.dxi>(acc[zr][id.yp], A[a * 2 + 1][id.dxi], ring[f % NRING]);
            }
          });
        } else if constexpr (id.kind == 1) {         // dy = +2, planes (zp, zp + 1): output row yp - 2 of plane zp (dz pair -2, -1)
          if constexpr (id.zp < LZ)                  // ... and of plane zp - 2 (dz pair 0, +1)
            mfma_u<10 * 5 + id.dxi>(acc[id.zp][id.yp - 2], A[10][id.dxi], ring[f % NRING]);
          if constexpr (id.zp >= 2)
            mfma_u<11 * 5 + id.dxi>(acc[id.zp - 2][id.yp - 2], A[11][id.dxi], ring[f % NRING]);
        } else {                                     // (dz, dy) = (+2, +2): output plane zp - 4, row yp - 2
          mfma_u<12 * 5 + id.dxi>(acc[id.zp - 4][id.yp - 2], A[12][id.dxi], ring[f % NRING]);
        }
      });
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMA's result is architecturally visible to VALU

      // ---- epilogue: bias, partial statistics (one slot per step of a column), activation, [accumulate], 8-B NDHWC stores ---
      // A wave alone on its SIMD hides nothing: the epilogue is kept to ~15 instructions per row — the activation mode, the
      // accumulate form and the volume's end are wave-uniform and decided ONCE (they were three scalar branches per element),
      // statistics and bias as packed fp32, store addresses = a scalar row base + one lane offset.
      const bool want_stats = d.stats_slots > 0;
      const int co = kg * 4;
      f32x2 b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
      if (p.bias) {
        b01 = f32x2{co + 0 < d.Co ? p.bias[co + 0] : 0.f, co + 1 < d.Co ? p.bias[co + 1] : 0.f};
        b23 = f32x2{co + 2 < d.Co ? p.bias[co + 2] : 0.f, co + 3 < d.Co ? p.bias[co + 3] : 0.f};
      }
      f32x2 s1a = {0.f, 0.f}, s1b = {0.f, 0.f}, s2a = {0.f, 0.f}, s2b = {0.f, 0.f};
      const int oz0 = zs0 + step * SZ;
      char* out_c = p.out + ((size_t)n * d.Do * d.Ho * d.Wo + (size_t)(oy0 + wave * LY) * d.Wo) * d.out_cs * 2;
      const unsigned lane_o = (unsigned)((ox0 + col) * d.out_cs + d.out_co + co) * 2u;
      const bool st_lane = co < d.Co;
      const int nz = min(LZ, d.Do - oz0);          // (a segment's last step may hang over the volume)
      auto rows = [&](auto act_tag, auto acc_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr bool ACCUM = decltype(acc_tag)::value;
#pragma unroll
        for (int z = 0; z < LZ; ++z) {
          if (z >= nz) break;
#pragma unroll
          for (int y = 0; y < LY; ++y) {
            f32x2 v01 = f32x2{acc[z][y][0], acc[z][y][1]} + b01, v23 = f32x2{acc[z][y][2], acc[z][y][3]} + b23;
            s1a += v01; s1b += v23;
            s2a += v01 * v01; s2b += v23 * v23;
            if constexpr (ACT == GS_ACT_RELU) {
              v01 = f32x2{fmaxf(v01.x, 0.f), fmaxf(v01.y, 0.f)}; v23 = f32x2{fmaxf(v23.x, 0.f), fmaxf(v23.y, 0.f)};
            } else if constexpr (ACT == GS_ACT_LRELU) {
              v01 = f32x2{v01.x > 0.f ? v01.x : v01.x * d.slope, v01.y > 0.f ? v01.y : v01.y * d.slope};
              v23 = f32x2{v23.x > 0.f ? v23.x : v23.x * d.slope, v23.y > 0.f ? v23.y : v23.y * d.slope};
            } else if constexpr (ACT == GS_ACT_TANH) {
              v01 = f32x2{gs_tanh_call(v01.x), gs_tanh_call(v01.y)}; v23 = f32x2{gs_tanh_call(v23.x), gs_tanh_call(v23.y)};
            }
            uint2 o;
            o.x = pack_bf2(v01.x, v01.y);
            o.y = pack_bf2(v23.x, v23.y);
            char* rowp = out_c + ((size_t)(oz0 + z) * d.Ho + y) * d.Wo * d.out_cs * 2;      // wave-uniform
            uint2* dst = reinterpret_cast<uint2*>(rowp + lane_o);
            if (st_lane) {
              if constexpr (ACCUM) {     // bf16 read-modify-write, same rounding points as gconv_kernel / hconv_kernel
                const uint2 old = *dst;
                o.x = pack_bf2(bf_lo(o.x) + bf_lo(old.x), bf_hi(o.x) + bf_hi(old.x));
                o.y = pack_bf2(bf_lo(o.y) + bf_lo(old.y), bf_hi(o.y) + bf_hi(old.y));
              }
              *dst = o;
            }
          }
        }
      };
      using T_ = std::true_type; using F_ = std::false_type;
      if (d.accumulate) rows(std::integral_constant<int, GS_ACT_NONE>{}, T_{});        // (no bias / activation / statistics then)
      else if (d.act == GS_ACT_NONE) rows(std::integral_constant<int, GS_ACT_NONE>{}, F_{});
      else if (d.act == GS_ACT_RELU) rows(std::integral_constant<int, GS_ACT_RELU>{}, F_{});
      else if (d.act == GS_ACT_LRELU) rows(std::integral_constant<int, GS_ACT_LRELU>{}, F_{});
      else rows(std::integral_constant<int, GS_ACT_TANH>{}, F_{});
      const float s1[4] = {s1a.x, s1a.y, s1b.x, s1b.y}, s2[4] = {s2a.x, s2a.y, s2b.x, s2b.y};
      if (want_stats) {
        // (the LDS-DMA of the next step's planes is in flight: LDS accesses the compiler sees would wait for it — raw stores /
        // loads through inline asm, ordered by lgkmcnt and a bare barrier)
        const unsigned red0 = lds_addr(red);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row16_sum(s1[r]), q = row16_sum(s2[r]);
          if (col == 0) {
            const unsigned ad = red0 + (unsigned)((wave * 16 + co + r) * 8);
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4" ::"v"(ad), "v"(a), "v"(q) : "memory");
          }
        }
        lds_barrier();
        if (tid < 16 && tid < d.Co) {
          float a = 0.f, q = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            float ra, rq;
            const unsigned ad = red0 + (unsigned)((w * 16 + tid) * 8);
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(ra), "=v"(rq) : "v"(ad) : "memory");
            a += ra; q += rq;
          }
          const int slot = ((oz0 / SZ) * p.nbh + by) * p.nbw + bx;
          float* sp = p.stats + (((size_t)n * d.stats_slots + d.stats_slot0 + slot) * 2) * d.Co;
          sp[tid] = a;
          sp[d.Co + tid] = q;
        }
        lds_barrier();                               // red may be rewritten by the next step
      }
      // (counted wait of the next step: valid when every row's store was issued and nothing else — statistics stores of wave 0 only
      // make the count stricter)
      counted = oz0 + SZ <= d.Do;
    }
  }
}

bool hconv5_eligible(const gs_gconv_desc* d) {
  if (!gs_opt(GS_OPT_HCONV5)) return false;
  if (d->T != 125 || d->Ci != 16 || d->Co > 16 || d->Co < 8 || d->si != 1 || d->so != 1) return false;
  if (d->Dc != d->Do || d->Hc != d->Ho || d->Wc != d->Wo || d->pz || d->py || d->px) return false;
  if (d->Do % SZ || d->Ho % BY || d->Wo % BX || d->Di != d->Do || d->Hi != d->Ho || d->Wi != d->Wo) return false;
  for (int t = 0; t < 125; ++t)
    if (d->dd[t] < -2 || d->dd[t] > 2 || d->dh[t] < -2 || d->dh[t] > 2 || d->dw[t] < -2 || d->dw[t] > 2) return false;
  // voxels in units of 2048 (= the work of a CU-filling step pair): small volumes stay on hconv_kernel's 512-voxel boxes
  const long long units = (long long)d->N * d->Do * d->Ho * d->Wo / 2048;
  if (units < gs_opt(GS_OPT_HCONV5) || units >= (1LL << 31)) return false;
  if ((long long)d->Di * d->Hi * d->Wi * d->in_cs * 2 >= (1LL << 32)) return false;
  if ((long long)d->Do * d->Ho * d->Wo >= (1LL << 31)) return false;
  return true;
}
}  // namespace

// partial-statistics slots per image when the layer runs here (one per step of a column: 4 x 16 x 16 voxels), 0 when it does not
int gs_hconv5_slots(const gs_gconv_desc* d) {
  return hconv5_eligible(d) ? (d->Do / SZ) * (d->Ho / BY) * (d->Wo / BX) : 0;
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
  k.nbh = d->Ho / BY; k.nbw = d->Wo / BX;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  // columns are cut into z segments until every CU has a workgroup (a segment re-reads 4 planes of z halo: keep them long)
  const int columns = d->N * k.nbh * k.nbw, steps = d->Do / SZ;
  int nseg = gs_opt(GS_OPT_HCONV5_SEG) > 0 ? gs_opt(GS_OPT_HCONV5_SEG) : (cus + columns - 1) / columns;
  if (nseg > steps) nseg = steps;
  if (nseg < 1) nseg = 1;
  k.seg_steps = (steps + nseg - 1) / nseg;
  k.nseg = (steps + k.seg_steps - 1) / k.seg_steps;
  k.nwork = columns * k.nseg;
  k.d = *d;
  const int per = (k.nwork + cus - 1) / cus;               // persistent: equal shares of whole segments
  const int grid = (k.nwork + per - 1) / per;
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
