// Self-pipelined halo-resident kernel for the WIDE 3x3 stride-1 layers (the residual-block convs: ganslate/nn/generators/
// resnet/resnet2d.py:80-87) — an alternative to hconvw.hip's phase-locked loop, selected by gs_set_option("hconvx", 1 | 100),
// OFF by default: it needs 4-5 % fewer cycles per launch than hconvw_kernel (74.7k against 78.1k busy cycles per CU) and runs
// no faster, because the chip sustains a lower clock under the denser loop (1.97 GHz, 1.89 with loader waves, against 2.15;
// the energy per launch is the same 46-47 mJ; profiles/r03_power_probe.txt) — the forward form only; kept as the measured
// evidence for that limit and for A/B runs.
//
// What the s_memtime timeline of hconvw_kernel showed (tools/probe/timeline.py, profiles/r03_hconvw_timeline.txt): a K-step
// took 1724 cycles against 1024 of MFMA work because the two wave groups alternate "read fragments" and "run MFMAs"
// phases behind two 16-wave barriers per K-step — each phase is 570-630 cycles of LDS-DMA issue + fragment reads on one
// side, 430-510 of MFMA issue on the other, plus ~250 of barrier skew — and four chunk boundaries add ~1000 cycles each
// for the burst of halo DMA instructions. The same tile here (one 16 x 16 box x 128 output channels, 18 x 18 halo box of a
// 64-channel chunk staged once per 9 taps, weights through a 3-slot ring) runs as 8 waves of 64 pixels x 64 channels on
// v_mfma_f32_32x32x16_bf16, and every wave pipelines ITSELF: a unit is one 16-deep k-slice of a K-step (4 fragment reads,
// 4 MFMAs = 128 pipe cycles), the reads of unit u+1 are issued before the MFMAs of unit u (two fragment sets, counted
// lgkmcnt), the two waves of a SIMD fill each other's gaps, and there is ONE barrier per K-step (it publishes the next
// weight slot). LDS-DMA instructions are spread one per unit behind that unit's MFMAs: the halo box of chunk c+1 as one
// instruction per wave in each of the first 7 K-steps of chunk c, the two weight instructions of K-step ks+2 in units 1
// and 2 of K-step ks. Fragment reads per MFMA cycle drop by a third against the 64 x 32 wave tiles (128 KiB per K-step).
//
// LDS images (both filled lane-linearly by LDS-DMA, the layout lives in the per-lane SOURCE address):
//   halo   [18 rows][20 voxels (18 + 2 pad)][9 pieces of 16 B (8 channel octets + 1 pad)]: voxel pitch 144 B, row pitch 20
//          voxels. A B fragment (32 pixels x 16 k) is an 8-row x 4-column pixel block: lane l reads octet 2s + (l >> 5) of
//          pixel (row l/4 mod 8, column l mod 4) — every 16-lane group of ds_read_b128 lands on 16 distinct 4-bank slots
//          (brute-forced over pitches and block shapes: 144 / 20 / 8x4 is the smallest conflict-free image).
//   weights[128 rows][8 octets] per stage, octet o of row r at position o ^ ((r >> 1) & 7): an A fragment (32 rows x 16 k,
//          lane l: row l & 31, octet 2s + (l >> 5)) is conflict-free in every lane group.
// Accumulators: acc[ci][pj] (2 x 2 blocks of 32 channels x 32 pixels, 64 registers); D layout of the 32x32 MFMA: column
// (pixel) = lane & 31, row (channel) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
#include "hconvw.hpp"
#include <cstdlib>
#include <type_traits>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

namespace {
constexpr int XT = 9, XNW = 8, XBN = 128;
constexpr int XWT = XBN * 128;                  // weight stage: 128 rows x 64 k = 16 KiB
constexpr int XVP = 144, XRP = 20;              // halo voxel pitch (bytes), row pitch (voxels)
constexpr int XHPIECES = 18 * XRP * 9;          // 3240 pieces of 16 B
constexpr int XHINSTR = (XHPIECES + 63) / 64;   // 51 wave-instructions
constexpr int XHBUF = XHINSTR * 1024;           // 52224 B per buffer
constexpr int XHPW = (XHINSTR + XNW - 1) / XNW; // 7 halo DMA instructions per wave per chunk
constexpr int XLDS = 3 * XWT + 2 * XHBUF + 1024;
constexpr int XPJ = 8 * XRP * XVP;              // byte offset of the second pixel block (box rows 8..15)

template <int N>
__device__ __forceinline__ void lgkm_wait4(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

struct Frag { bf16x8 a0, a1, b0, b1; };

#ifdef GS_TIMELINE
// debug build only (tools/probe/timeline.py): see hconvw.hip
__device__ unsigned* g_tl_buf = nullptr;
constexpr int TLW = 8, TLS = 160;
#define TL_STAMP(slot)                                                                            \
  do {                                                                                            \
    if (tl_on) { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();                      \
      if (lane == 0) tl[wave * TLS + (slot)] = t_; }                                              \
  } while (0)
#else
#define TL_STAMP(slot) do {} while (0)
#endif
template <int I> using IC = std::integral_constant<int, I>;
}  // namespace

// ABL (debug builds only, results are garbage for ABL != 0), bit mask: 1 = no LDS-DMA in the loop, 2 = no fragment reads in
// the loop, 4 = no barrier in the loop — what each costs the K loop (tools/probe/timeline.py --abl)
template <bool RING, int ABL = 0, int LW = 0>
__global__ __launch_bounds__((XNW + LW) * 64) void hconvx_kernel(const HConvWK p) {
  static_assert(!RING, "hconvx: the fused data-gradient (RING) form is not built yet");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wring = smem;                            // 3 x 16 KiB
  char* hbuf = smem + 3 * XWT;                   // 2 x 51 KiB
  char* sink = hbuf + 2 * XHBUF;                 // 1 KiB
  const gs_gconv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

#ifdef GS_TIMELINE
  unsigned* tl = reinterpret_cast<unsigned*>(smem + XLDS);
  const bool tl_on = g_tl_buf != nullptr;
  TL_STAMP(0);
#endif
  int b;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int nt = b % p.tiles_n;
  b /= p.tiles_n;
  const int mt = b % p.tiles_m;
  const int n = b / p.tiles_m;
  const int oy0 = (mt / p.nbw) * 16, ox0 = (mt % p.nbw) * 16;

  // ---- LW > 0: waves XNW .. XNW+LW-1 are LOADERS -----------------------------------------------------------------------------
  // An LDS-DMA instruction holds its wave for 60-185 cycles at issue, and a compute wave issues in order: the three DMA
  // instructions per K-step of the LW == 0 form cost 4-7 us of a 45 us launch (ablation), the matrix pipe waiting behind
  // them. With loader waves (one per SIMD) the compute waves' stream is MFMAs, fragment reads, counted waits and one barrier
  // per K-step. A loader runs the same barrier sequence: [its share of K-step ks+1's weights has landed] barrier(ks)
  // [weights of K-step ks+2 into the slot K-step ks-1 used; in the first 7 K-steps of a chunk, pieces of the halo box of
  // chunk c+1 into the buffer chunk c-1 used]. Past the end the same instructions overwrite dead slots / the dead halo
  // buffer with clamped sources, so the vmcnt counts are compile-time constants. A loader ends after barrier(nk-1); the
  // epilogue's barriers then wait on the surviving (compute) waves only.
  if constexpr (LW > 0) {
    if (wave >= XNW) {
      const int lw = wave - XNW;
      constexpr int HPL = (XHINSTR + LW - 1) / LW;           // 13 halo instructions per loader per chunk
      constexpr int WPL = 16 / LW;                           // 4 weight instructions per loader per K-step
      constexpr int HPK = (HPL + 6) / 7;                     // 2 halo instructions per K-step (first 7 K-steps of a chunk)
      const char* in_n = p.in + ((size_t)n * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
      int hsrc[HPL];
#pragma unroll
      for (int i = 0; i < HPL; ++i) {
        const int inst = i * LW + lw;
        const int q = inst * 64 + lane;
        const int v = q / 9, part = q - v * 9;
        const int hy = v / XRP, hx = v - hy * XRP;
        bool ok = inst < XHINSTR && hy < 18 && hx < 18 && part < 8;
        int iy = border_index(oy0 + hy + p.hmin, d.Hi, d.border, ok);
        int ix = border_index(ox0 + hx + p.wmin, d.Wi, d.border, ok);
        iy = min(max(iy, 0), d.Hi - 1);
        ix = min(max(ix, 0), d.Wi - 1);
        hsrc[i] = ok ? ((iy * d.Wi + ix) * d.in_cs + part * 8) * 2 : -1;
      }
      auto halo_piece = [&](auto i_tag, int chunk, int buf) {
        constexpr int i = decltype(i_tag)::value;
        unsigned off = (unsigned)hsrc[i] + (unsigned)chunk * 128u;
        asm volatile("" : "+v"(off));
        const char* src = hsrc[i] >= 0 ? in_n + off : p.zero;
        const int inst = i * LW + lw;
        glds16(src, inst < XHINSTR ? hbuf + buf * XHBUF + inst * 1024 : sink);
      };
      const char* wsrc[WPL];
      int winc[WPL];
#pragma unroll
      for (int i = 0; i < WPL; ++i) {
        const int row = (lw * WPL + i) * 8 + (lane >> 3);
        const int oct = (lane & 7) ^ ((row >> 1) & 7);
        const int wco = nt * XBN + row;
        const bool wv = wco < d.w_rows;
        wsrc[i] = wv ? p.w + ((size_t)wco * d.Kp + oct * 8) * 2 : p.zero;
        winc[i] = wv ? 16 : 0;
      }
      auto weights = [&](int c, int t, int slot) {
        const int q0 = t * (d.Ci >> 3) + c * 8;
#pragma unroll
        for (int i = 0; i < WPL; ++i) glds16(wsrc[i] + (size_t)q0 * winc[i], wring + slot * XWT + (lw * WPL + i) * 1024);
      };
      static_for<0, HPL>([&](auto i_tag) { halo_piece(i_tag, 0, 0); });
      weights(0, 0, 0);
      weights(0, 1, 1);
      vm_wait<WPL>();                              // the halo box and K-step 0 landed
      __builtin_amdgcn_s_barrier();
      for (int c = 0; c < p.chunks; ++c) {
        const int cn = min(c + 1, p.chunks - 1);
        static_for<0, XT>([&](auto t_tag) {
          constexpr int t = decltype(t_tag)::value;
          constexpr int tp = (t + XT - 1) % XT;                                   // halo instructions of the previous K-step
          constexpr int hprev = tp < 7 ? ((tp + 1) * HPK <= HPL ? HPK : (HPL > tp * HPK ? HPL - tp * HPK : 0)) : 0;
          vm_wait<hprev>();
          __builtin_amdgcn_s_barrier();
          constexpr int t2 = (t + 2) % XT, dc2 = (t + 2) / XT;
          weights(min(c + dc2, p.chunks - 1), t2, (t + 2) % 3);
          if constexpr (t < 7) {
            static_for<0, HPK>([&](auto k_tag) {
              constexpr int i = t * HPK + decltype(k_tag)::value;
              if constexpr (i < HPL) halo_piece(IC<(i < HPL ? i : 0)>{}, cn, (c + 1) & 1);
            });
          }
        });
      }
      return;
    }
  }

  // ---- halo pieces of this thread (chunk 0 source byte offsets, or -1 for pad pieces / zero-padded positions) -----------
  const char* in_n = p.in + ((size_t)n * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
  int hsrc[XHPW];
#pragma unroll
  for (int i = 0; i < XHPW; ++i) {
    const int inst = i * XNW + wave;
    const int q = inst * 64 + lane;
    const int v = q / 9, part = q - v * 9;
    const int hy = v / XRP, hx = v - hy * XRP;
    bool ok = inst < XHINSTR && hy < 18 && hx < 18 && part < 8;
    int iy = border_index(oy0 + hy + p.hmin, d.Hi, d.border, ok);
    int ix = border_index(ox0 + hx + p.wmin, d.Wi, d.border, ok);
    iy = min(max(iy, 0), d.Hi - 1);
    ix = min(max(ix, 0), d.Wi - 1);
    hsrc[i] = ok ? ((iy * d.Wi + ix) * d.in_cs + part * 8) * 2 : -1;
  }
  auto issue_halo_piece = [&](auto i_tag, int chunk, int buf) {
    constexpr int i = decltype(i_tag)::value;
    if constexpr (ABL & 1) { if (chunk > 0) return; }
    if constexpr (LW > 0) return;
    const bool real = chunk < p.chunks;                      // wave-uniform; past the last chunk the piece is a dummy
    unsigned off = (unsigned)hsrc[i] + (unsigned)chunk * 128u;
    asm volatile("" : "+v"(off));
    const char* src = (real && hsrc[i] >= 0) ? in_n + off : p.zero;
    const int inst = i * XNW + wave;
    glds16(src, (real && inst < XHINSTR) ? hbuf + buf * XHBUF + inst * 1024 : sink);
  };
  // ---- weight stage: two LDS-DMA instructions per wave (8 rows x 8 octets each) --------------------------------------
  const char* wsrc[2];
  int winc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + (lane >> 3);
    const int oct = (lane & 7) ^ ((row >> 1) & 7);
    const int wco = nt * XBN + row;
    const bool wv = wco < d.w_rows;
    wsrc[i] = wv ? p.w + ((size_t)wco * d.Kp + oct * 8) * 2 : p.zero;
    winc[i] = wv ? 16 : 0;
  }
  auto issue_w_part = [&](auto i_tag, int c, int t, int slot) {
    constexpr int i = decltype(i_tag)::value;
    if constexpr (ABL & 1) { if (c > 0 || t > 1) return; }
    if constexpr (LW > 0) return;
    const bool real = c < p.chunks;
    const int q0 = t * (d.Ci >> 3) + c * 8;      // first 8-k piece of this K-step inside a pack row (tap-major pack)
    glds16(real ? wsrc[i] + (size_t)q0 * winc[i] : p.zero, real ? wring + slot * XWT + (wave * 2 + i) * 1024 : sink);
  };

  // ---- fragment addresses --------------------------------------------------------------------------------------------
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;       // 4-column pixel strip, 64-channel half
  const unsigned smem0 = lds_addr(smem);
  unsigned woffs[4];
  {
    const int row = wn * 64 + l31, f = (row >> 1) & 7;       // rows +32 (second channel block) share f
#pragma unroll
    for (int s = 0; s < 4; ++s) woffs[s] = smem0 + (unsigned)(row * 128 + (((2 * s + hh) ^ f) << 4));
  }
  const unsigned xlane = smem0 + 3 * XWT + (unsigned)(((l31 >> 2) * XRP + 4 * wm + (l31 & 3)) * XVP + hh * 16);
  int tb[XT];
#pragma unroll
  for (int t = 0; t < XT; ++t) tb[t] = (((int)d.dh[t] - p.hmin) * XRP + ((int)d.dw[t] - p.wmin)) * XVP;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // one fragment read of unit (stage, s): which = 0, 1 the two 32-channel weight blocks, 2, 3 the two 32-pixel blocks
  auto read_one = [&](auto stage_tag, auto s_tag, auto which_tag, unsigned xaddr, Frag& F) {
    constexpr int st = decltype(stage_tag)::value, s = decltype(s_tag)::value, which = decltype(which_tag)::value;
    if constexpr (ABL & 2) { asm volatile("" : "+v"(F.a0), "+v"(F.a1), "+v"(F.b0), "+v"(F.b1)); return; }
    if constexpr (which == 0) lds_read128<st * XWT>(F.a0, woffs[s]);
    if constexpr (which == 1) lds_read128<st * XWT + 32 * 128>(F.a1, woffs[s]);
    if constexpr (which == 2) lds_read128<32 * s>(F.b0, xaddr);
    if constexpr (which == 3) lds_read128<XPJ + 32 * s>(F.b1, xaddr);
  };
  auto issue_reads = [&](auto stage_tag, auto s_tag, unsigned xaddr, Frag& F) {
    read_one(stage_tag, s_tag, IC<0>{}, xaddr, F);
    read_one(stage_tag, s_tag, IC<1>{}, xaddr, F);
    read_one(stage_tag, s_tag, IC<2>{}, xaddr, F);
    read_one(stage_tag, s_tag, IC<3>{}, xaddr, F);
  };
  auto mma_one = [&](auto which_tag, const Frag& F) {
    constexpr int which = decltype(which_tag)::value;
    const bf16x8_t a = __builtin_bit_cast(bf16x8_t, (which & 2) ? F.a1 : F.a0);
    const bf16x8_t bb = __builtin_bit_cast(bf16x8_t, (which & 1) ? F.b1 : F.b0);
    acc[which >> 1][which & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc[which >> 1][which & 1], 0, 0, 0);
  };

  // ---- prologue: halo of chunk 0, weights of K-steps 0 and 1 ----------------------------------------------------------------
  static_for<0, XHPW>([&](auto i_tag) { issue_halo_piece(i_tag, 0, 0); });
  issue_w_part(IC<0>{}, 0, 0, 0);
  issue_w_part(IC<1>{}, 0, 0, 0);
  issue_w_part(IC<0>{}, 0, 1, 1);
  issue_w_part(IC<1>{}, 0, 1, 1);
  if constexpr (LW == 0) vm_wait<2>();           // the halo box and K-step 0 landed; K-step 1 is waited for at barrier(0)
  TL_STAMP(1);
  __builtin_amdgcn_s_barrier();
  TL_STAMP(2);
  Frag F0, F1, F2;
  if constexpr (ABL & 2) { F0 = Frag{}; F1 = Frag{}; F2 = Frag{}; }
  issue_reads(IC<0>{}, IC<0>{}, xlane + (unsigned)tb[0], F0);
  issue_reads(IC<0>{}, IC<1>{}, xlane + (unsigned)tb[0], F1);

  // ---- main loop ---------------------------------------------------------------------------------------------------------
  // Unit u = 4 ks + s (K-step ks = 9 c + t, slice s) uses fragment set u % 3 (36 units per chunk: the sets line up again at
  // every chunk). A unit is: wait for its set (lgkmcnt(4): the four reads of unit u+1 may still fly), then its 4 MFMAs with
  // the 4 reads of unit u+2 between them — one LDS instruction per MFMA gap instead of a burst in front of a burst, which left
  // the matrix pipe waiting for the read issue and the reads waiting behind the MFMA issue (ablations: the 16 reads of a
  // K-step cost 7-10 us of a 45 us launch that way, the 3 LDS-DMA instructions 4-7 us).
  // Units 2, 3 of K-step ks read slices 0, 1 of K-step ks+1, so barrier(ks) sits at the head of unit 2, behind this wave's
  // vmcnt(0) (its share of the weights of K-step ks+1 — issued a K-step ago — and of every halo piece so far has landed).
  // Every wave has by then waited for all reads of K-step ks-1 and of slices 0, 1 of K-step ks, so behind barrier(ks) the
  // weight slot of K-step ks-1 takes the weights of K-step ks+2 (units 2 and 3) and, in the first 7 K-steps of a chunk, one
  // piece of the halo box of chunk c+1 goes into the buffer chunk c-1 used (unit 2).
  for (int c = 0; c < p.chunks; ++c) {
    const unsigned xbase = xlane + (unsigned)((c & 1) * XHBUF);
    const unsigned xbase_next = xlane + (unsigned)(((c + 1) & 1) * XHBUF);
    const bool more = c + 1 < p.chunks;
    static_for<0, XT>([&](auto t_tag) {
      constexpr int t = decltype(t_tag)::value;
      constexpr int st = t % 3, stn = (t + 1) % 3;
      constexpr int t2 = (t + 2) % XT, dc2 = (t + 2) / XT;        // K-step ks+2
      const unsigned xcur = xbase + (unsigned)tb[t];
      const unsigned xnext = (t + 1 < XT ? xbase : xbase_next) + (unsigned)tb[(t + 1) % XT];
      const bool have_next = t + 1 < XT || more;                  // wave-uniform: a K-step follows this one
      static_for<0, 4>([&](auto s_tag) {
        constexpr int s = decltype(s_tag)::value;
        constexpr int u = 4 * t + s;
        auto& Fc = *[&]() { if constexpr (u % 3 == 0) return &F0; else if constexpr (u % 3 == 1) return &F1; else return &F2; }();
        auto& Fn = *[&]() { if constexpr ((u + 2) % 3 == 0) return &F0; else if constexpr ((u + 2) % 3 == 1) return &F1; else return &F2; }();
        if constexpr (s == 2) {
          TL_STAMP(8 + (c * XT + t) * 4 + 0);
          if constexpr (!(ABL & 1) && LW == 0) vm_wait<0>();
          TL_STAMP(8 + (c * XT + t) * 4 + 1);
          if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
          TL_STAMP(8 + (c * XT + t) * 4 + 2);
        }
        // the last two units of the launch have no successors' reads in flight behind their own
        if (s < 2 || have_next) lgkm_wait4<4>(Fc.a0, Fc.a1, Fc.b0, Fc.b1);
        else if (s == 2) lgkm_wait4<4>(Fc.a0, Fc.a1, Fc.b0, Fc.b1);
        else lgkm_wait4<0>(Fc.a0, Fc.a1, Fc.b0, Fc.b1);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 4>([&](auto w_tag) {
          mma_one(w_tag, Fc);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (s < 2) read_one(IC<st>{}, IC<s + 2>{}, w_tag, xcur, Fn);
          else if (have_next) read_one(IC<stn>{}, IC<(s + 2) % 4>{}, w_tag, xnext, Fn);
          __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (s == 2) {
          if constexpr (t < XHPW) issue_halo_piece(IC<(t < XHPW ? t : 0)>{}, c + 1, (c + 1) & 1);
          issue_w_part(IC<0>{}, c + dc2, t2, (st + 2) % 3);
        }
        if constexpr (s == 3) issue_w_part(IC<1>{}, c + dc2, t2, (st + 2) % 3);
      });
    });
  }
  if constexpr (LW == 0) vm_wait<0>();           // the dummy pieces of the tail have landed in the sink
  __syncthreads();                               // every wave is out of the loop: the LDS is free for the epilogue
  TL_STAMP(3);

  // ---- epilogue: bias, partial statistics (slot = box), activation, LDS-staged coalesced NHWC stores ---------------------
  // slab row = wm * 64 + pj * 32 + (lane & 31)  <->  box pixel (pj * 8 + (l31 >> 2), 4 wm + (l31 & 3))
  constexpr int SROW = XBN * 2 + 16;             // 272 B
  constexpr int RED_BYTES = 8 * XBN * 2 * 4;     // [wm][lane half][channel][sum, sum of squares]
  char* slab = smem + RED_BYTES;
  float* red = reinterpret_cast<float*>(smem);
  const bool want_stats = d.stats_slots > 0;
  f32x4 biav[2][4];                              // all eight loads in flight together (one dependent wait, not eight)
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      biav[ci][g] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nt * XBN + wn * 64 + ci * 32 + 8 * g + 4 * hh)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
  // the activation switch is hoisted out of the 128 values of a lane (apply_act's branches per value made this pass 3x longer
  // than the stores behind it); InstanceNorm layers — the residual convs — take the plain form
  auto tile_to_slab = [&](auto plain_tag) {
    constexpr bool PLAIN = decltype(plain_tag)::value;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cl = wn * 64 + ci * 32 + 8 * g + 4 * hh;      // first of this lane's 4 consecutive channels
        const f32x4 bia = biav[ci][g];
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) {
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[k] = acc[ci][pj][4 * g + k] + bia[k];
            s1[k] += v[k];
            s2[k] += v[k] * v[k];
            if constexpr (!PLAIN) v[k] = apply_act(v[k], d.act, d.slope);
          }
          uint2 o;
          o.x = pack_bf2(v[0], v[1]);
          o.y = pack_bf2(v[2], v[3]);
          *reinterpret_cast<uint2*>(slab + (wm * 64 + pj * 32 + l31) * SROW + cl * 2) = o;
        }
        if (want_stats) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float a = row16_sum(s1[k]), q = row16_sum(s2[k]);
            if ((lane & 15) == 0) {
              const int part = wm * 2 + ((lane >> 4) & 1);
              red[(part * XBN + cl + k) * 2 + 0] = a;
              red[(part * XBN + cl + k) * 2 + 1] = q;
            }
          }
        }
      }
    }
  };
  if (d.act == GS_ACT_NONE) tile_to_slab(std::true_type{}); else tile_to_slab(std::false_type{});
  __syncthreads();
  {
    const int sub = lane & 15, prow = lane >> 4;   // 16 lanes x 16 B = the 128 channels of a pixel, 4 pixels per instruction
    const int co = nt * XBN + sub * 8;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int sr = wave * 32 + it * 4 + prow;
      const int q = sr & 31;
      const int ly = ((sr >> 5) & 1) * 8 + (q >> 2), lx = (sr >> 6) * 4 + (q & 3);
      const size_t opix = ((size_t)n * d.Ho + (oy0 + ly)) * d.Wo + (ox0 + lx);
      const uint4 val = *reinterpret_cast<const uint4*>(slab + sr * SROW + sub * 16);
      *reinterpret_cast<uint4*>(p.out + (opix * d.out_cs + d.out_co + co) * 2) = val;
    }
  }
  TL_STAMP(4);
  if (want_stats && tid < XBN) {
    const int co = nt * XBN + tid;
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { a += red[(w * XBN + tid) * 2]; q += red[(w * XBN + tid) * 2 + 1]; }
    float* sp = p.stats + (((size_t)n * d.stats_slots + d.stats_slot0 + mt) * 2) * d.Co;
    sp[co] = a;
    sp[d.Co + co] = q;
  }
#ifdef GS_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TL_STAMP(5);
  __syncthreads();
  if (g_tl_buf != nullptr && (blockIdx.x == 0 || blockIdx.x == 101))
    for (int i = tid; i < TLW * TLS; i += XNW * 64) g_tl_buf[(blockIdx.x == 0 ? 0 : 1) * TLW * TLS + i] = tl[i];
#endif
}

#ifdef GS_TIMELINE
extern "C" int gs_debug_timeline_x(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_tl_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
constexpr int XLDS_LAUNCH = XLDS + TLW * TLS * 4;
#else
constexpr int XLDS_LAUNCH = XLDS;
#endif
// returns 0 and sets *handled when the layer ran here (same eligibility as hconvw.hip's forward form, which the caller has
// already checked: this is its drop-in, selected by gs_set_option("hconvx", 1))
int gs_hconvx_launch(const HConvWK& k, long long blocks, void* stream) {
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvx_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, XLDS_LAUNCH));
    configured = true;
  }
#ifdef GS_TIMELINE
  const int abl = gs_opt(GS_OPT_HCONVX) - 1;
  const dim3 g_((unsigned)blocks), b_(XNW * 64);
  hipStream_t st_ = static_cast<hipStream_t>(stream);
#define ABL_CASE(A)                                                                                                     \
  if (abl == A) {                                                                                                       \
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvx_kernel<false, A>),                           \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, XLDS_LAUNCH));                         \
    hipLaunchKernelGGL((hconvx_kernel<false, A>), g_, b_, XLDS_LAUNCH, st_, k);                                         \
    return 0;                                                                                                           \
  }
  ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(5) ABL_CASE(6) ABL_CASE(7)
#endif
  if (gs_opt(GS_OPT_HCONVX) >= 100) {            // 100: four loader waves beside the eight compute waves
    static bool configured_lw = false;
    if (!configured_lw) {
      GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvx_kernel<false, 0, 4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, XLDS_LAUNCH));
      configured_lw = true;
    }
    hipLaunchKernelGGL((hconvx_kernel<false, 0, 4>), dim3((unsigned)blocks), dim3((XNW + 4) * 64), XLDS_LAUNCH,
                       static_cast<hipStream_t>(stream), k);
    GS_CHECK_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL((hconvx_kernel<false>), dim3((unsigned)blocks), dim3(XNW * 64), XLDS_LAUNCH, static_cast<hipStream_t>(stream), k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
