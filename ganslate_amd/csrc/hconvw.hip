// Halo-resident forward kernel for the WIDE 3x3 stride-1 layers (the residual-block convs, Cin a multiple of 64,
// Cout a multiple of 128, image sides multiples of 16): ganslate/nn/generators/resnet/resnet2d.py:80-87.
//
// gconv_kernel's loop is bound by the LDS: per K-step it takes 48 KiB of LDS-DMA writes (16 KiB weights + 32 KiB of
// gathered pixels) and the fragment reads. The pixels of the 9 taps of a 64-channel chunk are the same
// 18x18 halo box read at 9 offsets, so this kernel stages that box ONCE per chunk (47 KiB, double-buffered, border
// handling resolved once per workgroup in the per-lane source offsets) and streams only the weights per K-step:
// 16 + 47/9 = 21 KiB of DMA per K-step instead of 48. The MFMA B operand of tap t is a ds_read_b128 at
// rowbase(j) + tapoff(t); a 144-byte voxel pitch makes 16 consecutive voxels cover all 64 banks.
// Tile 256 pixels (one 16x16 box) x 128 output channels, 16 waves as 4 x 4, 3-stage weight ring, same epilogue
// contract as gconv_kernel (bias, one statistics slot per box, activation, dense or sliced output).
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

#include "hconvw.hpp"

#ifdef GS_TIMELINE
// Debug build only (tools/probe/timeline.py, never the product library): s_memtime stamps of one workgroup's phases, kept in
// the 9 KiB of LDS the kernel leaves free and dumped to a global buffer at the end. TLW waves x TLS slots of 32 bits.
__device__ unsigned* g_tl_buf = nullptr;
extern "C" int gs_debug_timeline(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_tl_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
constexpr int TLW = 8, TLS = 160;
#define TL_STAMP(slot)                                                                            \
  do {                                                                                            \
    if (tl_on) { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();                      \
      if (lane == 0) tl[tl_w * TLS + (slot)] = t_; }                                              \
  } while (0)
#else
#define TL_STAMP(slot) do {} while (0)
#endif

// NW = 16: waves as 4 x 4, 64 pixels x 32 channels each; NW = 8: 4 x 2, 64 x 64 each (a third fewer fragment reads per
// MFMA — the loop is bound by LDS reads — for half the latency hiding)
//
// RING: data gradient of a reflect-padded (pad 1) 3x3 conv on the UNPADDED domain (resnet2d.py:80-87 backward). With
// C(u) = sum_t dY[u + off_t] W_t the zero-border conv on the extended domain u in [-1, H] x [-1, W], the gradient is
// dX[v] = sum over {u : reflect(u) = v} C(u): the box itself plus, for boxes on the image border, the ring pixels one
// step outside it (row -1 folds onto row 1, row H onto row H-2, same for columns, the four corners onto (1,1) ...).
// A ring pixel sees the image through 3 of the 9 taps only (1 for a corner), and the destination pixel lives in the same
// box, so each pixel-row group of waves takes one ring fragment as a side job — wm 0: top row (or a bottom corner),
// wm 3: bottom row (or a top corner), wm 1: left column, wm 2: right column — two extra MFMAs per half K-step on three
// taps, accumulated apart and added to the destination pixels in fp32 before rounding (through LDS, fixed order).
// gconv_kernel runs this layer on the 66 x 66 padded domain instead (+6 % pixels, 288-pixel im2col tiles: 59 us
// vs 41 here) and leaves the fold to the consumer. The epilogue is gs_gconv_forward_fused's: per-box sums of the
// consumer's InstanceNorm backward.
template <int T, int NW = 16, bool RING = false>
__global__ __launch_bounds__(NW * 64) void hconvw_kernel(const HConvWK p) {
  constexpr int BM = 256, BN = 128, WM = 4, WN = NW / 4;
  constexpr int WT = BN * 128;                   // weight stage: 128 rows x 64 k
  constexpr int HP = 160;                        // halo voxel pitch: 8 channel pieces of 16 B + 2 pad pieces. 40 banks: the
                                                 // 16 lanes of every ds_read_b128 lane group land on 16 distinct 4-bank
                                                 // slots (144 B left the k-chunk-1 lanes on the k-chunk-0 lanes' banks)
  constexpr int HPIECES = 18 * 18 * 10;          // 3240
  constexpr int HINSTR = (HPIECES + 63) / 64;    // 51 wave-instructions of 64 pieces
  constexpr int HBUF = HINSTR * 1024;            // 52224 B per buffer
  constexpr int HPW = (HINSTR + NW - 1) / NW;    // halo DMA instructions per wave per chunk (uniform: surplus ones
                                                 // copy the zero page into a 1-KiB sink so vmcnt counts stay equal)
  constexpr int WPI = 16 / NW;                   // weight DMA instructions per wave per K-step (8 rows each)
  constexpr int TI = BN / WN / 16, TJ = 4;
  constexpr int CWV = TI * 16;                   // output channels per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wring = smem;                            // 3 x 16 KiB
  char* hbuf = smem + 3 * WT;                    // 2 x 51 KiB
  char* sink = hbuf + 2 * HBUF;                  // 1 KiB
  const gs_gconv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

#ifdef GS_TIMELINE
  unsigned* tl = reinterpret_cast<unsigned*>(smem + 3 * WT + 2 * HBUF + 1024);
  const int tl_w = NW == 8 ? wave : ((wave & 3) | ((wave >> 3) << 2));          // NW 16: waves 0-3 and 8-11
  const bool tl_on = g_tl_buf != nullptr && (NW == 8 || (wave & 4) == 0);
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

  // ---- halo pieces of this thread (box-invariant, resolved once): source byte offset of channel chunk 0, or -1 ----
  const char* in_n = p.in + ((size_t)n * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
  int hsrc[HPW];
#pragma unroll
  for (int i = 0; i < HPW; ++i) {
    const int q = (i * NW + wave) * 64 + lane;   // wave-instruction i*16+wave covers pieces [inst*64, inst*64+64)
    const int v = q / 10, part = q - v * 10;
    const int hy = v / p.hw, hx = v - hy * p.hw;
    bool ok = q < HPIECES && part < 8;
    int iy = border_index(oy0 + hy + p.hmin, d.Hi, d.border, ok);
    int ix = border_index(ox0 + hx + p.wmin, d.Wi, d.border, ok);
    iy = min(max(iy, 0), d.Hi - 1);
    ix = min(max(ix, 0), d.Wi - 1);
    hsrc[i] = ok ? ((iy * d.Wi + ix) * d.in_cs + part * 8) * 2 : -1;
  }
  auto issue_halo = [&](int chunk, int buf) {
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
      unsigned off = (unsigned)hsrc[i] + (unsigned)chunk * 128u;
      asm volatile("" : "+v"(off));
      const char* src = hsrc[i] >= 0 ? in_n + off : p.zero;
      const int inst = i * NW + wave;
      glds16(src, inst < HINSTR ? hbuf + buf * HBUF + inst * 1024 : sink);
    }
  };
  // ---- weight stage: one LDS-DMA instruction per wave (128 rows x 8 pieces), rows swizzled like gconv_kernel ----
  const int lrow = lane >> 3;
  const int wchunk = (lane & 7) ^ lrow;
  const char* wsrc[WPI];
  int winc[WPI];                                 // bytes per 8-k piece step
#pragma unroll
  for (int i = 0; i < WPI; ++i) {
    const int wco = nt * BN + (wave * WPI + i) * 8 + lrow;
    const bool wv = wco < d.w_rows;
    wsrc[i] = wv ? p.w + ((size_t)wco * d.Kp + wchunk * 8) * 2 : p.zero;
    winc[i] = wv ? 16 : 0;
  }
  auto issue_w = [&](int c, int t, int buf) {
    const int q0 = (t * (d.Ci >> 3)) + c * 8;    // first 8-k piece of this K-step inside a pack row (tap-major pack)
#pragma unroll
    for (int i = 0; i < WPI; ++i) glds16(wsrc[i] + (size_t)q0 * winc[i], wring + buf * WT + (wave * WPI + i) * 1024);
  };

  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 15, fk = lane >> 4, swz = lane & 7;
  int rowb[TJ];                                  // halo byte offset of this lane's pixel in box row wm*4+j, k-chunk fk
#pragma unroll
  for (int j = 0; j < TJ; ++j) rowb[j] = ((wm * 4 + j) * p.hw + frow) * HP + fk * 16;
  int tb[T];                                     // tap byte offsets inside the halo
#pragma unroll
  for (int t = 0; t < T; ++t) tb[t] = (((int)d.dh[t] - p.hmin) * p.hw + ((int)d.dw[t] - p.wmin)) * HP;

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- RING: this wave's ring fragment (lane frow = one ring pixel), its taps and its slot in the ring buffer ----
  // taps come in the data-gradient order t = 3*ry + rx with (dh, dw) = (1 - ry, 1 - rx) (checked by the launcher)
  [[maybe_unused]] const bool e_top = oy0 == 0, e_bot = oy0 + 16 == d.Ho, e_lef = ox0 == 0, e_rig = ox0 + 16 == d.Wo;
  [[maybe_unused]] unsigned e_mask = 0;          // wave-uniform: taps that reach the image from this wave's ring pixels
  [[maybe_unused]] int e_rb = 0, e_slot = 0;
  [[maybe_unused]] bool e_lane = true;           // false: this lane carries no ring pixel (reads the zero sink)
  [[maybe_unused]] f32x4 accE[TI];
  if constexpr (RING) {
#pragma unroll
    for (int i = 0; i < TI; ++i) accE[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int py = 0, px = 0, sy = 0, sx = 0;          // ring pixel of lane frow: (py + sy*frow, px + sx*frow), box coordinates
    const bool side = e_lef || e_rig;
    if (wm == 0) {
      if (e_top) { e_mask = 0x007u; py = -1; sx = 1; e_slot = 0; }
      else if (e_bot && side) { e_mask = e_lef ? 0x040u : 0x100u; py = 16; px = e_lef ? -1 : 16; e_lane = frow == 0; e_slot = 4; }
    } else if (wm == 3) {
      if (e_bot) { e_mask = 0x1C0u; py = 16; sx = 1; e_slot = 1; }
      else if (e_top && side) { e_mask = e_lef ? 0x001u : 0x004u; py = -1; px = e_lef ? -1 : 16; e_lane = frow == 0; e_slot = 4; }
    } else if (wm == 1) {
      if (e_lef) { e_mask = 0x049u; px = -1; sy = 1; e_slot = 2; }
    } else {
      if (e_rig) { e_mask = 0x124u; px = 16; sy = 1; e_slot = 3; }
    }
    e_mask = __builtin_amdgcn_readfirstlane(e_mask);
    e_rb = ((py + sy * frow) * p.hw + (px + sx * frow)) * HP + fk * 16;
  }

  // ---- main loop: chunks x taps, software-pipelined over half K-steps --------------------------------------------
  // Weights run 3 K-steps ahead in a 3-slot ring, halo boxes 2 chunks ahead in 2 buffers. Inside a K-step the fragment
  // reads of the second half are issued before the MFMAs of the first, and the reads of the NEXT step's first half
  // (after the one barrier per K-step) before the MFMAs of the second: LDS latency hides under the matrix pipe.
  const int nk = p.chunks * T;
  // Fragment reads go through lds_read128 (common.hpp): the compiler would wait lgkmcnt(0) before every MFMA block
  // because of the LDS-DMA in the loop; here block A waits with lgkmcnt(TI + TJ), i.e. only for ITS fragments while the
  // TI + TJ reads of block B issued after them are still in flight.
  const unsigned smem0 = lds_addr(smem);
  const unsigned woff = (unsigned)((wn * CWV + frow) * 128);
  const unsigned c0 = (unsigned)(((0 * 4 + fk) ^ swz) << 4), c1 = (unsigned)(((1 * 4 + fk) ^ swz) << 4);
  auto load_frags = [&](unsigned wb, unsigned xb, auto kk_tag, bf16x8 (&wf)[TI], bf16x8 (&xf)[TJ]) {
    constexpr int kk = decltype(kk_tag)::value;
    const unsigned wa = wb + (kk ? c1 : c0);
    lds_read128<0>(wf[0], wa);
    lds_read128<2048>(wf[1], wa);
    if constexpr (TI == 4) { lds_read128<4096>(wf[2], wa); lds_read128<6144>(wf[3], wa); }
#pragma unroll
    for (int j = 0; j < TJ; ++j) lds_read128<kk * 64>(xf[j], xb + (unsigned)rowb[j]);
  };
  auto wait_frags = [&](auto n_tag, bf16x8 (&wf)[TI], bf16x8 (&xf)[TJ]) {
    constexpr int n = decltype(n_tag)::value;
    if constexpr (TI == 2) gs_lgkm_wait<n>(wf[0], wf[1], xf[0], xf[1], xf[2], xf[3]);
    else gs_lgkm_wait<n>(wf[0], wf[1], wf[2], wf[3], xf[0], xf[1], xf[2], xf[3]);
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using NF = std::integral_constant<int, TI + TJ>;
  auto mma = [&](const bf16x8 (&wf)[TI], const bf16x8 (&xf)[TJ]) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
  };
  auto ct_of = [&](int ks, int& c, int& t) { c = ks / T; t = ks - c * T; };
  // ---- main loop: two wave groups, one phase apart ----------------------------------------------------------------
  // Measured on the barrier-per-K-step loop this replaces (profiles/r01_hconvw_pmc.txt): matrix pipe busy 45 %, LDS
  // array busy 42 %, and a K-step took the SUM of its fragment-read time and its MFMA time — the barrier phase-locks all
  // 16 waves, so everybody queues on the LDS (SQ_WAIT_INST_LDS 17 % of the wave cycles) and then everybody queues on the
  // matrix pipe. Here a K-step is two phases separated by barriers, L = issue the 2*(TI+TJ) fragment reads of the step
  // and wait for them, M = its 2*TI*TJ MFMAs, and the upper half of the waves executes ONE extra barrier up front: from
  // then on one group is always in L while the other is in M (each SIMD hosts waves of both groups), the LDS and the
  // matrix pipe work at the same time, and priority is raised for the M phase (cdna_hip_programming.md T3-T5).
  // LDS-DMA: weights of step ks+2 go into the slot of step ks-1 at the start of L(ks) (that slot's last reader finished
  // a phase ago), the halo of chunk c+1 into the buffer of chunk c-1 at the first L of chunk c; a wave waits for its
  // share of step ks+1's weights at the end of the last phase before the first reader (group 0: end of M(ks), group 1:
  // end of L(ks)) and the barrier that follows publishes it.
  const bool grp = wave >= NW / 2;
  issue_halo(0, 0);
  if (p.chunks > 1) issue_halo(1, 1);
#pragma unroll
  for (int s0 = 0; s0 < 3; ++s0)
    if (s0 < nk) { int c0_, t0; ct_of(s0, c0_, t0); issue_w(c0_, t0, s0); }
  // halo 0 (and, in order, halo 1) and weights 0 landed; weights 1, 2 may still fly
  if (nk >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WPI) : "memory");
  else if (nk == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPI) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TL_STAMP(1);
  __builtin_amdgcn_s_barrier();
  if (grp) __builtin_amdgcn_s_barrier();
  TL_STAMP(2);
  const unsigned wring0 = smem0 + woff, hbuf0 = smem0 + 3 * WT;
  bf16x8 wA[TI], xA[TJ], wB[TI], xB[TJ];
  [[maybe_unused]] bf16x8 xE0, xE1;
  [[maybe_unused]] const unsigned sink0 = smem0 + 3 * WT + 2 * HBUF;
  int stage = 0;
  for (int c = 0; c < p.chunks; ++c) {
    const unsigned hb = hbuf0 + (unsigned)((c & 1) * HBUF);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int ks = c * T + t;
      // ---- L(ks) ----
      const bool halo_now = t == 0 && c >= 1 && c + 1 < p.chunks;
      if (halo_now) issue_halo(c + 1, (c + 1) & 1);
      if (ks >= 1 && ks + 2 < nk) { int c2, t2; ct_of(ks + 2, c2, t2); issue_w(c2, t2, stage == 0 ? 2 : stage - 1); }
      load_frags(wring0 + (unsigned)(stage * WT), hb + (unsigned)tb[t], K0{}, wA, xA);
      load_frags(wring0 + (unsigned)(stage * WT), hb + (unsigned)tb[t], K1{}, wB, xB);
      [[maybe_unused]] const bool e_now = RING && ((e_mask >> t) & 1u);
      if constexpr (RING) {
        if (e_now) {
          const unsigned ea = e_lane ? hb + (unsigned)tb[t] + (unsigned)e_rb : sink0;
          lds_read128<0>(xE0, ea);
          lds_read128<64>(xE1, ea);
        }
      }
      wait_frags(std::integral_constant<int, 0>{}, wA, xA);
      wait_frags(std::integral_constant<int, 0>{}, wB, xB);
      if constexpr (RING) {
        if (e_now) { reg_fence(xE0); reg_fence(xE1); }
      }
      TL_STAMP(8 + ks * 4 + 0);
      auto wait_next_weights = [&]() {       // this wave's share of step ks+1's weights (and anything older) has landed
        if (ks + 1 >= nk) return;
        if (ks + 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (ks == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPI) : "memory");     // w1 | w2 outstanding
        else if (halo_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HPW + WPI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPI) : "memory");
      };
      if (grp) wait_next_weights();
      __builtin_amdgcn_s_barrier();
      TL_STAMP(8 + ks * 4 + 1);
      // ---- M(ks) ----
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      mma(wA, xA);
      mma(wB, xB);
      if constexpr (RING) {
        if (e_now) {
#pragma unroll
          for (int i = 0; i < TI; ++i) accE[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[i], xE0, accE[i], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < TI; ++i) accE[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wB[i], xE1, accE[i], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      TL_STAMP(8 + ks * 4 + 2);
      if (!grp) wait_next_weights();
      __builtin_amdgcn_s_barrier();
      TL_STAMP(8 + ks * 4 + 3);
      stage = stage == 2 ? 0 : stage + 1;
    }
  }
  if (!grp) __builtin_amdgcn_s_barrier();       // group 1 ran one barrier ahead of the loop: every wave has passed the same count
  if constexpr (RING) {
    // ---- RING epilogue: ring sums -> LDS -> added (fp32) to the pixels they fold onto, bf16 tile through the per-wave
    // slabs, coalesced stores with the consumer's InstanceNorm-backward sums (same contract as gconv_kernel's fused
    // epilogue: sums over the box of ghat = (g + g2) * act'(yhat), ghat * yhat, yhat; one slot per box) -------------
    constexpr int CW = CWV, PW = 64, SROW = CW * 2 + 16;
    constexpr int RED_BYTES = WM * BN * 3 * 4;
    constexpr int SLAB0 = ((RED_BYTES + 255) / 256) * 256;
    constexpr int RING0 = SLAB0 + NW * PW * SROW;            // [5 slots: top, bottom, left, right, corner][16 pixels][BN] fp32
    float* ringbuf = reinterpret_cast<float*>(smem + RING0);
    __syncthreads();                                         // the last K-step's operands have been read by every wave
    if (e_mask) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
        *reinterpret_cast<f32x4*>(ringbuf + (e_slot * 16 + frow) * BN + wn * CWV + i * 16 + fk * 4) = accE[i];
    }
    __syncthreads();
    const int cy = e_top ? 1 : 14, cx = e_lef ? 1 : 14;      // where this box's image corner (if it has one) folds onto
    const bool corner = (e_top || e_bot) && (e_lef || e_rig);
    char* slab = smem + SLAB0 + wave * (PW * SROW);
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int y = wm * 4 + j;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int cl = wn * CWV + i * 16 + fk * 4;
        f32x4 v = acc[i][j];
        if (e_top && y == 1) v += *reinterpret_cast<const f32x4*>(ringbuf + (0 * 16 + frow) * BN + cl);
        if (e_bot && y == 14) v += *reinterpret_cast<const f32x4*>(ringbuf + (1 * 16 + frow) * BN + cl);
        if (e_lef && frow == 1) v += *reinterpret_cast<const f32x4*>(ringbuf + (2 * 16 + y) * BN + cl);
        if (e_rig && frow == 14) v += *reinterpret_cast<const f32x4*>(ringbuf + (3 * 16 + y) * BN + cl);
        if (corner && y == cy && frow == cx) v += *reinterpret_cast<const f32x4*>(ringbuf + (4 * 16 + 0) * BN + cl);
        uint2 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(slab + (j * 16 + frow) * SROW + (i * 16 + fk * 4) * 2) = o;
      }
    }
    __syncthreads();
    constexpr int LPR = CW / 8, PPI = 64 / LPR;            // 4 (16 waves: 32 channels per wave) or 8 (8 waves: 64) lanes per pixel
    static_assert(LPR == 4 || LPR == 8, "ring epilogue: 32 or 64 channels per wave");
    const int sub = lane % LPR, prow = lane / LPR;
    const int co = nt * BN + wn * CW + sub * 8;
    float fa1[8], fa2[8], fa3[8], fmu[8], frs[8];
    {
      const float* mr = p.f.mean_rstd + (size_t)n * 2 * d.Co;
#pragma unroll
      for (int k = 0; k < 8; ++k) { fa1[k] = fa2[k] = fa3[k] = 0.f; fmu[k] = mr[co + k]; frs[k] = mr[d.Co + co + k]; }
    }
#pragma unroll
    for (int it = 0; it < PW / PPI; ++it) {
      const int pl = it * PPI + prow;
      const int ly = wm * 4 + (pl >> 4), lx = pl & 15;
      const size_t opix = ((size_t)n * d.Ho + (oy0 + ly)) * d.Wo + (ox0 + lx);
      const uint4 val = *reinterpret_cast<const uint4*>(slab + pl * SROW + sub * 16);
      const uint4 yv = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.f.y) + (opix * d.Co + co) * 2);
      *reinterpret_cast<uint4*>(p.out + (opix * d.out_cs + d.out_co + co) * 2) = val;
      float g[8] = {bf_lo(val.x), bf_hi(val.x), bf_lo(val.y), bf_hi(val.y),
                    bf_lo(val.z), bf_hi(val.z), bf_lo(val.w), bf_hi(val.w)};
      const float yr[8] = {bf_lo(yv.x), bf_hi(yv.x), bf_lo(yv.y), bf_hi(yv.y),
                           bf_lo(yv.z), bf_hi(yv.z), bf_lo(yv.w), bf_hi(yv.w)};
      if (p.f.g2) {
        const uint4 gv = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.f.g2) + (opix * d.Co + co) * 2);
        g[0] += bf_lo(gv.x); g[1] += bf_hi(gv.x); g[2] += bf_lo(gv.y); g[3] += bf_hi(gv.y);
        g[4] += bf_lo(gv.z); g[5] += bf_hi(gv.z); g[6] += bf_lo(gv.w); g[7] += bf_hi(gv.w);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float yh = (yr[k] - fmu[k]) * frs[k];
        const float gh = g[k] * act_grad_from_out(yh, p.f.act, p.f.slope);
        fa1[k] += gh;
        fa2[k] += gh * yh;
        fa3[k] += yh;
      }
    }
    float* red3 = reinterpret_cast<float*>(smem);            // [WM][BN][3]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (LPR == 4) {
        fa1[k] = row_sum_stride4(fa1[k]); fa2[k] = row_sum_stride4(fa2[k]); fa3[k] = row_sum_stride4(fa3[k]);
      } else {
        fa1[k] = row_sum_stride8(fa1[k]); fa2[k] = row_sum_stride8(fa2[k]); fa3[k] = row_sum_stride8(fa3[k]);
      }
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
        red3[(wm * BN + cl) * 3 + 0] = fa1[k];
        red3[(wm * BN + cl) * 3 + 1] = fa2[k];
        red3[(wm * BN + cl) * 3 + 2] = fa3[k];
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int c = nt * BN + tid;
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        t0 += red3[(w * BN + tid) * 3]; t1 += red3[(w * BN + tid) * 3 + 1]; t2 += red3[(w * BN + tid) * 3 + 2];
      }
      float* sp = p.f.partial + ((size_t)n * p.tiles_m + mt) * 3 * d.Co;
      sp[c] = t0; sp[d.Co + c] = t1; sp[2 * d.Co + c] = t2;
    }
    return;
  }
  f32x4 bia[TI];                                // loaded after the loop: inside it they would spill (128-VGPR cap)
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int co = nt * BN + wn * CWV + i * 16 + fk * 4;
    bia[i] = (p.bias && co < d.Co) ? *reinterpret_cast<const f32x4*>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  TL_STAMP(3);

  // ---- epilogue: bias, partial statistics (slot = box), activation, LDS-staged coalesced NHWC stores ------------
  const bool want_stats = d.stats_slots > 0;
  float s1[TI][4], s2[TI][4];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
  constexpr int CW = CWV, PW = 64, SROW = CW * 2 + 16;
  constexpr int RED_BYTES = WM * BN * 2 * 4;
  char* slab = smem + ((RED_BYTES + 255) / 256) * 256 + wave * (PW * SROW);
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] + bia[i][r];
        s1[i][r] += v[r];
        s2[i][r] += v[r] * v[r];
        v[r] = apply_act(v[r], d.act, d.slope);
      }
      uint2 o;
      o.x = pack_bf2(v[0], v[1]);
      o.y = pack_bf2(v[2], v[3]);
      *reinterpret_cast<uint2*>(slab + (j * 16 + frow) * SROW + (i * 16 + fk * 4) * 2) = o;
    }
  }
  __syncthreads();
  {
    constexpr int LPR = CW / 8, PPI = 64 / LPR;   // 4 lanes per pixel, 16 pixels per store instruction
    const int sub = lane % LPR, prow = lane / LPR;
    const int co = nt * BN + wn * CW + sub * 8;
#pragma unroll
    for (int it = 0; it < PW / PPI; ++it) {
      const int pl = it * PPI + prow;             // pixel inside the wave's 4 box rows
      const int ly = wm * 4 + (pl >> 4), lx = pl & 15;
      if (co < d.Co) {
        const size_t opix = ((size_t)n * d.Ho + (oy0 + ly)) * d.Wo + (ox0 + lx);
        const uint4 val = *reinterpret_cast<const uint4*>(slab + pl * SROW + sub * 16);
        *reinterpret_cast<uint4*>(p.out + (opix * d.out_cs + d.out_co + co) * 2) = val;
      }
    }
  }
  TL_STAMP(4);
  if (want_stats) {
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[i][r], q = s2[i][r];
        a = row16_sum(a);
        q = row16_sum(q);
        if (frow == 0) {
          const int cl = wn * CW + i * 16 + fk * 4 + r;
          red[(wm * BN + cl) * 2 + 0] = a;
          red[(wm * BN + cl) * 2 + 1] = q;
        }
      }
    __syncthreads();
    if (tid < BN) {
      const int co = nt * BN + tid;
      if (co < d.Co) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { a += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
        float* sp = p.stats + (((size_t)n * d.stats_slots + d.stats_slot0 + mt) * 2) * d.Co;
        sp[co] = a;
        sp[d.Co + co] = q;
      }
    }
  }
#ifdef GS_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tile's stores have been accepted
  TL_STAMP(5);
  __syncthreads();
  if (g_tl_buf != nullptr && (blockIdx.x == 0 || blockIdx.x == 101))
    for (int i = tid; i < TLW * TLS; i += NW * 64) g_tl_buf[(blockIdx.x == 0 ? 0 : 1) * TLW * TLS + i] = tl[i];
#endif
}

int gs_hconvx_launch(const HConvWK& k, long long blocks, void* stream);   // hconvx.hip

static bool hconvw_eligible(const gs_gconv_desc* d, int* lo) {
  const bool enabled = gs_opt(GS_OPT_HCONV_WIDE) != 0;
  if (!enabled || d->si != 1 || d->so != 1 || d->T != 9 || d->Ci % 64 != 0 || d->Co % 128 != 0 || d->accumulate) return false;
  if (d->Di != 1 || d->Do != 1 || d->Dc != 1 || d->Hc != d->Ho || d->Wc != d->Wo || d->py || d->px || d->pz) return false;
  if (d->Ho % 16 != 0 || d->Wo % 16 != 0) return false;
  int hi[2] = {-128, -128};
  lo[0] = lo[1] = 127;
  for (int t = 0; t < d->T; ++t) {
    if (d->dd[t] != 0) return false;
    const int o[2] = {d->dh[t], d->dw[t]};
    for (int a = 0; a < 2; ++a) { if (o[a] < lo[a]) lo[a] = o[a]; if (o[a] > hi[a]) hi[a] = o[a]; }
  }
  if (hi[0] - lo[0] != 2 || hi[1] - lo[1] != 2) return false;    // 18 x 18 halo
  const long long blocks = (long long)d->N * (d->Ho / 16) * (d->Wo / 16) * (d->Co / 128);
  if (blocks < 192 || blocks >= (1LL << 31)) return false;        // small grids: the 128-pixel tiles fill the chip better
  if ((long long)d->Hi * d->Wi * d->in_cs * 2 >= (1LL << 31)) return false;
  return true;
}

// partial-statistics slots per image when the layer runs here (one per 16x16 box), 0 when it does not
int gs_hconvw_slots(const gs_gconv_desc* d) {
  int lo[2];
  return hconvw_eligible(d, lo) ? (d->Ho / 16) * (d->Wo / 16) : 0;
}

// returns 0 and sets *handled when the layer ran here
int gs_hconvw_try(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out,
                  float* stats, void* stream, int* handled) {
  *handled = 0;
  int lo[2];
  if (!hconvw_eligible(d, lo)) return 0;
  const long long blocks = (long long)d->N * (d->Ho / 16) * (d->Wo / 16) * (d->Co / 128);
  HConvWK k;
  k.in = static_cast<const char*>(in);
  k.w = static_cast<const char*>(w_pack);
  k.bias = bias;
  k.out = static_cast<char*>(out);
  k.stats = stats;
  k.zero = static_cast<const char*>(gs_zero_page());
  GS_REQUIRE(k.zero, "gs_gconv_forward: library not initialised (call gs_init)");
  k.tiles_m = (d->Ho / 16) * (d->Wo / 16);
  k.tiles_n = d->Co / 128;
  k.nbw = d->Wo / 16;
  k.hh = 18; k.hw = 18; k.hmin = lo[0]; k.wmin = lo[1];
  k.chunks = d->Ci / 64;
  k.d = *d;
  k.f = gs_gconv_fuse{};
#ifdef GS_TIMELINE
  const int lds = 3 * 128 * 128 + 2 * ((18 * 18 * 10 + 63) / 64) * 1024 + 1024 + TLW * TLS * 4;
#else
  const int lds = 3 * 128 * 128 + 2 * ((18 * 18 * 10 + 63) / 64) * 1024 + 1024;
#endif
  const int nw = gs_opt(GS_OPT_HCONVW_WAVES);
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvw_kernel<9, 16>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvw_kernel<9, 8>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  *handled = 1;
  if (gs_opt(GS_OPT_HCONVX)) return gs_hconvx_launch(k, blocks, stream);
  if (nw == 8)
    hipLaunchKernelGGL((hconvw_kernel<9, 8>), dim3((unsigned)blocks), dim3(512), lds, static_cast<hipStream_t>(stream), k);
  else
    hipLaunchKernelGGL((hconvw_kernel<9, 16>), dim3((unsigned)blocks), dim3(1024), lds, static_cast<hipStream_t>(stream), k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- RING form: fused data gradient of a reflect-padded 3x3 conv on the unpadded domain ---------------------------------
static bool hconvw_ring_eligible(const gs_gconv_desc* d) {
  int lo[2];
  if (gs_opt(GS_OPT_HCONVW_RING) == 0 || !hconvw_eligible(d, lo)) return false;
  if (d->border != GS_BORDER_ZERO || d->Hi != d->Ho || d->Wi != d->Wo || d->Ho < 32 || d->Wo < 32) return false;
  if (d->stats_slots != 0 || d->act != GS_ACT_NONE || d->out_cs != d->Co || d->out_co != 0) return false;
  for (int t = 0; t < 9; ++t)
    if (d->dh[t] != 1 - t / 3 || d->dw[t] != 1 - t % 3) return false;   // the data-gradient tap order the kernel assumes
  return true;
}

// slots of partial sums per image of the ring form (one per 16x16 box), 0 when the layer does not run in it: then the
// caller lowers the data gradient onto the padded domain (gs_gconv_forward_fused with the padded output extent)
extern "C" int gs_gconv_ring_slots(const gs_gconv_desc* d) {
  return (d && hconvw_ring_eligible(d)) ? (d->Ho / 16) * (d->Wo / 16) : 0;
}

int gs_hconvw_ring(const gs_gconv_desc* d, const void* in, const void* w_pack, void* out, const gs_gconv_fuse* fuse,
                   void* stream) {
  GS_REQUIRE(hconvw_ring_eligible(d), "gs_gconv_forward_fused: layer is not eligible for the unpadded (ring) form, see "
                                      "gs_gconv_ring_slots");
  GS_REQUIRE(fuse->fold == 1 && fuse->fold_mode == GS_BORDER_REFLECT && fuse->Dy == 1,
             "gs_gconv_forward_fused: the unpadded form folds a reflect padding of 1");
  HConvWK k;
  k.in = static_cast<const char*>(in);
  k.w = static_cast<const char*>(w_pack);
  k.bias = nullptr;
  k.out = static_cast<char*>(out);
  k.stats = nullptr;
  k.zero = static_cast<const char*>(gs_zero_page());
  GS_REQUIRE(k.zero, "gs_gconv_forward: library not initialised (call gs_init)");
  k.tiles_m = (d->Ho / 16) * (d->Wo / 16);
  k.tiles_n = d->Co / 128;
  k.nbw = d->Wo / 16;
  k.hh = 18; k.hw = 18; k.hmin = -1; k.wmin = -1;
  k.chunks = d->Ci / 64;
  k.d = *d;
  k.f = *fuse;
  const long long blocks = (long long)d->N * k.tiles_m * k.tiles_n;
  const int lds = 3 * 128 * 128 + 2 * ((18 * 18 * 10 + 63) / 64) * 1024 + 1024;
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvw_kernel<9, 16, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvw_kernel<9, 8, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  // 8 waves of 64 x 64 (209 registers x 2 waves per SIMD) against 16 waves of 64 x 32 (127 x 4, the whole register file)
  if (gs_opt(GS_OPT_HCONVW_RING_WAVES) == 8)
    hipLaunchKernelGGL((hconvw_kernel<9, 8, true>), dim3((unsigned)blocks), dim3(512), lds,
                       static_cast<hipStream_t>(stream), k);
  else
    hipLaunchKernelGGL((hconvw_kernel<9, 16, true>), dim3((unsigned)blocks), dim3(1024), lds,
                       static_cast<hipStream_t>(stream), k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
