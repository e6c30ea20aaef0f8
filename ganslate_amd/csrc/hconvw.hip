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
//
// PERSISTENT, SEVERAL TILES PER WORKGROUP (round 4). A launch over 2N images — the same layer of the two generators of a
// CycleGAN phase as one twin batch, p.nsplit / p.w_delta / p.bias_delta select the weight set per image — has 512 tiles
// for 256 CUs. Workgroup b walks tiles b, b + gridDim.x, ...: the K-step sequence simply continues — the weight ring and
// the two halo buffers are fed with the NEXT tile's first chunk / first K-steps while the current tile's last chunk
// computes — so only the first tile of a workgroup pays the 9k-cycle prologue, and a tile's stores are issued and never
// waited for: they drain under the next tile's K loop (profiles/r03_hconvw_timeline.txt: prologue 9.2k + epilogue 10.2k of
// 84k cycles per launch, plus the launch ramp between launches). At a tile boundary the two wave groups of the loop are
// brought back into step (one barrier), the epilogue runs out of the ONE halo buffer that is free at that point (two
// passes of 32 pixels through 40 KiB instead of one through 80), and the phase shift is set up again.
// vmcnt bookkeeping across the boundary (per wave, VMEM retires in issue order): the epilogue's 4 output stores sit between
// the next tile's K-step-2 weights (issued before them) and its chunk-1 halo (after); the first two waits of the new tile
// leave exactly those younger operations outstanding.
// APPLY (round 5, RING only): the apply pass of the consumer's InstanceNorm backward rides in the epilogue too. The sums it needs
// are over the whole image — the 16 boxes of an image sit on 16 workgroups that run the same tile index at the same time (one
// persistent workgroup per CU, all resident) — so a box writes its partial sums through to memory (agent-scope stores),
// arrives at a per-(image, channel tile) counter, waits for the others (bounded spin), adds the 16 slots up in slot order
// (the arithmetic of inorm_bwd_apply_cg_kernel: double accumulation, then dy = rstd * (ghat - mean ghat - yhat * mean ghat yhat)
// on the bf16-rounded gradient) and stores dy — and the total gradient gx + g2 for the skip path — instead of gx: the
// 23-us apply launch, its re-read of y / g2 and the write + read of gx are gone. Bit-identical to the two launches
// (tests/test_ops_gpu.py::test_ring_form_with_the_norm_backward_applied_in_the_launch).
template <int T, bool RING = false, bool APPLY = false>
__global__ __launch_bounds__(1024) void hconvw_kernel(const HConvWK p) {
  static_assert(RING || !APPLY, "the in-launch norm backward belongs to the fused data gradient");
  constexpr int NW = 16;
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
  constexpr int NST = 4;                         // output stores per wave per tile (a lower bound is all the waits need)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wring = smem;                            // 3 x 16 KiB
  char* hbuf = smem + 3 * WT;                    // 2 x 51 KiB
  char* sink = hbuf + 2 * HBUF;                  // 1 KiB
  char* spare = sink + 1024;                     // 9 KiB: the halo source table (8 KiB)
  const gs_gconv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // Workgroup b walks tiles b, b + gridDim.x, ...; the launcher makes gridDim.x a multiple of the tiles per image, so every
  // tile of a workgroup is the SAME box and channel tile of another image: all per-lane state (DMA source offsets, ring
  // jobs, epilogue coordinates) is built once, a tile only changes the image (two base pointers).
  int b0;
  const int nwg = gridDim.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    b0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int ntl = (p.ntiles - b0 + nwg - 1) / nwg;          // tiles of this workgroup
  const int tpi = p.tiles_m * p.tiles_n;
  const int nt = b0 % p.tiles_n, mt = (b0 / p.tiles_n) % p.tiles_m, n0 = b0 / tpi;
  const int nstep = nwg / tpi;                              // images between two tiles of a workgroup (ntl > 1 only)
  const int oy0 = (mt / p.nbw) * 16, ox0 = (mt % p.nbw) * 16;

  // ---- halo pieces of this thread (box-invariant, resolved once) --------------------------------------------------------
  // piece = 16 B of one halo voxel; its source is pixel (ry, rx) of the 18 x 18 window around the box AFTER border handling
  // (reflect / clamp keep it inside the window) and channel piece `part`. Kept as 4 x 16 bits per thread in LDS —
  // ok << 15 | part << 10 | ry << 5 | rx — and read back once per chunk: four registers less in a loop that has none to spare.
  const size_t img_bytes = (size_t)d.Hi * d.Wi * d.in_cs * 2;
  {
    unsigned short* htab = reinterpret_cast<unsigned short*>(spare) + tid * HPW;
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
      const int q = (i * NW + wave) * 64 + lane;   // wave-instruction i*16+wave covers pieces [inst*64, inst*64+64)
      const int v = q / 10, part = q - v * 10;
      const int hy = v / 18, hx = v - hy * 18;
      bool ok = q < HPIECES && part < 8;
      int iy = border_index(oy0 + hy + p.hmin, d.Hi, d.border, ok);
      int ix = border_index(ox0 + hx + p.wmin, d.Wi, d.border, ok);
      iy = min(max(iy, 0), d.Hi - 1) - (oy0 + p.hmin);
      ix = min(max(ix, 0), d.Wi - 1) - (ox0 + p.wmin);
      htab[i] = (unsigned short)(ok ? (0x8000 | (part << 10) | (iy << 5) | ix) : 0);
    }
  }
  const unsigned htab0 = lds_addr(spare);
  // window origin of the box (may lie one pixel outside the image: only added to offsets of pixels inside it)
  const long long box0 = (((long long)(oy0 + p.hmin) * d.Wi + (ox0 + p.wmin)) * d.in_cs + d.in_co) * 2;
  auto issue_halo = [&](int n, int chunk, int buf) {        // image n
    const char* in_b = p.in + (size_t)n * img_bytes + box0 + chunk * 128;
    uint2 e;
    unsigned ta = (unsigned)tid;
    asm volatile("" : "+v"(ta));                // (address rebuilt per use: one register less across the loop)
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(htab0 + ta * (HPW * 2)) : "memory");
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
      const unsigned ent = ((i & 2) ? e.y : e.x) >> ((i & 1) * 16) & 0xffffu;
      unsigned off = (((ent >> 5) & 31u) * (unsigned)d.Wi + (ent & 31u)) * (unsigned)(d.in_cs * 2) + ((ent >> 10) & 15u) * 16u;
      asm volatile("" : "+v"(off));
      const char* src = (ent & 0x8000u) ? in_b + off : p.zero;
      const int inst = i * NW + wave;
      glds16(src, inst < HINSTR ? hbuf + buf * HBUF + inst * 1024 : sink);
    }
  };
  // ---- weight stage: one LDS-DMA instruction per wave (128 rows x 8 pieces), rows swizzled like gconv_kernel ----
  // (every row exists: Co is a multiple of 128 and the pack holds at least Co rows, see hconvw_eligible)
  unsigned wsrc[WPI];                            // byte offset of this lane's 16-B piece of K-step 0 inside the pack
  {
    // stage row r (= output channel nt*BN + r) keeps its 16-B pieces at slot ^ swz(r), swz(r) = bits 1, 3, 4 of r: with the
    // fragment rows below (a lane's 8 output channels contiguous) every ds_read_b128 lane group covers all 64 banks
    const int lrow = lane >> 3;
#pragma unroll
    for (int i = 0; i < WPI; ++i) {
      const int r = (wave * WPI + i) * 8 + lrow;
      const int wchunk = (lane & 7) ^ (((r >> 1) & 1) | (((r >> 3) & 1) << 1) | (((r >> 4) & 1) << 2));
      wsrc[i] = (unsigned)(((nt * BN + r) * d.Kp + wchunk * 8) * 2);
    }
  }
  auto issue_w = [&](int n, int c, int t, int buf) {        // image n picks the network in a twin batch
    const char* wbase = p.w + (n >= p.nsplit ? p.w_delta : 0);
    const unsigned q0 = (unsigned)((t * (d.Ci >> 3)) + c * 8) * 16u;   // first 8-k piece of this K-step (tap-major pack)
#pragma unroll
    for (int i = 0; i < WPI; ++i) {
      unsigned off = wsrc[i];
      asm volatile("" : "+v"(off));             // (keeps the compiler from holding one address per tap in registers)
      off += q0;
      glds16(wbase + off, wring + buf * WT + (wave * WPI + i) * 1024);
    }
  };

  const int wm = wave / WN, wn = wave % WN;
  int tb[T];                                     // tap byte offsets inside the halo
#pragma unroll
  for (int t = 0; t < T; ++t) tb[t] = (((int)d.dh[t] - p.hmin) * 18 + ((int)d.dw[t] - p.wmin)) * HP;

  f32x4 acc[TI][TJ];
  [[maybe_unused]] f32x4 accE[TI];

  // ---- main loop: chunks x taps, software-pipelined over half K-steps --------------------------------------------
  // Weights run 3 K-steps ahead in a 3-slot ring, halo boxes 2 chunks ahead in 2 buffers. Fragment reads go through
  // lds_read128 (common.hpp): the compiler would wait lgkmcnt(0) before every MFMA block because of the LDS-DMA in the loop.
  const int nk = p.chunks * T;
  const unsigned smem0 = lds_addr(smem);
  // Fragment addresses and the ring job are derived from an opaque copy of the lane id at the top of every tile, the
  // epilogue's coordinates from another one: the compiler then REBUILDS them (a few VALU) instead of keeping the K loop's
  // address registers alive across the epilogue and the epilogue's across the loop — 128 registers per lane is all a
  // 16-wave workgroup has, and a spill reload in the loop would wait vmcnt(0) behind the DMA stream.
  int rowb;                                      // halo byte offset of this lane's pixel in box row wm*4, k-chunk fk (the
                                                 // wave's other three rows and the second k-half are immediate offsets)
  unsigned woff, c0;
  int frow, fk;
  static_assert(3 * 18 * HP + 64 + 16 <= 65535, "row offsets must fit the ds_read immediate");
  auto load_frags = [&](unsigned wb, unsigned xb, auto kk_tag, bf16x8 (&wf)[TI], bf16x8 (&xf)[TJ]) {
    constexpr int kk = decltype(kk_tag)::value;
    const unsigned wa = wb + (kk ? (c0 ^ 64u) : c0);      // 16-B slot (kk * 4 + fk) ^ swz
    lds_read128<0>(wf[0], wa);
    lds_read128<512>(wf[1], wa);
    const unsigned xa = xb + (unsigned)rowb;
    lds_read128<0 * 18 * HP + kk * 64>(xf[0], xa);
    lds_read128<1 * 18 * HP + kk * 64>(xf[1], xa);
    lds_read128<2 * 18 * HP + kk * 64>(xf[2], xa);
    lds_read128<3 * 18 * HP + kk * 64>(xf[3], xa);
  };
  auto wait_frags = [&](bf16x8 (&wf)[TI], bf16x8 (&xf)[TJ]) {
    gs_lgkm_wait<0>(wf[0], wf[1], xf[0], xf[1], xf[2], xf[3]);
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  auto mma = [&](const bf16x8 (&wf)[TI], const bf16x8 (&xf)[TJ]) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
  };
  auto ct_of = [&](int ks, int& c, int& t) { c = ks / T; t = ks - c * T; };
  // ---- two wave groups, one phase apart ---------------------------------------------------------------------------
  // Measured on the barrier-per-K-step loop this replaced (profiles/r01_hconvw_pmc.txt): matrix pipe busy 45 %, LDS
  // array busy 42 %, and a K-step took the SUM of its fragment-read time and its MFMA time — the barrier phase-locks all
  // 16 waves, so everybody queues on the LDS and then everybody queues on the matrix pipe. Here a K-step is two phases
  // separated by barriers, L = issue the 2*(TI+TJ) fragment reads of the step and wait for them, M = its 2*TI*TJ MFMAs,
  // and the upper half of the waves executes ONE extra barrier up front: from then on one group is always in L while the
  // other is in M (each SIMD hosts waves of both groups), and priority is raised for the M phase.
  // LDS-DMA: weights of step ks+2 go into the slot of step ks-1 at the start of L(ks) (that slot's last reader finished
  // a phase ago), the halo of chunk c+1 into the buffer of chunk c-1 at the first L of chunk c; a wave waits for its
  // share of step ks+1's weights at the end of the last phase before the first reader (group 0: end of M(ks), group 1:
  // end of L(ks)) and the barrier that follows publishes it.
  const bool grp = wave >= NW / 2;
  issue_halo(n0, 0, 0);
  if (p.chunks > 1) issue_halo(n0, 1, 1);
#pragma unroll
  for (int s0 = 0; s0 < 3; ++s0)
    if (s0 < nk) { int c0_, t0; ct_of(s0, c0_, t0); issue_w(n0, c0_, t0, s0); }
  // halo 0 (and, in order, halo 1) and weights 0 landed; weights 1, 2 may still fly
  if (nk >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WPI) : "memory");
  else if (nk == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPI) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp) __builtin_amdgcn_s_barrier();
  const unsigned hbuf0 = smem0 + 3 * WT;
  bf16x8 wA[TI], xA[TJ], wB[TI], xB[TJ];
  [[maybe_unused]] bf16x8 xE0, xE1;
  [[maybe_unused]] const unsigned sink0 = smem0 + 3 * WT + 2 * HBUF;
  int stage = 0;                                 // ring slot of the current K-step (runs on across tiles)
  int hpar = 0;                                  // halo buffer of the current chunk (alternates across tiles too)
  // which image borders this box touches, as ONE scalar: every tile re-derives its flags from an opaque copy (the compiler
  // otherwise hoists each derived select out of the tile loop and runs out of scalar registers, then of vector ones)
  [[maybe_unused]] const int e_flags = __builtin_amdgcn_readfirstlane(
      (oy0 == 0 ? 1 : 0) | (oy0 + 16 == d.Ho ? 2 : 0) | (ox0 == 0 ? 4 : 0) | (ox0 + 16 == d.Wo ? 8 : 0));
#pragma clang loop unroll(disable)
  for (int it = 0; it < ntl; ++it) {
    const int n = n0 + it * nstep;               // this tile's image; the next tile's is n + nstep
    const bool has_next = it + 1 < ntl;
    {
      int ll = lane;
      asm volatile("" : "+v"(ll));
      frow = ll & 15; fk = ll >> 4;
      // MFMA row m of channel tile i is output channel wn*CWV + (m >> 2) * 8 + i * 4 + (m & 3): lane (fk = m >> 2 of the
      // accumulator rows it holds) ends up with the 8 CONTIGUOUS channels wn*CWV + fk*8 .. +8 over i = 0, 1 — 16-byte loads of
      // y / g2 in the fused epilogue, 16-byte slab writes — at no cost in the loop
      const int swz = (ll >> 1) & 7;
      rowb = ((wm * 4) * 18 + frow) * HP + fk * 16;
      woff = (unsigned)((wn * CWV + (frow >> 2) * 8 + (frow & 3)) * 128);
      c0 = (unsigned)((fk ^ swz) << 4);
    }
    const unsigned wring0 = smem0 + woff;
    // RING: this wave's ring fragment (lane frow = one ring pixel), its taps and its slot in the ring buffer;
    // taps come in the data-gradient order t = 3*ry + rx with (dh, dw) = (1 - ry, 1 - rx) (checked by the launcher)
    [[maybe_unused]] unsigned e_mask = 0;        // wave-uniform: taps that reach the image from this wave's ring pixels
    [[maybe_unused]] int e_rb = 0, e_slot = 0;
    [[maybe_unused]] bool e_lane = true;         // false: this lane carries no ring pixel (reads the zero sink)
    if constexpr (RING) {
      int ef = e_flags;
      asm volatile("" : "+s"(ef));
      const bool e_top = ef & 1, e_bot = ef & 2, e_lef = ef & 4, e_rig = ef & 8;
      int py = 0, px = 0, sy = 0, sx = 0;        // ring pixel of lane frow: (py + sy*frow, px + sx*frow), box coordinates
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
      if (!APPLY && (p.dbg & 1)) e_mask = 0;      // (ring_dbg ablation; APPLY's dbg bits mean something else)
      e_mask = __builtin_amdgcn_readfirstlane(e_mask);
      e_rb = ((py + sy * frow) * 18 + (px + sx * frow)) * HP + fk * 16;
#pragma unroll
      for (int i = 0; i < TI; ++i) accE[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma clang loop unroll(disable)
    for (int c = 0; c < p.chunks; ++c) {
      const unsigned hb = hbuf0 + (unsigned)(hpar * HBUF);
      const bool last_chunk = c + 1 == p.chunks;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        // ---- L(ks), ks = c * T + t ----
        // halo of the chunk after this one (of the NEXT tile behind the last chunk) into the other buffer; the first chunk
        // of the launch finds chunk 1 issued by the prologue, the first chunk of a later tile finds the other buffer freed
        // by the epilogue that just ran out of it
        const bool halo_now = t == 0 && (c >= 1 || it > 0) && (!last_chunk || has_next);
        if (halo_now) issue_halo(last_chunk ? n + nstep : n, last_chunk ? 0 : c + 1, hpar ^ 1);
        // weights two K-steps ahead (K-steps 0 / 1 of the next tile behind this tile's last two; K-step 2 of a later tile
        // is issued in front of the previous tile's epilogue stores)
        const bool tail2 = last_chunk && t >= T - 2;          // K-step ks + 2 belongs to the next tile
        const bool w_now = (c > 0 || t >= 1) && (!tail2 || has_next);
        if (w_now) {
          if (tail2) issue_w(n + nstep, 0, t - (T - 2), stage == 0 ? 2 : stage - 1);
          else issue_w(n, t + 2 >= T ? c + 1 : c, t + 2 >= T ? t + 2 - T : t + 2, stage == 0 ? 2 : stage - 1);
        }
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
        wait_frags(wA, xA);
        wait_frags(wB, xB);
        if constexpr (RING) {
          if (e_now) { reg_fence(xE0); reg_fence(xE1); }
        }
        auto wait_next_weights = [&]() {       // this wave's share of step ks+1's weights (and anything older) has landed
          if (last_chunk && t == T - 1 && !has_next) return;
          if (it > 0 && c == 0 && t <= 1) {
            // first K-steps behind a tile boundary: the weights waited for are older than the previous tile's NST output
            // stores and this tile's chunk-1 halo (ks 0: w2 | stores | halo outstanding; ks 1: stores | halo | w3)
            if (APPLY && p.out2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPI + 2 * NST + HPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPI + NST + HPW) : "memory");
            return;
          }
          if (!w_now && (c > 0 || t >= 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tail of the last tile
          else if (halo_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HPW + WPI) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPI) : "memory");               // (ks 0 of the launch: w1 | w2)
        };
        if (grp) wait_next_weights();
        __builtin_amdgcn_s_barrier();
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
        if (!grp) wait_next_weights();
        __builtin_amdgcn_s_barrier();
        stage = stage == 2 ? 0 : stage + 1;
      }
      hpar ^= 1;
    }
    // ---- tile boundary -----------------------------------------------------------------------------------------------------
    // Epilogue coordinates (rebuilt from the lane id, see above) and its GLOBAL operands first: the loads are issued before
    // the barrier that brings the two wave groups back into step, so their latency runs under that wait; everything the
    // epilogue reads from memory is in registers (or on its way) before the first output store is issued, and no wait
    // behind the stores ever names them — they drain under the next tile's K loop.
    {
      int le = lane;
      asm volatile("" : "+v"(le));
      frow = le & 15; fk = le >> 4;
    }
    const int elane = fk * 16 + frow;             // (= lane, rebuilt)
    const size_t pix0 = ((size_t)n * d.Ho + oy0) * d.Wo + ox0;
    constexpr int SROW = CWV * 2 + 16, PH = 32;             // slab row pitch; pixels per pass (two of the wave's four box rows)
    constexpr int LPR = CWV / 8, PPI = 64 / LPR;            // 4 lanes per pixel, 16 pixels per store instruction
    constexpr int NQ = PH / PPI;                            // store instructions per pass (2)
    const int sub = elane % LPR, prow = elane / LPR;
    // (store q of pass ph covers two of the wave's four box rows per pass, 16 pixels per instruction: see the stores below)
    static_assert(NQ == 2, "four store values per lane");
    uint4 val0, val1, val2, val3;                  // the tile in store layout: 4 x 16 B per lane (named: an indexed array
                                                   // ended up in scratch memory)
    [[maybe_unused]] uint4 tot0, tot1, tot2, tot3; // APPLY: the total gradient gx + g2 of the same four pixels
    [[maybe_unused]] uint4 yv[TJ], gv[TJ];         // y / g2 of the lane's 8 channels (.x .y: tile i = 0, .z .w: i = 1), per box row
    [[maybe_unused]] f32x4 mrv;
    [[maybe_unused]] f32x4 bia[TI];
    // y / g2 of the consumer's norm backward in the ACCUMULATOR layout (pixel (wm*4 + j, frow), channels i*16 + fk*4 ..):
    // the sums are then taken before the tile goes through the store slabs, i.e. before the first output store
    // (32-bit lane offsets from a wave-uniform per-image base: the 64-bit form cost ~20 VALU instructions per load)
    const size_t oimg = (size_t)n * d.Ho * d.Wo * d.Co * 2;
    auto load_yg = [&]() {
      const char* yb = static_cast<const char*>(p.f.y) + oimg;
      const char* gb = static_cast<const char*>(p.f.g2) + oimg;
      const unsigned o0 = (unsigned)(((oy0 + wm * 4) * d.Wo + ox0 + frow) * d.Co + nt * BN + wn * CWV + fk * 8) * 2u;
      const unsigned rowb2 = (unsigned)(d.Wo * d.Co) * 2u;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const unsigned o = o0 + (unsigned)j * rowb2;
        if (!APPLY && (p.dbg & 2)) { yv[j] = uint4{o, o, o0, o}; gv[j] = uint4{o0, o, o, o0}; continue; }
        yv[j] = *reinterpret_cast<const uint4*>(yb + o);
        gv[j] = p.f.g2 ? *reinterpret_cast<const uint4*>(gb + o) : uint4{0u, 0u, 0u, 0u};
      }
    };
    if constexpr (RING) {
      load_yg();
      // mean / rstd of this tile's 128 channels: 64 threads fetch 4 floats each, everybody reads them back from LDS
      if (tid < 2 * BN / 4) {
        const float* mr = p.f.mean_rstd + (size_t)n * 2 * d.Co + (tid >= BN / 4 ? d.Co : 0) + nt * BN + (tid % (BN / 4)) * 4;
        mrv = *reinterpret_cast<const f32x4*>(mr);
      }
    } else {
      const float* bias_n = p.bias ? p.bias + (n >= p.nsplit ? p.bias_delta : 0) : nullptr;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int cb = nt * BN + wn * CWV + fk * 8 + i * 4;
        bia[i] = bias_n ? *reinterpret_cast<const f32x4*>(bias_n + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    // both groups back in step; every LDS operand of this tile has been read
    if (!grp) __builtin_amdgcn_s_barrier();       // group 1 ran one barrier ahead of the loop
    // the epilogue works out of the halo buffer the next chunk does NOT use: hpar now names the next tile's chunk-0
    // buffer (already filled), the other one is free
    char* const ebuf = hbuf + (hpar ^ 1) * HBUF;
    static_assert(NW * PH * SROW + WM * BN * 3 * 4 + 4 * BN * 4 <= HBUF, "epilogue scratch must fit one halo buffer");
    char* const slab = ebuf + wave * (PH * SROW);            // 16 x 2560 B = 40 KiB
    float* const red = reinterpret_cast<float*>(ebuf + NW * PH * SROW);                       // [WM][BN][2 | 3], behind the slabs
    [[maybe_unused]] float* const mrs = reinterpret_cast<float*>(ebuf + NW * PH * SROW + WM * BN * 3 * 4);   // [2][BN]
    // RING: zeros behind everything else — what the lanes that are NOT on a fold column read in the ring pre-pass
    constexpr int ZPAD = ((TJ - 1) * BN + (TI - 1) * 4 + 4) * 4;
    static_assert(NW * PH * SROW + WM * BN * 3 * 4 + 4 * BN * 4 + (ZPAD + 15) / 16 * 16 <= HBUF, "zero pad must fit the halo buffer");
    [[maybe_unused]] float* const zpad = reinterpret_cast<float*>(ebuf + NW * PH * SROW + WM * BN * 3 * 4 + 4 * BN * 4);
    if constexpr (RING) {
      // ---- ring sums -> LDS -> added (fp32) to the pixels they fold onto; bf16 tile through the per-wave slabs; coalesced
      // stores with the consumer's InstanceNorm-backward sums (contract of gconv_kernel's fused epilogue: sums over the
      // box of ghat = (g + g2) * act'(yhat), ghat * yhat, yhat; one slot per box) -------------------------------------
      float* ringbuf = reinterpret_cast<float*>(ebuf);       // [5 slots: top, bottom, left, right, corner][16 pixels][BN] fp32
      if (e_mask) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
          *reinterpret_cast<f32x4*>(ringbuf + (e_slot * 16 + frow) * BN + wn * CWV + fk * 8 + i * 4) = accE[i];
      }
      if (tid < 2 * BN / 4) *reinterpret_cast<f32x4*>(mrs + tid * 4) = mrv;
      if (tid >= 512 && tid < 512 + (ZPAD + 15) / 16) {
        float z0;                                  // (made here: a zero vector hoisted out of the tile loop went to scratch memory)
        asm volatile("v_mov_b32 %0, 0" : "=v"(z0));
        *reinterpret_cast<f32x4*>(zpad + (tid - 512) * 4) = f32x4{z0, z0, z0, z0};
      }
      lds_barrier();
      uint2 pk[TI][TJ];
      // Ring sums onto the pixels they fold onto, in place in the accumulators (fp32, fixed order: row, column, corner). Which
      // wave / box row takes what is wave-uniform — top row: wm 0, j 1; bottom row: wm 3, j 2; the corner lands on the same
      // row as the row job — so these are scalar branches; the column (pixel column 1 or 14 of every box row) is one read per
      // box row whose ADDRESS selects the lane's column: the other lanes read zeros (zpad). (They were five exec-masked
      // conditional reads per (i, j), each with its own LDS round trip.)
      {
        int ef = e_flags;
        asm volatile("" : "+s"(ef));
        const bool e_top = ef & 1, e_bot = ef & 2, e_lef = ef & 4, e_rig = ef & 8;
        const bool noadd = !APPLY && (p.dbg & 8);
        const bool rowtop = e_top && wm == 0 && !noadd, rowbot = e_bot && wm == 3 && !noadd;
        const bool colside = (e_lef || e_rig) && !noadd;
        const bool oncol = frow == (e_lef ? 1 : 14);
        const float* colp = oncol ? ringbuf + ((e_lef ? 2 : 3) * 16 + wm * 4) * BN + wn * CWV + fk * 8 : zpad;
        const float* corp = oncol ? ringbuf + (4 * 16) * BN + wn * CWV + fk * 8 : zpad;
        const float* rowp = ringbuf + ((rowtop ? 0 : 1) * 16 + frow) * BN + wn * CWV + fk * 8;
        if (rowtop) {
#pragma unroll
          for (int i = 0; i < TI; ++i) acc[i][1] += *reinterpret_cast<const f32x4*>(rowp + i * 4);
        }
        if (rowbot) {
#pragma unroll
          for (int i = 0; i < TI; ++i) acc[i][2] += *reinterpret_cast<const f32x4*>(rowp + i * 4);
        }
        if (colside) {
#pragma unroll
          for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(colp + j * BN + i * 4);
          if (rowtop) {
#pragma unroll
            for (int i = 0; i < TI; ++i) acc[i][1] += *reinterpret_cast<const f32x4*>(corp + i * 4);
          }
          if (rowbot) {
#pragma unroll
            for (int i = 0; i < TI; ++i) acc[i][2] += *reinterpret_cast<const f32x4*>(corp + i * 4);
          }
        }
      }
      // The sums are VALU work of every lane (16 waves x 32 elements each) with nothing else running on the CU: the
      // instruction count IS the time (measured: 2400 -> 1100 instructions took the launch from 97 to 90 us). Two channels
      // per instruction where the ISA has packed fp32 (v_pk_add / v_pk_mul / v_pk_fma), act'(yhat) as one compare + select:
      // 1 above zero and `neg` below (none: 1, relu: 0, lrelu: slope; tanh never sits in front of this layer — the launcher
      // refuses it).
      const float neg = p.f.act == GS_ACT_RELU ? 0.f : (p.f.act == GS_ACT_LRELU ? p.f.slope : 1.f);
      const f32x2 neg2 = {neg, neg};
      const bool has_g2 = p.f.g2 != nullptr;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int cl = wn * CWV + fk * 8 + i * 4;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mrs + cl), rs = *reinterpret_cast<const f32x4*>(mrs + BN + cl);
        const f32x2 mu2[2] = {{mu[0], mu[1]}, {mu[2], mu[3]}}, rs2[2] = {{rs[0], rs[1]}, {rs[2], rs[3]}};
        f32x2 s1[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2] = {{0.f, 0.f}, {0.f, 0.f}}, s3[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const f32x4 v = acc[i][j];
          pk[i][j].x = pack_bf2(v[0], v[1]);                 // rounded here: the sums see the gradient as it is stored
          pk[i][j].y = pack_bf2(v[2], v[3]);
          if (!APPLY && (p.dbg & 4)) { s1[0].x += bf_lo(yv[j].x ^ gv[j].y); continue; }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const unsigned pw = h ? pk[i][j].y : pk[i][j].x;
            const unsigned yw = i ? (h ? yv[j].w : yv[j].z) : (h ? yv[j].y : yv[j].x);
            const unsigned gw = i ? (h ? gv[j].w : gv[j].z) : (h ? gv[j].y : gv[j].x);
            f32x2 g = {bf_lo(pw), bf_hi(pw)};
            const f32x2 yq = {bf_lo(yw), bf_hi(yw)};
            if (has_g2) g += f32x2{bf_lo(gw), bf_hi(gw)};
            const f32x2 yh = (yq - mu2[h]) * rs2[h];
            const f32x2 gn = g * neg2;
            const f32x2 gh = {yh.x > 0.f ? g.x : gn.x, yh.y > 0.f ? g.y : gn.y};
            s1[h] += gh;
            s2[h] += gh * yh;
            s3[h] += yh;
          }
        }
        // per-wave sums over the 16 pixel columns, then over the four pixel-row groups of waves through LDS (fixed order)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row16_sum(s1[r >> 1][r & 1]), b2 = row16_sum(s2[r >> 1][r & 1]), c3 = row16_sum(s3[r >> 1][r & 1]);
          if (frow == 0) {
            red[(wm * BN + cl + r) * 3 + 0] = a;
            red[(wm * BN + cl + r) * 3 + 1] = b2;
            red[(wm * BN + cl + r) * 3 + 2] = c3;
          }
        }
      }
      lds_barrier();                                         // the ring sums have been read: the slabs may overwrite them
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        static_assert(TI == 2, "a lane's two channel tiles make one 16-byte slab write");
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          *reinterpret_cast<uint4*>(slab + (jj * 16 + frow) * SROW + fk * 16) =
              uint4{pk[0][ph * 2 + jj].x, pk[0][ph * 2 + jj].y, pk[1][ph * 2 + jj].x, pk[1][ph * 2 + jj].y};
        __builtin_amdgcn_wave_barrier();                     // wave-private slab: LDS operations of a wave complete in order
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const uint4 v = *reinterpret_cast<const uint4*>(slab + (q * PPI + prow) * SROW + sub * 16);
          if (ph == 0 && q == 0) val0 = v; else if (ph == 0) val1 = v; else if (q == 0) val2 = v; else val3 = v;
        }
        __builtin_amdgcn_wave_barrier();
      }
      int tq = tid;
      asm volatile("" : "+v"(tq));                           // (addresses below are built here, not carried through the tile)
      if (tq < BN) {                                         // (red was published by the barrier in front of the slab passes)
        const int cc = nt * BN + tq;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          t0 += red[(w * BN + tq) * 3]; t1 += red[(w * BN + tq) * 3 + 1]; t2 += red[(w * BN + tq) * 3 + 2];
        }
        float* sp = p.f.partial + ((size_t)n * p.tiles_m + mt) * 3 * d.Co;
        if constexpr (APPLY) {     // written through (sc1): read by the other boxes' workgroups of this launch
          __hip_atomic_store(sp + cc, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(sp + d.Co + cc, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(sp + 2 * d.Co + cc, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          sp[cc] = t0; sp[d.Co + cc] = t1; sp[2 * d.Co + cc] = t2;
        }
      }
      if constexpr (APPLY) {
        // ---- rendezvous of the boxes of (image n, channel tile nt), then the norm backward on this box ------------------------
        float* const tots = mrs + 2 * BN;                     // [2][BN]: mean ghat, mean ghat * yhat
        // y / g2 of this lane's four stores (8 channels of one pixel each, coalesced) are requested before the rendezvous: their
        // latency runs under it
        uint4 y0, y1, y2, y3, q0 = {0u, 0u, 0u, 0u}, q1 = q0, q2 = q0, q3 = q0;
        {
          const char* yb = static_cast<const char*>(p.f.y);
          const char* gb = static_cast<const char*>(p.f.g2);
          // (the offsets are built from an opaque copy of the lane id: shared with the store addresses below they stayed live —
          // and went to scratch memory — through the whole apply pass)
          int l2 = lane;
          asm volatile("" : "+v"(l2));
          const int sub2 = l2 % LPR, prow2 = l2 / LPR;
          auto eoff = [&](int ph, int q) {
            const int pl = q * PPI + prow2;
            const size_t px = pix0 + (size_t)(wm * 4 + ph * 2 + (pl >> 4)) * d.Wo + (pl & 15);
            return (px * d.Co + nt * BN + wn * CWV + sub2 * 8) * 2;
          };
          const size_t e0 = eoff(0, 0), e1 = eoff(0, 1), e2 = eoff(1, 0), e3 = eoff(1, 1);
          y0 = *reinterpret_cast<const uint4*>(yb + e0); y1 = *reinterpret_cast<const uint4*>(yb + e1);
          y2 = *reinterpret_cast<const uint4*>(yb + e2); y3 = *reinterpret_cast<const uint4*>(yb + e3);
          if (has_g2) {
            q0 = *reinterpret_cast<const uint4*>(gb + e0); q1 = *reinterpret_cast<const uint4*>(gb + e1);
            q2 = *reinterpret_cast<const uint4*>(gb + e2); q3 = *reinterpret_cast<const uint4*>(gb + e3);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's partial sums are on their way out of the CU ...
        lds_barrier();                                        // ... and so are everybody's
        if (tid == 0 && (p.dbg & 16)) {                        // debug: arrival time of this tile (tools/probe/archive/e8_ring_skew.py)
          const unsigned long long t = __builtin_amdgcn_s_memrealtime();      // (100 MHz, one clock for all XCDs)
          const int tile = (n * p.tiles_m + mt) * p.tiles_n + nt;
          p.sync[256 + 2 * tile] = (int)(t & 0xffffffffu);
          p.sync[256 + 2 * tile + 1] = (int)(t >> 32);
        }
        if (tid == 0 && !(p.dbg & 8)) {
          int* cnt = p.sync + 2 * (n * p.tiles_n + nt);
          __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int spins = 0;
          while (!(p.dbg & 2) && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.tiles_m) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 22)) {                        // a grid that is not fully resident: flag it and go on (wrong
              __hip_atomic_store(p.sync + 2 * d.N * p.tiles_n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // results,
              break;                                          // no hang)
            }
          }
          // the last one to leave puts the counters back for the next launch
          if (__hip_atomic_fetch_add(cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.tiles_m - 1) {
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        lds_barrier();
        if (tid < 3 * BN && !(p.dbg & 4)) {
          const int r = tid / BN, ch = tid - r * BN;
          const float* src = p.f.partial + ((size_t)n * p.tiles_m * 3 + r) * d.Co + nt * BN + ch;
          double acc2 = 0.0;                                  // slot order, double accumulation: inorm_bwd_apply_cg_kernel's sum
          int sl = 0;
          for (; sl + 8 <= p.tiles_m; sl += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
              v[k] = __hip_atomic_load(src + (size_t)(sl + k) * 3 * d.Co, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc2 += (double)v[k];
          }
          for (; sl < p.tiles_m; ++sl)
            acc2 += (double)__hip_atomic_load(src + (size_t)sl * 3 * d.Co, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const float tot = (float)acc2;
          if (r < 2) tots[r * BN + ch] = tot * p.inv_hw;
          // per-image totals for the bias gradient of the conv in front of the norm (gs_norm_bias_grads), once per image
          if (mt == 0) p.f.partial[((size_t)d.N * p.tiles_m + n) * 3 * d.Co + (size_t)r * d.Co + nt * BN + ch] = tot;
        }
        lds_barrier();
        // dy (into the val registers) and the total gradient gx + g2 (into the q registers) of the four stores: the arithmetic
        // of the apply pass, pixel by pixel
        {
          const int cl8 = wn * CWV + sub * 8;
          auto apply4 = [&](uint4& v, const uint4& yv4, uint4& gq) {
            const int c8 = cl8;
            unsigned vw[4] = {v.x, v.y, v.z, v.w};
            unsigned gw[4] = {gq.x, gq.y, gq.z, gq.w};
            const unsigned yw[4] = {yv4.x, yv4.y, yv4.z, yv4.w};
#pragma unroll
            for (int h = 0; h < 4; ++h) {                     // two channels at a time: the constants come from LDS as they are used
              const f32x2 mu2 = *reinterpret_cast<const f32x2*>(mrs + c8 + 2 * h), rs2 = *reinterpret_cast<const f32x2*>(mrs + BN + c8 + 2 * h);
              const f32x2 a2 = *reinterpret_cast<const f32x2*>(tots + c8 + 2 * h), b2 = *reinterpret_cast<const f32x2*>(tots + BN + c8 + 2 * h);
              float g[2] = {bf_lo(vw[h]), bf_hi(vw[h])};
              if (has_g2) {
                g[0] += bf_lo(gw[h]); g[1] += bf_hi(gw[h]);
                gw[h] = pack_bf2(g[0], g[1]);
              }
              const float yy[2] = {bf_lo(yw[h]), bf_hi(yw[h])};
              float dd[2];
#pragma unroll
              for (int k = 0; k < 2; ++k) {
                const float yh = (yy[k] - mu2[k]) * rs2[k];
                dd[k] = inorm_dy(g[k], yh, yh > 0.f ? 1.f : neg, a2[k], b2[k], rs2[k]);   // (inorm_bwd_apply_cg_kernel's arithmetic)
              }
              vw[h] = pack_bf2(dd[0], dd[1]);
            }
            v.x = vw[0]; v.y = vw[1]; v.z = vw[2]; v.w = vw[3];
            gq.x = gw[0]; gq.y = gw[1]; gq.z = gw[2]; gq.w = gw[3];
          };
          apply4(val0, y0, q0);
          apply4(val1, y1, q1);
          apply4(val2, y2, q2);
          apply4(val3, y3, q3);
          tot0 = q0; tot1 = q1; tot2 = q2; tot3 = q3;
        }
      }
    } else {
      // ---- bias, partial statistics (slot = box), activation, LDS-staged coalesced NHWC stores ------------------------
      const bool want_stats = d.stats_slots > 0;
      float s1[TI][4], s2[TI][4];
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
      // (the activation mode is uniform: the usual case — none, the norm follows — skips the per-element switch)
      const bool plain = d.act == GS_ACT_NONE;
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int i = 0; i < TI; ++i) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = acc[i][ph * 2 + jj][r] + bia[i][r];
              s1[i][r] += v[r];
              s2[i][r] += v[r] * v[r];
            }
            if (!plain) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], d.act, d.slope);
            }
            uint2 o;
            o.x = pack_bf2(v[0], v[1]);
            o.y = pack_bf2(v[2], v[3]);
            *reinterpret_cast<uint2*>(slab + (jj * 16 + frow) * SROW + (fk * 8 + i * 4) * 2) = o;
          }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < NQ; ++q)               // pixel q * PPI + prow of this pass's two box rows
        {
          const uint4 v = *reinterpret_cast<const uint4*>(slab + (q * PPI + prow) * SROW + sub * 16);
          if (ph == 0 && q == 0) val0 = v; else if (ph == 0) val1 = v; else if (q == 0) val2 = v; else val3 = v;
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (want_stats) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float a = s1[i][r], q = s2[i][r];
            a = row16_sum(a);
            q = row16_sum(q);
            if (frow == 0) {
              const int cl = wn * CWV + fk * 8 + i * 4 + r;
              red[(wm * BN + cl) * 2 + 0] = a;
              red[(wm * BN + cl) * 2 + 1] = q;
            }
          }
        lds_barrier();
        if (tid < BN) {
          const int cc = nt * BN + tid;
          float a = 0.f, q = 0.f;
#pragma unroll
          for (int w = 0; w < WM; ++w) { a += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
          float* sp = p.stats + (((size_t)n * d.stats_slots + d.stats_slot0 + mt) * 2) * d.Co;
          sp[cc] = a;
          sp[d.Co + cc] = q;
        }
      }
    }
    // Everything the epilogue reads from LDS or memory has been read. Now, in this order: K-step 2 of the next tile into the
    // slot of this tile's last step (an LDS-DMA in front of any LDS access would make the compiler wait vmcnt(0) there),
    // then the tile's four output stores, which nothing ever waits for (see the counted waits of the next tile's first
    // K-steps: the stores are younger than these weights and older than the chunk-1 halo).
    if (has_next) {
      lds_barrier();                              // slabs and sums read by everybody: the next chunk-1 halo may land there
      int c2, t2;
      ct_of(2, c2, t2);
      issue_w(n + nstep, c2, t2, stage == 0 ? 2 : stage - 1);
    }
    if constexpr (!APPLY) {
      // (32-bit lane offsets from the image's base, rebuilt here from an opaque lane id: nothing of them lives through the tile)
      char* const ob = p.out + (size_t)n * d.Ho * d.Wo * d.out_cs * 2;
      int ls = lane;
      asm volatile("" : "+v"(ls));
      const int prow_s = ls / LPR, sub_s = ls % LPR;
      auto ooff = [&](int ph, int q) {
        const int pl = q * PPI + prow_s;
        const int px = (oy0 + wm * 4 + ph * 2 + (pl >> 4)) * d.Wo + ox0 + (pl & 15);
        return (unsigned)(px * d.out_cs + d.out_co + nt * BN + wn * CWV + sub_s * 8) * 2u;
      };
      *reinterpret_cast<uint4*>(ob + ooff(0, 0)) = val0;
      *reinterpret_cast<uint4*>(ob + ooff(0, 1)) = val1;
      *reinterpret_cast<uint4*>(ob + ooff(1, 0)) = val2;
      *reinterpret_cast<uint4*>(ob + ooff(1, 1)) = val3;
    } else {
      // (store offsets from a fresh opaque lane id: computed up front they lived — in scratch memory — through the apply pass;
      // dy and the total gradient are dense [pixel][Co] tensors)
      int l3 = lane;
      asm volatile("" : "+v"(l3));
      const int sub3 = l3 % LPR, prow3 = l3 / LPR;
      auto soff = [&](int ph, int q) {
        const int pl = q * PPI + prow3;
        const size_t px = pix0 + (size_t)(wm * 4 + ph * 2 + (pl >> 4)) * d.Wo + (pl & 15);
        return (px * d.Co + nt * BN + wn * CWV + sub3 * 8) * 2;
      };
      const size_t o0 = soff(0, 0), o1 = soff(0, 1), o2 = soff(1, 0), o3 = soff(1, 1);
      *reinterpret_cast<uint4*>(p.out + o0) = val0;
      *reinterpret_cast<uint4*>(p.out + o1) = val1;
      *reinterpret_cast<uint4*>(p.out + o2) = val2;
      *reinterpret_cast<uint4*>(p.out + o3) = val3;
      if (p.out2) {                                           // (uniform: four more stores per lane, see the counted waits)
        *reinterpret_cast<uint4*>(p.out2 + o0) = tot0;
        *reinterpret_cast<uint4*>(p.out2 + o1) = tot1;
        *reinterpret_cast<uint4*>(p.out2 + o2) = tot2;
        *reinterpret_cast<uint4*>(p.out2 + o3) = tot3;
      }
    }
    if (has_next && grp) __builtin_amdgcn_s_barrier();      // group 1 one barrier ahead again
  }
}

static bool hconvw_eligible(const gs_gconv_desc* d, int* lo) {
  const bool enabled = gs_opt(GS_OPT_HCONV_WIDE) != 0;
  if (!enabled || d->si != 1 || d->so != 1 || d->T != 9 || d->Ci % 64 != 0 || d->Co % 128 != 0 || d->accumulate) return false;
  if (d->Di != 1 || d->Do != 1 || d->Dc != 1 || d->Hc != d->Ho || d->Wc != d->Wo || d->py || d->px || d->pz) return false;
  if (d->Ho % 16 != 0 || d->Wo % 16 != 0 || d->w_rows < d->Co) return false;
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
  if ((long long)d->Ho * d->Wo * (d->out_cs > d->Co ? d->out_cs : d->Co) * 2 >= (1LL << 31)) return false;   // 32-bit offsets per image
  return true;
}

// partial-statistics slots per image when the layer runs here (one per 16x16 box), 0 when it does not
int gs_hconvw_slots(const gs_gconv_desc* d) {
  int lo[2];
  return hconvw_eligible(d, lo) ? (d->Ho / 16) * (d->Wo / 16) : 0;
}

// Grid of a launch over `images` x `tpi` tiles (tpi = boxes x channel tiles of one image): one workgroup per tile while
// they fit in one round of the chip (one workgroup per CU: 160 KiB of LDS); else the persistent form — G = floor(CUs / tpi)
// images per round, ceil(images / G) tiles per workgroup, and a grid that is a MULTIPLE of tpi, so that all tiles of a
// workgroup are the same box and channel tile of different images (the kernel builds its per-lane state once). It needs
// >= 2 channel chunks (a tile's last chunk leaves one halo buffer for the next tile's first). Option hconvw_persist = 0:
// always one tile per workgroup.
static int hconvw_grid(int images, int tpi, int chunks) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  const long long tiles = (long long)images * tpi;
  if (tiles <= cus || tpi > cus || chunks < 2 || gs_opt(GS_OPT_HCONVW_PERSIST) == 0) return (int)tiles;
  const int G = cus / tpi;                                 // images per round
  const int per = (images + G - 1) / G;                    // tiles per workgroup
  return tpi * ((images + per - 1) / per);
}

static void hconvw_twin(HConvWK& k, const gs_twin* tw) {
  k.nsplit = tw ? tw->n_split : 0x7fffffff;
  k.w_delta = tw ? tw->w_delta : 0;
  k.bias_delta = tw ? tw->bias_delta / 4 : 0;
}

// returns 0 and sets *handled when the layer ran here
int gs_hconvw_try(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out,
                  float* stats, const gs_twin* tw, void* stream, int* handled) {
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
  k.ntiles = (int)blocks;
  hconvw_twin(k, tw);
  k.d = *d;
  k.f = gs_gconv_fuse{};
  k.out2 = nullptr; k.sync = nullptr; k.inv_hw = 0.f; k.dbg = 0;
  const int lds = 160 * 1024;
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvw_kernel<9>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  *handled = 1;
  hipLaunchKernelGGL((hconvw_kernel<9>), dim3((unsigned)hconvw_grid(d->N, k.tiles_m * k.tiles_n, k.chunks)), dim3(1024), lds,
                     static_cast<hipStream_t>(stream), k);
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
                   const gs_twin* tw, void* stream) {
  GS_REQUIRE(hconvw_ring_eligible(d), "gs_gconv_forward_fused: layer is not eligible for the unpadded (ring) form, see "
                                      "gs_gconv_ring_slots");
  GS_REQUIRE(fuse->fold == 1 && fuse->fold_mode == GS_BORDER_REFLECT && fuse->Dy == 1,
             "gs_gconv_forward_fused: the unpadded form folds a reflect padding of 1");
  GS_REQUIRE(fuse->act != GS_ACT_TANH, "gs_gconv_forward_fused: the unpadded form takes none / relu / lrelu in front of the "
                                       "consumer's norm (tanh: use the padded form)");
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
  k.out2 = nullptr; k.sync = nullptr; k.inv_hw = 0.f; k.dbg = gs_opt(GS_OPT_RING_DBG);
  const long long blocks = (long long)d->N * k.tiles_m * k.tiles_n;
  k.ntiles = (int)blocks;
  hconvw_twin(k, tw);
  const int lds = 160 * 1024;
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvw_kernel<9, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  hipLaunchKernelGGL((hconvw_kernel<9, true>), dim3((unsigned)hconvw_grid(d->N, k.tiles_m * k.tiles_n, k.chunks)), dim3(1024), lds,
                     static_cast<hipStream_t>(stream), k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- RING + APPLY: the fused data gradient with the consumer's whole InstanceNorm backward in the launch -----------------------
// int32 words of the rendezvous buffer the launch wants (zero-filled once by the caller, left zero by every launch), 0 when the
// launch cannot run in this form: it needs every workgroup resident at once (a persistent grid of at most one workgroup per CU)
extern "C" int gs_gconv_ring_apply_words(const gs_gconv_desc* d) {
  if (!d || !gs_opt(GS_OPT_RING_APPLY) || !hconvw_ring_eligible(d)) return 0;
  const int tpi = (d->Ho / 16) * (d->Wo / 16) * (d->Co / 128);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  if (hconvw_grid(d->N, tpi, d->Ci / 64) > cus) return 0;
  return 2 * d->N * (d->Co / 128) + 1;
}

extern "C" int gs_gconv_ring_apply(const gs_gconv_desc* d, const void* in, const void* w_pack, const gs_gconv_fuse* fuse, void* dy,
                                   void* total, int32_t* sync, const gs_twin* tw, void* stream) {
  GS_REQUIRE(d && in && w_pack && fuse && dy && sync && fuse->y && fuse->mean_rstd && fuse->partial, "gs_gconv_ring_apply: null argument");
  GS_REQUIRE(gs_gconv_ring_apply_words(d) > 0, "gs_gconv_ring_apply: the launch does not qualify (gs_gconv_ring_apply_words)");
  GS_REQUIRE(fuse->fold == 1 && fuse->fold_mode == GS_BORDER_REFLECT && fuse->Dy == 1 && fuse->Hy == d->Ho && fuse->Wy == d->Wo,
             "gs_gconv_ring_apply: the unpadded form folds a reflect padding of 1");
  GS_REQUIRE(fuse->act != GS_ACT_TANH, "gs_gconv_ring_apply: none / relu / lrelu in front of the consumer's norm");
  GS_REQUIRE(!total || fuse->g2, "gs_gconv_ring_apply: the total gradient is gx + g2 (without g2 it is not produced: dy only)");
  HConvWK k;
  k.in = static_cast<const char*>(in);
  k.w = static_cast<const char*>(w_pack);
  k.bias = nullptr;
  k.out = static_cast<char*>(dy);
  k.out2 = static_cast<char*>(total);
  k.sync = sync;
  k.stats = nullptr;
  k.zero = static_cast<const char*>(gs_zero_page());
  GS_REQUIRE(k.zero, "gs_gconv_forward: library not initialised (call gs_init)");
  k.tiles_m = (d->Ho / 16) * (d->Wo / 16);
  k.tiles_n = d->Co / 128;
  k.nbw = d->Wo / 16;
  k.hh = 18; k.hw = 18; k.hmin = -1; k.wmin = -1;
  k.chunks = d->Ci / 64;
  k.inv_hw = 1.0f / (float)(d->Ho * d->Wo);
  k.dbg = gs_opt(GS_OPT_RING_APPLY);
  k.d = *d;
  k.f = *fuse;
  k.ntiles = d->N * k.tiles_m * k.tiles_n;
  hconvw_twin(k, tw);
  const int lds = 160 * 1024;
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvw_kernel<9, true, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  hipLaunchKernelGGL((hconvw_kernel<9, true, true>), dim3((unsigned)hconvw_grid(d->N, k.tiles_m * k.tiles_n, k.chunks)), dim3(1024),
                     lds, static_cast<hipStream_t>(stream), k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
