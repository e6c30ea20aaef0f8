// Halo-resident kernel for the output-parity classes of a stride-2 layer in ONE pass over the input: the forward of
// ConvTranspose2d(k3, s2, p1, op1) (resnet2d.py:52-57: u128, u64 — classes of 1 / 2 / 2 / 4 taps) and the data gradient of a
// stride-2 conv (resnet2d.py:35 backward: same classes; patchgan2d.py:36-48 backward, k4: 4 / 4 / 4 / 4 taps).
//
// gs_gconv_forward_multi runs these as im2col tiles: every class gathers every input pixel once per tap from L2 — 442 MB
// of L2 -> LDS traffic for the 100 MB u64 layer at ~4.7 TB/s, whatever the tile size. All classes read
// the SAME input through offsets inside a 3 x 3 window, so here a workgroup owns a 16 x 16 box of class-grid pixels,
// stages its 18 x 18 halo box ONCE per 64-channel chunk (double-buffered LDS-DMA, like hconvw.hip) and runs every
// (class, tap) K-step out of it: 51 KB of halo + 8 KB of weights per K-step instead of 16 + 8 KB per K-step per 128
// pixels. A wave holds 64 pixels x 16 output channels of ALL four classes (64 accumulator registers); weights stream
// through a 3-slot ring of PAIRS of K-steps (one 1-KiB DMA instruction per wave per pair). Fragment reads roll two
// half-K-step sets ahead of the MFMAs with counted waits (common.hpp). Epilogue: bias, statistics (one slot per box over
// all classes, the rest of the layer's slots zeroed), activation, class by class through an LDS slab into 128-B stores at
// out[2i + py][2j + px].
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

namespace {
__device__ __forceinline__ void lds_write64(unsigned addr, uint2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
constexpr int NC = 4;
struct HConvTK {
  const char* in;
  const char* w;               // class 0's pack; w_off[c] = byte offset of class c's [w_rows][Kp] block
  const float* bias;
  char* out;
  float* stats;
  const char* zero;
  long long w_off[NC];
  int kp[NC], py[NC], px[NC];
  int tb[16];                  // halo byte offset of K-step s of a chunk (tap offset relative to the window's corner)
  int wtap[16];                // tap index inside its class of K-step s (k offset = wtap * Ci + chunk * 64)
  int tiles_m, tiles_n, nbw, hmin, wmin, chunks;
  int ntiles;                  // boxes x channel tiles x images of the launch; workgroup b walks tiles b, b + gridDim.x, ...
  int nsplit;                  // twin batch (gs_twin): images [nsplit, N) take the packs / bias w_delta / bias_delta bytes
  long long w_delta, bias_delta;   // further on (a box never straddles images)
  gs_gconv_desc d;             // class 0's descriptor (shared fields)
  gs_gconv_fuse f;             // fused != 0: the reduction pass of the consumer's InstanceNorm backward rides in the epilogue
  int fused;                   // (gs_gconv_forward_multi_fused; contract of gs_gconv_forward_fused, one slot per box)
};

// PAT 0: taps per class 1 / 2 / 2 / 4 (k3 transposed conv / stride-2 k3 gradient), PAT 1: 4 / 4 / 4 / 4 (k4)
template <int PAT> struct Pat;
template <> struct Pat<0> { static constexpr int NS = 9; static constexpr int CPS = 2; };
template <> struct Pat<1> { static constexpr int NS = 16; static constexpr int CPS = 1; };
template <int PAT>
__device__ __host__ constexpr int cls_of(int s) {
  if (PAT == 0) return s < 1 ? 0 : (s < 3 ? 1 : (s < 5 ? 2 : 3));
  return s >> 2;
}

// PLAIN: a data-gradient launch — no bias, no statistics, no activation (their 24 registers are what the kernel spills)
template <int PAT, bool FUSED = false, bool PLAIN = false>
__global__ __launch_bounds__(512) void hconvt_kernel(const HConvTK p) {
  // 8 waves as 4 (pixel rows) x 2 (channels): 64 pixels x 32 channels x 4 classes = 128 accumulator registers per lane —
  // the 16-wave split (64 x 16 per wave) fits the 128-VGPR cap only with spills, whose reloads wait vmcnt(0) right behind
  // the weight DMA (measured 80 us for the u64 layer against 93 on the im2col path)
  constexpr int NW = 8, BN = 64, TI = 2, WPI = 2;
  constexpr int NS = Pat<PAT>::NS, CPS = Pat<PAT>::CPS;      // K-steps per chunk, chunks per unrolled super-chunk
  constexpr int NPAIR = NS * CPS / 2;                        // pairs of K-steps per super-chunk
  constexpr int WT = 2 * BN * 128;                           // weight stage: a PAIR of K-steps, 64 rows x 64 k each
  constexpr int HP = 160, HPIECES = 18 * 18 * 10, HINSTR = (HPIECES + 63) / 64, HBUF = HINSTR * 1024;
  constexpr int HPW = (HINSTR + NW - 1) / NW;
  constexpr int TJ = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wring = smem;                                        // 3 x 16 KiB
  char* hbuf = smem + 3 * WT;                                // 2 x 51 KiB
  char* sink = hbuf + 2 * HBUF;                              // 1 KiB
  const gs_gconv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // PERSISTENT (round 5): workgroup b0 walks tiles b0, b0 + G, b0 + 2G, ... (G = gridDim.x, a multiple of tiles_n: one channel
  // tile per workgroup). The K-step stream simply continues across tiles: the weight ring is fed with the NEXT tile's first two
  // pairs while this tile's last two compute, its first halo chunk goes into the buffer this tile's second-to-last chunk left,
  // the epilogue works out of the buffer the LAST chunk leaves (the next tile's second chunk lands there afterwards), and the
  // tile's 16 output stores are never waited for (counted vmcnt at the next tile's first pair). Measured before: one
  // load -> compute -> store chain per workgroup and CU, matrix pipes 16 % busy, 2.2 TB/s (profiles/r04_trunk_pmc_v4.txt).
  const int G = gridDim.x;
  int b0;
  {
    const int q = G >> 3, r = G & 7, xcd = blockIdx.x & 7;
    b0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int ntl = (p.ntiles - b0 + G - 1) / G;
  const int nt = b0 % p.tiles_n;
  struct Tile { int n, mt, oy0, ox0; const char* in_n; const char* wnet; };
  auto tile_of = [&](int it) {
    const int bb = (b0 + it * G) / p.tiles_n;
    Tile t;
    t.mt = bb % p.tiles_m;
    t.n = bb / p.tiles_m;
    t.oy0 = (t.mt / p.nbw) * 16;
    t.ox0 = (t.mt % p.nbw) * 16;
    t.in_n = p.in + ((size_t)t.n * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
    t.wnet = p.w + (t.n >= p.nsplit ? p.w_delta : 0);
    return t;
  };

  // ---- halo pieces of this thread: source byte offset of channel chunk 0, or -1 (zero border) ----
  // (recomputed at every issue — a few dozen VALU instructions two times per super-chunk — instead of held in registers:
  //  the 64 accumulator + 40 fragment registers leave no room for them under the 128-VGPR cap of a 16-wave workgroup)
  auto issue_halo = [&](const Tile& tl, int chunk, int buf) {
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
      int q = (i * NW + wave) * 64 + lane;
      asm volatile("" : "+v"(q));
      const int v = q / 10, part = q - v * 10;
      const int hy = v / 18, hx = v - hy * 18;
      const int iy = tl.oy0 + hy + p.hmin, ix = tl.ox0 + hx + p.wmin;
      const bool ok = q < HPIECES && part < 8 && (unsigned)iy < (unsigned)d.Hi && (unsigned)ix < (unsigned)d.Wi;
      const unsigned off = (unsigned)(((iy * d.Wi + ix) * d.in_cs + part * 8) * 2) + (unsigned)chunk * 128u;
      const char* src = ok ? tl.in_n + off : p.zero;
      const int inst = i * NW + wave;
      glds16(src, inst < HINSTR ? hbuf + buf * HBUF + inst * 1024 : sink);
    }
  };
  // ---- weight stage of a pair (s0, s0 + 1) of K-steps: waves 0-7 bring step s0's 64 rows, waves 8-15 step s0 + 1's ----
  const int lrow = lane >> 3;
  const int wpiece = (lane & 7) ^ lrow;
  const int wrow0 = nt * BN + (wave & 3) * 16 + lrow;        // output channel this lane fetches (+ 8 for its second instruction)
  auto issue_w = [&](const Tile& tl, int chunk, int s, int stage) {   // s: this wave's K-step inside its chunk (compile-time after unrolling)
    const int c = cls_of<PAT>(s);
    int wr = wrow0;
    asm volatile("" : "+v"(wr));                             // keep the 18 per-step addresses out of loop-invariant registers
    const char* base = tl.wnet + (p.w_off[c] + ((long long)p.wtap[s] * d.Ci + chunk * 64) * 2);
#pragma unroll
    for (int i = 0; i < WPI; ++i)
      glds16(base + (unsigned)(((wr + i * 8) * p.kp[c] + wpiece * 8) * 2), wring + stage * WT + (wave * WPI + i) * 1024);
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fk = lane >> 4, swz = lane & 7;
  const unsigned rowb0 = (unsigned)(((wm * 4) * 18 + frow) * HP + fk * 16);   // box row wm*4 + j: + j * 18 * HP as the read's offset

  f32x4 acc[NC][TI][TJ];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned smem0 = lds_addr(smem);
  const unsigned wfrag0 = smem0 + (unsigned)((wn * 32 + frow) * 128);
  const unsigned kc0 = (unsigned)(((0 * 4 + fk) ^ swz) << 4);        // k-piece of half 1: (4 + fk) ^ swz = (fk ^ swz) ^ 4 -> kc0 ^ 64
  const unsigned hbuf0 = smem0 + 3 * WT;
  struct Frags { bf16x8 w[TI], x[TJ]; };
  auto load = [&](Frags& f, unsigned wbase, unsigned xbase, auto kk_tag) {
    constexpr int kk = decltype(kk_tag)::value;
    const unsigned wa = wbase + (kk ? (kc0 ^ 64u) : kc0);
    lds_read128<0>(f.w[0], wa);
    lds_read128<2048>(f.w[1], wa);
    const unsigned xa = xbase + rowb0;
    lds_read128<kk * 64 + 0 * 18 * HP>(f.x[0], xa);
    lds_read128<kk * 64 + 1 * 18 * HP>(f.x[1], xa);
    lds_read128<kk * 64 + 2 * 18 * HP>(f.x[2], xa);
    lds_read128<kk * 64 + 3 * 18 * HP>(f.x[3], xa);
  };
  auto mma = [&](const Frags& f, f32x4 (&a)[TI][TJ]) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) a[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[i], f.x[j], a[i][j], 0, 0, 0);
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;

  // the bias of this channel tile for both networks of a twin batch, into LDS before the first LDS-DMA (an ordinary load used
  // while a DMA is in flight is waited for with vmcnt(0): inside the tile loop that would drain the next tile's operands)
  float* const bias_lds = reinterpret_cast<float*>(sink + 1024);          // [2 networks][BN]
  if (!PLAIN && !FUSED && tid < 2 * BN) {
    const int net = tid / BN, c = nt * BN + tid % BN;
    const float* bsrc = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.bias) + (net ? p.bias_delta : 0));
    bias_lds[tid] = p.bias ? bsrc[c] : 0.f;
  }

  // ---- prologue: halo of chunks 0 (and 1), weight pairs 0, 1, 2 of the first tile -------------------------------------------
  const int NP = (p.chunks * NS) >> 1;                        // pairs of K-steps per tile (launcher: a whole number of super-chunks)
  Tile cur = tile_of(0);
  auto issue_pair = [&](const Tile& tl, int pr) {             // runtime pair index: only used in the prologue
    const int ks = pr * 2 + (wave >> 2);
    const int c = ks / NS, s = ks - c * NS;
    // (runtime s: the class tables are indexed dynamically here, three times per launch)
    const int cl = PAT == 0 ? (s < 1 ? 0 : (s < 3 ? 1 : (s < 5 ? 2 : 3))) : (s >> 2);
    const char* base = tl.wnet + (p.w_off[cl] + ((long long)p.wtap[s] * d.Ci + c * 64) * 2);
#pragma unroll
    for (int i = 0; i < WPI; ++i)
      glds16(base + (unsigned)(((wrow0 + i * 8) * p.kp[cl] + wpiece * 8) * 2), wring + (pr % 3) * WT + (wave * WPI + i) * 1024);
  };
  __syncthreads();                                            // bias_lds (nothing in flight yet)
  issue_halo(cur, 0, 0);
  if (p.chunks > 1) issue_halo(cur, 1, 1);
#pragma unroll
  for (int pr = 0; pr < 3; ++pr)
    if (pr < NP) issue_pair(cur, pr);
  if (NP >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WPI) : "memory");
  else if (NP == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPI) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  Frags S0, S1;
  int stage = 0;
  int hb0 = 0;                                                // halo buffer of the current tile's chunk 0
  constexpr int NSTORE = 16;                                  // output store instructions per lane per tile (4 classes x 4)
#pragma clang loop unroll(disable)
  for (int it = 0; it < ntl; ++it) {
    const bool has_next = it + 1 < ntl;
    Tile nxt = cur;
    if (has_next) nxt = tile_of(it + 1);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma clang loop unroll(disable)
    for (int sc = 0; sc < p.chunks / CPS; ++sc) {
      const int cbase = sc * CPS;
      static_for<0, NPAIR>([&](auto pr_tag) {
        constexpr int pr = decltype(pr_tag)::value;
        constexpr int sA = (2 * pr) % NS, sB = (2 * pr + 1) % NS;          // K-steps inside their chunks
        constexpr int cA = (2 * pr) / NS, cB = (2 * pr + 1) / NS;          // chunk inside the super-chunk
        const int q = sc * NPAIR + pr;                                     // pair index inside the tile
        // weights of the pair after next into the slot the previous pair used (the first tile's pairs 0 - 2 come from the
        // prologue, a later tile's pairs 0 / 1 from the previous tile's last two pairs); the halo box of a later chunk into the
        // buffer of the chunk that just finished (issued AFTER the weights: it may stay in flight for one more pair)
        bool halo_now = false;
        {
          constexpr int p2 = pr + 2;
          constexpr int sW0 = (2 * (p2 % NPAIR)) % NS, sW1 = (2 * (p2 % NPAIR) + 1) % NS;
          constexpr int cW0 = (2 * (p2 % NPAIR)) / NS, cW1 = (2 * (p2 % NPAIR) + 1) / NS;
          const int st2 = stage == 0 ? 2 : stage - 1;
          if (q + 2 < NP) {
            if (it > 0 || q >= 1) {
              const int cw = cbase + (p2 / NPAIR) * CPS;
              if (wave < 4) issue_w(cur, cw + cW0, sW0, st2); else issue_w(cur, cw + cW1, sW1, st2);
            }
          } else if (has_next) {                              // (q + 2 - NP is pair p2 % NPAIR of the next tile's first super-chunk)
            if (wave < 4) issue_w(nxt, cW0, sW0, st2); else issue_w(nxt, cW1, sW1, st2);
          }
        }
        if constexpr (PAT == 0) {
          // chunk cbase + 1: its box goes into the buffer chunk cbase - 1 left at pair 0 (the first tile's chunk 1 comes from the
          // prologue; a later tile's lands where the previous tile's epilogue worked); the box of chunk cbase + 2 — or of the next
          // tile's chunk 0 — into the buffer chunk cbase leaves at pair 4 (its last K-step), from pair 5 on
          if (pr == 0 && (sc >= 1 || it > 0) && cbase + 1 < p.chunks) { issue_halo(cur, cbase + 1, (hb0 + cbase + 1) & 1); halo_now = true; }
          if (pr == 5) {
            if (cbase + 2 < p.chunks) { issue_halo(cur, cbase + 2, (hb0 + cbase) & 1); halo_now = true; }
            else if (has_next) { issue_halo(nxt, 0, (hb0 + cbase) & 1); halo_now = true; }
          }
        } else {
          if (pr == 0) {
            if (cbase + 1 < p.chunks) {
              if (sc >= 1 || it > 0) { issue_halo(cur, cbase + 1, (hb0 + cbase + 1) & 1); halo_now = true; }
            } else if (has_next) {
              // last chunk of the tile: the next tile's chunk 0 into the other buffer (a one-chunk tile's other buffer is
              // where the previous tile's epilogue worked: finished, the barrier of the previous pair is behind us)
              issue_halo(nxt, 0, (hb0 + cbase + 1) & 1); halo_now = true;
            }
          }
        }
        const unsigned wst = wfrag0 + (unsigned)(stage * WT);
        unsigned hA = hbuf0 + (unsigned)(((hb0 + cbase + cA) & 1) * HBUF) + (unsigned)p.tb[sA];
        unsigned hB = hbuf0 + (unsigned)(((hb0 + cbase + cB) & 1) * HBUF) + (unsigned)p.tb[sB];
        // (with two chunks per super-chunk the buffer parity is loop-invariant and the compiler would keep all 18 x 4 fragment
        //  addresses of the unrolled body in registers: 160 spilled VGPRs)
        asm volatile("" : "+s"(hA), "+s"(hB));
        load(S0, wst, hA, K0{});
        load(S1, wst, hA, K1{});
        gs_lgkm_wait<6>(S0.w[0], S0.w[1], S0.x[0], S0.x[1], S0.x[2], S0.x[3]);
        mma(S0, acc[cls_of<PAT>(sA)]);
        load(S0, wst + BN * 128, hB, K0{});
        gs_lgkm_wait<6>(S1.w[0], S1.w[1], S1.x[0], S1.x[1], S1.x[2], S1.x[3]);
        mma(S1, acc[cls_of<PAT>(sA)]);
        load(S1, wst + BN * 128, hB, K1{});
        gs_lgkm_wait<6>(S0.w[0], S0.w[1], S0.x[0], S0.x[1], S0.x[2], S0.x[3]);
        mma(S0, acc[cls_of<PAT>(sB)]);
        gs_lgkm_wait<0>(S1.w[0], S1.w[1], S1.x[0], S1.x[1], S1.x[2], S1.x[3]);
        mma(S1, acc[cls_of<PAT>(sB)]);
        // the weights of the next pair (issued a pair ago) and everything older have landed; what may still fly: the weights
        // of the pair after it, a halo box issued in this pair and — at a later tile's first pair — the previous tile's output
        // stores (issued between the two: VMEM retires in issue order)
        if (q + 1 < NP || has_next) {
          if (q + 2 >= NP && !has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if (it > 0 && q == 0) {
            if (halo_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE + HPW + WPI) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE + WPI) : "memory");
          }
          else if (halo_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HPW + WPI) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPI) : "memory");
        }
        __builtin_amdgcn_s_barrier();
        stage = stage == 2 ? 0 : stage + 1;
      });
    }

    // ---- epilogue out of the halo buffer the tile's last chunk left (the other one holds the next tile's chunk 0) -------------
    const int n = cur.n, mt = cur.mt, oy0 = cur.oy0, ox0 = cur.ox0;
    char* const ebuf = hbuf + ((hb0 + p.chunks - 1) & 1) * HBUF;
    f32x4 bia[TI];                                             // this lane's 2 x 4 output channels
#pragma unroll
    for (int i = 0; i < TI; ++i)
      bia[i] = (!PLAIN && !FUSED) ? *reinterpret_cast<const f32x4*>(bias_lds + (n >= p.nsplit ? BN : 0) + wn * 32 + i * 16 + fk * 4)
                                  : f32x4{0.f, 0.f, 0.f, 0.f};
    const bool want_stats = !FUSED && !PLAIN && d.stats_slots > 0;
    float s1[TI][4], s2[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
    constexpr int SROW = BN * 2 + 16;                          // slab row: 64 channels + a pad piece
    char* slab = ebuf;                                         // [256 pixels][SROW]
    float* red = reinterpret_cast<float*>(ebuf + 256 * SROW);  // [4 wm][64][2]  (fused: [8 waves][64][3])
    static_assert(256 * SROW + NW * BN * 3 * 4 <= HBUF, "epilogue scratch must fit one halo buffer");
    // fused norm-backward sums of this thread's 8 channels (the store loop's `piece`) over the pixels it stores:
    // ghat = (g + g2) * act'(yhat), ghat * yhat, yhat with yhat = (y - mean) * rstd of the CONSUMER's forward output y
    float fa1[8], fa2[8], fa3[8], fmu[8], frs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) fa1[k] = fa2[k] = fa3[k] = fmu[k] = frs[k] = 0.f;
    if constexpr (FUSED) {
      const float* mr = p.f.mean_rstd + (size_t)n * 2 * d.Co + nt * BN + (tid & 7) * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) { fmu[k] = mr[k]; frs[k] = mr[d.Co + k]; }
    }
    // (no wait: the last pair's barrier is behind every wave, and nobody reads this buffer any more)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      // (the activation is chosen once per class, not per element: the switch inside these loops was 3800 scalar instructions)
      auto to_slab = [&](auto none_tag) {
        constexpr bool ACT_NONE = decltype(none_tag)::value;
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int i = 0; i < TI; ++i) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if constexpr (PLAIN || FUSED) {
                v[r] = acc[c][i][j][r];
              } else {
                v[r] = acc[c][i][j][r] + bia[i][r];
                s1[i][r] += v[r];
                s2[i][r] += v[r] * v[r];
                if constexpr (!ACT_NONE) v[r] = apply_act_small(v[r], d.act, d.slope);
              }
            }
            uint2 o;
            o.x = pack_bf2(v[0], v[1]);
            o.y = pack_bf2(v[2], v[3]);
            // (inline asm: in front of a compiler-visible LDS access the compiler waits vmcnt(0) — the next tile's operands are
            // in flight into the other buffers)
            lds_write64(lds_addr(slab + ((wm * 4 + j) * 16 + frow) * SROW + (wn * 32 + i * 16 + fk * 4) * 2), o);
          }
      };
      if (PLAIN || FUSED || d.act == GS_ACT_NONE) to_slab(std::true_type{}); else to_slab(std::false_type{});
      lds_barrier();
      {
        const int piece = tid & 7;                             // 8 lanes x 16 B = the 64 channels of a pixel: 128-B stores
        // (FUSED: the consumer's y / g2 of two pixels are requested before the first is used — as load, use, load, use the fused
        // epilogue tripled the launch; all four at once spill the accumulators of the classes still to come)
#pragma unroll 1
        for (int it0 = 0; it0 < 4; it0 += 2) {      // (not unrolled: with all four pixels' loads hoisted the FUSED form spilled 170 registers)
          uint4 val[2], yv[2], gv[2];
          size_t opx[2];
          {
            bf16x8 r0, r1;
            const unsigned ra = lds_addr(slab + (it0 * 64 + (tid >> 3)) * SROW + piece * 16);
            lds_read128<0>(r0, ra);
            lds_read128<64 * SROW>(r1, ra);
            gs_lgkm_wait_only<0>();
            reg_fence(r0); reg_fence(r1);
            val[0] = __builtin_bit_cast(uint4, r0);
            val[1] = __builtin_bit_cast(uint4, r1);
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int pl = (it0 + u) * 64 + (tid >> 3);
            const int ly = pl >> 4, lx = pl & 15;
            opx[u] = ((size_t)n * d.Ho + ((oy0 + ly) * 2 + p.py[c])) * d.Wo + ((ox0 + lx) * 2 + p.px[c]);
            if constexpr (FUSED) {
              const size_t e = (opx[u] * d.Co + nt * BN + piece * 8) * 2;
              yv[u] = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.f.y) + e);
              if (p.f.g2) gv[u] = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.f.g2) + e);
            }
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<uint4*>(p.out + (opx[u] * d.out_cs + d.out_co + nt * BN + piece * 8) * 2) = val[u];
            if constexpr (FUSED) {
              float g[8] = {bf_lo(val[u].x), bf_hi(val[u].x), bf_lo(val[u].y), bf_hi(val[u].y),
                            bf_lo(val[u].z), bf_hi(val[u].z), bf_lo(val[u].w), bf_hi(val[u].w)};
              const float yr[8] = {bf_lo(yv[u].x), bf_hi(yv[u].x), bf_lo(yv[u].y), bf_hi(yv[u].y),
                                   bf_lo(yv[u].z), bf_hi(yv[u].z), bf_lo(yv[u].w), bf_hi(yv[u].w)};
              if (p.f.g2) {
                g[0] += bf_lo(gv[u].x); g[1] += bf_hi(gv[u].x); g[2] += bf_lo(gv[u].y); g[3] += bf_hi(gv[u].y);
                g[4] += bf_lo(gv[u].z); g[5] += bf_hi(gv[u].z); g[6] += bf_lo(gv[u].w); g[7] += bf_hi(gv[u].w);
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
          }
        }
      }
      lds_barrier();
    }
    if constexpr (FUSED) {
      // lanes with equal (lane & 7) hold different pixels of the same 8 channels: sum inside the wave, then over the 8 waves
      float* red3 = red;                                       // [8 waves][64 channels][3]
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        fa1[k] = row_sum_stride8(fa1[k]); fa2[k] = row_sum_stride8(fa2[k]); fa3[k] = row_sum_stride8(fa3[k]);
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
          fa1[k] += __shfl_xor(fa1[k], o, 64);
          fa2[k] += __shfl_xor(fa2[k], o, 64);
          fa3[k] += __shfl_xor(fa3[k], o, 64);
        }
      }
      if (lane < 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          red3[((wave * BN) + lane * 8 + k) * 3 + 0] = fa1[k];
          red3[((wave * BN) + lane * 8 + k) * 3 + 1] = fa2[k];
          red3[((wave * BN) + lane * 8 + k) * 3 + 2] = fa3[k];
        }
      }
      lds_barrier();
      if (tid < BN) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { t0 += red3[(w * BN + tid) * 3]; t1 += red3[(w * BN + tid) * 3 + 1]; t2 += red3[(w * BN + tid) * 3 + 2]; }
        float* sp = p.f.partial + ((size_t)n * p.tiles_m + mt) * 3 * d.Co;
        const int cc = nt * BN + tid;
        sp[cc] = t0; sp[d.Co + cc] = t1; sp[2 * d.Co + cc] = t2;
      }
    } else if (want_stats) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row16_sum(s1[i][r]), q = row16_sum(s2[i][r]);
          if (frow == 0) {
            red[(wm * BN + wn * 32 + i * 16 + fk * 4 + r) * 2 + 0] = a;
            red[(wm * BN + wn * 32 + i * 16 + fk * 4 + r) * 2 + 1] = q;
          }
        }
      lds_barrier();
      int tq = tid;
      asm volatile("" : "+v"(tq));                            // (addresses below are built here, not carried through the tile)
      if (tq < BN) {
        const int cc = nt * BN + tq;
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += red[(w * BN + tq) * 2]; q += red[(w * BN + tq) * 2 + 1]; }
        // slot `mt` holds this box (all classes); the layer's other slots — counted per class by the caller — are zeroed, each
        // by the box whose index it is congruent to
        for (int slot = mt; slot < d.stats_slots; slot += p.tiles_m) {
          float* sp = p.stats + (((size_t)n * d.stats_slots + slot) * 2) * d.Co;
          sp[cc] = slot == mt ? a : 0.f;
          sp[d.Co + cc] = slot == mt ? q : 0.f;
        }
      }
    }
    // (wave 0's statistics / sums stores are younger than its 16 output stores: its counted wait at the next tile's first pair
    // then also covers some of the output stores — an over-wait, never an under-wait)
    if (has_next) lds_barrier();                               // the next tile's second chunk lands in this buffer
    cur = nxt;
    hb0 = (hb0 + p.chunks) & 1;
  }
}

int hconvt_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  return cus;
}

bool window(const gs_gconv_desc* const* descs, int* lo) {
  int hi[2] = {-128, -128};
  lo[0] = lo[1] = 127;
  for (int c = 0; c < NC; ++c)
    for (int t = 0; t < descs[c]->T; ++t) {
      if (descs[c]->dd[t] != 0) return false;
      const int o[2] = {descs[c]->dh[t], descs[c]->dw[t]};
      for (int a = 0; a < 2; ++a) { if (o[a] < lo[a]) lo[a] = o[a]; if (o[a] > hi[a]) hi[a] = o[a]; }
    }
  return hi[0] - lo[0] <= 2 && hi[1] - lo[1] <= 2;
}
}  // namespace

// pattern of the classes (0: 1/2/2/4 taps, 1: 4/4/4/4) when the layer runs here, -1 when it does not
int gs_hconvt_pattern(const gs_gconv_desc* const* descs, int count) {
  if (gs_opt(GS_OPT_HCONVT) == 0 || count != NC) return -1;
  const gs_gconv_desc* d = descs[0];
  if (d->so != 2 || d->si != 1 || d->Ci % 64 != 0 || d->Co % 64 != 0 || d->border != GS_BORDER_ZERO) return -1;
  if (d->Di != 1 || d->Do != 1 || d->Dc != 1 || d->Hc % 16 != 0 || d->Wc % 16 != 0) return -1;
  if (d->Ho != 2 * d->Hc || d->Wo != 2 * d->Wc || d->accumulate) return -1;
  int pat = -1;
  if (descs[0]->T == 1 && descs[1]->T == 2 && descs[2]->T == 2 && descs[3]->T == 4) pat = 0;
  if (descs[0]->T == 4 && descs[1]->T == 4 && descs[2]->T == 4 && descs[3]->T == 4) pat = 1;
  if (pat < 0) return -1;
  for (int c = 0; c < NC; ++c) {
    const gs_gconv_desc* e = descs[c];
    if (e->pz != 0 || e->py < 0 || e->py > 1 || e->px < 0 || e->px > 1) return -1;
    // the kernel takes geometry, border, strides and the epilogue contract from descs[0]: every class must agree with it
    if (e->N != d->N || e->Hi != d->Hi || e->Wi != d->Wi || e->Di != d->Di || e->Ci != d->Ci || e->Co != d->Co ||
        e->Ho != d->Ho || e->Wo != d->Wo || e->Do != d->Do || e->Hc != d->Hc || e->Wc != d->Wc || e->Dc != d->Dc ||
        e->in_cs != d->in_cs || e->in_co != d->in_co || e->out_cs != d->out_cs || e->out_co != d->out_co ||
        e->so != d->so || e->si != d->si || e->border != d->border || e->act != d->act || e->slope != d->slope ||
        e->stats_slots != d->stats_slots || e->accumulate != d->accumulate)
      return -1;
  }
  int lo[2];
  if (!window(descs, lo)) return -1;
  const int chunks = d->Ci / 64;
  if (pat == 0 && chunks % 2 != 0) return -1;                 // K-steps are consumed in pairs over super-chunks of 2 chunks
  const long long blocks = (long long)d->N * (d->Hc / 16) * (d->Wc / 16) * (d->Co / 64);
  if (blocks < gs_opt(GS_OPT_HCONVT) || blocks >= (1LL << 31)) return -1;      // option value = smallest grid taken
  if ((long long)d->Hi * d->Wi * d->in_cs * 2 >= (1LL << 31)) return -1;
  if (d->stats_slots > 0 && d->stats_slots < (d->Hc / 16) * (d->Wc / 16)) return -1;
  return pat;
}

int gs_hconvt_launch(const gs_gconv_desc* const* descs, int pat, const void* in, const void* const* w_packs,
                     const float* bias, void* out, float* stats, const gs_gconv_fuse* fuse, void* stream, const gs_twin* tw) {
  const gs_gconv_desc* d = descs[0];
  HConvTK k;   // a plain local: autograd issues launches from its own host threads
  k.nsplit = tw ? tw->n_split : 0x7fffffff;
  k.w_delta = tw ? tw->w_delta : 0;
  k.bias_delta = tw ? tw->bias_delta : 0;
  k.in = static_cast<const char*>(in);
  k.w = static_cast<const char*>(w_packs[0]);
  k.bias = bias;
  k.out = static_cast<char*>(out);
  k.stats = stats;
  k.zero = static_cast<const char*>(gs_zero_page());
  GS_REQUIRE(k.zero, "gs_gconv_forward_multi: library not initialised (call gs_init)");
  int lo[2];
  window(descs, lo);
  int s = 0;
  for (int c = 0; c < NC; ++c) {
    const gs_gconv_desc* dc = descs[c];
    k.w_off[c] = static_cast<const char*>(w_packs[c]) - k.w;
    k.kp[c] = dc->Kp; k.py[c] = dc->py; k.px[c] = dc->px;
    for (int t = 0; t < dc->T; ++t, ++s) {
      k.tb[s] = ((dc->dh[t] - lo[0]) * 18 + (dc->dw[t] - lo[1])) * 160;
      k.wtap[s] = t;
    }
  }
  k.tiles_m = (d->Hc / 16) * (d->Wc / 16);
  k.tiles_n = d->Co / 64;
  k.nbw = d->Wc / 16;
  k.hmin = lo[0]; k.wmin = lo[1];
  k.chunks = d->Ci / 64;
  k.d = *d;
  k.fused = fuse != nullptr;
  k.f = fuse ? *fuse : gs_gconv_fuse{};
  const long long tiles = (long long)d->N * k.tiles_m * k.tiles_n;
  k.ntiles = (int)tiles;
  // one workgroup per CU is resident (152 KB of LDS): more tiles than CUs run as persistent workgroups over every G-th tile,
  // G a multiple of the channel tiles (option hconvt_persist = 0: one tile per workgroup)
  long long blocks = tiles;
  if (gs_opt(GS_OPT_HCONVT_PERSIST) && tiles > hconvt_cus() && k.tiles_n <= hconvt_cus())
    blocks = hconvt_cus() / k.tiles_n * k.tiles_n;
  const int lds = 3 * 2 * 64 * 128 + 2 * ((18 * 18 * 10 + 63) / 64) * 1024 + 1024 + 2 * 64 * 4;
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvt_kernel<0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvt_kernel<1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (fuse) {
    static bool configured_f = false;
    if (!configured_f) {
      GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvt_kernel<0, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvt_kernel<1, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      configured_f = true;
    }
    if (pat == 0) hipLaunchKernelGGL((hconvt_kernel<0, true>), dim3((unsigned)blocks), dim3(512), lds, st, k);
    else hipLaunchKernelGGL((hconvt_kernel<1, true>), dim3((unsigned)blocks), dim3(512), lds, st, k);
  } else if (!bias && d->stats_slots == 0 && d->act == GS_ACT_NONE) {
    static bool configured_p = false;
    if (!configured_p) {
      GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvt_kernel<0, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hconvt_kernel<1, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      configured_p = true;
    }
    if (pat == 0) hipLaunchKernelGGL((hconvt_kernel<0, false, true>), dim3((unsigned)blocks), dim3(512), lds, st, k);
    else hipLaunchKernelGGL((hconvt_kernel<1, false, true>), dim3((unsigned)blocks), dim3(512), lds, st, k);
  } else if (pat == 0) hipLaunchKernelGGL((hconvt_kernel<0>), dim3((unsigned)blocks), dim3(512), lds, st, k);
  else hipLaunchKernelGGL((hconvt_kernel<1>), dim3((unsigned)blocks), dim3(512), lds, st, k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
