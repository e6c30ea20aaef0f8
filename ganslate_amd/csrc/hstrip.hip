// Halo-resident kernel for the W-folded k7 boundary convolutions of the ResNets (stem c7s1-64, output c7s1-3 and their data
// gradients: resnet2d.py:25,65 — after csrc/wfold.hip moved the W taps into channels they are 7 VERTICAL taps over 32 / 64
// channels, 21-64 output channels, at full image resolution).
//
// 1.4 GFLOP over 100 MB of activations: pure streaming. On the im2col kernel every tap gathers the input again from L2
// (7 x 67 MB for the output conv) and the launches sit at 68-86 us against an HBM floor of ~21. With taps in one column
// the input window of a tile of R rows x 8 columns is the (R + 6) x 8 strip above and below it — contiguous pixels, tap t
// = a shift by dh_t x 8 pixels — so a workgroup stages that strip once (LDS-DMA, row border resolved per lane), keeps ALL
// the layer's weights in LDS (<= 30 KB) and runs the taps out of it: input read once, no K loop over global memory.
// 4 waves x (64 pixels x Co), 256-pixel tiles (32 x 8), <= 78 KB of LDS: two workgroups per CU overlap each other's
// load / compute / store phases. Epilogue contract of gconv_kernel: bias, one statistics slot per tile, activation,
// dense or sliced output.
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

namespace {
// the activations other than "none" as a CALL: inlined into the unrolled 64-element epilogues (tanhf is ~40 instructions) they
// made a tile's code longer than the instruction cache holds
__device__ __noinline__ float act_call(float v, int act, float slope) { return apply_act(v, act, slope); }

constexpr int TR = 32, TC = 8;          // tile rows x columns (256 pixels)
struct HStripK {
  const char* in;
  const char* w;
  const float* bias;
  char* out;
  float* stats;
  const char* zero;
  int tiles_y, tiles_x, hmin, span, dwc;   // span = rows of the input window beyond the tile (max dh - min dh)
  int nsplit;                              // twin batch (gs_twin): images [nsplit, N) use the weights / bias w_delta /
  long long w_delta, bias_delta;           // bias_delta bytes further on; gsplit: workgroups of the first network
  int gsplit;                              // (persistent form: a workgroup keeps ONE network's weights in registers)
  int toff[8];                             // tap -> row offset inside the window (dh - hmin)
  gs_gconv_desc d;
};

template <int CI, int CO>
__global__ __launch_bounds__(256, 2) void hstrip_kernel(const HStripK p) {
  // bytes per staged pixel: data + 2 pad pieces. A ds_read_b128 is served in groups of 8 consecutive pixels x 2 k-pieces
  // (hconvw.hip): the pitch in banks must be 8 mod 16 for the 16 lanes to land on 16 distinct 4-bank slots — 160 B (40
  // banks) for 64 channels, 96 B (24) for 32. (80 B / one pad piece measured 48 % of the LDS-active cycles in conflicts.)
  constexpr int PITCH = CI == 64 ? 160 : 96;
  constexpr int PIECES = PITCH / 16;             // 10 / 6
  constexpr int TI = CO / 16, TJ = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const gs_gconv_desc& d = p.d;
  const int T = d.T;
  const int WROW = T * CI * 2 + 32;              // weight row pitch: T * CI / 2 banks = 0 mod 16, + 8 banks of pad (same rule)
  const int wbytes = ((CO * WROW + 1023) / 1024) * 1024;
  char* wl = smem;                               // [CO][WROW]
  char* halo = smem + wbytes;                    // [(TR + span) * TC pixels][PITCH]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y;
  const int n = b / p.tiles_y;
  const int oy0 = ty * TR, ox0 = tx * TC;

  // ---- input strip: (TR + span) rows x TC columns, all CI channels, by LDS-DMA (border resolved in the source address) ----
  const char* in_n = p.in + ((size_t)n * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
  const int hpx = (TR + p.span) * TC;
  const int hinstr = (hpx * PIECES + 63) / 64;
  for (int inst = wave; inst < hinstr; inst += 4) {
    const int q = inst * 64 + lane;
    const int v = q / PIECES, part = q - v * PIECES;
    const int hy = v / TC, hx = v - hy * TC;
    bool ok = v < hpx && part < CI / 8;
    int iy = border_index(oy0 + hy + p.hmin, d.Hi, d.border, ok);
    int ix = border_index(ox0 + hx + p.dwc, d.Wi, d.border, ok);
    iy = min(max(iy, 0), d.Hi - 1);
    ix = min(max(ix, 0), d.Wi - 1);
    const char* src = ok ? in_n + ((size_t)(iy * d.Wi + ix) * d.in_cs + part * 8) * 2 : p.zero;
    glds16(src, halo + inst * 1024);
  }
  // ---- weights: CO rows x (T * CI) k plus one pad piece per row (row pitch WROW), by LDS-DMA as well: lane q -> (row,
  // piece); L2-resident after the first workgroups. (A register copy loop here — 14 dependent load / store rounds per
  // thread — was the longest phase of the workgroup.) ----
  {
    const int rp1 = T * CI / 8 + 2;              // 16-B pieces per LDS row incl. the two pad pieces
    const int winstr = (CO * rp1 + 63) / 64;
    for (int inst = wave; inst < winstr; inst += 4) {
      const int q = inst * 64 + lane;
      const int r = q / rp1, piece = q - r * rp1;
      const bool ok = r < CO && r < d.w_rows && piece < rp1 - 2;
      const char* src = ok ? p.w + (n >= p.nsplit ? p.w_delta : 0) + ((size_t)r * d.Kp + piece * 8) * 2 : p.zero;
      glds16(src, wl + inst * 1024);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- taps out of LDS ------------------------------------------------------------------------------------------------------
  const int frow = lane & 15, fk = lane >> 4;
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned smem0 = lds_addr(smem);
  const unsigned wa0 = smem0 + (unsigned)(frow * WROW + fk * 16);
  const unsigned xa0 = smem0 + (unsigned)wbytes + (unsigned)((wave * 64 + frow) * PITCH + fk * 16);
  for (int t = 0; t < T; ++t) {
    const unsigned xt = xa0 + (unsigned)(p.toff[t] * TC * PITCH);
    const unsigned wt = wa0 + (unsigned)(t * CI * 2);
#pragma unroll
    for (int kk = 0; kk < CI / 32; ++kk) {
      bf16x8 wf[TI], xf[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(smem + (wt - smem0) + i * 16 * WROW + kk * 64);
#pragma unroll
      for (int j = 0; j < TJ; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(smem + (xt - smem0) + j * 16 * PITCH + kk * 64);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();                               // the strip is free: its LDS becomes the output slab

  // ---- epilogue: bias, statistics (slot = tile), activation, coalesced stores through an LDS slab -------------------------
  constexpr int SROW = CO * 2 + 16;
  char* slab = halo;                             // [256 pixels][SROW]  (CO * 2 + 16 <= PITCH * ... checked by the launcher)
  float* red = reinterpret_cast<float*>(smem);   // [4 waves][CO][2] over the weight area
  const bool want_stats = d.stats_slots > 0;
  float s1[TI][4], s2[TI][4];
  auto epilogue = [&](auto plain_tag) {            // (activation chosen once, not per element: see hstripr_kernel)
    constexpr bool PLAIN = decltype(plain_tag)::value;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int co = i * 16 + fk * 4;
    const float* bias_n = p.bias ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.bias) +
                                                                  (n >= p.nsplit ? p.bias_delta : 0)) : nullptr;
    const f32x4 bia = (bias_n && co < d.Co) ? *reinterpret_cast<const f32x4*>(bias_n + co) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int pl = wave * 64 + j * 16 + frow;
      const bool inside = oy0 + pl / TC < d.Hc && ox0 + pl % TC < d.Wc;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] + bia[r];
        if (inside) { s1[i][r] += v[r]; s2[i][r] += v[r] * v[r]; }
        if constexpr (!PLAIN) v[r] = act_call(v[r], d.act, d.slope);
      }
      uint2 o;
      o.x = pack_bf2(v[0], v[1]);
      o.y = pack_bf2(v[2], v[3]);
      *reinterpret_cast<uint2*>(slab + pl * SROW + co * 2) = o;
    }
  }
  };
  if (d.act == GS_ACT_NONE) epilogue(std::true_type{}); else epilogue(std::false_type{});
  __syncthreads();
  {
    constexpr int LPP = CO / 8;                  // lanes (16 B each) per pixel
    for (int q = tid; q < 256 * LPP; q += 256) {
      const int pl = q / LPP, piece = q - pl * LPP;
      const int oy = oy0 + pl / TC, ox = ox0 + pl % TC;
      if (oy < d.Hc && ox < d.Wc && piece * 8 < d.Co) {
        const size_t opix = ((size_t)n * d.Ho + oy) * d.Wo + ox;
        *reinterpret_cast<uint4*>(p.out + (opix * d.out_cs + d.out_co + piece * 8) * 2) =
            *reinterpret_cast<const uint4*>(slab + pl * SROW + piece * 16);
      }
    }
  }
  if (want_stats) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s1[i][r]), q = row16_sum(s2[i][r]);
        if (frow == 0) {
          red[(wave * CO + i * 16 + fk * 4 + r) * 2 + 0] = a;
          red[(wave * CO + i * 16 + fk * 4 + r) * 2 + 1] = q;
        }
      }
    __syncthreads();
    if (tid < CO && tid < d.Co) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { a += red[(w * CO + tid) * 2]; q += red[(w * CO + tid) * 2 + 1]; }
      float* sp = p.stats + (((size_t)n * d.stats_slots + d.stats_slot0 + ty * p.tiles_x + tx) * 2) * d.Co;
      sp[tid] = a;
      sp[d.Co + tid] = q;
    }
  }
}


// ---- round 4: persistent form with the weights in registers -----------------------------------------------------------------
// Ablations of hstrip_kernel (profiles/r04_hstrip_ablations.txt: no stores -5 us, no weight DMA -4, no strip DMA -8, one tap
// instead of seven -13, of 43-55 us; HBM floor 21) say no single resource bounds it: a workgroup's life is one serial chain —
// kernel arguments, piece decode, DMA issue, HBM latency, barrier, taps, barrier, slab, stores — and with 30 KB of weights next
// to the strip only two such chains fit a CU. Here a workgroup keeps the layer's weights in REGISTERS (7 taps x CI/32 x CO/16
// fragments = 112 VGPRs at two waves per SIMD), loaded once, and walks tiles: LDS holds nothing but two strip buffers, the
// next tile's strip is on its way while this one's taps, slab and stores run, the piece decode is done once per workgroup
// and the border rules once per launch (two small tables, as in hwgrad.hip). Still two workgroups per CU.
// TRr = tile rows (32 for 32 input channels; 16 for 64, whose 38-row strips would leave one workgroup per CU).
template <int CI, int CO, int TRr>
__global__ __launch_bounds__(256, 2) void hstripr_kernel(const HStripK p) {
  constexpr int T = 7;
  constexpr int PITCH = CI == 64 ? 160 : 96;
  constexpr int PIECES = PITCH / 16;
  constexpr int TI = CO / 16, TJ = TRr * TC / 64;      // accumulator tiles per wave: CO x (64 or 32 pixels)
  constexpr int KK = CI / 32;
  constexpr int TPX = TRr * TC;                        // pixels per tile
  constexpr int SROW = CO * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const gs_gconv_desc& d = p.d;
  const int hrows = TRr + p.span;
  const int hpx = hrows * TC;
  const int hinstr = (hpx * PIECES + 63) / 64;
  const int bbytes = max(hinstr * 1024, TPX * SROW);   // a buffer is a strip, then the tile's output slab
  char* buf0 = smem;
  float* red = reinterpret_cast<float*>(smem + 2 * bbytes);          // [4 waves][CO][2]
  unsigned short* ytab = reinterpret_cast<unsigned short*>(red + 4 * CO * 2);
  unsigned short* xtab = ytab + p.tiles_y * hrows;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;
  for (int e = tid; e < p.tiles_y * hrows + p.tiles_x * TC; e += 256) {
    bool ok = true;
    int v;
    if (e < p.tiles_y * hrows) {
      const int ty = e / hrows, hy = e - ty * hrows;
      v = border_index(ty * TRr + hy + p.hmin, d.Hi, d.border, ok);
      v = min(max(v, 0), d.Hi - 1);
    } else {
      const int e2 = e - p.tiles_y * hrows;
      const int tx = e2 / TC, hx = e2 - tx * TC;
      v = border_index(tx * TC + hx + p.dwc, d.Wi, d.border, ok);
      v = min(max(v, 0), d.Wi - 1);
    }
    ytab[e] = ok ? (unsigned short)v : (unsigned short)0x8000;
  }
  // twin batch: workgroups [0, gsplit) serve the first network's images, the others the second's
  const int net = (int)blockIdx.x >= p.gsplit ? 1 : 0;
  const int gx = (int)blockIdx.x - net * p.gsplit, gnum = net ? (int)gridDim.x - p.gsplit : p.gsplit;
  const char* wnet = p.w + (net ? p.w_delta : 0);
  const float* bias_net = p.bias ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.bias) + (net ? p.bias_delta : 0))
                                 : nullptr;
  // the layer's weights: fragment (t, kk, i) = rows i*16 + frow, k = t*CI + kk*32 + fk*8 .. + 8 of the [rows][Kp] pack
  bf16x8 wf[T][KK][TI];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int r = i * 16 + frow;
        wf[t][kk][i] = r < d.w_rows ? *reinterpret_cast<const bf16x8*>(wnet + ((size_t)r * d.Kp + t * CI + kk * 32 + fk * 8) * 2)
                                    : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
  // the strip pieces this thread stages of every tile
  constexpr int NP = CI == 64 ? 8 : 8;                 // <= 2048 pieces of 16 B (launcher)
  int h_p[NP];                                         // strip row << 8 | column << 4 | channel piece; -1: pad / beyond the strip
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int q = (i * 4 + wave) * 64 + lane;
    const int v = q / PIECES, part = q - v * PIECES;
    const int hy = v / TC;
    h_p[i] = (v < hpx && part < CI / 8) ? (hy << 8 | (v - hy * TC) << 4 | part) : -1;
  }
  const int per_img = p.tiles_y * p.tiles_x;
  const int nimg_net = p.gsplit < (int)gridDim.x ? p.nsplit : d.N;       // images of this workgroup's network
  const int tile0 = net * p.nsplit * per_img;                            // its first tile, its tile count
  const int ntiles = tile0 + nimg_net * per_img;
  auto issue_strip = [&](int tile, char* buf) {
    const int n = tile / per_img, r = tile - n * per_img;
    const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
    const char* in_n = p.in + ((size_t)n * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
    const unsigned short* yrow = ytab + ty * hrows;
    const unsigned short* xrow = xtab + tx * TC;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (i * 4 + wave < hinstr) {                     // wave-uniform
        int hp = max(h_p[i], 0);
        asm volatile("" : "+v"(hp));                   // decode per tile: hoisted, the three fields of 8 pieces spill
        const unsigned iy = yrow[hp >> 8], ix = xrow[(hp >> 4) & 15];
        const bool ok = h_p[i] >= 0 && !((iy | ix) & 0x8000u);
        unsigned off = ((iy * (unsigned)d.Wi + ix) * (unsigned)d.in_cs + (unsigned)(hp & 15) * 8u) * 2u;
        asm volatile("" : "+v"(off));
        glds16(ok ? in_n + off : p.zero, buf + (size_t)(i * 4 + wave) * 1024);
      }
  };
  __syncthreads();                                     // tables
  int cur = 0;
  if (tile0 + gx < ntiles) issue_strip(tile0 + gx, buf0);
  const bool want_stats = d.stats_slots > 0;
  for (int tile = tile0 + gx; tile < ntiles; tile += gnum) {
    char* buf = buf0 + (size_t)cur * bbytes;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's strip (and the previous tile's stores, older)
    __syncthreads();                                   // ... for every wave; the other buffer's slab has been read
    if (tile + gnum < ntiles) issue_strip(tile + gnum, buf0 + (size_t)(cur ^ 1) * bbytes);
    const int n = tile / per_img, rr = tile - n * per_img;
    const int ty = rr / p.tiles_x, tx = rr - ty * p.tiles_x;
    const int oy0 = ty * TRr, ox0 = tx * TC;
    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* xa0 = buf + (size_t)((wave * 16 * TJ + frow) * PITCH + fk * 16);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const char* xt = xa0 + (size_t)(p.toff[t] * TC * PITCH);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        bf16x8 xf[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xt + j * 16 * PITCH + kk * 64);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][kk][i], xf[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);               // (hoisting every tap's fragment reads to the top spilled 46 registers)
    }
    __syncthreads();                                   // the strip is free: its buffer becomes the output slab
    char* slab = buf;                                  // [TPX pixels][SROW]
    // (the activation is chosen once per tile, not per element: with the runtime switch inside the 64-element loops the
    // epilogue was ~6000 lines of mostly skipped code per tile and the kernel VALU / branch bound)
    auto epilogue = [&](auto plain_tag) {
      constexpr bool PLAIN = decltype(plain_tag)::value;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int co = i * 16 + fk * 4;
      const f32x4 bia = (bias_net && co < d.Co) ? *reinterpret_cast<const f32x4*>(bias_net + co) : f32x4{0.f, 0.f, 0.f, 0.f};
      float s1[4], s2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) s1[r] = s2[r] = 0.f;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int pl = wave * 16 * TJ + j * 16 + frow;
        const bool inside = oy0 + pl / TC < d.Hc && ox0 + pl % TC < d.Wc;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r] + bia[r];
          if (inside) { s1[r] += v[r]; s2[r] += v[r] * v[r]; }
          if constexpr (!PLAIN) v[r] = act_call(v[r], d.act, d.slope);
        }
        uint2 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(slab + pl * SROW + co * 2) = o;
      }
      if (want_stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row16_sum(s1[r]), q = row16_sum(s2[r]);
          if (frow == 0) {
            red[(wave * CO + co + r) * 2 + 0] = a;
            red[(wave * CO + co + r) * 2 + 1] = q;
          }
        }
      }
    }
    };
    if (d.act == GS_ACT_NONE) epilogue(std::true_type{}); else epilogue(std::false_type{});
    __syncthreads();
    {
      constexpr int LPP = CO / 8;                      // lanes (16 B each) per pixel
      for (int q = tid; q < TPX * LPP; q += 256) {
        const int pl = q / LPP, piece = q - pl * LPP;
        const int oy = oy0 + pl / TC, ox = ox0 + pl % TC;
        if (oy < d.Hc && ox < d.Wc && piece * 8 < d.Co) {
          const size_t opix = ((size_t)n * d.Ho + oy) * d.Wo + ox;
          *reinterpret_cast<uint4*>(p.out + (opix * d.out_cs + d.out_co + piece * 8) * 2) =
              *reinterpret_cast<const uint4*>(slab + pl * SROW + piece * 16);
        }
      }
    }
    if (want_stats && tid < CO && tid < d.Co) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { a += red[(w * CO + tid) * 2]; q += red[(w * CO + tid) * 2 + 1]; }
      float* sp = p.stats + (((size_t)n * d.stats_slots + d.stats_slot0 + ty * p.tiles_x + tx) * 2) * d.Co;
      sp[tid] = a;
      sp[d.Co + tid] = q;
    }
    cur ^= 1;
  }
}

struct Plan { bool ok; int hmin, span, dwc, ci, co, lds, tr; bool regs; };
Plan plan(const gs_gconv_desc* d) {
  Plan h{};
  const int minb = gs_opt(GS_OPT_HSTRIP);
  if (minb == 0 || d->si != 1 || d->so != 1 || d->T < 2 || d->T > 8 || d->accumulate) return h;
  if (d->Di != 1 || d->Do != 1 || d->Dc != 1 || d->pz || d->py || d->px || d->Hc != d->Ho || d->Wc != d->Wo) return h;
  if ((d->Ci != 32 && d->Ci != 64) || d->Co > 64 || d->Co % 8 != 0) return h;
  int lo = 127, hi = -128;
  for (int t = 0; t < d->T; ++t) {
    if (d->dd[t] != 0 || d->dw[t] != d->dw[0]) return h;          // vertical taps in one column
    if (d->dh[t] < lo) lo = d->dh[t];
    if (d->dh[t] > hi) hi = d->dh[t];
  }
  if (hi - lo > 7) return h;
  h.hmin = lo; h.span = hi - lo; h.dwc = d->dw[0];
  h.ci = d->Ci; h.co = d->Co <= 32 ? 32 : 64;
  h.tr = TR;
  if ((long long)d->Hi * d->Wi * d->in_cs * 2 >= (1LL << 31)) return h;
  // the persistent register-weight form (hstripr_kernel): the two shapes of the ResNets' W-folded k7 convs
  // (measured, profiles/r04_hstrip_forms.txt: ahead for the launches without statistics — the data gradients —, behind by 2 us
  // for the forwards, whose per-tile statistics epilogue it runs with fewer waves in flight; option value 3 takes every launch)
  if (gs_opt(GS_OPT_HSTRIP_REGS) && (d->stats_slots == 0 || gs_opt(GS_OPT_HSTRIP_REGS) >= 2) && d->T == 7 && ((h.ci == 32 && h.co == 64) || (h.ci == 64 && h.co == 32)) &&
      d->Hi < 32768 && d->Wi < 32768) {
    const int tr = h.ci == 32 ? 32 : 16;
    const long long ty = (d->Ho + tr - 1) / tr, tx = (d->Wo + TC - 1) / TC;
    const long long tiles = (long long)d->N * ty * tx;
    const int pitch = h.ci == 64 ? 160 : 96;
    const int hinstr = ((tr + h.span) * TC * (pitch / 16) + 63) / 64;
    const int slab = tr * TC * (h.co * 2 + 16);
    const int bbytes = hinstr * 1024 > slab ? hinstr * 1024 : slab;
    const long long tab = (ty * (tr + h.span) + tx * TC) * 2;
    const long long lds = 2LL * bbytes + 4 * h.co * 2 * 4 + (tab + 15) / 16 * 16;
    // (option value 1: launches of at least two rounds of 512 workgroups; >= 2: any launch — op tests)
    if (hinstr <= 32 && lds <= 80 * 1024 && tiles >= (gs_opt(GS_OPT_HSTRIP_REGS) >= 2 ? 1 : 1024) && tiles < (1LL << 31)) {
      h.regs = true; h.tr = tr; h.lds = (int)lds; h.ok = true;
      return h;
    }
  }
  const long long blocks = (long long)d->N * ((d->Ho + TR - 1) / TR) * ((d->Wo + TC - 1) / TC);
  if (blocks < minb || blocks >= (1LL << 31)) return h;
  const int pitch = d->Ci == 64 ? 160 : 96;
  const int wbytes = ((h.co * (d->T * d->Ci * 2 + 32) + 1023) / 1024) * 1024;
  const int hbytes = (((TR + h.span) * TC * (pitch / 16) + 63) / 64) * 1024;
  const int slab = 256 * (h.co * 2 + 16);
  h.lds = wbytes + (hbytes > slab ? hbytes : slab);
  if (h.lds > 80 * 1024 || wbytes < 4 * h.co * 2 * 4) return h;    // two workgroups per CU; red[] fits the weight area
  h.ok = true;
  return h;
}

template <int CI, int CO>
int launch_s(const HStripK& k, long long blocks, int lds, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hstrip_kernel<CI, CO>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    configured = true;
  }
  hipLaunchKernelGGL((hstrip_kernel<CI, CO>), dim3((unsigned)blocks), dim3(256), lds, st, k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
template <int CI, int CO, int TRr>
int launch_r(const HStripK& k, long long groups, int lds, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hstripr_kernel<CI, CO, TRr>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    configured = true;
  }
  hipLaunchKernelGGL((hstripr_kernel<CI, CO, TRr>), dim3((unsigned)groups), dim3(256), lds, st, k);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
}  // namespace

// statistics slots per image when the class runs here (one per 32 x 8 tile), 0 when it does not
int gs_hstrip_slots(const gs_gconv_desc* d) {
  gs_gconv_desc q = *d;
  q.stats_slots = 1;            // the question is about a launch WITH statistics (the forms differ in their tile rows)
  const Plan h = plan(&q);
  return h.ok ? ((d->Ho + h.tr - 1) / h.tr) * ((d->Wo + TC - 1) / TC) : 0;
}

int gs_hstrip_try(const gs_gconv_desc* d, const void* in, const void* w_pack, const float* bias, void* out, float* stats,
                  void* stream, int* handled, const gs_twin* tw) {
  *handled = 0;
  const Plan h = plan(d);
  if (!h.ok) return 0;
  HStripK k;
  k.nsplit = tw ? tw->n_split : 0x7fffffff;
  k.w_delta = tw ? tw->w_delta : 0;
  k.bias_delta = tw ? tw->bias_delta : 0;
  k.in = static_cast<const char*>(in);
  k.w = static_cast<const char*>(w_pack);
  k.bias = bias;
  k.out = static_cast<char*>(out);
  k.stats = stats;
  k.zero = static_cast<const char*>(gs_zero_page());
  GS_REQUIRE(k.zero, "gs_gconv_forward: library not initialised (call gs_init)");
  k.tiles_y = (d->Ho + h.tr - 1) / h.tr;
  k.tiles_x = (d->Wo + TC - 1) / TC;
  k.hmin = h.hmin; k.span = h.span; k.dwc = h.dwc;
  for (int t = 0; t < d->T; ++t) k.toff[t] = d->dh[t] - h.hmin;
  k.d = *d;
  const long long blocks = (long long)d->N * k.tiles_y * k.tiles_x;
  hipStream_t st = static_cast<hipStream_t>(stream);
  *handled = 1;
  k.gsplit = 0x7fffffff;
  if (h.regs) {
    long long groups = blocks < 512 ? blocks : 512;            // two persistent workgroups per CU
    if (tw) {                                                  // ... half of them per network
      const long long per_net = (long long)tw->n_split * k.tiles_y * k.tiles_x;
      groups = per_net < 256 ? per_net : 256;
      k.gsplit = (int)groups;
      groups *= 2;
    } else {
      k.gsplit = (int)groups;
    }
    if (h.ci == 32) return launch_r<32, 64, 32>(k, groups, h.lds, st);
    return launch_r<64, 32, 16>(k, groups, h.lds, st);
  }
  if (h.ci == 64 && h.co == 64) return launch_s<64, 64>(k, blocks, h.lds, st);
  if (h.ci == 64 && h.co == 32) return launch_s<64, 32>(k, blocks, h.lds, st);
  if (h.ci == 32 && h.co == 64) return launch_s<32, 64>(k, blocks, h.lds, st);
  return launch_s<32, 32>(k, blocks, h.lds, st);
}
