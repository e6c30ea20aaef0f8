// Round-3 experiment, removed from libganslate_hip.so in round 4 (hstrip.hip then): the persistent, double-buffered form
// of the W-folded k7 boundary-conv kernel. Not built; kept as the source of the measurements quoted in its header.
// ---- persistent form --------------------------------------------------------------------------------------------------------
// hstrip_kernel runs load -> compute -> store once per workgroup, two workgroups per CU: 55-58 us for launches whose HBM floor
// is ~20 (profiles/r02_halo_kernels_pmc.txt: input read once, 13-17 % matrix busy — the chain itself is what is left). Here ONE
// workgroup per CU loads the layer's weights once and walks over tiles (tile b + gridDim.x next) with the input strip
// double-buffered: the LDS-DMA of the next strip is issued before this tile's taps and waited for behind them, this tile's
// stores are issued after that wait and nothing waits for them until a whole tile later. Statistics: one slot per (tile, wave),
// so the tile needs no third barrier. Fragment reads through inline asm with counted waits (the DMA in the tile loop makes
// hipcc wait lgkmcnt(0) around every read group otherwise).
// OFF by default (gs_set_option("hstrip_persist", smallest number of tiles)): bit-identical results, but 72 / 62 / 69 / 74 us
// against 53 / 42 / 46 / 55 for the four boundary launches of the headline (tools/bench_kernels.py stemw / outw, same box) and
// -1 % on the step: with 125-148 KB of LDS (weights + two strips + slab) there is ONE 4-wave workgroup per CU, and its own
// chain (DMA issue, taps, slab, stores: ~8 us per tile) is longer than half of the two interleaved chains it replaces. A form
// that keeps two workgroups per CU needs 16-row tiles without the slab and a swizzled 128-byte pitch for 64 channels.
template <int CI, int CO>
__global__ __launch_bounds__(256) void hstripp_kernel(const HStripK p, const int tiles) {
  constexpr int PITCH = CI == 64 ? 160 : 96;
  constexpr int PIECES = PITCH / 16;
  constexpr int TI = CO / 16, TJ = 4, KK = CI / 32;
  constexpr int SROW = CO * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const gs_gconv_desc& d = p.d;
  const int T = d.T;
  const int WROW = T * CI * 2 + 32;
  const int wbytes = ((CO * WROW + 1023) / 1024) * 1024;
  const int hpx = (TR + p.span) * TC;
  const int hinstr = (hpx * PIECES + 63) / 64;
  const int hbytes = hinstr * 1024;
  char* wl = smem;                               // [CO][WROW]
  char* hbuf = smem + wbytes;                    // 2 x [(TR + span) * TC pixels][PITCH]
  char* slab = hbuf + 2 * hbytes;                // [256 pixels][SROW]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;

  // the pieces of a strip this thread stages are the same for every tile: (hy, hx, part) decoded once
  constexpr int NHP = 12;                        // <= 48 DMA instructions per strip over 4 waves
  int hcode[NHP];
#pragma unroll
  for (int i = 0; i < NHP; ++i) {
    const int q = (i * 4 + wave) * 64 + lane;
    const int v = q / PIECES, part = q - v * PIECES;
    const int hy = v / TC, hx = v - hy * TC;
    hcode[i] = (v < hpx && part < CI / 8) ? (part << 16 | hy << 8 | hx) : -1;
  }
  auto issue_strip = [&](int b, int buf) {
    const int tx = b % p.tiles_x;
    const int ty = (b / p.tiles_x) % p.tiles_y;
    const int n = b / (p.tiles_x * p.tiles_y);
    const char* in_n = p.in + ((size_t)n * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
    const int y0 = ty * TR + p.hmin, x0 = tx * TC + p.dwc;
#pragma unroll
    for (int i = 0; i < NHP; ++i) {
      const int inst = i * 4 + wave;
      if (inst < hinstr) {                       // wave-uniform
        const int c = hcode[i];
        bool ok = c >= 0;
        int iy = border_index(y0 + ((c >> 8) & 255), d.Hi, d.border, ok);
        int ix = border_index(x0 + (c & 255), d.Wi, d.border, ok);
        iy = min(max(iy, 0), d.Hi - 1);
        ix = min(max(ix, 0), d.Wi - 1);
        const char* src = ok ? in_n + ((size_t)(iy * d.Wi + ix) * d.in_cs + ((c >> 16) & 15) * 8) * 2 : p.zero;
        glds16(src, hbuf + (size_t)buf * hbytes + inst * 1024);
      }
    }
  };
  {
    const int rp1 = T * CI / 8 + 2;              // 16-B pieces per LDS row incl. the two pad pieces
    const int winstr = (CO * rp1 + 63) / 64;
    for (int inst = wave; inst < winstr; inst += 4) {
      const int q = inst * 64 + lane;
      const int r = q / rp1, piece = q - r * rp1;
      const bool ok = r < CO && r < d.w_rows && piece < rp1 - 2;
      const char* src = ok ? p.w + ((size_t)r * d.Kp + piece * 8) * 2 : p.zero;
      glds16(src, wl + inst * 1024);
    }
  }
  f32x4 bia[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int co = i * 16 + fk * 4;
    bia[i] = (p.bias && co < d.Co) ? *reinterpret_cast<const f32x4*>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const unsigned smem0 = lds_addr(smem);
  const unsigned wa0 = smem0 + (unsigned)(frow * WROW + fk * 16);
  const unsigned xrel = (unsigned)((wave * 64 + frow) * PITCH + fk * 16);

  int b = blockIdx.x;
  if (b < tiles) issue_strip(b, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // weights, the first strip
  int cur = 0;
  for (; b < tiles; b += gridDim.x, cur ^= 1) {
    __syncthreads();       // every wave's pieces of this strip landed (each waited for its own before the previous barrier);
                           // the other buffer and the slab are free
    const int nb = b + (int)gridDim.x;
    if (nb < tiles) issue_strip(nb, cur ^ 1);

    // ---- taps out of LDS: step s = (tap, 32-deep k slice), the reads of step s + 1 issued before the MFMAs of step s ----
    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned xa0 = smem0 + (unsigned)wbytes + (unsigned)(cur * hbytes) + xrel;
    bf16x8 wA[TI], xA[TJ], wB[TI], xB[TJ];
    auto load = [&](int t, auto kk_tag, bf16x8 (&wf)[TI], bf16x8 (&xf)[TJ]) {
      constexpr int kk = decltype(kk_tag)::value;
      const unsigned wt = wa0 + (unsigned)(t * CI * 2), xt = xa0 + (unsigned)(p.toff[t] * TC * PITCH);
      static_for<0, TI>([&](auto i_tag) { lds_read128<kk * 64>(wf[decltype(i_tag)::value], wt + (unsigned)(decltype(i_tag)::value * 16 * WROW)); });
      static_for<0, TJ>([&](auto j_tag) { lds_read128<decltype(j_tag)::value * 16 * PITCH + kk * 64>(xf[decltype(j_tag)::value], xt); });
    };
    auto wait_mma = [&](auto n_tag, bf16x8 (&wf)[TI], bf16x8 (&xf)[TJ]) {
      constexpr int N = decltype(n_tag)::value;
      if constexpr (TI == 2) gs_lgkm_wait<N>(wf[0], wf[1], xf[0], xf[1], xf[2], xf[3]);
      else gs_lgkm_wait<N>(wf[0], wf[1], wf[2], wf[3], xf[0], xf[1], xf[2], xf[3]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    using NF = std::integral_constant<int, TI + TJ>;
    using N0 = std::integral_constant<int, 0>;
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, KK - 1>;
    if constexpr (KK == 1) {
      // steps = taps; two per iteration (A, B sets); T may be odd
      load(0, K0{}, wA, xA);
      int t = 0;
      for (; t + 1 < T; t += 2) {
        load(t + 1, K0{}, wB, xB);
        wait_mma(NF{}, wA, xA);
        if (t + 2 < T) { load(t + 2, K0{}, wA, xA); wait_mma(NF{}, wB, xB); }
        else wait_mma(N0{}, wB, xB);
      }
      if (t < T) wait_mma(N0{}, wA, xA);
    } else {
      // two k slices per tap: slice 0 in set A, slice 1 in set B
      load(0, K0{}, wA, xA);
      for (int t = 0; t < T; ++t) {
        load(t, K1{}, wB, xB);
        wait_mma(NF{}, wA, xA);
        if (t + 1 < T) { load(t + 1, K0{}, wA, xA); wait_mma(NF{}, wB, xB); }
        else wait_mma(N0{}, wB, xB);
      }
    }

    // ---- epilogue: bias, statistics (slot = (tile, wave)), activation, coalesced stores through the slab ----
    const int tx = b % p.tiles_x;
    const int ty = (b / p.tiles_x) % p.tiles_y;
    const int n = b / (p.tiles_x * p.tiles_y);
    const int oy0 = ty * TR, ox0 = tx * TC;
    float s1[TI][4], s2[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int co = i * 16 + fk * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int pl = wave * 64 + j * 16 + frow;
        const bool inside = oy0 + pl / TC < d.Hc && ox0 + pl % TC < d.Wc;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r] + bia[i][r];
          if (inside) { s1[i][r] += v[r]; s2[i][r] += v[r] * v[r]; }
          v[r] = apply_act(v[r], d.act, d.slope);
        }
        uint2 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(slab + pl * SROW + co * 2) = o;
      }
    }
    // the next strip (issued a whole tap loop ago) and the previous tile's stores: waited for HERE, so that nothing waits on
    // the stores issued below until a tile later
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    if (d.stats_slots > 0) {
      float* sp = p.stats + (((size_t)n * d.stats_slots + d.stats_slot0 + (ty * p.tiles_x + tx) * 4 + wave) * 2) * d.Co;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row16_sum(s1[i][r]), q = row16_sum(s2[i][r]);
          const int c = i * 16 + fk * 4 + r;
          if (frow == 0 && c < d.Co) { sp[c] = a; sp[d.Co + c] = q; }
        }
    }
  }
}

