// Round-3 experiment, removed from libganslate_hip.so in round 4 (hconv.hip then): the persistent resident-weight form of
// the narrow halo kernel for 16 -> 16 channel k5 volume layers. Not built; kept as the source of the measurements quoted
// in its header (profiles/r03_hconv_forms.txt).
// ---- persistent form for the 16-output-channel volume layers (Vnet3D's coupling / input convs at the full resolution) --------
// hconv_kernel re-streams the layer's weights for every box: 64 KB per 256 voxels for a 16 -> 16 channel 5x5x5 kernel,
// 524 MB of L2 -> LDS traffic per launch at 128^3 next to 300 MB of halo boxes, each weight stage waited for behind a
// barrier. Here ONE workgroup per CU keeps the whole [16][Kp] weight block in LDS for its lifetime and walks over boxes
// (box b + gridDim.x next) with the halo box double-buffered: the DMA of the next box flies under the taps of this one,
// and the tap loop has no barrier and no wait on memory in it. 8 waves: wave w owns the 64 voxels of box plane w & 3 and
// the K-steps of parity w >> 2 (the two halves of K meet through 16 KB of LDS in front of the epilogue), so 8 waves share
// one 4 x 8 x 8 box and the second halo buffer fits: 66 + 2 x 37 + 16 KB. Statistics: one slot per (box, wave).
// OFF by default (gs_set_option("hconv_persist", smallest number of boxes)): measured 191 us against 204 for hconv_kernel's
// 8 x 8 x 8 form on the 16 -> 16 channel layer at 128^3 (tools/probe/hconv_forms.py) but 73.4 against 72.6 ms in the brats
// step, where two streams share the chip. What the exercise established: with 16 output channels every 1-KB B fragment
// feeds ONE MFMA, so all forms of this layer are LDS-read bound (5 ds_read_b128 per 4 MFMAs = 4.3 us of LDS time per 256
// voxels at 128 B/clk, 138 us per launch at 128^3) and run at 0.65-0.73 PFLOP/s; the weight stream was not the limit.
template <int CC, int NS>      // NS = K-steps per wave = Kp / 64 (the pack's zero padding makes every wave's count equal)
__global__ __launch_bounds__(512) void hconvp_kernel(const HConvK p, const int wpitch, const int nboxes) {
  constexpr int PP = CC / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const gs_gconv_desc& d = p.d;
  const int HV = p.HD * p.HH * p.HW, hhw = p.HH * p.HW;
  const int pieces = HV * PP;
  const int hbytes = (pieces * 16 + 1023) / 1024 * 1024;
  const int wpieces = 16 * (wpitch >> 4);
  const int wbytes = (wpieces * 16 + 1023) / 1024 * 1024;
  int* toff = reinterpret_cast<int*>(smem);                        // [GS_MAX_TAPS]
  char* wres = smem + GS_MAX_TAPS * 4;
  char* hbuf = wres + wbytes;                                      // 2 x hbytes
  float* scratch = reinterpret_cast<float*>(hbuf + 2 * hbytes);    // [4 planes][16 values][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = wave & 3, khalf = wave >> 2;
  const int row = lane & 15, kg = lane >> 4;

  for (int t = tid; t < d.T; t += 512)                             // offset << 8 | row-offset parity (hconv_kernel's swizzle)
    toff[t] = ((((int)d.dd[t] - p.dmin) * p.HH + ((int)d.dh[t] - p.hmin)) * p.HW + ((int)d.dw[t] - p.wmin)) << 8 |
              (((int)d.dh[t] - p.hmin) & 1);
  // ---- resident weights: LDS image [16 rows][wpitch bytes], the last piece of a row is padding ----
  {
    const int ppr = wpitch >> 4;
    for (int q0 = wave * 64; q0 < wpieces; q0 += 512) {
      const int q = q0 + lane;
      const int r = q / ppr, pc = q - r * ppr;
      const bool ok = q < wpieces && pc < ppr - 1 && r < d.w_rows;
      unsigned off = ((unsigned)r * (unsigned)d.Kp + (unsigned)pc * 8u) * 2u;
      asm volatile("" : "+v"(off));
      glds16(ok ? p.w + off : p.zero, wres + (size_t)q0 * 16);
    }
  }
  // the pieces this thread stages are the same for every box: decode them once (two runtime divisions per piece per box were
  // a third of the kernel's VALU work)
  constexpr int NHP = 5;                                           // <= 2560 pieces (8 x 12 x 12 voxels x 2)
  int hcode[NHP];                                                  // hz << 16 | hy << 8 | hx, or -1; the half rides in bit 24
#pragma unroll
  for (int i = 0; i < NHP; ++i) {
    const int q = (i * 8 + wave) * 64 + lane;
    const int v = q / PP, part = q - v * PP;
    const int hz = v / hhw, r2 = v - hz * hhw;
    const int hy = r2 / p.HW, hx = r2 - hy * p.HW;
    const int spart = CC == 16 ? part ^ (hy & 1) : part;           // bank swizzle, see hconv_kernel
    hcode[i] = q < pieces ? (spart << 24 | hz << 16 | hy << 8 | hx) : -1;
  }
  auto issue_halo = [&](int box, int buf) {
    int b = box;
    const int bx = b % p.nbw; b /= p.nbw;
    const int by = b % p.nbh; b /= p.nbh;
    const int bz = b % p.nbd;
    const int n = b / p.nbd;
    const int oz0 = bz * p.BD + p.dmin, oy0 = by * p.BH + p.hmin, ox0 = bx * p.BW + p.wmin;
    const char* in_n = p.in + ((size_t)n * d.Di * d.Hi * d.Wi * d.in_cs + d.in_co) * 2;
#pragma unroll
    for (int i = 0; i < NHP; ++i) {
      const int q0 = (i * 8 + wave) * 64;
      if (q0 < pieces) {                                           // wave-uniform
        const int c = hcode[i];
        bool ok = c >= 0;
        int iz = border_index(oz0 + ((c >> 16) & 255), d.Di, d.border, ok);
        int iy = border_index(oy0 + ((c >> 8) & 255), d.Hi, d.border, ok);
        int ix = border_index(ox0 + (c & 255), d.Wi, d.border, ok);
        iz = min(max(iz, 0), d.Di - 1);
        iy = min(max(iy, 0), d.Hi - 1);
        ix = min(max(ix, 0), d.Wi - 1);
        unsigned off = ((unsigned)((iz * d.Hi + iy) * d.Wi + ix) * (unsigned)d.in_cs + (unsigned)(((c >> 24) & 1) * 8)) * 2u;
        asm volatile("" : "+v"(off));
        glds16(ok ? in_n + off : p.zero, hbuf + (size_t)buf * hbytes + (size_t)q0 * 16);
      }
    }
  };

  // this wave's 4 x 16 voxels inside a box (plane = box depth slice), as halo byte offsets
  int pb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pl = j * 16 + row;                 // voxel inside the 8 x 8 plane
    const int ly = pl / p.BW, lx = pl - ly * p.BW;
    pb[j] = ((plane * p.HH + ly) * p.HW + lx) * CC * 2;
  }
  const char* wrow = wres + row * wpitch + kg * 16;
  // this lane's tap offset (bytes into the halo, incl. its swizzled half) of each of this wave's K-steps u = khalf + 2 s: the
  // same for every box, so they live in registers and the tap loop's reads depend on nothing but the buffer base
  __syncthreads();                                     // tap table visible
  int vo[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    int tap = ((khalf + 2 * s) * 32 + kg * 8) >> p.cc_shift;
    tap = tap < d.T ? tap : 0;                         // past the last tap the weights are zero: any finite B
    const int e = toff[tap];
    vo[s] = CC == 16 ? (e >> 8) * 32 + (((kg & 1) ^ ((e ^ (row >> 3)) & 1)) << 4) : (e >> 8) * 16;
  }
  const unsigned waddr = lds_addr(wrow) + (unsigned)(khalf * 64);
  // epilogue split: the K halves meet through LDS, and each half FINISHES two of the four 16-voxel groups of its plane
  // (waves 0-3: groups 0, 1; waves 4-7: groups 2, 3), so all 8 waves share the epilogue
  const int co = kg * 4;
  f32x4 bia = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias && co < d.Co) bia = *reinterpret_cast<const f32x4*>(p.bias + co);

  int box = blockIdx.x;
  if (box < nboxes) issue_halo(box, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first halo box, the weights
  int cur = 0;
  for (; box < nboxes; box += gridDim.x, cur ^= 1) {
    __syncthreads();                                   // this box's halo landed for every wave (each waited for its own
                                                       // pieces before the previous barrier); the other buffer and the scratch are free
    if (box + (int)gridDim.x < nboxes) issue_halo(box + gridDim.x, cur ^ 1);
    const char* halo = hbuf + (size_t)cur * hbytes;

    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Three fragment sets: the 5 reads of step s + 2 are issued before the 4 MFMAs of step s. Reads through inline asm with
    // counted waits (common.hpp): with the LDS-DMA in this loop hipcc waits lgkmcnt(0) in front of every MFMA group and
    // every read group, i.e. read - wait - MFMA.
    unsigned hb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) hb[j] = lds_addr(halo) + (unsigned)pb[j];
    bf16x8 wf[3], xf[3][4];
    auto load_frags = [&](auto s_tag) {
      constexpr int s_ = decltype(s_tag)::value, set = s_ % 3;
      lds_read128<s_ * 128>(wf[set], waddr);
#pragma unroll
      for (int j = 0; j < 4; ++j) lds_read128<0>(xf[set][j], hb[j] + (unsigned)vo[s_]);
    };
    load_frags(std::integral_constant<int, 0>{});
    if constexpr (NS > 1) load_frags(std::integral_constant<int, 1>{});
    static_for<0, NS>([&](auto s_tag) {
      constexpr int s_ = decltype(s_tag)::value, set = s_ % 3;
      if constexpr (s_ + 1 < NS) gs_lgkm_wait<5>(wf[set], xf[set][0], xf[set][1], xf[set][2], xf[set][3]);
      else gs_lgkm_wait<0>(wf[set], xf[set][0], xf[set][1], xf[set][2], xf[set][3]);
      if constexpr (s_ + 2 < NS) load_frags(std::integral_constant<int, s_ + 2>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[set], xf[set][j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- the two K halves meet: every wave hands the two groups it does not finish to its partner ----
    auto finish = [&](auto kh_tag) {
    constexpr int j0 = decltype(kh_tag)::value * 2;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 4; ++r) scratch[((wave ^ 4) * 8 + jj * 4 + r) * 64 + lane] = acc[(j0 ^ 2) + jj][r];
    // the next box's halo pieces of this wave (issued a whole tap loop ago) and the stores of the previous box: waited for
    // HERE, so that nothing waits on the stores issued below until a tap loop later
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue (hconv_kernel's): bias, statistics, activation, [accumulate], 8-B NHWC stores ----
    int b = box;
    const int bx = b % p.nbw; b /= p.nbw;
    const int by = b % p.nbh; b /= p.nbh;
    const int bz = b % p.nbd;
    const int n = b / p.nbd;
    const int bslot = (bz * p.nbh + by) * p.nbw + bx;
    const int oz = bz * p.BD + plane;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int pl = (j0 + jj) * 16 + row;
      const int ly = pl / p.BW, lx = pl - ly * p.BW;
      const int oy = by * p.BH + ly, ox = bx * p.BW + lx;
      const bool pv = oz < d.Do && oy < d.Ho && ox < d.Wo;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[j0 + jj][r] + scratch[(wave * 8 + jj * 4 + r) * 64 + lane] + bia[r];
        if (pv) { s1[r] += v[r]; s2[r] += v[r] * v[r]; }
        v[r] = apply_act(v[r], d.act, d.slope);
      }
      if (pv && co < d.Co) {
        const size_t opix = (((size_t)n * d.Do + oz) * d.Ho + oy) * d.Wo + ox;
        uint2* dst = reinterpret_cast<uint2*>(p.out + (opix * d.out_cs + d.out_co + co) * 2);
        uint2 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        if (d.accumulate) {
          const uint2 old = *dst;
          o.x = pack_bf2(bf_lo(o.x) + bf_lo(old.x), bf_hi(o.x) + bf_hi(old.x));
          o.y = pack_bf2(bf_lo(o.y) + bf_lo(old.y), bf_hi(o.y) + bf_hi(old.y));
        }
        *dst = o;
      }
    }
    if (d.stats_slots > 0) {
      float* sp = p.stats + (((size_t)n * d.stats_slots + d.stats_slot0 + bslot * 8 + wave) * 2) * d.Co;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s1[r]), q = row16_sum(s2[r]);
        if (row == 0 && co + r < d.Co) { sp[co + r] = a; sp[d.Co + co + r] = q; }
      }
    }
    };
    if (khalf == 0) finish(std::integral_constant<int, 0>{}); else finish(std::integral_constant<int, 1>{});
  }
}

