// Kernel arguments of the im2col conv kernels (gconv.hip: one tile per workgroup; pconv.hip: the persistent form).
#pragma once
#include "common.hpp"

// What differs between the output-parity classes of one stride-2 transposed conv / stride-2 data gradient (same tensors,
// same class extent, same Co): weight block, taps, output phase, statistics slot. Up to 8 classes (3-D) of up to 8 taps
// ride in one launch (gs_gconv_forward_multi): workgroup -> (class, image, pixel tile, channel tile).
constexpr int GS_MULTI_MAX_CLS = 8, GS_MULTI_MAX_TAPS = 8;
struct GConvCls {
  long long w_off;             // byte offset of this class's [w_rows][Kp] block from GConvK::w
  int T, Kp, pz, py, px, stats_slot0, nh, nw;
  signed char ud[GS_MULTI_MAX_TAPS], uh[GS_MULTI_MAX_TAPS], uw[GS_MULTI_MAX_TAPS];
  unsigned char tap_h[GS_MULTI_MAX_TAPS], tap_w[GS_MULTI_MAX_TAPS];
};

struct GConvK {
  const char* in;
  const char* w;
  const float* bias;
  char* out;
  float* stats;
  const char* zero;
  int tiles_m, tiles_n, ci_shift;
  float rcp_wc, rcp_hc;
  // taps factored into distinct (depth, row) / column offsets (every lowering produces a product grid of taps)
  int nh, nw;
  signed char ud[64], uh[64], uw[16];
  unsigned char tap_h[GS_MAX_TAPS], tap_w[GS_MAX_TAPS];
  gs_gconv_fuse f;             // f.partial != nullptr: first pass of the consumer's InstanceNorm backward in the epilogue
  int fuse_slots;
  // split-K (few output tiles, long K: the deep U-Net / PatchGAN-tail layers): workgroup (tile, sp) runs K-steps
  // [sp*nk/splits, (sp+1)*nk/splits) and writes raw fp32 sums to partial[sp][output pixel][Co]; gconv_splitk_finalize
  // adds them up and applies the usual epilogue
  int splits;
  float* partial;
  long long split_stride;      // floats per split = N * Do*Ho*Wo * Co
  // twin batch (gs_twin): images [nsplit, N) read the weight pack / bias w_delta / bias_delta bytes further on
  int nsplit;
  long long w_delta, bias_delta;
  int n_cls;                   // > 0: merged launch over cls[0..n_cls) (their fields replace the per-class ones of p / d)
  GConvCls cls[GS_MULTI_MAX_CLS];
  gs_gconv_desc d;
};

