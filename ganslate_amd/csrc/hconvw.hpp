// Kernel arguments shared by the halo-resident kernels of the wide 3x3 stride-1 layers (hconvw.hip, hconvx.hip)
#pragma once
#include "common.hpp"

struct HConvWK {
  const char* in;
  const char* w;
  const float* bias;
  char* out;
  float* stats;
  const char* zero;
  int tiles_m, tiles_n, nbw;   // boxes per image, channel tiles, boxes per row
  int hh, hw, hmin, wmin;      // halo extent and smallest tap offsets
  int chunks;                  // Ci / 64
  gs_gconv_desc d;
  gs_gconv_fuse f;             // RING: the consumer's InstanceNorm backward sums ride in the epilogue (gs_gconv_forward_fused)
};
