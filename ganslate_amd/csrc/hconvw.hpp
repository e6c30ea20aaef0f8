// Kernel arguments shared by the halo-resident kernels of the wide 3x3 stride-1 layers (hconvw.hip)
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
  int ntiles;                  // N * tiles_m * tiles_n; a workgroup walks tiles b, b + gridDim.x, ... (hconvw.hip)
  int nsplit;                  // twin batch: images >= nsplit use the second network's weights (INT_MAX: one network)
  long long w_delta;           // byte distance from the first network's pack to the second's
  long long bias_delta;        // the same for the bias vector, in floats
  gs_gconv_desc d;
  gs_gconv_fuse f;             // RING: the consumer's InstanceNorm backward sums ride in the epilogue (gs_gconv_forward_fused)
};
