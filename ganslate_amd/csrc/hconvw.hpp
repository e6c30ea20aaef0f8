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
  // APPLY (gs_gconv_ring_apply): the consumer's whole InstanceNorm backward rides in the epilogue — after an in-launch
  // rendezvous of the boxes of an image the kernel writes dy (to `out`) and, with a residual-join gradient, the total gradient
  char* out2;                  // total gradient gx + g2 (NULL: not wanted / no g2)
  int* sync;                   // [2 * N * tiles_n + 1] arrival / departure counters per (image, channel tile), zero between launches
  float inv_hw;
  int dbg;                     // timing ablations of the rendezvous (option ring_apply > 1; results are wrong then)
};
