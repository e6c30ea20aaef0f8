/* Compiled as C11 by tests/test_abi_cpu.py: proves include/ganslate_hip.h is a C header (no C++), prints the layout of
 * every descriptor struct for comparison with the ctypes mirrors in ganslate_amd/hip/lib.py, fills a gs_gconv_desc the
 * way a C caller would and resolves the entry points of libganslate_hip.so with dlopen/dlsym (no GPU call is made:
 * gs_gconv_stat_slots / gs_tile_m / gs_gconv_splitk_ws_floats are host-side planning functions). */
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "ganslate_hip.h"

#define SZ(T) printf("sizeof " #T " %zu\n", sizeof(T))
#define OFF(T, f) printf("offsetof " #T " " #f " %zu\n", offsetof(T, f))

typedef int (*stat_slots_fn)(const gs_gconv_desc*);
typedef int64_t (*ws_floats_fn)(const gs_gconv_desc*);
typedef const char* (*last_error_fn)(void);

int main(int argc, char** argv) {
  SZ(gs_gconv_desc); OFF(gs_gconv_desc, Di); OFF(gs_gconv_desc, Ho); OFF(gs_gconv_desc, T); OFF(gs_gconv_desc, slope);
  OFF(gs_gconv_desc, accumulate); OFF(gs_gconv_desc, dh); OFF(gs_gconv_desc, dw); OFF(gs_gconv_desc, dd);
  SZ(gs_wgrad_desc); OFF(gs_wgrad_desc, Hg); OFF(gs_wgrad_desc, dw_ld); OFF(gs_wgrad_desc, dh); OFF(gs_wgrad_desc, dd);
  SZ(gs_gconv_fuse); OFF(gs_gconv_fuse, partial); OFF(gs_gconv_fuse, Dy); OFF(gs_gconv_fuse, slope);
  SZ(gs_norm_ex_desc); OFF(gs_norm_ex_desc, slope); OFF(gs_norm_ex_desc, drop_p); OFF(gs_norm_ex_desc, seed_hi);
  SZ(gs_pnorm_desc); OFF(gs_pnorm_desc, N); OFF(gs_pnorm_desc, gres_co);
  SZ(gs_patchnce_desc); OFF(gs_patchnce_desc, channels); OFF(gs_patchnce_desc, lambda_nce);
  SZ(gs_attn_desc); OFF(gs_attn_desc, C);
  SZ(gs_attn_params); OFF(gs_attn_params, wq); OFF(gs_attn_params, bv);
  if (argc < 2) return 0;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { printf("dlopen failed: %s\n", dlerror()); return 3; }
  stat_slots_fn slots = (stat_slots_fn)dlsym(h, "gs_gconv_stat_slots");
  stat_slots_fn tile_m = (stat_slots_fn)dlsym(h, "gs_tile_m");
  ws_floats_fn wsf = (ws_floats_fn)dlsym(h, "gs_gconv_splitk_ws_floats");
  last_error_fn err = (last_error_fn)dlsym(h, "gs_last_error");
  if (!slots || !tile_m || !wsf || !err) { printf("dlsym failed\n"); return 4; }
  /* the residual-block conv of Resnet2D at the headline shape (resnet2d.py:80-87): 8 x 64 x 64 x 256 -> 256, k3 reflect */
  gs_gconv_desc d;
  memset(&d, 0, sizeof d);
  d.N = 8; d.Hi = d.Wi = d.Ho = d.Wo = d.Hc = d.Wc = 64; d.Ci = d.Co = d.in_cs = d.out_cs = 256;
  d.Di = d.Do = d.Dc = 1; d.so = d.si = 1; d.T = 9; d.Kp = 9 * 256; d.w_rows = 256; d.border = GS_BORDER_REFLECT;
  for (int t = 0; t < 9; ++t) { d.dh[t] = (int8_t)(t / 3 - 1); d.dw[t] = (int8_t)(t % 3 - 1); }
  printf("call stat_slots %d\n", slots(&d));
  printf("call tile_m %d\n", tile_m(&d));
  printf("call splitk_ws_floats %lld\n", (long long)wsf(&d));
  /* the PatchGAN tail 512 -> 1 (patchgan2d.py:62) at 31 x 31: few tiles, long K -> the split-K plan asks for workspace */
  memset(&d, 0, sizeof d);
  d.N = 1; d.Hi = d.Wi = 31; d.Ho = d.Wo = d.Hc = d.Wc = 30; d.Ci = d.in_cs = 512; d.Co = d.out_cs = 8;
  d.Di = d.Do = d.Dc = 1; d.so = d.si = 1; d.T = 16; d.Kp = 16 * 512; d.w_rows = 8; d.border = GS_BORDER_ZERO;
  for (int t = 0; t < 16; ++t) { d.dh[t] = (int8_t)(t / 4 - 1); d.dw[t] = (int8_t)(t % 4 - 1); }
  printf("call tail_splitk_ws_floats %lld\n", (long long)wsf(&d));
  /* SelfAttentionBlock(256) on the 10^3 map of the self-attention discriminator at 128^3 inputs (selfattention_patchgan3d.py:58) */
  typedef int64_t (*attn_bytes_fn)(const gs_attn_desc*);
  attn_bytes_fn ab = (attn_bytes_fn)dlsym(h, "gs_attn_work_bytes");
  if (!ab) { printf("dlsym failed\n"); return 4; }
  gs_attn_desc ad = {1, 1000, 256};
  printf("call attn_work_bytes %lld\n", (long long)ab(&ad));
  dlclose(h);
  return 0;
}
