// Shared device helpers for the gfx950 kernels of libganslate_hip.so.
// Wave = 64 lanes everywhere; fragment layouts pinned by tools/probe/probe.hip (profiles/r01_hw_probe.txt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ganslate_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

#define GS_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define GS_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// two fp32 -> packed bf16x2 (round-to-nearest-even, v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {
  f32x2 v = {a, b};
  bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<unsigned*>(&r);
}
__device__ __forceinline__ unsigned short f2bf(float a) { return (unsigned short)(pack_bf2(a, 0.f) & 0xffffu); }

// 16-byte LDS-DMA: LDS destination = wave-uniform base + lane*16; global source is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(GS_GLB(gsrc), GS_LDS(lds_wave_base), 16, 0, 0);
}

// m / d for 0 <= m < 2^24 with rcp = 1.0f/d (exact after one fix-up step each way)
__device__ __forceinline__ int div_small(int m, int d, float rcp) {
  int q = (int)((float)m * rcp);
  int r = m - q * d;
  if (r < 0) { --q; } else if (r >= d) { ++q; }
  return q;
}

// branch-free border handling (mode is wave-uniform): returns the source index and clears `ok` for a
// zero-padded tap that falls outside [0, n)
__device__ __forceinline__ int border_index(int x, int n, int mode, bool& ok) {
  const bool inb = (unsigned)x < (unsigned)n;
  int xr = x < 0 ? -x : x;
  xr = xr >= n ? 2 * n - 2 - xr : xr;
  int xc = x < 0 ? 0 : x;
  xc = xc >= n ? n - 1 : xc;
  ok = ok & (inb | (mode != GS_BORDER_ZERO));
  return mode == GS_BORDER_REFLECT ? xr : (mode == GS_BORDER_REPLICATE ? xc : x);
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == GS_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == GS_ACT_LRELU) return v > 0.f ? v : v * slope;
  if (act == GS_ACT_TANH) return tanhf(v);
  return v;
}
// derivative expressed through the activation OUTPUT o (valid for relu / lrelu with slope>0 / tanh)
__device__ __forceinline__ float act_grad_from_out(float o, int act, float slope) {
  if (act == GS_ACT_RELU) return o > 0.f ? 1.f : 0.f;
  if (act == GS_ACT_LRELU) return o > 0.f ? 1.f : slope;
  if (act == GS_ACT_TANH) return 1.f - o * o;
  return 1.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// kernel-selection switches (api.hip; set through gs_set_option, never read from the environment by the library)
enum GsOpt {
  GS_OPT_SPLITK, GS_OPT_SPLITK_MAX_BLOCKS, GS_OPT_SPLITK_TARGET, GS_OPT_HCONV, GS_OPT_HCONV_WIDE, GS_OPT_HCONVW_WAVES,
  GS_OPT_HWGRAD, GS_OPT_HWGRAD_WIDE, GS_OPT_HWGRAD_PLANES, GS_OPT_NORM_BWD_PPB, GS_OPT_NORM_APPLY_UNROLL,
  GS_OPT_COUNT
};
int gs_opt(int id);

// host-side error plumbing shared by the launchers
void gs_set_error(const char* fmt, ...);
const void* gs_zero_page();
#define GS_CHECK_HIP(x)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      gs_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)
#define GS_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      gs_set_error(__VA_ARGS__);   \
      return 2;                    \
    }                              \
  } while (0)
