// Shared device helpers for the gfx950 kernels of libganslate_hip.so.
// Wave = 64 lanes everywhere; fragment layouts pinned by tools/probe/probe.hip (profiles/r01_hw_probe.txt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
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

// ---- LDS fragment reads the compiler does not count --------------------------------------------------------------
// With an LDS-DMA (global_load_lds) anywhere in a loop, hipcc (ROCm 7.2) no longer emits counted lgkmcnt(N) waits for
// ds_read results: every consumer waits lgkmcnt(0), i.e. also for the reads issued AFTER the ones it needs (the
// software-pipelined "issue the next fragments, then run the MFMAs on the current ones" degenerates into read - wait -
// MFMA; verified on a 20-line kernel: lgkmcnt(6)/(5)/(4) ladders without the DMA, lgkmcnt(0) with it). Reads issued
// through these statements are invisible to that bookkeeping; the kernel waits for them itself with gs_lgkm_wait<N>(...)
// naming every destination ("+v": no consumer can be scheduled above the wait, cdna_hip_programming.md §5.7 form (ii)).
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
template <int OFF>
__device__ __forceinline__ void lds_read128(bf16x8& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void gs_lgkm_wait(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d, bf16x8& e, bf16x8& f) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void gs_lgkm_wait(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d, bf16x8& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void gs_lgkm_wait(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d, bf16x8& e, bf16x8& f,
                                             bf16x8& g, bf16x8& h) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "i"(N) : "memory");
}

// wait-only statement + per-fragment register fences (any number of fragments): nothing that consumes x can be scheduled
// above reg_fence(x), and the fences follow the wait in program order (all are asm volatile)
template <int N>
__device__ __forceinline__ void gs_lgkm_wait_only() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ void reg_fence(bf16x8& x) { asm volatile("" : "+v"(x)); }

// Workgroup barrier for data exchanged through LDS that does NOT drain the vector-memory counter: __syncthreads() carries
// a release fence, and with an LDS-DMA (or, in a persistent kernel, the previous tile's output stores) in flight that fence
// waits vmcnt(0). This one waits for this wave's LDS operations only; the "memory" clobbers keep the compiler from moving
// LDS accesses across it.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// 64-bit transpose read (pixel-major operands of the weight-gradient kernels): 4 bf16 of one k-column per lane
template <int OFF>
__device__ __forceinline__ void lds_read64_tr(uint2& v, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void gs_lgkm_wait64(uint2& a, uint2& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void gs_lgkm_wait64(uint2& a, uint2& b, uint2& c, uint2& d, uint2& e, uint2& f) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "i"(N) : "memory");
}
// compile-time loop: f(std::integral_constant<int, I>{}) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
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
// The same with tanh as a CALL: inlined into an unrolled 64-128 element conv epilogue, tanhf's ~40 instructions per element
// (skipped at run time for the other activations, but fetched around and allocated for) made hstrip's tile code longer than
// the instruction cache and cost its kernels up to 127 registers (profiles/r04_hstrip_forms.txt).
static __device__ __noinline__ float gs_tanh_call(float v) { return tanhf(v); }
__device__ __forceinline__ float apply_act_small(float v, int act, float slope) {
  if (act == GS_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == GS_ACT_LRELU) return v > 0.f ? v : v * slope;
  if (act == GS_ACT_TANH) return gs_tanh_call(v);
  return v;
}
// derivative expressed through the activation OUTPUT o (valid for relu / lrelu with slope>0 / tanh)
__device__ __forceinline__ float act_grad_from_out(float o, int act, float slope) {
  if (act == GS_ACT_RELU) return o > 0.f ? 1.f : 0.f;
  if (act == GS_ACT_LRELU) return o > 0.f ? 1.f : slope;
  if (act == GS_ACT_TANH) return 1.f - o * o;
  return 1.f;
}

// dy of the InstanceNorm backward, dy = rstd * (ghat - mean ghat - yhat * mean(ghat * yhat)) with ghat = g * act'(yhat), from the
// pieces both of its homes hold — norm.hip's apply pass by channel group and hconvw.hip's in-launch apply (gs_gconv_ring_apply)
// — with explicit roundings: left to the compiler the two sites contract it differently (an fma here, a select there), and the
// one-launch form must reproduce the two-launch form bit for bit
__device__ __forceinline__ float inorm_dy(float g, float yh, float m, float s1, float s2, float rs) {
  const float gh = __fmul_rn(g, m);
  return __fmul_rn(rs, __fsub_rn(__fsub_rn(gh, s1), __fmul_rn(yh, s2)));
}

// ---- lane reductions on the DPP path -----------------------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32: an LDS instruction with ~100 cycles of latency per step. The statistics
// epilogue of the conv kernels ran 64 of them per wave in dependent chains of four (4.5 us of a 44 us launch, measured
// by leaving the statistics out). The same sums through DPP modifiers are plain VALU adds (v_add_f32_dpp).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), result in all of them
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);    // row_half_mirror: quads 0<->1, 2<->3 (values are quad-uniform by now)
  v += dpp_mov<0x140>(v);    // row_mirror: halves of the row
  return v;
}
// v[i] + v[i ^ STEP] for lanes whose values are NOT uniform below STEP (lane i keeps its low bits): STEP = 4, 8 inside a
// row via rotations (ror:4 then ror:8 sums the 4 lanes with equal i mod 4; ror:8 alone the 2 with equal i mod 8)
__device__ __forceinline__ float row_sum_stride4(float v) { v += dpp_mov<0x124>(v); v += dpp_mov<0x128>(v); return v; }
__device__ __forceinline__ float row_sum_stride8(float v) { v += dpp_mov<0x128>(v); return v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- Adam (optim.hip; wgrad.hip's fused epilogue) ----------------------------------------------------------------------
// One element of the update; the same expression for the scalar tail and the four lanes of a 16-byte access.
__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float b1, float b2, float eps,
                                         float bc2_sqrt, float step_size, float gscale, int zero_grad) {
  const float gi = g * gscale;
  const float mi = m + (gi - m) * (1.f - b1);                  // torch: exp_avg.lerp_(grad, 1-beta1)
  const float vi = v * b2 + (1.f - b2) * gi * gi;              // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p = p - step_size * (mi / denom);
  m = mi;
  v = vi;
  if (zero_grad) g = 0.f;
}


// kernel-selection switches (api.hip; set through gs_set_option, never read from the environment by the library)
enum GsOpt {
  GS_OPT_SPLITK, GS_OPT_SPLITK_MAX_BLOCKS, GS_OPT_SPLITK_TARGET, GS_OPT_HCONV, GS_OPT_HCONV_WIDE,
  GS_OPT_HWGRAD, GS_OPT_HWGRAD_WIDE, GS_OPT_HWGRAD_PLANES, GS_OPT_NORM_BWD_PPB, GS_OPT_NORM_APPLY_UNROLL,
  GS_OPT_GCONV_TILE288, GS_OPT_GCONV_MULTI, GS_OPT_HCONVW_RING, GS_OPT_HCONVT, GS_OPT_HSTRIP, GS_OPT_WFOLD_ROWS, GS_OPT_HWGRAD_FT, GS_OPT_GCONV_BIG, GS_OPT_HCONV_BOX8, GS_OPT_HCONVW_PERSIST, GS_OPT_HSTRIP_REGS, GS_OPT_GCONV_TWIN, GS_OPT_WGRAD_TWIN, GS_OPT_GCONV_SMALLK, GS_OPT_GCONV_PERSIST, GS_OPT_HCONVT_PERSIST, GS_OPT_RING_APPLY, GS_OPT_NORM_XCD, GS_OPT_WGRAD_ROWS, GS_OPT_SPLITK_MULTI, GS_OPT_SPLITK_RING, GS_OPT_GCONV_RING4, GS_OPT_RING_DBG, GS_OPT_HCONV5, GS_OPT_HCONV5_SEG, GS_OPT_HWGRAD2, GS_OPT_HCONV2, GS_OPT_PWISE, GS_OPT_ADAM_BLOCKS,
  GS_OPT_COUNT
};
int gs_opt(int id);

// host-side error plumbing shared by the launchers
void gs_set_error(const char* fmt, ...);
const void* gs_zero_page();
void* gs_dump_page();          // 256 writable bytes nobody reads: destination of masked store lanes (pconv.hip)
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
