// Energy per FLOP of the two bf16 MFMA shapes on gfx950, with and without LDS operand traffic: loops one form on every CU for
// a few seconds; tools/probe/mfma_power.py samples rocm-smi beside it.   usage: mfma_power <form> <seconds> <waves per CU> <duty %> <random operands 0|1>
//   form 0: v_mfma_f32_16x16x32_bf16, operands in registers         form 1: v_mfma_f32_32x32x16_bf16, operands in registers
//   form 2: 16x16x32 with a 64 x 64 wave tile's fragment reads (8 ds_read_b128 per 16 MFMAs)
//   form 3: 32x32x16 with the same tile's reads (8 ds_read_b128 per 8 MFMAs of twice the work)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int FORM>
__global__ __launch_bounds__(1024) void burn(float* out, int iters, int rnd) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  const int lane = threadIdx.x & 63;
  // rnd != 0: operands with random signs and mantissas, magnitudes in [0.5, 2) — the switching activity of real activations;
  // rnd == 0: nearly constant operands (the multiplier arrays barely toggle)
  auto hash = [](unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; };
  auto rbf = [&](unsigned x) { const unsigned h = hash(x); return (short)((h & 0x80ff) | (0x3f00 + ((h >> 8) & 0x80))); };
  for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x)
    reinterpret_cast<int*>(lds)[i] = rnd ? (int)(((unsigned)(unsigned short)rbf(2 * i + blockIdx.x * 77777) << 16) | (unsigned short)rbf(2 * i + 1))
                                         : 0x3c003c00 + i;
  __syncthreads();
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 8; ++k) {
      a[i][k] = rnd ? rbf(threadIdx.x * 64 + i * 8 + k) : (short)(0x3c00 + lane + i);
      b[i][k] = rnd ? rbf(threadIdx.x * 64 + 32 + i * 8 + k + blockIdx.x * 4096) : (short)(0x3c10 + lane * 2 + i);
    }
  const char* base = lds + lane * 16;
  if (FORM == 0 || FORM == 2) {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      if (FORM == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = *reinterpret_cast<const bf16x8*>(base + ((it + i) & 15) * 1024);
          b[i] = *reinterpret_cast<const bf16x8*>(base + 16384 + ((it + i) & 15) * 1024);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
  } else {
    // 64 x 64 wave tile = 2 x 2 blocks of 32 x 32; a K-step of 32 = two MFMAs of K = 16 per block
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if (FORM == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = *reinterpret_cast<const bf16x8*>(base + ((it + i) & 15) * 1024);
          b[i] = *reinterpret_cast<const bf16x8*>(base + 16384 + ((it + i) & 15) * 1024);
        }
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk * 2 + i], b[kk * 2 + j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    if (s == 12345.678f) out[threadIdx.x] = s;
  }
}

int main(int argc, char** argv) {
  const int form = argc > 1 ? atoi(argv[1]) : 0;
  const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
  const int waves = argc > 3 ? atoi(argv[3]) : 8;
  const int duty = argc > 4 ? atoi(argv[4]) : 100;      // percent of wall time the kernel runs (idle gaps between launches)
  const int rnd = argc > 5 ? atoi(argv[5]) : 0;
  float* out;
  CK(hipMalloc(&out, 4096));
  const int iters = 20000;                     // x 16 MFMAs of 16x16x32 (or 8 of 32x32x16) = 64 x 64 x 32 MACs per iteration
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    dim3 g(256), b(waves * 64);
    if (form == 0) hipLaunchKernelGGL(burn<0>, g, b, 0, 0, out, iters, rnd);
    if (form == 1) hipLaunchKernelGGL(burn<1>, g, b, 0, 0, out, iters, rnd);
    if (form == 2) hipLaunchKernelGGL(burn<2>, g, b, 0, 0, out, iters, rnd);
    if (form == 3) hipLaunchKernelGGL(burn<3>, g, b, 0, 0, out, iters, rnd);
  };
  launch();
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  double busy_ms = 0;
  long n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    busy_ms += ms;
    ++n;
    if (duty < 100) {
      const auto w0 = std::chrono::steady_clock::now();
      const double idle = ms * 1e-3 * (100 - duty) / duty;
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < idle) {}
    }
  }
  const double flop = 2.0 * 64 * 64 * 32 * (double)iters * waves * 256;
  printf("form %d waves/CU %d duty %d%% %s: %ld launches, %.3f ms each, %.1f TFLOP/s while running\n", form, waves, duty, rnd ? "random operands" : "constant operands", n,
         busy_ms / n, flop / (busy_ms / n * 1e-3) / 1e12);
  return 0;
}
