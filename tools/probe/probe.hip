// Hardware-semantics probe for gfx950 (MI355X): run once on the GPU box to pin the
// lane layouts the kernels in ganslate_amd/csrc rely on (MFMA fragments, LDS-DMA image,
// ds_read_b64_tr_b16 transpose read). Prints PASS/FAIL lines plus raw dumps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void*)(p))

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// A: [16][32] row-major bf16, Bt: [16][32] (n-major, k contiguous), C: [16][16] f32
__global__ void mfma16(const uint16_t* A, const uint16_t* Bt, float* C) {
  int l = threadIdx.x;
  bf16x8 a = *(const bf16x8*)(A + (l & 15) * 32 + (l >> 4) * 8);
  bf16x8 b = *(const bf16x8*)(Bt + (l & 15) * 32 + (l >> 4) * 8);
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// A: [32][16], Bt: [32][16], C: [32][32]
__global__ void mfma32(const uint16_t* A, const uint16_t* Bt, float* C) {
  int l = threadIdx.x;
  bf16x8 a = *(const bf16x8*)(A + (l & 31) * 16 + (l >> 5) * 8);
  bf16x8 b = *(const bf16x8*)(Bt + (l & 31) * 16 + (l >> 5) * 8);
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

// LDS-DMA: 256 threads, 4 wave-instructions each moving 1 KiB. Source chunk is XOR-swizzled.
__global__ void glds_probe(const char* src, char* dump) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int t = threadIdx.x, w = t >> 6, l = t & 63;
  for (int i = 0; i < 4; ++i) {
    int inst = w * 4 + i;            // 16 instructions of 1 KiB = 16 KiB tile: 128 rows x 128 B
    int row = inst * 8 + (l >> 3);
    int slot = l & 7;
    int chunk = slot ^ (row & 7);
    const char* g = src + row * 128 + chunk * 16;
    __builtin_amdgcn_global_load_lds(GLBP(g), LDSP(smem + inst * 1024), 16, 0, 0);
  }
  __syncthreads();
  for (int i = t; i < 16384 / 16; i += 256) ((uint4*)dump)[i] = ((uint4*)smem)[i];
}

// transpose read: LDS holds int16 values = element index. mode 0: addr = lane*8.
// mode 1: per 16-lane group g a [4][16] block with row stride 64 elements (128 B):
//         lane l' -> row (l'&15)>>2, cols 4*(l'&3) ; block base g*16 elements (column offset).
__global__ void tr_probe(int mode, short* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  short* s = (short*)smem;
  int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) s[i] = (short)i;
  __syncthreads();
  int addr;
  if (mode == 0) addr = l * 8;
  else { int g = l >> 4, q = l & 15; addr = (((q >> 2) * 64) + g * 16 + (q & 3) * 4) * 2; }
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + addr));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device=%s arch=%s CUs=%d clock=%d MHz memclk=%d mem=%.1f GB lds/block=%zu regs/block=%d l2=%d\n",
         p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000, p.memoryClockRate / 1000,
         p.totalGlobalMem / 1e9, p.sharedMemPerBlock, p.regsPerBlock, p.l2CacheSize);
  srand(1);
  { // mfma16
    std::vector<uint16_t> A(16 * 32), B(16 * 32); std::vector<float> C(256), R(256, 0.f);
    for (auto& x : A) x = f2bf((float)(rand() % 7 - 3));
    for (auto& x : B) x = f2bf((float)(rand() % 5 - 2));
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += bf2f(A[i * 32 + k]) * bf2f(B[j * 32 + k]); R[i * 16 + j] = s; }
    uint16_t *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    mfma16<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(C[i] - R[i]));
    printf("mfma_16x16x32_bf16 layout (A[l&15][8*(l>>4)+j], C[(l>>4)*4+r][l&15]): %s maxerr=%g\n", e == 0 ? "PASS" : "FAIL", e);
  }
  { // mfma32
    std::vector<uint16_t> A(32 * 16), B(32 * 16); std::vector<float> C(1024), R(1024, 0.f);
    for (auto& x : A) x = f2bf((float)(rand() % 7 - 3));
    for (auto& x : B) x = f2bf((float)(rand() % 5 - 2));
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += bf2f(A[i * 16 + k]) * bf2f(B[j * 16 + k]); R[i * 32 + j] = s; }
    uint16_t *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    mfma32<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
    double e = 0; for (int i = 0; i < 1024; ++i) e = fmax(e, fabs(C[i] - R[i]));
    printf("mfma_32x32x16_bf16 layout (A[l&31][8*(l>>5)+j], C[(r&3)+8*(r>>2)+4*(l>>5)][l&31]): %s maxerr=%g\n", e == 0 ? "PASS" : "FAIL", e);
  }
  { // glds
    std::vector<uint32_t> src(4096), dump(4096);
    for (int i = 0; i < 4096; ++i) src[i] = i;  // dword index
    char *dS, *dD; CK(hipMalloc(&dS, 16384)); CK(hipMalloc(&dD, 16384));
    CK(hipMemcpy(dS, src.data(), 16384, hipMemcpyHostToDevice));
    glds_probe<<<1, 256, 16384>>>(dS, dD); CK(hipMemcpy(dump.data(), dD, 16384, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int row = 0; row < 128; ++row) for (int slot = 0; slot < 8; ++slot) for (int d = 0; d < 4; ++d) {
      uint32_t got = dump[row * 32 + slot * 4 + d];
      uint32_t exp = row * 32 + (slot ^ (row & 7)) * 4 + d;
      if (got != exp) { if (bad < 8) printf("  glds mismatch row %d slot %d d %d got %u exp %u\n", row, slot, d, got, exp); ++bad; }
    }
    printf("global_load_lds x16 lane-linear dest + swizzled source: %s (%d bad)\n", bad == 0 ? "PASS" : "FAIL", bad);
  }
  for (int mode = 0; mode < 2; ++mode) { // tr read
    short* dO; CK(hipMalloc(&dO, 64 * 4 * 2)); std::vector<short> o(256);
    tr_probe<<<1, 64, 8192>>>(mode, dO); CK(hipMemcpy(o.data(), dO, 512, hipMemcpyDeviceToHost));
    printf("ds_read_b64_tr_b16 mode %d raw (lane: 4 values):\n", mode);
    for (int l = 0; l < 64; ++l) printf("  l%02d: %4d %4d %4d %4d%s", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3], (l % 4 == 3) ? "\n" : "");
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      int exp = (mode == 0) ? ((l & 15) + j * 16 + (l >> 4) * 64) : (j * 64 + (l >> 4) * 16 + (l & 15));
      if (o[l * 4 + j] != exp) ++bad;
    }
    printf("ds_read_b64_tr_b16 mode %d matches [4][16]-block column model: %s (%d bad)\n", mode, bad == 0 ? "PASS" : "FAIL", bad);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
