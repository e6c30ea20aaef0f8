// How does the LDS-DMA (global_load_lds_dwordx4) throughput of a CU scale with the bytes it keeps in flight?
// 256 workgroups x 16 waves (one per CU, 144 KB of LDS as a ring of S stages), every wave issues IPS 1-KB instructions per stage
// and waits with vmcnt so that D stages stay in flight; no compute. Source: rows of 128 B gathered 8 per instruction (the im2col
// pattern), from a footprint that is L2-resident (per-XCD working set << 4 MB), MALL-resident or HBM-sized.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/dma_depth.hip -o /tmp/dma_depth && /tmp/dma_depth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define GLB(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS(p) ((__attribute__((address_space(3))) void*)(p))

template <int IPS, int D>
__global__ __launch_bounds__(1024) void dma_kernel(const char* src, size_t footprint, int steps, int stage_bytes, int nstage,
                                                   unsigned* sink, size_t row_stride = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // this workgroup's stream: consecutive 128-B rows; an instruction of wave w covers rows (w * IPS + i) * 8 + lane / 8 of a stage
  const size_t wg_base = ((size_t)blockIdx.x * 7919u * 4096u) % footprint;
  size_t off = wg_base;
  int st = 0;
  auto issue = [&](int k) {
#pragma unroll
    for (int i = 0; i < IPS; ++i) {
      const size_t row = (size_t)k * (stage_bytes / 128) + (wave * IPS + i) * 8 + (lane >> 3);
      // row_stride != 0: the weight-pack pattern — a stage is 128 B of each of (stage_bytes / 128) rows that lie row_stride
      // bytes apart, consecutive stages walk along the rows
      const size_t a = row_stride ? (off + (size_t)((wave * IPS + i) * 8 + (lane >> 3)) * row_stride + (size_t)k * 128 + (lane & 7) * 16) % footprint
                                  : (off + row * 128 + (lane & 7) * 16) % footprint;
      __builtin_amdgcn_global_load_lds(GLB(src + a), LDS(smem + st * stage_bytes + (wave * IPS + i) * 1024), 16, 0, 0);
    }
    st = st + 1 == nstage ? 0 : st + 1;
  };
  for (int k = 0; k < D; ++k) issue(k);
  for (int k = D; k < steps; ++k) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * IPS) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(k);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(smem);
}

template <int IPS, int D>
void run(const char* name, const char* src, size_t footprint, int nstage, unsigned* sink, size_t row_stride = 0, int steps = 400) {
  const int stage_bytes = IPS * 16 * 1024;
  const int lds = nstage * stage_bytes;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<IPS, D>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((dma_kernel<IPS, D>), dim3(256), dim3(1024), lds, 0, src, footprint, steps, stage_bytes, nstage, sink,
                       row_stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * steps * stage_bytes;
  printf("%-10s stage %3d KB x %2d stages, %2d in flight (%3d KB / CU), row stride %6zu, %3d steps: %7.1f us  %6.2f TB/s  %5.1f GB/s per CU\n",
         name, stage_bytes / 1024, nstage, D, D * stage_bytes / 1024, row_stride, steps, ms * 1e3, bytes / ms / 1e9,
         bytes / ms / 1e6 / 256);
}

int main() {
  const size_t big = 1ull << 30;
  char* src;
  unsigned* sink;
  hipMalloc(&src, big);
  hipMemset(src, 1, big);
  hipMalloc(&sink, 4096);
  struct { const char* name; size_t fp; } fps[] = {{"L2 (8 MB)", 8u << 20}, {"MALL 96MB", 96u << 20}, {"HBM 1 GB", big}};
  for (auto& f : fps) {
    run<3, 2>(f.name, src, f.fp, 3, sink);      // 48 KB stages, 2 in flight: gconv_kernel<256,128,...,3>
    run<3, 1>(f.name, src, f.fp, 3, sink);
    run<1, 2>(f.name, src, f.fp, 9, sink);      // 16 KB stages
    run<1, 4>(f.name, src, f.fp, 9, sink);
    run<1, 8>(f.name, src, f.fp, 9, sink);
    run<2, 4>(f.name, src, f.fp, 4, sink);      // 32 KB x 4 in flight (needs a 5th stage: aliasing is harmless here)
  }
  // the split-K weight stream: 128 rows of a pack per workgroup (16 KB stages), 8 / 64 K-steps per workgroup, rows 8-32 KB apart
  for (size_t stride : {(size_t)0, (size_t)8192, (size_t)16384, (size_t)32768})
    for (int steps : {8, 64}) {
      run<1, 1>("HBM 1 GB", src, big, 9, sink, stride, steps);
      run<1, 3>("HBM 1 GB", src, big, 9, sink, stride, steps);
    }
  return 0;
}
