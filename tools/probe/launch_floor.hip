// What does a launch cost before it does anything? Empty kernels in the launch shapes of the conv kernels (threads per workgroup,
// dynamic LDS, kernarg bytes), timed back to back on one stream with HIP events (100 launches each).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/launch_floor.hip -o tools/probe/launch_floor.bin
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { char pad[3400]; int* sink; };
__global__ void empty_kernel(Big b) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) b.sink[0] = smem[0];
}
__global__ void touch_kernel(Big b) {      // every wave reads one kernarg word at a dynamic offset and writes LDS once
  extern __shared__ char smem[];
  smem[threadIdx.x] = b.pad[(blockIdx.x * 64) % 3400];
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) b.sink[0] = smem[5];
}
template <typename K>
void run(const char* name, K kern, int grid, int threads, int lds) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  Big b{};
  (void)hipMalloc(&b.sink, 64);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, b);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-6s grid %5d x %4d threads, %3d KB LDS: %6.2f us per launch (back to back)\n", name, grid, threads, lds / 1024, ms * 10);
}
int main() {
  for (int lds : {0, 64 * 1024, 128 * 1024, 150 * 1024})
    for (int threads : {256, 512, 1024}) {
      run("empty", empty_kernel, 256, threads, lds);
      run("touch", touch_kernel, 256, threads, lds);
    }
  run("empty", empty_kernel, 512, 1024, 64 * 1024);
  run("empty", empty_kernel, 2048, 1024, 64 * 1024);
  run("empty", empty_kernel, 64, 1024, 128 * 1024);
  return 0;
}
