// Device-side training-patch path of the 3-D volume datasets: crop a patch out of a volume that is
// RESIDENT in HBM and normalise it the way the reference's dataset workers do on the host,
//     patch = volume[z:z+d, y:y+h, x:x+w];  z_score_normalize(patch, scale_to_range=(lo, hi))
// (projects/brats_mri_sequence_translation/datasets/train_dataset.py:83-86, ganslate/data/utils/normalization.py:18-30):
//     t = (v - mean) / std            mean, UNBIASED std of the patch
//     out = (hi - lo) * (t - min t) / (max t - min t) + lo
// A BraTS volume is 36 MB and a training set a few hundred of them: with 288 GB of HBM the volumes are uploaded once and
// only the patch coordinates (data/utils/stochastic_focal_patching.py) travel per iteration.
// Three launches, no host round trip: per-block partial sums (double sum / sum of squares, min, max; fixed order), one
// block that finishes them into {mean, std, min t, max t} — min t = (min v - mean) / std exactly, the map is monotone —
// and the apply pass, which evaluates the reference's expression operation for operation in fp32. HBM-bound, 4 B in +
// 4 B out per voxel twice over an 8 MB patch: not a kernel worth more than coalesced rows.
#include "common.hpp"

namespace {
constexpr int NB = 256;      // partial-sum blocks

struct PatchK {
  const void* vol;
  long long sz, sy;          // element strides of the volume's depth and row axes (x is dense)
  int z0, y0, x0, pd, ph, pw;
  long long n;
};

template <typename T>
__device__ __forceinline__ float voxel(const PatchK& k, long long i) {
  const int x = (int)(i % k.pw);
  const long long r = i / k.pw;
  const int y = (int)(r % k.ph), z = (int)(r / k.ph);
  return (float)static_cast<const T*>(k.vol)[(k.z0 + z) * k.sz + (k.y0 + y) * k.sy + (k.x0 + x)];
}

template <typename T>
__global__ __launch_bounds__(256) void patch_stats_kernel(const PatchK k, double* partial) {
  __shared__ double s1[256], s2[256];
  __shared__ float mn[256], mx[256];
  double a = 0.0, q = 0.0;
  float lo = INFINITY, hi = -INFINITY;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < k.n; i += (long long)NB * 256) {
    const float v = voxel<T>(k, i);
    a += v;
    q += (double)v * v;
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
    if (v != v) { lo = v; hi = v; }          // NaN voxels poison min / max like torch.min / torch.max do
  }
  s1[threadIdx.x] = a; s2[threadIdx.x] = q; mn[threadIdx.x] = lo; mx[threadIdx.x] = hi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      s1[threadIdx.x] += s1[threadIdx.x + o];
      s2[threadIdx.x] += s2[threadIdx.x + o];
      const float l2 = mn[threadIdx.x + o], h2 = mx[threadIdx.x + o];
      mn[threadIdx.x] = (l2 != l2 || mn[threadIdx.x] != mn[threadIdx.x]) ? NAN : fminf(mn[threadIdx.x], l2);
      mx[threadIdx.x] = (h2 != h2 || mx[threadIdx.x] != mx[threadIdx.x]) ? NAN : fmaxf(mx[threadIdx.x], h2);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 4 + 0] = s1[0];
    partial[blockIdx.x * 4 + 1] = s2[0];
    partial[blockIdx.x * 4 + 2] = mn[0];
    partial[blockIdx.x * 4 + 3] = mx[0];
  }
}

// stats = {mean, std, min t, max t} as the reference's fp32 scalars
__global__ __launch_bounds__(256) void patch_finalize_kernel(const double* partial, long long n, float* stats) {
  __shared__ double s1[256], s2[256];
  __shared__ float mn[256], mx[256];
  const int t = threadIdx.x;
  s1[t] = partial[t * 4]; s2[t] = partial[t * 4 + 1]; mn[t] = (float)partial[t * 4 + 2]; mx[t] = (float)partial[t * 4 + 3];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
      s1[t] += s1[t + o];
      s2[t] += s2[t + o];
      mn[t] = (mn[t + o] != mn[t + o] || mn[t] != mn[t]) ? NAN : fminf(mn[t], mn[t + o]);
      mx[t] = (mx[t + o] != mx[t + o] || mx[t] != mx[t]) ? NAN : fmaxf(mx[t], mx[t + o]);
    }
    __syncthreads();
  }
  if (t == 0) {
    const double mean = s1[0] / (double)n;
    double var = (s2[0] - (double)n * mean * mean) / (double)(n - 1);      // n == 1: 0 / 0 = NaN, like torch.std
    if (var < 0.0) var = 0.0;
    const float m = (float)mean, sd = (float)sqrt(var);
    stats[0] = m;
    stats[1] = sd;
    stats[2] = (mn[0] - m) / sd;
    stats[3] = (mx[0] - m) / sd;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void patch_apply_kernel(const PatchK k, const float* stats, float lo, float hi, int rescale,
                                                          float* out) {
  const float m = stats[0], sd = stats[1], tmin = stats[2], tmax = stats[3];
  const float span = hi - lo, delta = tmax - tmin;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < k.n; i += (long long)gridDim.x * 256) {
    float t = (voxel<T>(k, i) - m) / sd;
    if (rescale) t = (span * (t - tmin) / delta) + lo;
    out[i] = t;
  }
}

template <typename T>
int launch(const PatchK& k, float lo, float hi, int rescale, float* out, float* scratch, hipStream_t st) {
  double* partial = reinterpret_cast<double*>(scratch);
  float* stats = scratch + NB * 4 * 2;
  hipLaunchKernelGGL((patch_stats_kernel<T>), dim3(NB), dim3(256), 0, st, k, partial);
  hipLaunchKernelGGL(patch_finalize_kernel, dim3(1), dim3(256), 0, st, partial, k.n, stats);
  const long long blocks = (k.n + 256 * 4 - 1) / (256 * 4);
  hipLaunchKernelGGL((patch_apply_kernel<T>), dim3((unsigned)(blocks < 1 ? 1 : (blocks > 65535 ? 65535 : blocks))), dim3(256),
                     0, st, k, stats, lo, hi, rescale, out);
  GS_CHECK_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int64_t gs_patch_zscore_ws_floats(void) { return NB * 4 * 2 + 8; }

extern "C" int gs_patch_zscore(const void* vol, int32_t dtype, int32_t D, int32_t H, int32_t W, const int32_t* start,
                               const int32_t* size, int32_t rescale, float lo, float hi, float* out, float* scratch,
                               void* stream) {
  GS_REQUIRE(vol && start && size && out && scratch, "gs_patch_zscore: null argument");
  GS_REQUIRE(dtype == GS_VOL_F32 || dtype == GS_VOL_I16, "gs_patch_zscore: dtype %d (GS_VOL_F32 / GS_VOL_I16)", dtype);
  GS_REQUIRE(size[0] > 0 && size[1] > 0 && size[2] > 0 && start[0] >= 0 && start[1] >= 0 && start[2] >= 0 &&
                 start[0] + size[0] <= D && start[1] + size[1] <= H && start[2] + size[2] <= W,
             "gs_patch_zscore: patch [%d+%d, %d+%d, %d+%d] outside the %d x %d x %d volume", start[0], size[0], start[1],
             size[1], start[2], size[2], D, H, W);
  GS_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 7) == 0, "gs_patch_zscore: scratch must be 8-byte aligned");
  PatchK k;
  k.vol = vol;
  k.sz = (long long)H * W; k.sy = W;
  k.z0 = start[0]; k.y0 = start[1]; k.x0 = start[2];
  k.pd = size[0]; k.ph = size[1]; k.pw = size[2];
  k.n = (long long)size[0] * size[1] * size[2];
  hipStream_t st = static_cast<hipStream_t>(stream);
  return dtype == GS_VOL_F32 ? launch<float>(k, lo, hi, rescale, out, scratch, st)
                             : launch<short>(k, lo, hi, rescale, out, scratch, st);
}
