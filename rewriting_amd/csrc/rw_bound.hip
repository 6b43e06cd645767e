// hipcc-flags: -fno-slp-vectorize -fno-vectorize
// (no packed fp32 math beside the convolutions: rw_ops.hip header, tests/test_build_checks.py)
// Bounds handed from launch to launch (rw_common.h, "a BOUND on a map" in include/rewriting_hip.h): the reduction of a
// producer's slots, the stand-alone measurement of a map, and the host-side helpers of the by-value weight scale.
#include "rw_common.h"

// 64 workgroups: workgroup g reduces the slots g*256 + t + 16384 k and stores lane g of the bound.  Every one of the
// RW_BOUND_LANES floats is written on every call (0 where a workgroup had no slot: the maxima are of absolute values).
__global__ void __launch_bounds__(256) bound_reduce_kernel(float* __restrict__ bound, int64_t nslots) {
  const float* slots = bound + RW_BOUND_LANES;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nslots; i += (int64_t)RW_BOUND_LANES * 256)
    m = fmaxf(m, slots[i]);
  __shared__ float red[4];
  m = rw_wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) bound[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

int rw_bound_finish(float* bound, int64_t nslots, hipStream_t stream) {
  RW_CHECK_ARG(bound && nslots > 0);
  hipLaunchKernelGGL(bound_reduce_kernel, dim3(RW_BOUND_LANES), dim3(256), 0, stream, bound, nslots);
  return RW_LAUNCH_RESULT();
}

extern "C" long long rw_bound_floats(long long n_elems) {
  if (n_elems < 0) return -1;
  return RW_BOUND_LANES + rw_bound_slot_capacity(n_elems);
}

extern "C" float rw_split_weight_scale(float u_absmax) { return rw_weight_scale_of(u_absmax); }

// max |x| over n floats: one slot per workgroup (at most 2048), then the reduction
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ bound) {
  float m = 0.f;
  const int64_t n4 = n >> 2;
  const rw_f32x4* x4 = reinterpret_cast<const rw_f32x4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const rw_f32x4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
  __shared__ float red[4];
  rw_bound_store_block_256(bound, m, red);
}

extern "C" int rw_absmax_f32(const float* x, long long n, float* out, rw_stream_t stream) {
  RW_CHECK_ARG(x && out && n > 0);
  if (((size_t)x & 15) != 0) return RW_ERR_UNSUPPORTED;
  int grid = (int)((n / 4 + 255) / 256);
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, rw_s(stream), x, (int64_t)n, out);
  const int rc = RW_LAUNCH_RESULT();
  if (rc) return rc;
  return rw_bound_finish(out, grid, rw_s(stream));
}
