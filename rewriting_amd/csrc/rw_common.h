// Shared helpers for the gfx950 kernels of librewriting_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rewriting_hip.h"

#define RW_WAVE 64

#define RW_CHECK_ARG(cond) do { if (!(cond)) return RW_ERR_BAD_ARGUMENT; } while (0)
#define RW_LAUNCH_RESULT() ((int)hipGetLastError())

static inline hipStream_t rw_s(rw_stream_t s) { return (hipStream_t)s; }

static inline int64_t rw_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Compute units of the CURRENT device (queried once per device; 256 on a whole MI355X, fewer on a partitioned one): the
// workgroup count of the persistent kernels, which own a CU each.
static inline int rw_cu_count(void) {
  static int cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cached[dev];
}

// Grid for a memory-bound grid-stride kernel: enough blocks to fill 256 CUs x 8, no more.
static inline int rw_stream_grid(int64_t work_items, int block) {
  int64_t g = rw_cdiv(work_items, block);
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float rw_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float rw_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Block-wide sum for 256-thread blocks (4 waves); result valid in every thread.
__device__ __forceinline__ float rw_block_sum_256(float v, float* lds4) {
  v = rw_wave_sum(v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[wave] = v;
  __syncthreads();
  return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

typedef float rw_f32x16 __attribute__((ext_vector_type(16)));
typedef float rw_f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// Bounds (max |map|) handed from a launch to the next one -- see "a BOUND on a map" in include/rewriting_hip.h.
// Round 4 kept them in 4-byte device scalars: zeroed by a memset, raised by the producer's workgroups with (filtered,
// system-scope) atomics, read by the consumer with a system-scope load.  Occasionally a consumer read a stale value
// (images 0.01 - 0.05 off; GPUTEST_r04).  Nothing of that mechanism is left: a producer's wave (or workgroup) stores its
// maximum PLAINLY into ITS OWN slot, rw_bound_finish() reduces the slots of the launch into RW_BOUND_LANES floats with
// one small launch, and a consumer's wave loads those floats with an ordinary per-lane vector load.  No location is
// written by more than one workgroup, nothing is zeroed first, every slot that is read was written by the launch in
// front: what orders it is what orders every feature map -- the launch boundary.
// ---------------------------------------------------------------------------------------
// slots a producer may use for a result of n floats (rw_bound_floats(n) - RW_BOUND_LANES)
static inline int64_t rw_bound_slot_capacity(int64_t n_elems) { return 2048 + n_elems / 1024 + 1; }

// host: bound[0 .. RW_BOUND_LANES) <- the maxima of bound[RW_BOUND_LANES .. RW_BOUND_LANES + nslots)   (rw_bound.hip)
int rw_bound_finish(float* bound, int64_t nslots, hipStream_t stream);

// consumer: the bound (uniform over the wave).  ALL 64 lanes of the wave must be active.
__device__ __forceinline__ float rw_bound_load(const float* __restrict__ bound) {
  return rw_wave_max(bound[threadIdx.x & 63]);
}

// producer, one slot per WAVE: slot = workgroup * waves per workgroup + wave.  v = the wave's maximum (any lane's copy
// of the reduced value); every wave of the launch must call it exactly once.
__device__ __forceinline__ void rw_bound_store_wave(float* __restrict__ bound, float v) {
  if ((threadIdx.x & 63) == 0)
    bound[RW_BOUND_LANES + (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = v;
}

// producer, one slot per WORKGROUP of 256 threads: v = this thread's maximum; every thread of the workgroup calls it.
__device__ __forceinline__ void rw_bound_store_block_256(float* __restrict__ bound, float v, float* lds4) {
  v = rw_wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) bound[RW_BOUND_LANES + blockIdx.x] = fmaxf(fmaxf(lds4[0], lds4[1]), fmaxf(lds4[2], lds4[3]));
}

// u_scale = 2^(15 - e), max |U| < 2^e  (rw_split_weight_scale in the header; one definition for host and tests)
static inline float rw_weight_scale_of(float u_absmax) {
  union { float f; unsigned u; } b;
  b.f = u_absmax;
  int eu = (int)((b.u >> 23) & 0xff) - 126;
  if ((b.u & 0x7fffffffu) == 0u) eu = 15;
  eu = eu < -100 ? -100 : (eu > 100 ? 100 : eu);
  b.u = (unsigned)(127 + 15 - eu) << 23;
  return b.f;
}
