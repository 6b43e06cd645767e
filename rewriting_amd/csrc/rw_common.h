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

// max into a device scalar that holds a NON-NEGATIVE float (bit patterns then order like the values).  The scalar is
// zeroed by a memset, raised by workgroups on all eight XCDs and read by the next launch: the operations carry system
// scope (sc1: performed at the memory side, not in one XCD's L2), and a workgroup whose value cannot raise it -- almost
// all of them -- leaves after one coherent load instead of queueing on the same address.
__device__ __forceinline__ void rw_atomic_max_nonneg(float* addr, float v) {
  unsigned* a = reinterpret_cast<unsigned*>(addr);
  const unsigned bits = __float_as_uint(v);
  if (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < bits)
    __hip_atomic_fetch_max(a, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
