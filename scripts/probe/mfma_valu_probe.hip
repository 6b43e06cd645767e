// Do fp32 MFMA and vector-ALU instructions overlap on gfx950?  k independent VALU instructions between consecutive
// v_mfma_f32_16x16x4_f32 (two waves per SIMD).  Answer: no -- ~2.7 cycles per v_fma_f32, ~6 per v_pk_fma_f32.
// build: hipcc --offload-arch=gfx950 -O3 scripts/probe/mfma_valu_probe.hip -o scripts/probe/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int K, int PK>
__global__ void __launch_bounds__(256, 2) kern(float* out, int iters, float a, float b) {
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float x[8]; f2 y[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = f2{x[i], x[i] + 1.f}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (PK) y[(m + k) & 7] = y[(m + k) & 7] * f2{a, a} + f2{b, b};
        else x[(m + k) & 7] = x[(m + k) & 7] * a + b;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i] + y[i][0] + y[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K, int PK> void run(float* d, const char* name) {
  const int iters = 20000, blocks = 512;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL((kern<K, PK>), dim3(blocks), dim3(256), 0, 0, d, 100, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL((kern<K, PK>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  // per SIMD: 2 waves x iters x 8 MFMA; cycles per MFMA-pair slot
  double mfma_per_simd = 2.0 * iters * 8;
  double ns_per_mfma = ms * 1e6 / mfma_per_simd;
  printf("%s K=%d: %.3f ms, %.2f ns per MFMA per SIMD (32 cycles @2.4GHz = 13.3 ns), TFLOP/s %.1f\n", name, K, ms, ns_per_mfma,
         512.0 * 4 * iters * 8 * 2048 / (ms * 1e-3) / 1e12);
}
int main() {
  float* d; hipMalloc(&d, 512 * 256 * 4);
  run<0, 0>(d, "scalar"); run<2, 0>(d, "scalar"); run<4, 0>(d, "scalar"); run<6, 0>(d, "scalar"); run<8, 0>(d, "scalar"); run<12, 0>(d, "scalar");
  run<2, 1>(d, "packed"); run<4, 1>(d, "packed"); run<6, 1>(d, "packed"); run<8, 1>(d, "packed");
  return 0;
}
