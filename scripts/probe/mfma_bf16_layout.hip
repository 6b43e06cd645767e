// Probe: operand layout of v_mfma_f32_32x32x16_bf16 on gfx950.
// Hypothesis: lane l holds A[row = l & 31][k = 8 * (l >> 5) + j] and B[k = 8 * (l >> 5) + j][col = l & 31],
// j = 0..7 in the 8 bf16 of the operand; C/D as for 32x32x2_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // hipcc >= 6.x takes v8i16 / v8bf16
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8b;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned short f2bf(float x) { return (unsigned short)(__float_as_uint(x) >> 16); }  // exact for small ints

__global__ void probe(const float* A, const float* B, float* C) {   // A[32][16], B[16][32], C[32][32]
  const int l = threadIdx.x;
  bf16x8b a, b;
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * (l >> 5) + j;
    unsigned short ua = f2bf(A[(l & 31) * 16 + k]), ub = f2bf(B[k * 32 + (l & 31)]);
    a[j] = __builtin_bit_cast(__bf16, ua);
    b[j] = __builtin_bit_cast(__bf16, ub);
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    C[row * 32 + col] = acc[r];
  }
}

int main() {
  float hA[32 * 16], hB[16 * 32], hC[32 * 32], ref[32 * 32];
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (float)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (float)((k * 5 + j * 2 + (k * j) % 3) % 13 - 6);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
  float *dA, *dB, *dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; ++i) if (hC[i] != ref[i]) ++bad;
  printf("mfma_f32_32x32x16_bf16 layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
  return bad != 0;
}
