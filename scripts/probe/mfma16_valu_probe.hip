// Round 4 probe: what stands beside the 16-bit matrix pipe on gfx950?
//   * issue rate of v_mfma_f32_16x16x32_f16 / v_mfma_f32_16x16x16_f16 / v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x4_f32
//     with one and two waves per SIMD,
//   * what K independent VALU instructions of one kind between consecutive MFMAs cost (v_fma_f32, v_pk_fma_f32,
//     v_cvt_pk_f16_f32, v_fma_mix_f32, v_mov_b32, v_pk_mul_f32, v_pk_add_f32) -- i.e. whether the vector ALU overlaps
//     THIS pipe (round 2's probe: it does not overlap the fp32 MFMAs),
//   * the VALU kinds alone (no MFMA),
//   * LDS reads (ds_read_b64 / b128 per MFMA) beside the MFMAs.
// Prints one line per configuration: ns per MFMA slot and SIMD, and the same in cycles at the clock measured by
// s_memrealtime-free arithmetic (a dependent v_add chain of known length calibrates cycles per ns).
// build: hipcc --offload-arch=gfx950 -O3 scripts/probe/mfma16_valu_probe.hip -o scripts/probe/mfma16_valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

enum { MF_NONE = 0, MF_16x32_F16 = 1, MF_16x16_F16 = 2, MF_F32 = 3, MF_32x16_F16 = 4 };
enum { VT_FMA = 1, VT_PKFMA = 2, VT_CVTPK = 3, VT_FMAMIX = 4, VT_MOV = 5, VT_PKMUL = 6, VT_PKADD = 7 };

template <int VT> __device__ __forceinline__ void valu(float& x, double& y, float a, double b2) {
  if (VT == VT_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(a));
  if (VT == VT_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y) : "v"(b2));
  if (VT == VT_CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(a));
  if (VT == VT_FMAMIX) asm volatile("v_fma_mix_f32 %0, -%0, 1.0, %1 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(a));
  if (VT == VT_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(a));
  if (VT == VT_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y) : "v"(b2));
  if (VT == VT_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y) : "v"(b2));
}

template <int MF, int VT, int K, int LD, int WPS>
__global__ void __launch_bounds__(256, WPS) kern(float* out, int iters, float a, float b) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 0.5f;
  __syncthreads();
  f4 acc[8];
  f16v acc32[2];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
  float x[8]; double y[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = x[i]; }
  h8 ha, hb; h4 ha4, hb4;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b - i); }
  for (int i = 0; i < 4; ++i) { ha4[i] = ha[i]; hb4[i] = hb[i]; }
  const double b2 = (double)b;
  const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + (threadIdx.x & 63) * 16;
  f4 ld4[4];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MF == MF_16x32_F16) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[m], 0, 0, 0);
      if (MF == MF_16x16_F16) acc[m] = __builtin_amdgcn_mfma_f32_16x16x16f16(ha4, hb4, acc[m], 0, 0, 0);
      if (MF == MF_F32) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
      if (MF == MF_32x16_F16) acc32[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc32[m & 1], 0, 0, 0);
      if (LD == 1) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*(double*)&ld4[m & 3]) : "v"(laddr), "i"(1024 * (m & 7)));
      if (LD == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld4[m & 3]) : "v"(laddr), "i"(1024 * (m & 7)));
#pragma unroll
      for (int k = 0; k < K; ++k) valu<VT>(x[(m + k) & 7], y[(m + k) & 7], a, b2);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (LD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i] + (float)y[i];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 16; ++j) s += acc32[i][j];
  if (LD) for (int i = 0; i < 4; ++i) s += ld4[i][0] + ld4[i][1] + ld4[i][2] + ld4[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// clock calibration: a dependent chain of N v_add_f32 takes N * (cycles per dependent VALU op); with one wave per
// SIMD that is the VALU latency (documented: 4 quad... measured here), reported only as ns so the reader can convert.
static const char* mf_name[] = {"none", "16x16x32_f16", "16x16x16_f16", "16x16x4_f32", "32x32x16_f16"};
static const char* vt_name[] = {"-", "v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_f16_f32", "v_fma_mix_f32", "v_mov_b32",
                                "v_pk_mul_f32", "v_pk_add_f32"};
static const char* ld_name[] = {"-", "ds_read_b64", "ds_read_b128"};

template <int MF, int VT, int K, int LD, int WPS> void run(float* d) {
  const int iters = 4000, blocks = 256 * WPS;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL((kern<MF, VT, K, LD, WPS>), dim3(blocks), dim3(256), 0, 0, d, 50, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(s);
    hipLaunchKernelGGL((kern<MF, VT, K, LD, WPS>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    if (ms < best) best = ms;
  }
  // per SIMD: WPS waves x iters x 8 slots
  const double slots = (double)WPS * iters * 8;
  const double ns = best * 1e6 / slots;
  printf("{\"mfma\": \"%s\", \"valu\": \"%s\", \"k\": %d, \"lds\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, "
         "\"ns_per_slot_per_simd\": %.3f, \"cycles_at_2p4\": %.1f}\n",
         mf_name[MF], vt_name[VT], K, ld_name[LD], WPS, best, ns, ns * 2.4);
  fflush(stdout);
}

template <int MF, int WPS> void sweep_valu(float* d) {
  run<MF, VT_FMA, 0, 0, WPS>(d);
  run<MF, VT_FMA, 2, 0, WPS>(d); run<MF, VT_FMA, 4, 0, WPS>(d); run<MF, VT_FMA, 8, 0, WPS>(d); run<MF, VT_FMA, 12, 0, WPS>(d);
  run<MF, VT_PKFMA, 2, 0, WPS>(d); run<MF, VT_PKFMA, 4, 0, WPS>(d); run<MF, VT_PKFMA, 8, 0, WPS>(d);
  run<MF, VT_CVTPK, 4, 0, WPS>(d); run<MF, VT_CVTPK, 8, 0, WPS>(d);
  run<MF, VT_FMAMIX, 4, 0, WPS>(d); run<MF, VT_FMAMIX, 8, 0, WPS>(d);
  run<MF, VT_MOV, 4, 0, WPS>(d); run<MF, VT_MOV, 8, 0, WPS>(d);
  run<MF, VT_PKMUL, 4, 0, WPS>(d); run<MF, VT_PKADD, 4, 0, WPS>(d);
}

int main() {
  float* d; hipMalloc(&d, 512 * 256 * 4);
  // the matrix instructions alone, one and two waves per SIMD
  run<MF_16x32_F16, VT_FMA, 0, 0, 1>(d); run<MF_16x32_F16, VT_FMA, 0, 0, 2>(d);
  run<MF_16x16_F16, VT_FMA, 0, 0, 1>(d); run<MF_16x16_F16, VT_FMA, 0, 0, 2>(d);
  run<MF_32x16_F16, VT_FMA, 0, 0, 1>(d); run<MF_32x16_F16, VT_FMA, 0, 0, 2>(d);
  run<MF_F32, VT_FMA, 0, 0, 2>(d);
  // VALU kinds alone (8 per slot)
  run<MF_NONE, VT_FMA, 8, 0, 1>(d); run<MF_NONE, VT_FMA, 8, 0, 2>(d);
  run<MF_NONE, VT_PKFMA, 8, 0, 2>(d); run<MF_NONE, VT_CVTPK, 8, 0, 2>(d); run<MF_NONE, VT_FMAMIX, 8, 0, 2>(d);
  run<MF_NONE, VT_MOV, 8, 0, 2>(d); run<MF_NONE, VT_PKMUL, 8, 0, 2>(d); run<MF_NONE, VT_PKADD, 8, 0, 2>(d);
  // VALU beside the 16-bit pipe
  sweep_valu<MF_16x32_F16, 2>(d);
  sweep_valu<MF_16x32_F16, 1>(d);
  sweep_valu<MF_16x16_F16, 2>(d);
  // fp32 MFMA for reference (round 2's result)
  run<MF_F32, VT_FMA, 4, 0, 2>(d); run<MF_F32, VT_PKFMA, 4, 0, 2>(d);
  // LDS reads beside the MFMAs
  run<MF_16x32_F16, VT_FMA, 0, 1, 2>(d); run<MF_16x32_F16, VT_FMA, 0, 2, 2>(d);
  run<MF_16x32_F16, VT_FMA, 4, 1, 2>(d); run<MF_16x32_F16, VT_FMA, 4, 2, 2>(d);
  run<MF_16x32_F16, VT_FMA, 8, 1, 2>(d); run<MF_16x32_F16, VT_FMA, 8, 2, 2>(d);
  run<MF_NONE, VT_FMA, 0, 1, 2>(d); run<MF_NONE, VT_FMA, 0, 2, 2>(d);
  return 0;
}
