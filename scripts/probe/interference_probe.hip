// Round 6 probe: does a kernel on one stream change the RESULT of a packed-fp32 kernel on another stream?
// (profiles/r05i, r05l: with rw_tconv.hip's kernel on the launching stream, to_rgb_kernel's
// v_pk_fma_f32 results on a second stream came back wrong in the low half of lanes 48..63.)
//
// Stand-alone: HIP runtime + the library's C ABI only (no torch, no Python).  Two streams; per (aggressor, victim) pair the
// victim runs alone once (its reference), then REPS times beside the aggressor (aggressor launched first on stream A, the
// victim right behind it on stream B, the aggressor once more behind that on stream A, both drained), and every overlapped result is compared with the reference BIT FOR BIT.
// One JSON line per pair: overlapped launches whose result differs, differing elements, the largest deviation, and where
// the differing elements sit (lane of the wave that wrote them, component of the lane's four pixels, output colour).
//
// Aggressors:  none | the library's rw_tconv_blur_f32 on the layer-17 shape (persistent form; RW_TCONV_TY=16: one workgroup
//   per CU) | rw_dconv3x3_f32 on the layer-16 shape (control: never seen to disturb) | synthetic kernels of 256 workgroups x
//   512 threads x 256 registers: an MFMA-only loop (v_mfma_f32_16x16x32_f16), the same with s_setprio 3, with LDS traffic
//   beside it, with global loads beside it, and a VALU-only loop (v_pk_fma_f32) of the same length.
// Victims: a ToRGB-shaped stream kernel (64 channels of a 512^2 map -> 3 colours, four pixels per lane, 16-byte loads)
//   whose inner product is written with v_pk_fma_f32 | v_pk_mul_f32 + v_pk_add_f32 | v_fma_f32 (inline asm: the instruction
//   is what is under test, not the compiler's choice).
//
// build (after the library):
//   hipcc --offload-arch=gfx950 -O3 -I include scripts/probe/interference_probe.hip -o scripts/probe/interference_probe \
//         -L rewriting_amd -lrewriting_hip -Wl,-rpath,'$ORIGIN/../../rewriting_amd'
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "rewriting_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
#define RW(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s:%d rw error %d (%s)\n", __FILE__, __LINE__, r_, rw_error_string(r_)); exit(3); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------ victims
enum { V_PKFMA = 0, V_PKMULADD = 1, V_FMA = 2, V_COMPILER = 3, V_COMPILER_DUP = 4, V_COUNT = 5 };
static const char* const victim_name[V_COUNT] = {"v_pk_fma_f32", "v_pk_mul_f32+v_pk_add_f32", "v_fma_f32",
                                                  "compiler-written (float4 accumulate; the SLP vectoriser packs it: v_pk_fma_f32 with op_sel)",
                                                  "compiler-written, weights stored as {w, w} pairs (v_pk_fma_f32 without op_sel)"};

template <int KIND>
__global__ void __launch_bounds__(256) victim_kernel(const float* __restrict__ x, const float* __restrict__ w3,
                                                      float* __restrict__ y, int in_ch, long long hw) {
  __shared__ float wm[3 * 64];
  __shared__ f2 wm2[3 * 64];
  const int b = blockIdx.y;
  for (int t = threadIdx.x; t < 3 * in_ch; t += 256) { wm[t] = w3[t] * (1.0f + 0.01f * b); wm2[t] = f2{wm[t], wm[t]}; }
  __syncthreads();
  const long long hw4 = hw >> 2;
  const float* xb = x + (long long)b * in_ch * hw;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < hw4; q += (long long)gridDim.x * 256) {
    f2 a[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c) a[c][0] = a[c][1] = f2{0.f, 0.f};
#pragma unroll 8
    for (int i = 0; i < in_ch; ++i) {
      const f4 v = reinterpret_cast<const f4*>(xb + (long long)i * hw)[q];
      const f2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float w = wm[c * in_ch + i];
        const f2 ww = {w, w};
        if (KIND == V_COMPILER_DUP) {              // the same arithmetic on ready {w, w} operands
          const f2 w2 = wm2[c * in_ch + i];
          a[c][0] = __builtin_elementwise_fma(lo, w2, a[c][0]);
          a[c][1] = __builtin_elementwise_fma(hi, w2, a[c][1]);
        } else if (KIND == V_COMPILER) {           // to_rgb_kernel's own statements (round 5's victim)
          a[c][0][0] += w * v[0]; a[c][0][1] += w * v[1]; a[c][1][0] += w * v[2]; a[c][1][1] += w * v[3];
        } else if (KIND == V_PKFMA) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[c][0]) : "v"(lo), "v"(ww));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[c][1]) : "v"(hi), "v"(ww));
        } else if (KIND == V_PKMULADD) {
          f2 t0, t1;
          asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(lo), "v"(ww));
          asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(hi), "v"(ww));
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[c][0]) : "v"(t0));
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[c][1]) : "v"(t1));
        } else {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[c][0][0]) : "v"(v[0]), "v"(w));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[c][0][1]) : "v"(v[1]), "v"(w));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[c][1][0]) : "v"(v[2]), "v"(w));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[c][1][1]) : "v"(v[3]), "v"(w));
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
      reinterpret_cast<f4*>(y)[((long long)b * 3 + c) * hw4 + q] = f4{a[c][0][0], a[c][0][1], a[c][1][0], a[c][1][1]};
  }
}

// ------------------------------------------------------------------ synthetic aggressors
enum { S_MFMA = 0, S_MFMA_PRIO = 1, S_MFMA_LDS = 2, S_MFMA_GLOBAL = 3, S_VALU = 4, S_LDS_ONLY = 5, S_GLOBAL_ONLY = 6, S_COUNT = 7 };
static const char* const synth_name[S_COUNT] = {"synthetic: MFMA loop", "synthetic: MFMA loop + s_setprio 3", "synthetic: MFMA + LDS reads",
                                                "synthetic: MFMA + global loads", "synthetic: v_pk_fma_f32 loop, no MFMA",
                                                "synthetic: LDS reads, no MFMA", "synthetic: global loads, no MFMA"};

template <int KIND>
__global__ void __launch_bounds__(512, 2) synth_kernel(float* out, const float* src, long long src_elems, int iters, float a, float b) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = i * 0.25f;
  __syncthreads();
  // 160 accumulator registers: with the operands the kernel sits near the 256 registers of the library's kernel
  f4 acc[40];
#pragma unroll
  for (int i = 0; i < 40; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  h8 ha, hb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + 0.01f * i + 0.001f * (threadIdx.x & 63)); hb[i] = (_Float16)(b - 0.01f * i); }
  f2 pk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pk[i] = f2{a + i, b - i};
  const f2 pkm = {1.0000001f, 0.9999999f};
  if (KIND == S_MFMA_PRIO) __builtin_amdgcn_s_setprio(3);
  f4 g = {0.f, 0.f, 0.f, 0.f};
  long long gi = ((long long)blockIdx.x * 512 + threadIdx.x) * 4;
  for (int it = 0; it < iters; ++it) {
    if (KIND == S_VALU) {
#pragma unroll
      for (int m = 0; m < 40; ++m) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[m & 7]) : "v"(pkm));
    } else {
#pragma unroll
      for (int m = 0; m < 40; ++m) {
        if (KIND != S_LDS_ONLY && KIND != S_GLOBAL_ONLY) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[m], 0, 0, 0);
        else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(acc[m][0]) : "v"(a));
        if ((KIND == S_MFMA_LDS || KIND == S_LDS_ONLY) && (m & 3) == 0) {
          const f4 l = *reinterpret_cast<const f4*>(&lds[((threadIdx.x * 4 + 64 * m + 16 * it) & 16380)]);
          ha[0] += (_Float16)(l[0] * 1e-9f);
        }
      }
      if (KIND == S_MFMA_GLOBAL || KIND == S_GLOBAL_ONLY) {
        g += *reinterpret_cast<const f4*>(src + (gi % (src_elems - 4)));
        gi += (long long)gridDim.x * 512 * 4;
      }
    }
  }
  float s = g[0] + g[1] + g[2] + g[3];
#pragma unroll
  for (int i = 0; i < 40; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += pk[i][0] + pk[i][1];
  if (s == 12345.678f) out[threadIdx.x] = s;      // never: keeps the loop
}

// ------------------------------------------------------------------ host
static unsigned lcg = 12345u;
static float rnd() { lcg = lcg * 1664525u + 1013904223u; return ((lcg >> 8) & 0xffff) / 32768.0f - 1.0f; }
static float* dev_random(long long n, float scale = 1.f, float offset = 0.f) {
  std::vector<float> h(n);
  for (long long i = 0; i < n; ++i) h[i] = offset + scale * rnd();
  float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, REPS = argc > 2 ? atoi(argv[2]) : 12;
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));

  // ---- victim data: (B, 64, 512, 512) -> (B, 3, 512, 512)
  const int vc = 64; const long long vhw = 512LL * 512;
  float* vx = dev_random((long long)B * vc * vhw);
  float* vw = dev_random(3 * vc);
  float *vy, *vref_d;
  const long long vout = (long long)B * 3 * vhw;
  CK(hipMalloc(&vy, vout * 4)); CK(hipMalloc(&vref_d, vout * 4));
  std::vector<float> ref(vout), got(vout);
  auto launch_victim = [&](int kind, hipStream_t s) {
    dim3 grid((unsigned)((256 * 8 + B - 1) / B), B);
    if (kind == V_PKFMA) hipLaunchKernelGGL(victim_kernel<V_PKFMA>, grid, dim3(256), 0, s, vx, vw, vy, vc, vhw);
    if (kind == V_PKMULADD) hipLaunchKernelGGL(victim_kernel<V_PKMULADD>, grid, dim3(256), 0, s, vx, vw, vy, vc, vhw);
    if (kind == V_FMA) hipLaunchKernelGGL(victim_kernel<V_FMA>, grid, dim3(256), 0, s, vx, vw, vy, vc, vhw);
    if (kind == V_COMPILER) hipLaunchKernelGGL(victim_kernel<V_COMPILER>, grid, dim3(256), 0, s, vx, vw, vy, vc, vhw);
    if (kind == V_COMPILER_DUP) hipLaunchKernelGGL(victim_kernel<V_COMPILER_DUP>, grid, dim3(256), 0, s, vx, vw, vy, vc, vhw);
    CK(hipGetLastError());
  };

  // ---- the library's kernels: layer 17 (64 -> 32 channels, 512^2 -> 1024^2) and layer 16 (64 -> 64 at 512^2)
  const int ci = 64, co = 32, h = 512, w = 512;
  float* x = dev_random((long long)B * ci * h * w);
  float* wt = dev_random((long long)co * ci * 9);
  float* wt1 = dev_random((long long)64 * ci * 9);
  float* style = dev_random((long long)B * ci, 0.3f, 1.0f);
  float* demod = dev_random((long long)B * 64, 0.2f, 1.0f);
  float* bias = dev_random(64);
  float* noise = dev_random((long long)B * 4 * h * w);
  float nw_h = 0.1f; float* nw; CK(hipMalloc(&nw, 4)); CK(hipMemcpy(nw, &nw_h, 4, hipMemcpyHostToDevice));
  float k4_h[16]; { const float k1[4] = {1, 3, 3, 1}; for (int a = 0; a < 4; ++a) for (int c = 0; c < 4; ++c) k4_h[a * 4 + c] = k1[a] * k1[c] / 64.f * 4.f; }
  float* k4; CK(hipMalloc(&k4, 64)); CK(hipMemcpy(k4, k4_h, 64, hipMemcpyHostToDevice));
  float *y17, *y16, *xb, *yb17, *yb16, *wb;
  CK(hipMalloc(&y17, (long long)B * co * 4 * h * w * 4));
  CK(hipMalloc(&y16, (long long)B * 64 * h * w * 4));
  CK(hipMalloc(&xb, rw_bound_floats((long long)B * ci * h * w) * 4));
  CK(hipMalloc(&yb17, rw_bound_floats((long long)B * co * 4 * h * w) * 4));
  CK(hipMalloc(&yb16, rw_bound_floats((long long)B * 64 * h * w) * 4));
  CK(hipMalloc(&wb, rw_bound_floats(0) * 4));
  RW(rw_absmax_f32(x, (long long)B * ci * h * w, xb, sa));
  auto pack = [&](const float* wsrc, int oc, float** wp, float* u_inv) {
    RW(rw_dconv_weight_absmax_f32(wsrc, oc, ci, wb, sa));
    CK(hipStreamSynchronize(sa));
    float bh[RW_BOUND_LANES]; CK(hipMemcpy(bh, wb, sizeof(bh), hipMemcpyDeviceToHost));
    float um = 0.f; for (int i = 0; i < RW_BOUND_LANES; ++i) um = fmaxf(um, bh[i]);
    const float us = rw_split_weight_scale(um);
    CK(hipMalloc(wp, rw_packed_dconv_weight_elems(oc, ci) * 4));
    RW(rw_pack_dconv_weight_f32(wsrc, *wp, oc, ci, us, sa));
    *u_inv = 1.0f / us;
  };
  float *wp17, *wp16; float ui17, ui16;
  pack(wt, co, &wp17, &ui17);
  pack(wt1, 64, &wp16, &ui16);
  CK(hipStreamSynchronize(sa));
  const float wsc = 1.0f / sqrtf((float)ci * 9);
  rw_conv_epilogue ep17 = {style, demod, noise, nw, bias, 1};
  rw_conv_epilogue ep16 = {style, demod, noise, nw, bias, 1};     // (the stride-1 layer reads B x h*w of the noise)
  float* synth_out; CK(hipMalloc(&synth_out, 4096));
  const long long src_elems = (long long)B * ci * h * w;

  enum { A_NONE = 0, A_TCONV_WS, A_TCONV_T16, A_DCONV, A_SYNTH0 };
  const int n_aggr = A_SYNTH0 + S_COUNT;
  auto aggressor_name = [&](int a) -> const char* {
    switch (a) {
      case A_NONE: return "none";
      case A_TCONV_WS: return "rw_tconv_blur_f32 layer 17, persistent form";
      case A_TCONV_T16: return "rw_tconv_blur_f32 layer 17, one workgroup per CU (RW_TCONV_TY=16)";
      case A_DCONV: return "rw_dconv3x3_f32 layer 16 (control)";
      default: return synth_name[a - A_SYNTH0];
    }
  };
  // length of the synthetic loops: ~ the library kernel's duration at this batch
  int synth_iters = 1000 * B;
  auto launch_aggressor = [&](int a, hipStream_t s) {
    if (a == A_NONE) return;
    if (a == A_TCONV_WS || a == A_TCONV_T16) {
      setenv("RW_TCONV_TY", a == A_TCONV_WS ? "0" : "16", 1);
      RW(rw_tconv_blur_f32(x, wp17, k4, y17, B, ci, co, h, w, wsc, &ep17, nullptr, ui17, xb, yb17, s));
      return;
    }
    if (a == A_DCONV) { RW(rw_dconv3x3_f32(x, wp16, y16, B, ci, 64, h, w, wsc, &ep16, ui16, xb, yb16, s)); return; }
    const int k = a - A_SYNTH0;
    const int it = k == S_VALU ? synth_iters * 4 : synth_iters;
    if (k == S_MFMA) hipLaunchKernelGGL(synth_kernel<S_MFMA>, dim3(256), dim3(512), 0, s, synth_out, x, src_elems, it, 0.5f, 0.25f);
    if (k == S_MFMA_PRIO) hipLaunchKernelGGL(synth_kernel<S_MFMA_PRIO>, dim3(256), dim3(512), 0, s, synth_out, x, src_elems, it, 0.5f, 0.25f);
    if (k == S_MFMA_LDS) hipLaunchKernelGGL(synth_kernel<S_MFMA_LDS>, dim3(256), dim3(512), 0, s, synth_out, x, src_elems, it, 0.5f, 0.25f);
    if (k == S_MFMA_GLOBAL) hipLaunchKernelGGL(synth_kernel<S_MFMA_GLOBAL>, dim3(256), dim3(512), 0, s, synth_out, x, src_elems, it, 0.5f, 0.25f);
    if (k == S_VALU) hipLaunchKernelGGL(synth_kernel<S_VALU>, dim3(256), dim3(512), 0, s, synth_out, x, src_elems, it, 0.5f, 0.25f);
    if (k == S_LDS_ONLY) hipLaunchKernelGGL(synth_kernel<S_LDS_ONLY>, dim3(256), dim3(512), 0, s, synth_out, x, src_elems, it, 0.5f, 0.25f);
    if (k == S_GLOBAL_ONLY) hipLaunchKernelGGL(synth_kernel<S_GLOBAL_ONLY>, dim3(256), dim3(512), 0, s, synth_out, x, src_elems, it, 0.5f, 0.25f);
    CK(hipGetLastError());
  };

  hipEvent_t e0, e1, v0, v1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&v0)); CK(hipEventCreate(&v1));
  const char* only_a = getenv("AGG"); const char* only_v = getenv("VIC");
  for (int v = 0; v < V_COUNT; ++v) {
    if (only_v && !strstr(victim_name[v], only_v)) continue;
    CK(hipMemsetAsync(vy, 0, vout * 4, sb));
    launch_victim(v, sb);
    CK(hipStreamSynchronize(sb));
    CK(hipMemcpy(ref.data(), vy, vout * 4, hipMemcpyDeviceToHost));
    for (int a = 0; a < n_aggr; ++a) {
      if (only_a && !strstr(aggressor_name(a), only_a)) continue;
      // warm-up of the aggressor alone (+ its duration), then the overlapped repetitions
      launch_aggressor(a, sa); CK(hipStreamSynchronize(sa));
      CK(hipEventRecord(e0, sa)); launch_aggressor(a, sa); CK(hipEventRecord(e1, sa)); CK(hipStreamSynchronize(sa));
      float a_ms = 0.f; CK(hipEventElapsedTime(&a_ms, e0, e1));
      int bad_launches = 0; long long bad_elems = 0; double max_dev = 0.0; float v_ms_sum = 0.f;
      long long lane_hist[64] = {0}, comp_hist[4] = {0}, colour_hist[3] = {0};
      for (int r = 0; r < REPS; ++r) {
        CK(hipMemsetAsync(vy, 0, vout * 4, sb));
        CK(hipStreamSynchronize(sb));
        // the order of round 5's reproducer: aggressor, victim, aggressor -- the victim's workgroups take the compute
        // units the first launch's tail frees and give them to the second launch's workgroups
        launch_aggressor(a, sa);
        CK(hipEventRecord(v0, sb)); launch_victim(v, sb); CK(hipEventRecord(v1, sb));
        launch_aggressor(a, sa);
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        float v_ms = 0.f; CK(hipEventElapsedTime(&v_ms, v0, v1)); v_ms_sum += v_ms;
        CK(hipMemcpy(got.data(), vy, vout * 4, hipMemcpyDeviceToHost));
        long long nb = 0;
        for (long long i = 0; i < vout; ++i) {
          if (memcmp(&got[i], &ref[i], 4) != 0) {
            ++nb;
            const double d = fabs((double)got[i] - (double)ref[i]);
            if (d > max_dev || d != d) max_dev = d != d ? 1e30 : d;
            const long long q = (i % vhw) >> 2;          // the lane's item: thread (q % 256) of a workgroup
            ++lane_hist[q & 63]; ++comp_hist[i & 3]; ++colour_hist[(i / vhw) % 3];
          }
        }
        if (nb) ++bad_launches;
        bad_elems += nb;
      }
      int lane_lo = 64, lane_hi = -1;
      for (int l = 0; l < 64; ++l) if (lane_hist[l]) { lane_lo = l < lane_lo ? l : lane_lo; lane_hi = l; }
      printf("{\"victim\": \"%s\", \"aggressor\": \"%s\", \"batch\": %d, \"overlapped_launches\": %d, \"launches_wrong\": %d, "
             "\"elements_wrong\": %lld, \"elements_per_launch\": %lld, \"max_abs_dev\": %.6g, \"lanes\": [%d, %d], "
             "\"components_xyzw\": [%lld, %lld, %lld, %lld], \"colours\": [%lld, %lld, %lld], \"aggressor_ms_alone\": %.3f, "
             "\"victim_ms_overlapped\": %.3f}\n",
             victim_name[v], aggressor_name(a), B, REPS, bad_launches, bad_elems, vout, max_dev, lane_hi < 0 ? -1 : lane_lo, lane_hi,
             comp_hist[0], comp_hist[1], comp_hist[2], comp_hist[3], colour_hist[0], colour_hist[1], colour_hist[2], a_ms, v_ms_sum / REPS);
      fflush(stdout);
    }
  }
  return 0;
}
