// hipcc-flags: -fno-slp-vectorize -fno-vectorize
// (no PACKED fp32 math -- v_pk_fma_f32 / v_pk_mul_f32 -- in this file's kernels: they are the streaming kernels that run on the
// side streams BESIDE the convolutions' MFMAs.  Round 5 found (scripts/interference_repro.py, profiles/r05i, r05l): with
// rw_tconv.hip's kernel running on another stream, to_rgb_kernel's v_pk_fma_f32 results came back wrong in the low half of
// lanes 48..63 of some waves -- 12 of 12 overlapped launches, 0 of 24 once the same source is compiled without the SLP
// vectoriser (scalar v_fma_f32).  Packed fp32 VALU shares the matrix pipe on gfx950 (scripts/probe/mfma16_valu_probe: it
// costs 10 - 18 cycles per MFMA slot): without it the FORWARD is also 2 % faster, 1344 against 1318 img/s on one box.)
// Streaming (HBM-bound) kernels of the StyleGANv2 sequential generator for gfx950:
// fused bias+leaky-ReLU, upfirdn2d, pixel-norm, equalised linear, style multiply,
// demodulation factors, weight repacking, noise injection, blur+noise+activation, ToRGB.
//
// All of these are bandwidth-bound (arithmetic intensity <= ~2 FLOP/B): the rules that
// matter are coalesced 16-byte accesses along W (NCHW => W is the fast axis), one pass over
// each feature map, and >> 256 workgroups.  Nothing here is reshaped into a GEMM.
#include "rw_common.h"
#ifndef OPS_NTS
#define OPS_NTS 0         // A/B builds: non-temporal stores of upfirdn2d_plane (1) / to_rgb (2)
#endif
#include <stdlib.h>
#include <stdint.h>

extern "C" int rw_abi_version(void) { return 9; }

extern "C" const char* rw_error_string(int code) {
  if (code == 0) return "success";
  if (code == RW_ERR_BAD_ARGUMENT) return "rewriting_hip: bad argument";
  if (code == RW_ERR_UNSUPPORTED) return "rewriting_hip: unsupported configuration";
  return hipGetErrorString((hipError_t)code);
}

// ---------------------------------------------------------------------------------------
// fused_bias_act   (reference: utils/stylegan2/op/fused_bias_act_kernel.cu:18-49)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float rw_bias_act_one(float x, float ref, int code, float alpha, float scale) {
  float y;
  switch (code) {
    default:
    case 10: y = x; break;
    case 11: y = x; break;
    case 12: y = 0.0f; break;
    case 30: y = (x > 0.0f) ? x : x * alpha; break;
    case 31: y = (ref > 0.0f) ? x : x * alpha; break;
    case 32: y = 0.0f; break;
  }
  return y * scale;
}

template <bool VEC4>
__global__ void __launch_bounds__(256) fused_bias_act_kernel(
    const float* __restrict__ x, const float* __restrict__ b, const float* __restrict__ ref,
    float* __restrict__ y, int64_t n, int64_t step_b, int64_t size_b, int code, float alpha,
    float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (VEC4) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      float4 v = reinterpret_cast<const float4*>(x)[i];
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ref) r = reinterpret_cast<const float4*>(ref)[i];
      if (b) {  // step_b % 4 == 0: the four lanes of the vector share one bias entry
        const float bv = b[((i << 2) / step_b) % size_b];
        v.x += bv; v.y += bv; v.z += bv; v.w += bv;
      }
      float4 o;
      o.x = rw_bias_act_one(v.x, r.x, code, alpha, scale);
      o.y = rw_bias_act_one(v.y, r.y, code, alpha, scale);
      o.z = rw_bias_act_one(v.z, r.z, code, alpha, scale);
      o.w = rw_bias_act_one(v.w, r.w, code, alpha, scale);
      reinterpret_cast<float4*>(y)[i] = o;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      float v = x[i];
      if (b) v += b[(i / step_b) % size_b];
      const float r = ref ? ref[i] : 0.0f;
      y[i] = rw_bias_act_one(v, r, code, alpha, scale);
    }
  }
}

extern "C" int rw_fused_bias_act_f32(const float* x, const float* b, const float* ref, float* y,
                                     int64_t n, int64_t step_b, int64_t size_b, int act, int grad,
                                     float alpha, float scale, rw_stream_t stream) {
  if (n == 0) return 0;
  RW_CHECK_ARG(x && y && n > 0);
  RW_CHECK_ARG(!b || (step_b > 0 && size_b > 0));
  const int code = act * 10 + grad;
  const bool aligned = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)ref) & 15) == 0;
  const bool vec = aligned && (n % 4 == 0) && (!b || step_b % 4 == 0);
  if (vec) {
    hipLaunchKernelGGL(fused_bias_act_kernel<true>, dim3(rw_stream_grid(n / 4, 256)), dim3(256), 0,
                       rw_s(stream), x, b, ref, y, n, step_b, size_b, code, alpha, scale);
  } else {
    hipLaunchKernelGGL(fused_bias_act_kernel<false>, dim3(rw_stream_grid(n, 256)), dim3(256), 0,
                       rw_s(stream), x, b, ref, y, n, step_b, size_b, code, alpha, scale);
  }
  return RW_LAUNCH_RESULT();
}

// grad_bias[c] = sum_{outer, inner} g[o][c][i]        (op/fused_act.py:32-39)
__global__ void __launch_bounds__(256) bias_grad_kernel(const float* __restrict__ g,
                                                        float* __restrict__ gb, int64_t outer,
                                                        int64_t channels, int64_t inner) {
  __shared__ float red[4];
  const int64_t c = blockIdx.x;
  float acc = 0.f;
  for (int64_t o = 0; o < outer; ++o) {
    const float* row = g + (o * channels + c) * inner;
    for (int64_t i = threadIdx.x; i < inner; i += 256) acc += row[i];
  }
  acc = rw_block_sum_256(acc, red);
  if (threadIdx.x == 0) gb[c] = acc;
}

extern "C" int rw_bias_grad_f32(const float* g, float* gb, int64_t outer, int64_t channels,
                                int64_t inner, rw_stream_t stream) {
  RW_CHECK_ARG(g && gb && outer > 0 && channels > 0 && inner > 0);
  hipLaunchKernelGGL(bias_grad_kernel, dim3((unsigned)channels), dim3(256), 0, rw_s(stream), g, gb,
                     outer, channels, inner);
  return RW_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------
// upfirdn2d   (reference: utils/stylegan2/op/upfirdn2d_kernel.cu:52-137)
// One thread per output sample; taps that land on inserted zeros or padding are skipped, and
// the kernel is applied FLIPPED (:71-81).  minor == 1 for every NCHW caller, so consecutive
// threads walk W: coalesced stores, L1-served overlapping loads.
// ---------------------------------------------------------------------------------------
struct UpfirdnParams {
  int major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, px0, py0, out_h, out_w;
};

__global__ void __launch_bounds__(256) upfirdn2d_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ k,
                                                        float* __restrict__ y, UpfirdnParams p) {
  __shared__ float sk[64];
  for (int t = threadIdx.x; t < p.kh * p.kw; t += 256) {
    const int ky = t / p.kw, kx = t - ky * p.kw;
    sk[t] = k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
  }
  __syncthreads();
  const int64_t total = (int64_t)p.major * p.out_h * p.out_w * p.minor;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    int64_t r = idx;
    const int mi = (int)(r % p.minor); r /= p.minor;
    const int ox = (int)(r % p.out_w); r /= p.out_w;
    const int oy = (int)(r % p.out_h); r /= p.out_h;
    const int64_t ma = r;
    const float* xm = x + ma * (int64_t)p.in_h * p.in_w * p.minor + mi;
    float acc = 0.f;
    for (int a = 0; a < p.kh; ++a) {
      const int vy = oy * p.down_y + a - p.py0;     // position in the zero-inserted image
      if (vy < 0 || vy % p.up_y) continue;
      const int iy = vy / p.up_y;
      if (iy >= p.in_h) continue;
      for (int c = 0; c < p.kw; ++c) {
        const int vx = ox * p.down_x + c - p.px0;
        if (vx < 0 || vx % p.up_x) continue;
        const int ix = vx / p.up_x;
        if (ix >= p.in_w) continue;
        acc += xm[((int64_t)iy * p.in_w + ix) * p.minor] * sk[a * p.kw + c];
      }
    }
    y[idx] = acc;
  }
}

// minor == 1 with up / down factors in {1, 2} (every caller on the generator path): a 256 x 16
// output tile per workgroup, four consecutive samples per thread (one 16-byte store), 32-bit index
// arithmetic, compile-time factors and a polyphase tap walk instead of three 64-bit divisions and
// kh*kw guarded taps per sample.  Same taps in the same order as the kernel above -> bit-identical
// results.
#define UF_TW 256
#define UF_TH 16
template <int UP, int DOWN>
__global__ void __launch_bounds__(256) upfirdn2d_plane_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ k,
                                                              float* __restrict__ y, UpfirdnParams p,
                                                              int tiles_x, int tiles_y) {
  __shared__ float sk[64];
  for (int t = threadIdx.x; t < p.kh * p.kw; t += 256) {
    const int ky = t / p.kw, kx = t - ky * p.kw;
    sk[t] = k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
  }
  __syncthreads();
  int blk = blockIdx.x;
  const int tx = blk % tiles_x; blk /= tiles_x;
  const int ty = blk % tiles_y;
  const int ma = blk / tiles_y;
  const int ox = tx * UF_TW + (threadIdx.x & 63) * 4;
  if (ox >= p.out_w) return;
  const float* xm = x + (int64_t)ma * p.in_h * p.in_w;
  float* ym = y + (int64_t)ma * p.out_h * p.out_w;
  const bool vec = (p.out_w % 4 == 0);             // then ox + 3 < out_w and rows start 16-byte aligned
  for (int oy = ty * UF_TH + (threadIdx.x >> 6); oy < min((ty + 1) * UF_TH, p.out_h); oy += 4) {
    // polyphase: only taps a = a0, a0 + UP, ... land on real samples (ascending, like the loop above)
    const int a0 = (((p.py0 - oy * DOWN) % UP) + UP) % UP;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int a = a0; a < p.kh; a += UP) {
      const int vy = oy * DOWN + a - p.py0;
      const int iy = vy / UP;
      if (vy < 0 || iy >= p.in_h) continue;
      const float* xr = xm + (int64_t)iy * p.in_w;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c0 = (((p.px0 - (ox + e) * DOWN) % UP) + UP) % UP;
        for (int c = c0; c < p.kw; c += UP) {
          const int vx = (ox + e) * DOWN + c - p.px0;
          const int ix = vx / UP;
          if (vx < 0 || ix >= p.in_w) continue;
          acc[e] += xr[ix] * sk[a * p.kw + c];
        }
      }
    }
    float* yo = ym + (int64_t)oy * p.out_w + ox;
    if (vec) {
#if OPS_NTS & 1
      __builtin_nontemporal_store(rw_f32x4{acc[0], acc[1], acc[2], acc[3]}, reinterpret_cast<rw_f32x4*>(yo));
#else
      *reinterpret_cast<float4*>(yo) = make_float4(acc[0], acc[1], acc[2], acc[3]);
#endif
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (ox + e < p.out_w) yo[e] = acc[e];
    }
  }
}

// The generator's own upsampling (UpsampleO on the running RGB image: up 2, a 4 x 4 kernel, minor 1; the polyphase
// walk above spends ~40 integer instructions per tap on it).  Here everything that does not depend on the row is
// computed once per thread -- per output column its two input columns, their validity and kernel columns -- and a row
// costs two row pointers, sixteen loads and sixteen FMAs per four outputs.  Same taps in the same order as
// upfirdn2d_kernel (a ascending, then c ascending; out-of-map taps skipped, not added as zeros) -> bit-identical.
__global__ void __launch_bounds__(256) upfirdn2d_up2k4_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                              float* __restrict__ y, UpfirdnParams p, int tiles_x,
                                                              int tiles_y) {
  __shared__ float sk[16];
  if (threadIdx.x < 16) {
    const int ky = threadIdx.x >> 2, kx = threadIdx.x & 3;
    sk[threadIdx.x] = k[(3 - ky) * 4 + (3 - kx)];
  }
  __syncthreads();
  int blk = blockIdx.x;
  const int tx = blk % tiles_x; blk /= tiles_x;
  const int ty = blk % tiles_y;
  const int ma = blk / tiles_y;
  const int ox = tx * UF_TW + (threadIdx.x & 63) * 4;
  if (ox >= p.out_w) return;
  const float* xm = x + (int64_t)ma * p.in_h * p.in_w;
  float* ym = y + (int64_t)ma * p.out_h * p.out_w;
  int ix[4][2], kc[4][2];
  bool okc[4][2];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c0 = (p.px0 - (ox + e)) & 1;                      // kernel columns c0, c0 + 2 land on samples
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + 2 * j, vx = ox + e + c - p.px0;
      kc[e][j] = c;
      ix[e][j] = vx >> 1;
      okc[e][j] = vx >= 0 && (vx >> 1) < p.in_w;
      if (!okc[e][j]) ix[e][j] = 0;
    }
  }
  const bool vec = (p.out_w % 4 == 0);
  const int oy1 = min((ty + 1) * UF_TH, p.out_h);
  for (int oy = ty * UF_TH + (threadIdx.x >> 6); oy < oy1; oy += 4) {
    const int a0 = (p.py0 - oy) & 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int a = a0 + 2 * j2, vy = oy + a - p.py0, iy = vy >> 1;
      if (vy < 0 || iy >= p.in_h) continue;
      const float* xr = xm + (int64_t)iy * p.in_w;
      const float* kr = sk + 4 * a;
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (okc[e][j]) acc[e] += xr[ix[e][j]] * kr[kc[e][j]];
    }
    float* yo = ym + (int64_t)oy * p.out_w + ox;
    if (vec) {
      *reinterpret_cast<float4*>(yo) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (ox + e < p.out_w) yo[e] = acc[e];
    }
  }
}

extern "C" int rw_upfirdn2d_f32(const float* x, const float* k, float* y, int major, int in_h,
                                int in_w, int minor, int kh, int kw, int up_x, int up_y,
                                int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                                int pad_y1, rw_stream_t stream) {
  RW_CHECK_ARG(x && k && y && major >= 0 && in_h > 0 && in_w > 0 && minor > 0);
  RW_CHECK_ARG(kh > 0 && kw > 0 && kh * kw <= 64 && up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0);
  UpfirdnParams p;
  p.major = major; p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kh = kh; p.kw = kw;
  p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.px0 = pad_x0; p.py0 = pad_y0;
  p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
  p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
  if (p.out_h <= 0 || p.out_w <= 0 || major == 0) return 0;
  const int64_t total = (int64_t)major * p.out_h * p.out_w * minor;
  const int tiles_x = (int)rw_cdiv(p.out_w, UF_TW), tiles_y = (int)rw_cdiv(p.out_h, UF_TH);
  const int64_t blocks = (int64_t)major * tiles_x * tiles_y;
  if (minor == 1 && up_x == up_y && down_x == down_y && up_x <= 2 && down_x <= 2 &&
      blocks <= 0x7fffffff) {
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = rw_s(stream);
    if (up_x == 2 && down_x == 1 && kh == 4 && kw == 4)
      hipLaunchKernelGGL(upfirdn2d_up2k4_kernel, grid, block, 0, st, x, k, y, p, tiles_x, tiles_y);
    else if (up_x == 1 && down_x == 1)
      hipLaunchKernelGGL((upfirdn2d_plane_kernel<1, 1>), grid, block, 0, st, x, k, y, p, tiles_x, tiles_y);
    else if (up_x == 2 && down_x == 1)
      hipLaunchKernelGGL((upfirdn2d_plane_kernel<2, 1>), grid, block, 0, st, x, k, y, p, tiles_x, tiles_y);
    else if (up_x == 1 && down_x == 2)
      hipLaunchKernelGGL((upfirdn2d_plane_kernel<1, 2>), grid, block, 0, st, x, k, y, p, tiles_x, tiles_y);
    else
      hipLaunchKernelGGL((upfirdn2d_plane_kernel<2, 2>), grid, block, 0, st, x, k, y, p, tiles_x, tiles_y);
    return RW_LAUNCH_RESULT();
  }
  hipLaunchKernelGGL(upfirdn2d_kernel, dim3(rw_stream_grid(total, 256)), dim3(256), 0, rw_s(stream),
                     x, k, y, p);
  return RW_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------
// Mapping network pieces
// ---------------------------------------------------------------------------------------
// PixelNormL (models.py:609-614): one wave per latent row.
__global__ void __launch_bounds__(256) pixel_norm_kernel(const float* __restrict__ x,
                                                         float* __restrict__ y, int batch, int dim,
                                                         float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= batch) return;
  const float* xr = x + (int64_t)row * dim;
  float ss = 0.f;
  for (int i = lane; i < dim; i += 64) { const float v = xr[i]; ss += v * v; }
  ss = rw_wave_sum(ss);
  const float r = rsqrtf(ss / (float)dim + eps);
  for (int i = lane; i < dim; i += 64) y[(int64_t)row * dim + i] = xr[i] * r;
}

extern "C" int rw_pixel_norm_f32(const float* x, float* y, int batch, int dim, float eps,
                                 rw_stream_t stream) {
  RW_CHECK_ARG(x && y && batch > 0 && dim > 0);
  hipLaunchKernelGGL(pixel_norm_kernel, dim3((batch + 3) / 4), dim3(256), 0, rw_s(stream), x, y,
                     batch, dim, eps);
  return RW_LAUNCH_RESULT();
}

// EqualLinear (models.py:503-511).  One wave per output feature: the weight row stays in
// registers (scaled once, as the reference scales the weight before F.linear) while the
// wave walks the batch; per-row dot products finish with a 64-lane butterfly.
#define RW_LINEAR_MAX_PER_LANE 16  // in_dim <= 1024
#define RW_LINEAR_ROWS 4           // batch rows per wave pass (independent butterflies overlap)
__global__ void __launch_bounds__(256) equal_linear_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int batch, int in_dim, int out_dim, int64_t x_stride, float w_scale,
    float b_scale, int act, float alpha, float act_scale) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= out_dim) return;
  float wr[RW_LINEAR_MAX_PER_LANE];
  const int per = (in_dim + 63) / 64;
#pragma unroll
  for (int j = 0; j < RW_LINEAR_MAX_PER_LANE; ++j) {
    const int i = lane + j * 64;
    wr[j] = (j < per && i < in_dim) ? w[(int64_t)o * in_dim + i] * w_scale : 0.f;
  }
  const float bv = bias ? bias[o] * b_scale : 0.f;
  for (int b0 = blockIdx.y * RW_LINEAR_ROWS; b0 < batch; b0 += gridDim.y * RW_LINEAR_ROWS) {
    float acc[RW_LINEAR_ROWS];
#pragma unroll
    for (int r = 0; r < RW_LINEAR_ROWS; ++r) {
      acc[r] = 0.f;
      const int b = b0 + r;
      if (b < batch) {
        const float* xr = x + (int64_t)b * x_stride;
#pragma unroll
        for (int j = 0; j < RW_LINEAR_MAX_PER_LANE; ++j) {
          const int i = lane + j * 64;
          if (j < per && i < in_dim) acc[r] += xr[i] * wr[j];
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int r = 0; r < RW_LINEAR_ROWS; ++r) acc[r] += __shfl_xor(acc[r], off, 64);
    if (lane < RW_LINEAR_ROWS && b0 + lane < batch) {
      float v = acc[0];
#pragma unroll
      for (int r = 1; r < RW_LINEAR_ROWS; ++r) v = (lane == r) ? acc[r] : v;
      v += bv;
      if (act) v = ((v > 0.f) ? v : v * alpha) * act_scale;
      y[(int64_t)(b0 + lane) * out_dim + o] = v;
    }
  }
}

// The same on the matrix pipe for the shapes the generator uses (in_dim % 16 == 0, out_dim % 16 == 0): one wave =
// a 16 (batch) x 16 (out) tile, v_mfma_f32_16x16x4_f32 over K.  Lane (m | n = lane & 15, q = lane >> 4) loads
// x[b0 + m][16 c + 4 q .. + 3] and w[o0 + n][16 c + 4 q .. + 3] as one 16-byte load each per 16-wide K block c and feeds
// component j to MFMA j of the block (k = 16 c + 4 q + j on both operands: every k exactly once).  The butterfly
// kernel above is one memory latency + six shuffle rounds per four batch rows (18 us at batch 64, 51 us at 250 --
// 7 % of a key-statistics sweep); this one is 128 MFMAs behind 8-deep load batches.
typedef float rw_lin_f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) equal_linear_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int batch, int in_dim, int out_dim, int64_t x_stride, float w_scale,
    float b_scale, int act, float alpha, float act_scale) {
  const int lane = threadIdx.x, mn = lane & 15, q = lane >> 4;
  const int o0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int brow = min(b0 + mn, batch - 1);                 // rows past the batch: a legal row, results dropped
  const float* xp = x + (int64_t)brow * x_stride + 4 * q;
  const float* wp = w + (int64_t)(o0 + mn) * in_dim + 4 * q;
  rw_lin_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int blocks = in_dim >> 4;
  constexpr int DEPTH = 8;
  for (int c0 = 0; c0 < blocks; c0 += DEPTH) {
    rw_lin_f32x4 a[DEPTH], b[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int c = min(c0 + d, blocks - 1);
      a[d] = *reinterpret_cast<const rw_lin_f32x4*>(xp + 16 * c);
      b[d] = *reinterpret_cast<const rw_lin_f32x4*>(wp + 16 * c);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (c0 + d < blocks) {
        const rw_lin_f32x4 bs = b[d] * w_scale;            // the reference scales the weight before F.linear
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[d][j], bs[j], acc, 0, 0, 0);
      }
    }
  }
  // acc[j] = y[b0 + 4 q + j][o0 + mn]   (C/D layout: row = 4 (lane >> 4) + j, column = lane & 15)
  const int o = o0 + mn;
  const float bv = bias ? bias[o] * b_scale : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int b = b0 + 4 * q + j;
    if (b < batch) {
      float v = acc[j] + bv;
      if (act) v = ((v > 0.f) ? v : v * alpha) * act_scale;
      y[(int64_t)b * out_dim + o] = v;
    }
  }
}

extern "C" int rw_equal_linear_f32(const float* x, const float* w, const float* bias, float* y,
                                   int batch, int in_dim, int out_dim, int64_t x_stride,
                                   float w_scale, float b_scale, int act, float alpha,
                                   float act_scale, rw_stream_t stream) {
  RW_CHECK_ARG(x && w && y && batch > 0 && in_dim > 0 && out_dim > 0 && x_stride >= in_dim);
  const char* impl = getenv("RW_LINEAR_IMPL");              // 1 = the butterfly kernel (A/B runs, tests)
  const bool butterfly = impl && atoi(impl) == 1;
  if (!butterfly && in_dim % 16 == 0 && out_dim % 16 == 0 && x_stride % 4 == 0 && ((uintptr_t)x & 15) == 0 &&
      ((uintptr_t)w & 15) == 0) {
    hipLaunchKernelGGL(equal_linear_mfma_kernel, dim3(out_dim / 16, (batch + 15) / 16), dim3(64), 0, rw_s(stream), x,
                       w, bias, y, batch, in_dim, out_dim, x_stride, w_scale, b_scale, act, alpha, act_scale);
    return RW_LAUNCH_RESULT();
  }
  if (in_dim > 64 * RW_LINEAR_MAX_PER_LANE) return RW_ERR_UNSUPPORTED;
  int gy = (batch + RW_LINEAR_ROWS - 1) / RW_LINEAR_ROWS;
  if (gy > 64) gy = 64;
  hipLaunchKernelGGL(equal_linear_kernel, dim3((out_dim + 3) / 4, gy), dim3(256), 0, rw_s(stream), x, w,
                     bias, y, batch, in_dim, out_dim, x_stride, w_scale, b_scale, act, alpha,
                     act_scale);
  return RW_LAUNCH_RESULT();
}

// AdjustLatent (models.py:570-583)
__global__ void __launch_bounds__(256) adjust_latent_kernel(const float* __restrict__ w,
                                                            const float* __restrict__ avg,
                                                            float* __restrict__ out, int batch,
                                                            int n_latent, int dim, float psi) {
  const int64_t total = (int64_t)batch * n_latent * dim;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(idx % dim);
    const int b = (int)(idx / ((int64_t)n_latent * dim));
    float v = w[(int64_t)b * dim + d];
    if (avg) { const float a = avg[d]; v = a + psi * (v - a); }
    out[idx] = v;
  }
}

extern "C" int rw_adjust_latent_f32(const float* w, const float* avg, float* out, int batch,
                                    int n_latent, int dim, float psi, rw_stream_t stream) {
  RW_CHECK_ARG(w && out && batch > 0 && n_latent > 0 && dim > 0);
  const int64_t total = (int64_t)batch * n_latent * dim;
  hipLaunchKernelGGL(adjust_latent_kernel, dim3(rw_stream_grid(total, 256)), dim3(256), 0,
                     rw_s(stream), w, avg, out, batch, n_latent, dim, psi);
  return RW_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------
// ApplyStyle / NoiseInjection (per-row scalar broadcast over a contiguous row of hw floats)
// ---------------------------------------------------------------------------------------
// y[row][p] = x[row][p] * rowscale[row]                               (ApplyStyle)
// y[b][c][p] = x[b][c][p] + nw * noise[b][p]   (rows = b*C + c)       (NoiseInjectionF)
template <int MODE>  // 0 = style multiply, 1 = noise add
__global__ void __launch_bounds__(256) row_broadcast_kernel(
    const float* __restrict__ x, const float* __restrict__ aux, const float* __restrict__ nw_ptr,
    float* __restrict__ y, int64_t rows, int channels, int64_t hw) {
  const int64_t total = rows * hw;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float nw = (MODE == 1) ? nw_ptr[0] : 0.f;
  if ((hw & 3) == 0) {
    const int64_t hw4 = hw >> 2, total4 = total >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
      const int64_t row = i / hw4;
      float4 v = reinterpret_cast<const float4*>(x)[i];
      if (MODE == 0) {
        const float s = aux[row];
        v.x = s * v.x; v.y = s * v.y; v.z = s * v.z; v.w = s * v.w;
      } else {
        const int64_t b = row / channels;
        const float4 nz = reinterpret_cast<const float4*>(aux)[b * hw4 + (i - row * hw4)];
        v.x = v.x + nw * nz.x; v.y = v.y + nw * nz.y; v.z = v.z + nw * nz.z; v.w = v.w + nw * nz.w;
      }
      reinterpret_cast<float4*>(y)[i] = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int64_t row = i / hw;
      if (MODE == 0) {
        y[i] = aux[row] * x[i];
      } else {
        const int64_t b = row / channels;
        y[i] = x[i] + nw * aux[b * hw + (i - row * hw)];
      }
    }
  }
}

extern "C" int rw_style_mul_f32(const float* x, const float* style, float* y, int batch,
                                int channels, int64_t hw, rw_stream_t stream) {
  RW_CHECK_ARG(x && style && y && batch > 0 && channels > 0 && hw > 0);
  const int64_t rows = (int64_t)batch * channels;
  hipLaunchKernelGGL(row_broadcast_kernel<0>, dim3(rw_stream_grid(rows * hw / 4 + 1, 256)),
                     dim3(256), 0, rw_s(stream), x, style, (const float*)nullptr, y, rows, channels, hw);
  return RW_LAUNCH_RESULT();
}

extern "C" int rw_noise_add_f32(const float* x, const float* noise, const float* noise_w, float* y,
                                int batch, int channels, int64_t hw, rw_stream_t stream) {
  RW_CHECK_ARG(x && noise && noise_w && y && batch > 0 && channels > 0 && hw > 0);
  const int64_t rows = (int64_t)batch * channels;
  hipLaunchKernelGGL(row_broadcast_kernel<1>, dim3(rw_stream_grid(rows * hw / 4 + 1, 256)),
                     dim3(256), 0, rw_s(stream), x, noise, noise_w, y, rows, channels, hw);
  return RW_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------
// Demodulation factors (DemodulatedConv2dF.forward, models.py:320-328)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) weight_sqsum_kernel(const float* __restrict__ w,
                                                           float* __restrict__ wsq, int64_t pairs,
                                                           int taps, float w_scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs;
       i += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int t = 0; t < taps; ++t) { const float v = w_scale * w[i * taps + t]; acc += v * v; }
    wsq[i] = acc;
  }
}

extern "C" int rw_weight_sqsum_f32(const float* w, float* wsq, int out_ch, int in_ch, int taps,
                                   float w_scale, rw_stream_t stream) {
  RW_CHECK_ARG(w && wsq && out_ch > 0 && in_ch > 0 && taps > 0);
  const int64_t pairs = (int64_t)out_ch * in_ch;
  hipLaunchKernelGGL(weight_sqsum_kernel, dim3(rw_stream_grid(pairs, 256)), dim3(256), 0,
                     rw_s(stream), w, wsq, pairs, taps, w_scale);
  return RW_LAUNCH_RESULT();
}

// one wave per (b, o): 64-lane butterfly for the per-channel reduction
__global__ void __launch_bounds__(256) demod_kernel(const float* __restrict__ wsq,
                                                    const float* __restrict__ style,
                                                    float* __restrict__ demod, int batch, int out_ch,
                                                    int in_ch, float eps) {
  const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (idx >= (int64_t)batch * out_ch) return;
  const int b = (int)(idx / out_ch), o = (int)(idx % out_ch);
  float acc = 0.f;
  for (int i = lane; i < in_ch; i += 64) {
    const float s = style[(int64_t)b * in_ch + i];
    acc += (s * s) * wsq[(int64_t)o * in_ch + i];
  }
  acc = rw_wave_sum(acc);
  if (lane == 0) demod[idx] = rsqrtf(acc + eps);
}

// the same on the matrix pipe (in_ch % 16 == 0, out_ch % 16 == 0): a 16 (batch) x 16 (out) tile per wave, operands and
// k mapping as in equal_linear_mfma_kernel, A = style^2
__global__ void __launch_bounds__(64) demod_mfma_kernel(const float* __restrict__ wsq, const float* __restrict__ style,
                                                        float* __restrict__ demod, int batch, int out_ch, int in_ch,
                                                        float eps) {
  const int lane = threadIdx.x, mn = lane & 15, q = lane >> 4;
  const int o0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int brow = min(b0 + mn, batch - 1);
  const float* sp = style + (int64_t)brow * in_ch + 4 * q;
  const float* wp = wsq + (int64_t)(o0 + mn) * in_ch + 4 * q;
  rw_lin_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int blocks = in_ch >> 4;
  constexpr int DEPTH = 8;
  for (int c0 = 0; c0 < blocks; c0 += DEPTH) {
    rw_lin_f32x4 a[DEPTH], b[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int c = min(c0 + d, blocks - 1);
      a[d] = *reinterpret_cast<const rw_lin_f32x4*>(sp + 16 * c);
      b[d] = *reinterpret_cast<const rw_lin_f32x4*>(wp + 16 * c);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (c0 + d < blocks) {
        // (component by component: a vector multiply would be two v_pk_mul_f32 -- no packed fp32 math in this file)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[d][j] * a[d][j], b[d][j], acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int b = b0 + 4 * q + j;
    if (b < batch) demod[(int64_t)b * out_ch + o0 + mn] = rsqrtf(acc[j] + eps);
  }
}

extern "C" int rw_demod_f32(const float* wsq, const float* style, float* demod, int batch,
                            int out_ch, int in_ch, float eps, rw_stream_t stream) {
  RW_CHECK_ARG(wsq && style && demod && batch > 0 && out_ch > 0 && in_ch > 0);
  const char* impl = getenv("RW_LINEAR_IMPL");              // 1 = the butterfly kernel (A/B runs, tests)
  if (!(impl && atoi(impl) == 1) && in_ch % 16 == 0 && out_ch % 16 == 0 && ((uintptr_t)wsq & 15) == 0 &&
      ((uintptr_t)style & 15) == 0) {
    hipLaunchKernelGGL(demod_mfma_kernel, dim3(out_ch / 16, (batch + 15) / 16), dim3(64), 0, rw_s(stream), wsq, style,
                       demod, batch, out_ch, in_ch, eps);
    return RW_LAUNCH_RESULT();
  }
  const int64_t waves = (int64_t)batch * out_ch;
  hipLaunchKernelGGL(demod_kernel, dim3((unsigned)rw_cdiv(waves, 4)), dim3(256), 0, rw_s(stream),
                     wsq, style, demod, batch, out_ch, in_ch, eps);
  return RW_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------
// Weight repack for the implicit-GEMM convolutions: [o][i][tap] -> [slab][i][o], followed by the
// same weights in MFMA A-fragment order for the halo-tile kernels (see rewriting_hip.h)
// ---------------------------------------------------------------------------------------
__constant__ int rw_up_tap_order[9] = {0, 2, 6, 8, 1, 7, 3, 5, 4};

static inline bool rw_frag_ok(int out_ch, int in_ch) { return out_ch % 32 == 0 && in_ch % 16 == 0; }
static inline int rw_frag_ic(int out_ch, int mode) { return (mode == 1 || out_ch % 64 == 0) ? 16 : 8; }

extern "C" long long rw_packed_conv_weight_elems(int out_ch, int in_ch, int mode) {
  if (out_ch <= 0 || in_ch <= 0 || (mode != 0 && mode != 1)) return -1;
  const long long io = (long long)in_ch * out_ch;
  if (!rw_frag_ok(out_ch, in_ch)) return 9 * io;
  return 18 * io;
}

__global__ void __launch_bounds__(256) pack_conv_weight_kernel(const float* __restrict__ w,
                                                               float* __restrict__ wp, int out_ch,
                                                               int in_ch, int mode) {
  const int64_t total = (int64_t)9 * in_ch * out_ch;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int o = (int)(idx % out_ch);
    const int i = (int)((idx / out_ch) % in_ch);
    const int slab = (int)(idx / ((int64_t)out_ch * in_ch));
    const int tap = mode == 0 ? slab : rw_up_tap_order[slab];
    wp[idx] = w[((int64_t)o * in_ch + i) * 9 + tap];
  }
}

// Fragment order.  A lane of v_mfma_f32_32x32x2_f32 holds A[o = lane & 31][k = lane >> 5]; the
// kernels walk k-pairs kp of an IC-channel chunk c, so lane l of out-channel block ob needs
// W[32 ob + (l & 31)][c IC + 2 kp + (l >> 5)][tap].
//   mode 0: wf[tap][c][ob][kp / 4][lane][kp % 4]       -> one 16-byte load per lane gives 4 k-pairs,
//   mode 1: wf[c][ob][kp]{[2][lane][4], [lane]}        -> two give slabs 0..7 of one, a dword slab 8,
// and every 16-byte load instruction of a wave reads 1 KiB of consecutive addresses.
__global__ void __launch_bounds__(256) pack_conv_frag_kernel(const float* __restrict__ w,
                                                             float* __restrict__ wf, int out_ch,
                                                             int in_ch, int mode, int ic) {
  const int obn = out_ch >> 5, kpn = ic >> 1, chunks = in_ch / ic;
  const int64_t total = (int64_t)9 * in_ch * out_ch;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    int kp, tap, c, ob, lane;
    if (mode == 0) {
      const int e = (int)(r & 3); r >>= 2;
      lane = (int)(r & 63); r >>= 6;
      const int h = (int)(r % (kpn >> 2)); r /= (kpn >> 2);
      ob = (int)(r % obn); r /= obn;
      c = (int)(r % chunks); r /= chunks;
      tap = (int)r;
      kp = 4 * h + e;
    } else {
      const int u = (int)(r % 576); r /= 576;
      kp = (int)(r % kpn); r /= kpn;
      ob = (int)(r % obn); r /= obn;
      c = (int)r;
      int slab;
      if (u < 512) { slab = 4 * (u >> 8) + (u & 3); lane = (u >> 2) & 63; }
      else { slab = 8; lane = u - 512; }
      tap = rw_up_tap_order[slab];
    }
    const int o = 32 * ob + (lane & 31);
    const int i = c * ic + 2 * kp + (lane >> 5);
    wf[idx] = w[((int64_t)o * in_ch + i) * 9 + tap];
  }
}

extern "C" int rw_pack_conv_weight_f32(const float* w, float* wp, int out_ch, int in_ch, int mode,
                                       rw_stream_t stream) {
  RW_CHECK_ARG(w && wp && out_ch > 0 && in_ch > 0 && (mode == 0 || mode == 1));
  const int64_t total = (int64_t)9 * in_ch * out_ch;
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(rw_stream_grid(total, 256)), dim3(256), 0,
                     rw_s(stream), w, wp, out_ch, in_ch, mode);
  if (rw_frag_ok(out_ch, in_ch)) {
    hipLaunchKernelGGL(pack_conv_frag_kernel, dim3(rw_stream_grid(total, 256)), dim3(256), 0,
                       rw_s(stream), w, wp + total, out_ch, in_ch, mode, rw_frag_ic(out_ch, mode));
  }
  return RW_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------
// Blur(pad 1,1) + noise + bias + leaky-ReLU for upsampling layers: one pass
// ---------------------------------------------------------------------------------------
// Workgroup = 64 x 64 output tile of one (image, channel) plane: the 67 x 67 input patch is
// staged in LDS with coalesced row loads (the odd row length 2W+1 rules out vector loads), each
// thread then produces 4 horizontally adjacent outputs of four rows, one 16-byte store each.
#ifndef BL_TH
#define BL_TH 64        // 16 / 32 / 64 / 96 / 128 rows measured: 3.1 / 3.5 / 3.9 / 3.6 / 3.3 TB/s at 32 x 1024^2 x 64
#endif
#define BL_TW 64
#define BL_PITCH (BL_TW + 4)
#ifndef BL_NT
#define BL_NT 2           // bit 0: non-temporal loads of the (2H+1)^2 map (measured -8 %: its halo rows ARE re-read),
                          // bit 1: non-temporal stores of the result (+9 % on the kernel: 3.77 -> 4.15 TB/s at 64 x 512^2 x 64)
#endif
__global__ void __launch_bounds__(256) blur_noise_act_kernel(
    const float* __restrict__ x, const float* __restrict__ k4, const float* __restrict__ noise,
    const float* __restrict__ nw_ptr, const float* __restrict__ bias, float* __restrict__ y,
    int batch, int channels, int out_h, int out_w, int tiles_x, int tiles_y, const float* __restrict__ post,
    float* __restrict__ y_amax) {
  __shared__ float kf[16];
  __shared__ __attribute__((aligned(16))) float tile[BL_TH + 3][BL_PITCH];
  const int tid = threadIdx.x;
  if (tid < 16) {
    const int a = tid >> 2, c = tid & 3;
    kf[tid] = k4[(3 - a) * 4 + (3 - c)];     // flipped, as upfirdn2d applies it
  }
  int blk = blockIdx.x;
  const int tx = blk % tiles_x; blk /= tiles_x;
  const int ty = blk % tiles_y;
  const int64_t bc = blk / tiles_y;
  const int c = (int)(bc % channels);
  const int64_t b = bc / channels;
  const int in_h = out_h + 1, in_w = out_w + 1;
  const int oy0 = ty * BL_TH, ox0 = tx * BL_TW;
  const float* xp = x + bc * (int64_t)in_h * in_w;
  // input patch rows oy0-1 .. oy0+BL_TH+1, cols ox0-1 .. ox0+BL_TW+2 (one spare): 17 lanes per row
  // load four floats each as ONE 16-byte load (rows are 2W+1 floats long, so only 4-byte aligned;
  // the hardware takes unaligned vector loads), all loads issued before the first LDS write.
  {
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    constexpr int LPR = BL_TW / 4 + 1;                  // lanes per patch row
    constexpr int RPP = 256 / LPR;                      // rows per pass
    constexpr int NP = (BL_TH + 3 + RPP - 1) / RPP;
    const int rr = tid / LPR, q4 = (tid - rr * LPR) * 4;
    rw_f32x4 v[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int r = rr + RPP * q, iy = oy0 - 1 + r, ix = ox0 - 1 + q4;
      const bool rok = rr < RPP && r < BL_TH + 3 && iy >= 0 && iy < in_h;
      const float* src = xp + (int64_t)iy * in_w + ix;
      if (rok && ix >= 0 && ix + 3 < in_w) {
#if BL_NT & 1
        v[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_u*>(src));
#else
        v[q] = *reinterpret_cast<const f32x4_u*>(src);
#endif
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[q][e] = (rok && ix + e >= 0 && ix + e < in_w) ? src[e] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int r = rr + RPP * q;
      if (rr < RPP && r < BL_TH + 3) *reinterpret_cast<rw_f32x4*>(&tile[r][q4]) = v[q];
    }
  }
  __syncthreads();
  // thread = 4 consecutive outputs of rows ly, ly + 16, ...: two 16-byte LDS reads per tap row, one
  // 16-byte noise load and one 16-byte store per output row -- the store tail is instruction-issue
  // bound, so wide stores matter more than the (mild, 2-way) LDS bank overlap of this mapping
  const int lx = (tid & 15) * 4;
  const int ox = ox0 + lx;
  const bool live = ox < out_w;
  float ymax = 0.f;                             // max |result| of this thread -> y_amax (the next layer's x_amax)
  const float nw = noise ? nw_ptr[0] : 0.f;
  const float bv = bias ? bias[c] : 0.f;
  const float ps = post ? post[bc] : 1.f;       // per (image, channel) factor on the result: the NEXT layer's style
  const bool full = (out_w % 4 == 0);           // then ox + 3 < out_w and every row start is 16-byte aligned
#pragma unroll
  for (int half = 0; half < BL_TH / 16; ++half) {
    const int ly = (tid >> 4) + 16 * half;
    const int oy = oy0 + ly;
    if (oy >= out_h || !live) continue;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float4 lo = *reinterpret_cast<const float4*>(&tile[ly + a][lx]);
      const float4 hi = *reinterpret_cast<const float4*>(&tile[ly + a][lx + 4]);
      const float rowv[7] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) acc[q] += rowv[q + cc] * kf[a * 4 + cc];
    }
    float nzv[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t noff = b * (int64_t)out_h * out_w + (int64_t)oy * out_w + ox;
    if (noise) {
      if (full) {
        const float4 nz = *reinterpret_cast<const float4*>(noise + noff);
        nzv[0] = nz.x; nzv[1] = nz.y; nzv[2] = nz.z; nzv[3] = nz.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (ox + q < out_w) nzv[q] = noise[noff + q];
      }
    }
    float res[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = acc[q] + nw * nzv[q];
      if (bias) { v += bv; v = ((v > 0.f) ? v : v * 0.2f) * 1.4142135623730951f; }
      res[q] = post ? v * ps : v;
      if (ox + q < out_w) ymax = fmaxf(ymax, fabsf(res[q]));
    }
    float* yo = y + (bc * out_h + oy) * (int64_t)out_w + ox;
    if (full) {
#if BL_NT & 2
      __builtin_nontemporal_store(rw_f32x4{res[0], res[1], res[2], res[3]}, reinterpret_cast<rw_f32x4*>(yo));
#else
      *reinterpret_cast<float4*>(yo) = make_float4(res[0], res[1], res[2], res[3]);
#endif
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) if (ox + q < out_w) yo[q] = res[q];
    }
  }
  if (y_amax) {       // the workgroup's maximum -> its own slot of the bound (plain store: rw_common.h)
    __shared__ float wmax[4];
    rw_bound_store_block_256(y_amax, ymax, wmax);
  }
}

// The same on the small maps (outputs 8 x 8, 16 x 16, 32 x 32: layers 3, 5, 7): the kernel above gives every plane its
// own 64 x 64 tile, i.e. a workgroup whose threads are 2 - 25 % busy and a launch of batch x channels workgroups that is
// paced by their dispatch (three such launches were 1.0 of the 8 ms of a 250-seed key-statistics sweep, against 0.4 ms
// of HBM time).  Here a workgroup takes 4 / 16 / 64 whole planes -- 4096 outputs, four rows of four per thread -- with
// their zero-bordered (OW + 3)^2 input patches side by side in LDS.  Same taps in the same order: bit-identical.
template <int OW>
__global__ void __launch_bounds__(256) blur_noise_act_small_kernel(
    const float* __restrict__ x, const float* __restrict__ k4, const float* __restrict__ noise,
    const float* __restrict__ nw_ptr, const float* __restrict__ bias, float* __restrict__ y,
    int64_t planes, int channels, const float* __restrict__ post) {
  constexpr int IW = OW + 1;                        // the input plane is IW x IW
  constexpr int PP = 4096 / (OW * OW);              // planes per workgroup
  constexpr int PR = OW + 3, PITCH = OW + 4;        // patch: input rows / columns -1 .. OW + 1, pitch a multiple of 4
  constexpr int CPR = OW / 4;                       // 4-output items per row
  __shared__ float kf[16];
  __shared__ __attribute__((aligned(16))) float tile[PP][PR][PITCH];
  const int tid = threadIdx.x;
  if (tid < 16) {
    const int a = tid >> 2, c = tid & 3;
    kf[tid] = k4[(3 - a) * 4 + (3 - c)];
  }
  const int64_t plane0 = (int64_t)blockIdx.x * PP;
  for (int e = tid; e < PP * PR * PITCH; e += 256) {
    const int pl = e / (PR * PITCH), rem = e - pl * (PR * PITCH);
    const int r = rem / PITCH, c = rem - r * PITCH;
    const int iy = r - 1, ix = c - 1;
    float v = 0.f;
    if (plane0 + pl < planes && iy >= 0 && iy < IW && ix >= 0 && ix < IW)
      v = x[(plane0 + pl) * (int64_t)(IW * IW) + iy * IW + ix];
    (&tile[0][0][0])[e] = v;
  }
  __syncthreads();
  const float nw = noise ? nw_ptr[0] : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int item = tid + 256 * k;                 // PP * OW * CPR == 1024 items
    const int pl = item / (OW * CPR), rem = item - pl * (OW * CPR);
    const int oy = rem / CPR, lx = (rem - oy * CPR) * 4;
    const int64_t bc = plane0 + pl;
    if (bc >= planes) continue;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float4 lo = *reinterpret_cast<const float4*>(&tile[pl][oy + a][lx]);
      const float4 hi = *reinterpret_cast<const float4*>(&tile[pl][oy + a][lx + 4]);
      const float rowv[7] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) acc[q] += rowv[q + cc] * kf[a * 4 + cc];
    }
    const int c = (int)(bc % channels);
    const int64_t b = bc / channels;
    float nzv[4] = {0.f, 0.f, 0.f, 0.f};
    if (noise) {
      const float4 nz = *reinterpret_cast<const float4*>(noise + b * (int64_t)(OW * OW) + oy * OW + lx);
      nzv[0] = nz.x; nzv[1] = nz.y; nzv[2] = nz.z; nzv[3] = nz.w;
    }
    const float bv = bias ? bias[c] : 0.f;
    const float ps = post ? post[bc] : 1.f;
    float res[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = acc[q] + nw * nzv[q];
      if (bias) { v += bv; v = ((v > 0.f) ? v : v * 0.2f) * 1.4142135623730951f; }
      res[q] = post ? v * ps : v;
    }
    *reinterpret_cast<float4*>(y + (bc * OW + oy) * (int64_t)OW + lx) = make_float4(res[0], res[1], res[2], res[3]);
  }
}

extern "C" int rw_absmax_f32(const float* x, long long n, float* out, rw_stream_t stream);
extern "C" int rw_blur_noise_act_amax_f32(const float* x, const float* k4, const float* noise,
                                          const float* noise_w, const float* bias, const float* post_scale,
                                          float* y, int batch, int channels, int out_h, int out_w, float* y_amax,
                                          rw_stream_t stream) {
  RW_CHECK_ARG(x && k4 && y && batch > 0 && channels > 0 && out_h > 0 && out_w > 0);
  RW_CHECK_ARG(!noise || noise_w);
  if (out_h == out_w && (out_w == 8 || out_w == 16 || out_w == 32)) {
    if (y_amax) {       // the small maps' kernel has no reduction of its own: one more pass over a tiny map
      const int rc = rw_blur_noise_act_amax_f32(x, k4, noise, noise_w, bias, post_scale, y, batch, channels, out_h, out_w,
                                                nullptr, stream);
      if (rc) return rc;
      return rw_absmax_f32(y, (long long)batch * channels * out_h * out_w, y_amax, stream);
    }
    const int64_t planes = (int64_t)batch * channels;
    const int64_t wgs = rw_cdiv(planes, 4096 / (out_w * out_w));
    if (wgs > 0x7fffffff) return RW_ERR_UNSUPPORTED;
    hipStream_t s = rw_s(stream);
    if (out_w == 8)
      hipLaunchKernelGGL(blur_noise_act_small_kernel<8>, dim3((unsigned)wgs), dim3(256), 0, s, x, k4, noise, noise_w, bias,
                         y, planes, channels, post_scale);
    else if (out_w == 16)
      hipLaunchKernelGGL(blur_noise_act_small_kernel<16>, dim3((unsigned)wgs), dim3(256), 0, s, x, k4, noise, noise_w, bias,
                         y, planes, channels, post_scale);
    else
      hipLaunchKernelGGL(blur_noise_act_small_kernel<32>, dim3((unsigned)wgs), dim3(256), 0, s, x, k4, noise, noise_w, bias,
                         y, planes, channels, post_scale);
    return RW_LAUNCH_RESULT();
  }
  const int tiles_x = (int)rw_cdiv(out_w, BL_TW), tiles_y = (int)rw_cdiv(out_h, BL_TH);
  const int64_t blocks = (int64_t)batch * channels * tiles_x * tiles_y;
  if (blocks > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  const int64_t n_out = (int64_t)batch * channels * out_h * out_w;
  if (y_amax && blocks > rw_bound_slot_capacity(n_out)) {     // tiles too small for a slot each: measure the result instead
    const int rc = rw_blur_noise_act_amax_f32(x, k4, noise, noise_w, bias, post_scale, y, batch, channels, out_h, out_w,
                                              nullptr, stream);
    if (rc) return rc;
    return rw_absmax_f32(y, (long long)n_out, y_amax, stream);
  }
  hipLaunchKernelGGL(blur_noise_act_kernel, dim3((unsigned)blocks), dim3(256), 0, rw_s(stream), x, k4,
                     noise, noise_w, bias, y, batch, channels, out_h, out_w, tiles_x, tiles_y, post_scale, y_amax);
  const int rc = RW_LAUNCH_RESULT();
  if (rc || !y_amax) return rc;
  return rw_bound_finish(y_amax, blocks, rw_s(stream));
}

extern "C" int rw_blur_noise_act_scaled_f32(const float* x, const float* k4, const float* noise,
                                            const float* noise_w, const float* bias, const float* post_scale,
                                            float* y, int batch, int channels, int out_h, int out_w,
                                            rw_stream_t stream) {
  return rw_blur_noise_act_amax_f32(x, k4, noise, noise_w, bias, post_scale, y, batch, channels, out_h, out_w, nullptr,
                                    stream);
}

extern "C" int rw_blur_noise_act_f32(const float* x, const float* k4, const float* noise,
                                     const float* noise_w, const float* bias, float* y, int batch,
                                     int channels, int out_h, int out_w, rw_stream_t stream) {
  return rw_blur_noise_act_scaled_f32(x, k4, noise, noise_w, bias, nullptr, y, batch, channels, out_h, out_w, stream);
}

// ---------------------------------------------------------------------------------------
// ToRGB (models.py:628-655): 1x1 modulated conv to 3 channels + bias + skip, AI ~1.3 FLOP/B.
// Each thread owns 4 consecutive pixels (16-byte loads along W) and walks the input channels;
// the 3 x C modulated weight row of this image lives in LDS.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) to_rgb_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ style,
                                                     const float* __restrict__ bias,
                                                     const float* __restrict__ skip,
                                                     float* __restrict__ y, int in_ch, int64_t hw,
                                                     float w_scale) {
  extern __shared__ float wm[];  // [3][in_ch]
  const int b = blockIdx.y;
  for (int t = threadIdx.x; t < 3 * in_ch; t += 256) {
    const int i = t % in_ch;
    wm[t] = w_scale * w[t] * style[(int64_t)b * in_ch + i];
  }
  __syncthreads();
  const int64_t hw4 = hw >> 2;
  const float* xb = x + (int64_t)b * in_ch * hw;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < hw4; q += (int64_t)gridDim.x * 256) {
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
#pragma unroll 8
    for (int i = 0; i < in_ch; ++i) {             // (eight 16-byte loads in flight per thread)
      const float4 v = reinterpret_cast<const float4*>(xb + (int64_t)i * hw)[q];
      const float w0 = wm[i], w1 = wm[in_ch + i], w2 = wm[2 * in_ch + i];
      a0.x += w0 * v.x; a0.y += w0 * v.y; a0.z += w0 * v.z; a0.w += w0 * v.w;
      a1.x += w1 * v.x; a1.y += w1 * v.y; a1.z += w1 * v.z; a1.w += w1 * v.w;
      a2.x += w2 * v.x; a2.y += w2 * v.y; a2.z += w2 * v.z; a2.w += w2 * v.w;
    }
    float4 acc[3] = {a0, a1, a2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float bv = bias ? bias[c] : 0.f;
      float4 o = acc[c];
      o.x += bv; o.y += bv; o.z += bv; o.w += bv;
      const int64_t off = ((int64_t)b * 3 + c) * hw4 + q;
      if (skip) {
        const float4 s = reinterpret_cast<const float4*>(skip)[off];
        o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
      }
#if OPS_NTS & 2
      __builtin_nontemporal_store(rw_f32x4{o.x, o.y, o.z, o.w}, reinterpret_cast<rw_f32x4*>(y) + off);
#else
      reinterpret_cast<float4*>(y)[off] = o;
#endif
    }
  }
}

// ToRGB whose channel sums were left behind by the convolution that produced the feature map (rw_dconv3x3_rgb_partial_f32):
// out[b][c][p] = sum_k partial[k][b][c][p] + bias[c] + skip[b][c][p] -- n_part small images instead of a pass over the map.
__global__ void __launch_bounds__(256) rgb_combine_kernel(const float* __restrict__ part, int n_part,
                                                          const float* __restrict__ bias, const float* __restrict__ skip,
                                                          float* __restrict__ y, int64_t n4, int64_t hw4, int64_t stride4) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    float4 a = reinterpret_cast<const float4*>(part)[q];
    for (int k = 1; k < n_part; ++k) {
      const float4 v = reinterpret_cast<const float4*>(part)[q + k * stride4];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (bias) {
      const float bv = bias[(q / hw4) % 3];
      a.x += bv; a.y += bv; a.z += bv; a.w += bv;
    }
    if (skip) {
      const float4 s = reinterpret_cast<const float4*>(skip)[q];
      a.x += s.x; a.y += s.y; a.z += s.z; a.w += s.w;
    }
    reinterpret_cast<float4*>(y)[q] = a;
  }
}

extern "C" int rw_rgb_combine_f32(const float* partials, int n_part, const float* bias, const float* skip, float* y,
                                  int batch, int64_t hw, rw_stream_t stream) {
  RW_CHECK_ARG(partials && y && n_part > 0 && batch > 0 && hw > 0);
  if (hw % 4) return RW_ERR_UNSUPPORTED;
  const int64_t n4 = (int64_t)batch * 3 * hw / 4;
  hipLaunchKernelGGL(rgb_combine_kernel, dim3(rw_stream_grid(n4, 256)), dim3(256), 0, rw_s(stream), partials, n_part, bias,
                     skip, y, n4, hw / 4, n4);
  return RW_LAUNCH_RESULT();
}

// The same for maps whose pixel count is not a multiple of four (the cropped goal maps of a rewriter whose target
// spans a ToRGB: rows of 5 x 7, ...): one pixel per thread, same order of additions.
__global__ void __launch_bounds__(256) to_rgb_scalar_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ style,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ skip, float* __restrict__ y,
                                                            int in_ch, int64_t hw, float w_scale) {
  extern __shared__ float wm[];  // [3][in_ch]
  const int b = blockIdx.y;
  for (int t = threadIdx.x; t < 3 * in_ch; t += 256) wm[t] = w_scale * w[t] * style[(int64_t)b * in_ch + t % in_ch];
  __syncthreads();
  const float* xb = x + (int64_t)b * in_ch * hw;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < hw; q += (int64_t)gridDim.x * 256) {
    float a[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < in_ch; ++i) {
      const float v = xb[(int64_t)i * hw + q];
      a[0] += wm[i] * v; a[1] += wm[in_ch + i] * v; a[2] += wm[2 * in_ch + i] * v;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int64_t off = ((int64_t)b * 3 + c) * hw + q;
      float o = a[c] + (bias ? bias[c] : 0.f);
      if (skip) o += skip[off];
      y[off] = o;
    }
  }
}

extern "C" int rw_to_rgb_f32(const float* x, const float* w, const float* style, const float* bias,
                             const float* skip, float* y, int batch, int in_ch, int64_t hw,
                             float w_scale, rw_stream_t stream) {
  RW_CHECK_ARG(x && w && style && y && batch > 0 && in_ch > 0 && hw > 0);
  if (hw % 4) {
    int gs = (int)rw_cdiv(hw, 256);
    if (gs > 2048) gs = 2048;
    hipLaunchKernelGGL(to_rgb_scalar_kernel, dim3(gs, batch), dim3(256), 3 * in_ch * sizeof(float), rw_s(stream), x, w,
                       style, bias, skip, y, in_ch, hw, w_scale);
    return RW_LAUNCH_RESULT();
  }
  int gx = (int)rw_cdiv(hw / 4, 256);
  const int cap = (256 * 8 + batch - 1) / batch;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(to_rgb_kernel, dim3(gx, batch), dim3(256), 3 * in_ch * sizeof(float),
                     rw_s(stream), x, w, style, bias, skip, y, in_ch, hw, w_scale);
  return RW_LAUNCH_RESULT();
}
