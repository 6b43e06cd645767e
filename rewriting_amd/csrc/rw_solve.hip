// hipcc-keep-packed-fp32
// (csrc/build.sh compiles every other file without packed fp32 instructions.  This one keeps the code generation its multi-step
// goldens were established with: the solve is an Adam trajectory -- a re-ordered rounding moves the 11-step low-rank erase golden from
// 5.0e-5 to 1.02e-4 of the reference's update, against a bar of 1e-4 (profiles/r06ah) -- and it gains nothing from the change: 27.3 ms
// per 2001-step solve either way.  Its packed FMAs run beside its OWN MFMA waves only; tests/test_gpu_overlap.py holds the
// one-launch solver bit-exact beside the upsampling kernels.)
// The rank-constrained projected-gradient solve of ProgressiveGanRewriter.insert
// (rewrite/ganrewrite.py:254-298) for a stride-1 SeqStyleGAN2 layer, as four kernels per
// iteration (arithmetic: SURVEY.md section 10):
//
//   K1 solve_fwd      conv[o][p] = sum_k W[o][k] xcol[k][p]  (split-K fp32 MFMA GEMM, K = 9*Cin)
//                     + partial sums of the demodulation norm   sum_k (s W[o][k] sigma_i)^2
//   K2 solve_mid      demod, noise, bias, leaky-ReLU, L1 loss, dL/dpre, per-channel sum g*conv
//   K3 solve_bwd_adam dW = s * gd xcol^T - c2 * W * sigma^2  (fp32 MFMA GEMM, K = pixels) with the
//                     torch.optim.Adam update applied in the epilogue: the gradient never
//                     touches HBM; W, m, v are each read and written exactly once per step
//   K4 solve_project  W <- W_orth + P(W)  (every piter-th step; also the low-rank-gradient path)
//
// The per-iteration scalars (lr/(1-b1^t), sqrt(1-b2^t)) come from device tables indexed by a
// device-side step counter, so ten iterations can be captured in one HIP graph and replayed
// with no host synchronisation; losses are written to a device array.
#include "rw_common.h"
#include <stdlib.h>

#define SV_KC 16
#define SV_BM 64        // out channels per workgroup (both GEMMs)
#define SV_BN 64        // pixels per workgroup in K1
#ifndef SV_BNK
#define SV_BNK 64       // weight columns per workgroup in K3 (64 or 128: one or two 32-column tiles per wave)
#endif
#define SV_NB (SV_BNK / 64)
#define SV_BJ (SV_KC * SV_BNK / 256)      // B elements a thread stages per chunk
#define SV_DEPTH 4      // chunks of operands in flight per workgroup (K1, K3)

static inline int sv_pp(int p) { return (int)rw_cdiv(p, 64) * 64; }

// Number of positions of the map the convolution writes: h*w for the stride-1 layer, the
// (2h+1) x (2w+1) pre-blur map for an upsampling layer (conv_transpose2d stride 2,
// utils/stylegan2/models.py:315-316).
__host__ __device__ static inline int sv_conv_w(const rw_solve_problem& p) { return p.upsample ? 2 * p.w + 1 : p.w; }
__host__ __device__ static inline int sv_conv_h(const rw_solve_problem& p) { return p.upsample ? 2 * p.h + 1 : p.h; }

// xcol[k = (i, tap)][position]: the input sample that weight tap (ky,kx) of channel i multiplies
// at conv-output position (Y, X); zero outside the crop (quirk Q2) and, for the transposed conv,
// where the parity of (Y-ky, X-kx) does not land on an input sample.
__device__ __forceinline__ float sv_gather(const rw_solve_problem& p, const float* key_i, int tap,
                                           int Y, int X) {
  const int ky = tap / 3, kx = tap - 3 * ky;
  int iy, ix;
  if (p.upsample) {
    const int ty = Y - ky, tx = X - kx;
    if ((ty | tx) < 0 || ((ty | tx) & 1)) return 0.f;
    iy = ty >> 1; ix = tx >> 1;
  } else {
    iy = Y + ky - 1; ix = X + kx - 1;
  }
  if (iy < 0 || iy >= p.h || ix < 0 || ix >= p.w) return 0.f;
  return key_i[iy * p.w + ix];
}

extern "C" int rw_solve_ksplit(int out_ch, int in_ch, int h, int w) {
  const int blocks = (out_ch / SV_BM) * (int)rw_cdiv((int64_t)h * w, SV_BN);
  const int chunks = 9 * in_ch / SV_KC;
  int ks = (int)rw_cdiv(512, blocks > 0 ? blocks : 1);
  if (ks > 32) ks = 32;
  if (ks > chunks) ks = chunks;
  if (ks < 1) ks = 1;
  return ks;
}

// ---------------------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) solve_fwd_kernel(const rw_solve_problem p, int pp) {
  __shared__ __attribute__((aligned(16))) float As[2][SV_KC][SV_BM + 4];
  __shared__ float Bs[2][SV_KC][SV_BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  const int frow = lane >> 5, fcol = lane & 31;
  const int o0 = blockIdx.x * SV_BM;
  const int n0 = blockIdx.y * SV_BN;
  const int ks = blockIdx.z;
  const int CW = sv_conv_w(p);
  const int P = sv_conv_h(p) * CW;
  const int K = 9 * p.in_ch;
  const int chunks = K / SV_KC;
  const int cbeg = (int)((int64_t)chunks * ks / p.ksplit), cend = (int)((int64_t)chunks * (ks + 1) / p.ksplit);

  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) p.step_counter[0] += 1;

  // A staging: thread -> (row = tid>>2, 4 consecutive k); B staging: thread -> pixel tid&63, 4 k rows
  const int arow = tid >> 2, apart = (tid & 3) * 4;
  const int bn = tid & 63, bk0 = tid >> 6;
  const int pix = n0 + bn;
  const bool pix_ok = pix < P;
  const int py = pix_ok ? pix / CW : 0, px = pix_ok ? pix - py * CW : 0;

  // Operands are fetched SV_DEPTH chunks ahead through a ring of register sets: the K range of a workgroup is
  // only 9 - 18 chunks of 8 MFMAs, and with a one-ahead prefetch every chunk cost a full memory latency (13 us
  // for 1.2 GFLOP).  Nothing touches a fetched value before its stash, SV_DEPTH - 1 chunks later.
  float4 areg[SV_DEPTH];
  float sreg[SV_DEPTH][4];
  float breg[SV_DEPTH][4];
  float wsq_acc = 0.f;
  auto fetch = [&](int c, int slot) __attribute__((always_inline)) {
    const int k0 = c * SV_KC;
    areg[slot] = *reinterpret_cast<const float4*>(p.weight + (int64_t)(o0 + arow) * K + k0 + apart);
#pragma unroll
    for (int e = 0; e < 4; ++e) sreg[slot][e] = p.style[(k0 + apart + e) / 9];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + bk0 + 4 * j;
      const int i = k / 9, tap = k - 9 * i;
      breg[slot][j] = pix_ok ? sv_gather(p, p.key + (int64_t)i * p.h * p.w, tap, py, px) : 0.f;
    }
  };
  auto stash = [&](int buf, int slot) __attribute__((always_inline)) {
    const float* av = &areg[slot].x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {                  // demodulation partial, in chunk order
      const float t = p.w_scale * av[e] * sreg[slot][e];
      wsq_acc += t * t;
    }
    As[buf][apart + 0][arow] = areg[slot].x; As[buf][apart + 1][arow] = areg[slot].y;
    As[buf][apart + 2][arow] = areg[slot].z; As[buf][apart + 3][arow] = areg[slot].w;
#pragma unroll
    for (int j = 0; j < 4; ++j) Bs[buf][bk0 + 4 * j][bn] = breg[slot][j];
  };

  rw_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int d = 0; d < SV_DEPTH; ++d)
    if (cbeg + d < cend) fetch(cbeg + d, d);
  if (cbeg < cend) stash(0, 0);
  __syncthreads();
  for (int base = cbeg; base < cend; base += SV_DEPTH) {
#pragma unroll
    for (int d = 0; d < SV_DEPTH; ++d) {
      const int c = base + d;
      if (c < cend) {                                // uniform
        const int buf = d & 1;                       // SV_DEPTH is even
        if (c + SV_DEPTH < cend) fetch(c + SV_DEPTH, d);       // slot d was stashed one chunk ago
#pragma unroll
        for (int kp = 0; kp < SV_KC / 2; ++kp) {
          const float af = As[buf][2 * kp + frow][wm0 + fcol];
          const float bf = Bs[buf][2 * kp + frow][wn0 + fcol];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
        }
        if (c + 1 < cend) stash(buf ^ 1, (d + 1) % SV_DEPTH);
        __syncthreads();
      }
    }
  }
  float* cpart = p.conv + (int64_t)ks * p.out_ch * pp;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = o0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * frow;
    const int n = n0 + wn0 + fcol;
    cpart[(int64_t)o * pp + n] = acc[r];
  }
  if (blockIdx.y == 0) {   // demodulation partial: 4 adjacent lanes share an out channel
    wsq_acc += __shfl_xor(wsq_acc, 1, 64);
    wsq_acc += __shfl_xor(wsq_acc, 2, 64);
    if ((tid & 3) == 0) p.wsq[(int64_t)ks * p.out_ch + o0 + arow] = wsq_acc;
  }
}

// Sum of the split-K partials base[s * stride], s = 0 .. ksplit-1 (ksplit <= 32, rw_solve_ksplit), in that order:
// all loads are issued before the first add -- a loop with a running sum serialises ksplit memory latencies
// (measured: 17 us of the 57 us iteration went to K2 that way).
__device__ __forceinline__ float sv_sum_partials(const float* base, int64_t stride, int ksplit) {
  float v[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) v[s] = s < ksplit ? base[s * stride] : 0.f;
  float acc = 0.f;
#pragma unroll
  for (int s = 0; s < 32; ++s) acc += v[s];          // + 0.f past ksplit leaves the sum unchanged
  return acc;
}

// ---------------------------------------------------------------------------------------
// K2: one workgroup per out channel
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) solve_mid_kernel(const rw_solve_problem p, int pp, float* lpart) {
  // one WAVE per out channel (4 per workgroup): the crop has a few dozen pixels, so a lane owns one or two
  // of them and the two per-channel sums are wave butterflies -- no LDS, no barrier
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= p.out_ch) return;
  const int P = sv_conv_h(p) * sv_conv_w(p);
  const float wsq = sv_sum_partials(p.wsq + o, p.out_ch, p.ksplit);
  const float demod = rsqrtf(wsq + 1e-8f);
  // bias == NULL: the target is the demodulated convolution alone (SeqTinyStyleGanRewriter,
  // rewrite/ganrewrite.py:731-738): no noise, no bias, no activation between it and the loss
  const bool plain = p.bias == nullptr;
  const float nw = plain ? 0.f : p.noise_w[0], bv = plain ? 0.f : p.bias[o];
  const float inv_numel = 1.0f / ((float)p.out_ch * (float)P);
  float lsum = 0.f, tsum = 0.f;
  for (int n = lane; n < pp; n += 64) {
    float gdv = 0.f;
    if (n < P) {
      float conv = sv_sum_partials(p.conv + (int64_t)o * pp + n, (int64_t)p.out_ch * pp, p.ksplit);
      conv *= p.w_scale;
      float out, pre;
      if (plain) {
        pre = conv * demod;
        out = pre;
      } else {
        pre = conv * demod + nw * p.noise[n] + bv;
        out = 1.4142135623730951f * ((pre > 0.f) ? pre : 0.2f * pre);
      }
      const float diff = out - p.val[(int64_t)o * P + n];
      lsum += fabsf(diff);
      const float sgn = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
      const float g_out = sgn * inv_numel;                      // l1_loss backward, mean reduction
      const float g_pre = plain ? g_out
                                : ((pre > 0.f) ? g_out : g_out * 0.2f) * 1.4142135623730951f;  // kernel case 31
      gdv = g_pre * demod;
      tsum += g_pre * conv;
    }
    p.gd[(int64_t)o * pp + n] = gdv;
  }
  lsum = rw_wave_sum(lsum);
  tsum = rw_wave_sum(tsum);
  if (lane == 0) {
    lpart[o] = lsum * inv_numel;
    p.c2[o] = p.w_scale * p.w_scale * demod * demod * demod * tsum;
  }
}

// ---------------------------------------------------------------------------------------
// K2 for upsampling layers: the whole (2h+1)x(2w+1) pre-blur map of one out channel lives in LDS.
//   wide = convT*demod -> blur (upfirdn2d 4x4, pad (1,1), flipped taps) -> + noise + bias -> lrelu
//   loss; g_pre -> blur adjoint (op/upfirdn2d.py:100-115: flipped kernel, pad (2,2)) -> g_wide
//   gd = g_wide*demod,  c2 from sum g_wide*conv
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) solve_mid_up_kernel(const rw_solve_problem p, int pp, float* lpart) {
  extern __shared__ float lds[];          // [P] conv (scaled), [Pout] g_pre
  __shared__ float red[4];
  __shared__ float kf[16];
  const int o = blockIdx.x, tid = threadIdx.x;
  const int CH = 2 * p.h + 1, CW = 2 * p.w + 1, P = CH * CW;
  const int OH = 2 * p.h, OW = 2 * p.w, PO = OH * OW;
  float* conv = lds;
  float* gpre = lds + P;
  if (tid < 16) kf[tid] = p.blur_k[(3 - (tid >> 2)) * 4 + (3 - (tid & 3))];
  const float wsq = sv_sum_partials(p.wsq + o, p.out_ch, p.ksplit);
  const float demod = rsqrtf(wsq + 1e-8f);
  for (int n = tid; n < P; n += 256)
    conv[n] = sv_sum_partials(p.conv + (int64_t)o * pp + n, (int64_t)p.out_ch * pp, p.ksplit) * p.w_scale;
  __syncthreads();
  const float nw = p.noise_w[0], bv = p.bias[o];
  const float inv_numel = 1.0f / ((float)p.out_ch * (float)PO);
  float lsum = 0.f;
  for (int n = tid; n < PO; n += 256) {
    const int y = n / OW, x = n - y * OW;
    float b = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int Y = y + a - 1;
      if (Y < 0 || Y >= CH) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int X = x + c - 1;
        if (X < 0 || X >= CW) continue;
        b += (conv[Y * CW + X] * demod) * kf[a * 4 + c];
      }
    }
    const float pre = b + nw * p.noise[n] + bv;
    const float out = 1.4142135623730951f * ((pre > 0.f) ? pre : 0.2f * pre);
    const float diff = out - p.val[(int64_t)o * PO + n];
    lsum += fabsf(diff);
    const float sgn = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
    const float g_out = sgn * inv_numel;
    gpre[n] = ((pre > 0.f) ? g_out : g_out * 0.2f) * 1.4142135623730951f;
  }
  __syncthreads();
  float tsum = 0.f;
  for (int n = tid; n < pp; n += 256) {
    float gdv = 0.f;
    if (n < P) {
      const int Y = n / CW, X = n - Y * CW;
      float gw = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int y = Y - a + 1;
        if (y < 0 || y >= OH) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int x = X - c + 1;
          if (x < 0 || x >= OW) continue;
          gw += gpre[y * OW + x] * kf[a * 4 + c];
        }
      }
      gdv = gw * demod;
      tsum += gw * conv[n];
    }
    p.gd[(int64_t)o * pp + n] = gdv;
  }
  lsum = rw_block_sum_256(lsum, red);
  tsum = rw_block_sum_256(tsum, red);
  if (tid == 0) {
    lpart[o] = lsum * inv_numel;
    p.c2[o] = p.w_scale * p.w_scale * demod * demod * demod * tsum;
  }
}

// ---------------------------------------------------------------------------------------
// Adam, as torch.optim.Adam's single-tensor path computes it (rewrite/ganrewrite.py:277,287)
// ---------------------------------------------------------------------------------------
// omb1 / omb2 are (1 - beta) evaluated in DOUBLE on the host and rounded once, as torch passes them
// (1.0f - 0.999f differs from float(1 - 0.999) by 1.3e-5 relative).
__device__ __forceinline__ void adam_update(float g, float& w, float& m, float& v, float omb1,
                                            float b2, float omb2, float eps, float step_size, float bc2s) {
  m = m + (g - m) * omb1;                      // exp_avg.lerp_(grad, 1 - beta1)
  v = v * b2 + (omb2 * g) * g;                 // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
  const float denom = sqrtf(v) / bc2s + eps;   // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
  w = w + (-step_size * m) / denom;            // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// ---------------------------------------------------------------------------------------
// K3
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) solve_bwd_adam_kernel(const rw_solve_problem p, int pp,
                                                             const float* lpart) {
  __shared__ __attribute__((aligned(16))) float As[2][SV_KC][SV_BM + 4];
  __shared__ float Bs[2][SV_KC][SV_BNK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * (SV_BNK / 2);     // wave tile 32 (o) x SV_BNK/2 (k)
  const int frow = lane >> 5, fcol = lane & 31;
  const int o0 = blockIdx.x * SV_BM;
  const int k0 = blockIdx.y * SV_BNK;
  const int CW = sv_conv_w(p);
  const int P = sv_conv_h(p) * CW;
  const int K = 9 * p.in_ch;
  const int it = p.step_counter[0];

  if (blockIdx.x == 0 && blockIdx.y == 0 && wave == 0) {   // deterministic loss reduction
    float l = 0.f;
    for (int o = lane; o < p.out_ch; o += 64) l += lpart[o];
    l = rw_wave_sum(l);
    if (lane == 0) p.losses[it] = l;
  }

  const int arow = tid >> 2, apart = (tid & 3) * 4;
  const int bcol = tid & (SV_BNK - 1), bp0 = tid / SV_BNK;
  const int kmine = k0 + bcol;
  const bool col_ok = kmine < K;
  const int ci = col_ok ? kmine / 9 : 0, ctap = col_ok ? kmine - 9 * ci : 0;
  const float* kch = p.key + (int64_t)ci * p.h * p.w;

  // operand ring as in K1 (the K dimension here is the crop: 4 - 16 chunks)
  float4 areg[SV_DEPTH];
  float breg[SV_DEPTH][SV_BJ];
  auto fetch = [&](int c, int slot) __attribute__((always_inline)) {
    const int p0 = c * SV_KC;
    areg[slot] = *reinterpret_cast<const float4*>(p.gd + (int64_t)(o0 + arow) * pp + p0 + apart);
#pragma unroll
    for (int j = 0; j < SV_BJ; ++j) {
      const int n = p0 + bp0 + (256 / SV_BNK) * j;
      float v = 0.f;
      if (n < P && col_ok) {
        const int y = n / CW;
        v = sv_gather(p, kch, ctap, y, n - y * CW);
      }
      breg[slot][j] = v;
    }
  };
  auto stash = [&](int buf, int slot) __attribute__((always_inline)) {
    As[buf][apart + 0][arow] = areg[slot].x; As[buf][apart + 1][arow] = areg[slot].y;
    As[buf][apart + 2][arow] = areg[slot].z; As[buf][apart + 3][arow] = areg[slot].w;
#pragma unroll
    for (int j = 0; j < SV_BJ; ++j) Bs[buf][bp0 + (256 / SV_BNK) * j][bcol] = breg[slot][j];
  };

  rw_f32x16 acc[SV_NB];
#pragma unroll
  for (int b = 0; b < SV_NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  const int chunks = (P + SV_KC - 1) / SV_KC;
#pragma unroll
  for (int d = 0; d < SV_DEPTH; ++d)
    if (d < chunks) fetch(d, d);
  stash(0, 0);
  __syncthreads();
  for (int base = 0; base < chunks; base += SV_DEPTH) {
#pragma unroll
    for (int d = 0; d < SV_DEPTH; ++d) {
      const int c = base + d;
      if (c < chunks) {                              // uniform
        const int buf = d & 1;
        if (c + SV_DEPTH < chunks) fetch(c + SV_DEPTH, d);
#pragma unroll
        for (int kp = 0; kp < SV_KC / 2; ++kp) {
          const float af = As[buf][2 * kp + frow][wm0 + fcol];
#pragma unroll
          for (int b = 0; b < SV_NB; ++b) {
            const float bf = Bs[buf][2 * kp + frow][wn0 + 32 * b + fcol];
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[b], 0, 0, 0);
          }
        }
        if (c + 1 < chunks) stash(buf ^ 1, (d + 1) % SV_DEPTH);
        __syncthreads();
      }
    }
  }

  // Epilogue: gradient, then Adam in place.  W, m, v of the 16 rows a lane owns are fetched in one
  // batch (48 loads in flight) before anything is stored: weight/exp_avg/exp_avg_sq are read AND
  // written here, so element-by-element code serialises into 32 dependent L2 round trips per wave.
  // (Requesting them before the GEMM instead -- they do not depend on it -- was measured slower: 30 vs 21 us.)
  const float step_size = p.step_size[it], bc2s = p.bc2_sqrt[it];
  float c2v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) c2v[r] = p.c2[o0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * frow];
#pragma unroll
  for (int b = 0; b < SV_NB; ++b) {
    const int k = k0 + wn0 + 32 * b + fcol;
    if (k >= K) continue;
    const int i = k / 9;
    const float sg = p.style[i];
    const float sig2 = sg * sg;
    const int64_t base = (int64_t)(o0 + wm0 + 4 * frow) * K + k;
    float wv[16], mv[16], vv[16];
    const bool adam = !(p.low_rank_gradient || p.linear_insert);
#pragma unroll
    for (int r = 0; r < 16; ++r) wv[r] = p.weight[base + (int64_t)((r & 3) + 8 * (r >> 2)) * K];
    if (adam) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mv[r] = p.exp_avg[base + (int64_t)((r & 3) + 8 * (r >> 2)) * K];
#pragma unroll
      for (int r = 0; r < 16; ++r) vv[r] = p.exp_avg_sq[base + (int64_t)((r & 3) + 8 * (r >> 2)) * K];
    }
    float g[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = p.w_scale * acc[b][r] - c2v[r] * wv[r] * sig2;
    if (!adam) {
#pragma unroll
      for (int r = 0; r < 16; ++r) p.grad[base + (int64_t)((r & 3) + 8 * (r >> 2)) * K] = g[r];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        adam_update(g[r], wv[r], mv[r], vv[r], p.one_minus_beta1, p.beta2, p.one_minus_beta2, p.eps, step_size, bc2s);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t idx = base + (int64_t)((r & 3) + 8 * (r >> 2)) * K;
        p.weight[idx] = wv[r]; p.exp_avg[idx] = mv[r]; p.exp_avg_sq[idx] = vv[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// K4 and rw_project_weight: P(W)[o][i][t] = sum_r (sum_j W[o][j][t] d[r][j]) d[r][i]
// (projected_conv, rewrite/ganrewrite.py:806-813).  One workgroup per out channel; W[o] is held
// in LDS, each rank costs `taps` block reductions.
//   MODE 0: out = (base ? base : 0) + P(src)                    (projection / ortho / zero())
//   MODE 1: g = P(grad); Adam(g) on weight                       (low_rank_gradient=True)
//   MODE 2: linear_insert (rewrite/ganrewrite.py:201-252): dLambda[o][r][t] = sum_i grad[o][i][t] d[r][i];
//           Adam on Lambda (state in exp_avg / exp_avg_sq, first out_ch*rank*9 floats; Lambda in
//           p.lambda); weight = base(W0) + sum_r Lambda[o][r][t] d[r][i]
// ---------------------------------------------------------------------------------------
#define SV_MAX_TAPS 9
template <int MODE>
__global__ void __launch_bounds__(256) project_kernel(const float* __restrict__ src,
                                                      const float* __restrict__ ctx,
                                                      const float* __restrict__ base,
                                                      float* __restrict__ out, int in_ch, int taps,
                                                      int rank, rw_solve_problem p) {
  extern __shared__ float lds[];            // [in_ch*taps] source row, [in_ch*taps] projection
  __shared__ float red[4];
  __shared__ float cosv[SV_MAX_TAPS];
  const int o = blockIdx.x, tid = threadIdx.x;
  const int rowlen = in_ch * taps;
  float* srow = lds;
  float* prow = lds + rowlen;
  const float* s = src + (int64_t)o * rowlen;
  for (int e = tid; e < rowlen; e += 256) { srow[e] = s[e]; prow[e] = 0.f; }
  __syncthreads();
  for (int r = 0; r < rank; ++r) {
    const float* d = ctx + (int64_t)r * in_ch;
    float part[SV_MAX_TAPS];
    float lam_new = 0.f;
#pragma unroll
    for (int t = 0; t < SV_MAX_TAPS; ++t) part[t] = 0.f;
    for (int i = tid; i < in_ch; i += 256) {
      const float dv = d[i];
#pragma unroll
      for (int t = 0; t < SV_MAX_TAPS; ++t)
        if (t < taps) part[t] += srow[i * taps + t] * dv;
    }
#pragma unroll
    for (int t = 0; t < SV_MAX_TAPS; ++t) {
      if (t < taps) {
        const float v = rw_block_sum_256(part[t], red);
        if (tid == 0) cosv[t] = v;
      }
    }
    __syncthreads();
    if (MODE == 2) {                      // cosv = dLambda[o][r][:]; one thread per tap steps Adam on Lambda
      if (tid < taps) {
        const int it = p.step_counter[0];
        const int64_t li = ((int64_t)o * rank + r) * taps + tid;
        float lv = p.lambda[li], m = p.exp_avg[li], v = p.exp_avg_sq[li];
        adam_update(cosv[tid], lv, m, v, p.one_minus_beta1, p.beta2, p.one_minus_beta2, p.eps,
                    p.step_size[it], p.bc2_sqrt[it]);
        p.lambda[li] = lv; p.exp_avg[li] = m; p.exp_avg_sq[li] = v;
        cosv[tid] = lv;
      }
      __syncthreads();
    }
    (void)lam_new;
    for (int i = tid; i < in_ch; i += 256) {
      const float dv = d[i];
      for (int t = 0; t < taps; ++t) prow[i * taps + t] += cosv[t] * dv;
    }
    __syncthreads();
  }
  if (MODE == 0 || MODE == 2) {
    const float* b = base ? base + (int64_t)o * rowlen : nullptr;
    for (int e = tid; e < rowlen; e += 256) out[(int64_t)o * rowlen + e] = (b ? b[e] : 0.f) + prow[e];
  } else {
    const int it = p.step_counter[0];
    const float step_size = p.step_size[it], bc2s = p.bc2_sqrt[it];
    for (int e = tid; e < rowlen; e += 256) {
      const int64_t idx = (int64_t)o * rowlen + e;
      float wv = p.weight[idx], m = p.exp_avg[idx], v = p.exp_avg_sq[idx];
      adam_update(prow[e], wv, m, v, p.one_minus_beta1, p.beta2, p.one_minus_beta2, p.eps, step_size, bc2s);
      p.weight[idx] = wv; p.exp_avg[idx] = m; p.exp_avg_sq[idx] = v;
    }
  }
}

extern "C" int rw_project_weight_f32(const float* w, const float* context, const float* base,
                                     float* out, int out_ch, int in_ch, int taps, int rank,
                                     float scale_w, rw_stream_t stream) {
  (void)scale_w;
  RW_CHECK_ARG(w && context && out && out_ch > 0 && in_ch > 0 && taps > 0 && taps <= SV_MAX_TAPS && rank > 0);
  const size_t lds = 2 * (size_t)in_ch * taps * sizeof(float);
  if (lds > 64 * 1024) return RW_ERR_UNSUPPORTED;
  rw_solve_problem dummy = {};
  hipLaunchKernelGGL(project_kernel<0>, dim3(out_ch), dim3(256), lds, rw_s(stream), w, context, base,
                     out, in_ch, taps, rank, dummy);
  return RW_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------
// The whole solve in ONE launch, with the weight never leaving the compute unit (round 3).
//
// `insert` decomposes over OUT channels: the convolution, the demodulation norm, the loss terms, both parts of the
// gradient, Adam and the rank-r projection of out-channel o read and write W[o, :, :] only --
// the loss VALUE is the one quantity that sums over o, and nothing in the update depends on it.  So a workgroup owns
// two out-channels for all niter iterations; thread i of it owns input channel i: the 2 x 9 weights W[o][i][:], their
// Adam moments and the ortho part sit in its registers (72), the key crop sits in LDS, and per iteration
//   A  partial conv of my channel at every position, summed over the wave by a transposed DPP tree and over the
//      waves through LDS -- likewise the demodulation norm;
//   B  one wave per out-channel: demod, noise, bias, leaky ReLU, L1 loss, dL/dpre, sum g*conv, gd -> LDS;
//   C  dW of my 18 weights (the same LDS reads), the demodulation term, [projection of the gradient], Adam,
//      [projection of the weight];
// three workgroup barriers, NO global-memory traffic besides one loss value per out-channel, NO kernel boundary, no
// inter-workgroup synchronisation at all (so no residency requirement and nothing to deadlock).  The kernel is bound
// by the INSTRUCTIONS a wave issues (one per ~4 cycles), not by FLOPs, so the layout is chosen for instruction count:
//   * a channel's crop is stored zero-padded with row pitch w+1 -- one zero row above and below, one zero column that
//     serves as the right border of row y and the left border of row y+1 -- so the nine taps of position n are at the
//     FIXED offsets ky*(w+1) + kx from n, no bounds logic at all; positions are visited linearly over the padded rows,
//     two at a time (the values computed at pad columns are never read, gd there is 0);
//   * the two positions of a pair share their key reads: per kernel row four adjacent floats (two ds_read2_b32)
//     feed six packed FMAs (v_pk_fma_f32: both out-channels in one instruction);
//   * the four sums of a pair (2 positions x 2 out-channels) are reduced over the 64 lanes TOGETHER: two exchange
//     steps leave lane l with value (l & 3) summed over its quad, row rotations by 4 and 8 and the gfx950 row / half
//     swaps (v_permlane16_swap, v_permlane32_swap) finish it in 15 instructions instead of 4 x 6.
// The channel row pitch is odd, so consecutive channels fall on different LDS banks.
// Limits: stride-1 target (with or without the noise / bias / activation stage), in_ch % 64 == 0, in_ch <= 512,
// out_ch % 2 == 0, rank <= 8, in_ch * ((h+2)(w+1) + 3 | 1) floats + ~6 KB within the 160 KB of LDS (512 channels: 5 x 8,
// 6 x 8, 4 x 11 crops; 256 channels: up to 10 x 12).  Everything else (upsampling targets, linear_insert, larger crops)
// takes rw_solve_step_f32.
// ---------------------------------------------------------------------------------------
#define SVP_RMAX 8
typedef float svp_f2 __attribute__((ext_vector_type(2)));
typedef float svp_f4 __attribute__((ext_vector_type(4)));
typedef unsigned svp_u2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float svp_dpp(float v) {        // the DPP-selected lane's v (controls that read valid lanes)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float svp_dpp_add(float v) {
  // v + (the DPP-selected lane's v, 0 where the row is masked or the source invalid)
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// sum over the 64 lanes of a wave; the total is in lane 63 (rocPRIM's DPP sequence for gfx9)
__device__ __forceinline__ float svp_wave_sum63(float v) {
  v = svp_dpp_add<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
  v = svp_dpp_add<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
  v = svp_dpp_add<0x141, 0xf>(v);     // row_half_mirror
  v = svp_dpp_add<0x140, 0xf>(v);     // row_mirror: every lane of a row of 16 holds the row's sum
  v = svp_dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
  v = svp_dpp_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3
  return v;
}
// v[l] + v[l ^ 16] and then + [l ^ 32]: the rows of 16 and the halves of the wave exchanged by the gfx950 swaps
__device__ __forceinline__ float svp_rows_sum(float v) {
  svp_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Four values per lane (a: position n, b: position n+1; [0], [1]: the two out-channels) summed over the wave; every
// lane l returns the total of value (l & 3), numbered 2 * position + channel.
__device__ __forceinline__ float svp_wave_sum4(svp_f2 a, svp_f2 b, bool odd, bool hi) {
  const float k01 = odd ? a[1] : a[0], s01 = odd ? a[0] : a[1];
  const float k23 = odd ? b[1] : b[0], s23 = odd ? b[0] : b[1];
  const float v01 = k01 + svp_dpp<0xB1>(s01);              // quad_perm [1,0,3,2]: even lanes hold channel 0, odd 1
  const float v23 = k23 + svp_dpp<0xB1>(s23);
  const float kk = hi ? v23 : v01, ss = hi ? v01 : v23;
  float v = kk + svp_dpp<0x4E>(ss);                        // quad_perm [2,3,0,1]: lanes 2, 3 of a quad hold n+1
  v += svp_dpp<0x124>(v);                                  // row_ror:4
  v += svp_dpp<0x128>(v);                                  // row_ror:8  -> the row's sum of value (l & 3)
  return svp_rows_sum(v);
}

__host__ __device__ static inline int svp_row_pitch(int h, int w) { return ((h + 2) * (w + 1) + 3) | 1; }
__host__ __device__ static inline int svp_positions(int h, int w) { return (h * (w + 1) + 1) & ~1; }
__host__ __device__ static inline int svp_pair_row(int h, int w) { return (2 * svp_positions(h, w) + 3) & ~3; }
__host__ __device__ static inline int svp_part_floats(int nw, int h, int w) {
  const int a = nw * svp_pair_row(h, w), b = nw * 18 * SVP_RMAX;     // wave partials; aliased by the projection's
  return a > b ? a : b;
}

__global__ void __launch_bounds__(512) solve_persistent_kernel(const rw_solve_problem p, int it0, int it1, int niter,
                                                                int piter, int low_rank_insert, float* lpart_all) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nth = blockDim.x, NW = nth >> 6;
  const int P = p.h * p.w, WP = p.w + 1;
  const int PS = svp_row_pitch(p.h, p.w);           // floats per channel row
  const int NP = svp_positions(p.h, p.w);           // positions visited: h padded rows, rounded up to a pair
  const int PA = svp_pair_row(p.h, p.w);            // floats per (position, channel)-interleaved row
  float* Ks = lds;                                  // [in_ch][PS]   zero-padded key crops
  float* part = Ks + (size_t)p.in_ch * PS;          // [NW][PA]      per-wave conv sums (n, o); the projection's scratch
  float* gds = part + svp_part_floats(NW, p.h, p.w);   // [PA]       g_pre * demod at (n, o); 0 at pad positions
  float* vals = gds + PA;                           // [PA]          the target values at (n, o)
  float* nz = vals + PA;                            // [PA]          noise_w * noise at n
  float* wq = nz + PA;                              // [NW][2]       demodulation partials
  float* chn = wq + 16;                             // [2][4]        c2 per out-channel
  int* nmap = reinterpret_cast<int*>(chn + 8);      // [P]           position index n of crop element q
  float* red = part;
  const int o0 = 2 * blockIdx.x;
  const int i = tid;                                // my input channel
  const int K = 9 * p.in_ch;
  const bool plain = p.bias == nullptr;
  const bool constrained = p.context != nullptr && p.rank > 0;
  const bool odd = lane & 1, hi = lane & 2;

  for (int e = tid; e < p.in_ch * PS; e += nth) Ks[e] = 0.f;
  for (int e = tid; e < 3 * PA; e += nth) gds[e] = 0.f;       // gds, vals, nz
  __syncthreads();
  for (int e = tid; e < p.in_ch * P; e += nth) {
    const int c = e / P, q = e - c * P, y = q / p.w, x = q - y * p.w;
    Ks[c * PS + (y + 1) * WP + x + 1] = p.key[e];
  }
  for (int q = tid; q < P; q += nth) {
    const int y = q / p.w, n = y * WP + (q - y * p.w);
    nmap[q] = n;
    vals[2 * n] = p.val[(int64_t)o0 * P + q];
    vals[2 * n + 1] = p.val[(int64_t)(o0 + 1) * P + q];
    if (!plain) nz[n] = p.noise_w[0] * p.noise[q];
  }
  svp_f2 W[9], M[9], V[9], Or[9];                   // [tap]{out-channel o0, o0 + 1}
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int64_t idx = (int64_t)(o0 + o) * K + i * 9 + t;
      W[t][o] = p.weight[idx]; M[t][o] = p.exp_avg[idx]; V[t][o] = p.exp_avg_sq[idx];
      Or[t][o] = (low_rank_insert && p.ortho) ? p.ortho[idx] : 0.f;
    }
  float dctx[SVP_RMAX];
#pragma unroll
  for (int r = 0; r < SVP_RMAX; ++r) dctx[r] = (constrained && r < p.rank) ? p.context[(int64_t)r * p.in_ch + i] : 0.f;
  const float sg = p.style[i], sig2 = sg * sg, s = p.w_scale;
  const float inv_numel = 1.0f / ((float)p.out_ch * (float)P);
  const float* kb = Ks + i * PS;
  const float bias0 = plain ? 0.f : p.bias[o0], bias1 = plain ? 0.f : p.bias[o0 + 1];
  __syncthreads();

  // x[t][o] <- sum_r (sum_i x[t][o][i] d[r][i]) d[r][i] over the workgroup's channels; two barriers
  auto project_rows = [&](svp_f2 (&x)[9]) __attribute__((always_inline)) {
    for (int r = 0; r < p.rank; ++r) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const float sum = svp_wave_sum63(x[t][o] * dctx[r]);
          if (lane == 63) red[(wave * SVP_RMAX + r) * 18 + 2 * t + o] = sum;
        }
    }
    __syncthreads();
    svp_f2 out[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) out[t] = svp_f2{0.f, 0.f};
    for (int r = 0; r < p.rank; ++r) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        svp_f2 c = {0.f, 0.f};
        for (int w2 = 0; w2 < NW; ++w2) c += *reinterpret_cast<const svp_f2*>(red + (w2 * SVP_RMAX + r) * 18 + 2 * t);
        out[t] += c * dctx[r];
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) x[t] = out[t];
    __syncthreads();
  };

  float ss_next = p.step_size[it0 < it1 ? it0 : 0], bc_next = p.bc2_sqrt[it0 < it1 ? it0 : 0];
  for (int it = it0; it < it1; ++it) {
    // Adam's step tables one iteration ahead of their use: the load's latency is never waited for
    const float step_size = ss_next, bc2s = bc_next;
    if (it + 1 < it1) { ss_next = p.step_size[it + 1]; bc_next = p.bc2_sqrt[it + 1]; }
    // ---- A: partial convolution of my channel, the demodulation partial
    {
      svp_f2 wq2 = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const svp_f2 a = (s * W[t]) * sg;
        wq2 += a * a;
      }
      const float wq0 = svp_wave_sum63(wq2[0]), wq1 = svp_wave_sum63(wq2[1]);
      if (lane == 63) { wq[wave * 2] = wq0; wq[wave * 2 + 1] = wq1; }
      for (int n = 0; n < NP; n += 2) {
        svp_f2 a = {0.f, 0.f}, b = {0.f, 0.f};
        float kk[3][4];                                       // all twelve reads in flight before the first FMA
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* r = kb + n + ky * WP;
#pragma unroll
          for (int c = 0; c < 4; ++c) kk[ky][c] = r[c];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float k0 = kk[ky][0], k1 = kk[ky][1], k2 = kk[ky][2], k3 = kk[ky][3];
          a = __builtin_elementwise_fma(W[3 * ky], svp_f2{k0, k0}, a);
          b = __builtin_elementwise_fma(W[3 * ky], svp_f2{k1, k1}, b);
          a = __builtin_elementwise_fma(W[3 * ky + 1], svp_f2{k1, k1}, a);
          b = __builtin_elementwise_fma(W[3 * ky + 1], svp_f2{k2, k2}, b);
          a = __builtin_elementwise_fma(W[3 * ky + 2], svp_f2{k2, k2}, a);
          b = __builtin_elementwise_fma(W[3 * ky + 2], svp_f2{k3, k3}, b);
        }
        const float v = svp_wave_sum4(a, b, odd, hi);
        if (lane < 4) part[wave * PA + 2 * n + lane] = v;
      }
    }
    __syncthreads();
    // ---- B: wave o finishes out-channel o0 + o
    for (int o = wave; o < 2; o += NW) {            // (a 64-channel layer has one wave: it takes both)
      float wsq = 0.f;
      for (int w2 = 0; w2 < NW; ++w2) wsq += wq[w2 * 2 + o];
      const float demod = rsqrtf(wsq + 1e-8f);
      const float bv = o ? bias1 : bias0;
      float lsum = 0.f, tsum = 0.f;
      for (int q = lane; q < P; q += 64) {
        const int n2 = 2 * nmap[q] + o;
        float conv = 0.f;
        for (int w2 = 0; w2 < NW; ++w2) conv += part[w2 * PA + n2];
        conv *= s;
        float out, pre;
        if (plain) { pre = conv * demod; out = pre; }
        else {
          pre = conv * demod + nz[n2 >> 1] + bv;
          out = 1.4142135623730951f * ((pre > 0.f) ? pre : 0.2f * pre);
        }
        const float diff = out - vals[n2];
        lsum += fabsf(diff);
        const float sgn = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
        const float g_out = sgn * inv_numel;
        const float g_pre = plain ? g_out : ((pre > 0.f) ? g_out : g_out * 0.2f) * 1.4142135623730951f;
        gds[n2] = g_pre * demod;
        tsum += g_pre * conv;
      }
      lsum = svp_wave_sum63(lsum);
      tsum = svp_wave_sum63(tsum);
      if (lane == 63) {
        chn[o * 4] = s * s * demod * demod * demod * tsum;
        lpart_all[(int64_t)it * p.out_ch + o0 + o] = lsum * inv_numel;
      }
    }
    __syncthreads();
    // ---- C: gradient of my 18 weights, Adam, projections
    {
      svp_f2 G[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) G[t] = svp_f2{0.f, 0.f};
      for (int n = 0; n < NP; n += 2) {
        const svp_f4 gq = *reinterpret_cast<const svp_f4*>(gds + 2 * n);      // (n, o0) (n, o1) (n+1, o0) (n+1, o1)
        const svp_f2 ga = {gq[0], gq[1]}, gb = {gq[2], gq[3]};
        float kk[3][4];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* r = kb + n + ky * WP;
#pragma unroll
          for (int c = 0; c < 4; ++c) kk[ky][c] = r[c];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float k0 = kk[ky][0], k1 = kk[ky][1], k2 = kk[ky][2], k3 = kk[ky][3];
          G[3 * ky] = __builtin_elementwise_fma(ga, svp_f2{k0, k0}, G[3 * ky]);
          G[3 * ky] = __builtin_elementwise_fma(gb, svp_f2{k1, k1}, G[3 * ky]);
          G[3 * ky + 1] = __builtin_elementwise_fma(ga, svp_f2{k1, k1}, G[3 * ky + 1]);
          G[3 * ky + 1] = __builtin_elementwise_fma(gb, svp_f2{k2, k2}, G[3 * ky + 1]);
          G[3 * ky + 2] = __builtin_elementwise_fma(ga, svp_f2{k2, k2}, G[3 * ky + 2]);
          G[3 * ky + 2] = __builtin_elementwise_fma(gb, svp_f2{k3, k3}, G[3 * ky + 2]);
        }
      }
      const svp_f2 c2 = {chn[0], chn[4]};
#pragma unroll
      for (int t = 0; t < 9; ++t) G[t] = s * G[t] - c2 * W[t] * sig2;
      if (p.low_rank_gradient) project_rows(G);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          float wv = W[t][o], mv = M[t][o], vv = V[t][o];
          adam_update(G[t][o], wv, mv, vv, p.one_minus_beta1, p.beta2, p.one_minus_beta2, p.eps, step_size, bc2s);
          W[t][o] = wv; M[t][o] = mv; V[t][o] = vv;
        }
      if (low_rank_insert && (it % piter == 0 || it == niter - 1)) {
        svp_f2 pw[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) pw[t] = W[t];
        project_rows(pw);
#pragma unroll
        for (int t = 0; t < 9; ++t) W[t] = Or[t] + pw[t];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int64_t idx = (int64_t)(o0 + o) * K + i * 9 + t;
      p.weight[idx] = W[t][o]; p.exp_avg[idx] = M[t][o]; p.exp_avg_sq[idx] = V[t][o];
    }
}

// ---------------------------------------------------------------------------------------
// The same solve for key crops that do NOT fit the LDS (round 4: the watermark erase solves on whole 16 x 16 maps of
// 512 channels, 633 KB of key).  Thread i of a workgroup reads only ITS channel's crop, so the crop need not be shared
// at all: it is streamed from a position-major, zero-padded copy in global memory (keyT[e][i], e = the padded index of
// the resident kernel's layout; built once per launch range by solve_key_transpose_kernel, L2-resident: every
// workgroup reads the same 0.6 MB) ROW BY ROW into registers -- four row buffers of w + 4 values, the row three ahead in
// flight while a row is convolved, every load 256 contiguous bytes per wave.  Positions are paired within a padded row
// (the last pair of an odd row re-computes the first position of the next one: the same value to the same place in
// the forward phase, masked out of the gradient phase).
// With low_rank_gradient (the erase's setting) the gradient phase needs no key at all: the projected gradient is
//   P(dW)[o][i][t] = sum_r (s sum_n g[o][n] kd_r[n + t] - c2[o] sum_j W[o][j][t] sig2[j] d_r[j]) d_r[i],
//   kd_r[e] = sum_i d_r[i] key[i][e]   (a one-channel map per context row, computed once per launch),
// i.e. 18 correlations of 256 positions per context row instead of 2.4 M multiply-adds per out-channel pair.
// Everything else -- phase B, Adam, the projections, the per-channel loss parts -- is the resident kernel's.
// Limits: as solve_persistent_kernel but w <= 16 and the (small) LDS footprint below instead of the crop's.
// ---------------------------------------------------------------------------------------
#define SVS_WP_MAX 17
#define SVS_ROW (SVS_WP_MAX + 3)

__global__ void __launch_bounds__(256) solve_key_transpose_kernel(const float* __restrict__ key, float* __restrict__ keyT,
                                                                  int in_ch, int h, int w) {
  const int PS = svp_row_pitch(h, w), WP = w + 1;
  const int64_t total = (int64_t)PS * in_ch;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx / in_ch), i = (int)(idx - (int64_t)e * in_ch);
    const int yy = e / WP - 1, xx = e - (e / WP) * WP - 1;        // crop coordinates of padded index e
    keyT[idx] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? key[((int64_t)i * h + yy) * w + xx] : 0.f;
  }
}

__global__ void __launch_bounds__(512) solve_stream_kernel(const rw_solve_problem p, int it0, int it1, int niter, int piter,
                                                            int low_rank_insert, float* lpart_all,
                                                            const float* __restrict__ keyT) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nth = blockDim.x, NW = nth >> 6;
  const int P = p.h * p.w, WP = p.w + 1;
  const int PS = svp_row_pitch(p.h, p.w);
  const int NP = svp_positions(p.h, p.w);
  const int PA = svp_pair_row(p.h, p.w);
  const bool lrg = p.low_rank_gradient != 0;
  float* part = lds;                                // [NW][PA]      per-wave conv sums (n, o); the projection's scratch
  float* gds = part + svp_part_floats(NW, p.h, p.w);   // [PA]       g_pre * demod at (n, o); 0 at pad positions
  float* vals = gds + PA;
  float* nz = vals + PA;
  float* wq = nz + PA;                              // [NW][2]
  float* chn = wq + 16;                             // [2][4]
  float* cgs = chn + 8;                             // [SVP_RMAX][18]  s * correlation of g with kd_r, per (tap, out-channel)
  int* nmap = reinterpret_cast<int*>(cgs + SVP_RMAX * 18);   // [P]
  float* kd = reinterpret_cast<float*>(nmap + P);   // [rank][PS]    low_rank_gradient only
  // Adam's moments of my 18 weights live in LDS (column tid of [36][threads]): the four row buffers need the registers
  float* mv = kd + (lrg ? p.rank : 0) * PS + tid;   // [2 * 18][nth]
  float* red = part;
  const int o0 = 2 * blockIdx.x;
  const int i = tid;
  const int K = 9 * p.in_ch;
  const bool plain = p.bias == nullptr;
  const bool constrained = p.context != nullptr && p.rank > 0;
  const bool odd = lane & 1, hi = lane & 2;

  for (int e = tid; e < 3 * PA; e += nth) gds[e] = 0.f;       // gds, vals, nz
  __syncthreads();
  for (int q = tid; q < P; q += nth) {
    const int y = q / p.w, n = y * WP + (q - y * p.w);
    nmap[q] = n;
    vals[2 * n] = p.val[(int64_t)o0 * P + q];
    vals[2 * n + 1] = p.val[(int64_t)(o0 + 1) * P + q];
    if (!plain) nz[n] = p.noise_w[0] * p.noise[q];
  }
  if (lrg) {
    // kd_r[e] = sum_j d_r[j] key[j][e]: thread e walks the channels of its position (consecutive floats of keyT)
    for (int e = tid; e < PS; e += nth) {
      const svp_f4* row = reinterpret_cast<const svp_f4*>(keyT + (int64_t)e * p.in_ch);
      for (int r = 0; r < p.rank; ++r) {
        const svp_f4* d4 = reinterpret_cast<const svp_f4*>(p.context + (int64_t)r * p.in_ch);
        float acc = 0.f;
        for (int j = 0; j < p.in_ch / 4; ++j) {
          const svp_f4 kv = row[j], dv = d4[j];
          acc += kv[0] * dv[0] + kv[1] * dv[1] + kv[2] * dv[2] + kv[3] * dv[3];
        }
        kd[r * PS + e] = acc;
      }
    }
  }
  svp_f2 W[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int64_t idx = (int64_t)(o0 + o) * K + i * 9 + t;
      W[t][o] = p.weight[idx];
      mv[(2 * t + o) * nth] = p.exp_avg[idx];
      mv[(18 + 2 * t + o) * nth] = p.exp_avg_sq[idx];
    }
  float dctx[SVP_RMAX];
#pragma unroll
  for (int r = 0; r < SVP_RMAX; ++r) dctx[r] = (constrained && r < p.rank) ? p.context[(int64_t)r * p.in_ch + i] : 0.f;
  const float sg = p.style[i], sig2 = sg * sg, s = p.w_scale;
  const float inv_numel = 1.0f / ((float)p.out_ch * (float)P);
  const float bias0 = plain ? 0.f : p.bias[o0], bias1 = plain ? 0.f : p.bias[o0 + 1];
  __syncthreads();

  // sums over the workgroup's channels of x[t][o] d_r: red[(wave * RMAX + r) * 18 + 2 t + o] (one barrier to publish)
  auto reduce_rows = [&](const svp_f2 (&x)[9], bool weighted) __attribute__((always_inline)) {
    for (int r = 0; r < p.rank; ++r) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const float sum = svp_wave_sum63(x[t][o] * (weighted ? dctx[r] * sig2 : dctx[r]));
          if (lane == 63) red[(wave * SVP_RMAX + r) * 18 + 2 * t + o] = sum;
        }
    }
    __syncthreads();
  };
  auto project_rows = [&](svp_f2 (&x)[9]) __attribute__((always_inline)) {
    reduce_rows(x, false);
    svp_f2 out[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) out[t] = svp_f2{0.f, 0.f};
    for (int r = 0; r < p.rank; ++r) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        svp_f2 c = {0.f, 0.f};
        for (int w2 = 0; w2 < NW; ++w2) c += *reinterpret_cast<const svp_f2*>(red + (w2 * SVP_RMAX + r) * 18 + 2 * t);
        out[t] += c * dctx[r];
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) x[t] = out[t];
    __syncthreads();
  };
  // padded row r of this thread's channel -> registers (entries beyond the row / the crop's padded extent are 0)
  // (buffer loads: one vector register of lane offsets and a scalar offset per element -- twenty 64-bit vector
  // addresses per row spilled the kernel --, and an element past the copy's end reads 0 by the range check)
  const __amdgpu_buffer_rsrc_t ksrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(keyT), 0, (int)((int64_t)PS * p.in_ch * 4), 0x00020000);
  const int klane = i * 4, krow4 = p.in_ch * 4;
  auto load_row = [&](int r, float (&Kr)[SVS_ROW]) __attribute__((always_inline)) {
    const int s0 = r * WP * krow4;
#pragma unroll
    for (int j = 0; j < SVS_ROW; ++j)
      Kr[j] = j < WP + 3 ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ksrc, klane, s0 + j * krow4, 0)) : 0.f;
  };

  float ss_next = p.step_size[it0 < it1 ? it0 : 0], bc_next = p.bc2_sqrt[it0 < it1 ? it0 : 0];
  for (int it = it0; it < it1; ++it) {
    const float step_size = ss_next, bc2s = bc_next;
    if (it + 1 < it1) { ss_next = p.step_size[it + 1]; bc_next = p.bc2_sqrt[it + 1]; }
    // ---- A: partial convolution of my channel (rows streamed), the demodulation partial
    {
      svp_f2 wq2 = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const svp_f2 a = (s * W[t]) * sg;
        wq2 += a * a;
      }
      const float wq0 = svp_wave_sum63(wq2[0]), wq1 = svp_wave_sum63(wq2[1]);
      if (lane == 63) { wq[wave * 2] = wq0; wq[wave * 2 + 1] = wq1; }
      auto conv_row = [&](int y, const float (&R0)[SVS_ROW], const float (&R1)[SVS_ROW], const float (&R2)[SVS_ROW])
          __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < SVS_WP_MAX; x += 2) {
          if (x < WP) {
            svp_f2 a = {0.f, 0.f}, b = {0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              const float (&R)[SVS_ROW] = ky == 0 ? R0 : (ky == 1 ? R1 : R2);
              const float k0 = R[x], k1 = R[x + 1], k2 = R[x + 2], k3 = R[x + 3];
              a = __builtin_elementwise_fma(W[3 * ky], svp_f2{k0, k0}, a);
              b = __builtin_elementwise_fma(W[3 * ky], svp_f2{k1, k1}, b);
              a = __builtin_elementwise_fma(W[3 * ky + 1], svp_f2{k1, k1}, a);
              b = __builtin_elementwise_fma(W[3 * ky + 1], svp_f2{k2, k2}, b);
              a = __builtin_elementwise_fma(W[3 * ky + 2], svp_f2{k2, k2}, a);
              b = __builtin_elementwise_fma(W[3 * ky + 2], svp_f2{k3, k3}, b);
            }
            const float v = svp_wave_sum4(a, b, odd, hi);
            const int n2 = 2 * (y * WP + x) + lane;
            if (lane < 4 && n2 < 2 * NP) part[wave * PA + n2] = v;
          }
        }
      };
      float KA[SVS_ROW], KB[SVS_ROW], KC[SVS_ROW], KD[SVS_ROW];
      load_row(0, KA); load_row(1, KB); load_row(2, KC); load_row(3, KD);
      for (int y = 0; y < p.h; y += 4) {
        conv_row(y, KA, KB, KC);
        if (y + 4 <= p.h + 1) load_row(y + 4, KA);
        if (y + 1 < p.h) { conv_row(y + 1, KB, KC, KD); if (y + 5 <= p.h + 1) load_row(y + 5, KB); }
        if (y + 2 < p.h) { conv_row(y + 2, KC, KD, KA); if (y + 6 <= p.h + 1) load_row(y + 6, KC); }
        if (y + 3 < p.h) { conv_row(y + 3, KD, KA, KB); if (y + 7 <= p.h + 1) load_row(y + 7, KD); }
      }
    }
    __syncthreads();
    // ---- B: wave o finishes out-channel o0 + o
    for (int o = wave; o < 2; o += NW) {
      float wsq = 0.f;
      for (int w2 = 0; w2 < NW; ++w2) wsq += wq[w2 * 2 + o];
      const float demod = rsqrtf(wsq + 1e-8f);
      const float bv = o ? bias1 : bias0;
      float lsum = 0.f, tsum = 0.f;
      for (int q = lane; q < P; q += 64) {
        const int n2 = 2 * nmap[q] + o;
        float conv = 0.f;
        for (int w2 = 0; w2 < NW; ++w2) conv += part[w2 * PA + n2];
        conv *= s;
        float out, pre;
        if (plain) { pre = conv * demod; out = pre; }
        else {
          pre = conv * demod + nz[n2 >> 1] + bv;
          out = 1.4142135623730951f * ((pre > 0.f) ? pre : 0.2f * pre);
        }
        const float diff = out - vals[n2];
        lsum += fabsf(diff);
        const float sgn = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
        const float g_out = sgn * inv_numel;
        const float g_pre = plain ? g_out : ((pre > 0.f) ? g_out : g_out * 0.2f) * 1.4142135623730951f;
        gds[n2] = g_pre * demod;
        tsum += g_pre * conv;
      }
      lsum = svp_wave_sum63(lsum);
      tsum = svp_wave_sum63(tsum);
      if (lane == 63) {
        chn[o * 4] = s * s * demod * demod * demod * tsum;
        lpart_all[(int64_t)it * p.out_ch + o0 + o] = lsum * inv_numel;
      }
    }
    __syncthreads();
    // ---- C: gradient of my 18 weights, Adam, projections
    {
      svp_f2 G[9];
      const svp_f2 c2 = {chn[0], chn[4]};
      if (lrg) {
        // s * sum_n g[n][o] kd_r[n + tap]: wave w takes the (context row, tap) pairs w, w + NW, ...
        for (int idx = wave; idx < p.rank * 9; idx += NW) {
          const int r = idx / 9, t = idx - 9 * r;
          const float* kr = kd + r * PS + (t / 3) * WP + (t % 3);
          svp_f2 acc = {0.f, 0.f};
          for (int n = lane; n < NP; n += 64) acc += *reinterpret_cast<const svp_f2*>(gds + 2 * n) * kr[n];
          const float a0 = svp_wave_sum63(acc[0]), a1 = svp_wave_sum63(acc[1]);
          if (lane == 63) { cgs[r * 18 + 2 * t] = s * a0; cgs[r * 18 + 2 * t + 1] = s * a1; }
        }
        reduce_rows(W, true);                       // sum_j W[t][o][j] sig2[j] d_r[j] (publishes cgs too)
#pragma unroll
        for (int t = 0; t < 9; ++t) G[t] = svp_f2{0.f, 0.f};
        for (int r = 0; r < p.rank; ++r) {
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            svp_f2 wsd = {0.f, 0.f};
            for (int w2 = 0; w2 < NW; ++w2) wsd += *reinterpret_cast<const svp_f2*>(red + (w2 * SVP_RMAX + r) * 18 + 2 * t);
            const svp_f2 cg = *reinterpret_cast<const svp_f2*>(cgs + r * 18 + 2 * t);
            G[t] += (cg - c2 * wsd) * dctx[r];
          }
        }
        __syncthreads();                            // red (= part) and cgs are free again
      } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) G[t] = svp_f2{0.f, 0.f};
        auto grad_row = [&](int y, const float (&R0)[SVS_ROW], const float (&R1)[SVS_ROW], const float (&R2)[SVS_ROW])
            __attribute__((always_inline)) {
#pragma unroll
          for (int x = 0; x < SVS_WP_MAX; x += 2) {
            if (x < WP) {
              const int n = y * WP + x;
              const svp_f2 ga = *reinterpret_cast<const svp_f2*>(gds + 2 * n);
              // the second position of a row's last pair belongs to the next row (its first pair counts it)
              const svp_f2 gb = (x + 1 < WP && n + 1 < NP) ? *reinterpret_cast<const svp_f2*>(gds + 2 * n + 2) : svp_f2{0.f, 0.f};
#pragma unroll
              for (int ky = 0; ky < 3; ++ky) {
                const float (&R)[SVS_ROW] = ky == 0 ? R0 : (ky == 1 ? R1 : R2);
                const float k0 = R[x], k1 = R[x + 1], k2 = R[x + 2], k3 = R[x + 3];
                G[3 * ky] = __builtin_elementwise_fma(ga, svp_f2{k0, k0}, G[3 * ky]);
                G[3 * ky] = __builtin_elementwise_fma(gb, svp_f2{k1, k1}, G[3 * ky]);
                G[3 * ky + 1] = __builtin_elementwise_fma(ga, svp_f2{k1, k1}, G[3 * ky + 1]);
                G[3 * ky + 1] = __builtin_elementwise_fma(gb, svp_f2{k2, k2}, G[3 * ky + 1]);
                G[3 * ky + 2] = __builtin_elementwise_fma(ga, svp_f2{k2, k2}, G[3 * ky + 2]);
                G[3 * ky + 2] = __builtin_elementwise_fma(gb, svp_f2{k3, k3}, G[3 * ky + 2]);
              }
            }
          }
        };
        float KA[SVS_ROW], KB[SVS_ROW], KC[SVS_ROW], KD[SVS_ROW];
        load_row(0, KA); load_row(1, KB); load_row(2, KC); load_row(3, KD);
        for (int y = 0; y < p.h; y += 4) {
          grad_row(y, KA, KB, KC);
          if (y + 4 <= p.h + 1) load_row(y + 4, KA);
          if (y + 1 < p.h) { grad_row(y + 1, KB, KC, KD); if (y + 5 <= p.h + 1) load_row(y + 5, KB); }
          if (y + 2 < p.h) { grad_row(y + 2, KC, KD, KA); if (y + 6 <= p.h + 1) load_row(y + 6, KC); }
          if (y + 3 < p.h) { grad_row(y + 3, KD, KA, KB); if (y + 7 <= p.h + 1) load_row(y + 7, KD); }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) G[t] = s * G[t] - c2 * W[t] * sig2;
        if (p.low_rank_gradient) project_rows(G);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          float wv = W[t][o], mm = mv[(2 * t + o) * nth], vv = mv[(18 + 2 * t + o) * nth];
          adam_update(G[t][o], wv, mm, vv, p.one_minus_beta1, p.beta2, p.one_minus_beta2, p.eps, step_size, bc2s);
          W[t][o] = wv; mv[(2 * t + o) * nth] = mm; mv[(18 + 2 * t + o) * nth] = vv;
        }
      if (low_rank_insert && (it % piter == 0 || it == niter - 1)) {
        svp_f2 pw[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) pw[t] = W[t];
        project_rows(pw);
#pragma unroll
        for (int t = 0; t < 9; ++t)          // ortho part: read where it is used (every piter-th iteration)
          W[t] = svp_f2{p.ortho[(int64_t)o0 * K + i * 9 + t], p.ortho[(int64_t)(o0 + 1) * K + i * 9 + t]} + pw[t];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int64_t idx = (int64_t)(o0 + o) * K + i * 9 + t;
      p.weight[idx] = W[t][o]; p.exp_avg[idx] = mv[(2 * t + o) * nth]; p.exp_avg_sq[idx] = mv[(18 + 2 * t + o) * nth];
    }
}

// losses[it] = sum over out-channels of the per-channel parts, in channel order (deterministic)
__global__ void __launch_bounds__(64) solve_loss_sum_kernel(const float* __restrict__ lpart_all, float* __restrict__ losses,
                                                            int out_ch, int it0, int it1) {
  const int it = it0 + blockIdx.x;
  if (it >= it1) return;
  float l = 0.f;
  for (int o = threadIdx.x; o < out_ch; o += 64) l += lpart_all[(int64_t)it * out_ch + o];
  l = rw_wave_sum(l);
  if (threadIdx.x == 0) losses[it] = l;
}

static size_t svp_lds_bytes(int in_ch, int h, int w) {
  const int NW = in_ch / 64;
  return ((size_t)in_ch * svp_row_pitch(h, w) + svp_part_floats(NW, h, w) + 3 * (size_t)svp_pair_row(h, w) + 16 + 8 +
          (size_t)h * w) * sizeof(float);
}
// the streaming form: no crop in the LDS; kd maps of the context rows for low_rank_gradient
static size_t svs_lds_bytes(int in_ch, int h, int w, int rank) {
  const int NW = in_ch / 64;
  return ((size_t)svp_part_floats(NW, h, w) + 3 * (size_t)svp_pair_row(h, w) + 16 + 8 + SVP_RMAX * 18 + (size_t)h * w +
          (size_t)rank * svp_row_pitch(h, w) + 36 * (size_t)in_ch) * sizeof(float);
}
static bool svp_common_ok(int out_ch, int in_ch, int h, int w, int rank, int upsample, int linear_insert) {
  if (out_ch <= 0 || in_ch <= 0 || h <= 0 || w <= 0 || upsample || linear_insert) return false;
  return !(out_ch % 2 || in_ch % 64 || in_ch > 512 || rank > SVP_RMAX);
}
static bool svp_resident(int in_ch, int h, int w) { return svp_lds_bytes(in_ch, h, w) <= 160 * 1024; }
static bool svs_streamable(int in_ch, int h, int w, int rank) {
  return w + 1 <= SVS_WP_MAX && in_ch % 4 == 0 && svs_lds_bytes(in_ch, h, w, rank) <= 150 * 1024;
}

// which kernel a launch takes: the streaming one where the crop does not fit, or everywhere it can run under
// RW_SOLVE_STREAM=1 (tests) -- ONE statement, used by the scratch size and by the launch
static bool svs_use_stream(int in_ch, int h, int w, int rank) {
  const char* force = getenv("RW_SOLVE_STREAM");
  return !svp_resident(in_ch, h, w) || (force && force[0] == '1' && svs_streamable(in_ch, h, w, rank));
}

// 1 when rw_solve_run_f32 takes the target, 0 otherwise (then rw_solve_step_f32 is the way)
extern "C" int rw_solve_run_supported(int out_ch, int in_ch, int h, int w, int rank, int upsample, int linear_insert) {
  if (!svp_common_ok(out_ch, in_ch, h, w, rank, upsample, linear_insert)) return 0;
  return (svp_resident(in_ch, h, w) || svs_streamable(in_ch, h, w, rank)) ? 1 : 0;
}

// floats of `lpart` rw_solve_run_f32 needs: niter * out_ch loss parts + (crops beyond the LDS) the position-major copy
// of the key crop the streaming kernel reads
extern "C" long long rw_solve_run_scratch_elems(int out_ch, int in_ch, int h, int w, int niter) {
  if (out_ch <= 0 || in_ch <= 0 || h <= 0 || w <= 0 || niter < 0) return -1;
  long long n = ((long long)niter * out_ch + 3) / 4 * 4;
  if (svs_use_stream(in_ch, h, w, 0)) n += (long long)svp_row_pitch(h, w) * in_ch;
  return n;
}

// Iterations [it_begin, it_end) of the niter-iteration solve in one launch (state is read from and written back to
// weight / exp_avg / exp_avg_sq, so consecutive calls continue each other); project on the iterations the reference
// projects on (it % piter == 0 or it == niter - 1) when low_rank_insert != 0.  lpart: rw_solve_run_scratch_elems floats
// of scratch; losses[it] is written for the iterations run.  Uses no other scratch of rw_solve_problem and not its
// step counter.
extern "C" int rw_solve_run_f32(const rw_solve_problem* pr, int it_begin, int it_end, int niter, int piter,
                                int low_rank_insert, float* lpart, rw_stream_t stream) {
  RW_CHECK_ARG(pr && lpart);
  const rw_solve_problem& p = *pr;
  RW_CHECK_ARG(p.key && p.style && p.val && p.weight && p.exp_avg && p.exp_avg_sq && p.step_size && p.bc2_sqrt &&
               p.losses);
  RW_CHECK_ARG(!p.bias || (p.noise && p.noise_w));
  RW_CHECK_ARG(0 <= it_begin && it_begin <= it_end && it_end <= niter && piter > 0);
  RW_CHECK_ARG(!(low_rank_insert || p.low_rank_gradient) || (p.context && p.rank > 0));
  RW_CHECK_ARG(!low_rank_insert || p.ortho);
  if (!rw_solve_run_supported(p.out_ch, p.in_ch, p.h, p.w, p.rank, p.upsample, p.linear_insert)) return RW_ERR_UNSUPPORTED;
  if (it_begin == it_end) return 0;
  hipStream_t s = rw_s(stream);
  if (svs_use_stream(p.in_ch, p.h, p.w, p.rank)) {
    float* keyT = lpart + ((long long)niter * p.out_ch + 3) / 4 * 4;
    const int64_t total = (int64_t)svp_row_pitch(p.h, p.w) * p.in_ch;
    hipLaunchKernelGGL(solve_key_transpose_kernel, dim3(rw_stream_grid(total, 256)), dim3(256), 0, s, p.key, keyT, p.in_ch,
                       p.h, p.w);
    const size_t ldsb = svs_lds_bytes(p.in_ch, p.h, p.w, p.low_rank_gradient ? p.rank : 0);
    hipError_t e = hipFuncSetAttribute((const void*)solve_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(solve_stream_kernel, dim3(p.out_ch / 2), dim3(p.in_ch), ldsb, s, p, it_begin, it_end, niter, piter,
                       low_rank_insert, lpart, (const float*)keyT);
  } else {
    const size_t ldsb = svp_lds_bytes(p.in_ch, p.h, p.w);
    hipError_t e = hipFuncSetAttribute((const void*)solve_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)ldsb);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(solve_persistent_kernel, dim3(p.out_ch / 2), dim3(p.in_ch), ldsb, s, p, it_begin, it_end, niter,
                       piter, low_rank_insert, lpart);
  }
  hipLaunchKernelGGL(solve_loss_sum_kernel, dim3(it_end - it_begin), dim3(64), 0, s, (const float*)lpart, p.losses,
                     p.out_ch, it_begin, it_end);
  return RW_LAUNCH_RESULT();
}

// Shape limits of the solver kernels, in one place (rw_solve_step_f32 calls it before its first launch and
// the host mirrors it in _hip_solvable): 64 out-channels per workgroup in both GEMMs, 16-channel K chunks,
// the blur / blur-backward staging of an upsampling target in <= 64 KB of LDS, and the two weight rows of the
// rank-r projection in <= 64 KB.
static int sv_check_shape(int out_ch, int in_ch, int h, int w, int upsample, int plain, int constrained) {
  if (out_ch <= 0 || in_ch <= 0 || h <= 0 || w <= 0) return RW_ERR_BAD_ARGUMENT;
  if (out_ch % SV_BM || in_ch % SV_KC) return RW_ERR_UNSUPPORTED;
  if (upsample && !plain) {
    const size_t P = (size_t)(2 * h + 1) * (2 * w + 1);
    if ((P + (size_t)4 * h * w) * sizeof(float) > 64 * 1024) return RW_ERR_UNSUPPORTED;
  }
  if (constrained && 2 * (size_t)in_ch * 9 * sizeof(float) > 64 * 1024) return RW_ERR_UNSUPPORTED;
  return 0;
}

// 1 when rw_solve_step_f32 takes the shape, 0 otherwise -- the convention of every other *_supported export.
extern "C" int rw_solve_supported(int out_ch, int in_ch, int h, int w, int upsample, int plain, int constrained) {
  return sv_check_shape(out_ch, in_ch, h, w, upsample, plain, constrained) == 0 ? 1 : 0;
}

// Element counts of the scratch buffers of rw_solve_problem for this shape: sizes[0..4] =
// {conv, wsq, gd, c2, grad} (floats), and sizes[5] = the split-K factor they were sized for (= rw_solve_ksplit of the
// map the convolution writes: the value rw_solve_problem.ksplit has to carry).  Rows of conv / gd are padded to ceil64 of the positions of the map the
// convolution writes (h*w, or (2h+1)(2w+1) for an upsampling target); c2 carries the per-channel loss behind it.
extern "C" int rw_solve_scratch_elems(int out_ch, int in_ch, int h, int w, int upsample, long long* sizes) {
  if (!sizes || out_ch <= 0 || in_ch <= 0 || h <= 0 || w <= 0) return RW_ERR_BAD_ARGUMENT;
  const int ch = upsample ? 2 * h + 1 : h, cw = upsample ? 2 * w + 1 : w;
  const long long pp = sv_pp(ch * cw);
  const long long ks = rw_solve_ksplit(out_ch, in_ch, ch, cw);
  sizes[0] = ks * out_ch * pp;
  sizes[1] = ks * out_ch;
  sizes[2] = (long long)out_ch * pp;
  sizes[3] = 2LL * out_ch;
  sizes[4] = (long long)out_ch * in_ch * 9;
  sizes[5] = ks;
  return 0;
}

extern "C" int rw_solve_step_f32(const rw_solve_problem* pr, int project, rw_stream_t stream) {
  RW_CHECK_ARG(pr);
  const rw_solve_problem& p = *pr;
  RW_CHECK_ARG(p.key && p.style && p.val && p.weight && p.exp_avg &&
               p.exp_avg_sq && p.step_size && p.bc2_sqrt && p.step_counter && p.losses && p.conv &&
               p.wsq && p.gd && p.c2);
  RW_CHECK_ARG(!p.bias || (p.noise && p.noise_w));          // bias == NULL: plain dconv target
  RW_CHECK_ARG(p.out_ch > 0 && p.in_ch > 0 && p.h > 0 && p.w > 0 && p.ksplit > 0 && p.ksplit <= 32);
  RW_CHECK_ARG(!(project || p.low_rank_gradient || p.linear_insert) || (p.context && p.rank > 0));
  RW_CHECK_ARG(!(project || p.linear_insert) || p.ortho);
  RW_CHECK_ARG(!(p.low_rank_gradient || p.linear_insert) || p.grad);
  RW_CHECK_ARG(!p.linear_insert || (p.lambda && !project && !p.low_rank_gradient));
  RW_CHECK_ARG(!p.upsample || !p.bias || p.blur_k);
  // every shape / LDS limit is checked BEFORE the first launch: a step either runs completely or not at all
  {
    const int rc = sv_check_shape(p.out_ch, p.in_ch, p.h, p.w, p.upsample, p.bias ? 0 : 1,
                                  (project || p.low_rank_gradient || p.linear_insert) ? 1 : 0);
    if (rc) return rc;
  }
  hipStream_t s = rw_s(stream);
  const int P = sv_conv_h(p) * sv_conv_w(p);
  const int pp = sv_pp(P);
  float* lpart = p.c2 + p.out_ch;   // c2 is allocated with 2*out_ch floats: [c2 | per-channel loss]
  hipLaunchKernelGGL(solve_fwd_kernel, dim3(p.out_ch / SV_BM, pp / SV_BN, p.ksplit), dim3(256), 0, s, p, pp);
  if (p.upsample && p.bias) {
    const size_t mid_lds = ((size_t)P + (size_t)4 * p.h * p.w) * sizeof(float);
    hipLaunchKernelGGL(solve_mid_up_kernel, dim3(p.out_ch), dim3(256), mid_lds, s, p, pp, lpart);
  } else {
    hipLaunchKernelGGL(solve_mid_kernel, dim3((unsigned)rw_cdiv(p.out_ch, 4)), dim3(256), 0, s, p, pp, lpart);
  }
  hipLaunchKernelGGL(solve_bwd_adam_kernel, dim3(p.out_ch / SV_BM, (unsigned)rw_cdiv(9 * p.in_ch, SV_BNK)), dim3(256), 0,
                     s, p, pp, (const float*)lpart);
  const size_t lds = 2 * (size_t)p.in_ch * 9 * sizeof(float);
  if (p.low_rank_gradient) {
    hipLaunchKernelGGL(project_kernel<1>, dim3(p.out_ch), dim3(256), lds, s, (const float*)p.grad,
                       p.context, (const float*)nullptr, (float*)nullptr, p.in_ch, 9, p.rank, p);
  }
  if (p.linear_insert) {
    hipLaunchKernelGGL(project_kernel<2>, dim3(p.out_ch), dim3(256), lds, s, (const float*)p.grad,
                       p.context, p.ortho, p.weight, p.in_ch, 9, p.rank, p);
  }
  if (project) {
    hipLaunchKernelGGL(project_kernel<0>, dim3(p.out_ch), dim3(256), lds, s, (const float*)p.weight,
                       p.context, p.ortho, p.weight, p.in_ch, 9, p.rank, p);
  }
  return RW_LAUNCH_RESULT();
}
