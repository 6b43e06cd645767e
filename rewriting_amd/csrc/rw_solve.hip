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
