// Gradient kernels of the demodulated 3x3 convolution (DemodulatedConv2dF, utils/stylegan2/models.py:291-329) for the
// autograd path: `insert` on a target the fused solver does not restate (several layers, a hooked module, a goal
// batch > 1) runs the reference's loop -- loss.backward() through the module chain (rewrite/ganrewrite.py:265-283) --
// and torch.autograd then needs, per convolution,
//
//   d fmap  = the SAME forward kernels on the transposed (and, stride 1, flipped) weights   (host side: hip.py)
//   d W     = s * sum_{b,p} (g[b,o,p] demod[b,o]) xcol_b[(i,tap), p]                          (rw_conv_wgrad_f32, here)
//             - s^2 W[o,i,t] sum_b sigma[b,i]^2 demod[b,o]^2 sum_p g[b,o,p] y[b,o,p]          (rw_rowdot_f32 + host)
//
// (Q3: the demodulation factor is differentiable).  The weight gradient is the GEMM of the
// solver's K3 (rw_solve.hip) without its Adam epilogue, generalised to a batch and to maps of any size: M = out
// channels, N = (in channel, tap) columns, K = batch x conv-output positions, fp32 MFMA, split-K over workgroups
// with the partial sums reduced in a fixed order (no atomics: the result is deterministic).
#include "rw_common.h"

#define WG_KC 16        // K chunk: conv-output positions per staging step
#define WG_BM 64        // out channels per workgroup
#define WG_BN 64        // (i, tap) columns per workgroup
#define WG_DEPTH 4      // chunks of operands in flight

struct WgradProblem {
  const float* g;       // (batch, out_ch, CH, CW): gradient w.r.t. the convolution's output map
  const float* x;       // (batch, in_ch, h, w): the (modulated) input map
  const float* gscale;  // (batch, out_ch) factor on g (the demodulation factor), nullable
  const float* xscale;  // (batch, in_ch) factor on x (a style applied on load), nullable
  float* part;          // (ksplit, out_ch, 9 in_ch) partial sums
  int batch, in_ch, out_ch, h, w, upsample, ksplit;
};

__host__ __device__ static inline int wg_conv_w(const WgradProblem& p) { return p.upsample ? 2 * p.w + 1 : p.w; }
__host__ __device__ static inline int wg_conv_h(const WgradProblem& p) { return p.upsample ? 2 * p.h + 1 : p.h; }

// the input sample that weight tap (ky, kx) of one channel multiplies at conv-output position (Y, X): zero padding
// for the stride-1 convolution (pad 1); for the stride-2 transposed convolution only where the parity of
// (Y - ky, X - kx) lands on an input sample (F.conv_transpose2d, models.py:315-316)
__device__ __forceinline__ float wg_gather(const WgradProblem& p, const float* xi, int tap, int Y, int X) {
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
  return xi[iy * p.w + ix];
}

__global__ void __launch_bounds__(256) conv_wgrad_kernel(const WgradProblem p) {
  __shared__ float As[2][WG_KC][WG_BM + 4];
  __shared__ float Bs[2][WG_KC][WG_BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;      // wave tile: 32 out channels x 32 columns
  const int frow = lane >> 5, fcol = lane & 31;
  const int o0 = blockIdx.x * WG_BM, k0 = blockIdx.y * WG_BN, ks = blockIdx.z;
  const int CW = wg_conv_w(p), P = wg_conv_h(p) * CW;
  const int K = 9 * p.in_ch;
  const int cpb = (P + WG_KC - 1) / WG_KC;                     // chunks per batch item (none straddles two items)
  const int chunks = cpb * p.batch;
  const int cbeg = (int)((int64_t)chunks * ks / p.ksplit), cend = (int)((int64_t)chunks * (ks + 1) / p.ksplit);

  // A staging: thread -> (out channel o0 + tid / 4, four consecutive positions); B: thread -> (column tid % 64,
  // positions tid / 64 + 4 j)
  const int arow = tid >> 2, apart = (tid & 3) * 4;
  const int ao = o0 + arow;
  const bool a_ok = ao < p.out_ch;
  const int bcol = tid & 63, bp0 = tid >> 6;
  const int kmine = k0 + bcol;
  const bool col_ok = kmine < K;
  const int ci = col_ok ? kmine / 9 : 0, ctap = col_ok ? kmine - 9 * ci : 0;
  const int64_t hw = (int64_t)p.h * p.w;

  float areg[WG_DEPTH][4], breg[WG_DEPTH][4];
  auto fetch = [&](int c, int slot) __attribute__((always_inline)) {
    const int b = c / cpb, p0 = (c - b * cpb) * WG_KC;
    const float gs = (a_ok && p.gscale) ? p.gscale[(int64_t)b * p.out_ch + ao] : 1.f;
    const float* gr = p.g + ((int64_t)b * p.out_ch + (a_ok ? ao : 0)) * P;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = p0 + apart + e;
      areg[slot][e] = (a_ok && n < P) ? gr[n] * gs : 0.f;
    }
    const float xs = (col_ok && p.xscale) ? p.xscale[(int64_t)b * p.in_ch + ci] : 1.f;
    const float* xi = p.x + ((int64_t)b * p.in_ch + ci) * hw;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = p0 + bp0 + 4 * j;
      float v = 0.f;
      if (n < P && col_ok) {
        const int y = n / CW;
        v = wg_gather(p, xi, ctap, y, n - y * CW) * xs;
      }
      breg[slot][j] = v;
    }
  };
  auto stash = [&](int buf, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) As[buf][apart + e][arow] = areg[slot][e];
#pragma unroll
    for (int j = 0; j < 4; ++j) Bs[buf][bp0 + 4 * j][bcol] = breg[slot][j];
  };

  rw_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int d = 0; d < WG_DEPTH; ++d)
    if (cbeg + d < cend) fetch(cbeg + d, d);
  if (cbeg < cend) stash(0, 0);
  __syncthreads();
  for (int base = cbeg; base < cend; base += WG_DEPTH) {
#pragma unroll
    for (int d = 0; d < WG_DEPTH; ++d) {
      const int c = base + d;
      if (c < cend) {                                // uniform
        const int buf = d & 1;                       // WG_DEPTH is even
        if (c + WG_DEPTH < cend) fetch(c + WG_DEPTH, d);
#pragma unroll
        for (int kp = 0; kp < WG_KC / 2; ++kp) {
          const float af = As[buf][2 * kp + frow][wm0 + fcol];
          const float bf = Bs[buf][2 * kp + frow][wn0 + fcol];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
        }
        if (c + 1 < cend) stash(buf ^ 1, (d + 1) % WG_DEPTH);
        __syncthreads();
      }
    }
  }
  float* out = p.part + (int64_t)ks * p.out_ch * K;
  const int k = k0 + wn0 + fcol;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = o0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * frow;
    if (o < p.out_ch && k < K) out[(int64_t)o * K + k] = acc[r];
  }
}

// out[e] = scale * sum_s part[s][e], s in order (deterministic)
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                           int64_t n, int ksplit, float scale) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    float acc = 0.f;
    for (int s = 0; s < ksplit; ++s) acc += part[(int64_t)s * n + e];
    out[e] = acc * scale;
  }
}

extern "C" int rw_conv_wgrad_ksplit(int batch, int in_ch, int out_ch, int h, int w, int upsample) {
  if (batch <= 0 || in_ch <= 0 || out_ch <= 0 || h <= 0 || w <= 0) return 0;
  const int CH = upsample ? 2 * h + 1 : h, CW = upsample ? 2 * w + 1 : w;
  const int64_t blocks = rw_cdiv(out_ch, WG_BM) * rw_cdiv(9 * (int64_t)in_ch, WG_BN);
  const int64_t chunks = rw_cdiv((int64_t)CH * CW, WG_KC) * batch;
  int64_t ks = rw_cdiv(1024, blocks);                 // ~4 workgroups per CU
  if (ks > 64) ks = 64;
  if (ks > chunks) ks = chunks;
  if (ks < 1) ks = 1;
  return (int)ks;
}

extern "C" int rw_conv_wgrad_f32(const float* g, const float* x, const float* gscale, const float* xscale,
                                 float* scratch, float* dw, int batch, int in_ch, int out_ch, int h, int w,
                                 int upsample, float scale, rw_stream_t stream) {
  RW_CHECK_ARG(g && x && scratch && dw && batch > 0 && in_ch > 0 && out_ch > 0 && h > 0 && w > 0);
  WgradProblem p;
  p.g = g; p.x = x; p.gscale = gscale; p.xscale = xscale; p.part = scratch;
  p.batch = batch; p.in_ch = in_ch; p.out_ch = out_ch; p.h = h; p.w = w; p.upsample = upsample ? 1 : 0;
  p.ksplit = rw_conv_wgrad_ksplit(batch, in_ch, out_ch, h, w, upsample);
  hipStream_t s = rw_s(stream);
  const dim3 grid((unsigned)rw_cdiv(out_ch, WG_BM), (unsigned)rw_cdiv(9 * (int64_t)in_ch, WG_BN), (unsigned)p.ksplit);
  hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), 0, s, p);
  const int64_t n = (int64_t)out_ch * in_ch * 9;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rw_stream_grid(n, 256)), dim3(256), 0, s, (const float*)scratch, dw, n,
                     p.ksplit, scale);
  return RW_LAUNCH_RESULT();
}

// out[r] = sum_j a[r][j] b[r][j]: one workgroup per row, float32 products summed in a fixed order per thread,
// then a block tree (per-(image, channel) sums over a feature map: the demodulation term, bias-like gradients)
__global__ void __launch_bounds__(256) rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ out, int64_t n) {
  __shared__ float red[4];
  const int64_t r = blockIdx.x;
  const float* ar = a + r * n;
  const float* br = b + r * n;
  float acc = 0.f;
  for (int64_t j = threadIdx.x; j < n; j += 256) acc += ar[j] * br[j];
  acc = rw_block_sum_256(acc, red);
  if (threadIdx.x == 0) out[r] = acc;
}

extern "C" int rw_rowdot_f32(const float* a, const float* b, float* out, long long rows, long long n,
                             rw_stream_t stream) {
  RW_CHECK_ARG(a && b && out && rows > 0 && n > 0 && rows < (1LL << 31));
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)rows), dim3(256), 0, rw_s(stream), a, b, out, (int64_t)n);
  return RW_LAUNCH_RESULT();
}
