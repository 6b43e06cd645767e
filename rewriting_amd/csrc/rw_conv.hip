// 3x3 modulated convolutions of the StyleGANv2 generator as fp32 implicit GEMMs on the CDNA4
// matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-for-bit an fmaf chain --
// there is no TF32 on gfx950, and the fp32 MFMA rate equals the fp32 VALU peak, 157 TFLOP/s,
// while leaving the VALU free for staging and the epilogue).
//
//   D[o][n] = sum_k A[o][k] * B[k][n]
//   A = repacked weights wp[tap][i][o]         (o contiguous: 16-byte LDS staging, conflict-free
//                                               32-lane fragment reads)
//   B = im2col of the NCHW input, gathered on the fly: a K-chunk is 16 input channels of ONE
//       tap, so a thread's 8 or 16 gathers share one (dy,dx) and one bounds test, and
//       consecutive lanes read consecutive W positions (coalesced NCHW rows).
//   n = (image, y, x) flattened so that small feature maps (4x4 .. 16x16) still fill a tile.
//
// The same kernel runs the stride-1 convolution (9 taps) and each output-parity phase of the
// stride-2 transposed convolution (4/2/2/1 taps, no multiplications by inserted zeros).
// The epilogue applies, in registers, the weight scale, the demodulation factor and -- for
// stride-1 layers -- noise, bias and leaky-ReLU, so a styled-conv block reads its input once
// and writes its output once.
#include "rw_common.h"

__host__ __device__ __forceinline__ int rw_tap_off(unsigned bits, int t) {
  return (int)((bits >> (2 * t)) & 3u) - 1;
}

struct ConvProblem {
  const float* x; const float* wp; float* y;
  const float* style; const float* demod; const float* noise; const float* noise_w; const float* bias;
  int batch, in_ch, out_ch, h, w;   // input tensor
  int ph, pw;                       // positions per image in this problem
  int oh, ow;                       // output tensor
  int sy, sx, oy0, ox0;             // output pixel = (sy*yy + oy0, sx*xx + ox0)
  int ntaps;
  unsigned dy_bits, dx_bits;        // 2 bits per tap: (dy + 1), (dx + 1); input pixel = (yy + dy, xx + dx)
  float w_scale;
  int act;
};

#define RW_KC 16

// Block id -> work item so that consecutive work items (the out-channel tiles of one pixel
// tile, which share the gathered input) sit on ONE XCD's L2.  Bijective for any total.
__device__ __forceinline__ int rw_xcd_remap(int id, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, slot = id >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

template <int TM, int TN, int WGM, int WGN>
__global__ void __launch_bounds__(256) conv_mfma_kernel(const ConvProblem p) {
  constexpr int BM = 32 * TM * WGM;
  constexpr int BN = 32 * TN * WGN;
  constexpr int B_ELEMS = RW_KC * BN / 256;
  constexpr int B_KSTEP = 256 / BN;
  constexpr int A_VEC = (RW_KC * BM / 4 + 255) / 256;
  constexpr bool A_FULL = (RW_KC * BM / 4) % 256 == 0;
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) float As[2][RW_KC][BM];
  __shared__ float Bs[2][RW_KC][BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WGN) * 32 * TM;
  const int wn0 = (wave % WGN) * 32 * TN;

  const int o_tiles = p.out_ch / BM;
  const int ppi = p.ph * p.pw;
  const int64_t n_total = (int64_t)p.batch * ppi;
  const int work = rw_xcd_remap(blockIdx.x, gridDim.x);
  const int o0 = (work % o_tiles) * BM;
  const int64_t n0 = (int64_t)(work / o_tiles) * BN;

  // ---- this thread's gather column
  const int nl = tid % BN;
  const int kk0 = tid / BN;
  const int64_t n_mine = n0 + nl;
  const bool n_ok = n_mine < n_total;
  int gb = 0, gy = 0, gx = 0;
  if (n_ok) {
    gb = (int)(n_mine / ppi);
    const int r = (int)(n_mine - (int64_t)gb * ppi);
    gy = r / p.pw;
    gx = r - gy * p.pw;
  }
  const int64_t hw = (int64_t)p.h * p.w;
  const float* xb = p.x + (int64_t)gb * p.in_ch * hw;
  const float* sb = p.style ? p.style + (int64_t)gb * p.in_ch : nullptr;

  const int chunks_per_tap = p.in_ch / RW_KC;
  const int n_chunks = chunks_per_tap * p.ntaps;

  float breg[B_ELEMS];
  rw_f32x4 areg[A_VEC];

  auto gather = [&](int c) {
    const int t = c % p.ntaps;
    const int i0 = (c / p.ntaps) * RW_KC;
    const int iy = gy + rw_tap_off(p.dy_bits, t), ix = gx + rw_tap_off(p.dx_bits, t);
    const bool ok = n_ok && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
    const float* src = xb + (int64_t)i0 * hw + (int64_t)iy * p.w + ix;
#pragma unroll
    for (int j = 0; j < B_ELEMS; ++j) {
      const int kk = kk0 + j * B_KSTEP;
      float v = 0.f;
      if (ok) {
        v = src[(int64_t)kk * hw];
        if (sb) v *= sb[i0 + kk];
      }
      breg[j] = v;
    }
    const float* wrow = p.wp + ((int64_t)t * p.in_ch + i0) * p.out_ch + o0;
#pragma unroll
    for (int j = 0; j < A_VEC; ++j) {
      const int q = tid + j * 256;
      if (A_FULL || q < RW_KC * BM / 4) {
        const int kk = q / (BM / 4), o4 = q % (BM / 4);
        areg[j] = *reinterpret_cast<const rw_f32x4*>(wrow + (int64_t)kk * p.out_ch + o4 * 4);
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int j = 0; j < B_ELEMS; ++j) Bs[buf][kk0 + j * B_KSTEP][nl] = breg[j];
#pragma unroll
    for (int j = 0; j < A_VEC; ++j) {
      const int q = tid + j * 256;
      if (A_FULL || q < RW_KC * BM / 4) {
        const int kk = q / (BM / 4), o4 = q % (BM / 4);
        *reinterpret_cast<rw_f32x4*>(&As[buf][kk][o4 * 4]) = areg[j];
      }
    }
  };

  rw_f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  gather(0);
  stash(0);
  __syncthreads();

  const int frow = lane >> 5;    // k within the pair
  const int fcol = lane & 31;    // row of A / column of B inside the 32x32 tile
  for (int c = 0; c < n_chunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < n_chunks) gather(c + 1);
#pragma unroll
    for (int kp = 0; kp < RW_KC / 2; ++kp) {
      float af[TM], bf[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) af[a] = As[buf][2 * kp + frow][wm0 + 32 * a + fcol];
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[b] = Bs[buf][2 * kp + frow][wn0 + 32 * b + fcol];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    if (c + 1 < n_chunks) stash(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (o)
  const float nw = p.noise ? p.noise_w[0] : 0.f;
  const int64_t ohw = (int64_t)p.oh * p.ow;
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int64_t n = n0 + wn0 + 32 * b + fcol;
    if (n >= n_total) continue;
    const int ib = (int)(n / ppi);
    const int r0 = (int)(n - (int64_t)ib * ppi);
    const int yy = r0 / p.pw, xx = r0 - yy * p.pw;
    const int64_t pix = (int64_t)(p.sy * yy + p.oy0) * p.ow + (p.sx * xx + p.ox0);
    const float nz = p.noise ? nw * p.noise[(int64_t)ib * ohw + pix] : 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * frow;
        float v = acc[a][b][r] * p.w_scale;
        if (p.demod) v *= p.demod[(int64_t)ib * p.out_ch + o];
        if (p.noise) v += nz;
        if (p.act) {
          v += p.bias[o];
          v = ((v > 0.f) ? v : v * 0.2f) * 1.4142135623730951f;
        }
        p.y[((int64_t)ib * p.out_ch + o) * ohw + pix] = v;
      }
    }
  }
}

// Direct VALU statement of the same problem (one thread per output sample).  Kept as an
// independent on-device cross-check of the MFMA fragment layouts (impl = 1).
__global__ void __launch_bounds__(256) conv_direct_kernel(const ConvProblem p) {
  const int ppi = p.ph * p.pw;
  const int64_t total = (int64_t)p.batch * p.out_ch * ppi;
  const int64_t hw = (int64_t)p.h * p.w;
  const int64_t ohw = (int64_t)p.oh * p.ow;
  const float nw = p.noise ? p.noise_w[0] : 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int r0 = (int)(idx % ppi);
    const int o = (int)((idx / ppi) % p.out_ch);
    const int ib = (int)(idx / ((int64_t)ppi * p.out_ch));
    const int yy = r0 / p.pw, xx = r0 - yy * p.pw;
    float acc = 0.f;
    for (int t = 0; t < p.ntaps; ++t) {
      const int iy = yy + rw_tap_off(p.dy_bits, t), ix = xx + rw_tap_off(p.dx_bits, t);
      if (iy < 0 || iy >= p.h || ix < 0 || ix >= p.w) continue;
      const float* src = p.x + (int64_t)ib * p.in_ch * hw + (int64_t)iy * p.w + ix;
      const float* wt = p.wp + (int64_t)t * p.in_ch * p.out_ch + o;
      for (int i = 0; i < p.in_ch; ++i) {
        float v = src[(int64_t)i * hw];
        if (p.style) v *= p.style[(int64_t)ib * p.in_ch + i];
        acc += wt[(int64_t)i * p.out_ch] * v;
      }
    }
    const int64_t pix = (int64_t)(p.sy * yy + p.oy0) * p.ow + (p.sx * xx + p.ox0);
    float v = acc * p.w_scale;
    if (p.demod) v *= p.demod[(int64_t)ib * p.out_ch + o];
    if (p.noise) v += nw * p.noise[(int64_t)ib * ohw + pix];
    if (p.act) {
      v += p.bias[o];
      v = ((v > 0.f) ? v : v * 0.2f) * 1.4142135623730951f;
    }
    p.y[((int64_t)ib * p.out_ch + o) * ohw + pix] = v;
  }
}

static int launch_problem(const ConvProblem& p, int impl, hipStream_t s) {
  const int64_t n_total = (int64_t)p.batch * p.ph * p.pw;
  if (n_total == 0) return 0;
  if (impl == 1) {
    const int64_t total = n_total * p.out_ch;
    hipLaunchKernelGGL(conv_direct_kernel, dim3(rw_stream_grid(total, 256) * 4), dim3(256), 0, s, p);
    return RW_LAUNCH_RESULT();
  }
  if (p.in_ch % RW_KC || p.out_ch % 32) return RW_ERR_UNSUPPORTED;
  if (p.out_ch % 128 == 0) {
    const int64_t grid = rw_cdiv(n_total, 128) * (p.out_ch / 128);
    hipLaunchKernelGGL((conv_mfma_kernel<2, 2, 2, 2>), dim3((unsigned)grid), dim3(256), 0, s, p);
  } else if (p.out_ch % 64 == 0) {
    const int64_t grid = rw_cdiv(n_total, 256) * (p.out_ch / 64);
    hipLaunchKernelGGL((conv_mfma_kernel<2, 2, 1, 4>), dim3((unsigned)grid), dim3(256), 0, s, p);
  } else {
    const int64_t grid = rw_cdiv(n_total, 256) * (p.out_ch / 32);
    hipLaunchKernelGGL((conv_mfma_kernel<1, 2, 1, 4>), dim3((unsigned)grid), dim3(256), 0, s, p);
  }
  return RW_LAUNCH_RESULT();
}

static void fill_common(ConvProblem& p, const float* x, const float* wp, float* y, int batch,
                        int in_ch, int out_ch, int h, int w, float w_scale,
                        const rw_conv_epilogue* ep) {
  p.x = x; p.wp = wp; p.y = y;
  p.style = ep ? ep->style : nullptr;
  p.demod = ep ? ep->demod : nullptr;
  p.noise = ep ? ep->noise : nullptr;
  p.noise_w = ep ? ep->noise_w : nullptr;
  p.bias = ep ? ep->bias : nullptr;
  p.act = ep ? ep->act : 0;
  p.batch = batch; p.in_ch = in_ch; p.out_ch = out_ch; p.h = h; p.w = w;
  p.w_scale = w_scale;
  p.dy_bits = 0; p.dx_bits = 0;
}

static void set_tap(ConvProblem& p, int t, int dy, int dx) {
  p.dy_bits |= (unsigned)(dy + 1) << (2 * t);
  p.dx_bits |= (unsigned)(dx + 1) << (2 * t);
}

extern "C" int rw_conv3x3_f32(const float* x, const float* wp, float* y, int batch, int in_ch,
                              int out_ch, int h, int w, float w_scale, const rw_conv_epilogue* ep,
                              int impl, rw_stream_t stream) {
  RW_CHECK_ARG(x && wp && y && batch > 0 && in_ch > 0 && out_ch > 0 && h > 0 && w > 0);
  RW_CHECK_ARG(!ep || ((!ep->noise || ep->noise_w) && (!ep->act || ep->bias)));
  ConvProblem p;
  fill_common(p, x, wp, y, batch, in_ch, out_ch, h, w, w_scale, ep);
  p.ph = h; p.pw = w; p.oh = h; p.ow = w; p.sy = 1; p.sx = 1; p.oy0 = 0; p.ox0 = 0;
  p.ntaps = 9;
  for (int t = 0; t < 9; ++t) set_tap(p, t, t / 3 - 1, t % 3 - 1);
  return launch_problem(p, impl, rw_s(stream));
}

extern "C" int rw_conv_transpose3x3s2_f32(const float* x, const float* wp, float* y, int batch,
                                          int in_ch, int out_ch, int h, int w, float w_scale,
                                          const rw_conv_epilogue* ep, int impl, rw_stream_t stream) {
  RW_CHECK_ARG(x && wp && y && batch > 0 && in_ch > 0 && out_ch > 0 && h > 0 && w > 0);
  RW_CHECK_ARG(!ep || (!ep->noise && !ep->act));
  // out[2y+ky][2x+kx] += x[y][x] * W[ky][kx]; per output parity: even rows take ky in {0,2}
  // (input row yy, yy-1), odd rows take ky = 1 (input row yy).  Slab order matches
  // rw_pack_conv_weight_f32 mode 1.
  static const int ntaps[4] = {4, 2, 2, 1};
  static const int slab0[4] = {0, 4, 6, 8};
  static const int tdy[4][4] = {{0, 0, -1, -1}, {0, -1, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  static const int tdx[4][4] = {{0, -1, 0, -1}, {0, 0, 0, 0}, {0, -1, 0, 0}, {0, 0, 0, 0}};
  const int64_t slab = (int64_t)in_ch * out_ch;
  for (int phase = 0; phase < 4; ++phase) {
    const int py = phase >> 1, px = phase & 1;
    ConvProblem p;
    fill_common(p, x, wp + slab0[phase] * slab, y, batch, in_ch, out_ch, h, w, w_scale, ep);
    p.ph = py ? h : h + 1; p.pw = px ? w : w + 1;
    p.oh = 2 * h + 1; p.ow = 2 * w + 1; p.sy = 2; p.sx = 2; p.oy0 = py; p.ox0 = px;
    p.ntaps = ntaps[phase];
    for (int t = 0; t < p.ntaps; ++t) set_tap(p, t, tdy[phase][t], tdx[phase][t]);
    const int rc = launch_problem(p, impl, rw_s(stream));
    if (rc) return rc;
  }
  return 0;
}
