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
#include <stdlib.h>
#include <string.h>

__host__ __device__ __forceinline__ int rw_tap_off(unsigned bits, int t) {
  return (int)((bits >> (2 * t)) & 3u) - 1;
}

struct ConvProblem {
  const float* x; const float* wp; float* y;
  const float* style; const float* demod; const float* noise; const float* noise_w; const float* bias;
  int batch, in_ch, out_ch, h, w;   // input tensor
  int ph, pw;                       // positions per image in this problem
  int yoff, xoff;                   // origin of the position grid: position = (yy + yoff, xx + xoff)
  int oh, ow;                       // output tensor
  int sy, sx, oy0, ox0;             // output pixel = (sy*yy + oy0, sx*xx + ox0)
  int ntaps;
  unsigned dy_bits, dx_bits;        // 2 bits per tap: (dy + 1), (dx + 1); input pixel = (yy + dy, xx + dx)
  float w_scale;
  int act;
};

#define RW_KC 16

// Up to four problems that share x / y / out_ch (the parity phases of a transposed conv, or the
// border strips of its halo variant) executed by ONE launch: workgroup -> problem by work0[].
struct ConvBatch {
  int n;
  int work0[5];
  ConvProblem p[4];
};

// Block id -> work item so that consecutive work items (the out-channel tiles of one pixel
// tile, which share the gathered input) sit on ONE XCD's L2.  Bijective for any total.
__device__ __forceinline__ int rw_xcd_remap(int id, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, slot = id >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// Epilogue of the im2col kernels for a wave's TM x TN accumulator tiles (C/D layout: col = lane&31
// is the position n, row r of a tile is out-channel o_first + 32a + (r&3) + 8(r>>2)).  Every value
// it reads is fetched in batches behind ONE uniform branch each: a load per store, each behind its
// own `if`, costs an L2 round trip plus the drain of the previous store (vmcnt counts both).
template <int TM_, int TN_>
__device__ __forceinline__ void rw_tile_epilogue(const ConvProblem& p, const rw_f32x16 (&acc)[TM_][TN_],
                                                 int o_first, int64_t n_first, int ppi, int64_t n_total) {
  const int64_t ohw = (int64_t)p.oh * p.ow;
  int ib[TN_];
  int64_t pix[TN_];
  bool live[TN_];
  float nz[TN_];
#pragma unroll
  for (int b = 0; b < TN_; ++b) {
    const int64_t n = n_first + 32 * b;
    live[b] = n < n_total;
    const int64_t nn = live[b] ? n : 0;
    ib[b] = (int)(nn / ppi);
    const int r0 = (int)(nn - (int64_t)ib[b] * ppi);
    const int yq = r0 / p.pw;
    const int yy = yq + p.yoff, xx = r0 - yq * p.pw + p.xoff;
    pix[b] = (int64_t)(p.sy * yy + p.oy0) * p.ow + (p.sx * xx + p.ox0);
    nz[b] = 0.f;
  }
  if (p.noise) {
    const float nw = p.noise_w[0];
#pragma unroll
    for (int b = 0; b < TN_; ++b) nz[b] = p.noise[(int64_t)ib[b] * ohw + pix[b]];
#pragma unroll
    for (int b = 0; b < TN_; ++b) nz[b] *= nw;
  }
#pragma unroll
  for (int a = 0; a < TM_; ++a) {
    const int ob = o_first + 32 * a;
    float bias[16];
    if (p.act) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bias[r] = p.bias[ob + (r & 3) + 8 * (r >> 2)];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) bias[r] = 0.f;
    }
#pragma unroll
    for (int b = 0; b < TN_; ++b) {
      float scale[16];
      if (p.demod) {
        const float* dm = p.demod + (int64_t)ib[b] * p.out_ch + ob;
#pragma unroll
        for (int r = 0; r < 16; ++r) scale[r] = dm[(r & 3) + 8 * (r >> 2)];
#pragma unroll
        for (int r = 0; r < 16; ++r) scale[r] *= p.w_scale;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) scale[r] = p.w_scale;
      }
      if (!live[b]) continue;
      float* yo = p.y + ((int64_t)ib[b] * p.out_ch + ob) * ohw + pix[b];
      if (p.act) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[a][b][r] * scale[r] + nz[b] + bias[r];
          yo[((r & 3) + 8 * (r >> 2)) * ohw] = ((v > 0.f) ? v : v * 0.2f) * 1.4142135623730951f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) yo[((r & 3) + 8 * (r >> 2)) * ohw] = acc[a][b][r] * scale[r] + nz[b];
      }
    }
  }
}

template <int TM, int TN, int WGM, int WGN>
__global__ void __launch_bounds__(256) conv_mfma_kernel(const ConvBatch cb) {
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

  int work = rw_xcd_remap(blockIdx.x, gridDim.x);
  int q = 0;
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (j < cb.n && work >= cb.work0[j]) q = j;
  work -= cb.work0[q];
  const ConvProblem& p = cb.p[q];
  const int o_tiles = p.out_ch / BM;
  const int ppi = p.ph * p.pw;
  const int64_t n_total = (int64_t)p.batch * ppi;
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
    gx = r - gy * p.pw + p.xoff;
    gy += p.yoff;
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
    // fragments one k-pair ahead of the MFMAs that use them, pinned there (see conv_halo_kernel)
    float af[TM], bf[TN], an[TM], bn[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) af[a] = As[buf][frow][wm0 + 32 * a + fcol];
#pragma unroll
    for (int b = 0; b < TN; ++b) bf[b] = Bs[buf][frow][wn0 + 32 * b + fcol];
#pragma unroll
    for (int kp = 0; kp < RW_KC / 2; ++kp) {
      if (kp + 1 < RW_KC / 2) {
#pragma unroll
        for (int a = 0; a < TM; ++a) an[a] = As[buf][2 * kp + 2 + frow][wm0 + 32 * a + fcol];
#pragma unroll
        for (int b = 0; b < TN; ++b) bn[b] = Bs[buf][2 * kp + 2 + frow][wn0 + 32 * b + fcol];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < TM; ++a) af[a] = an[a];
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[b] = bn[b];
    }
    if (c + 1 < n_chunks) stash(buf ^ 1);
    __syncthreads();
  }

  rw_tile_epilogue<TM, TN>(p, acc, o0 + wm0 + 4 * frow, n0 + wn0 + fcol, ppi, n_total);
}

// ---------------------------------------------------------------------------------------
// Split-K variant for LOW-RESOLUTION layers (4x4 .. 16x16 maps at small batch): there the
// implicit GEMM has few output tiles (N = B*H*W is a few hundred) but a long K = 9*Cin = 4608,
// so the tiling above leaves most of the 256 CUs idle behind a handful of long serial K loops.
// Here a workgroup owns one small (32T x 32T) output tile and its FOUR WAVES SPLIT K: all
// threads stage a chunk of 4*KW k-values, wave w multiplies k-slice w, and the four partial
// tiles are summed through LDS (reusing the staging buffers) before wave 0 runs the epilogue.
// ---------------------------------------------------------------------------------------
template <int T, int KW>
__global__ void __launch_bounds__(256) conv_mfma_ksplit_kernel(const ConvBatch cb) {
  constexpr int BM = 32 * T, BN = 32 * T, KC = 4 * KW;
  constexpr int B_ELEMS = KC * BN / 256;
  constexpr int B_KSTEP = 256 / BN;
  constexpr int A_VEC = KC * BM / 4 / 256;
  static_assert(KC * BM / 4 % 256 == 0 && 2 * KC * (BM + BN) >= 2 * BM * BN, "tile/chunk shape");
  __shared__ __attribute__((aligned(16))) float smem[2 * KC * BM + 2 * KC * BN];
  float(*As)[KC][BM] = reinterpret_cast<float(*)[KC][BM]>(smem);
  float(*Bs)[KC][BN] = reinterpret_cast<float(*)[KC][BN]>(smem + 2 * KC * BM);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int work = rw_xcd_remap(blockIdx.x, gridDim.x);
  int q = 0;
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (j < cb.n && work >= cb.work0[j]) q = j;
  work -= cb.work0[q];
  const ConvProblem& p = cb.p[q];
  const int o_tiles = p.out_ch / BM;
  const int ppi = p.ph * p.pw;
  const int64_t n_total = (int64_t)p.batch * ppi;
  const int o0 = (work % o_tiles) * BM;
  const int64_t n0 = (int64_t)(work / o_tiles) * BN;

  const int nl = tid % BN;
  const int kk0 = tid / BN;
  const int64_t n_mine = n0 + nl;
  const bool n_ok = n_mine < n_total;
  int gb = 0, gy = 0, gx = 0;
  if (n_ok) {
    gb = (int)(n_mine / ppi);
    const int r = (int)(n_mine - (int64_t)gb * ppi);
    gy = r / p.pw;
    gx = r - gy * p.pw + p.xoff;
    gy += p.yoff;
  }
  const int64_t hw = (int64_t)p.h * p.w;
  const float* xb = p.x + (int64_t)gb * p.in_ch * hw;
  const float* sb = p.style ? p.style + (int64_t)gb * p.in_ch : nullptr;
  const int chunks_per_tap = p.in_ch / KC;
  const int n_chunks = chunks_per_tap * p.ntaps;

  // Chunks are fetched DEPTH ahead into a ring of register sets: a chunk is only 16 MFMAs per wave (~0.5 us), one
  // chunk of look-ahead left the loop at one memory latency per chunk (260 us for layer 2's 144 chunks)
  constexpr int DEPTH = 4;
  float bring[DEPTH][B_ELEMS];
  rw_f32x4 aring[DEPTH][A_VEC];
  auto gather = [&](int c, float (&breg)[B_ELEMS], rw_f32x4 (&areg)[A_VEC]) __attribute__((always_inline)) {
    const int t = c % p.ntaps;
    const int i0 = (c / p.ntaps) * KC;
    const int iy = gy + rw_tap_off(p.dy_bits, t), ix = gx + rw_tap_off(p.dx_bits, t);
    const bool ok = n_ok && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
    const float* src = xb + (int64_t)i0 * hw + (int64_t)(ok ? iy : 0) * p.w + (ok ? ix : 0);
    const float m = ok ? 1.0f : 0.0f;
#pragma unroll
    for (int j = 0; j < B_ELEMS; ++j) {
      const int kk = kk0 + j * B_KSTEP;
      float v = src[(int64_t)kk * hw] * m;
      if (sb) v *= sb[i0 + kk];
      breg[j] = v;
    }
    const float* wrow = p.wp + ((int64_t)t * p.in_ch + i0) * p.out_ch + o0;
#pragma unroll
    for (int j = 0; j < A_VEC; ++j) {
      const int qq = tid + j * 256;
      const int kk = qq / (BM / 4), o4 = qq % (BM / 4);
      areg[j] = *reinterpret_cast<const rw_f32x4*>(wrow + (int64_t)kk * p.out_ch + o4 * 4);
    }
  };
  auto stash = [&](int buf, const float (&breg)[B_ELEMS], const rw_f32x4 (&areg)[A_VEC]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < B_ELEMS; ++j) Bs[buf][kk0 + j * B_KSTEP][nl] = breg[j];
#pragma unroll
    for (int j = 0; j < A_VEC; ++j) {
      const int qq = tid + j * 256;
      const int kk = qq / (BM / 4), o4 = qq % (BM / 4);
      *reinterpret_cast<rw_f32x4*>(&As[buf][kk][o4 * 4]) = areg[j];
    }
  };

  rw_f32x16 acc[T][T];
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < n_chunks) gather(d, bring[d], aring[d]);
  stash(0, bring[0], aring[0]);
  __syncthreads();
  const int frow = lane >> 5, fcol = lane & 31;
  for (int c0 = 0; c0 < n_chunks; c0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int c = c0 + d;
      if (c < n_chunks) {
        const int buf = c & 1;
        // ring slot d held chunk c, which the previous iteration stored to LDS: free for chunk c + DEPTH
        if (c > 0 && c - 1 + DEPTH < n_chunks) gather(c - 1 + DEPTH, bring[(d + DEPTH - 1) % DEPTH], aring[(d + DEPTH - 1) % DEPTH]);
#pragma unroll
        for (int kp = 0; kp < KW / 2; ++kp) {
          const int kr = wave * KW + 2 * kp + frow;
          float af[T], bf[T];
#pragma unroll
          for (int a = 0; a < T; ++a) af[a] = As[buf][kr][32 * a + fcol];
#pragma unroll
          for (int b = 0; b < T; ++b) bf[b] = Bs[buf][kr][32 * b + fcol];
#pragma unroll
          for (int a = 0; a < T; ++a)
#pragma unroll
            for (int b = 0; b < T; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (c + 1 < n_chunks) stash(buf ^ 1, bring[(d + 1) % DEPTH], aring[(d + 1) % DEPTH]);
        // LDS writes complete + barrier; NOT __syncthreads(), whose vmcnt(0) would drain the ring every chunk
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
      }
    }
  }

  // ---- cross-wave reduction tree through LDS: (2,3) -> (0,1), then 1 -> 0
  float* red = smem;
  auto slot = [&](int a, int b, int r) { return ((a * T + b) * 16 + r) * 64 + lane; };
  if (wave >= 2) {
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int b = 0; b < T; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave - 2) * BM * BN + slot(a, b, r)] = acc[a][b][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int b = 0; b < T; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] += red[wave * BM * BN + slot(a, b, r)];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int b = 0; b < T; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[slot(a, b, r)] = acc[a][b][r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] += red[slot(a, b, r)];

  rw_tile_epilogue<T, T>(p, acc, o0 + 4 * frow, n0 + fcol, ppi, n_total);
}

// ---------------------------------------------------------------------------------------
// Halo-tile variant for feature maps at least one MFMA tile wide (W >= 24): the workgroup owns
// a (TH rows x 32 columns) patch of ONE image and BM out-channels.  Per chunk of IC input
// channels the (TH+2) x 34 input halo is staged in LDS ONCE (coalesced NCHW row pieces, style
// multiplied on the way in) and all taps read it at shifted addresses, so global gathers drop
// 9x versus the im2col staging above and there is one barrier per 9*IC/2 MFMA steps instead
// of per 8.  Weight fragments are not staged at all: the weights are also kept in MFMA fragment
// order (rw_pack_conv_weight_f32), so every lane fetches its A operand for four k-pairs with ONE
// 16-byte load (1 KiB of consecutive addresses per wave) straight from L1/L2, half a tap ahead of
// its use, which frees the LDS and the VALU for the matrix pipe.  (FRAG = false: the per-phase
// variant of the transposed convolution, impl 4, reads wp[tap][i][o] with dword loads.)
// ---------------------------------------------------------------------------------------
template <int N> struct rw_int { static constexpr int value = N; };

// Timing ablations for kernel work (build with -DRW_ABLATION, select with RW_CONV_ABL=<bits>; results
// are WRONG when a bit is set): 1 = no LDS operand reads in the loop, 2 = no weight refills,
// 4 = no halo fetch / staging, 8 = no barrier, 16 = no epilogue stores.  Compiled out otherwise.
#ifdef RW_ABLATION
#include <stdlib.h>
#define RW_ABL(p, bit) ((p).abl & (bit))
static int rw_abl_env() { const char* e = getenv("RW_CONV_ABL"); return e ? atoi(e) : 0; }
#else
#define RW_ABL(p, bit) false
static int rw_abl_env() { return 0; }
#endif

struct PhaseDesc {
  int ntaps;
  unsigned dy_bits, dx_bits;
  int ph, pw, oy0, ox0;
  int tiles_x, tiles_y, work0;
  long long wp_off;
};

struct HaloProblem {
  const float* x; const float* wp; const float* wfrag; float* y;
  const float* style; const float* demod; const float* noise; const float* noise_w; const float* bias;
  int batch, in_ch, out_ch, h, w, oh, ow, sy, sx;
  float w_scale;
  int act, nphase, abl;
  // ToRGB fused into the epilogue (rw_conv3x3_to_rgb_f32; the workgroup holds ALL out-channels): nullable
  const float* rgb_weight; const float* rgb_style; const float* rgb_bias; const float* rgb_skip; float* rgb_out;
  float rgb_scale;
  PhaseDesc phase[4];
};

// Epilogue of the stride-1 halo kernels for a wave's TM x TN accumulator tiles: row r of tile a is
// out-channel o_first + 32a + (r&3) + 8(r>>2), tile b is image row y_first + b * RPT, the lane's
// column is xx.  Everything it reads is fetched in batches behind ONE uniform branch each: a load per
// store, each behind its own `if`, costs an L2 round trip plus the drain of the previous store
// (vmcnt counts both) per element -- as long as the whole K loop on the 32/64-channel layers.
template <int TM, int TN, int RPT>
__device__ __forceinline__ void rw_halo_epilogue(const HaloProblem& p, const PhaseDesc& d,
                                                 const rw_f32x16 (&acc)[TM][TN], int ib, int o_first,
                                                 int y_first, int xx) {
  const int64_t ohw = (int64_t)p.oh * p.ow;
  int64_t pix[TN];
  float nz[TN];
  bool live[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int yy = y_first + b * RPT;
    live[b] = yy < d.ph && xx < d.pw;
    pix[b] = live[b] ? (int64_t)(p.sy * yy + d.oy0) * p.ow + (p.sx * xx + d.ox0) : 0;
    nz[b] = 0.f;
  }
  if (p.noise) {
    const float nw = p.noise_w[0];
    const float* np = p.noise + (int64_t)ib * ohw;
#pragma unroll
    for (int b = 0; b < TN; ++b) nz[b] = np[pix[b]];
#pragma unroll
    for (int b = 0; b < TN; ++b) nz[b] *= nw;
  }
  // Fused ToRGB (models.py:639-655): rgb[c] = sum_o (s W[c][o] style[b][o]) * out[o] + bias[c] + skip.  The
  // workgroup holds every out-channel of its pixels: a lane sums its 16 * TM channels, the partner lane
  // (other half of the wave, same pixel) the rest.
  const bool rgb = p.rgb_weight != nullptr;
  float part[TN][3];
#pragma unroll
  for (int b = 0; b < TN; ++b) part[b][0] = part[b][1] = part[b][2] = 0.f;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int ob = o_first + 32 * a;
    float scale[16], bias[16];
    if (p.demod) {
      const float* dm = p.demod + (int64_t)ib * p.out_ch + ob;
#pragma unroll
      for (int r = 0; r < 16; ++r) scale[r] = dm[(r & 3) + 8 * (r >> 2)];
#pragma unroll
      for (int r = 0; r < 16; ++r) scale[r] *= p.w_scale;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) scale[r] = p.w_scale;
    }
    if (p.act) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bias[r] = p.bias[ob + (r & 3) + 8 * (r >> 2)];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) bias[r] = 0.f;
    }
    float wr[3][16];
    if (rgb) {
      float sr[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sr[r] = p.rgb_style[(int64_t)ib * p.out_ch + ob + (r & 3) + 8 * (r >> 2)];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) wr[c][r] = p.rgb_weight[c * p.out_ch + ob + (r & 3) + 8 * (r >> 2)];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) wr[c][r] = p.rgb_scale * wr[c][r] * sr[r];
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      if (!live[b]) continue;
      if (RW_ABL(p, 16) && acc[0][b][0] != 12345.f) continue;
      float* yo = p.y ? p.y + ((int64_t)ib * p.out_ch + ob) * ohw + pix[b] : nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[a][b][r] * scale[r] + nz[b];
        if (p.act) {
          v += bias[r];
          v = ((v > 0.f) ? v : v * 0.2f) * 1.4142135623730951f;
        }
        if (yo) yo[((r & 3) + 8 * (r >> 2)) * ohw] = v;
        if (rgb) {
#pragma unroll
          for (int c = 0; c < 3; ++c) part[b][c] += v * wr[c][r];
        }
      }
    }
  }
  if (rgb) {
    const bool low_half = (threadIdx.x & 32) == 0;
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) part[b][c] += __shfl_xor(part[b][c], 32, 64);
    // bias and the running image are fetched in one batch each (not one dependent load per store)
    float rb[3] = {0.f, 0.f, 0.f}, sk[TN][3];
    if (p.rgb_bias) {
#pragma unroll
      for (int c = 0; c < 3; ++c) rb[c] = p.rgb_bias[c];
    }
    if (p.rgb_skip) {
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) sk[b][c] = p.rgb_skip[((int64_t)ib * 3 + c) * ohw + pix[b]];
    } else {
#pragma unroll
      for (int b = 0; b < TN; ++b) sk[b][0] = sk[b][1] = sk[b][2] = 0.f;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      if (!live[b] || !low_half) continue;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        p.rgb_out[((int64_t)ib * 3 + c) * ohw + pix[b]] = part[b][c] + rb[c] + sk[b][c];
    }
  }
}

// TW = 32: an MFMA column tile is 32 consecutive pixels of one row; TW = 16 (maps 9..16 wide): two
// rows of 16; TW = 8 (maps 5..8 wide): four rows of 8 -- the low-resolution layers keep every lane of
// the tile busy.
template <int TM, int TN, int WGM, int WGN, int IC, bool FRAG, int TW>
__global__ void __launch_bounds__(256, 2) conv_halo_kernel(const HaloProblem p) {
  constexpr int BM = 32 * TM * WGM;
  constexpr int RPT = 32 / TW;                // image rows per 32-lane column tile
  constexpr int TH = TN * WGN * RPT;
  constexpr int XH = TH + 2, XW = TW == 32 ? 36 : (TW == 16 ? 20 : 12), XUSED = TW + 2;
  constexpr int KP = IC / 2;
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  __shared__ float Xs[2][IC][XH][XW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WGN) * 32 * TM;
  const int wrow0 = (wave % WGN) * TN * RPT;
  const int frow = lane >> 5, fcol = lane & 31;
  const int lc = fcol & (TW - 1), lr = fcol / TW;     // this lane's column / row inside a column tile

  const int work = rw_xcd_remap(blockIdx.x, gridDim.x);
  int phase = 0;
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (q < p.nphase && work >= p.phase[q].work0) phase = q;
  const PhaseDesc d = p.phase[phase];
  int local = work - d.work0;
  const int o_tiles = p.out_ch / BM;
  const int o0 = (local % o_tiles) * BM; local /= o_tiles;
  const int tx = local % d.tiles_x; local /= d.tiles_x;
  const int ty = local % d.tiles_y;
  const int ib = local / d.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int64_t hw = (int64_t)p.h * p.w;
  const float* xb = p.x + (int64_t)ib * p.in_ch * hw;
  const float* st = p.style ? p.style + (int64_t)ib * p.in_ch : nullptr;     // uniform: scalar loads
  const float* wph = p.wp + d.wp_off + o0;            // wave-uniform
  const int a_lane = frow * p.out_ch + wm0 + fcol;    // this lane's offset inside a k-pair of rows

  // Halo staging: this thread owns up to PSLOT fixed positions (r, c) of the (TH+2) x 34 patch and
  // walks the IC channels of a chunk for each: one 32-bit offset per slot, channel stride uniform.
  // Positions outside the image are loaded from a legal address and multiplied by 0, slots past the
  // patch are written to a padding column nobody reads, so the staging is branch-free.
  constexpr int NPOS = XH * XUSED;
  constexpr int PSLOT = (NPOS + 255) / 256;
  int xoff[PSLOT], xlds[PSLOT];
  float xmask[PSLOT];
#pragma unroll
  for (int sl = 0; sl < PSLOT; ++sl) {
    const int pos = tid + 256 * sl;
    const int r = pos / XUSED, c = pos - r * XUSED;
    const int iy = y0 - 1 + r, ix = x0 - 1 + c;
    const bool ok = pos < NPOS && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
    xoff[sl] = ok ? iy * p.w + ix : 0;            // invalid -> any legal address, value masked to 0
    xmask[sl] = ok ? 1.0f : 0.0f;
    xlds[sl] = pos < NPOS ? r * XW + c : XW - 1;
  }
  float xreg[PSLOT][IC];
  float sty[IC];                            // style of the chunk held in xreg (SGPRs; 1.0 when not fused)
  auto xfetch = [&](int i0) {
    const float* xc = xb + (int64_t)i0 * hw;               // uniform
#pragma unroll
    for (int ic = 0; ic < IC; ++ic)
#pragma unroll
      for (int sl = 0; sl < PSLOT; ++sl) xreg[sl][ic] = xc[(int64_t)ic * hw + xoff[sl]];
    if (st) {
#pragma unroll
      for (int ic = 0; ic < IC; ++ic) sty[ic] = st[i0 + ic];
    } else {
#pragma unroll
      for (int ic = 0; ic < IC; ++ic) sty[ic] = 1.0f;
    }
  };
  // One element of the staging (mask + style applied on the way into LDS).  In the stride-1 kernel
  // the steps ride between the MFMAs of the last two taps of a chunk, so there is no VALU/LDS-only
  // phase in front of the barrier.
  constexpr int NST = PSLOT * IC;
  auto stash_step = [&](int buf, int j) {
    const int sl = j / IC, ic = j % IC;
    (&Xs[buf][0][0][0])[ic * XH * XW + xlds[sl]] = xreg[sl][ic] * (xmask[sl] * sty[ic]);
  };
  // A operand.  FRAG: the weights come in fragment order (rw_pack_conv_weight_f32): one 16-byte load
  // per lane fetches four k-pairs and a wave's load covers 1 KiB of consecutive addresses, so a tap
  // costs TM * KP/4 load instructions.  With KP = 8 each half is refilled right after its last use,
  // four k-pairs (>= 1k cycles of MFMA) ahead of the next tap; with KP = 4 the next tap is
  // double-buffered.  !FRAG (per-phase transposed conv, impl 4): per-lane dword loads of wp[tap][i][o].
  constexpr int KH = KP / 4;
  rw_f32x4 av[KH][TM], an[TM];
  const int n_chunks = p.in_ch / IC;
  const int c_stride = (p.out_ch >> 5) * KH * 256;
  const int64_t t_stride = (int64_t)n_chunks * c_stride;
  const float* wf = FRAG ? p.wfrag + (int64_t)((o0 + wm0) >> 5) * KH * 256 : nullptr;   // uniform
  auto fload = [&](rw_f32x4 (&dst)[TM], int h, int t, int c) {
    const float* base = wf + (int64_t)t * t_stride + (int64_t)c * c_stride;
#pragma unroll
    for (int a = 0; a < TM; ++a)
      dst[a] = *reinterpret_cast<const rw_f32x4*>(base + (a * KH + h) * 256 + lane * 4);
  };
  auto aload1 = [&](int kp, int t, int i0) {
    const float* base = wph + ((int64_t)t * p.in_ch + i0) * p.out_ch;   // uniform: SGPR base + 32-bit lane offset
#pragma unroll
    for (int a = 0; a < TM; ++a) av[kp >> 2][a][kp & 3] = base[a_lane + (2 * kp) * p.out_ch + 32 * a];
  };

  rw_f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  xfetch(0);
  if (FRAG) {
#pragma unroll
    for (int h = 0; h < KH; ++h) fload(av[h], h, 0, 0);
  } else {
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) aload1(kp, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < NST; ++j) stash_step(0, j);
  __syncthreads();
  // The loop body is straight-line code: prefetches past the end are clamped to valid (unused)
  // addresses instead of being branched around, and __builtin_amdgcn_sched_barrier pins every
  // load BEFORE the MFMA group that covers its latency (left alone, the scheduler sinks the loads
  // next to their first use and each k-pair waits for a full L2 / LDS round trip).
  constexpr int MPK = TM * TN;               // MFMAs per k-pair
  static_assert(NST <= 2 * KP * MPK, "staging steps fit the last two taps");
  for (int c = 0; c < n_chunks; ++c) {
    const int buf = c & 1;
    const int cn = c + 1 < n_chunks ? c + 1 : c;      // last chunk: a redundant, unused refill
    if (!RW_ABL(p, 4)) xfetch(cn * IC);    // consumed by the staging steps at the end of the chunk
    const float* xs = &Xs[buf][frow][wrow0 + lr + rw_tap_off(d.dy_bits, 0) + 1][lc + rw_tap_off(d.dx_bits, 0) + 1];
    float bf[TN], bnext[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) bf[b] = xs[b * RPT * XW];
    // one tap; STAGE >= 0 carries staging steps [STAGE * KP * MPK, (STAGE + 1) * KP * MPK)
    auto tap = [&](int t, auto stage_tag) {
      constexpr int STAGE = decltype(stage_tag)::value;
      int nt = t + 1, nc = c;
      if (nt == d.ntaps) { nt = 0; nc = cn; }
      // first k-pair of the next tap (for the last tap: re-read after the barrier, this one is unused)
      const float* xs_next = &Xs[buf][frow][wrow0 + lr + rw_tap_off(d.dy_bits, nt) + 1][lc + rw_tap_off(d.dx_bits, nt) + 1];
      if (FRAG && KH == 1 && !RW_ABL(p, 2)) fload(an, 0, nt, nc);
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) {
#pragma unroll
        for (int b = 0; b < TN; ++b)       // B fragments one k-pair ahead of the MFMAs that use them
          if (!RW_ABL(p, 1)) bnext[b] = kp + 1 < KP ? xs[(2 * kp + 2) * XH * XW + b * RPT * XW] : xs_next[b * RPT * XW];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp >> 2][a][kp & 3], bf[b], acc[a][b], 0, 0, 0);
            if (STAGE >= 0 && (STAGE * KP + kp) * MPK + a * TN + b < NST && !RW_ABL(p, 4)) {
              stash_step(buf ^ 1, (STAGE * KP + kp) * MPK + a * TN + b);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        // these weight registers are free again: refill them for the next tap, >= 1k cycles of MFMA
        // ahead of their use
        if (RW_ABL(p, 2)) {
        } else if (!FRAG) aload1(kp, nt, nc * IC);
        else if (KH > 1 && (kp & 3) == 3) fload(av[kp >> 2], kp >> 2, nt, nc);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[b] = bnext[b];
      }
      if (FRAG && KH == 1) {
#pragma unroll
        for (int a = 0; a < TM; ++a) av[0][a] = an[a];
      }
      xs = xs_next;
    };
    if (FRAG) {                            // stride-1 convolution: always nine taps
      for (int t = 0; t < 7; ++t) tap(t, rw_int<-1>());
      tap(7, rw_int<0>());
      tap(8, rw_int<1>());
    } else {
      for (int t = 0; t < d.ntaps; ++t) tap(t, rw_int<-1>());
#pragma unroll
      for (int j = 0; j < NST; ++j) stash_step(buf ^ 1, j);
    }
    if (!RW_ABL(p, 8)) __syncthreads();
  }

  rw_halo_epilogue<TM, TN, RPT>(p, d, acc, ib, o0 + wm0 + 4 * frow, y0 + wrow0 + lr, x0 + lc);
}

// ---------------------------------------------------------------------------------------
// Stride-2 transposed convolution, all four output-parity phases in ONE workgroup.
// The workgroup owns BM out-channels and a (TH x 32) patch of the (H+1) x (W+1) grid of 2x2
// output quads.  Per chunk of IC input channels the (TH+1) x 33 input halo (rows y-1..y,
// columns x-1..x) is staged once; the four distinct input shifts {0,-1}^2 are read from LDS once
// per k-pair and feed the nine weight slabs (4+2+2+1 taps of the four phases), so each wave keeps
// four accumulator sets (one per phase) and the epilogue writes COMPLETE output rows: even rows
// as aligned 8-byte (px=0, px=1) pairs, i.e. 256 contiguous bytes per half-wave.
// Quads y < H, x < W are tiled exactly; output row 2H and column 2W (the "+1" of 2H+1) are a
// strip of H+W+1 quads computed by four strip problems of one batched im2col launch.
// ---------------------------------------------------------------------------------------
struct UpProblem {
  const float* x; const float* wfrag; float* y;
  const float* style; const float* demod;
  int batch, in_ch, out_ch, h, w;
  int tiles_x, tiles_y;
  float w_scale;
  int abl;
};

template <int WGM, int WGN, int IC, int TW>
__global__ void __launch_bounds__(256, 2) conv_up_halo_kernel(const UpProblem p) {
  constexpr int TN = 2;                       // rows of quads per wave
  constexpr int BM = 32 * WGM;
  constexpr int RPT = 32 / TW;                // quad rows per 32-lane column tile (TW = 16: two, narrow maps)
  constexpr int TH = TN * WGN * RPT;
  constexpr int XH = TH + 1, XW = TW == 32 ? 36 : (TW == 16 ? 20 : 12), XUSED = TW + 1;
  constexpr int NPOS = XH * XUSED;
  constexpr int PSLOT = (NPOS + 255) / 256;
  constexpr int KP = IC / 2;
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  __shared__ float Xs[2][IC][XH][XW];
  __shared__ float Sc[BM];                    // w_scale * demod of the workgroup's out-channels (see the epilogue)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WGN) * 32;
  const int wrow0 = (wave % WGN) * TN * RPT;
  const int frow = lane >> 5, fcol = lane & 31;
  const int lc = fcol & (TW - 1), lr = fcol / TW;

  int local = rw_xcd_remap(blockIdx.x, gridDim.x);
  const int o_tiles = p.out_ch / BM;
  const int o0 = (local % o_tiles) * BM; local /= o_tiles;
  const int tx = local % p.tiles_x; local /= p.tiles_x;
  const int ty = local % p.tiles_y;
  const int ib = local / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int64_t hw = (int64_t)p.h * p.w;
  const float* xb = p.x + (int64_t)ib * p.in_ch * hw;
  const float* st = p.style ? p.style + (int64_t)ib * p.in_ch : nullptr;     // uniform: scalar loads
  // weights in fragment order (rw_pack_conv_weight_f32 mode 1): [chunk][o / 32][kp][576]
  const float* wf = p.wfrag + (int64_t)((o0 + wm0) >> 5) * KP * 576;    // uniform
  const int c_stride = (p.out_ch >> 5) * KP * 576;
  // A global load in the epilogue would queue behind whatever the loop issued last (vmcnt retires in order) and cost
  // a memory latency per workgroup: the per-channel factors go to LDS up front (visible after the first barrier).
  if (tid < BM) Sc[tid] = p.demod ? p.demod[(int64_t)ib * p.out_ch + o0 + tid] * p.w_scale : p.w_scale;

  // Halo staging: this thread owns up to PSLOT fixed positions of the (TH+1) x 33 patch; positions
  // outside the image are loaded from a legal address and multiplied by 0, slots past the patch
  // are written to a padding column nobody reads, so the staging is branch-free.
  int xoff[PSLOT], xlds[PSLOT];
  float xmask[PSLOT];
#pragma unroll
  for (int sl = 0; sl < PSLOT; ++sl) {
    const int pos = tid + 256 * sl;
    const int r = pos / XUSED, c = pos - r * XUSED;
    const int iy = y0 - 1 + r, ix = x0 - 1 + c;
    const bool ok = pos < NPOS && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
    xoff[sl] = ok ? iy * p.w + ix : 0;
    xmask[sl] = ok ? 1.0f : 0.0f;
    xlds[sl] = pos < NPOS ? r * XW + c : XW - 1;
  }
  float xreg[PSLOT][IC];
  float sty[IC];                            // style of the chunk held in xreg (SGPRs)
  auto xfetch = [&](int i0) {
    const float* xc = xb + (int64_t)i0 * hw;
#pragma unroll
    for (int ic = 0; ic < IC; ++ic)
#pragma unroll
      for (int sl = 0; sl < PSLOT; ++sl) xreg[sl][ic] = xc[(int64_t)ic * hw + xoff[sl]];
    if (st) {
#pragma unroll
      for (int ic = 0; ic < IC; ++ic) sty[ic] = st[i0 + ic];
    } else {
#pragma unroll
      for (int ic = 0; ic < IC; ++ic) sty[ic] = 1.0f;
    }
  };
  // one element of the staging (style and the zero padding applied on the way into LDS); the steps
  // are spread between the MFMAs of the last k-pairs of a chunk instead of forming a VALU/LDS-only
  // phase in front of the barrier
  constexpr int NST = PSLOT * IC;
  auto stash_step = [&](int buf, int j) {
    const int sl = j / IC, ic = j % IC;
    (&Xs[buf][0][0][0])[ic * XH * XW + xlds[sl]] = xreg[sl][ic] * (xmask[sl] * sty[ic]);
  };
  // nine weight slabs of one k-pair: two 16-byte loads (slabs 0-3, 4-7; 1 KiB per wave each) and a
  // dword (slab 8) per lane
  // The slabs are prefetched THREE k-pairs ahead through a ring of four register sets.  vmcnt retires in order: a
  // weight load issued right behind the patch fetch of a chunk (48 loads from HBM, ~2 us) cannot be waited for before
  // that fetch has landed, and with a one-ahead prefetch every wave stalled on it once per chunk (measured: loads
  // cost 15 - 22 % of the kernel).  Three ahead, the first slab issued behind the fetch is consumed four k-pairs
  // later -- when the staging steps need the patch anyway.
  struct ASlabs { rw_f32x4 v4[2]; float s8; };
  static_assert(KP % 4 == 0, "ring of four slab sets");
  ASlabs aring[4];
  auto aload = [&](ASlabs& dst, int kp, int c) {
    const float* base = wf + (int64_t)c * c_stride + kp * 576;
    dst.v4[0] = *reinterpret_cast<const rw_f32x4*>(base + lane * 4);
    dst.v4[1] = *reinterpret_cast<const rw_f32x4*>(base + 256 + lane * 4);
    dst.s8 = base[512 + lane];
  };

  rw_f32x16 acc[4][TN];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][b][r] = 0.f;

  const int n_chunks = p.in_ch / IC;
  xfetch(0);
  aload(aring[0], 0, 0);
  aload(aring[1], 1, 0);
  aload(aring[2], 2, 0);
#pragma unroll
  for (int j = 0; j < NST; ++j) stash_step(0, j);
  __syncthreads();
  // Straight-line body, loads pinned ahead of the MFMA group that hides them (see conv_halo_kernel).
  constexpr int SLOT0 = 9 * (KP - 4);        // first MFMA statement (of 9 * KP) that carries a staging step
  static_assert(NST <= 36, "staging steps fit the last four k-pairs");
  for (int c = 0; c < n_chunks; ++c) {
    const int buf = c & 1;
    const int cn = c + 1 < n_chunks ? c + 1 : c;     // last chunk: a redundant, unused refill
    // LDS row r holds input row y0-1+r, column c holds input column x0-1+c:
    // shift (dy,dx) of quad (row, col) -> Xs[.][row + 1 + dy][col + 1 + dx]
    const float* xs = &Xs[buf][frow][wrow0 + lr][lc];
    float b00[TN], b0m[TN], bm0[TN], bmm[TN], n00[TN], n0m[TN], nm0[TN], nmm[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const float* q = xs + b * RPT * XW;
      b00[b] = q[XW + 1]; b0m[b] = q[XW]; bm0[b] = q[1]; bmm[b] = q[0];
    }
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      int nkp = kp + 3, nc = c;
      if (nkp >= KP) { nkp -= KP; nc = cn; }
      if (!RW_ABL(p, 2)) aload(aring[(kp + 3) & 3], nkp, nc);
      // after the weight load (its wait must not drain these), and UNCONDITIONAL -- the last chunk re-fetches
      // itself: with a branch around the fetch the compiler has to count vmcnt for the path without it, and
      // every weight wait of the chunk then drains the fetch on the path with it
      if (kp == 0 && !RW_ABL(p, 4)) xfetch(cn * IC);
      if (kp + 1 < KP && !RW_ABL(p, 1)) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const float* q = xs + (2 * kp + 2) * XH * XW + b * RPT * XW;
          n00[b] = q[XW + 1]; n0m[b] = q[XW]; nm0[b] = q[1]; nmm[b] = q[0];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // nine slabs x TN rows; consecutive MFMAs never accumulate into the same registers
#define RW_UP_MFMA(m, q, sl, bv)                                                                        \
  _Pragma("unroll") for (int b = 0; b < TN; ++b) acc[q][b] =                                            \
      __builtin_amdgcn_mfma_f32_32x32x2f32((sl) < 8 ? aring[kp & 3].v4[((sl) >> 2) & 1][(sl) & 3] : aring[kp & 3].s8, bv[b],  \
                                           acc[q][b], 0, 0, 0);                                        \
  if (9 * kp + (m) >= SLOT0 && 9 * kp + (m) - SLOT0 < NST && !RW_ABL(p, 4)) {                                          \
    stash_step(buf ^ 1, 9 * kp + (m) - SLOT0);                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  }
      RW_UP_MFMA(0, 0, 0, b00)
      RW_UP_MFMA(1, 1, 4, b00)
      RW_UP_MFMA(2, 0, 1, b0m)
      RW_UP_MFMA(3, 2, 6, b00)
      RW_UP_MFMA(4, 0, 2, bm0)
      RW_UP_MFMA(5, 3, 8, b00)
      RW_UP_MFMA(6, 0, 3, bmm)
      RW_UP_MFMA(7, 1, 5, bm0)
      RW_UP_MFMA(8, 2, 7, b0m)
#undef RW_UP_MFMA
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < TN; ++b) { b00[b] = n00[b]; b0m[b] = n0m[b]; bm0[b] = nm0[b]; bmm[b] = nmm[b]; }
    }
    __syncthreads();
  }

  // Epilogue.  A lane holds the 2x2 outputs of quad (yy, xx); neighbouring lanes exchange halves
  // (DPP quad_perm [1,0,3,2]) so that the even lane owns four consecutive floats of output row 2yy
  // and the odd lane four of row 2yy+1: one 16-byte store per lane instead of an 8-byte and two
  // strided 4-byte ones (planes have odd size, so the stores are only 4-byte aligned).
  typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
  typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
  const int oh = 2 * p.h + 1, ow = 2 * p.w + 1;
  const int64_t ohw = (int64_t)oh * ow;
  const int xx = x0 + lc;
  const bool odd_lane = fcol & 1;
  const bool pair_ok = (xx | 1) < p.w;        // both quads of the lane pair are inside the tiled area
  float scale[16];                            // w_scale * demod of this lane's 16 out-channels (LDS table)
#pragma unroll
  for (int r = 0; r < 16; ++r) scale[r] = Sc[wm0 + 4 * frow + (r & 3) + 8 * (r >> 2)];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int yy = y0 + wrow0 + b * RPT + lr;
    if (yy >= p.h) continue;                    // row 2H comes from the strip launch
    if (RW_ABL(p, 16) && acc[0][b][0] != 12345.f) continue;
    // row r of the tile is channel o0 + wm0 + 4 frow + (r&3) + 8(r>>2): one 64-bit address per b, the
    // per-row plane offsets are scalar
    float* yb = p.y + ((int64_t)ib * p.out_ch + o0 + wm0 + 4 * frow) * ohw + (int64_t)(2 * yy) * ow + 2 * xx;
    if (pair_ok) {
      float* yv = odd_lane ? yb + ow - 2 : yb;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float sc = scale[r];
        const float v00 = acc[0][b][r] * sc, v01 = acc[1][b][r] * sc;
        const float v10 = acc[2][b][r] * sc, v11 = acc[3][b][r] * sc;
        const float g0 = __int_as_float(__builtin_amdgcn_update_dpp(
            0, __float_as_int(odd_lane ? v00 : v10), 0xB1, 0xf, 0xf, true));
        const float g1 = __int_as_float(__builtin_amdgcn_update_dpp(
            0, __float_as_int(odd_lane ? v01 : v11), 0xB1, 0xf, 0xf, true));
        f32x4_u v;
        v[0] = odd_lane ? g0 : v00; v[1] = odd_lane ? g1 : v01;
        v[2] = odd_lane ? v10 : g0; v[3] = odd_lane ? v11 : g1;
        *reinterpret_cast<f32x4_u*>(yv + (int64_t)((r & 3) + 8 * (r >> 2)) * ohw) = v;
      }
    } else if (xx < p.w) {                      // odd W: last quad column of the tiled area
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float sc = scale[r];
        float* yo = yb + (int64_t)((r & 3) + 8 * (r >> 2)) * ohw;
        f32x2_u e = {acc[0][b][r] * sc, acc[1][b][r] * sc}, d = {acc[2][b][r] * sc, acc[3][b][r] * sc};
        *reinterpret_cast<f32x2_u*>(yo) = e;
        *reinterpret_cast<f32x2_u*>(yo + ow) = d;
      }
    }
  }
}

static int launch_batch(const ConvProblem* ps, int n, int impl, hipStream_t s);

// part: 0 = everything, 1 = the quad tiles only, 2 = output row 2H / column 2W only (the two parts write
// disjoint elements, so a caller may issue them on two streams)
// ---------------------------------------------------------------------------------------
// The border strips of a stride-2 transposed convolution: output row 2H and output column 2W of the (2H+1) x (2W+1)
// map, which the quad-tile kernels (here and in rw_upwino.hip) leave out.  Both are 1-D transposed convolutions of
// the LAST input row / column with one row / column of the 3 x 3 kernel,
//   even output 2m    = W_a x[m] + W_b x[m-1]      row strip: W[2][0], W[2][2]   column strip: W[0][2], W[2][2]
//   odd  output 2m+1  = W_c x[m]                               W[2][1]                         W[1][2]
// i.e. three (out_ch x in_ch) GEMMs over the strip's samples of ALL images: M = 32 out-channels per workgroup, N = 64
// strip positions (image, m) -- m = 0..W for the row strip (x[W] = 0 closes it with the corner pixel), 0..H-1 for the
// column strip --, K = in_ch in chunks of 16 on v_mfma_f32_16x16x4_f32; a wave owns 16 positions, its even-output tile
// accumulates both taps (x[m-1] is the neighbouring column of the same LDS tile).  The batched im2col launch this
// replaces walked 9 x in_ch gathered columns per position in 64-position tiles that do not span images: 0.96 ms for
// the 65 border pixels of the 16 -> 33 layer at 250 images, beside 2.7 ms for the other 1024.
// Weights: the [slab][in_ch][out_ch] part of rw_pack_conv_weight_f32 mode 1 (slab -> tap: rw_up_tap_order).
// ---------------------------------------------------------------------------------------
struct StripProblem {
  const float* x; const float* wp; float* y; const float* style; const float* demod;
  int batch, in_ch, out_ch, h, w;
  float w_scale;
  int n_row_tiles;                  // workgroups along x that belong to the row strip; the rest: column strip
};
#define ST_KC 16
__global__ void __launch_bounds__(256) up_strip_kernel(const StripProblem p) {
  __shared__ float As[3][ST_KC][32];
  __shared__ float Bs[ST_KC][64 + 4];            // column 0 = the position before the tile's first
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool col = (int)blockIdx.x >= p.n_row_tiles;
  const int n0 = 64 * (col ? blockIdx.x - p.n_row_tiles : blockIdx.x);
  const int o0 = 32 * blockIdx.y;
  const int L = col ? p.h : p.w;                 // samples of the last row / column
  const int P = col ? p.h : p.w + 1;             // positions per image
  const int ntot = p.batch * P;
  const int slab_a = col ? 1 : 2, slab_b = 3, slab_c = col ? 7 : 5;
  const int64_t hw = (int64_t)p.h * p.w, slab = (int64_t)p.in_ch * p.out_ch;
  // B staging: thread -> (channel tid / 16 of the chunk, positions n0 - 1 + (tid % 16) + 16 j, j = 0..4)
  const int bk = tid >> 4, bc = tid & 15;
  int64_t boff[5]; int bimg[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int cidx = bc + 16 * j;                // column of Bs: position n0 - 1 + cidx
    const int n = n0 - 1 + cidx;
    boff[j] = -1; bimg[j] = 0;
    if (cidx < 65 && n >= 0 && n < ntot) {
      const int img = n / P, m = n - img * P;
      if (m < L) {
        bimg[j] = img;
        boff[j] = (int64_t)img * p.in_ch * hw + (col ? (int64_t)m * p.w + (p.w - 1) : (int64_t)(p.h - 1) * p.w + m);
      }
    }
  }
  // A staging: thread -> slab tid / 85.. : 3 x 16 x 32 floats = 1536 = 6 per thread as (row = e / 32, o = e % 32)
  float areg[6], breg[5];
  auto fetch = [&](int c) __attribute__((always_inline)) {
    const int i0 = c * ST_KC;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int f = tid + 256 * e;               // 0 .. 1535
      const int t = f / (ST_KC * 32), r = f - t * (ST_KC * 32), k = r >> 5, o = r & 31;
      const int sl = t == 0 ? slab_a : (t == 1 ? slab_b : slab_c);
      areg[e] = p.wp[sl * slab + (int64_t)(i0 + k) * p.out_ch + o0 + o];
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float v = 0.f;
      if (boff[j] >= 0) {
        v = p.x[boff[j] + (int64_t)(i0 + bk) * hw];
        if (p.style) v *= p.style[(int64_t)bimg[j] * p.in_ch + i0 + bk];
      }
      breg[j] = v;
    }
  };
  auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int f = tid + 256 * e;
      const int t = f / (ST_KC * 32), r = f - t * (ST_KC * 32);
      (&As[0][0][0])[t * (ST_KC * 32) + r] = areg[e];
    }
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (bc + 16 * j < 65) Bs[bk][bc + 16 * j] = breg[j];
  };
  rw_f32x4 accE[2], accO[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) { accE[mb] = rw_f32x4{0.f, 0.f, 0.f, 0.f}; accO[mb] = accE[mb]; }
  const int lk = lane >> 4, ln = lane & 15;
  // this lane's position and whether x[m-1] exists for it (m >= 1: the previous position is the same image's)
  const int n = n0 + 16 * wave + ln;
  const bool n_ok = n < ntot;
  const int img = n_ok ? n / P : 0, m = n_ok ? n - img * P : 0;
  const float prev_mask = (n_ok && m >= 1) ? 1.f : 0.f;
  const int chunks = p.in_ch / ST_KC;
  fetch(0);
  for (int c = 0; c < chunks; ++c) {
    __syncthreads();                             // the previous chunk's reads are done
    stash();
    __syncthreads();
    if (c + 1 < chunks) fetch(c + 1);
#pragma unroll
    for (int ks = 0; ks < ST_KC / 4; ++ks) {
      const int k = 4 * ks + lk;
      const float bcur = Bs[k][16 * wave + ln + 1];
      const float bprev = Bs[k][16 * wave + ln] * prev_mask;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float a0 = As[0][k][16 * mb + ln], a1 = As[1][k][16 * mb + ln], a2 = As[2][k][16 * mb + ln];
        accE[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bcur, accE[mb], 0, 0, 0);
        accE[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bprev, accE[mb], 0, 0, 0);
        accO[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bcur, accO[mb], 0, 0, 0);
      }
    }
  }
  if (!n_ok) return;
  const int oh = 2 * p.h + 1, ow = 2 * p.w + 1;
  const int64_t ohw = (int64_t)oh * ow;
  // even output 2m, odd output 2m + 1 (none behind the last sample: m == L only exists on the row strip, as the corner)
  const int64_t pos_e = col ? (int64_t)(2 * m) * ow + 2 * p.w : (int64_t)(2 * p.h) * ow + 2 * m;
  const int64_t step_o = col ? ow : 1;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = o0 + 16 * mb + 4 * lk + r;
      const float sc = p.demod ? p.demod[(int64_t)img * p.out_ch + o] * p.w_scale : p.w_scale;
      float* yo = p.y + ((int64_t)img * p.out_ch + o) * ohw + pos_e;
      yo[0] = accE[mb][r] * sc;
      if (m < L) yo[step_o] = accO[mb][r] * sc;
    }
}

static int launch_up_strips(const ConvProblem& c, const float* wp_all, hipStream_t s) {
  StripProblem p;
  p.x = c.x; p.wp = wp_all; p.y = c.y; p.style = c.style; p.demod = c.demod;
  p.batch = c.batch; p.in_ch = c.in_ch; p.out_ch = c.out_ch; p.h = c.h; p.w = c.w; p.w_scale = c.w_scale;
  const int64_t nrow = rw_cdiv((int64_t)c.batch * (c.w + 1), 64), ncol = rw_cdiv((int64_t)c.batch * c.h, 64);
  if (nrow + ncol > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  p.n_row_tiles = (int)nrow;
  hipLaunchKernelGGL(up_strip_kernel, dim3((unsigned)(nrow + ncol), (unsigned)(c.out_ch / 32)), dim3(256), 0, s, p);
  return RW_LAUNCH_RESULT();
}

static int launch_up_halo(const ConvProblem* ps, const float* wp_all, int part, hipStream_t s) {
  const ConvProblem& c = ps[0];
  UpProblem u;
  u.x = c.x; u.wfrag = wp_all + (int64_t)9 * c.in_ch * c.out_ch; u.y = c.y; u.style = c.style; u.demod = c.demod;
  u.batch = c.batch; u.in_ch = c.in_ch; u.out_ch = c.out_ch; u.h = c.h; u.w = c.w; u.w_scale = c.w_scale;
  u.abl = rw_abl_env();
  const int tw = c.w > 16 ? 32 : (c.w > 8 ? 16 : 8), rpt = 32 / tw;
  u.tiles_x = (int)rw_cdiv(c.w, tw);
  if (part == 2) {
  } else if (tw == 8) {                            // 128 out-channels x (8 x 8 quads)
    u.tiles_y = (int)rw_cdiv(c.h, 8);
    const int work = c.batch * u.tiles_x * u.tiles_y * (c.out_ch / 128);
    hipLaunchKernelGGL((conv_up_halo_kernel<4, 1, 16, 8>), dim3(work), dim3(256), 0, s, u);
  } else if (c.out_ch % 64 == 0) {
    u.tiles_y = (int)rw_cdiv(c.h, 4 * rpt);
    const int work = c.batch * u.tiles_x * u.tiles_y * (c.out_ch / 64);
    if (tw == 16) hipLaunchKernelGGL((conv_up_halo_kernel<2, 2, 16, 16>), dim3(work), dim3(256), 0, s, u);
    else hipLaunchKernelGGL((conv_up_halo_kernel<2, 2, 16, 32>), dim3(work), dim3(256), 0, s, u);
  } else {
    u.tiles_y = (int)rw_cdiv(c.h, 8 * rpt);
    const int work = c.batch * u.tiles_x * u.tiles_y * (c.out_ch / 32);
    if (tw == 16) hipLaunchKernelGGL((conv_up_halo_kernel<1, 4, 16, 16>), dim3(work), dim3(256), 0, s, u);
    else hipLaunchKernelGGL((conv_up_halo_kernel<1, 4, 16, 32>), dim3(work), dim3(256), 0, s, u);
  }
  if (part == 1) return RW_LAUNCH_RESULT();
  // Output row 2H and column 2W: quads y' = H (phases (0,0),(0,1)) and x' = W, y' < H (phases (0,0),(1,0)): the strip
  // kernel above; RW_UP_STRIPS=im2col keeps the four strip problems of one batched im2col launch (A/B, cross-check)
  const char* strips_env = getenv("RW_UP_STRIPS");                      // read per call: the tests flip it
  const bool strips_gemm = !(strips_env && !strcmp(strips_env, "im2col"));
  if (strips_gemm && c.in_ch % ST_KC == 0 && c.out_ch % 32 == 0) return launch_up_strips(c, wp_all, s);
  ConvProblem e[4] = {ps[0], ps[1], ps[0], ps[2]};
  e[0].ph = 1; e[0].pw = c.w + 1; e[0].yoff = c.h;
  e[1].ph = 1; e[1].pw = c.w;     e[1].yoff = c.h;
  e[2].ph = c.h; e[2].pw = 1;     e[2].xoff = c.w;
  e[3].ph = c.h; e[3].pw = 1;     e[3].xoff = c.w;
  return launch_batch(e, 4, 0, s);
}

static bool halo_applicable(const ConvProblem* ps, int n) {
  // column tiles of 32 pixels for maps at least 24 wide, of 2 x 16 for maps 9..16 wide, of 4 x 8 for
  // maps 5..8 wide (128 out-channel tiles only)
  // (for the four phases of a transposed convolution: by the input width, pw = W or W + 1)
  for (int q = 0; q < n; ++q) {
    const int wref = n == 1 ? ps[q].pw : ps[q].w;
    const bool wide = ps[q].pw >= 24, narrow = wref >= 9 && wref <= 16;
    const bool tiny = wref >= 5 && wref <= 8 && ps[q].out_ch % 128 == 0;
    if (!(wide || narrow || tiny) || ps[q].in_ch % 16 || ps[q].in_ch > 1024 || ps[q].out_ch % 32) return false;
  }
  return true;
}

template <int TW>
static void launch_halo_frag(int bm, int work, const HaloProblem& h, hipStream_t s) {
  if (bm == 128)
    hipLaunchKernelGGL((conv_halo_kernel<2, 2, 2, 2, 16, true, TW>), dim3(work), dim3(256), 0, s, h);
  else if (bm == 64)
    hipLaunchKernelGGL((conv_halo_kernel<2, 2, 1, 4, 16, true, TW>), dim3(work), dim3(256), 0, s, h);
  else
    hipLaunchKernelGGL((conv_halo_kernel<1, 4, 1, 4, 8, true, TW>), dim3(work), dim3(256), 0, s, h);
}

// ps: 1 (stride-1 conv) or 4 (transposed-conv phases) problems sharing x / y / epilogue.
struct RgbFusion { const float* weight; const float* style; const float* bias; const float* skip; float* out; float scale; };

static int launch_halo(const ConvProblem* ps, int n, const float* wfrag, hipStream_t s, const RgbFusion* rgb = nullptr) {
  const ConvProblem& c = ps[0];
  HaloProblem h;
  h.rgb_weight = rgb ? rgb->weight : nullptr; h.rgb_style = rgb ? rgb->style : nullptr;
  h.rgb_bias = rgb ? rgb->bias : nullptr; h.rgb_skip = rgb ? rgb->skip : nullptr; h.rgb_out = rgb ? rgb->out : nullptr;
  h.rgb_scale = rgb ? rgb->scale : 0.f;
  h.x = c.x; h.wp = ps[0].wp; h.wfrag = wfrag; h.y = c.y; h.style = c.style; h.demod = c.demod; h.noise = c.noise;
  h.noise_w = c.noise_w; h.bias = c.bias; h.batch = c.batch; h.in_ch = c.in_ch; h.out_ch = c.out_ch;
  h.h = c.h; h.w = c.w; h.oh = c.oh; h.ow = c.ow; h.sy = c.sy; h.sx = c.sx; h.w_scale = c.w_scale;
  h.act = c.act; h.nphase = n; h.abl = rw_abl_env();
  // 2 (4) image rows per column tile on maps 9..16 (5..8) wide
  const int tw = !wfrag || c.pw > 16 ? 32 : (c.pw > 8 ? 16 : 8);
  int th, bm;
  if (c.out_ch % 128 == 0) { th = 4; bm = 128; }
  else if (c.out_ch % 64 == 0) { th = 8; bm = 64; }
  else { th = 16; bm = 32; }
  th *= 32 / tw;
  if (tw == 8) { th = 8; bm = 128; }                   // one variant: 128 out-channels x (8 rows x 8 columns)
  int work = 0;
  for (int q = 0; q < 4; ++q) {
    PhaseDesc& d = h.phase[q];
    const ConvProblem& pq = ps[q < n ? q : 0];
    d.ntaps = pq.ntaps; d.dy_bits = pq.dy_bits; d.dx_bits = pq.dx_bits; d.ph = pq.ph; d.pw = pq.pw;
    d.oy0 = pq.oy0; d.ox0 = pq.ox0;
    d.tiles_x = (int)rw_cdiv(pq.pw, tw); d.tiles_y = (int)rw_cdiv(pq.ph, th);
    d.work0 = work; d.wp_off = (long long)(pq.wp - ps[0].wp);
    if (q < n) work += c.batch * d.tiles_x * d.tiles_y * (c.out_ch / bm);
  }
  if (work == 0) return 0;
  if (wfrag) {        // stride-1 convolution, weights in fragment order
    if (tw == 8)
      hipLaunchKernelGGL((conv_halo_kernel<2, 1, 2, 2, 16, true, 8>), dim3(work), dim3(256), 0, s, h);
    else if (tw == 16) launch_halo_frag<16>(bm, work, h, s);
    else launch_halo_frag<32>(bm, work, h, s);
  } else {
    if (bm == 128)
      hipLaunchKernelGGL((conv_halo_kernel<2, 2, 2, 2, 16, false, 32>), dim3(work), dim3(256), 0, s, h);
    else if (bm == 64)
      hipLaunchKernelGGL((conv_halo_kernel<2, 2, 1, 4, 16, false, 32>), dim3(work), dim3(256), 0, s, h);
    else
      hipLaunchKernelGGL((conv_halo_kernel<1, 4, 1, 4, 8, false, 32>), dim3(work), dim3(256), 0, s, h);
  }
  return RW_LAUNCH_RESULT();
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
    const int yq = r0 / p.pw;
    const int yy = yq + p.yoff, xx = r0 - yq * p.pw + p.xoff;
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

static int64_t std_blocks(const ConvProblem& p, int bm, int bn) {
  return rw_cdiv((int64_t)p.batch * p.ph * p.pw, bn) * (p.out_ch / bm);
}

// impl: 1 = direct VALU kernel, 5 = im2col MFMA without split-K, otherwise im2col MFMA with the
// split-K variant chosen automatically for launches that would leave most CUs idle.
static int launch_batch(const ConvProblem* ps, int n, int impl, hipStream_t s) {
  if (impl == 1) {
    for (int q = 0; q < n; ++q) {
      const int64_t total = (int64_t)ps[q].batch * ps[q].ph * ps[q].pw * ps[q].out_ch;
      if (total == 0) continue;
      hipLaunchKernelGGL(conv_direct_kernel, dim3(rw_stream_grid(total, 256) * 4), dim3(256), 0, s, ps[q]);
    }
    return RW_LAUNCH_RESULT();
  }
  const ConvProblem& c = ps[0];
  if (c.in_ch % RW_KC || c.out_ch % 32) return RW_ERR_UNSUPPORTED;
  int bm, bn;
  if (c.out_ch % 128 == 0) { bm = 128; bn = 128; }
  else if (c.out_ch % 64 == 0) { bm = 64; bn = 256; }
  else { bm = 32; bn = 256; }
  int64_t blocks = 0, blocks64 = 0;
  for (int q = 0; q < n; ++q) {
    blocks += std_blocks(ps[q], bm, bn);
    if (c.out_ch % 64 == 0) blocks64 += std_blocks(ps[q], 64, 64);
  }
  ConvBatch cb;
  cb.n = n;
  int ksplit = 0;   // 0 = off, 1 = 32x32 tiles (chunks of 64 k), 2 = 64x64 tiles (chunks of 32 k)
  if (impl == 6 || (impl != 5 && blocks < 192)) {      // 6: force split-K (A/B measurements)
    // 64x64 tiles only when they still give two workgroups per CU (one wave per SIMD leaves every LDS read ->
    // MFMA dependency exposed: layer 2 of the 1024 model at batch 64 runs 0.19 ms on 32x32 tiles, 0.31 ms on 64x64)
    if (c.out_ch % 64 == 0 && c.in_ch % 32 == 0 && (blocks64 >= 512 || c.in_ch % 64)) ksplit = 2;
    else if (c.in_ch % 64 == 0) ksplit = 1;
    else if (c.out_ch % 64 == 0 && c.in_ch % 32 == 0) ksplit = 2;
  }
  if (ksplit) {
    const char* ek = getenv("RW_KSPLIT");             // experiment: force the 32x32 (1) or 64x64 (2) tile
    if (ek && atoi(ek) == 1 && c.in_ch % 64 == 0) ksplit = 1;
    if (ek && atoi(ek) == 2 && c.out_ch % 64 == 0 && c.in_ch % 32 == 0) ksplit = 2;
    bm = bn = (ksplit == 2) ? 64 : 32;
  }
  int64_t work = 0;
  for (int q = 0; q < 4; ++q) {
    cb.p[q] = ps[q < n ? q : 0];
    cb.work0[q] = (int)work;
    if (q < n) work += std_blocks(ps[q], bm, bn);
  }
  cb.work0[4] = (int)work;
  if (work == 0) return 0;
  if (work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)work), block(256);
  if (ksplit == 2) hipLaunchKernelGGL((conv_mfma_ksplit_kernel<2, 8>), grid, block, 0, s, cb);
  else if (ksplit == 1) hipLaunchKernelGGL((conv_mfma_ksplit_kernel<1, 16>), grid, block, 0, s, cb);
  else if (bm == 128) hipLaunchKernelGGL((conv_mfma_kernel<2, 2, 2, 2>), grid, block, 0, s, cb);
  else if (bm == 64) hipLaunchKernelGGL((conv_mfma_kernel<2, 2, 1, 4>), grid, block, 0, s, cb);
  else hipLaunchKernelGGL((conv_mfma_kernel<1, 2, 1, 4>), grid, block, 0, s, cb);
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
  p.yoff = 0; p.xoff = 0;
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
  if (impl == 3 && !halo_applicable(&p, 1)) return RW_ERR_UNSUPPORTED;
  if (impl == 3 || (impl == 0 && halo_applicable(&p, 1)))
    return launch_halo(&p, 1, wp + (int64_t)9 * in_ch * out_ch, rw_s(stream));
  return launch_batch(&p, 1, impl == 2 ? 0 : impl, rw_s(stream));
}

extern "C" int rw_conv3x3_to_rgb_f32(const float* x, const float* wp, float* y, int batch, int in_ch, int out_ch,
                                    int h, int w, float w_scale, const rw_conv_epilogue* ep,
                                    const rw_rgb_epilogue* rgb, rw_stream_t stream) {
  RW_CHECK_ARG(x && wp && rgb && rgb->weight && rgb->style && rgb->out && batch > 0 && in_ch > 0 && out_ch > 0);
  RW_CHECK_ARG(h > 0 && w > 0 && (!ep || ((!ep->noise || ep->noise_w) && (!ep->act || ep->bias))));
  // ONE WAVE must hold every out-channel of its pixels (the 32- and 64-channel tile shapes), 32-pixel column tiles
  if (!(out_ch == 32 || out_ch == 64) || w < 24 || in_ch % 16 || in_ch > 1024)
    return RW_ERR_UNSUPPORTED;
  ConvProblem p;
  fill_common(p, x, wp, y, batch, in_ch, out_ch, h, w, w_scale, ep);
  p.ph = h; p.pw = w; p.oh = h; p.ow = w; p.sy = 1; p.sx = 1; p.oy0 = 0; p.ox0 = 0;
  p.ntaps = 9;
  for (int t = 0; t < 9; ++t) set_tap(p, t, t / 3 - 1, t % 3 - 1);
  const RgbFusion f = {rgb->weight, rgb->style, rgb->bias, rgb->skip, rgb->out, rgb->scale};
  return launch_halo(&p, 1, wp + (int64_t)9 * in_ch * out_ch, rw_s(stream), &f);
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
  ConvProblem ps[4];
  for (int phase = 0; phase < 4; ++phase) {
    const int py = phase >> 1, px = phase & 1;
    ConvProblem& p = ps[phase];
    fill_common(p, x, wp + slab0[phase] * slab, y, batch, in_ch, out_ch, h, w, w_scale, ep);
    p.ph = py ? h : h + 1; p.pw = px ? w : w + 1;
    p.oh = 2 * h + 1; p.ow = 2 * w + 1; p.sy = 2; p.sx = 2; p.oy0 = py; p.ox0 = px;
    p.ntaps = ntaps[phase];
    for (int t = 0; t < p.ntaps; ++t) set_tap(p, t, tdy[phase][t], tdx[phase][t]);
  }
  // impl 8 = the border strips alone (output row 2H, column 2W): a batched im2col launch that needs the channel
  // multiples of the halo kernels but no tile shape, so it also serves maps the tiles do not (4 wide, rw_upwino.hip)
  const bool strips_ok = in_ch % 16 == 0 && in_ch <= 1024 && out_ch % 32 == 0;
  if (impl == 8 && !strips_ok) return RW_ERR_UNSUPPORTED;
  if ((impl == 3 || impl == 4 || impl == 7) && !halo_applicable(ps, 4)) return RW_ERR_UNSUPPORTED;
  if (impl == 7 || impl == 8) return launch_up_halo(ps, wp, impl - 6, rw_s(stream));
  if (impl == 4) return launch_halo(ps, 4, nullptr, rw_s(stream));        // per-phase halo tiles (kept for A/B)
  if (impl == 3 || (impl == 0 && halo_applicable(ps, 4))) return launch_up_halo(ps, wp, 0, rw_s(stream));
  return launch_batch(ps, 4, impl == 2 ? 0 : impl, rw_s(stream));
}

// ---------------------------------------------------------------------------------------
// OPT-IN split-precision variant of the stride-1 halo kernel ("bf16x6").  fp32 MFMA runs at the fp32
// vector rate (157 TFLOP/s); the bf16 matrix rate is 16x that.  Every fp32 operand is split EXACTLY into
// three bf16 pieces (8 + 8 + 8 mantissa bits, by truncation: x = x1 + x2 + x3), and a product keeps the six
// piece products down to 2^-16 relative (x1y1, x1y2, x2y1, x2y2, x1y3, x3y1), dropping terms below
// 2^-23 -- fp32-product accuracy, fp32 accumulation in the MFMA, no range loss (bf16 has fp32's exponent).
// Six v_mfma_f32_32x32x16_bf16 (32 cycles each, K = 16 = one whole channel chunk) replace eight
// v_mfma_f32_32x32x2_f32 (64 cycles each): 192 instead of 512 matrix-pipe cycles per tile and tap.
// Operand layout (probed on the device, scripts/probe/mfma_bf16_layout.hip): lane l holds
// A[row l&31][k = 8 (l>>5) + j], B[k = 8 (l>>5) + j][col l&31], j = 0..7.
//   weights: wb[tap][chunk][o/32][piece][lane] 16-byte cells (rw_pack_conv_weight_bf16x3)
//   input:   Xb[buf][piece][k-group][row][col] 16-byte cells in LDS, split while staging
// The default path stays exact fp32 MFMA; this one is selected explicitly (rw_conv3x3_bf16x6_f32).
// ---------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 rw_bf16x8;

__device__ __forceinline__ void rw_split3(float x, unsigned& p1, unsigned& p2, unsigned& p3) {
  const unsigned u = __float_as_uint(x) & 0xffff0000u;
  const float r = x - __uint_as_float(u);                    // exact
  const unsigned v = __float_as_uint(r) & 0xffff0000u;
  const float r2 = r - __uint_as_float(v);                   // exact, <= 8 significant bits left
  p1 = u >> 16; p2 = v >> 16; p3 = __float_as_uint(r2) >> 16;
}

__global__ void __launch_bounds__(256) pack_conv_bf16x3_kernel(const float* __restrict__ w, uint4* __restrict__ wb,
                                                               int out_ch, int in_ch) {
  const int obn = out_ch >> 5, chunks = in_ch >> 4;
  const int64_t total = (int64_t)9 * chunks * obn * 64;       // one thread: the three pieces of one lane cell
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int lane = (int)(r & 63); r >>= 6;
    const int ob = (int)(r % obn); r /= obn;
    const int c = (int)(r % chunks); r /= chunks;
    const int tap = (int)r;
    const int o = 32 * ob + (lane & 31), i0 = 16 * c + 8 * (lane >> 5);
    unsigned pc[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rw_split3(w[((int64_t)o * in_ch + i0 + j) * 9 + tap], pc[0][j], pc[1][j], pc[2][j]);
    uint4* dst = wb + (((int64_t)tap * chunks + c) * obn + ob) * 192 + lane;
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
      dst[s3 * 64] = make_uint4(pc[s3][0] | (pc[s3][1] << 16), pc[s3][2] | (pc[s3][3] << 16),
                                pc[s3][4] | (pc[s3][5] << 16), pc[s3][6] | (pc[s3][7] << 16));
  }
}

template <int TM, int TN, int WGM, int WGN>
__global__ void __launch_bounds__(256, 2) conv_halo_bf16x6_kernel(const HaloProblem p) {
  constexpr int IC = 16;
  constexpr int BM = 32 * TM * WGM;
  constexpr int TH = TN * WGN;
  constexpr int XH = TH + 2, XW = 36, XUSED = 34;
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  __shared__ uint4 Xb[2][3][2][XH][XW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WGN) * 32 * TM;
  const int wrow0 = (wave % WGN) * TN;
  const int frow = lane >> 5, fcol = lane & 31;

  const PhaseDesc d = p.phase[0];
  int local = rw_xcd_remap(blockIdx.x, gridDim.x);
  const int o_tiles = p.out_ch / BM;
  const int o0 = (local % o_tiles) * BM; local /= o_tiles;
  const int tx = local % d.tiles_x; local /= d.tiles_x;
  const int ty = local % d.tiles_y;
  const int ib = local / d.tiles_y;
  const int y0 = ty * TH, x0 = tx * 32;
  const int64_t hw = (int64_t)p.h * p.w;
  const float* xb = p.x + (int64_t)ib * p.in_ch * hw;
  const float* st = p.style ? p.style + (int64_t)ib * p.in_ch : nullptr;

  constexpr int NPOS = XH * XUSED;
  constexpr int PSLOT = (NPOS + 255) / 256;
  int xoff[PSLOT], xlds[PSLOT];
  float xmask[PSLOT];
#pragma unroll
  for (int sl = 0; sl < PSLOT; ++sl) {
    const int pos = tid + 256 * sl;
    const int r = pos / XUSED, c = pos - r * XUSED;
    const int iy = y0 - 1 + r, ix = x0 - 1 + c;
    const bool ok = pos < NPOS && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
    xoff[sl] = ok ? iy * p.w + ix : 0;
    xmask[sl] = ok ? 1.0f : 0.0f;
    xlds[sl] = pos < NPOS ? r * XW + c : XW - 1;
  }
  float xreg[PSLOT][IC];
  float sty[IC];
  auto xfetch = [&](int i0) {
    const float* xc = xb + (int64_t)i0 * hw;
#pragma unroll
    for (int ic = 0; ic < IC; ++ic)
#pragma unroll
      for (int sl = 0; sl < PSLOT; ++sl) xreg[sl][ic] = xc[(int64_t)ic * hw + xoff[sl]];
    if (st) {
#pragma unroll
      for (int ic = 0; ic < IC; ++ic) sty[ic] = st[i0 + ic];
    } else {
#pragma unroll
      for (int ic = 0; ic < IC; ++ic) sty[ic] = 1.0f;
    }
  };
  // style and zero padding in fp32, then the exact three-way split, 8 channels per 16-byte LDS cell;
  // one unit = the 8 channels (sl, g) of one staged position
  constexpr int NU = 2 * PSLOT;
  static_assert(NU <= 8, "staging units fit the taps of a chunk");
  auto stash_unit = [&](int buf, int u) {
    const int sl = u >> 1, g = u & 1;
    unsigned pc[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      rw_split3(xreg[sl][8 * g + j] * (xmask[sl] * sty[8 * g + j]), pc[0][j], pc[1][j], pc[2][j]);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
      (&Xb[buf][s3][g][0][0])[xlds[sl]] =
          make_uint4(pc[s3][0] | (pc[s3][1] << 16), pc[s3][2] | (pc[s3][3] << 16),
                     pc[s3][4] | (pc[s3][5] << 16), pc[s3][6] | (pc[s3][7] << 16));
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NU; ++u) stash_unit(buf, u);
  };

  const int n_chunks = p.in_ch / IC;
  const int obn = p.out_ch >> 5;
  const uint4* wb = reinterpret_cast<const uint4*>(p.wfrag) + (int64_t)((o0 + wm0) >> 5) * 192 + lane;
  const int c_stride = obn * 192;
  const int64_t t_stride = (int64_t)n_chunks * c_stride;
  uint4 acur[TM][3], anxt[TM][3];
  auto aload = [&](uint4 (&dst)[TM][3], int t, int c) {
    const uint4* base = wb + (int64_t)t * t_stride + (int64_t)c * c_stride;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) dst[a][s3] = base[a * 192 + s3 * 64];
  };

  rw_f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  xfetch(0);
  aload(acur, 0, 0);
  stash(0);
  __syncthreads();
  for (int c = 0; c < n_chunks; ++c) {
    const int buf = c & 1;
    const int cn = c + 1 < n_chunks ? c + 1 : c;
    if (!RW_ABL(p, 4)) xfetch(cn * IC);
    // one tap; UNIT >= 0 also converts and stages unit UNIT of the next chunk between the MFMAs
    auto tap = [&](int t, auto unit_tag) {
      constexpr int UNIT = decltype(unit_tag)::value;
      int nt = t + 1, nc = c;
      if (nt == 9) { nt = 0; nc = cn; }
      if (!RW_ABL(p, 2)) aload(anxt, nt, nc);
      const int dy = rw_tap_off(d.dy_bits, t), dx = rw_tap_off(d.dx_bits, t);
      uint4 bq[TN][3];
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3)
          bq[b][s3] = RW_ABL(p, 1) ? acur[0][s3] : Xb[buf][s3][frow][wrow0 + b + dy + 1][fcol + dx + 1];
      __builtin_amdgcn_sched_barrier(0);
#define RW_BF(v) __builtin_bit_cast(rw_bf16x8, v)
      // six piece products (small terms first); consecutive MFMAs go to different accumulators
      constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(RW_BF(acur[a][PA[q]]), RW_BF(bq[b][PB[q]]),
                                                                acc[a][b], 0, 0, 0);
#undef RW_BF
      if (UNIT >= 0 && !RW_ABL(p, 4)) stash_unit(buf ^ 1, UNIT);      // VALU + LDS writes in the shadow of the MFMAs above
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) acur[a][s3] = anxt[a][s3];
    };
#pragma unroll 1
    for (int t = 0; t < 9 - NU; ++t) tap(t, rw_int<-1>());
    if (NU >= 4) { tap(9 - NU, rw_int<(NU >= 4 ? 0 : -1)>()); tap(10 - NU, rw_int<(NU >= 4 ? 1 : -1)>()); }
    tap(7, rw_int<NU - 2>());
    tap(8, rw_int<NU - 1>());
    __syncthreads();
  }
  rw_halo_epilogue<TM, TN, 1>(p, d, acc, ib, o0 + wm0 + 4 * frow, y0 + wrow0, x0 + fcol);
}

extern "C" long long rw_packed_conv_weight_bf16x3_bytes(int out_ch, int in_ch) {
  if (out_ch <= 0 || in_ch <= 0 || out_ch % 32 || in_ch % 16) return -1;
  return (long long)9 * in_ch * out_ch * 6;
}

extern "C" int rw_pack_conv_weight_bf16x3(const float* w, void* wb, int out_ch, int in_ch, rw_stream_t stream) {
  RW_CHECK_ARG(w && wb && out_ch > 0 && in_ch > 0);
  if (out_ch % 32 || in_ch % 16) return RW_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)9 * (in_ch >> 4) * (out_ch >> 5) * 64;
  hipLaunchKernelGGL(pack_conv_bf16x3_kernel, dim3(rw_stream_grid(total, 256)), dim3(256), 0, rw_s(stream), w,
                     reinterpret_cast<uint4*>(wb), out_ch, in_ch);
  return RW_LAUNCH_RESULT();
}

extern "C" int rw_conv3x3_bf16x6_f32(const float* x, const void* wb, float* y, int batch, int in_ch, int out_ch,
                                     int h, int w, float w_scale, const rw_conv_epilogue* ep, rw_stream_t stream) {
  RW_CHECK_ARG(x && wb && y && batch > 0 && in_ch > 0 && out_ch > 0 && h > 0 && w > 0);
  RW_CHECK_ARG(!ep || ((!ep->noise || ep->noise_w) && (!ep->act || ep->bias)));
  if (w < 24 || in_ch % 16 || in_ch > 1024 || out_ch % 64) return RW_ERR_UNSUPPORTED;
  HaloProblem hp;
  hp.x = x; hp.wp = nullptr; hp.wfrag = reinterpret_cast<const float*>(wb); hp.y = y;
  hp.style = ep ? ep->style : nullptr; hp.demod = ep ? ep->demod : nullptr; hp.noise = ep ? ep->noise : nullptr;
  hp.noise_w = ep ? ep->noise_w : nullptr; hp.bias = ep ? ep->bias : nullptr; hp.act = ep ? ep->act : 0;
  hp.batch = batch; hp.in_ch = in_ch; hp.out_ch = out_ch; hp.h = h; hp.w = w; hp.oh = h; hp.ow = w;
  hp.sy = 1; hp.sx = 1; hp.w_scale = w_scale; hp.nphase = 1; hp.abl = rw_abl_env();
  hp.rgb_weight = nullptr; hp.rgb_style = nullptr; hp.rgb_bias = nullptr; hp.rgb_skip = nullptr; hp.rgb_out = nullptr;
  hp.rgb_scale = 0.f;
  const int bm = out_ch % 128 == 0 ? 128 : 64, th = bm == 128 ? 4 : 8;
  PhaseDesc& d = hp.phase[0];
  d.ntaps = 9; d.dy_bits = 0; d.dx_bits = 0;
  for (int t = 0; t < 9; ++t) {
    d.dy_bits |= (unsigned)(t / 3) << (2 * t);          // (dy + 1), (dx + 1)
    d.dx_bits |= (unsigned)(t % 3) << (2 * t);
  }
  d.ph = h; d.pw = w; d.oy0 = 0; d.ox0 = 0;
  d.tiles_x = (int)rw_cdiv(w, 32); d.tiles_y = (int)rw_cdiv(h, th); d.work0 = 0; d.wp_off = 0;
  for (int q = 1; q < 4; ++q) hp.phase[q] = d;
  const int64_t work = (int64_t)batch * d.tiles_x * d.tiles_y * (out_ch / bm);
  if (work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  if (bm == 128)
    hipLaunchKernelGGL((conv_halo_bf16x6_kernel<2, 2, 2, 2>), dim3((unsigned)work), dim3(256), 0, rw_s(stream), hp);
  else
    hipLaunchKernelGGL((conv_halo_bf16x6_kernel<2, 2, 1, 4>), dim3((unsigned)work), dim3(256), 0, rw_s(stream), hp);
  return RW_LAUNCH_RESULT();
}
