// hipcc-flags: -fno-slp-vectorize
// Stride-2 transposed 3x3 modulated convolution (DemodulatedConv2dF with upsample, utils/stylegan2/models.py:313-329:
// F.conv_transpose2d(x, W^T, stride=2) -> (2H+1) x (2W+1)) by the minimal-filtering algorithm F(2,2) in fp32.
//
// Along one axis  y[2i] = w0 x[i] + w2 x[i-1]  (a 2-tap filter over the input) and  y[2i+1] = w1 x[i]  (1 tap).  Two
// consecutive even outputs come from three inputs with THREE multiplications instead of four (F(2,2): points
// d0-d1, d1, d2-d1 against w2, w2+w0, w0; y[2i] = m0+m1, y[2i+2] = m1+m2), the odd ones take two.  In 2-D a block
// of 2x2 quads (4x4 output pixels) from a 3x3 input window costs 9 + 6 + 6 + 4 = 25 multiplications per channel
// pair for its four output-parity phases instead of 36 -- 1.44x fewer matrix FLOPs than rw_conv.hip's
// conv_up_halo_kernel, which already skips the inserted zeros.  The 25 points use only 16 distinct transformed
// inputs (vertical d0-d1, d1, d2-d1, d2 x the same horizontally): 14 subtractions per window and channel.  The
// transforms have coefficients 0, +-1 only: the error class of the direct sum.
//
//   xi =  0.. 8  phase (0,0): rows (d0-d1, d1, d2-d1) x cols (same)      weights (w2, w2+w0, w0) x (same)
//   xi =  9..14  phase (0,1): rows (d0-d1, d1, d2-d1) x cols (d1, d2)    weights (w2, w2+w0, w0) x (w1, w1)
//   xi = 15..20  phase (1,0): rows (d1, d2) x cols (d0-d1, d1, d2-d1)
//   xi = 21..24  phase (1,1): rows (d1, d2) x cols (d1, d2)
//
// Kernel: the design of rw_wino4.hip's conv_wino36b_kernel<2,2> (see there).  The B operand of
// v_mfma_f32_16x16x4_f32 puts element (k, n) in lane 16 k + n; with k = channel of the k-quad and n = block, a
// lane transforms the window of ITS (block, channel) in registers and the results are its B operands -- here at
// 0.9 VALU instructions per MFMA (9 style multiplies + 14 subtractions for 25 MFMAs), which matters because fp32
// MFMA and VALU do not overlap on gfx950.  A wave holds the 25 points of 16 out-channels x 16
// blocks (100 accumulator registers); the output transform is lane-local and a lane ends with a 4x4 pixel block of
// four channels.  Patch (5 rows x 33 columns per channel, `buffer_load_dword ... lds`, out-of-image lanes write 0)
// and weights (`global_load_lds_dwordx4`) arrive by LDS-direct loads issued from inline assembly, the patch two
// 8-channel intervals ahead, the weights one; waits are hand-written (vmcnt(N) + raw s_barrier).
// Workgroup = 4 waves = 32 out-channels x 2 block rows of 16 blocks (4 x 32 quads = 8 x 64 output pixels); it
// walks a run of groups along x.  Quads y < H, x < W only (H % 4 == 0, W % 32 == 0): output row 2H and column 2W
// are the strip problems of rw_conv.hip (rw_conv_transpose3x3s2_f32 impl 8), as for conv_up_halo_kernel.
//   weights: uf[o / 16][i / 4][q = 0..6][lane = 16 (i % 4) + o % 16][xi % 4], xi = 4 q + e (25..27 are zero)
#include "rw_common.h"
//
// H16 (round 4): the 25 GEMMs on the 16-bit matrix pipe with an exact operand split (the scheme of rw_wino4.hip's H16
// kernels: T 2^eV = Th + Tl, U 2^eU = Uh + Ul in f16, all four products accumulated in fp32).  This kernel is bound by
// its fp32 MFMAs (74 % busy, 0.9 VALU per MFMA), so the split pays here: an interval already spans two k-quads, and
// ONE v_mfma_f32_16x16x32_f16 takes both -- lane group lk holds channels (lk, lk + 4) of the interval,
//   A = [Uh0, Uh1, Ul0, Ul1, Uh0, Uh1, Ul0, Ul1]    (two weight words, duplicated)
//   B = [Th0, Th1, Th0, Th1, Tl0, Tl1, Tl0, Tl1]    (v_cvt_pk_f16_f32 converts the pair of channels at once)
// -- 25 MFMAs of ~17 cycles per 8 channels instead of 50 of 32, the 16 distinct transformed inputs split once per
// interval (4 VALU per pair) and kept as operand quads.  |T| <= 4 max |x style| and |U| <= 4 max |w|: eV from the
// caller's x_amax, eU at pack time (trailer, as in rw_wino4.hip).  Packed weights: the 25 points carry 16 distinct values
// (9 of phase (0,0), 3 + 3 of the mixed phases, 1 of phase (1,1)): uf[o / 16][i / 8][wi = 0..15][lane][{0: Uh pair,
// 1: Ul pair}] -- 16 KB per interval and workgroup where the fp32 packing streams 28.
#include <stdlib.h>
typedef float uw_f32x4 __attribute__((ext_vector_type(4)));
typedef float uw_f32x2 __attribute__((ext_vector_type(2)));
typedef int uw_i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned uw_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned uw_u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 uw_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 uw_f16x8 __attribute__((ext_vector_type(8)));

struct UpWinoProblem {
  const float* x; const float* uf; float* y;
  const float* style; const float* demod;
  int batch, in_ch, out_ch, h, w;
  int groups_x, groups_y, gpw;
  float w_scale;
  const float* x_amax;          // H16: the bound of x (RW_BOUND_LANES floats, rw_common.h)
  float u_inv;                  // H16: 1 / (the packed weights' scale), by value
};

#ifndef UW_ABL
#define UW_ABL 0          // timing ablations (results WRONG): 2 = no patch pieces, 4 = no weight pieces, 8 = no epilogue;
                          // H16: 16 = no operand split, 32 = no MFMAs, 64 = no weight reads from LDS
#endif
#ifndef UW_NTS
#define UW_NTS 0          // non-temporal stores of the (2H+1)^2 map (A/B builds)
#endif
#define UW_PITCH 36             // row pitch of a patch channel in LDS: 33 columns + 3

__device__ __forceinline__ int uw_xcd_remap(int id, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, slot = id >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}
__device__ __forceinline__ void uw_dma_buffer_b32(unsigned lds_addr, int voffset, uw_i32x4 rsrc, int soffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}
__device__ __forceinline__ void uw_dma_global_b128_s(unsigned lds_addr, int voffset, const void* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds_addr), "v"(voffset), "s"(sbase)
               : "memory");
}

// NRW = 16: input maps 16 pixels wide (layer 7 of the generators: 16^2 -> 33^2, a fifth of a key-statistics sweep at
// layer 8): a wave's 16 blocks are then TWO block rows of 8 -- block lt sits at block row 2 wn + (lt >> 3), column
// lt & 7 --, the patch of a channel is 9 rows x 17 columns at pitch 20 (the same 180 floats = three pieces as 5 x 33 at
// pitch 36), a workgroup covers 8 quad rows and the map's whole width (groups_x = gpw = 1).
// NRW = 8, 4: the 8^2 and 4^2 maps (layers 5 and 3: a quarter of that sweep on kernels built for big maps).  A wave's
// 16 blocks are then ONE whole 8 x 8 image (4 x 4 blocks) or FOUR 4 x 4 images (2 x 2 blocks each), a workgroup covers
// 2 / 8 consecutive images of the batch: the patch of a channel is 2 x (9 rows x 9 columns at pitch 10) = 180 floats
// resp. 8 x (5 rows x 5 columns at pitch 6, 32 floats apart) = 256 floats = four pieces; the demodulation factors
// are per lane (registers) instead of per workgroup, and the input arrives ALREADY multiplied by its style (the host
// side does that on these tiny maps: a per-image style table for 8 images would not fit beside the rings).
// Everything else is unchanged.
template <int NRW, bool H16 = false>
__device__ __forceinline__ void conv_up_wino_body(const UpWinoProblem& p) {
  static_assert(!H16 || NRW == 0, "the split-operand form is written for the wide maps");
  constexpr int PITCH = NRW == 16 ? 20 : (NRW == 8 ? 10 : (NRW == 4 ? 6 : UW_PITCH));      // row pitch of a patch channel
  constexpr int PROWS = NRW == 16 ? 9 : (NRW == 8 ? 9 : 5);
  constexpr int PCOLS = NRW == 16 ? 17 : (NRW == 8 ? 9 : (NRW == 4 ? 5 : 33));
  constexpr int IPW = NRW == 8 ? 2 : (NRW == 4 ? 8 : 1);        // images per workgroup
  constexpr int ISTRIDE = NRW == 8 ? 90 : (NRW == 4 ? 32 : 256); // floats between the images of a channel's patch
  constexpr int UW_PIECES = NRW == 4 ? 4 : 3;     // 64-float pieces per channel
  constexpr int IC = 8;                           // channels per interval: two k-quads
  constexpr int PSZ = IC * UW_PIECES * 64;        // floats per patch ring slot
  // floats per weight ring slot: [16-channel half][k-quad][7][256]; H16: [half][16 distinct weights][64 lanes][2 words]
  constexpr int USZ = H16 ? 2 * 16 * 128 : 2 * 2 * 7 * 256;
  __shared__ __attribute__((aligned(16))) float Ps[3 * PSZ];
  __shared__ __attribute__((aligned(16))) float Us[2 * USZ];
  __shared__ float St[IPW == 1 ? 512 : 1];        // IPW > 1: the input carries its style already
  __shared__ float Sc[IPW == 1 ? 32 : 1];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;        // out-channel half / block row
  const int lk = lane >> 4, lt = lane & 15;       // channel of the k-quad / block column

  const int local = uw_xcd_remap(blockIdx.x, gridDim.x);
  const int o_tiles = p.out_ch / 32;
  const int runs_x = p.groups_x / p.gpw;
  const int ot = local % o_tiles;
  int pg = local / o_tiles;
  const int run = pg % runs_x; pg /= runs_x;
  const int gy = pg % p.groups_y;
  const int ib = pg / p.groups_y;
  const int o0 = ot * 32;
  const int q0y = (NRW == 16 ? 8 : 4) * gy, gx0 = run * p.gpw;  // first quad row; groups of 32 quad columns
  // this lane's block: image within the workgroup, block row within the workgroup (image), block column within the group
  const int img_l = NRW == 8 ? wn : (NRW == 4 ? 4 * wn + (lt >> 2) : 0);
  const int b_row = NRW == 16 ? 2 * wn + (lt >> 3) : (NRW == 8 ? lt >> 2 : (NRW == 4 ? (lt >> 1) & 1 : wn));
  const int b_col = NRW == 16 ? (lt & 7) : (NRW == 8 ? lt & 3 : (NRW == 4 ? lt & 1 : lt));
  const int64_t hw = (int64_t)p.h * p.w;
  const int img0 = ib * IPW;                       // first image of this workgroup
  const int nimg = min(IPW, p.batch - img0);
  const float* xb = p.x + (int64_t)img0 * p.in_ch * hw;
  const int NC = p.in_ch / IC;
  const int VT = p.gpw * NC;

  if (IPW == 1) {
    float in_scale = 1.f, out_scale = 1.f;       // H16: 2^eV on the input (style table), 2^-(eU + eV) on the result
    if (H16) {
      __shared__ float Red[4];
      float smax = p.style ? 0.f : 1.f;
      if (p.style)
        for (int i = tid; i < p.in_ch; i += 256) smax = fmaxf(smax, fabsf(p.style[(int64_t)ib * p.in_ch + i]));
#pragma unroll
      for (int off = 32; off; off >>= 1) smax = fmaxf(smax, __shfl_xor(smax, off));
      if (lane == 0) Red[wave] = smax;
      __syncthreads();
      smax = fmaxf(fmaxf(Red[0], Red[1]), fmaxf(Red[2], Red[3]));
      const float am = rw_bound_load(p.x_amax) * smax;            // |T| <= 4 am < 2^(e + 2)
      int e = (int)((__float_as_uint(am) >> 23) & 0xff) - 126;    // am < 2^e
      e = e < -100 ? -100 : (e > 100 ? 100 : e);
      in_scale = __uint_as_float((unsigned)(127 + 12 - e) << 23);
      out_scale = __uint_as_float((unsigned)(127 + e - 12) << 23) * p.u_inv;
    }
    for (int i = tid; i < p.in_ch; i += 256) St[i] = (p.style ? p.style[(int64_t)ib * p.in_ch + i] : 1.0f) * in_scale;
    if (tid < 32) Sc[tid] = (p.demod ? p.demod[(int64_t)ib * p.out_ch + o0 + tid] * p.w_scale : p.w_scale) * out_scale;
  }
  const bool img_ok = img_l < nimg;
  float scl[4] = {p.w_scale, p.w_scale, p.w_scale, p.w_scale};        // IPW > 1: this lane's demodulation factors
  if (IPW > 1 && p.demod && img_ok) {
#pragma unroll
    for (int j = 0; j < 4; ++j) scl[j] = p.demod[(int64_t)(img0 + img_l) * p.out_ch + o0 + 16 * wm + 4 * lk + j] * p.w_scale;
  }

  typedef __attribute__((address_space(3))) float* lds_f;
  const unsigned ps_base = (unsigned)(size_t)(lds_f)Ps, us_base = (unsigned)(size_t)(lds_f)Us;
  const unsigned long long xaddr = (unsigned long long)xb;
  const uw_i32x4 xsrc = {(int)(unsigned)xaddr, (int)(unsigned)(xaddr >> 32), (int)((int64_t)nimg * p.in_ch * hw * 4),
                         0x00020000};
  const int hw4 = (int)hw * 4;
  // patch pieces of this wave: channels 2 wave, 2 wave + 1 of the interval, three pieces each.  Patch row r = input
  // row q0y - 1 + r (r = 0..4), column c = input column 32 (gx0 + g) - 1 + c (c = 0..32).
  int xoff[UW_PIECES];
  auto set_group = [&](int g) __attribute__((always_inline)) {
    const int x0 = (gx0 + g) * 32;
#pragma unroll
    for (int s = 0; s < UW_PIECES; ++s) {
      const int f = 64 * s + lane;
      const int m = IPW > 1 ? f / ISTRIDE : 0, fm = f - m * ISTRIDE;
      const int r = fm / PITCH, c = fm - r * PITCH;
      const int iy = q0y - 1 + r, ix = x0 - 1 + c;
      const bool ok = m < nimg && r < PROWS && c < PCOLS && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
      xoff[s] = ok ? (m * p.in_ch * (int)hw + iy * p.w + ix) * 4 : 0x7fffffff;
    }
  };
  int p_soff = 0;
  unsigned p_dst = 0;
  auto pload_begin = [&](int ring, int fg, int fc) __attribute__((always_inline)) {
    if (fc == 0) set_group(fg);
    p_soff = (IC * fc + 2 * wave) * hw4;
    p_dst = ps_base + (unsigned)((ring * PSZ + 2 * wave * (UW_PIECES * 64)) * 4);
  };
  // piece s = 0 .. 2 UW_PIECES - 1: channel 2 wave + s / UW_PIECES, piece s % UW_PIECES
  auto pload_piece = [&](int s) __attribute__((always_inline)) {
    if (UW_ABL & 2) return;
    uw_dma_buffer_b32(p_dst + 256 * s, xoff[s % UW_PIECES], xsrc, p_soff + (s / UW_PIECES) * hw4);
  };
  // weights of an interval: 28 one-KB pieces [half 2][k-quad 2][7]; wave w copies pieces 7 w .. 7 w + 6
  const int kq_total = p.in_ch >> 2;
  const int a_lane = lane * 4;
  auto uload = [&](int slot, int fc) __attribute__((always_inline)) {
    if (UW_ABL & 4) return;
    if (H16) {
      // 16 one-KB pieces [half 2][8]; wave w copies pieces 4 w .. 4 w + 3 (half w / 2)
      const float* src = p.uf + ((int64_t)((o0 >> 4) + (wave >> 1)) * (p.in_ch >> 3) + fc) * (16 * 128) + (wave & 1) * 1024;
      const unsigned dst = us_base + (unsigned)((slot * USZ + wave * 1024) * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) uw_dma_global_b128_s(dst + q * 1024, a_lane * 4, src + q * 256);
      return;
    }
    const int hf = wave >> 1, kql = wave & 1;
    const float* src = p.uf + ((int64_t)((o0 >> 4) + hf) * kq_total + 2 * fc + kql) * (7 * 256);       // uniform
    const unsigned dst = us_base + (unsigned)((slot * USZ + (hf * 2 + kql) * (7 * 256)) * 4);
#pragma unroll
    for (int q = 0; q < 7; ++q) uw_dma_global_b128_s(dst + q * 1024, a_lane * 4, src + q * 256);
  };

  uw_f32x4 acc[25];
#pragma unroll
  for (int xi = 0; xi < 25; ++xi) acc[xi] = uw_f32x4{0.f, 0.f, 0.f, 0.f};

  // this lane's window: patch rows 2 wn .. 2 wn + 2, columns 2 lt .. 2 lt + 2 of channel lk (+ 4 per k-quad)
  const int item_off = lk * (UW_PIECES * 64) + (IPW > 1 ? img_l * ISTRIDE : 0) + (2 * b_row) * PITCH + 2 * b_col;
  auto compute = [&](int ring, int uslot, int c, bool spread) __attribute__((always_inline)) {
#pragma unroll
    for (int kql = 0; kql < 2; ++kql) {
      const float* src = &Ps[ring * PSZ + kql * 4 * (UW_PIECES * 64) + item_off];
      const float sv = IPW == 1 ? St[IC * c + 4 * kql + lk] : 1.0f;
      float d[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const uw_f32x2 lo = *reinterpret_cast<const uw_f32x2*>(src + r * PITCH);
        d[r][0] = lo[0] * sv; d[r][1] = lo[1] * sv; d[r][2] = src[r * PITCH + 2] * sv;
      }
      // T[v][h]: v, h in (d0-d1, d1, d2-d1, d2)
      float t[4][3];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        t[0][cc] = d[0][cc] - d[1][cc]; t[1][cc] = d[1][cc]; t[2][cc] = d[2][cc] - d[1][cc]; t[3][cc] = d[2][cc];
      }
      float T[4][4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        T[v][0] = t[v][0] - t[v][1]; T[v][1] = t[v][1]; T[v][2] = t[v][2] - t[v][1]; T[v][3] = t[v][2];
      }
      const float* ub = &Us[uslot * USZ + (wm * 2 + kql) * (7 * 256) + a_lane];
      uw_f32x4 a4[3];
      a4[0] = *reinterpret_cast<const uw_f32x4*>(ub);
      a4[1] = *reinterpret_cast<const uw_f32x4*>(ub + 256);
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        if (q + 2 < 7) a4[(q + 2) % 3] = *reinterpret_cast<const uw_f32x4*>(ub + (q + 2) * 256);
        if (spread && q < UW_PIECES) pload_piece(UW_PIECES * kql + q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int xi = 4 * q + e;
          if (xi < 25) {
            // the transformed input of point xi (see the table in the header)
            const int vr = xi < 9 ? xi / 3 : (xi < 15 ? (xi - 9) / 2 : (xi < 21 ? 1 + 2 * ((xi - 15) / 3) : 1 + 2 * ((xi - 21) / 2)));
            const int hc = xi < 9 ? xi % 3 : (xi < 15 ? 1 + 2 * ((xi - 9) % 2) : (xi < 21 ? (xi - 15) % 3 : 1 + 2 * ((xi - 21) % 2)));
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q % 3][e], T[vr][hc], acc[xi], 0, 0, 0);
          }
        }
      }
    }
  };

  // H16: both k-quads of the interval in one K = 32 MFMA per point (see the header)
  auto compute16 = [&](int ring, int uslot, int c, bool spread) __attribute__((always_inline)) {
    float T[2][4][4];
#pragma unroll
    for (int kql = 0; kql < 2; ++kql) {
      const float* src = &Ps[ring * PSZ + kql * 4 * (UW_PIECES * 64) + item_off];
      const float sv = St[IC * c + 4 * kql + lk];
      float d[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const uw_f32x2 lo = *reinterpret_cast<const uw_f32x2*>(src + r * PITCH);
        d[r][0] = lo[0] * sv; d[r][1] = lo[1] * sv; d[r][2] = src[r * PITCH + 2] * sv;
      }
      float t[4][3];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        t[0][cc] = d[0][cc] - d[1][cc]; t[1][cc] = d[1][cc]; t[2][cc] = d[2][cc] - d[1][cc]; t[3][cc] = d[2][cc];
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        T[kql][v][0] = t[v][0] - t[v][1]; T[kql][v][1] = t[v][1]; T[kql][v][2] = t[v][2] - t[v][1]; T[kql][v][3] = t[v][2];
      }
    }
    // the 16 operand quads (Th pair, Th pair, Tl pair, Tl pair)
    uw_f16x8 B[4][4];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int hc = 0; hc < 4; ++hc) {
        const float t0 = T[0][v][hc], t1 = T[1][v][hc];
        if (UW_ABL & 16) {
          const uw_f16x2 h0 = __builtin_bit_cast(uw_f16x2, t0), h1 = __builtin_bit_cast(uw_f16x2, t1);
          B[v][hc] = uw_f16x8{h0[0], h0[1], h1[0], h1[1], h0[0], h0[1], h1[0], h1[1]};
        } else {
          const uw_f16x2 hh = __builtin_convertvector(uw_f32x2{t0, t1}, uw_f16x2);
          float r0, r1;
          asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hh), "v"(t0));
          asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hh), "v"(t1));
          const uw_f16x2 ll = __builtin_convertvector(uw_f32x2{r0, r1}, uw_f16x2);
          B[v][hc] = uw_f16x8{hh[0], hh[1], hh[0], hh[1], ll[0], ll[1], ll[0], ll[1]};
        }
      }
    // weights: the 25 points carry 16 distinct values (E x O, O x E and O x O repeat theirs), (Uh pair, Ul pair) of
    // value wi at byte 512 wi + 8 lane (consecutive lanes 8 bytes apart: no bank conflicts -- 16 bytes apart cost
    // this loop half its time in SQ_WAIT_INST_LDS).  The operand wants the two words twice: the vector ALU binds this
    // loop and the LDS is a quarter busy, so ds_read2_b64 reads the same eight bytes into both halves of the operand
    // instead of two v_mov copying them.  The reads are inline assembly (the compiler has no way to say "the same
    // address twice"), four values per batch, one batch ahead.  The compiler takes an asm's outputs for ready: the
    // operands reach the MFMAs only through the asm that waits (scripts/check_asm_loads.py checks the assembly).
    const unsigned ua = us_base + (unsigned)((uslot * USZ + wm * (16 * 128) + 2 * lane) * 4);
    uw_u32x4 aq[2][4];
    auto aload = [&](int wi, uw_u32x4& dst) __attribute__((always_inline)) {
      // offsets of ds_read2_b64 count 8-byte units and stop at 255: one base per four values
      if (UW_ABL & 64) { dst = uw_u32x4{ua, (unsigned)wi, ua, 1u}; return; }
      asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%2" : "=&v"(dst) : "v"(ua + (wi >> 2) * 2048), "n"((wi & 3) * 64));
    };
    auto point = [&](int xi, const uw_u32x4& a) __attribute__((always_inline)) {
      const int vr = xi < 9 ? xi / 3 : (xi < 15 ? (xi - 9) / 2 : (xi < 21 ? 1 + 2 * ((xi - 15) / 3) : 1 + 2 * ((xi - 21) / 2)));
      const int hc = xi < 9 ? xi % 3 : (xi < 15 ? 1 + 2 * ((xi - 9) % 2) : (xi < 21 ? (xi - 15) % 3 : 1 + 2 * ((xi - 21) % 2)));
      if (UW_ABL & 32) asm volatile("" :: "v"(a), "v"(B[vr][hc]));
      else acc[xi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(uw_f16x8, a), B[vr][hc], acc[xi], 0, 0, 0);
    };
#pragma unroll
    for (int e = 0; e < 4; ++e) aload(e, aq[0][e]);
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) {
      uw_u32x4 (&cur)[4] = aq[bt & 1];
      if (bt < 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) aload(4 * (bt + 1) + e, aq[(bt + 1) & 1][e]);
        // this batch's reads are older than the four just issued
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]) :: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]) :: "memory");
      }
      if (spread) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (2 * bt + t < 2 * UW_PIECES) pload_piece(2 * bt + t);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int wi = 4 * bt + e;
        // value wi serves: wi < 9 the point wi of phase (0,0); 9..11 the two columns of row wi - 9 of phase (0,1);
        // 12..14 column wi - 12 of both rows of phase (1,0); 15 all of phase (1,1)
        if (wi < 9) point(wi, cur[e]);
        else if (wi < 12) { point(9 + 2 * (wi - 9), cur[e]); point(10 + 2 * (wi - 9), cur[e]); }
        else if (wi < 15) { point(15 + (wi - 12), cur[e]); point(18 + (wi - 12), cur[e]); }
        else { point(21, cur[e]); point(22, cur[e]); point(23, cur[e]); point(24, cur[e]); }
      }
    }
  };

  // epilogue of one group: acc[xi][j] = M[xi] of out-channel o0 + 16 wm + 4 lk + j, block (row wn, column lt):
  // quads (q0y + 2 wn + a, 32 (gx0 + g) + 2 lt + b), pixels (2 quad + phase)
  typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
  const int oh = 2 * p.h + 1, ow = 2 * p.w + 1;
  const int64_t ohw = (int64_t)oh * ow;
  auto group_epilogue = [&](int g) __attribute__((always_inline)) {
    const int Y0 = 2 * (q0y + 2 * b_row), X0 = 2 * (32 * (gx0 + g) + 2 * b_col);
    float* yb = p.y + ((int64_t)(img0 + img_l) * p.out_ch + o0 + 16 * wm + 4 * lk) * ohw + (int64_t)Y0 * ow + X0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sc = IPW > 1 ? scl[j] : Sc[IPW == 1 ? 16 * wm + 4 * lk + j : 0];
      float px[4][4];                              // [2 a + py][2 b + px]
      // phase (0,0): 3 x 3 -> 2 x 2
      {
        float cs[3][2];
#pragma unroll
        for (int a = 0; a < 3; ++a) { cs[a][0] = acc[3 * a][j] + acc[3 * a + 1][j]; cs[a][1] = acc[3 * a + 1][j] + acc[3 * a + 2][j]; }
#pragma unroll
        for (int b = 0; b < 2; ++b) { px[0][2 * b] = cs[0][b] + cs[1][b]; px[2][2 * b] = cs[1][b] + cs[2][b]; }
      }
      // phase (0,1): 3 x 2 -> 2 x 2
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        px[0][2 * b + 1] = acc[9 + b][j] + acc[11 + b][j];
        px[2][2 * b + 1] = acc[11 + b][j] + acc[13 + b][j];
      }
      // phase (1,0): 2 x 3 -> 2 x 2
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        px[2 * a + 1][0] = acc[15 + 3 * a][j] + acc[16 + 3 * a][j];
        px[2 * a + 1][2] = acc[16 + 3 * a][j] + acc[17 + 3 * a][j];
      }
      // phase (1,1)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) px[2 * a + 1][2 * b + 1] = acc[21 + 2 * a + b][j];
      if (IPW > 1 && !img_ok) continue;              // past the batch: nothing to write
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4_u v = {px[r][0] * sc, px[r][1] * sc, px[r][2] * sc, px[r][3] * sc};
#if UW_NTS
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4_u*>(yb + (int64_t)j * ohw + (int64_t)r * ow));
#else
        *reinterpret_cast<f32x4_u*>(yb + (int64_t)j * ohw + (int64_t)r * ow) = v;
#endif
      }
    }
#pragma unroll
    for (int xi = 0; xi < 25; ++xi) acc[xi] = uw_f32x4{0.f, 0.f, 0.f, 0.f};
  };

#define UW_WAIT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | 0x70)
  // in flight across the barrier of an interval: the 2 UW_PIECES patch pieces issued in it (the weights of v + 1 go
  // first and have landed); + 16 stores after a group's epilogue (a lane past the batch issues none: it then waits for
  // older pieces than it needs to, which is harmless)
  auto sync_interval = [&](bool stores) __attribute__((always_inline)) {
    if (stores) UW_WAIT(2 * UW_PIECES + 16); else UW_WAIT(2 * UW_PIECES);
    __builtin_amdgcn_s_barrier();
  };

  int fg = 0, fc = 0;                               // (group, chunk) of the next interval to fetch
  auto advance = [&]() __attribute__((always_inline)) { if (++fc == NC) { fc = 0; ++fg; } };
  // ---- prologue: tables complete before any LDS-direct load; then U(0), patch 0, patch 1
  __builtin_amdgcn_s_waitcnt(0x0070);
  uload(0, 0);
  pload_begin(0, fg, fc);
#pragma unroll
  for (int s = 0; s < 2 * UW_PIECES; ++s) pload_piece(s);
  advance();
  pload_begin(1, fg, fc);
#pragma unroll
  for (int s = 0; s < 2 * UW_PIECES; ++s) pload_piece(s);
  advance();
  sync_interval(false);                             // U(0), patch 0 landed; patch 1 may be in flight

  int c = 0, g = 0, ring = 0;
  for (int v = 0; v < VT; ++v) {
    const int ring2 = ring == 0 ? 2 : ring - 1;     // (v + 2) % 3
    uload((v + 1) & 1, c + 1 < NC ? c + 1 : 0);     // weights of interval v + 1 (same slices for every group)
    pload_begin(ring2, fg, fc);                     // patch of interval v + 2: pieces issued between the MFMAs
    advance();
    if (H16) compute16(ring, v & 1, c, true); else compute(ring, v & 1, c, true);
    const bool last = c == NC - 1;
    if (last) { if (!(UW_ABL & 8) || acc[0][0] == 12345.f) group_epilogue(g); c = 0; ++g; } else { ++c; }
    ring = ring == 2 ? 0 : ring + 1;
    sync_interval(last);
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
#undef UW_WAIT
}

__global__ void __launch_bounds__(256, 2) conv_up_wino_kernel(const UpWinoProblem p) { conv_up_wino_body<0>(p); }
__global__ void __launch_bounds__(256, 2) conv_up_winoh_kernel(const UpWinoProblem p) { conv_up_wino_body<0, true>(p); }
__global__ void __launch_bounds__(256, 2) conv_up_wino_narrow_kernel(const UpWinoProblem p) { conv_up_wino_body<16>(p); }
__global__ void __launch_bounds__(256, 2) conv_up_wino_8x8_kernel(const UpWinoProblem p) { conv_up_wino_body<8>(p); }
__global__ void __launch_bounds__(256, 2) conv_up_wino_4x4_kernel(const UpWinoProblem p) { conv_up_wino_body<4>(p); }

// the 25 (+3 zero) values of one (o, i): g[3 ky + kx] of W[o][i] as rw_conv_transpose3x3s2_f32 sees it
__device__ __forceinline__ void uw_weight_points(const float* g, float (&u)[28]) {
    // vertical transforms of the three kernel rows: E -> (w[2], w[2] + w[0], w[0]); O -> (w[1], w[1])
    float ve[3][3], vo[2][3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      ve[0][kx] = g[6 + kx]; ve[1][kx] = g[6 + kx] + g[kx]; ve[2][kx] = g[kx];
      vo[0][kx] = g[3 + kx]; vo[1][kx] = g[3 + kx];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      u[3 * a + 0] = ve[a][2]; u[3 * a + 1] = ve[a][2] + ve[a][0]; u[3 * a + 2] = ve[a][0];       // E x E
      u[9 + 2 * a + 0] = ve[a][1]; u[9 + 2 * a + 1] = ve[a][1];                                   // E x O
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      u[15 + 3 * a + 0] = vo[a][2]; u[15 + 3 * a + 1] = vo[a][2] + vo[a][0]; u[15 + 3 * a + 2] = vo[a][0];   // O x E
      u[21 + 2 * a + 0] = vo[a][1]; u[21 + 2 * a + 1] = vo[a][1];                                  // O x O
    }
    u[25] = u[26] = u[27] = 0.f;
}

// One thread: the values of one (o, i).  W[o][i][ky][kx] as rw_conv_transpose3x3s2_f32 sees it.
__global__ void __launch_bounds__(256) pack_up_wino_kernel(const float* __restrict__ w, float* __restrict__ uf,
                                                           int out_ch, int in_ch) {
  const int64_t total = (int64_t)out_ch * in_ch;
  const int kqn = in_ch >> 2;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    int64_t rest = idx >> 6;
    const int kq = (int)(rest % kqn);
    const int ob = (int)(rest / kqn);
    const int o = 16 * ob + (lane & 15), i = 4 * kq + (lane >> 4);
    float u[28];
    uw_weight_points(w + ((int64_t)o * in_ch + i) * 9, u);
    float* dst = uf + ((int64_t)ob * kqn + kq) * (7 * 256) + lane * 4;
#pragma unroll
    for (int q = 0; q < 7; ++q)
      *reinterpret_cast<uw_f32x4*>(dst + q * 256) = uw_f32x4{u[4 * q], u[4 * q + 1], u[4 * q + 2], u[4 * q + 3]};
  }
}

// H16 packing.  PASS 1 (rw_conv_transpose_weight_winoh_absmax_f32): max |U| as a bound, one slot per workgroup.  PASS 2:
// one thread per (o, channel pair (c, c + 4) of an 8-channel interval): the words Uh pair / Ul pair of U su, su BY VALUE
// (rw_split_weight_scale of the maximum the host read back: see rw_wino4.hip); the trailer is for inspection only.
__device__ __forceinline__ unsigned uw_f16_bits(float v) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v); }
template <int PASS>
__global__ void __launch_bounds__(256) pack_up_winoh_kernel(const float* __restrict__ w, float* __restrict__ uf,
                                                            int out_ch, int in_ch, float su, float* __restrict__ bound) {
  const int ivn = in_ch >> 3;
  const int64_t total = (int64_t)out_ch * (in_ch >> 1);
  float m = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    int64_t rest = idx >> 6;
    const int iv = (int)(rest % ivn);
    const int ob = (int)(rest / ivn);
    const int o = 16 * ob + (lane & 15), c0 = 8 * iv + (lane >> 4), c1 = c0 + 4;
    float u0[28], u1[28];
    uw_weight_points(w + ((int64_t)o * in_ch + c0) * 9, u0);
    uw_weight_points(w + ((int64_t)o * in_ch + c1) * 9, u1);
    if (PASS == 1) {
#pragma unroll
      for (int i = 0; i < 25; ++i) m = fmaxf(m, fmaxf(fabsf(u0[i]), fabsf(u1[i])));
    } else {
      unsigned* dst = reinterpret_cast<unsigned*>(uf) + ((int64_t)ob * ivn + iv) * (16 * 128) + lane * 2;
#pragma unroll
      for (int wi = 0; wi < 16; ++wi) {
        const int xi = wi < 9 ? wi : (wi < 12 ? 9 + 2 * (wi - 9) : (wi < 15 ? 15 + (wi - 12) : 21));    // a point that carries it
        const float a = u0[xi] * su, b = u1[xi] * su;
        const _Float16 ah = (_Float16)a, bh = (_Float16)b;
        *reinterpret_cast<uw_u32x2*>(dst + wi * 128) =
            uw_u32x2{uw_f16_bits(a) | (uw_f16_bits(b) << 16), uw_f16_bits(a - (float)ah) | (uw_f16_bits(b - (float)bh) << 16)};
      }
    }
  }
  __shared__ float red[4];
  if (PASS == 1) rw_bound_store_block_256(bound, m, red);
  if (PASS == 2 && blockIdx.x == 0 && threadIdx.x < 4)
    uf[(int64_t)16 * out_ch * in_ch + threadIdx.x] = threadIdx.x == 0 ? 1.f / su : (threadIdx.x == 1 ? su : 0.f);
}

static bool up_wino_narrow(int h, int w) { return w == 16 && h % 8 == 0; }
static bool up_wino_whole(int h, int w) { return (w == 8 && h == 8) || (w == 4 && h == 4); }    // whole images per wave

static bool up_wino_shape_ok(int out_ch, int in_ch, int h, int w) {
  if (!(out_ch > 0 && in_ch >= 16 && in_ch <= 512 && out_ch % 32 == 0 && in_ch % 8 == 0)) return false;
  return (w % 32 == 0 && h % 4 == 0) || up_wino_narrow(h, w) || up_wino_whole(h, w);
}

extern "C" int rw_conv_transpose3x3s2_wino_supported(int out_ch, int in_ch, int h, int w) {
  return up_wino_shape_ok(out_ch, in_ch, h, w) ? 1 : 0;
}

extern "C" long long rw_packed_conv_transpose_wino_elems(int out_ch, int in_ch) {
  if (out_ch <= 0 || in_ch <= 0 || out_ch % 16 || in_ch % 4) return -1;
  return 28LL * out_ch * in_ch;
}

extern "C" int rw_pack_conv_transpose_wino_f32(const float* w, float* uf, int out_ch, int in_ch, rw_stream_t stream) {
  RW_CHECK_ARG(w && uf && out_ch > 0 && in_ch > 0);
  if (out_ch % 16 || in_ch % 4) return RW_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)out_ch * in_ch;
  hipLaunchKernelGGL(pack_up_wino_kernel, dim3(rw_stream_grid(total, 256)), dim3(256), 0, rw_s(stream), w, uf, out_ch,
                     in_ch);
  return RW_LAUNCH_RESULT();
}

// The quads y < H, x < W of the transposed convolution (everything but output row 2H and column 2W, which
// rw_conv_transpose3x3s2_f32 impl 8 writes): y (B, out_ch, 2H+1, 2W+1).
static int up_wino_launch(const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch, int h, int w,
                          float w_scale, const float* style, const float* demod, bool h16, float u_inv,
                          const float* x_amax, rw_stream_t stream) {
  RW_CHECK_ARG(x && uf && y && batch > 0 && in_ch > 0 && out_ch > 0 && h > 0 && w > 0);
  RW_CHECK_ARG(!h16 || (x_amax && u_inv > 0.f));
  if (!up_wino_shape_ok(out_ch, in_ch, h, w)) return RW_ERR_UNSUPPORTED;
  const bool whole = up_wino_whole(h, w);
  if (whole && style) return RW_ERR_UNSUPPORTED;       // 8^2 and 4^2 maps arrive already multiplied by their style
  UpWinoProblem p;
  p.x = x; p.uf = uf; p.y = y; p.style = style; p.demod = demod;
  p.batch = batch; p.in_ch = in_ch; p.out_ch = out_ch; p.h = h; p.w = w; p.w_scale = w_scale;
  p.x_amax = x_amax; p.u_inv = u_inv;
  const bool narrow = up_wino_narrow(h, w);
  if (h16 && (whole || narrow || w % 32 != 0)) return RW_ERR_UNSUPPORTED;
  p.groups_x = (narrow || whole) ? 1 : w / 32;
  p.groups_y = whole ? 1 : (narrow ? h / 8 : h / 4);
  const int ipw = whole ? (w == 8 ? 2 : 8) : 1;        // images per workgroup
  const int wg_batch = (batch + ipw - 1) / ipw;
  const int o_tiles = out_ch / 32;
  const char* e = getenv("RW_UPWINO_GPW");
  int gpw = e ? atoi(e) : 4;
  if (gpw < 1) gpw = 1;
  if (gpw > p.groups_x) gpw = p.groups_x;
  while (p.groups_x % gpw) --gpw;
  while (gpw > 1 && (int64_t)wg_batch * p.groups_y * (p.groups_x / gpw) * o_tiles < 1024) {
    --gpw;
    while (p.groups_x % gpw) --gpw;
  }
  p.gpw = gpw;
  const int64_t work = (int64_t)wg_batch * p.groups_y * (p.groups_x / gpw) * o_tiles;
  if (work <= 0 || work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  if (h16) hipLaunchKernelGGL(conv_up_winoh_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  else if (whole && w == 8) hipLaunchKernelGGL(conv_up_wino_8x8_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  else if (whole) hipLaunchKernelGGL(conv_up_wino_4x4_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  else if (narrow) hipLaunchKernelGGL(conv_up_wino_narrow_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  else hipLaunchKernelGGL(conv_up_wino_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  return RW_LAUNCH_RESULT();
}

extern "C" int rw_conv_transpose3x3s2_wino_f32(const float* x, const float* uf, float* y, int batch, int in_ch,
                                               int out_ch, int h, int w, float w_scale, const float* style,
                                               const float* demod, rw_stream_t stream) {
  return up_wino_launch(x, uf, y, batch, in_ch, out_ch, h, w, w_scale, style, demod, false, 1.f, nullptr, stream);
}

// ---- H16: the same quads with the products on the 16-bit matrix pipe (maps with w % 32 == 0, h % 4 == 0)
extern "C" int rw_conv_transpose3x3s2_winoh_supported(int out_ch, int in_ch, int h, int w) {
  return up_wino_shape_ok(out_ch, in_ch, h, w) && w % 32 == 0 && h % 4 == 0 ? 1 : 0;
}

extern "C" long long rw_packed_conv_transpose_winoh_elems(int out_ch, int in_ch) {
  if (out_ch <= 0 || in_ch <= 0 || out_ch % 16 || in_ch % 8) return -1;
  return 16LL * out_ch * in_ch + 4;
}

extern "C" int rw_conv_transpose_weight_winoh_absmax_f32(const float* w, int out_ch, int in_ch, float* bound,
                                                         rw_stream_t stream) {
  RW_CHECK_ARG(w && bound && out_ch > 0 && in_ch > 0);
  if (out_ch % 16 || in_ch % 8) return RW_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)out_ch * (in_ch >> 1);
  const int grid = rw_stream_grid(total, 256);
  hipLaunchKernelGGL(pack_up_winoh_kernel<1>, dim3(grid), dim3(256), 0, rw_s(stream), w, (float*)nullptr, out_ch, in_ch, 1.f,
                     bound);
  const int rc = RW_LAUNCH_RESULT();
  if (rc) return rc;
  return rw_bound_finish(bound, grid, rw_s(stream));
}

extern "C" int rw_pack_conv_transpose_winoh_f32(const float* w, float* uf, int out_ch, int in_ch, float u_scale,
                                                rw_stream_t stream) {
  RW_CHECK_ARG(w && uf && out_ch > 0 && in_ch > 0 && u_scale > 0.f);
  if (out_ch % 16 || in_ch % 8) return RW_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)out_ch * (in_ch >> 1);
  hipLaunchKernelGGL(pack_up_winoh_kernel<2>, dim3(rw_stream_grid(total, 256)), dim3(256), 0, rw_s(stream), w, uf,
                     out_ch, in_ch, u_scale, (float*)nullptr);
  return RW_LAUNCH_RESULT();
}

extern "C" int rw_conv_transpose3x3s2_winoh_f32(const float* x, const float* uf, float* y, int batch, int in_ch,
                                                int out_ch, int h, int w, float w_scale, const float* style,
                                                const float* demod, float u_inv, const float* x_amax,
                                                rw_stream_t stream) {
  return up_wino_launch(x, uf, y, batch, in_ch, out_ch, h, w, w_scale, style, demod, true, u_inv, x_amax, stream);
}
