// Upsampling StyledConv in one pass WITHOUT the fourfold multiply count of the phase kernels (round 5):
// conv_transpose2d(stride 2) as a direct sum on the 16-bit matrix pipe, the 4x4 blur from LDS, noise + bias + leaky ReLU +
// the next layer's style in the same epilogue  (utils/stylegan2/models.py:313-316 F.conv_transpose2d(stride=2), :275-281
// Blur(pad 1,1), :539-546 NoiseInjection, :232-257 FusedLeakyReLU).
//
// Why another form.  The one-pass kernels of rw_wino4.hip / rw_dconv.hip compose the blur INTO the weights: four
// output-parity phases, each a dense 3x3 convolution -- 36 taps per 2x2 block of outputs where the transposed convolution
// itself has 9 (layer 17 of the 1024 generator: 38.6 GFLOP per image instead of 9.66; 10.2 ms per launch at batch 64, bound
// by MFMA issue).  The two-pass route (F(2,2) transposed convolution -> (2H+1)^2 map in HBM -> blur pass) multiplies less
// but writes and re-reads the largest map of the layer.  Here the transposed convolution is computed as it is defined,
//     z[o][2i + ky][2j + kx] += W[o][c][ky][kx] x[c][i][j]        (9 multiplies per input position and channel pair),
// its result stays in LDS, and the blur reads it there:
//     y[Y][X] = sum_{a,b} kf[a][b] z[Y - 1 + a][X - 1 + b]        (kf = the FIR as upfirdn2d applies it: flipped).
//
// Decomposition.  A POSITION (i, j) of the input grid owns the four z values (2i + py, 2j + px), py, px in {0, 1}:
//     z[2i + py][2j + px] = sum_{a,b in {0,1}, py + 2a <= 2, px + 2b <= 2}  W[py + 2a][px + 2b] . x[i - a][j - b]
// -- phase (0,0) has four taps, (0,1) and (1,0) two, (1,1) one: nine in all, on FOUR pixel operands x[i-a][j-b].  Positions
// are the M side of v_mfma_f32_16x16x32_f16 (16 consecutive positions of the flattened (TY + 2) x (TX + 2) position window:
// a shift in the flattened index is the same for every lane, so the four pixel operands are four LDS reads at fixed
// offsets), 16 out-channels the N side, a 16-channel chunk the K side with the exact f16 operand split of rw_dconv.hip
// ([Vh c0..3 | Vl c0..3] pixel words against [Uh | Uh] and [Ul | Ul]: 18 MFMAs per block and chunk; the packed weights ARE
// rw_pack_dconv_weight_f32's -- the plain 3x3 kernel, no composition).  The blur needs z one row above and two below its
// output rows (one column left, two right): a workgroup computes the positions (TY + 2) x (TX + 2) around its TY x TX
// tile (halo factor 1.2 at 16 x 32) from an x window of (TY + 3) x (TX + 3) pixels staged as in rw_dconv.hip.
//
// Workgroup = 8 waves, ONE per CU (LDS: two 42.5 KB window buffers + 18 KB of weights): wave v owns the position blocks
// v, v + 8, ... (5 of 39) x 4 phases = 80 accumulator registers.  Per chunk: the chunk's 9 KB of weights and the next
// window are requested a chunk ahead (weights through LDS: 512 threads x 16 bytes + a tail), the window converted and
// written behind the first / last blocks' MFMAs, one barrier per chunk.  Epilogue in two halves of 8 channels (the z tile
// of 8 channels, 36 x 72 floats each, reuses the window buffers): accumulators -> LDS, barrier, every thread blurs eight
// groups of four outputs (16-byte LDS reads), applies demodulation, noise, bias, leaky ReLU and the post scale, tracks
// max |y| for the bound, and stores 16 bytes.
#include "rw_common.h"
#include <stdlib.h>
typedef float tc_f32x4 __attribute__((ext_vector_type(4)));
typedef float tc_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 tc_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 tc_f16x8 __attribute__((ext_vector_type(8)));

struct TconvProblem {
  const float* x; const unsigned char* wp; float* y; const float* k4;
  const float* style; const float* demod; const float* noise; const float* noise_w; const float* bias; int act;
  const float* post;
  int batch, in_ch, out_ch, h, w;
  int tiles_x, tiles_y, o_tiles;
  float w_scale, u_inv;
  const float* x_amax; float* y_amax;
};

#define TC_TY 16
#define TC_TX 32
#define TC_PR (TC_TY + 2)                 // position rows
#define TC_PC (TC_TX + 2)                 // position columns
#define TC_NPOS (TC_PR * TC_PC)           // 612
#define TC_NBLK ((TC_NPOS + 15) / 16)     // 39 blocks of 16 positions
#define TC_WR (TC_TY + 3)                 // window rows: input rows I0 - 2 .. I0 + TY
#define TC_WC (TC_TX + 3)                 // window columns: J0 - 2 .. J0 + TX
#define TC_NPIX (TC_WR * TC_WC)           // 665
#define TC_BUFB (TC_NPIX * 64)            // bytes of a window buffer: 16 channels x (2 + 2) bytes per pixel
#define TC_WAVES 8
#define TC_BPW ((TC_NBLK + TC_WAVES - 1) / TC_WAVES)      // 5 blocks per wave
#define TC_ZP 72                          // z row pitch (floats): columns 3 .. 70 are written, 4 .. 70 read
#define TC_ZR (2 * TC_PR)                 // z rows of the tile: 36
#define TC_CHS (TC_ZR * TC_ZP + 4)        // z channel stride (floats)
#define TC_WCH 9216                       // bytes of a chunk's weights of one 16-channel block: 9 taps x [Uh | Ul] x 512

static_assert(8 * TC_CHS * 4 <= 2 * TC_BUFB, "the z tile of eight channels fits the window buffers");

#ifndef TC_ABL
#define TC_ABL 0          // timing ablations (results WRONG): 1 = no staging loads, 2 = no MFMAs, 8 = no epilogue
#endif

__device__ __forceinline__ int tc_xcd_remap(int id, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, slot = id >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}
__device__ __forceinline__ int tc_swz(int cc) { return (cc >> 1) & 3; }       // rw_dconv.hip's dc_swz
__device__ __forceinline__ tc_f16x8 tc_expand(tc_f32x2 w) {
  const tc_f32x4 d = {w[0], w[1], w[0], w[1]};
  return __builtin_bit_cast(tc_f16x8, d);
}
template <int N> struct tc_int { static constexpr int value = N; };

__global__ void __launch_bounds__(512, 1) tconv_blur_kernel(const TconvProblem p) {
  __shared__ __attribute__((aligned(16))) unsigned char Ls[2 * TC_BUFB];
  __shared__ __attribute__((aligned(16))) unsigned char Wl[2 * TC_WCH];
  __shared__ __attribute__((aligned(16))) float St[512];
  __shared__ float Sc[16], Bs[16], Po[16], Kf[16], Red[TC_WAVES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4, lt = lane & 15;

  const int local = tc_xcd_remap(blockIdx.x, gridDim.x);
  const int ot = local % p.o_tiles;
  int pg = local / p.o_tiles;
  const int tx = pg % p.tiles_x; pg /= p.tiles_x;
  const int ty = pg % p.tiles_y;
  const int ib = pg / p.tiles_y;
  const int I0 = ty * TC_TY, J0 = tx * TC_TX;
  const int64_t hw = (int64_t)p.h * p.w;
  const int NC = p.in_ch >> 4, T = 9 * NC;

  // ---- scales (rw_dconv.hip): |x style| <= am < 2^e  ->  |V| = |x style 2^(14 - e)| < 2^14
  float in_scale, out_scale;
  {
    float smax = p.style ? 0.f : 1.f;
    if (p.style)
      for (int i = tid; i < p.in_ch; i += 512) smax = fmaxf(smax, fabsf(p.style[(int64_t)ib * p.in_ch + i]));
    smax = rw_wave_max(smax);
    if (lane == 0) Red[wave] = smax;
    __syncthreads();
    smax = Red[0];
#pragma unroll
    for (int v = 1; v < TC_WAVES; ++v) smax = fmaxf(smax, Red[v]);
    const float am = rw_bound_load(p.x_amax) * smax;
    int e = (int)((__float_as_uint(am) >> 23) & 0xff) - 126;      // am < 2^e
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    in_scale = __uint_as_float((unsigned)(127 + 14 - e) << 23);
    out_scale = __uint_as_float((unsigned)(127 + e - 14) << 23) * p.u_inv;
  }
  const float gain = p.act ? 1.4142135623730951f : 1.f, slope = p.act ? 0.2f : 1.f;
  for (int i = tid; i < p.in_ch; i += 512) St[i] = (p.style ? p.style[(int64_t)ib * p.in_ch + i] : 1.0f) * in_scale;
  if (tid < 16) {
    const int o = 16 * ot + tid;
    Sc[tid] = (p.demod ? p.demod[(int64_t)ib * p.out_ch + o] * p.w_scale : p.w_scale) * out_scale * gain;
    Bs[tid] = p.act ? p.bias[o] * gain : 0.f;
    Po[tid] = p.post ? p.post[(int64_t)ib * p.out_ch + o] : 1.f;
    const int a = tid >> 2, c = tid & 3;
    Kf[tid] = p.k4[(3 - a) * 4 + (3 - c)];        // flipped, as upfirdn2d applies it
  }
  const float noise_wg = p.noise ? p.noise_w[0] * gain : 0.f;
  __syncthreads();                                  // the tables are read by other threads than their writers

  // ---- staging: wave v stages channel quad v & 3 of pixels 64 (v >> 2) + lane + 128 s of the flattened window
  const int g = wave & 3, hsel = wave >> 2;
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + (int64_t)ib * p.in_ch * hw), 0, (int)((int64_t)p.in_ch * hw * 4), 0x00020000);
  const int hw4 = (int)hw * 4;
  constexpr int SI = (TC_NPIX + 127) / 128, SH = (SI + 1) / 2;
  int xoff[SI], loff[SI];
#pragma unroll
  for (int s = 0; s < SI; ++s) {
    const int pi = 128 * s + 64 * hsel + lane;
    const int r = pi / TC_WC, cc = pi - r * TC_WC;
    const int iy = I0 - 2 + r, ix = J0 - 2 + cc;
    const bool ok = pi < TC_NPIX && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
    xoff[s] = ok ? (iy * p.w + ix) * 4 : 0x7fffffff;
    loff[s] = pi < TC_NPIX ? pi * 64 + ((g ^ tc_swz(cc)) << 4) : -1;
  }
  float raw[SH][4];
  auto stage_load = [&](int c, auto half_tag) __attribute__((always_inline)) {
    constexpr int S0 = decltype(half_tag)::value ? SH : 0, S1 = decltype(half_tag)::value ? SI : SH;
    const int s0 = (16 * c + 4 * g) * hw4;
#pragma unroll
    for (int s = S0; s < S1; ++s)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        raw[s - S0][k] = (TC_ABL & 1) ? 1.f : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xsrc, xoff[s], s0 + k * hw4, 0));
  };
  auto stage_store = [&](int c, int buf, auto half_tag) __attribute__((always_inline)) {
    constexpr int S0 = decltype(half_tag)::value ? SH : 0, S1 = decltype(half_tag)::value ? SI : SH;
    const tc_f32x4 sv = *reinterpret_cast<const tc_f32x4*>(&St[16 * c + 4 * g]);
    unsigned char* dst = Ls + buf * TC_BUFB;
#pragma unroll
    for (int s = S0; s < S1; ++s) {
      const float (&rw)[4] = raw[s - S0];
      const float v0 = rw[0] * sv[0], v1 = rw[1] * sv[1], v2 = rw[2] * sv[2], v3 = rw[3] * sv[3];
      const tc_f16x2 h01 = __builtin_convertvector(tc_f32x2{v0, v1}, tc_f16x2);
      const tc_f16x2 h23 = __builtin_convertvector(tc_f32x2{v2, v3}, tc_f16x2);
      float r0, r1, r2, r3;                        // v - (float)h, exact
      asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h01), "v"(v0));
      asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h01), "v"(v1));
      asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h23), "v"(v2));
      asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h23), "v"(v3));
      const tc_f16x2 l01 = __builtin_convertvector(tc_f32x2{r0, r1}, tc_f16x2);
      const tc_f16x2 l23 = __builtin_convertvector(tc_f32x2{r2, r3}, tc_f16x2);
      const tc_f16x8 word = {h01[0], h01[1], h23[0], h23[1], l01[0], l01[1], l23[0], l23[1]};
      if (loff[s] >= 0) *reinterpret_cast<tc_f16x8*>(dst + loff[s]) = word;
    }
  };
  // the chunk's weights of this workgroup's 16 out-channels: 9216 contiguous bytes of the packed array -> LDS
  const unsigned char* wsrc = p.wp + (int64_t)ot * T * 1024;
  tc_f32x4 wraw0, wraw1;
  auto wstage_load = [&](int c) __attribute__((always_inline)) {
    const unsigned char* src = wsrc + (int64_t)c * TC_WCH;
    wraw0 = *reinterpret_cast<const tc_f32x4*>(src + tid * 16);
    if (tid < (TC_WCH - 512 * 16) / 16) wraw1 = *reinterpret_cast<const tc_f32x4*>(src + 512 * 16 + tid * 16);
  };
  auto wstage_store = [&](int buf) __attribute__((always_inline)) {
    unsigned char* dst = Wl + buf * TC_WCH;
    *reinterpret_cast<tc_f32x4*>(dst + tid * 16) = wraw0;
    if (tid < (TC_WCH - 512 * 16) / 16) *reinterpret_cast<tc_f32x4*>(dst + 512 * 16 + tid * 16) = wraw1;
  };

  // ---- this wave's position blocks: operand addresses of the lane's position q = 16 blk + lt (clamped), pixel offsets
  // (a, b) = x[i - a][j - b] at window pixel (r + 1 - a, c + 1 - b)
  unsigned pb0[TC_BPW], pb1[TC_BPW];               // column offset b = 0 / 1 at row offset a = 0 (a = 1: - TC_WC * 64)
#pragma unroll
  for (int b = 0; b < TC_BPW; ++b) {
    int q = 16 * (wave + TC_WAVES * b) + lt;
    q = q < TC_NPOS ? q : TC_NPOS - 1;
    const int r = q / TC_PC, c = q - r * TC_PC;
    pb0[b] = (unsigned)(((r + 1) * TC_WC + c + 1) * 64 + ((lk ^ tc_swz(c + 1)) << 4));
    pb1[b] = (unsigned)(((r + 1) * TC_WC + c) * 64 + ((lk ^ tc_swz(c)) << 4));
  }

  tc_f32x4 acc[TC_BPW][4];
#pragma unroll
  for (int b = 0; b < TC_BPW; ++b)
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) acc[b][ph] = tc_f32x4{0.f, 0.f, 0.f, 0.f};

  tc_f16x8 W[9][2];                                // [tap 3 ky + kx][Uh | Ul], each doubled into the operand
  auto wread = [&](int buf) __attribute__((always_inline)) {
    const unsigned char* wb = Wl + buf * TC_WCH + lane * 8;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int part = 0; part < 2; ++part)
        W[t][part] = tc_expand(*reinterpret_cast<const tc_f32x2*>(wb + t * 1024 + part * 512));
  };
  auto chunk = [&](int c, auto last_tag) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value != 0;
    const int buf = c & 1;
    const unsigned char* lb = Ls + buf * TC_BUFB;
    if (!LAST) { stage_load(c + 1, tc_int<0>()); wstage_load(c + 1); }
    wread(buf);
#pragma unroll
    for (int b = 0; b < TC_BPW; ++b) {
      if (wave + TC_WAVES * b < TC_NBLK) {          // wave-uniform
        const tc_f16x8 P00 = *reinterpret_cast<const tc_f16x8*>(lb + pb0[b]);
        const tc_f16x8 P01 = *reinterpret_cast<const tc_f16x8*>(lb + pb1[b]);
        const tc_f16x8 P10 = *reinterpret_cast<const tc_f16x8*>(lb + pb0[b] - TC_WC * 64);
        const tc_f16x8 P11 = *reinterpret_cast<const tc_f16x8*>(lb + pb1[b] - TC_WC * 64);
        if (TC_ABL & 2) {
          asm volatile("" :: "v"(P00), "v"(P01), "v"(P10), "v"(P11));
        } else {
          // phase ph = 2 py + px; tap (py + 2a, px + 2b) on the operand (a, b)
          acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P00, W[0][0], acc[b][0], 0, 0, 0);
          acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P00, W[1][0], acc[b][1], 0, 0, 0);
          acc[b][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P00, W[3][0], acc[b][2], 0, 0, 0);
          acc[b][3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P00, W[4][0], acc[b][3], 0, 0, 0);
          acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P00, W[0][1], acc[b][0], 0, 0, 0);
          acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P00, W[1][1], acc[b][1], 0, 0, 0);
          acc[b][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P00, W[3][1], acc[b][2], 0, 0, 0);
          acc[b][3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P00, W[4][1], acc[b][3], 0, 0, 0);
          acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P01, W[2][0], acc[b][0], 0, 0, 0);       // (0, 0): tap (0, 2)
          acc[b][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P01, W[5][0], acc[b][2], 0, 0, 0);       // (1, 0): tap (1, 2)
          acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P01, W[2][1], acc[b][0], 0, 0, 0);
          acc[b][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P01, W[5][1], acc[b][2], 0, 0, 0);
          acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P10, W[6][0], acc[b][0], 0, 0, 0);       // (0, 0): tap (2, 0)
          acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P10, W[7][0], acc[b][1], 0, 0, 0);       // (0, 1): tap (2, 1)
          acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P10, W[6][1], acc[b][0], 0, 0, 0);
          acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P10, W[7][1], acc[b][1], 0, 0, 0);
          acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P11, W[8][0], acc[b][0], 0, 0, 0);       // (0, 0): tap (2, 2)
          acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(P11, W[8][1], acc[b][0], 0, 0, 0);
        }
      }
      if (!LAST && b == 1) {                        // the first half of the next window: converted behind two blocks' MFMAs
        stage_store(c + 1, buf ^ 1, tc_int<0>());
        stage_load(c + 1, tc_int<1>());
      }
    }
    if (!LAST) {
      stage_store(c + 1, buf ^ 1, tc_int<1>());
      wstage_store(buf ^ 1);
      __syncthreads();
    }
  };

  // ---- prologue
  wstage_load(0);
  stage_load(0, tc_int<0>());
  stage_store(0, 0, tc_int<0>());
  stage_load(0, tc_int<1>());
  stage_store(0, 0, tc_int<1>());
  wstage_store(0);
  __syncthreads();

  for (int c = 0; c + 1 < NC; ++c) chunk(c, tc_int<0>());
  chunk(NC - 1, tc_int<1>());
  if (TC_ABL & 8) { if (acc[0][0][0] != 12345.f) return; }
  __syncthreads();                                  // every wave has read its last operands: the windows become the z tile

  // ---- epilogue, eight channels at a time
  float* Z = reinterpret_cast<float*>(Ls);
  float ymax = 0.f;
  float kf[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) kf[t] = Kf[t];
  const int W2 = 2 * p.w;
  const int64_t hw2 = 4 * hw;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if ((lt >> 3) == pass) {
      float* zc = Z + (lt & 7) * TC_CHS;
#pragma unroll
      for (int b = 0; b < TC_BPW; ++b) {
        if (wave + TC_WAVES * b >= TC_NBLK) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = 16 * (wave + TC_WAVES * b) + 4 * lk + j;
          if (q < TC_NPOS) {
            const int r = q / TC_PC, c = q - r * TC_PC;
            float* zp = zc + (2 * r) * TC_ZP + 2 * c + 3;
            zp[0] = acc[b][0][j];
            zp[1] = acc[b][1][j];
            zp[TC_ZP] = acc[b][2][j];
            zp[TC_ZP + 1] = acc[b][3][j];
          }
        }
      }
    }
    __syncthreads();
#pragma unroll 2
    for (int k = 0; k < 8; ++k) {
      const int gid = tid + 512 * k;                // 8 channels x 32 rows x 16 groups of four outputs
      const int og = gid & 15, oy = (gid >> 4) & 31, ch = gid >> 9;
      const float* zb = Z + ch * TC_CHS + (oy + 1) * TC_ZP + 4 * og + 4;
      float res[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const tc_f32x4 lo = *reinterpret_cast<const tc_f32x4*>(zb + a * TC_ZP);
        const tc_f32x4 hi = *reinterpret_cast<const tc_f32x4*>(zb + a * TC_ZP + 4);
        const float rowv[7] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2]};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) res[q] += rowv[q + cc] * kf[a * 4 + cc];
      }
      const int cl = 8 * pass + ch;                 // channel within the workgroup's 16
      const int64_t pix = (int64_t)(2 * I0 + oy) * W2 + 2 * J0 + 4 * og;
      tc_f32x4 nz = {0.f, 0.f, 0.f, 0.f};
      if (p.noise) nz = *reinterpret_cast<const tc_f32x4*>(p.noise + (int64_t)ib * hw2 + pix) * noise_wg;
      const float sc = Sc[cl], bs = Bs[cl], post = Po[cl];
      tc_f32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float u = res[q] * sc + nz[q] + bs;
        v[q] = fmaxf(u, u * slope) * post;
        ymax = fmaxf(ymax, fabsf(v[q]));
      }
      *reinterpret_cast<tc_f32x4*>(p.y + ((int64_t)ib * p.out_ch + 16 * ot + cl) * hw2 + pix) = v;
    }
    __syncthreads();
  }
  if (p.y_amax) rw_bound_store_wave(p.y_amax, rw_wave_max(ymax));
}

static bool tconv_shape_ok(int out_ch, int in_ch, int h, int w) {
  return out_ch > 0 && out_ch % 16 == 0 && in_ch >= 16 && in_ch % 16 == 0 && in_ch <= 512 && w % TC_TX == 0 && h % TC_TY == 0;
}

extern "C" int rw_tconv_blur_supported(int out_ch, int in_ch, int h, int w) { return tconv_shape_ok(out_ch, in_ch, h, w) ? 1 : 0; }

extern "C" int rw_tconv_blur_f32(const float* x, const float* wp, const float* k4, float* y, int batch, int in_ch,
                                 int out_ch, int h, int w, float w_scale, const rw_conv_epilogue* ep,
                                 const float* post_scale, float u_inv, const float* x_amax, float* y_amax,
                                 rw_stream_t stream) {
  RW_CHECK_ARG(x && wp && k4 && y && x_amax && u_inv > 0.f && batch > 0 && in_ch > 0 && out_ch > 0 && h > 0 && w > 0);
  RW_CHECK_ARG(!ep || ((!ep->noise || ep->noise_w) && (!ep->act || ep->bias)));
  if (!tconv_shape_ok(out_ch, in_ch, h, w)) return RW_ERR_UNSUPPORTED;
  if ((int64_t)in_ch * h * w * 4 > 0x7fffffffLL) return RW_ERR_UNSUPPORTED;
  TconvProblem p = {};
  p.x = x; p.wp = reinterpret_cast<const unsigned char*>(wp); p.y = y; p.k4 = k4;
  p.style = ep ? ep->style : nullptr; p.demod = ep ? ep->demod : nullptr; p.noise = ep ? ep->noise : nullptr;
  p.noise_w = ep ? ep->noise_w : nullptr; p.bias = ep ? ep->bias : nullptr; p.act = ep ? ep->act : 0;
  p.post = post_scale;
  p.batch = batch; p.in_ch = in_ch; p.out_ch = out_ch; p.h = h; p.w = w; p.w_scale = w_scale; p.u_inv = u_inv;
  p.x_amax = x_amax; p.y_amax = y_amax;
  p.tiles_x = w / TC_TX; p.tiles_y = h / TC_TY; p.o_tiles = out_ch / 16;
  const int64_t work = (int64_t)batch * p.tiles_y * p.tiles_x * p.o_tiles;
  if (work <= 0 || work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  if (y_amax && TC_WAVES * work > rw_bound_slot_capacity((int64_t)batch * out_ch * 4 * h * w)) return RW_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(tconv_blur_kernel, dim3((unsigned)work), dim3(512), 0, rw_s(stream), p);
  const int rc = RW_LAUNCH_RESULT();
  if (rc || !y_amax) return rc;
  return rw_bound_finish(y_amax, TC_WAVES * work, rw_s(stream));
}
