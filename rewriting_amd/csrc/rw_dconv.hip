// Direct 3x3 convolution on the 16-bit matrix pipe with exact f16 operand splits (round 4).
//
// Why a direct sum beside the minimal-filtering kernels: on gfx950 v_mfma_f32_16x16x32_f16 retires 16 k FLOPs in 16
// cycles -- 16x the fp32 MFMA -- and the F(4x4,3x3) kernels of rw_wino4.hip, once their multiplies cost that little, are
// paced by their load -> barrier -> transform -> split -> multiply chains (4.1 - 5.7 ms per layer
// where the multiplies alone need 0.5).  The direct sum of the same layer is 1.24 TFLOP x 4 piece products = 2.0 ms of
// the 16-bit pipe, and it needs NO transform: an input value is scaled and split ONCE per workgroup when it is staged
// into LDS, then read nine times as a ready operand.  Wherever the transforms and their chains cost more than 3/4 of
// the multiplies they save -- the layers with few channels and large maps -- the direct sum wins.
//
// The operand split is rw_wino4.hip's (V 2^eV = Vh + Vl, U 2^eU = Uh + Ul, f16 round-to-nearest twice, all four products
// accumulated in fp32), arranged so that neither operand needs a shuffle: a lane's eight k values of
// v_mfma_f32_16x16x32_f16 are FOUR input channels,
//     pixel operand   [Vh c0..c3, Vl c0..c3]                       (what staging writes: 16 bytes per pixel and channel quad)
//     weight operand  [Uh c0..c3, Uh c0..c3]  then  [Ul c0..c3, Ul c0..c3]    (two MFMAs on the same pixel operand)
// so the k-groups lk = 0..3 of an instruction are the four channel quads of a 16-channel chunk.  Pixels are the M side
// (first operand), out-channels the N side: a lane ends with four CONSECUTIVE pixels of one out-channel (16-byte stores).
//
// Workgroup = 4 waves = WM pairs of out-channel blocks x WN = 4 / WM strips of 4 rows x 32 columns; a wave holds
// 2 blocks of 16 out-channels x 8 blocks of 16 pixels (64 accumulator registers).  Per 16-channel chunk the (4 WN + 2) x 34
// window is staged ONCE: wave g loads channel quad g (buffer loads: out-of-image lanes read 0 = the zero padding), scales by
// style 2^eV, splits and writes 16-byte operand words at [pixel][quad ^ ((column >> 2) & 3)] -- the swizzle makes the
// ds_read_b128 of 16 consecutive pixels of one quad hit 16 distinct 16-byte bank groups.  Two chunk buffers: the loads of
// chunk c + 1 are in flight during the MFMAs of chunk c, its conversion follows them, one barrier per chunk.  The weight
// operands come straight from global memory (L2-resident: <= 300 KB for the layers this kernel takes) into registers,
// two taps ahead -- the LDS carries only the pixel operands (one 1-KB read per four MFMAs and wave).
//
// Packed weights: wp[vb][t = 9 c + tap][part 0: Uh, 1: Ul][lane = 16 g + n][4 halves], out-channel 16 vb + n, input
// channels 16 c + 4 g + (0..3) (the kernel doubles them into the operand), + 4 trailing floats [2^-eU, 0, max |U| bits, 0] as in rw_wino4.hip.
//
// MODE 1 (UP) = conv_transpose(stride 2) + 4x4 blur + noise + bias + leaky ReLU in one pass, as rw_wino4.hip's UP: the
// four output-parity phases are 'same' 3x3 convolutions of the INPUT map with composed kernels; here a block of 16
// virtual channels is ONE phase of 16 real channels, and a wave holds the phases (py = wm, px = 0 / 1) -- a lane's two
// accumulators interleave to eight consecutive output pixels.
// MODE 2 (RGB) = the last styled convolution with ToRGB in the epilogue (out_ch == 32, WM == 1: a wave holds all 32
// channels of its pixels; the sum over channels is a 16-lane DPP reduction).
#include "rw_common.h"
#include <stdlib.h>
typedef float dc_f32x4 __attribute__((ext_vector_type(4)));
typedef float dc_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned dc_u32x4 __attribute__((ext_vector_type(4)));
typedef int dc_i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 dc_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 dc_f16x8 __attribute__((ext_vector_type(8)));

struct DconvProblem {
  const float* x; const unsigned char* wp; float* y;
  const float* style; const float* demod; const float* noise; const float* noise_w; const float* bias; int act;
  const float* post;                                // UP: per (image, real channel) factor on the result
  const float* rgb_weight; const float* rgb_style; const float* rgb_bias; const float* rgb_skip; float* rgb_out;
  float rgb_scale;
  int batch, in_ch, out_ch, h, w;                   // out_ch: virtual (UP: 4 x the real count)
  int tiles_x, tiles_y, o_tiles;
  int strided;                                      // specialised kernels: tile k of workgroup b = k * grid + b' instead of a contiguous run
  float w_scale;
  float u_inv;                                      // 1 / (the packed weights' scale), by value
  const float* x_amax; float* y_amax;               // bounds (RW_BOUND_LANES floats; y: + slots), rw_common.h
};

#ifndef DC_ABL
#define DC_ABL 0          // timing ablations (results WRONG): 1 = no staging loads, 2 = no MFMAs, 4 = no weight loads, 8 = no epilogue,
                          // 16 = the epilogue without its global stores (specialised kernels)
#endif
#define DC_PW 34          // columns of the staged window
#ifndef DC_PROF
#define DC_PROF 0         // 1: workgroups 0 and 100 of the specialised kernels leave cycle counts in dc_prof (rw_dconv_prof)
#endif
#if DC_PROF
__device__ unsigned long long dc_prof[32];
extern "C" int rw_dconv_prof(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dc_prof), sizeof(unsigned long long) * 32);
}
#define DC_T() ((unsigned long long)clock64())
#endif

__device__ __forceinline__ int dc_xcd_remap(int id, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, slot = id >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

template <int N> struct rw_int { static constexpr int value = N; };

// position of channel quad g inside the 64 bytes of window column cc: g ^ dc_swz(cc).  ds_read_b128 is serviced in the
// lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32): with this swizzle the 16 lanes of every group -- 16 pixels of
// one quad in the MFMA operand order, at any of the three tap columns -- hit 16 distinct 16-byte slots of the 256-byte bank
// row (searched exhaustively; (cc >> 2) & 3 is 2-way: SQ_LDS_BANK_CONFLICT was 48 % of the LDS cycles), and the eight
// consecutive pixels of a ds_write_b128 group hit eight distinct slots of its 128-byte row.
__device__ __forceinline__ int dc_swz(int cc) { return (cc >> 1) & 3; }
// the operand [u0 u1 u2 u3 u0 u1 u2 u3] of the four halves in w
__device__ __forceinline__ dc_f16x8 dc_expand(dc_f32x2 w) {
  const dc_f32x4 d = {w[0], w[1], w[0], w[1]};
  return __builtin_bit_cast(dc_f16x8, d);
}
// ---- THREE piece products per multiply where four were issued (round 5).  Vl Ul is <= 2^-22 of a product whose other
// pieces are already rounded at 2^-22: it buys nothing.  A pixel word is [Vh c0..3 | Vl c0..3]; against [Uh | Uh] one MFMA
// gives Vh Uh + Vl Uh, and the Vh Ul of TWO taps share a second one: the rows r and r + 1 of a window column meet the taps
// ky = 0 and ky = 1 of the SAME output block, so [Vh(row r) | Vh(row r + 1)] x [Ul(ky 0) | Ul(ky 1)] is both taps' third
// product (two register moves build the pixel operand, the weight operand is two 8-byte loads side by side).  Tap ky = 2
// has no partner in its column and keeps its fourth product ([Ul | Ul]): 5 MFMAs per kernel column and output block
// where 6 were issued, 240 per 16-channel chunk instead of 288 -- the kernels are paced by MFMA issue (profiles/r04s).
// DC_PRODUCTS=4 (compile time) keeps the old form for comparison.
#ifndef DC_PRODUCTS
#define DC_PRODUCTS 3
#endif
// [a0 a1 a2 a3 b0 b1 b2 b3]: the first four halves of two operands (their Vh parts) / two 4-half weight pieces
__device__ __forceinline__ dc_f16x8 dc_pair(dc_f16x8 a, dc_f16x8 b) {
  return dc_f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ dc_f16x8 dc_pair(dc_f32x2 a, dc_f32x2 b) {
  const dc_f32x4 d = {a[0], a[1], b[0], b[1]};
  return __builtin_bit_cast(dc_f16x8, d);
}

// LDS-direct load: lane L's 16 bytes at sbase + voffset land at LDS byte address lds_addr + 16 L
__device__ __forceinline__ void dc_dma_global_b128(unsigned lds_addr, int voffset, const void* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds_addr), "v"(voffset), "s"(sbase)
               : "memory");
}

// sum over the 16 lanes of a DPP row (every lane ends with the sum)
__device__ __forceinline__ float dc_row_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror
  return v;
}

template <int MODE, int WM>
__device__ __forceinline__ void dconv_body(const DconvProblem& p) {
  constexpr bool UP = MODE == 1, RGB = MODE == 2, RGBP = MODE == 3;
  static_assert(!RGB || WM == 1, "ToRGB: one wave holds all out-channels of its pixels");
  static_assert(!UP || WM == 2, "UP: the wave pairs are the two row phases");
  constexpr int WN = 4 / WM, TR = 4 * WN, PR = TR + 2;
  constexpr int NPIX = PR * DC_PW;                 // pixels of the staged window
  constexpr int SI = (NPIX + 63) / 64;             // staging iterations of a wave (64 pixels each)
  constexpr int BUFB = NPIX * 64;                  // bytes per chunk buffer: 16 channels x 4 bytes per pixel
  constexpr int VCH = 32 * WM;                     // (virtual) out-channels of a workgroup
  __shared__ __attribute__((aligned(16))) unsigned char Ls[2 * BUFB];
  __shared__ __attribute__((aligned(16))) float St[512];
  __shared__ float Ct[2][VCH];
  __shared__ float Cr[3][(RGB || RGBP) ? 32 * WM : 1];
  __shared__ float Red[4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lk = lane >> 4, lt = lane & 15;

  const int local = dc_xcd_remap(blockIdx.x, gridDim.x);
  const int ot = local % p.o_tiles;
  int pg = local / p.o_tiles;
  const int tx = pg % p.tiles_x; pg /= p.tiles_x;
  const int ty = pg % p.tiles_y;
  const int ib = pg / p.tiles_y;
  const int y0 = ty * TR, x0 = tx * 32;
  const int64_t hw = (int64_t)p.h * p.w;
  const int NC = p.in_ch >> 4, T = 9 * NC;
  const int real_ch = UP ? p.out_ch >> 2 : p.out_ch;

  // ---- scales (see rw_wino4.hip H16): |x style| <= am < 2^e  ->  |V| = |x style 2^(14 - e)| < 2^14
  float in_scale, out_scale;
  {
    float smax = p.style ? 0.f : 1.f;
    if (p.style)
      for (int i = tid; i < p.in_ch; i += 256) smax = fmaxf(smax, fabsf(p.style[(int64_t)ib * p.in_ch + i]));
    smax = rw_wave_max(smax);
    if (lane == 0) Red[wave] = smax;
    __syncthreads();
    smax = fmaxf(fmaxf(Red[0], Red[1]), fmaxf(Red[2], Red[3]));
    const float am = rw_bound_load(p.x_amax) * smax;
    int e = (int)((__float_as_uint(am) >> 23) & 0xff) - 126;      // am < 2^e
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    in_scale = __uint_as_float((unsigned)(127 + 14 - e) << 23);
    out_scale = __uint_as_float((unsigned)(127 + e - 14) << 23) * p.u_inv;
  }
  const float gain = p.act ? 1.4142135623730951f : 1.f, slope = p.act ? 0.2f : 1.f;
  for (int i = tid; i < p.in_ch; i += 256) St[i] = (p.style ? p.style[(int64_t)ib * p.in_ch + i] : 1.0f) * in_scale;
  if (tid < VCH) {
    const int o = UP ? 16 * ot + (tid & 15) : ot * VCH + tid;
    Ct[0][tid] = (p.demod ? p.demod[(int64_t)ib * real_ch + o] * p.w_scale : p.w_scale) * out_scale * gain;
    Ct[1][tid] = p.act ? p.bias[o] * gain : 0.f;
    if (RGB || RGBP) {
      const float sr = p.rgb_scale * p.rgb_style[(int64_t)ib * p.out_ch + o];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) Cr[cc][tid] = sr * p.rgb_weight[cc * p.out_ch + o];
    }
  }
  const float noise_wg = p.noise ? p.noise_w[0] * gain : 0.f;
  __syncthreads();                                  // St / Ct / Cr are read by other threads than their writers

  // ---- staging: wave g = channel quad g of the chunk; lane = pixel 64 s + lane of the flattened window
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + (int64_t)ib * p.in_ch * hw), 0, (int)((int64_t)p.in_ch * hw * 4), 0x00020000);
  const int hw4 = (int)hw * 4;
  int xoff[SI], loff[SI];
#pragma unroll
  for (int s = 0; s < SI; ++s) {
    const int pi = 64 * s + lane;
    const int r = pi / DC_PW, cc = pi - r * DC_PW;
    const int iy = y0 - 1 + r, ix = x0 - 1 + cc;
    const bool ok = pi < NPIX && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
    xoff[s] = ok ? (iy * p.w + ix) * 4 : 0x7fffffff;
    loff[s] = pi < NPIX ? pi * 64 + ((wave ^ dc_swz(cc)) << 4) : -1;
  }
  // the window of a chunk is staged in two halves (iterations [0, SH) and [SH, SI)): the first is in flight during the
  // first third of the previous chunk's MFMAs and converted after it, the second during the rest -- half the registers
  constexpr int SH = (SI + 1) / 2;
  float raw[SH][4];
  auto stage_load = [&](int c, auto half_tag) __attribute__((always_inline)) {
    constexpr int S0 = decltype(half_tag)::value ? SH : 0, S1 = decltype(half_tag)::value ? SI : SH;
    const int s0 = (16 * c + 4 * wave) * hw4;
#pragma unroll
    for (int s = S0; s < S1; ++s)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        raw[s - S0][k] = (DC_ABL & 1) ? 1.f : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xsrc, xoff[s], s0 + k * hw4, 0));
  };
  auto stage_store = [&](int c, int buf, auto half_tag) __attribute__((always_inline)) {
    constexpr int S0 = decltype(half_tag)::value ? SH : 0, S1 = decltype(half_tag)::value ? SI : SH;
    const dc_f32x4 sv = *reinterpret_cast<const dc_f32x4*>(&St[16 * c + 4 * wave]);
    unsigned char* dst = Ls + buf * BUFB;
#pragma unroll
    for (int s = S0; s < S1; ++s) {
      const float (&rw)[4] = raw[s - S0];
      const float v0 = rw[0] * sv[0], v1 = rw[1] * sv[1], v2 = rw[2] * sv[2], v3 = rw[3] * sv[3];
      const dc_f16x2 h01 = __builtin_convertvector(dc_f32x2{v0, v1}, dc_f16x2);
      const dc_f16x2 h23 = __builtin_convertvector(dc_f32x2{v2, v3}, dc_f16x2);
      float r0, r1, r2, r3;                        // v - (float)h, exact
      asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h01), "v"(v0));
      asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h01), "v"(v1));
      asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h23), "v"(v2));
      asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h23), "v"(v3));
      const dc_f16x2 l01 = __builtin_convertvector(dc_f32x2{r0, r1}, dc_f16x2);
      const dc_f16x2 l23 = __builtin_convertvector(dc_f32x2{r2, r3}, dc_f16x2);
      const dc_f16x8 word = {h01[0], h01[1], h23[0], h23[1], l01[0], l01[1], l23[0], l23[1]};
      if (loff[s] >= 0) *reinterpret_cast<dc_f16x8*>(dst + loff[s]) = word;
    }
  };

  // ---- weight operands of this wave's two blocks: the three taps (ky = 0..2) of one kernel COLUMN kx in registers.  In
  // memory (and in flight) an operand is its four distinct halves; dc_expand() doubles them on arrival.
  const int vb0 = ot * (2 * WM) + 2 * wm;
  const unsigned char* wbase = p.wp + (int64_t)vb0 * T * 1024 + lane * 8;
  dc_f16x8 W[3][2][2];
  dc_f32x2 Wc[3][2][2];
  auto wload = [&](int ky, int t) __attribute__((always_inline)) {
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int part = 0; part < 2; ++part) {
        if (DC_ABL & 4) Wc[ky][ob][part] = dc_f32x2{1.f, 1.f};
        else Wc[ky][ob][part] = *reinterpret_cast<const dc_f32x2*>(wbase + ((int64_t)ob * T + t) * 1024 + part * 512);
      }
  };
  // W[ky][ob][0] = [Uh | Uh] of tap ky; DC_PRODUCTS == 3: W[0][ob][1] = [Ul(ky 0) | Ul(ky 1)], W[2][ob][1] = [Ul | Ul] of
  // tap 2 (W[1][ob][1] unused); == 4: W[ky][ob][1] = [Ul | Ul] of tap ky
  auto wexpand = [&](int ky) __attribute__((always_inline)) {
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
      W[ky][ob][0] = dc_expand(Wc[ky][ob][0]);
      if (DC_PRODUCTS == 4 || ky == 2) W[ky][ob][1] = dc_expand(Wc[ky][ob][1]);
      else if (ky == 1) W[0][ob][1] = dc_pair(Wc[0][ob][1], Wc[1][ob][1]);       // first used in row 1
    }
  };

  dc_f32x4 acc[2][8];
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int pb = 0; pb < 8; ++pb) acc[ob][pb] = dc_f32x4{0.f, 0.f, 0.f, 0.f};

  // pixel operand of window row 4 wn + r (r = 0..5), column 16 half + lt + kx, quad lk: it serves the taps (ky, kx) of
  // the pixel blocks of row r - ky -- read ONCE per kernel column (36 reads per chunk for 288 MFMAs)
  unsigned bbase[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int cc = lt + kx;
    bbase[kx] = (unsigned)((4 * wn * DC_PW + cc) * 64 + ((lk ^ dc_swz(cc)) << 4));
  }
  auto bread = [&](const unsigned char* lb, int kx, int idx) __attribute__((always_inline)) {
    return *reinterpret_cast<const dc_f16x8*>(lb + bbase[kx] + ((idx >> 1) * DC_PW + 16 * (idx & 1)) * 64);
  };

  // One chunk.  Kernel column kx: rows r = 0..5, halves 0 / 1 (idx = 2 r + half); the operand of idx + 1 is read before
  // the MFMAs of idx.  The weights of the NEXT column (or the next chunk's first) replace a tap's registers right after its
  // last use: ky = 0 after row 3, ky = 1 after row 4, ky = 2 after row 5 -- 16 to 24 MFMAs ahead of their first use.
  auto chunk = [&](int c, auto last_tag) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value != 0;
    const int buf = c & 1;
    const unsigned char* lb = Ls + buf * BUFB;
    if (!LAST) stage_load(c + 1, rw_int<0>());
    dc_f16x8 bcur = bread(lb, 0, 0);
    dc_f16x8 bprev[2] = {bcur, bcur};               // the operand one row up, per half (DC_PRODUCTS == 3)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const bool more = !LAST || kx < 2;            // is there a next column
      const int tn = kx < 2 ? 9 * c + kx + 1 : 9 * (c + 1);        // tap (0, next column)
#pragma unroll
      for (int idx = 0; idx < 12; ++idx) {
        const int r = idx >> 1, half = idx & 1;
        if (half == 0 && r < 3) wexpand(r);         // tap ky = r is first used in row r
        dc_f16x8 bnext = bcur;
        if (idx + 1 < 12) bnext = bread(lb, kx, idx + 1);
        else if (kx < 2) bnext = bread(lb, kx + 1, 0);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int pr = r - ky;
          if (pr < 0 || pr > 3) continue;
          const int pb = 2 * pr + half;
          if (DC_ABL & 2) { asm volatile("" :: "v"(bcur), "v"(W[ky][0][0]), "v"(W[ky][0][1]), "v"(W[ky][1][0]), "v"(W[ky][1][1])); continue; }
          acc[0][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bcur, W[ky][0][0], acc[0][pb], 0, 0, 0);
          acc[1][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bcur, W[ky][1][0], acc[1][pb], 0, 0, 0);
          if (DC_PRODUCTS == 4 || ky == 2) {
            acc[0][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bcur, W[ky][0][1], acc[0][pb], 0, 0, 0);
            acc[1][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bcur, W[ky][1][1], acc[1][pb], 0, 0, 0);
          }
        }
        if (DC_PRODUCTS == 3 && r >= 1 && r <= 4 && !(DC_ABL & 2)) {      // Vh Ul of the taps (0, kx) on row r - 1 and (1, kx) on row r
          const int pb = 2 * (r - 1) + half;
          const dc_f16x8 hh = dc_pair(bprev[half], bcur);
          acc[0][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hh, W[0][0][1], acc[0][pb], 0, 0, 0);
          acc[1][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hh, W[0][1][1], acc[1][pb], 0, 0, 0);
        }
        bprev[half] = bcur;
        __builtin_amdgcn_sched_barrier(0);
        if (more && half == 1 && r >= 3) wload(r - 3, tn + 3 * (r - 3));
        bcur = bnext;
      }
      if (!LAST && kx == 0) {
        stage_store(c + 1, buf ^ 1, rw_int<0>());
        stage_load(c + 1, rw_int<1>());
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!LAST) {
      stage_store(c + 1, buf ^ 1, rw_int<1>());
      __syncthreads();
    }
  };

  // ---- prologue
  wload(0, 0);
  wload(1, 3);
  wload(2, 6);
  stage_load(0, rw_int<0>());
  stage_store(0, 0, rw_int<0>());
  stage_load(0, rw_int<1>());
  stage_store(0, 0, rw_int<1>());
  __syncthreads();

  for (int c = 0; c + 1 < NC; ++c) chunk(c, rw_int<0>());
  chunk(NC - 1, rw_int<1>());
  if (DC_ABL & 8) { if (acc[0][0][0] != 12345.f) return; }

  // ---- epilogue: lane (lk, lt) holds out-channel lt of block ob, pixels 4 lk .. 4 lk + 3 of pixel block pb
  float ymax = 0.f;
  if (MODE == 0 || RGBP) {
    const int oc = 16 * (2 * wm) + lt;              // + 16 ob
    const float sc[2] = {Ct[0][oc], Ct[0][oc + 16]}, bs[2] = {Ct[1][oc], Ct[1][oc + 16]};
    float* yb = p.y + ((int64_t)ib * p.out_ch + ot * VCH + oc) * hw;
    // RGBP: this wave's share of the ToRGB that reads the result (models.py:639-655) -- the sum over ITS 32 channels of
    // rgb weight x rgb style x activated value, per colour: partial (ot WM + wm) of out_ch / 32, planes [partial][image][colour]
    float cr[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    float* rb = nullptr;
    if (RGBP) {
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) cr[ob][cc] = Cr[cc][oc + 16 * ob];
      rb = p.rgb_out + (((int64_t)(ot * WM + wm) * p.batch + ib) * 3 + (lt < 3 ? lt : 0)) * hw;
    }
#pragma unroll
    for (int pb = 0; pb < 8; ++pb) {
      const int64_t pix = (int64_t)(y0 + 4 * wn + (pb >> 1)) * p.w + x0 + 16 * (pb & 1) + 4 * lk;
      dc_f32x4 nz = {0.f, 0.f, 0.f, 0.f};
      if (p.noise) nz = *reinterpret_cast<const dc_f32x4*>(p.noise + (int64_t)ib * hw + pix) * noise_wg;
      dc_f32x4 v[2];
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float u = acc[ob][pb][j] * sc[ob] + nz[j] + bs[ob];
          v[ob][j] = fmaxf(u, u * slope);
          ymax = fmaxf(ymax, fabsf(v[ob][j]));
        }
        *reinterpret_cast<dc_f32x4*>(yb + (int64_t)(16 * ob) * hw + pix) = v[ob];
      }
      if (RGBP) {
        dc_f32x4 sum[3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) sum[cc][j] = dc_row_sum(v[0][j] * cr[0][cc] + v[1][j] * cr[1][cc]);
        // every lane of the row holds the sums: lane lt == cc stores colour cc
        if (lt < 3) *reinterpret_cast<dc_f32x4*>(rb + pix) = lt == 0 ? sum[0] : (lt == 1 ? sum[1] : sum[2]);
      }
    }
  } else if (UP) {
    // blocks: ob = px, wm = py; real channel 16 ot + lt; output row 2 (input row) + py, columns 2 (input column) + px
    const float sc = Ct[0][lt], bs = Ct[1][lt];
    const int ch = 16 * ot + lt;
    const float post = p.post ? p.post[(int64_t)ib * real_ch + ch] : 1.f;
    const int W2 = 2 * p.w;
    const int64_t hw2 = 4 * hw;
    float* yb = p.y + ((int64_t)ib * real_ch + ch) * hw2;
#pragma unroll
    for (int pb = 0; pb < 8; ++pb) {
      const int64_t pix = (int64_t)(2 * (y0 + 4 * wn + (pb >> 1)) + wm) * W2 + 2 * (x0 + 16 * (pb & 1) + 4 * lk);
      dc_f32x4 n0 = {0.f, 0.f, 0.f, 0.f}, n1 = n0;
      if (p.noise) {
        const float* np = p.noise + (int64_t)ib * hw2 + pix;
        n0 = *reinterpret_cast<const dc_f32x4*>(np) * noise_wg;
        n1 = *reinterpret_cast<const dc_f32x4*>(np + 4) * noise_wg;
      }
      dc_f32x4 q0 = {acc[0][pb][0], acc[1][pb][0], acc[0][pb][1], acc[1][pb][1]};
      dc_f32x4 q1 = {acc[0][pb][2], acc[1][pb][2], acc[0][pb][3], acc[1][pb][3]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float u0 = q0[k] * sc + n0[k] + bs, u1 = q1[k] * sc + n1[k] + bs;
        q0[k] = fmaxf(u0, u0 * slope) * post; q1[k] = fmaxf(u1, u1 * slope) * post;
        ymax = fmaxf(ymax, fmaxf(fabsf(q0[k]), fabsf(q1[k])));
      }
      *reinterpret_cast<dc_f32x4*>(yb + pix) = q0;
      *reinterpret_cast<dc_f32x4*>(yb + pix + 4) = q1;
    }
  } else {
    // RGB: channels lt and 16 + lt of the wave's pixels; the sum over channels = over the 16 lanes of a DPP row
    const float sc[2] = {Ct[0][lt], Ct[0][lt + 16]}, bs[2] = {Ct[1][lt], Ct[1][lt + 16]};
    float cr[2][3];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) cr[ob][cc] = Cr[cc][lt + 16 * ob];
#pragma unroll
    for (int pb = 0; pb < 8; ++pb) {
      const int64_t pix = (int64_t)(y0 + 4 * wn + (pb >> 1)) * p.w + x0 + 16 * (pb & 1) + 4 * lk;
      dc_f32x4 nz = {0.f, 0.f, 0.f, 0.f};
      if (p.noise) nz = *reinterpret_cast<const dc_f32x4*>(p.noise + (int64_t)ib * hw + pix) * noise_wg;
      dc_f32x4 sum[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u0 = acc[0][pb][j] * sc[0] + nz[j] + bs[0], u1 = acc[1][pb][j] * sc[1] + nz[j] + bs[1];
        u0 = fmaxf(u0, u0 * slope); u1 = fmaxf(u1, u1 * slope);
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) sum[cc][j] = dc_row_sum(u0 * cr[0][cc] + u1 * cr[1][cc]);
      }
      // every lane of the row holds the sums: lane lt == cc stores colour cc
      if (lt < 3) {
        const dc_f32x4 mine = lt == 0 ? sum[0] : (lt == 1 ? sum[1] : sum[2]);
        const int64_t off = ((int64_t)ib * 3 + lt) * hw + pix;
        dc_f32x4 o4 = mine + (p.rgb_bias ? p.rgb_bias[lt] : 0.f);
        if (p.rgb_skip) o4 += *reinterpret_cast<const dc_f32x4*>(p.rgb_skip + off);
        *reinterpret_cast<dc_f32x4*>(p.rgb_out + off) = o4;
      }
    }
  }
  if (!RGB && p.y_amax) rw_bound_store_wave(p.y_amax, rw_wave_max(ymax));       // this wave's slot
}

// ---------------------------------------------------------------------------------------
// Second version: the waves of a workgroup SPECIALISE.  Measured on the kernel above (profiles/r04p): every part of it
// costs about a millisecond of the 4.6 per layer and nothing overlaps -- staging loads (HBM latency), weights (L2), the
// MFMAs and the epilogue take turns, because one wave does all four and its loads return in order: a wait for weights
// (L2, short) also waits for the window loads (HBM, long) issued before them.  Here ONE workgroup of twelve waves owns a CU:
//   waves 0..7 (two per SIMD: a workgroup's waves are dealt to the SIMDs in turn) only multiply: pixel operands AND
//     weight operands from LDS (the weights one half-step = 48 MFMAs ahead; through the L1 they were 288 loads per chunk
//     and CU), the tile's noise requested a chunk ahead, the epilogue from tables in LDS, 16-byte stores;
//   waves 8..11 only stage, one channel quad and one weight block each: piece s (a 16-byte aligned run of four columns
//     per lane and channel + three 1-KB tap pieces of weights) of chunk n + 1 is converted and written, then piece s of
//     chunk n + 2 requested into the same registers -- loads are in flight all the time, a whole interval ahead;
//     everything the compiler can count (LDS-direct loads from inline assembly made its waits nine loads too strict,
//     loop-carried assembly outputs are not safe against its copies).
// One raw s_barrier per chunk joins them (LDS writes waited for, loads in flight across it).  A workgroup takes every
// (grid)th tile -- those running at the same time are neighbours, the 32 of an XCD consecutive.  Tiles are 8 rows x 64
// columns x 64 out-channels; in_ch >= 32 (tables double-buffered by tile parity), a style on load, w % 64 == 0.
// What it reaches (profiles/r04s: cycle counters inside the kernel): the multiplying waves are never short of operands,
// and the two of a SIMD retire an MFMA every 20.5 cycles -- the rate ONE wave issues them at (scripts/probe: 20 - 21.6
// cycles alone, 17.2 - 18.7 with a second wave's MFMAs between) -- at the 1.7 GHz the chip sustains under this load:
// 302 M MFMAs x 20.5 / 1024 SIMDs = 3.6 ms + the tiles' epilogues = 4.2 ms, where the 2.5 PFLOP/s the pipe is priced at
// (16 cycles, 2.4 GHz) would be 2.0.  That is the same 4.2 - 4.4 ms the split F(4x4,3x3) kernel takes with a quarter of
// the multiplies; the direct sums stay an opt-in (closer to the fp32 sum: 4e-7 against 1.1e-6).
// ---------------------------------------------------------------------------------------
// MW multiplying waves (wave < MW) = WM pairs of out-channel blocks x (MW / WM) strips of 4 rows x 32 columns, WC of them side
// by side (a tile is 4 MW / (WM WC) rows x 32 WC columns: the wider, the fewer 128-byte lines its window touches per
// pixel -- a 34-float row is three lines, a 66-float row is three lines too); NL staging waves (4 / NL channel quads each)
template <int MODE, int WM, int MW, int NL, int WC>
__device__ __forceinline__ void dconv_ws_body(const DconvProblem& p) {
  constexpr bool UP = MODE == 1, RGB = MODE == 2, RGBP = MODE == 3, PLAIN = MODE == 0 || MODE == 3;      // RGBP: dconv_body's
  static_assert(!RGB || WM == 1, "ToRGB: one wave holds all out-channels of its pixels");
  static_assert(!UP || WM == 2, "UP: the wave pairs are the two row phases");
  constexpr int WN = MW / WM, WR = WN / WC, TR = 4 * WR, PR = TR + 2, TC = 32 * WC, PW = TC + 2;
  static_assert(WR * WC == WN && WR >= 1, "strips tile the workgroup's pixels");
  constexpr int QW = 4 / NL, NLL = 64 * NL;       // quads per staging wave; staging lanes
  constexpr int NPIX = PR * PW, BUFB = NPIX * 64;
  // staging items: 16-byte aligned runs of four columns x0 - 4 + 4 j .. + 3 (j = 0 .. IPR - 1) of one row -- window columns
  // 4 j - 3 .. 4 j; a lane loads its item of the quad's four channels (4 x 16 bytes) and writes four operand words
  constexpr int IPR = TC / 4 + 2, NITEM = PR * IPR, SI = (NITEM + 63) / 64;
  constexpr int VCH = 32 * WM;
  constexpr int WBUF = 2 * WM * 9 * 1024;           // bytes of a chunk's weight operands: [block][tap][Uh | Ul][lane][4 halves]
  __shared__ __attribute__((aligned(16))) unsigned char Ls[2 * BUFB];
  __shared__ __attribute__((aligned(16))) unsigned char Wl[2 * WBUF];
  __shared__ float Ct[2][2][VCH];
  __shared__ float Cr[2][3][VCH];
  __shared__ float Po[2][16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t hw = (int64_t)p.h * p.w;
  const int NC = p.in_ch >> 4, T = 9 * NC;
  const int real_ch = UP ? p.out_ch >> 2 : p.out_ch;
  const float gain = p.act ? 1.4142135623730951f : 1.f, slope = p.act ? 0.2f : 1.f;

  // this workgroup's run of tiles (ot fastest, then x, y, image: the two out-channel halves of an upsampling layer read
  // the same window back to back)
  const int64_t total = (int64_t)p.batch * p.tiles_y * p.tiles_x * p.o_tiles;
  const int per = (int)((total + gridDim.x - 1) / gridDim.x);
  // strided: the workgroups that run at the same time work on NEIGHBOURING tiles (those of one XCD on 32 consecutive ones)
  // -- their stores fill whole rows of the output planes together; contiguous: a workgroup walks its own run of tiles
  const int bx = p.strided ? dc_xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int t0 = p.strided ? 0 : bx * per;
  const int t1 = p.strided ? (int)((total - bx + gridDim.x - 1) / gridDim.x)
                           : (int)((int64_t)t0 + per < total ? t0 + per : total);       // run = positions [t0, t1)
  if (t0 >= t1) {                                  // (every wave of the launch owns a slot of the bound: rw_common.h)
    if (!RGB && p.y_amax) rw_bound_store_wave(p.y_amax, 0.f);
    return;
  }
  const int N = (t1 - t0) * NC;                    // chunks of the run
  const int t_mul = p.strided ? gridDim.x : 1, t_add = p.strided ? bx : 0;            // tile id of run position k
  auto decode = [&](int pos, int& ot, int& tx, int& ty, int& ib) __attribute__((always_inline)) {
    const int tile = pos * t_mul + t_add;
    ot = tile % p.o_tiles;
    int pg = tile / p.o_tiles;
    tx = pg % p.tiles_x; pg /= p.tiles_x;
    ty = pg % p.tiles_y;
    ib = pg / p.tiles_y;
  };
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): vector loads stay in flight across the barrier
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (wave >= MW) {
    // =========================== staging waves: channel quad g of every chunk ===========================
    const int g2 = QW * (wave - MW), lid = (wave - MW) * 64 + lane;    // quads g2 .. g2 + QW - 1
    const float xam = rw_bound_load(p.x_amax);
    const int hw4 = (int)hw * 4;
    int loff[SI];                                   // LDS byte offset of the item's window column 4 j - 3 (may lie left of the row)
#pragma unroll
    for (int s = 0; s < SI; ++s) {
      const int it = 64 * s + lane;
      const int r = it / IPR, j = it - r * IPR;
      loff[s] = it < NITEM ? (r * PW + 4 * j - 3) * 64 : -1000000;
    }
    // the tile being REQUESTED
    int l_tile = t0, l_c = 0, l_ib = -1, l_ot = 0, l_y0 = 0, l_x0 = 0;
    float in_scale = 1.f, out_scale = 1.f;
    int xoff[SI];
    __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0, 0x00020000);
    // the chunk that is in flight / waits for its conversion
    dc_f32x4 raw[QW][SI][4];                         // [channel k]: four pixels
    float psv[QW][4];
    float a_demod = 1.f, a_bias = 0.f, a_rgb = 0.f, a_post = 1.f, a_oscale = 1.f, a_iscale = 1.f;
    float a_crw[3] = {0.f, 0.f, 0.f};                // RGBP: the ToRGB weights of the tile's out-channel block
    bool a_first = false;
    int a_par = 0;
    float crw[3] = {0.f, 0.f, 0.f};
    if (RGB && lid < 32) {
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) crw[cc] = p.rgb_weight[cc * p.out_ch + lid] * p.rgb_scale;
    }

    // A chunk travels in SI pieces of 64 pixels: piece s of the chunk in flight is converted and written, then piece s of
    // the chunk after it is requested into the same registers -- the loads never arrive as one burst followed by silence
    // (a CU sustains ~10 B / cycle from HBM; a whole chunk requested at once took 5 k cycles to issue and left the
    // memory idle for the other 10 k of the interval: cycle counters, profiles/r04q).
    float sv[QW][4];
    int l_s0 = 0;
    const unsigned char* l_wsrc = p.wp;           // the weight block of the chunk being requested
    auto setup = [&]() __attribute__((always_inline)) {            // the chunk to REQUEST: (l_tile, l_c); loads only, none used here
      if (l_tile >= t1) { l_tile = t1 - 1; l_c = NC - 1; }         // past the run: the last chunk again (never read)
      a_first = l_c == 0;
      if (l_c == 0) {
        int tx, ty, ib;
        decode(l_tile, l_ot, tx, ty, ib);
        l_y0 = ty * TR; l_x0 = tx * TC;
        if (ib != l_ib) {
          l_ib = ib;
          float smax = 0.f;
          for (int i = lane; i < p.in_ch; i += 64) smax = fmaxf(smax, fabsf(p.style[(int64_t)ib * p.in_ch + i]));
          smax = rw_wave_max(smax);
          const float am = xam * smax;
          int e = (int)((__float_as_uint(am) >> 23) & 0xff) - 126;      // am < 2^e
          e = e < -100 ? -100 : (e > 100 ? 100 : e);
          in_scale = __uint_as_float((unsigned)(127 + 14 - e) << 23);
          out_scale = __uint_as_float((unsigned)(127 + e - 14) << 23) * p.u_inv;
          xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)ib * p.in_ch * hw), 0,
                                                   (int)((int64_t)p.in_ch * hw * 4), 0x00020000);
        }
#pragma unroll
        for (int s = 0; s < SI; ++s) {
          const int it = 64 * s + lane;
          const int r = it / IPR, j = it - r * IPR;
          const int iy = l_y0 - 1 + r, ix = l_x0 - 4 + 4 * j;        // the item lies inside the row or outside it as a whole
          const bool ok = it < NITEM && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
          xoff[s] = ok ? (iy * p.w + ix) * 4 : 0x7ffffff0;
        }
        // the tile's tables and noise, requested with its first chunk
        a_par = (l_tile - t0) & 1;
        a_oscale = out_scale;
        if (lid < VCH) {
          const int o = UP ? 16 * l_ot + (lid & 15) : l_ot * VCH + lid;
          a_demod = p.demod ? p.demod[(int64_t)l_ib * real_ch + o] : 1.f;
          a_bias = p.act ? p.bias[o] : 0.f;
          if (RGB || RGBP) a_rgb = p.rgb_style[(int64_t)l_ib * p.out_ch + o];
          if (RGBP) {
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) a_crw[cc] = p.rgb_weight[cc * p.out_ch + o] * p.rgb_scale;
          }
          if (UP && lid < 16 && p.post) a_post = p.post[(int64_t)l_ib * real_ch + o];
        }
      }
#pragma unroll
      for (int q = 0; q < QW; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) psv[q][k] = p.style[(int64_t)l_ib * p.in_ch + 16 * l_c + 4 * (g2 + q) + k];
      a_iscale = in_scale;
      l_s0 = (16 * l_c + 4 * g2) * hw4;
      l_wsrc = p.wp + ((int64_t)(l_ot * (2 * WM) + (wave - MW)) * T + 9 * l_c) * 1024;      // wave-uniform
      if (++l_c == NC) { l_c = 0; ++l_tile; }
    };
    // piece s of a chunk: the lane's pixel item (4 channels x 16 bytes) and three of the nine 1-KB tap pieces of the weight
    // block this wave carries (NL == 2 WM: wave j <-> block j, a 9 KB run of the packed array) -- everything the compiler can
    // count, so its waits leave exactly the younger loads in flight (LDS-direct weight loads from inline assembly, which it
    // cannot see, made every count nine too small: each interval waited for the weights it had just requested)
    static_assert(NL == 2 * WM && SI == 3, "wave j stages weight block j, three tap pieces per pixel piece");
    dc_f32x4 wraw[SI][3];
    auto request_s = [&](int s) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < QW; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (DC_ABL & 1) raw[q][s][k] = dc_f32x4{1.f, 1.f, 1.f, 1.f};
          else raw[q][s][k] = __builtin_bit_cast(dc_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xoff[s], l_s0 + (4 * q + k) * hw4, 0));
        }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (DC_ABL & 4) wraw[s][k] = dc_f32x4{1.f, 1.f, 1.f, 1.f};
        else wraw[s][k] = *reinterpret_cast<const dc_f32x4*>(l_wsrc + (3 * s + k) * 1024 + lane * 16);
      }
    };
    // what setup() requested beside the pixels -> registers / LDS (the first wait of an interval)
    auto tables = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < QW; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) sv[q][k] = psv[q][k] * a_iscale;
      if (a_first) {
        if (lid < VCH) {
          Ct[a_par][0][lid] = a_demod * p.w_scale * a_oscale * gain;
          Ct[a_par][1][lid] = a_bias * gain;
          if (RGB || RGBP) {
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) Cr[a_par][cc][lid] = a_rgb * (RGBP ? a_crw[cc] : crw[cc]);
          }
          if (UP && lid < 16) Po[a_par][lid] = a_post;
        }
      }
    };
    auto deliver_s = [&](int buf, int s) __attribute__((always_inline)) {
      unsigned char* dst = Ls + buf * BUFB;
      unsigned char* wdst = Wl + buf * WBUF + (wave - MW) * 9216 + lane * 16;
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<dc_f32x4*>(wdst + (3 * s + k) * 1024) = wraw[s][k];
      const int it = 64 * s + lane;
      const int j = it % IPR;
#pragma unroll
      for (int q = 0; q < QW; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v0 = raw[q][s][0][e] * sv[q][0], v1 = raw[q][s][1][e] * sv[q][1], v2 = raw[q][s][2][e] * sv[q][2],
                      v3 = raw[q][s][3][e] * sv[q][3];
          const dc_f16x2 h01 = __builtin_convertvector(dc_f32x2{v0, v1}, dc_f16x2);
          const dc_f16x2 h23 = __builtin_convertvector(dc_f32x2{v2, v3}, dc_f16x2);
          float r0, r1, r2, r3;                      // v - (float)h, exact
          asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h01), "v"(v0));
          asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h01), "v"(v1));
          asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h23), "v"(v2));
          asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h23), "v"(v3));
          const dc_f16x2 l01 = __builtin_convertvector(dc_f32x2{r0, r1}, dc_f16x2);
          const dc_f16x2 l23 = __builtin_convertvector(dc_f32x2{r2, r3}, dc_f16x2);
          const dc_f16x8 word = {h01[0], h01[1], h23[0], h23[1], l01[0], l01[1], l23[0], l23[1]};
          const int cc = 4 * j - 3 + e;              // window column of this pixel
          if (it < NITEM && cc >= 0 && cc < PW)
            *reinterpret_cast<dc_f16x8*>(dst + loff[s] + e * 64 + (((g2 + q) ^ dc_swz(cc)) << 4)) = word;
        }
    };

    // chunk 0: requested and delivered; chunk 1: requested
    setup();
#pragma unroll
    for (int s = 0; s < SI; ++s) request_s(s);
    tables();
    __builtin_amdgcn_sched_barrier(0);
    setup();
#pragma unroll
    for (int s = 0; s < SI; ++s) { deliver_s(0, s); __builtin_amdgcn_sched_barrier(0); request_s(s); __builtin_amdgcn_sched_barrier(0); }
    lds_barrier();
#if DC_PROF
    unsigned long long pt_del = 0, pt_req = 0, pt_bar = 0, pt_all = DC_T();
#endif
    for (int n = 0; n < N; ++n) {
#if DC_PROF
      const unsigned long long ta = DC_T();
#endif
      // chunk n + 1 (in flight) -> LDS piece by piece, chunk n + 2 requested behind it (past the run: harmless repeats)
      tables();
      __builtin_amdgcn_sched_barrier(0);
      setup();
#if DC_PROF
      const unsigned long long tb = DC_T();
#endif
#pragma unroll
      for (int s = 0; s < SI; ++s) { deliver_s((n + 1) & 1, s); __builtin_amdgcn_sched_barrier(0); request_s(s); __builtin_amdgcn_sched_barrier(0); }
#if DC_PROF
      const unsigned long long tc = DC_T();
#endif
      lds_barrier();
#if DC_PROF
      const unsigned long long td = DC_T();
      pt_del += tb - ta; pt_req += tc - tb; pt_bar += td - tc;
#endif
    }
#if DC_PROF
    if (wave == MW && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {
      unsigned long long* o = dc_prof + (blockIdx.x == 0 ? 0 : 16);
      o[8] = DC_T() - pt_all; o[9] = pt_del; o[10] = pt_req; o[11] = pt_bar; o[12] = N;
    }
#endif
    if (!RGB && p.y_amax) rw_bound_store_wave(p.y_amax, 0.f);       // a staging wave produced nothing: its slot says so
    return;
  }

  // =========================== multiplying waves ===========================
  const int wm = wave / WN, wn = wave % WN, wr = wn / WC, wc = wn % WC;       // strip: rows 4 wr .., columns 32 wc ..
  const int lk = lane >> 4, lt = lane & 15;
  unsigned bbase[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int cc = lt + kx;
    bbase[kx] = (unsigned)((4 * wr * PW + 32 * wc + cc) * 64 + ((lk ^ dc_swz(cc)) << 4));
  }
  auto bread = [&](const unsigned char* lb, int kx, int idx) __attribute__((always_inline)) {
    return *reinterpret_cast<const dc_f16x8*>(lb + bbase[kx] + ((idx >> 1) * PW + 16 * (idx & 1)) * 64);
  };
  // weights.  A chunk is six HALF-steps (kernel column kx, piece Uh / Ul): 48 MFMAs on 48 different accumulators with the
  // six operands [ky][ob] of that piece -- 24 registers where a whole column's twelve take 48, which is what lets three waves
  // share a SIMD (168 registers; the pixel operands are then read once per half-step: 72 reads per chunk).  They come from
  // the chunk's copy in LDS (8 bytes per lane, doubled into the operand), one half-step ahead: through the L1 they were
  // 288 loads per chunk and CU -- three quarters of everything the CU's memory path carried, and it was full (r04q).
  dc_f16x8 W[3][2];
  dc_f32x2 Wc[3][2];
  auto wread = [&](const unsigned char* wb, int kx, int part) __attribute__((always_inline)) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) {
        if (DC_ABL & 4) Wc[ky][ob] = dc_f32x2{1.f, 1.f};
        else Wc[ky][ob] = *reinterpret_cast<const dc_f32x2*>(wb + ((2 * wm + ob) * 9 + 3 * ky + kx) * 1024 + part * 512 + lane * 8);
      }
  };
  const float noise_wg = p.noise ? p.noise_w[0] * gain : 0.f;
  dc_f32x4 nzp[PLAIN ? 8 : 1];
  dc_f32x4 acc[2][8];
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int pb = 0; pb < 8; ++pb) acc[ob][pb] = dc_f32x4{0.f, 0.f, 0.f, 0.f};
  float ymax = 0.f;
  float rgb_b[3] = {0.f, 0.f, 0.f};
  if (RGB && p.rgb_bias) { rgb_b[0] = p.rgb_bias[0]; rgb_b[1] = p.rgb_bias[1]; rgb_b[2] = p.rgb_bias[2]; }

  int tile = t0, c = 0;
  int ot, tx, ty, ib;
  decode(tile, ot, tx, ty, ib);
  lds_barrier();                                    // chunk 0 is in LDS

#if DC_PROF
  unsigned long long pt_mm = 0, pt_bar = 0, pt_epi = 0, pt_all = DC_T();
#endif
  for (int n = 0; n < N; ++n) {
#if DC_PROF
    const unsigned long long ta = DC_T();
#endif
    const unsigned char* lb = Ls + (n & 1) * BUFB;
    const unsigned char* wb = Wl + (n & 1) * WBUF;
    if (PLAIN && c == NC - 1 && p.noise) {          // the tile's noise, requested a chunk of MFMAs before its epilogue
      const float* np = p.noise + (int64_t)ib * hw + (int64_t)(ty * TR + 4 * wr) * p.w + tx * TC + 32 * wc + 4 * lk;
#pragma unroll
      for (int pb = 0; pb < 8; ++pb) nzp[pb] = *reinterpret_cast<const dc_f32x4*>(np + (int64_t)(pb >> 1) * p.w + 16 * (pb & 1));
    }
    dc_f16x8 bq[3];
    wread(wb, 0, 0);
    bq[0] = bread(lb, 0, 0);
    bq[1] = bread(lb, 0, 1);
    dc_f16x8 bprev[2] = {bq[0], bq[1]};             // the operand one row up, per half (DC_PRODUCTS == 3, part 1)
#pragma unroll
    for (int hs = 0; hs < 6; ++hs) {
      const int kx = hs >> 1, part = hs & 1;
      // part 0: W[ky][ob] = [Uh | Uh] of tap ky.  Part 1, DC_PRODUCTS == 3: W[0][ob] = [Ul(ky 0) | Ul(ky 1)] (meets
      // [Vh(row r - 1) | Vh(row r)]), W[2][ob] = [Ul | Ul] of tap 2; DC_PRODUCTS == 4: [Ul | Ul] of every tap
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) {
        if (DC_PRODUCTS == 4 || part == 0) {
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) W[ky][ob] = dc_expand(Wc[ky][ob]);
        } else {
          W[0][ob] = dc_pair(Wc[0][ob], Wc[1][ob]);
          W[2][ob] = dc_expand(Wc[2][ob]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (hs + 1 < 6) wread(wb, (hs + 1) >> 1, (hs + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int idx = 0; idx < 12; ++idx) {
        const int r = idx >> 1, half = idx & 1;
        const dc_f16x8 b = bq[idx % 3];
        if (idx + 2 < 12) bq[(idx + 2) % 3] = bread(lb, kx, idx + 2);
        else if (hs < 5) bq[(idx + 2) % 3] = bread(lb, (hs + 1) >> 1, idx + 2 - 12);
        if (DC_PRODUCTS == 4 || part == 0) {
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int pr = r - ky;
            if (pr < 0 || pr > 3) continue;
            const int pb = 2 * pr + half;
            if (DC_ABL & 2) { asm volatile("" :: "v"(b), "v"(W[ky][0]), "v"(W[ky][1])); continue; }
            acc[0][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, W[ky][0], acc[0][pb], 0, 0, 0);
            acc[1][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, W[ky][1], acc[1][pb], 0, 0, 0);
          }
        } else if (DC_ABL & 2) {
          asm volatile("" :: "v"(b), "v"(W[0][0]), "v"(W[0][1]), "v"(W[2][0]), "v"(W[2][1]));
        } else {
          if (r >= 1 && r <= 4) {                   // Vh Ul of the taps (0, kx) on row r - 1 and (1, kx) on row r
            const int pb = 2 * (r - 1) + half;
            const dc_f16x8 hh = dc_pair(bprev[half], b);
            acc[0][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hh, W[0][0], acc[0][pb], 0, 0, 0);
            acc[1][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hh, W[0][1], acc[1][pb], 0, 0, 0);
          }
          if (r >= 2) {                             // tap (2, kx): all four products
            const int pb = 2 * (r - 2) + half;
            acc[0][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, W[2][0], acc[0][pb], 0, 0, 0);
            acc[1][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, W[2][1], acc[1][pb], 0, 0, 0);
          }
          bprev[half] = b;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#if DC_PROF
    const unsigned long long tb = DC_T();
#endif
    lds_barrier();
#if DC_PROF
    const unsigned long long tc = DC_T();
    pt_mm += tb - ta; pt_bar += tc - tb;
    if (c + 1 == NC) pt_epi -= tc;
#endif
    if (++c < NC) continue;
    // ---- epilogue of the tile: lane (lk, lt) holds out-channel lt of block ob, pixels 4 lk .. 4 lk + 3 of pixel block pb
    c = 0;
    const int par = (tile - t0) & 1;
    const int y0 = ty * TR, x0 = tx * TC;
    if (!(DC_ABL & 8)) {
      if (PLAIN) {
        const int oc = 16 * (2 * wm) + lt;          // + 16 ob
        const float sc[2] = {Ct[par][0][oc], Ct[par][0][oc + 16]}, bs[2] = {Ct[par][1][oc], Ct[par][1][oc + 16]};
        float* yb = p.y + ((int64_t)ib * p.out_ch + ot * VCH + oc) * hw;
        // RGBP: this wave's share of the ToRGB that reads the result (dconv_body): the sum over ITS 32 channels, per colour --
        // partial (ot WM + wm) of out_ch / 32, planes [partial][image][colour]
        float cr[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        float* rb = nullptr;
        if (RGBP) {
#pragma unroll
          for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) cr[ob][cc] = Cr[par][cc][oc + 16 * ob];
          rb = p.rgb_out + (((int64_t)(ot * WM + wm) * p.batch + ib) * 3 + (lt < 3 ? lt : 0)) * hw;
        }
#pragma unroll
        for (int pb = 0; pb < 8; ++pb) {
          const int64_t pix = (int64_t)(y0 + 4 * wr + (pb >> 1)) * p.w + x0 + 32 * wc + 16 * (pb & 1) + 4 * lk;
          dc_f32x4 nz = {0.f, 0.f, 0.f, 0.f};
          if (p.noise) nz = nzp[pb] * noise_wg;
          dc_f32x4 v[2];
#pragma unroll
          for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float u = acc[ob][pb][j] * sc[ob] + nz[j] + bs[ob];
              v[ob][j] = fmaxf(u, u * slope);
              ymax = fmaxf(ymax, fabsf(v[ob][j]));
            }
            if (!(DC_ABL & 16)) *reinterpret_cast<dc_f32x4*>(yb + (int64_t)(16 * ob) * hw + pix) = v[ob];
          }
          if (RGBP) {
            dc_f32x4 sum[3];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int cc = 0; cc < 3; ++cc) sum[cc][j] = dc_row_sum(v[0][j] * cr[0][cc] + v[1][j] * cr[1][cc]);
            // every lane of the row holds the sums: lane lt == cc stores colour cc
            if (lt < 3 && !(DC_ABL & 16)) *reinterpret_cast<dc_f32x4*>(rb + pix) = lt == 0 ? sum[0] : (lt == 1 ? sum[1] : sum[2]);
          }
        }
      } else if (UP) {
        const float sc = Ct[par][0][lt], bs = Ct[par][1][lt];
        const int ch = 16 * ot + lt;
        const float post = Po[par][lt];
        const int W2 = 2 * p.w;
        float* yb = p.y + ((int64_t)ib * real_ch + ch) * (4 * hw);
#pragma unroll
        for (int pb = 0; pb < 8; ++pb) {
          const int orow = 2 * (4 * wr + (pb >> 1)) + wm, ocol = 2 * (32 * wc + 16 * (pb & 1) + 4 * lk);
          const int64_t pix = (int64_t)(2 * y0 + orow) * W2 + 2 * x0 + ocol;
          dc_f32x4 n0 = {0.f, 0.f, 0.f, 0.f}, n1 = n0;
          if (p.noise) {
            const float* np = p.noise + (int64_t)ib * (4 * hw) + pix;
            n0 = *reinterpret_cast<const dc_f32x4*>(np) * noise_wg;
            n1 = *reinterpret_cast<const dc_f32x4*>(np + 4) * noise_wg;
          }
          dc_f32x4 q0 = {acc[0][pb][0], acc[1][pb][0], acc[0][pb][1], acc[1][pb][1]};
          dc_f32x4 q1 = {acc[0][pb][2], acc[1][pb][2], acc[0][pb][3], acc[1][pb][3]};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float u0 = q0[k] * sc + n0[k] + bs, u1 = q1[k] * sc + n1[k] + bs;
            q0[k] = fmaxf(u0, u0 * slope) * post; q1[k] = fmaxf(u1, u1 * slope) * post;
            ymax = fmaxf(ymax, fmaxf(fabsf(q0[k]), fabsf(q1[k])));
          }
          if (!(DC_ABL & 16)) {
            *reinterpret_cast<dc_f32x4*>(yb + pix) = q0;
            *reinterpret_cast<dc_f32x4*>(yb + pix + 4) = q1;
          }
        }
      } else {
        const float sc[2] = {Ct[par][0][lt], Ct[par][0][lt + 16]}, bs[2] = {Ct[par][1][lt], Ct[par][1][lt + 16]};
        float cr[2][3];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) cr[ob][cc] = Cr[par][cc][lt + 16 * ob];
#pragma unroll
        for (int pb = 0; pb < 8; ++pb) {
          const int64_t pix = (int64_t)(y0 + 4 * wr + (pb >> 1)) * p.w + x0 + 32 * wc + 16 * (pb & 1) + 4 * lk;
          dc_f32x4 nz = {0.f, 0.f, 0.f, 0.f};
          if (p.noise) nz = *reinterpret_cast<const dc_f32x4*>(p.noise + (int64_t)ib * hw + pix) * noise_wg;
          dc_f32x4 sum[3];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float u0 = acc[0][pb][j] * sc[0] + nz[j] + bs[0], u1 = acc[1][pb][j] * sc[1] + nz[j] + bs[1];
            u0 = fmaxf(u0, u0 * slope); u1 = fmaxf(u1, u1 * slope);
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) sum[cc][j] = dc_row_sum(u0 * cr[0][cc] + u1 * cr[1][cc]);
          }
          if (lt < 3) {
            const dc_f32x4 mine = lt == 0 ? sum[0] : (lt == 1 ? sum[1] : sum[2]);
            const int64_t off = ((int64_t)ib * 3 + lt) * hw + pix;
            dc_f32x4 o4 = mine + (lt == 0 ? rgb_b[0] : (lt == 1 ? rgb_b[1] : rgb_b[2]));
            if (p.rgb_skip) o4 += *reinterpret_cast<const dc_f32x4*>(p.rgb_skip + off);
            *reinterpret_cast<dc_f32x4*>(p.rgb_out + off) = o4;
          }
        }
      }
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int pb = 0; pb < 8; ++pb) acc[ob][pb] = dc_f32x4{0.f, 0.f, 0.f, 0.f};
    if (++tile < t1) decode(tile, ot, tx, ty, ib);
#if DC_PROF
    pt_epi += DC_T();
#endif
  }
#if DC_PROF
  if (wave == 0 && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {
    unsigned long long* o = dc_prof + (blockIdx.x == 0 ? 0 : 16);
    o[0] = DC_T() - pt_all; o[1] = pt_mm; o[2] = pt_bar; o[3] = pt_epi; o[4] = N;
  }
#endif
  if (!RGB && p.y_amax) rw_bound_store_wave(p.y_amax, rw_wave_max(ymax));       // this wave's slot
}

// one workgroup per CU: 8 multiplying waves (two per SIMD: the waves of a workgroup are dealt to the SIMDs in turn) + 4 staging;
// tiles 64 columns wide: 16 rows (32 out-channels), 8 rows (64), 4 rows (128)
__global__ void __launch_bounds__(768, 3) dconv_ws_w2_kernel(const DconvProblem p) { dconv_ws_body<0, 2, 8, 4, 2>(p); }
__global__ void __launch_bounds__(768, 3) dconv_ws_up_kernel(const DconvProblem p) { dconv_ws_body<1, 2, 8, 4, 2>(p); }
// (round 6) the same + the ToRGB channel sums of its result: one partial image per multiplying wave's 32 out-channels
__global__ void __launch_bounds__(768, 3) dconv_ws_w2_rgbp_kernel(const DconvProblem p) { dconv_ws_body<3, 2, 8, 4, 2>(p); }
// ToRGB in the epilogue (out_ch == 32: a wave holds every out-channel of its pixels): four multiplying waves (one per SIMD,
// 8 rows x 64 columns per tile) + two staging waves (two channel quads and one weight block each); 120 KB of LDS
__global__ void __launch_bounds__(384, 2) dconv_ws_rgb_kernel(const DconvProblem p) { dconv_ws_body<2, 1, 4, 2, 2>(p); }

__global__ void __launch_bounds__(256, 2) dconv_w1_kernel(const DconvProblem p) { dconv_body<0, 1>(p); }
__global__ void __launch_bounds__(256, 2) dconv_w2_kernel(const DconvProblem p) { dconv_body<0, 2>(p); }
__global__ void __launch_bounds__(256, 2) dconv_w4_kernel(const DconvProblem p) { dconv_body<0, 4>(p); }
__global__ void __launch_bounds__(256, 2) dconv_up_kernel(const DconvProblem p) { dconv_body<1, 2>(p); }
__global__ void __launch_bounds__(256, 2) dconv_rgb_kernel(const DconvProblem p) { dconv_body<2, 1>(p); }
// MODE 3: the stride-1 convolution that ALSO leaves the partial sums of the ToRGB that reads its result (one per 32 channels)
__global__ void __launch_bounds__(256, 2) dconv_w1_rgbp_kernel(const DconvProblem p) { dconv_body<3, 1>(p); }
__global__ void __launch_bounds__(256, 2) dconv_w2_rgbp_kernel(const DconvProblem p) { dconv_body<3, 2>(p); }
__global__ void __launch_bounds__(256, 2) dconv_w4_rgbp_kernel(const DconvProblem p) { dconv_body<3, 4>(p); }

// ---------------------------------------------------------------------------------------
// Packing.  PASS 1 (rw_dconv_*_absmax_f32): max |U| as a bound; PASS 2: the f16 pieces of U su in operand order, su BY VALUE
// (rw_split_weight_scale of the maximum the host read back: see rw_wino4.hip).
// One thread per (virtual out-channel v, input channel i): nine taps.  UP: v's block vb = 4 ot + 2 py + px is phase (py, px)
// of the real channels 16 ot + (v % 16); its 3x3 kernel is rw_wino4.hip's composition of the transposed convolution with the
// blur: h[a][b] = g6[2 - 2a + py][2 - 2b + px], g6 = k' (*) w.
// ---------------------------------------------------------------------------------------
template <int PASS, bool UP>
__global__ void __launch_bounds__(256) pack_dconv_kernel(const float* __restrict__ w, const float* __restrict__ k4,
                                                         unsigned char* __restrict__ wp, float* __restrict__ trailer,
                                                         int vch, int in_ch, float su, float* __restrict__ bound) {
  const int64_t total = (int64_t)vch * in_ch;
  const int NC = in_ch >> 4, T = 9 * NC;
  float m = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % in_ch), v = (int)(idx / in_ch);
    const int vb = v >> 4, n = v & 15;
    float h[9];
    if (UP) {
      const int ot = vb >> 2, py = (vb >> 1) & 1, px = vb & 1;
      const float* g = w + ((int64_t)(16 * ot + n) * in_ch + i) * 9;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const int ty = 2 - 2 * a + py, tx = 2 - 2 * b + px;        // -2 .. 3
          float sum = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const int ky = ty - 1 + c, kx = tx - 1 + d;
              if (ky >= 0 && ky < 3 && kx >= 0 && kx < 3) sum += k4[(3 - c) * 4 + (3 - d)] * g[3 * ky + kx];
            }
          h[3 * a + b] = sum;
        }
    } else {
      const float* g = w + ((int64_t)v * in_ch + i) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) h[t] = g[t];
    }
    if (PASS == 1) {
#pragma unroll
      for (int t = 0; t < 9; ++t) m = fmaxf(m, fabsf(h[t]));
      continue;
    }
    const int c = i >> 4, g4 = (i & 15) >> 2, k = i & 3, lane = 16 * g4 + n;
    _Float16* dst = reinterpret_cast<_Float16*>(wp + ((int64_t)vb * T + 9 * c) * 1024) + lane * 4 + k;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float u = h[t] * su;
      const _Float16 hi = (_Float16)u;
      const _Float16 lo = (_Float16)(u - (float)hi);
      _Float16* d = dst + t * 512;                  // 1024 bytes per tap: [Uh | Ul][64 lanes][4 halves]
      d[0] = hi;
      d[256] = lo;                                  // part 1: + 512 bytes
    }
  }
  __shared__ float red[4];
  if (PASS == 1) rw_bound_store_block_256(bound, m, red);
  if (PASS == 2 && blockIdx.x == 0 && threadIdx.x < 4)       // for inspection only: no kernel reads the trailer
    trailer[threadIdx.x] = threadIdx.x == 0 ? 1.f / su : (threadIdx.x == 1 ? su : 0.f);
}

// RW_DCONV_V=1: the one-role kernels (two workgroups per CU); default: the specialised ones (in_ch >= 32)
static bool dconv_specialised(int in_ch, const rw_conv_epilogue* ep) {
  const char* e = getenv("RW_DCONV_V");
  return in_ch >= 32 && ep && ep->style && !(e && e[0] == '1');
}
// as many workgroups as fit the chip at once, each taking every (grid)th tile
static unsigned dconv_ws_grid(int64_t tiles, int per_cu = 1) {
  const char* e = getenv("RW_DCONV_GRID");
  int64_t g = e ? atoi(e) : (int64_t)rw_cu_count() * per_cu;
  if (g < 1) g = 1;
  return (unsigned)(tiles < g ? tiles : g);
}

static bool dconv_shape_ok(int out_ch, int in_ch, int h, int w) {
  return out_ch > 0 && out_ch % 32 == 0 && in_ch >= 16 && in_ch % 16 == 0 && in_ch <= 512 && w % 32 == 0 && h % 16 == 0;
}

extern "C" int rw_dconv3x3_supported(int out_ch, int in_ch, int h, int w) { return dconv_shape_ok(out_ch, in_ch, h, w) ? 1 : 0; }

extern "C" long long rw_packed_dconv_weight_elems(int out_ch, int in_ch) {
  if (out_ch <= 0 || in_ch <= 0 || out_ch % 16 || in_ch % 16) return -1;
  return 9LL * out_ch * in_ch + 4;
}

template <bool UP>
static int dconv_absmax(const float* w, const float* k4, int vch, int in_ch, float* bound, rw_stream_t stream) {
  const int64_t total = (int64_t)vch * in_ch;
  const int grid = rw_stream_grid(total, 256);
  hipLaunchKernelGGL((pack_dconv_kernel<1, UP>), dim3(grid), dim3(256), 0, rw_s(stream), w, k4, (unsigned char*)nullptr,
                     (float*)nullptr, vch, in_ch, 1.f, bound);
  const int rc = RW_LAUNCH_RESULT();
  if (rc) return rc;
  return rw_bound_finish(bound, grid, rw_s(stream));
}

template <bool UP>
static int dconv_pack(const float* w, const float* k4, float* wp, int vch, int in_ch, float u_scale, rw_stream_t stream) {
  const int64_t total = (int64_t)vch * in_ch;
  float* trailer = wp + 9 * total;
  unsigned char* bytes = reinterpret_cast<unsigned char*>(wp);
  hipLaunchKernelGGL((pack_dconv_kernel<2, UP>), dim3(rw_stream_grid(total, 256)), dim3(256), 0, rw_s(stream), w, k4, bytes,
                     trailer, vch, in_ch, u_scale, (float*)nullptr);
  return RW_LAUNCH_RESULT();
}

extern "C" int rw_dconv_weight_absmax_f32(const float* w, int out_ch, int in_ch, float* bound, rw_stream_t stream) {
  RW_CHECK_ARG(w && bound && out_ch > 0 && in_ch > 0);
  if (out_ch % 16 || in_ch % 16) return RW_ERR_UNSUPPORTED;
  return dconv_absmax<false>(w, nullptr, out_ch, in_ch, bound, stream);
}

extern "C" int rw_pack_dconv_weight_f32(const float* w, float* wp, int out_ch, int in_ch, float u_scale, rw_stream_t stream) {
  RW_CHECK_ARG(w && wp && out_ch > 0 && in_ch > 0 && u_scale > 0.f);
  if (out_ch % 16 || in_ch % 16) return RW_ERR_UNSUPPORTED;
  return dconv_pack<false>(w, nullptr, wp, out_ch, in_ch, u_scale, stream);
}

// after a launch whose waves stored their maxima: the bound of the result
static int dconv_finish(float* y_amax, int64_t nslots, rw_stream_t stream) {
  const int rc = RW_LAUNCH_RESULT();
  if (rc || !y_amax) return rc;
  return rw_bound_finish(y_amax, nslots, rw_s(stream));
}

static void dconv_fill(DconvProblem& p, const float* x, const float* wp, int batch, int in_ch, int vch, int h, int w,
                       float w_scale, const rw_conv_epilogue* ep, float u_inv, const float* x_amax, float* y_amax) {
  p.x = x; p.wp = reinterpret_cast<const unsigned char*>(wp);
  p.u_inv = u_inv;
  p.style = ep ? ep->style : nullptr; p.demod = ep ? ep->demod : nullptr; p.noise = ep ? ep->noise : nullptr;
  p.noise_w = ep ? ep->noise_w : nullptr; p.bias = ep ? ep->bias : nullptr; p.act = ep ? ep->act : 0;
  p.batch = batch; p.in_ch = in_ch; p.out_ch = vch; p.h = h; p.w = w; p.w_scale = w_scale;
  p.x_amax = x_amax; p.y_amax = y_amax;
  p.tiles_x = w / 32;
  const char* e = getenv("RW_DCONV_ORDER");
  p.strided = e ? atoi(e) : 1;
}

extern "C" int rw_dconv3x3_f32(const float* x, const float* wp, float* y, int batch, int in_ch, int out_ch, int h, int w,
                               float w_scale, const rw_conv_epilogue* ep, float u_inv, const float* x_amax, float* y_amax,
                               rw_stream_t stream) {
  RW_CHECK_ARG(x && wp && y && x_amax && u_inv > 0.f && batch > 0 && in_ch > 0 && out_ch > 0 && h > 0 && w > 0);
  RW_CHECK_ARG(!ep || ((!ep->noise || ep->noise_w) && (!ep->act || ep->bias)));
  if (!dconv_shape_ok(out_ch, in_ch, h, w)) return RW_ERR_UNSUPPORTED;
  if ((int64_t)in_ch * h * w * 4 > 0x7fffffffLL) return RW_ERR_UNSUPPORTED;
  DconvProblem p = {};
  dconv_fill(p, x, wp, batch, in_ch, out_ch, h, w, w_scale, ep, u_inv, x_amax, y_amax);
  p.y = y;
  const int64_t cap = rw_bound_slot_capacity((int64_t)batch * out_ch * h * w);
  // out-channels of a workgroup: 128 / 64 / 32 -- the widest that divides (the window is staged once for all of them)
  const char* e = getenv("RW_DCONV_WM");
  int wm = out_ch % 128 == 0 ? 4 : (out_ch % 64 == 0 ? 2 : 1);
  if (e && (atoi(e) == 1 || atoi(e) == 2 || atoi(e) == 4) && out_ch % (32 * atoi(e)) == 0) wm = atoi(e);
  p.o_tiles = out_ch / (32 * wm);
  p.tiles_y = h / (16 / wm);
  const int64_t work = (int64_t)batch * p.tiles_y * p.tiles_x * p.o_tiles;
  if (work <= 0 || work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  if (dconv_specialised(in_ch, ep) && out_ch % 64 == 0 && h % 8 == 0 && w % 64 == 0) {
    // eight multiplying waves: tiles of 64 out-channels x 8 rows x 64 columns (a chunk's weights sit in LDS beside two windows:
    // 128 out-channels per workgroup would not fit, 32 would need twice the pixels)
    p.o_tiles = out_ch / 64; p.tiles_y = h / 8; p.tiles_x = w / 64;
    const unsigned grid = dconv_ws_grid((int64_t)batch * p.tiles_y * p.tiles_x * p.o_tiles);
    if (y_amax && 12LL * grid > cap) return RW_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dconv_ws_w2_kernel, dim3(grid), dim3(768), 0, rw_s(stream), p);
    return dconv_finish(y_amax, 12LL * grid, stream);
  }
  if (y_amax && 4 * work > cap) return RW_ERR_UNSUPPORTED;
  if (wm == 4) hipLaunchKernelGGL(dconv_w4_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  else if (wm == 2) hipLaunchKernelGGL(dconv_w2_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  else hipLaunchKernelGGL(dconv_w1_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  return dconv_finish(y_amax, 4 * work, stream);
}

// ---- the same convolution + the partial sums of the ToRGB that consumes its result (ToRGBF.forward, models.py:639-655: a 1x1
// modulated convolution to three colours, no demodulation): every wave adds up ITS 32 out-channels' contributions while the
// activated values are in registers -- rgb->out receives out_ch / 32 partial images (partial, image, colour, pixel), which
// rw_rgb_combine_f32 sums with the bias and the upsampled running image.  The feature map is written as usual (the next
// layer reads it); what disappears is the second pass over it.  rgb->bias / rgb->skip are not used here.
extern "C" int rw_dconv3x3_rgb_partials(int out_ch) { return out_ch > 0 && out_ch % 32 == 0 ? out_ch / 32 : -1; }
extern "C" int rw_dconv3x3_rgb_partial_f32(const float* x, const float* wp, float* y, int batch, int in_ch, int out_ch, int h,
                                           int w, float w_scale, const rw_conv_epilogue* ep, const rw_rgb_epilogue* rgb,
                                           float u_inv, const float* x_amax, float* y_amax, rw_stream_t stream) {
  RW_CHECK_ARG(x && wp && y && x_amax && u_inv > 0.f && batch > 0 && in_ch > 0 && out_ch > 0 && h > 0 && w > 0);
  RW_CHECK_ARG(rgb && rgb->weight && rgb->style && rgb->out);
  RW_CHECK_ARG(!ep || ((!ep->noise || ep->noise_w) && (!ep->act || ep->bias)));
  if (!dconv_shape_ok(out_ch, in_ch, h, w)) return RW_ERR_UNSUPPORTED;
  if ((int64_t)in_ch * h * w * 4 > 0x7fffffffLL) return RW_ERR_UNSUPPORTED;
  DconvProblem p = {};
  dconv_fill(p, x, wp, batch, in_ch, out_ch, h, w, w_scale, ep, u_inv, x_amax, y_amax);
  p.y = y;
  p.rgb_weight = rgb->weight; p.rgb_style = rgb->style; p.rgb_out = rgb->out; p.rgb_scale = rgb->scale;
  const int64_t cap = rw_bound_slot_capacity((int64_t)batch * out_ch * h * w);
  const char* e = getenv("RW_DCONV_WM");
  int wm = out_ch % 128 == 0 ? 4 : (out_ch % 64 == 0 ? 2 : 1);
  if (e && (atoi(e) == 1 || atoi(e) == 2 || atoi(e) == 4) && out_ch % (32 * atoi(e)) == 0) wm = atoi(e);
  p.o_tiles = out_ch / (32 * wm);
  p.tiles_y = h / (16 / wm);
  const int64_t work = (int64_t)batch * p.tiles_y * p.tiles_x * p.o_tiles;
  if (work <= 0 || work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  if (dconv_specialised(in_ch, ep) && out_ch % 64 == 0 && h % 8 == 0 && w % 64 == 0) {      // as rw_dconv3x3_f32: a style on load
    p.o_tiles = out_ch / 64; p.tiles_y = h / 8; p.tiles_x = w / 64;
    const unsigned grid = dconv_ws_grid((int64_t)batch * p.tiles_y * p.tiles_x * p.o_tiles);
    if (y_amax && 12LL * grid > cap) return RW_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dconv_ws_w2_rgbp_kernel, dim3(grid), dim3(768), 0, rw_s(stream), p);
    return dconv_finish(y_amax, 12LL * grid, stream);
  }
  if (y_amax && 4 * work > cap) return RW_ERR_UNSUPPORTED;
  if (wm == 4) hipLaunchKernelGGL(dconv_w4_rgbp_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  else if (wm == 2) hipLaunchKernelGGL(dconv_w2_rgbp_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  else hipLaunchKernelGGL(dconv_w1_rgbp_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  return dconv_finish(y_amax, 4 * work, stream);
}

// ---- conv_transpose(stride 2) + blur + noise + bias + leaky ReLU in one pass (rw_conv_transpose3x3s2_blur_wino4_f32's operation)
static bool dconv_up_shape_ok(int out_ch, int in_ch, int h, int w) {
  return out_ch > 0 && out_ch % 16 == 0 && in_ch >= 16 && in_ch % 16 == 0 && in_ch <= 512 && w % 32 == 0 && h % 8 == 0;
}
extern "C" int rw_dconv_transpose_blur_supported(int out_ch, int in_ch, int h, int w) {
  return dconv_up_shape_ok(out_ch, in_ch, h, w) ? 1 : 0;
}
extern "C" long long rw_packed_dconv_transpose_blur_weight_elems(int out_ch, int in_ch) {
  if (out_ch <= 0 || in_ch <= 0 || out_ch % 16 || in_ch % 16) return -1;
  return 36LL * out_ch * in_ch + 4;
}
extern "C" int rw_dconv_transpose_blur_weight_absmax_f32(const float* w, const float* k4, int out_ch, int in_ch,
                                                         float* bound, rw_stream_t stream) {
  RW_CHECK_ARG(w && k4 && bound && out_ch > 0 && in_ch > 0);
  if (out_ch % 16 || in_ch % 16) return RW_ERR_UNSUPPORTED;
  return dconv_absmax<true>(w, k4, 4 * out_ch, in_ch, bound, stream);
}
extern "C" int rw_pack_dconv_transpose_blur_weight_f32(const float* w, const float* k4, float* wp, int out_ch, int in_ch,
                                                       float u_scale, rw_stream_t stream) {
  RW_CHECK_ARG(w && k4 && wp && out_ch > 0 && in_ch > 0 && u_scale > 0.f);
  if (out_ch % 16 || in_ch % 16) return RW_ERR_UNSUPPORTED;
  return dconv_pack<true>(w, k4, wp, 4 * out_ch, in_ch, u_scale, stream);
}
extern "C" int rw_dconv_transpose3x3s2_blur_f32(const float* x, const float* wp, float* y, int batch, int in_ch, int out_ch,
                                                int h, int w, float w_scale, const rw_conv_epilogue* ep,
                                                const float* post_scale, float u_inv, const float* x_amax, float* y_amax,
                                                rw_stream_t stream) {
  RW_CHECK_ARG(x && wp && y && x_amax && u_inv > 0.f && batch > 0 && in_ch > 0 && out_ch > 0 && h > 0 && w > 0);
  RW_CHECK_ARG(!ep || ((!ep->noise || ep->noise_w) && (!ep->act || ep->bias)));
  if (!dconv_up_shape_ok(out_ch, in_ch, h, w)) return RW_ERR_UNSUPPORTED;
  if ((int64_t)in_ch * h * w * 4 > 0x7fffffffLL) return RW_ERR_UNSUPPORTED;
  DconvProblem p = {};
  dconv_fill(p, x, wp, batch, in_ch, 4 * out_ch, h, w, w_scale, ep, u_inv, x_amax, y_amax);
  p.y = y; p.post = post_scale;
  const int64_t cap = rw_bound_slot_capacity((int64_t)batch * out_ch * 4 * h * w);
  p.o_tiles = out_ch / 16;
  p.tiles_y = h / 8;
  const int64_t work = (int64_t)batch * p.tiles_y * p.tiles_x * p.o_tiles;
  if (work <= 0 || work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  if (dconv_specialised(in_ch, ep) && w % 64 == 0) {
    p.tiles_y = h / 8; p.tiles_x = w / 64;
    const unsigned grid = dconv_ws_grid((int64_t)batch * p.tiles_y * p.tiles_x * p.o_tiles);
    if (y_amax && 12LL * grid > cap) return RW_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dconv_ws_up_kernel, dim3(grid), dim3(768), 0, rw_s(stream), p);
    return dconv_finish(y_amax, 12LL * grid, stream);
  }
  if (y_amax && 4 * work > cap) return RW_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dconv_up_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  return dconv_finish(y_amax, 4 * work, stream);
}

// ---- the last styled convolution with ToRGB in the epilogue (rw_conv3x3_wino4_to_rgb_f32's operation)
extern "C" int rw_dconv3x3_to_rgb_supported(int out_ch, int in_ch, int h, int w) {
  return out_ch == 32 && dconv_shape_ok(out_ch, in_ch, h, w) ? 1 : 0;
}
extern "C" int rw_dconv3x3_to_rgb_f32(const float* x, const float* wp, int batch, int in_ch, int out_ch, int h, int w,
                                      float w_scale, const rw_conv_epilogue* ep, const rw_rgb_epilogue* rgb, float u_inv,
                                      const float* x_amax, rw_stream_t stream) {
  RW_CHECK_ARG(x && wp && rgb && rgb->weight && rgb->style && rgb->out && x_amax && u_inv > 0.f && batch > 0 && in_ch > 0 && out_ch > 0);
  RW_CHECK_ARG(!ep || ((!ep->noise || ep->noise_w) && (!ep->act || ep->bias)));
  if (!rw_dconv3x3_to_rgb_supported(out_ch, in_ch, h, w)) return RW_ERR_UNSUPPORTED;
  if ((int64_t)in_ch * h * w * 4 > 0x7fffffffLL) return RW_ERR_UNSUPPORTED;
  DconvProblem p = {};
  dconv_fill(p, x, wp, batch, in_ch, out_ch, h, w, w_scale, ep, u_inv, x_amax, nullptr);
  p.rgb_weight = rgb->weight; p.rgb_style = rgb->style; p.rgb_bias = rgb->bias; p.rgb_skip = rgb->skip;
  p.rgb_out = rgb->out; p.rgb_scale = rgb->scale;
  p.o_tiles = 1;
  p.tiles_y = h / 16;
  const int64_t work = (int64_t)batch * p.tiles_y * p.tiles_x;
  if (work <= 0 || work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  // RW_DCONV_V=2 (opt-in: at layer 18's shape it takes 7.4 ms where the kernel below takes 6.7 and the F(4x4) kernel 5.8 --
  // four multiplying waves per CU issue half the matrix rate of eight, profiles/r06l), a style on load, maps 64 columns wide:
  // the persistent workgroup with specialised waves
  const char* ve = getenv("RW_DCONV_V");
  if (ve && ve[0] == '2' && dconv_specialised(in_ch, ep) && h % 8 == 0 && w % 64 == 0) {
    p.tiles_y = h / 8; p.tiles_x = w / 64;
    const unsigned grid = dconv_ws_grid((int64_t)batch * p.tiles_y * p.tiles_x);
    hipLaunchKernelGGL(dconv_ws_rgb_kernel, dim3(grid), dim3(384), 0, rw_s(stream), p);
    return RW_LAUNCH_RESULT();
  }
  hipLaunchKernelGGL(dconv_rgb_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  return RW_LAUNCH_RESULT();
}
