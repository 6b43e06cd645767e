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
// Workgroup shapes (template): <TY = 8, 4 waves> two workgroups per CU (67 KB of LDS, <= 256 registers: a workgroup's
// epilogue runs beside the other one's MFMAs) -- the default; <TY = 16, 8 waves> one per CU (106 KB; halo factor 1.2 instead
// of 1.33).  Wave v owns the position blocks v, v + WAVES, ... x 4 phases (96 / 80 accumulator registers).  Per chunk: the
// chunk's 9 KB of weights and the next window are requested a chunk ahead (weights through LDS), the window converted and
// written behind the first / last blocks' MFMAs, one barrier per chunk.
// THREE piece products (rw_dconv.hip, DC_PRODUCTS): the taps of a phase share the Vh Ul instruction in pairs --
// [Vh(P00) | Vh(P01)] x [Ul(t0) | Ul(t2)] and so on -- 14 MFMAs per block and chunk instead of 18, issued in three tap
// groups (the taps on x[i][j] alone, those that also need x[i][j-1], those on the row above: six operand reads per block
// where four would do, at most five weight operands live).
// Epilogue in two halves of 8 channels (the z tile of 8 channels, 2 (TY + 2) x 72 floats each, reuses the window buffers):
// accumulators -> LDS, barrier, blur, demodulation, noise, bias, leaky ReLU, post scale, max |y| for the bound, 16-byte
// stores.  The blur: when the FIR is an outer product kv x kh (the reference's always is: make_kernel([1,3,3,1])) a
// thread walks DOWN a strip of four output columns -- every z row is read once (two 16-byte LDS reads), filtered
// horizontally (16 multiply-adds), and the last four filtered rows give an output row (16 more): 9.5 multiply-adds per
// output where the direct 16-tap form spends 16, and a third of its LDS reads; any other FIR takes the 16-tap form.
#include "rw_common.h"
#include <stdlib.h>
#include <stdio.h>
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

#define TC_TX 32
#define TC_PC (TC_TX + 2)                 // position columns
#define TC_WC (TC_TX + 3)                 // window columns: J0 - 2 .. J0 + TX
#define TC_ZP 72                          // z row pitch (floats): columns 3 .. 70 are written, 4 .. 70 read
#define TC_WCH 9216                       // bytes of a chunk's weights of one 16-channel block: 9 taps x [Uh | Ul] x 512

#ifndef TC_LDS_PAD
#define TC_LDS_PAD 0
#endif
#ifndef TC_ITEM_G
#define TC_ITEM_G 1       // tconv_body: lane -> item order within a piece (1: consecutive; 4: see there)
#endif
#ifndef TC_AHEAD
#define TC_AHEAD 1        // tconv_body: how many position blocks ahead the pixel operands are requested (1 or 2)
#endif
#ifndef TC_STAGE_PRIO
#define TC_STAGE_PRIO 0   // persistent forms: s_setprio of the staging waves (0: none)
#endif
#ifndef TC_ABL
#define TC_ABL 0          // timing ablations (results WRONG): 1 = no staging loads, 2 = no MFMAs, 8 = no epilogue,
                          // 16 = the epilogue without its global stores, 32 = without its noise loads, 64 = without the blur
                          // arithmetic (z written, one row read per output row), 128 = three position blocks per wave only
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
__device__ __forceinline__ tc_f16x8 tc_pair(tc_f32x2 a, tc_f32x2 b) {
  const tc_f32x4 d = {a[0], a[1], b[0], b[1]};
  return __builtin_bit_cast(tc_f16x8, d);
}
// [a0 a1 a2 a3 b0 b1 b2 b3]: the Vh parts of two pixel words
__device__ __forceinline__ tc_f16x8 tc_pair(tc_f16x8 a, tc_f16x8 b) {
  return tc_f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// [h0 h1 h2 h3 q0 q1 q2 q3]: a Vh half-word and the Vh part of a pixel word
__device__ __forceinline__ tc_f16x8 tc_pair_hq(tc_f32x2 h, tc_f16x8 q) {
  const tc_f32x4 qq = __builtin_bit_cast(tc_f32x4, q);
  const tc_f32x4 d = {h[0], h[1], qq[0], qq[1]};
  return __builtin_bit_cast(tc_f16x8, d);
}
#ifndef TC_PROF
#define TC_PROF 0         // 1: workgroups 0 and 100 leave cycle counts of wave 0 (multiplying) and wave 4 (staging) in tc_prof
#endif
#if TC_PROF
__device__ unsigned long long tc_prof[64];
extern "C" int rw_tconv_prof(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tc_prof), sizeof(unsigned long long) * 64);
}
#define TP_DECL(...) unsigned long long __VA_ARGS__
#define TP_NOW(t) t = (unsigned long long)clock64()
#define TP_ADD(var, t) { const unsigned long long tp_n = (unsigned long long)clock64(); var += tp_n - t; t = tp_n; }
#else
#define TP_DECL(...)
#define TP_NOW(t)
#define TP_ADD(var, t)
#endif
template <int N> struct tc_int { static constexpr int value = N; };
// a * b as ONE v_mul_f32: left to the vectoriser, products of values that sit in different 16-byte load results become
// v_pk_mul_f32 on register pairs assembled with two v_mov_b32 each
__device__ __forceinline__ float tc_mul(float a, float b) {
  float r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int TY, int WAVES, int NT>
__device__ __forceinline__ void tconv_body(const TconvProblem& p) {
  constexpr int THREADS = 64 * WAVES;
  constexpr int PR = TY + 2;                        // position rows
  constexpr int NPOS = PR * TC_PC, NBLK = (NPOS + 15) / 16, BPW = (NBLK + WAVES - 1) / WAVES;
  constexpr int WR = TY + 3, NPIX = WR * TC_WC;    // window rows: input rows I0 - 2 .. I0 + TY
  constexpr int BUFB = NPIX * 64;                   // bytes of a window buffer: 16 channels x (2 + 2) bytes per pixel
  constexpr int ZR = 2 * PR, CHS = ZR * TC_ZP + 4;  // z rows of the tile; z channel stride (floats)
  static_assert(8 * CHS * 4 <= 2 * BUFB, "the z tile of eight channels fits the window buffers");
  constexpr int SR = 2 * TY * 128 / THREADS;        // output rows of a thread's strip in the epilogue
  static_assert(WAVES % 4 == 0 && THREADS % 128 == 0 && SR * (THREADS / 128) == 2 * TY && (SR == 4 || SR == 8),
                "four or eight output rows per strip segment");
  // NT blocks of 16 out-channels per workgroup (p.o_tiles = out_ch / OC): a pixel operand read from LDS multiplies into NT
  // accumulator blocks, the window is staged (requested, converted, written) once per OC out-channels
  constexpr int OC = 16 * NT, WCH = NT * TC_WCH;
  __shared__ __attribute__((aligned(16))) unsigned char Ls[2 * BUFB];
  __shared__ __attribute__((aligned(16))) unsigned char Wl[2 * WCH];
  __shared__ __attribute__((aligned(16))) float St[512];
  __shared__ float Sc[OC], Bs[OC], Po[OC], Kf[16], Red[WAVES];
#if TC_LDS_PAD
  // (experiment / workaround, TY == 16 only: claim the rest of the CU's 160 KB of LDS so that no workgroup that uses LDS --
  // to_rgb_kernel -- can share the CU: profiles/r05i)
  constexpr int PADB = TY == 16 ? 163840 - 2 * BUFB - 2 * WCH - 2048 - 4 * 64 - 4 * WAVES - 64 : 16;
  __shared__ unsigned char Pad[PADB];
  if (p.batch < 0) Pad[threadIdx.x] = 1;           // never true: keeps the array allocated
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4, lt = lane & 15;

  const int local = tc_xcd_remap(blockIdx.x, gridDim.x);
  const int ot = local % p.o_tiles;
  int pg = local / p.o_tiles;
  const int tx = pg % p.tiles_x; pg /= p.tiles_x;
  const int ty = pg % p.tiles_y;
  const int ib = pg / p.tiles_y;
  const int I0 = ty * TY, J0 = tx * TC_TX;
  const int64_t hw = (int64_t)p.h * p.w;
  const int NC = p.in_ch >> 4, T = 9 * NC;

  // ---- scales (rw_dconv.hip): |x style| <= am < 2^e  ->  |V| = |x style 2^(14 - e)| < 2^14
  float in_scale, out_scale;
  {
    float smax = p.style ? 0.f : 1.f;
    if (p.style)
      for (int i = tid; i < p.in_ch; i += THREADS) smax = fmaxf(smax, fabsf(p.style[(int64_t)ib * p.in_ch + i]));
    smax = rw_wave_max(smax);
    if (lane == 0) Red[wave] = smax;
    __syncthreads();
    smax = Red[0];
#pragma unroll
    for (int v = 1; v < WAVES; ++v) smax = fmaxf(smax, Red[v]);
    const float am = rw_bound_load(p.x_amax) * smax;
    int e = (int)((__float_as_uint(am) >> 23) & 0xff) - 126;      // am < 2^e
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    in_scale = __uint_as_float((unsigned)(127 + 14 - e) << 23);
    out_scale = __uint_as_float((unsigned)(127 + e - 14) << 23) * p.u_inv;
  }
  const float gain = p.act ? 1.4142135623730951f : 1.f, slope = p.act ? 0.2f : 1.f;
  for (int i = tid; i < p.in_ch; i += THREADS) St[i] = (p.style ? p.style[(int64_t)ib * p.in_ch + i] : 1.0f) * in_scale;
  if (tid < OC) {
    const int o = OC * ot + tid;
    Sc[tid] = (p.demod ? p.demod[(int64_t)ib * p.out_ch + o] * p.w_scale : p.w_scale) * out_scale * gain;
    Bs[tid] = p.act ? p.bias[o] * gain : 0.f;
    Po[tid] = p.post ? p.post[(int64_t)ib * p.out_ch + o] : 1.f;
  }
  if (tid < 16) {
    const int a = tid >> 2, c = tid & 3;
    Kf[tid] = p.k4[(3 - a) * 4 + (3 - c)];        // flipped, as upfirdn2d applies it
  }
  const float noise_wg = p.noise ? p.noise_w[0] * gain : 0.f;
  __syncthreads();                                  // the tables are read by other threads than their writers

  // ---- staging: wave v stages channel quad g = v & 3
  // A lane's ITEM is a 16-byte aligned run of four window columns (4 j - 2 .. 4 j + 1 of the window: input columns
  // J0 - 4 + 4 j ..) of one row: four 16-byte loads -- one per channel of the quad -- bring 4 pixels x 4 channels, i.e. four
  // LDS words.  (Round 5 staged one pixel per lane with 4-byte loads: the 24 loads a thread issued per chunk took ~20 cycles
  // EACH to issue -- cycle counters, profiles/r06q: the vector-memory address path, not latency or bandwidth, paced the
  // window; requesting both halves a whole chunk ahead made it slower.)  Items outside the image read 0 through the
  // descriptor's range check; the two columns left of the window (j = 0) and the three right of it (j = IPR - 1) are loaded
  // and dropped.  Piece s of wave v: item LPP (NSL s + (v >> 2)) + lane of the WR x IPR items, lanes < LPP -- every wave has
  // every piece (48 of 64 lanes each at TY = 16): no wave-dependent branch around a request, which would make hipcc's vmcnt
  // waits for the OLDER requests assume it absent (the software pipeline below keeps up to eleven in flight).
  constexpr int NSL = WAVES / 4;
  const int g = wave & 3, hsel = wave >> 2;
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + (int64_t)ib * p.in_ch * hw), 0, (int)((int64_t)p.in_ch * hw * 4), 0x00020000);
  const int hw4 = (int)hw * 4;
  constexpr int IPR = TC_TX / 4 + 2, NITEM = WR * IPR;
  constexpr int SI = (NITEM + 64 * NSL - 1) / (64 * NSL), LPP = ((NITEM + SI * NSL - 1) / (SI * NSL) + 3) & ~3;
  static_assert(SI <= 2 && LPP <= 64 && LPP * SI * NSL >= NITEM, "one or two pieces per wave cover the window");
  int xoff[SI], lofa[SI], lofb[SI], lok[SI];        // LDS byte offsets of the item's pixels 0 / 2 (1 / 3: + 64); bit e of lok: pixel e is a window pixel
#pragma unroll
  for (int s = 0; s < SI; ++s) {
    // (TC_ITEM_G == 4: lane l takes item (l & 3) LPP / 4 + (l >> 2) of the piece -- the eight lanes of a ds_write_b128 group then
    // write to two rows and both item parities: the four 16-byte slots the operand swizzle can give pixels 4 columns apart)
    const int li = TC_ITEM_G == 4 ? (lane & 3) * (LPP / 4) + (lane >> 2) : lane;
    const int it = LPP * (NSL * s + hsel) + li;
    const bool mine = lane < LPP && it < NITEM;
    const int r = it / IPR, j = it - r * IPR;
    const int iy = I0 - 2 + r, ix = J0 - 4 + 4 * j;                // the item lies inside the row or outside it as a whole
    const bool ok = mine && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
    xoff[s] = ok ? (iy * p.w + ix) * 4 : 0x7ffffff0;
    const int cc0 = 4 * j - 2;                                      // window column of pixel 0
    lofa[s] = (r * TC_WC + cc0) * 64 + ((g ^ tc_swz(cc0 & 0x7ffffffe)) << 4);          // pixels 0, 1 share a swizzle ((cc >> 1) & 3) ...
    lofb[s] = (r * TC_WC + cc0 + 2) * 64 + ((g ^ tc_swz(cc0 + 2)) << 4);               // ... pixels 2, 3 the next one
    int m = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) m |= (mine && cc0 + e >= 0 && cc0 + e < TC_WC) ? (1 << e) : 0;
    lok[s] = m;
  }
  tc_f32x4 raw[SI][4];                              // [piece][channel]: four pixels
  // channels K0 .. K1 - 1 of piece s of chunk c requested
  auto stage_load_part = [&](int c, auto s_tag, auto k0_tag, auto k1_tag) __attribute__((always_inline)) {
    constexpr int s = decltype(s_tag)::value, K0 = decltype(k0_tag)::value, K1 = decltype(k1_tag)::value;
    const int s0 = (16 * c + 4 * g) * hw4;
#pragma unroll
    for (int k = K0; k < K1; ++k) {
      if (TC_ABL & 1) raw[s][k] = tc_f32x4{1.f, 1.f, 1.f, 1.f};
      else raw[s][k] = __builtin_bit_cast(tc_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xoff[s], s0 + k * hw4, 0));
    }
  };
  // pixels E0 .. E1 - 1 of piece s converted (style, exact f16 pair split) and written
  auto stage_store_part = [&](int c, int buf, auto s_tag, auto e0_tag, auto e1_tag) __attribute__((always_inline)) {
    constexpr int s = decltype(s_tag)::value, E0 = decltype(e0_tag)::value, E1 = decltype(e1_tag)::value;
    const tc_f32x4 sv = *reinterpret_cast<const tc_f32x4*>(&St[16 * c + 4 * g]);
    unsigned char* dst = Ls + buf * BUFB;
#pragma unroll
    for (int e = E0; e < E1; ++e) {
      const float v0 = tc_mul(raw[s][0][e], sv[0]), v1 = tc_mul(raw[s][1][e], sv[1]), v2 = tc_mul(raw[s][2][e], sv[2]),
                  v3 = tc_mul(raw[s][3][e], sv[3]);
      const tc_f16x2 h01 = __builtin_convertvector(tc_f32x2{v0, v1}, tc_f16x2);
      const tc_f16x2 h23 = __builtin_convertvector(tc_f32x2{v2, v3}, tc_f16x2);
      float r0, r1, r2, r3;                        // v - (float)h, exact
      asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h01), "v"(v0));
      asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h01), "v"(v1));
      asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h23), "v"(v2));
      asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h23), "v"(v3));
      const tc_f16x2 l01 = __builtin_convertvector(tc_f32x2{r0, r1}, tc_f16x2);
      const tc_f16x2 l23 = __builtin_convertvector(tc_f32x2{r2, r3}, tc_f16x2);
      tc_f16x8 word = {h01[0], h01[1], h23[0], h23[1], l01[0], l01[1], l23[0], l23[1]};
      if (TC_ABL & 256)                            // (timing ablation: no conversion arithmetic; results wrong)
        word = __builtin_bit_cast(tc_f16x8, tc_f32x4{raw[s][0][e], raw[s][1][e], raw[s][2][e], raw[s][3][e]});
      if (TC_ABL & 512) { asm volatile("" :: "v"(word)); }            // (timing ablation: no LDS store of the window)
      else if (lok[s] & (1 << e)) *reinterpret_cast<tc_f16x8*>(dst + (e < 2 ? lofa[s] : lofb[s]) + (e & 1) * 64) = word;
    }
  };
  // the chunk's weights of this workgroup's NT out-channel blocks: NT x 9216 contiguous bytes of the packed array -> LDS
  const unsigned char* wsrc = p.wp + (int64_t)(NT * ot) * T * 1024;
  constexpr int NPC = NT * (TC_WCH / 16), NWP = (NPC + THREADS - 1) / THREADS;      // 16-byte pieces, per thread
  tc_f32x4 wraw[NWP];
  auto wstage_load = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NWP; ++k) {                 // (no branch around a request: the threads past the end read the last piece again)
      const int pc = tid + k * THREADS < NPC ? tid + k * THREADS : NPC - 1;
      const int n = pc / (TC_WCH / 16), r = pc - n * (TC_WCH / 16);
      wraw[k] = *reinterpret_cast<const tc_f32x4*>(wsrc + ((int64_t)n * T + 9 * c) * 1024 + r * 16);
    }
  };
  auto wstage_store = [&](int buf) __attribute__((always_inline)) {
    unsigned char* dst = Wl + buf * WCH;
#pragma unroll
    for (int k = 0; k < NWP; ++k)
      if ((k + 1) * THREADS <= NPC || tid + k * THREADS < NPC)
        *reinterpret_cast<tc_f32x4*>(dst + (tid + k * THREADS) * 16) = wraw[k];
  };

  // ---- this wave's position blocks: operand addresses of the lane's position q = 16 blk + lt (clamped), pixel offsets
  // (a, b) = x[i - a][j - b] at window pixel (r + 1 - a, c + 1 - b)
  // (the ROW ABOVE, a = 1; the position's own row is + TC_WC * 64: an immediate.  They include the window buffer's offset and
  // move by +- BUFB at the end of a chunk -- 2 BPW additions per chunk where `buffer base + offset` cost one per operand read)
  unsigned pb0[BPW], pb1[BPW];                     // column offset b = 0 / 1 at row offset a = 1
#pragma unroll
  for (int b = 0; b < BPW; ++b) {
    int q = 16 * (wave + WAVES * b) + lt;
    q = q < NPOS ? q : NPOS - 1;
    const int r = q / TC_PC, c = q - r * TC_PC;
    pb0[b] = (unsigned)((r * TC_WC + c + 1) * 64 + ((lk ^ tc_swz(c + 1)) << 4));
    pb1[b] = (unsigned)((r * TC_WC + c) * 64 + ((lk ^ tc_swz(c)) << 4));
  }

  tc_f32x4 acc[BPW][NT][4];
#pragma unroll
  for (int b = 0; b < BPW; ++b)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) acc[b][n][ph] = tc_f32x4{0.f, 0.f, 0.f, 0.f};

  // One chunk in THREE tap groups, so that at most five weight operands per out-channel block (20 NT registers) are live
  // beside the accumulators: G0 = the taps on x[i][j] alone, G1 = those that also need x[i][j-1], G2 = those on the row above.
  // Tap t = 3 ky + kx of phase ph = 2 py + px sits on the operand (a, b) = x[i-a][j-b] with ky = py + 2a, kx = px + 2b.
  // Three piece products (rw_dconv.hip): Vh Uh + Vl Uh by [Vh | Vl] x [Uh | Uh]; the Vh Ul of two taps of one phase share an
  // instruction, [Vh(P) | Vh(Q)] x [Ul(tP) | Ul(tQ)]; tap (1, 1), alone in its phase, keeps all four ([Ul | Ul]).
#define TB_MFMA(PH, A, B) acc[b][n][PH] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc[b][n][PH], 0, 0, 0)
#define TB_UH(T_) tc_expand(*reinterpret_cast<const tc_f32x2*>(wb + n * TC_WCH + (T_) * 1024))
#define TB_UL(T_) (*reinterpret_cast<const tc_f32x2*>(wb + n * TC_WCH + (T_) * 1024 + 512))
#define TC_PIX(OFF) (*reinterpret_cast<const tc_f16x8*>(lb + (OFF)))
#define TC_VH(OFF) (*reinterpret_cast<const tc_f32x2*>(lb + (OFF)))
// (the persistent kernels below: one out-channel block per workgroup)
#define TC_MFMA(PH, A, B) acc[b][PH] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc[b][PH], 0, 0, 0)
#define TC_UH(T_) tc_expand(*reinterpret_cast<const tc_f32x2*>(wb + (T_) * 1024))
#define TC_UL(T_) (*reinterpret_cast<const tc_f32x2*>(wb + (T_) * 1024 + 512))
  TP_DECL(tp = 0, tp_all = 0, tp_g0 = 0, tp_s0 = 0, tp_g1 = 0, tp_g2 = 0, tp_s1 = 0, tp_bar = 0, tp_pro = 0);
  auto chunk = [&](int c, auto last_tag) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value != 0;
    const int buf = c & 1;
    const unsigned char* lb = Ls;                   // (pb0 / pb1 carry the buffer's offset)
    const unsigned char* wb = Wl + buf * WCH + lane * 8;
    TP_NOW(tp);
    // The software pipeline: what is requested during chunk c (piece 0 behind the first tap group's blocks, piece 1 behind the
    // second group's, the weights behind the third's) belongs to chunk c + 2 and is converted / written during chunk c + 1
    // (behind the first blocks of the same groups, just before the registers are requested again): every load has a whole
    // chunk to arrive, the requests are spread (eight waves requesting a window at once stall at the issue of their loads and
    // the MFMAs behind them wait: profiles/r06s), the conversion sits between MFMAs.  The chunk barrier waits for LDS only.
    // No branches around the requests: past the end they read the last chunk again, into registers nobody converts -- with
    // them inside conditionals hipcc waits for vmcnt(0) at the first conversion of every chunk.
    constexpr bool st = !LAST;
    const int c2 = c + 2 < NC ? c + 2 : NC - 1;
    auto hook = [&](auto grp_tag, int b) __attribute__((always_inline)) {
      constexpr int grp = decltype(grp_tag)::value;
#if TC_PROF
      if (b < 4) { if (grp == 0) { TP_ADD(tp_g0, tp); } else if (grp == 1) { TP_ADD(tp_g1, tp); } else { TP_ADD(tp_g2, tp); } }
#endif
      if constexpr (grp < SI) {
        if (b == 0 && st) stage_store_part(c + 1, buf ^ 1, tc_int<grp>(), tc_int<0>(), tc_int<2>());
        if (b == 1 && st) stage_store_part(c + 1, buf ^ 1, tc_int<grp>(), tc_int<2>(), tc_int<4>());
        if constexpr (BPW >= 4) {
          if (b == 2) stage_load_part(c2, tc_int<grp>(), tc_int<0>(), tc_int<2>());
          if (b == 3) stage_load_part(c2, tc_int<grp>(), tc_int<2>(), tc_int<4>());
        } else {
          if (b == 2) stage_load_part(c2, tc_int<grp>(), tc_int<0>(), tc_int<4>());
        }
      } else if constexpr (grp == 2) {
        if (b == 0 && st) wstage_store(buf ^ 1);
        if (b == 1) wstage_load(c2);
      }
#if TC_PROF
      if (b < 2) { TP_ADD(tp_s0, tp); } else if (b < 4) { TP_ADD(tp_s1, tp); }
#endif
    };
    // (operands one block ahead, the order pinned: left to itself the scheduler hoists every block's LDS reads to the top
    // of a group and spills; only the last block of a wave can be missing: NBLK > WAVES (BPW - 1))
    constexpr int LB = BPW - 1;
    const bool last_ok = wave + WAVES * LB < NBLK;  // wave-uniform
    {
      tc_f16x8 u0[NT], u1[NT], u3[NT], u4[NT], l4[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) { u0[n] = TB_UH(0); u1[n] = TB_UH(1); u3[n] = TB_UH(3); u4[n] = TB_UH(4); l4[n] = tc_expand(TB_UL(4)); }
      // (TC_AHEAD blocks ahead: 1 = round 5; 2 = the pixel operands of block b + 2 requested before block b's MFMAs)
      tc_f16x8 pc = TC_PIX(pb0[0] + TC_WC * 64), pd = pc;
      if (TC_AHEAD == 2 && BPW > 1) pd = TC_PIX(pb0[1] + TC_WC * 64);
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        tc_f16x8 pn = TC_AHEAD == 2 ? pd : pc;
        if (b + TC_AHEAD < BPW) pn = TC_PIX(pb0[b + TC_AHEAD] + TC_WC * 64);
        if (b < LB || last_ok) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if (TC_ABL & 2) { asm volatile("" :: "v"(pc), "v"(u0[n]), "v"(u1[n]), "v"(u3[n]), "v"(u4[n]), "v"(l4[n])); }
            else {
              TB_MFMA(3, pc, u4[n]); TB_MFMA(0, pc, u0[n]); TB_MFMA(1, pc, u1[n]); TB_MFMA(2, pc, u3[n]);
              TB_MFMA(3, pc, l4[n]);                // (never two dependent MFMAs back to back)
            }
          }
        }
        hook(tc_int<0>(), b);
        __builtin_amdgcn_sched_barrier(0);
        if (TC_AHEAD == 2) { pc = pd; pd = pn; } else pc = pn;
      }
    }
    TP_ADD(tp_g0, tp);
    __builtin_amdgcn_sched_barrier(0);
    {
      tc_f16x8 u2[NT], u5[NT], m02[NT], m35[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) { u2[n] = TB_UH(2); u5[n] = TB_UH(5); m02[n] = tc_pair(TB_UL(0), TB_UL(2)); m35[n] = tc_pair(TB_UL(3), TB_UL(5)); }
      tc_f32x2 hc = TC_VH(pb0[0] + TC_WC * 64), hd = hc;    // Vh of x[i][j]: the first half of its word
      tc_f16x8 qc = TC_PIX(pb1[0] + TC_WC * 64), qd = qc;
      if (TC_AHEAD == 2 && BPW > 1) { hd = TC_VH(pb0[1] + TC_WC * 64); qd = TC_PIX(pb1[1] + TC_WC * 64); }
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        tc_f32x2 hn = TC_AHEAD == 2 ? hd : hc;
        tc_f16x8 qn = TC_AHEAD == 2 ? qd : qc;
        if (b + TC_AHEAD < BPW) { hn = TC_VH(pb0[b + TC_AHEAD] + TC_WC * 64); qn = TC_PIX(pb1[b + TC_AHEAD] + TC_WC * 64); }
        if (b < LB || last_ok) {
          const tc_f16x8 M = tc_pair_hq(hc, qc);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if (TC_ABL & 2) { asm volatile("" :: "v"(hc), "v"(qc), "v"(u2[n]), "v"(u5[n]), "v"(m02[n]), "v"(m35[n])); }
            else {
              TB_MFMA(0, qc, u2[n]); TB_MFMA(2, qc, u5[n]);                       // taps (0, 2), (1, 2)
              TB_MFMA(0, M, m02[n]); TB_MFMA(2, M, m35[n]);                       // Vh Ul of (0, 0) + (0, 2); (1, 0) + (1, 2)
            }
          }
        }
        hook(tc_int<1>(), b);
        __builtin_amdgcn_sched_barrier(0);
        if (TC_AHEAD == 2) { hc = hd; qc = qd; hd = hn; qd = qn; } else { hc = hn; qc = qn; }
      }
    }
    TP_ADD(tp_g1, tp);
    {
      tc_f16x8 u6[NT], u7[NT], u8[NT], m17[NT], m68[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        u6[n] = TB_UH(6); u7[n] = TB_UH(7); u8[n] = TB_UH(8);
        m17[n] = tc_pair(TB_UL(1), TB_UL(7)); m68[n] = tc_pair(TB_UL(6), TB_UL(8));
      }
      tc_f32x2 hc = TC_VH(pb0[0] + TC_WC * 64), hd = hc;
      tc_f16x8 qc = TC_PIX(pb0[0]), rc = TC_PIX(pb1[0]), qd = qc, rd = rc;
      if (TC_AHEAD == 2 && BPW > 1) { hd = TC_VH(pb0[1] + TC_WC * 64); qd = TC_PIX(pb0[1]); rd = TC_PIX(pb1[1]); }
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        tc_f32x2 hn = TC_AHEAD == 2 ? hd : hc;
        tc_f16x8 qn = TC_AHEAD == 2 ? qd : qc, rn = TC_AHEAD == 2 ? rd : rc;
        if (b + TC_AHEAD < BPW) { hn = TC_VH(pb0[b + TC_AHEAD] + TC_WC * 64); qn = TC_PIX(pb0[b + TC_AHEAD]); rn = TC_PIX(pb1[b + TC_AHEAD]); }
        if (b < LB || last_ok) {
          const tc_f16x8 M02 = tc_pair_hq(hc, qc), M23 = tc_pair(qc, rc);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if (TC_ABL & 2) { asm volatile("" :: "v"(hc), "v"(qc), "v"(rc), "v"(u6[n]), "v"(u7[n]), "v"(u8[n]), "v"(m17[n]), "v"(m68[n])); }
            else {
              TB_MFMA(0, qc, u6[n]); TB_MFMA(1, qc, u7[n]);                       // taps (2, 0), (2, 1)
              TB_MFMA(0, rc, u8[n]); TB_MFMA(1, M02, m17[n]);                     // tap (2, 2); Vh Ul of (0, 1) + (2, 1)
              TB_MFMA(0, M23, m68[n]);                                            // Vh Ul of (2, 0) + (2, 2)
            }
          }
        }
        hook(tc_int<2>(), b);
        __builtin_amdgcn_sched_barrier(0);
        if (TC_AHEAD == 2) { hc = hd; qc = qd; rc = rd; hd = hn; qd = qn; rd = rn; } else { hc = hn; qc = qn; rc = rn; }
      }
    }
    TP_ADD(tp_g2, tp);
    if (!LAST) {
      const unsigned delta = buf ? (unsigned)-BUFB : (unsigned)BUFB;
#pragma unroll
      for (int b = 0; b < BPW; ++b) { pb0[b] += delta; pb1[b] += delta; }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0): the requests of chunk c + 2 stay in flight across the barrier
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      TP_ADD(tp_bar, tp);
    }
  };

  // ---- prologue: chunk 0 requested, converted, written; chunk 1 (in_ch == 16: chunk 0 again, never converted) requested, in
  // the order of a chunk's requests, pinned -- the waits of the loop count the requests behind the one they need
  TP_NOW(tp_all);
  wstage_load(0);
#pragma unroll
  for (int s = 0; s < SI; ++s) {
    if (s == 0) stage_load_part(0, tc_int<0>(), tc_int<0>(), tc_int<4>());
    else stage_load_part(0, tc_int<SI - 1>(), tc_int<0>(), tc_int<4>());
  }
  stage_store_part(0, 0, tc_int<0>(), tc_int<0>(), tc_int<4>());
  if constexpr (SI > 1) stage_store_part(0, 0, tc_int<SI - 1>(), tc_int<0>(), tc_int<4>());
  wstage_store(0);
  __syncthreads();
  {
    const int c1 = NC > 1 ? 1 : 0;
    __builtin_amdgcn_sched_barrier(0);
    stage_load_part(c1, tc_int<0>(), tc_int<0>(), tc_int<4>());
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SI > 1) stage_load_part(c1, tc_int<SI - 1>(), tc_int<0>(), tc_int<4>());
    __builtin_amdgcn_sched_barrier(0);
    wstage_load(c1);
    __builtin_amdgcn_sched_barrier(0);
  }

#if TC_PROF
  TP_NOW(tp); tp_pro = tp - tp_all;
#endif
  for (int c = 0; c + 1 < NC; ++c) chunk(c, tc_int<0>());
  chunk(NC - 1, tc_int<1>());
#if TC_PROF
  const unsigned long long tp_loop_end = (unsigned long long)clock64();
#endif
  if (TC_ABL & 8) { if (acc[0][0][0][0] != 12345.f) return; }
  __syncthreads();                                  // every wave has read its last operands: the windows become the z tile

  // ---- epilogue, eight channels at a time
  float* Z = reinterpret_cast<float*>(Ls);
  float ymax = 0.f;
  float kf[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) kf[t] = Kf[t];
  // is the FIR an outer product kv x kh?  (kh = its first row, kv = its first column / its corner)
  bool sep = kf[0] != 0.f;
  float kv[4], kmax = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) kmax = fmaxf(kmax, fabsf(kf[t]));
#pragma unroll
  for (int a = 0; a < 4; ++a) kv[a] = sep ? kf[4 * a] / kf[0] : 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) sep = sep && fabsf(kf[t] - kv[t >> 2] * kf[t & 3]) <= 1e-6f * kmax;
  const int W2 = 2 * p.w;
  const int64_t hw2 = 4 * hw;
  // strip of this thread (separable FIR): four output columns, SR output rows, one channel per pass.  Its SR noise
  // vectors are the same in every pass: requested here, ahead of the z writes and their barrier (one L2 round trip
  // instead of one per output row)
  const int strip = tid & 127, seg = tid >> 7;
  const int s_og = strip & 15, s_ch = strip >> 4, s_oy0 = SR * seg;
  const int64_t s_pix = (int64_t)(2 * I0 + s_oy0) * W2 + 2 * J0 + 4 * s_og;
  tc_f32x4 nzr[SR];
#pragma unroll
  for (int oy = 0; oy < SR; ++oy) {
    nzr[oy] = tc_f32x4{0.f, 0.f, 0.f, 0.f};
    if (sep && p.noise && !(TC_ABL & 32))
      nzr[oy] = *reinterpret_cast<const tc_f32x4*>(p.noise + (int64_t)ib * hw2 + s_pix + (int64_t)oy * W2);
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if ((lt >> 3) == pass) {
      float* zc = Z + (lt & 7) * CHS;
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        if (wave + WAVES * b >= NBLK) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = 16 * (wave + WAVES * b) + 4 * lk + j;
          if (q < NPOS) {
            const int r = q / TC_PC, c = q - r * TC_PC;
            float* zp = zc + (2 * r) * TC_ZP + 2 * c + 3;
            zp[0] = acc[b][n][0][j];
            zp[1] = acc[b][n][1][j];
            zp[TC_ZP] = acc[b][n][2][j];
            zp[TC_ZP + 1] = acc[b][n][3][j];
          }
        }
      }
    }
    __syncthreads();
    if (sep) {
      // strip of four output columns, SR output rows: z rows oy0 + 1 .. oy0 + SR + 3, each read once
      const float* zb = Z + s_ch * CHS + (s_oy0 + 1) * TC_ZP + 4 * s_og + 4;
      const int cl = 16 * n + 8 * pass + s_ch;      // channel within the workgroup's OC
      const float sc = Sc[cl], bs = Bs[cl], post = Po[cl];
      float* yb = p.y + ((int64_t)ib * p.out_ch + OC * ot + cl) * hw2 + s_pix;
      tc_f32x4 hrow[4];                             // the last four horizontally filtered rows
      tc_f32x4 lo = *reinterpret_cast<const tc_f32x4*>(zb), hi = *reinterpret_cast<const tc_f32x4*>(zb + 4);
#pragma unroll
      for (int zr = 0; zr < SR + 3; ++zr) {
        tc_f32x4 lon = lo, hin = hi;                // the next z row, requested before this one is filtered
        if (zr + 1 < SR + 3) {
          lon = *reinterpret_cast<const tc_f32x4*>(zb + (zr + 1) * TC_ZP);
          hin = *reinterpret_cast<const tc_f32x4*>(zb + (zr + 1) * TC_ZP + 4);
        }
        tc_f32x4 hsum = lo * kf[0];
        if (!(TC_ABL & 64)) {
          hsum += tc_f32x4{lo[1], lo[2], lo[3], hi[0]} * kf[1];
          hsum += tc_f32x4{lo[2], lo[3], hi[0], hi[1]} * kf[2];
          hsum += tc_f32x4{lo[3], hi[0], hi[1], hi[2]} * kf[3];
        }
        hrow[zr & 3] = hsum;
        if (zr >= 3) {
          const int oy = zr - 3;                    // output row oy0 + oy: filtered rows zr - 3 .. zr
          tc_f32x4 res = hrow[(zr - 3) & 3] * kv[0];
          if (!(TC_ABL & 64)) {
            res += hrow[(zr - 2) & 3] * kv[1];
            res += hrow[(zr - 1) & 3] * kv[2];
            res += hrow[zr & 3] * kv[3];
          }
          const tc_f32x4 nz = nzr[oy] * noise_wg;
          tc_f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float u = res[q] * sc + nz[q] + bs;
            v[q] = fmaxf(u, u * slope) * post;
            ymax = fmaxf(ymax, fabsf(v[q]));
          }
          if (!(TC_ABL & 16)) *reinterpret_cast<tc_f32x4*>(yb + (int64_t)oy * W2) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
        lo = lon; hi = hin;
      }
    } else {
#pragma unroll 1
      for (int k = 0; k < 8 * 2 * TY * 16 / THREADS; ++k) {
        const int gid = tid + THREADS * k;          // 8 channels x 2 TY rows x 16 groups of four outputs
        const int og = gid & 15, oy = (gid >> 4) % (2 * TY), ch = gid / (32 * TY);
        const float* zb = Z + ch * CHS + (oy + 1) * TC_ZP + 4 * og + 4;
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
        const int cl = 16 * n + 8 * pass + ch;
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
        *reinterpret_cast<tc_f32x4*>(p.y + ((int64_t)ib * p.out_ch + OC * ot + cl) * hw2 + pix) = v;
      }
    }
    __syncthreads();
  }
  }
#if TC_PROF
  if (lane == 0 && (wave == 0 || wave == 7) && (blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x / 2 + 777)) {
    unsigned long long* o = tc_prof + (blockIdx.x == gridDim.x / 2 ? 0 : 32) + (wave == 0 ? 0 : 16);
    const unsigned long long now = (unsigned long long)clock64();
    o[0] = now - tp_all; o[1] = tp_pro; o[2] = tp_g0; o[3] = tp_s0; o[4] = tp_g1; o[5] = tp_g2; o[6] = tp_s1; o[7] = tp_bar;
    o[8] = now - tp_loop_end; o[9] = NC;
  }
#endif
  if (p.y_amax) rw_bound_store_wave(p.y_amax, rw_wave_max(ymax));
}

__global__ void __launch_bounds__(256, 2) tconv_blur_t8_kernel(const TconvProblem p) { tconv_body<8, 4, 1>(p); }
__global__ void __launch_bounds__(512, 1) tconv_blur_t16_kernel(const TconvProblem p) { tconv_body<16, 8, 1>(p); }
// 32 out-channels per workgroup (round 6): an 8 x 32 tile, eight waves of three position blocks x two out-channel blocks
__global__ void __launch_bounds__(512, 1) tconv_blur_n32_kernel(const TconvProblem p) { tconv_body<8, 8, 2>(p); }
// (four waves per SIMD -- <8, 8> twice per CU, <16, 16> once -- do not fit: 128 registers against 48 accumulators + ~125
// for the operands and the epilogue, 120 - 150 spilled)

// ---------------------------------------------------------------------------------------
// Third form: the waves of a workgroup SPECIALISE and the workgroup is PERSISTENT (rw_dconv.hip's dconv_ws_body, applied to
// the kernel above).  Measured on the forms above (profiles/r05g, layer 17): the window loads cost 2.2 ms of 7.8 (every
// half-chunk's HBM latency is exposed: four chunks, nothing else to run), the epilogue 3.9 (its noise loads alone 1.0), and
// nothing overlaps -- one wave does every job in turn.  Here ONE workgroup of eight waves owns a CU and walks every
// (grid)th tile:
//   waves 4..7 (one per SIMD) stage: one channel quad each, a lane's item = a 16-byte aligned run of four window columns x
//     four channels; piece s of chunk n + 1 is converted and written, then piece s of chunk n + 2 requested into the same
//     registers -- across tile boundaries and through the epilogue: loads are in flight all the time.  They carry the
//     chunk's 9 KB of weights (three 1-KB tap pieces per wave at most) and the tile's tables too;
//   waves 0..3 (one per SIMD) multiply: six position blocks each (96 accumulator registers), the tap groups of the kernel
//     above, pixel AND weight operands from LDS, and write the z tile (its own 46 KB of LDS: the window buffers belong to
//     the staging waves);
//   ALL eight run the blur: 512 strips of four output columns x four rows per pass of eight channels (noise requested a
//     chunk of MFMAs earlier), 16-byte stores.
// One raw s_barrier per chunk and two more per tile (z written -> blurred | the other eight channels written -> blurred).
// Tile: 8 x 32 positions (16 x 64 outputs) x 16 out-channels; in_ch >= 32 (the tables are double-buffered by tile parity).
// ---------------------------------------------------------------------------------------
#ifndef TC_NT
#define TC_NT 0           // pipelined form: 1 = the result leaves with non-temporal stores
#endif
#ifndef TC_PP_MQ
#define TC_PP_MQ 2        // pipelined form: of a tile's four blur passes, how many the multiplying waves take (0 .. 4)
#endif

// MW multiplying waves (4: the form of round 5, one per SIMD beside a staging wave; 8: two per SIMD beside a staging wave --
// twelve waves of at most 168 registers, which the kernel fits since the library is compiled without packed fp32 math: round 6)
template <int MW>
__device__ __forceinline__ void tconv_ws_body(const TconvProblem& p) {
  constexpr int TY = 8;
  constexpr int PR = TY + 2, NPOS = PR * TC_PC, NBLK = (NPOS + 15) / 16, BPW = (NBLK + MW - 1) / MW;
  constexpr int WR = TY + 3, NPIX = WR * TC_WC, BUFB = NPIX * 64;
  constexpr int IPR = TC_TX / 4 + 2, NITEM = WR * IPR, SI = (NITEM + 63) / 64;   // items: window columns 4 j - 2 .. 4 j + 1
  // z tile: rows 0 .. 2 PR - 1 + two spare rows (the last position block runs 12 positions past the window: written, never
  // read -- no per-position test in the write loop); position column c owns z columns 2 c + 4, 2 c + 5 (8-byte aligned
  // pairs; the blur reads the aligned 16-byte pieces 4 og + 4 .. 4 og + 11 and uses seven of them)
  constexpr int ZR = 2 * PR + 2, CHS = ZR * TC_ZP + 4;
  constexpr int SR = 4, CT = 512;                   // output rows of a strip; threads of the blur (all)
  static_assert(SI == 2 && BPW == 24 / MW && (MW == 4 || MW == 8) && SR * (CT / 128) == 2 * TY, "piece / block / strip counts the code below is written for");
  constexpr bool STAGERS_BLUR = MW == 4;            // (MW == 8: the 512 threads of the multiplying waves are the blur's)
  __shared__ __attribute__((aligned(16))) unsigned char Ls[2 * BUFB];
  __shared__ __attribute__((aligned(16))) unsigned char Wl[2 * TC_WCH];
  __shared__ __attribute__((aligned(16))) float Zs[8 * CHS];
  __shared__ float Sc[2][16], Bs[2][16], Po[2][16], Kf[16], Ks[12];   // Ks: kh[4], kv[4], [8] = 1 when the FIR is kv x kh

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t hw = (int64_t)p.h * p.w;
  const int NC = p.in_ch >> 4, T = 9 * NC;
  const float gain = p.act ? 1.4142135623730951f : 1.f, slope = p.act ? 0.2f : 1.f;

  // this workgroup's run: tile k * grid + bx (ot fastest, then x, y, image); those running at the same time are neighbours
  const int64_t total = (int64_t)p.batch * p.tiles_y * p.tiles_x * p.o_tiles;
  const int bx = tc_xcd_remap(blockIdx.x, gridDim.x);
  const int count = (int)((total - bx + gridDim.x - 1) / gridDim.x);
  if (count <= 0) {                                 // (every wave of the launch owns a slot of the bound: rw_common.h)
    if (p.y_amax) rw_bound_store_wave(p.y_amax, 0.f);
    return;
  }
  const int N = count * NC;                         // chunks of the run
  // tile coordinates: (ot, tx, ty, ib) of tile bx once, then + the grid's digits per step (64-bit divisions per tile cost the
  // staging waves 1.7 k cycles per chunk on layer 17: profiles/r05p)
  auto digits = [&](unsigned tile, int& ot, int& tx, int& ty, int& ib) __attribute__((always_inline)) {
    ot = (int)(tile % (unsigned)p.o_tiles);
    unsigned pg = tile / (unsigned)p.o_tiles;
    tx = (int)(pg % (unsigned)p.tiles_x); pg /= (unsigned)p.tiles_x;
    ty = (int)(pg % (unsigned)p.tiles_y);
    ib = (int)(pg / (unsigned)p.tiles_y);
  };
  int g_ot, g_tx, g_ty, g_ib;
  digits(gridDim.x, g_ot, g_tx, g_ty, g_ib);
  auto advance = [&](int& ot, int& tx, int& ty, int& ib) __attribute__((always_inline)) {
    ot += g_ot;
    int carry = ot >= p.o_tiles ? 1 : 0;
    ot -= carry ? p.o_tiles : 0;
    tx += g_tx + carry;
    carry = tx >= p.tiles_x ? 1 : 0;
    tx -= carry ? p.tiles_x : 0;
    ty += g_ty + carry;
    carry = ty >= p.tiles_y ? 1 : 0;
    ty -= carry ? p.tiles_y : 0;
    ib += g_ib + carry;
  };
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): vector loads and stores stay in flight across the barrier
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  if (tid < 16) {
    const int a = tid >> 2, c = tid & 3;
    Kf[tid] = p.k4[(3 - a) * 4 + (3 - c)];          // flipped, as upfirdn2d applies it (read after the first barrier)
  }

  // ---- the blur of one pass (every thread): strip = four output columns x SR rows of one channel
  const float noise_wg = p.noise ? p.noise_w[0] * gain : 0.f;
  const int W2 = 2 * p.w;
  const int64_t hw2 = 4 * hw;
  const int strip = tid & 127, seg = tid >> 7;
  const int s_og = strip & 15, s_ch = strip >> 4, s_oy0 = SR * seg;
  tc_f32x4 nzr[SR];
#pragma unroll
  for (int oy = 0; oy < SR; ++oy) nzr[oy] = tc_f32x4{0.f, 0.f, 0.f, 0.f};
  float ymax = 0.f;
  // is the FIR an outer product kv x kh?  (tconv_body; kh, kv and the answer live in LDS: every register counts beside the
  // multiplying waves' accumulators and operands.)  Thread 0 decides, before the first barrier.
  auto fir_setup = [&]() __attribute__((always_inline)) {
    if (tid == 0) {
      float kf[16], kvv[4], kmax = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) { kf[t] = p.k4[(3 - (t >> 2)) * 4 + (3 - (t & 3))]; kmax = fmaxf(kmax, fabsf(kf[t])); }
      bool ok = kf[0] != 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) { kvv[a] = ok ? kf[4 * a] / kf[0] : 0.f; Ks[a] = kf[a]; Ks[4 + a] = kvv[a]; }
#pragma unroll
      for (int t = 0; t < 16; ++t) ok = ok && fabsf(kf[t] - kvv[t >> 2] * kf[t & 3]) <= 1e-6f * kmax;
      Ks[8] = ok ? 1.f : 0.f;
    }
  };
  fir_setup();
  // the strip's noise (the same in both passes), requested a chunk of MFMAs before the epilogue
  auto noise_request = [&](int ty, int tx, int ib) __attribute__((always_inline)) {
    if (Ks[8] != 0.f && p.noise) {
      const int64_t s_pix = (int64_t)(2 * ty * TY + s_oy0) * W2 + 2 * tx * TC_TX + 4 * s_og;
#pragma unroll
      for (int oy = 0; oy < SR; ++oy)
        nzr[oy] = *reinterpret_cast<const tc_f32x4*>(p.noise + (int64_t)ib * hw2 + s_pix + (int64_t)oy * W2);
    }
  };
  auto blur = [&](int pass, int par, int ot, int tx, int ty, int ib) __attribute__((always_inline)) {
    if (Ks[8] != 0.f) {
      const float kh[4] = {Ks[0], Ks[1], Ks[2], Ks[3]}, kv[4] = {Ks[4], Ks[5], Ks[6], Ks[7]};
      const int cl = 8 * pass + s_ch;               // channel within the workgroup's 16
      const int64_t s_pix = (int64_t)(2 * ty * TY + s_oy0) * W2 + 2 * tx * TC_TX + 4 * s_og;
      const float* zb = Zs + s_ch * CHS + (s_oy0 + 1) * TC_ZP + 4 * s_og + 4;
      const float sc = Sc[par][cl], bs = Bs[par][cl], post = Po[par][cl];
      float* yb = p.y + ((int64_t)ib * p.out_ch + 16 * ot + cl) * hw2 + s_pix;
      tc_f32x4 hrow[4];                             // the last four horizontally filtered rows
      tc_f32x4 lo = *reinterpret_cast<const tc_f32x4*>(zb), hi = *reinterpret_cast<const tc_f32x4*>(zb + 4);
#pragma unroll
      for (int zr = 0; zr < SR + 3; ++zr) {
        tc_f32x4 lon = lo, hin = hi;                // the next z row, requested before this one is filtered
        if (zr + 1 < SR + 3) {
          lon = *reinterpret_cast<const tc_f32x4*>(zb + (zr + 1) * TC_ZP);
          hin = *reinterpret_cast<const tc_f32x4*>(zb + (zr + 1) * TC_ZP + 4);
        }
        tc_f32x4 hsum = tc_f32x4{lo[1], lo[2], lo[3], hi[0]} * kh[0];
        hsum += tc_f32x4{lo[2], lo[3], hi[0], hi[1]} * kh[1];
        hsum += tc_f32x4{lo[3], hi[0], hi[1], hi[2]} * kh[2];
        hsum += hi * kh[3];
        hrow[zr & 3] = hsum;
        if (zr >= 3) {
          const int oy = zr - 3;                    // output row oy0 + oy: filtered rows zr - 3 .. zr
          tc_f32x4 res = hrow[(zr - 3) & 3] * kv[0];
          res += hrow[(zr - 2) & 3] * kv[1];
          res += hrow[(zr - 1) & 3] * kv[2];
          res += hrow[zr & 3] * kv[3];
          const tc_f32x4 nz = nzr[oy] * noise_wg;
          tc_f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float u = res[q] * sc + nz[q] + bs;
            v[q] = fmaxf(u, u * slope) * post;
            ymax = fmaxf(ymax, fabsf(v[q]));
          }
          *reinterpret_cast<tc_f32x4*>(yb + (int64_t)oy * W2) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
        lo = lon; hi = hin;
      }
    } else {
#pragma unroll 1
      for (int k = 0; k < 8 * 2 * TY * 16 / CT; ++k) {
        const int gid = tid + CT * k;               // 8 channels x 2 TY rows x 16 groups of four outputs
        const int og = gid & 15, oy = (gid >> 4) % (2 * TY), ch = gid / (32 * TY);
        const float* zb = Zs + ch * CHS + (oy + 1) * TC_ZP + 4 * og + 4;
        float res[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const tc_f32x4 lo = *reinterpret_cast<const tc_f32x4*>(zb + a * TC_ZP);
          const tc_f32x4 hi = *reinterpret_cast<const tc_f32x4*>(zb + a * TC_ZP + 4);
          const float rowv[7] = {lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) res[q] += rowv[q + cc] * Kf[a * 4 + cc];
        }
        const int cg = 8 * pass + ch;
        const int64_t pix = (int64_t)(2 * ty * TY + oy) * W2 + 2 * tx * TC_TX + 4 * og;
        tc_f32x4 nz = {0.f, 0.f, 0.f, 0.f};
        if (p.noise) nz = *reinterpret_cast<const tc_f32x4*>(p.noise + (int64_t)ib * hw2 + pix) * noise_wg;
        const float sc = Sc[par][cg], bs = Bs[par][cg], post = Po[par][cg];
        tc_f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float u = res[q] * sc + nz[q] + bs;
          v[q] = fmaxf(u, u * slope) * post;
          ymax = fmaxf(ymax, fabsf(v[q]));
        }
        *reinterpret_cast<tc_f32x4*>(p.y + ((int64_t)ib * p.out_ch + 16 * ot + cg) * hw2 + pix) = v;
      }
    }
  };

  if (wave >= MW) {
    // =========================== staging waves: channel quad g of every chunk ===========================
#if TC_STAGE_PRIO
    __builtin_amdgcn_s_setprio(TC_STAGE_PRIO);      // (static: the second-dispatched half loses the VALU arbitration by age otherwise)
#endif
    const int g = wave - MW, lid = g * 64 + lane;
    const float xam = rw_bound_load(p.x_amax);
    const int hw4 = (int)hw * 4;
    int l_pos = 0, l_c = 0, l_ib = -1, l_ot, l_tx, l_ty, l_ibn;       // (l_ot, l_tx, l_ty, l_ibn): the tile being requested
    bool l_past = false;
    digits((unsigned)bx, l_ot, l_tx, l_ty, l_ibn);
    float in_scale = 1.f, out_scale = 1.f;
    int xoff[SI];
    __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0, 0x00020000);
    // Two chunks are in flight at any time (registers: slot n & 1 holds chunk n's window pieces, weight pieces and
    // scalars): chunk n + 3 is requested when chunk n + 1 has been written -- two intervals between a request and its use
    // (one left layer 17 waiting: 1.3 of 6.5 ms disappeared with the window loads, profiles/r05u)
    struct Flight {
      tc_f32x4 raw[SI][4];                          // [piece][channel]: four pixels
      float psv[4];
      float demod, bias, post, oscale, iscale;
      bool first;
      int par;
    };
    Flight fl[2];
    // (the weights come from the L2 and stay ONE interval ahead: tap pieces g, g + 4 and (g == 0) 8 of the chunk after the
    // one being written; w_next = where the chunk requested last finds its weights)
    tc_f32x4 wraw[3];
    float sv[4];
    int l_s0 = 0;
    const unsigned char* l_wsrc = p.wp;
    const unsigned char* w_next = p.wp;
    auto setup = [&](auto tag) __attribute__((always_inline)) {   // the chunk to REQUEST: (l_pos, l_c); loads only, none used here
      Flight& F = fl[decltype(tag)::value];
      F.first = l_c == 0 && !l_past;                              // (past the run: the last chunk again, never read)
      if (F.first) {
        const int ib = l_ibn;
        const int i0 = l_ty * TY, j0 = l_tx * TC_TX;
        if (ib != l_ib) {
          l_ib = ib;
          float smax = p.style ? 0.f : 1.f;
          if (p.style)
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
          const int iy = i0 - 2 + r, ix = j0 - 4 + 4 * j;            // the item lies inside the row or outside it as a whole
          const bool ok = it < NITEM && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
          xoff[s] = ok ? (iy * p.w + ix) * 4 : 0x7ffffff0;
        }
        F.par = l_pos & 1;
        F.oscale = out_scale;
        if (lid < 16) {
          const int o = 16 * l_ot + lid;
          F.demod = p.demod ? p.demod[(int64_t)l_ib * p.out_ch + o] : 1.f;
          F.bias = p.act ? p.bias[o] : 0.f;
          F.post = p.post ? p.post[(int64_t)l_ib * p.out_ch + o] : 1.f;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) F.psv[k] = p.style ? p.style[(int64_t)l_ib * p.in_ch + 16 * l_c + 4 * g + k] : 1.f;
      F.iscale = in_scale;
      l_s0 = (16 * l_c + 4 * g) * hw4;
      l_wsrc = w_next;                              // the weights of the chunk set up ONE call ago
      w_next = p.wp + ((int64_t)l_ot * T + 9 * l_c) * 1024 + lane * 16;
      if (++l_c == NC) {
        if (l_pos + 1 < count) { l_c = 0; ++l_pos; advance(l_ot, l_tx, l_ty, l_ibn); }
        else { l_c = NC - 1; l_past = true; }
      }
    };
    auto request_s = [&](auto tag, int s, bool with_w = true) __attribute__((always_inline)) {
      Flight& F = fl[decltype(tag)::value];
      // (the weights FIRST, all with piece 0: loads return in order -- the wait for them, an interval later, must not reach
      // past this interval's window requests)
      if (with_w && s == 0) {
        wraw[0] = *reinterpret_cast<const tc_f32x4*>(l_wsrc + g * 1024);
        wraw[1] = *reinterpret_cast<const tc_f32x4*>(l_wsrc + (g + 4) * 1024);
        if (g == 0) wraw[2] = *reinterpret_cast<const tc_f32x4*>(l_wsrc + 8 * 1024);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (TC_ABL & 1) F.raw[s][k] = tc_f32x4{1.f, 1.f, 1.f, 1.f};     // (timing ablation: no window loads; results wrong)
        else F.raw[s][k] = __builtin_bit_cast(tc_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xoff[s], l_s0 + k * hw4, 0));
      }
    };
    // what setup() requested beside the pixels -> registers / LDS (the first wait of an interval)
    auto tables = [&](auto tag) __attribute__((always_inline)) {
      Flight& F = fl[decltype(tag)::value];
#pragma unroll
      for (int k = 0; k < 4; ++k) sv[k] = F.psv[k] * F.iscale;
      if (F.first && lid < 16) {
        Sc[F.par][lid] = F.demod * p.w_scale * F.oscale * gain;
        Bs[F.par][lid] = F.bias * gain;
        Po[F.par][lid] = F.post;
      }
    };
    auto deliver_s = [&](auto tag, int buf, int s) __attribute__((always_inline)) {
      Flight& F = fl[decltype(tag)::value];
      unsigned char* dst = Ls + buf * BUFB;
      unsigned char* wdst = Wl + buf * TC_WCH + lane * 16;
      if (s == 0) {
        *reinterpret_cast<tc_f32x4*>(wdst + g * 1024) = wraw[0];
        *reinterpret_cast<tc_f32x4*>(wdst + (g + 4) * 1024) = wraw[1];
        if (g == 0) *reinterpret_cast<tc_f32x4*>(wdst + 8 * 1024) = wraw[2];
      }
      const int it = 64 * s + lane;
      const int r = it / IPR, j = it - r * IPR;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v0 = F.raw[s][0][e] * sv[0], v1 = F.raw[s][1][e] * sv[1], v2 = F.raw[s][2][e] * sv[2],
                    v3 = F.raw[s][3][e] * sv[3];
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
        const int cc = 4 * j - 2 + e;                // window column of this pixel
        if (it < NITEM && cc >= 0 && cc < TC_WC)
          *reinterpret_cast<tc_f16x8*>(dst + (r * TC_WC + cc) * 64 + ((g ^ tc_swz(cc)) << 4)) = word;
      }
    };

    // chunks 0 and 1 requested (chunk 0's weights with chunk 1's window); chunk 0 written; chunk 2 requested
    setup(tc_int<0>());
#pragma unroll
    for (int s = 0; s < SI; ++s) request_s(tc_int<0>(), s, false);
    setup(tc_int<1>());
#pragma unroll
    for (int s = 0; s < SI; ++s) request_s(tc_int<1>(), s);
    tables(tc_int<0>());
    __builtin_amdgcn_sched_barrier(0);
    setup(tc_int<0>());
#pragma unroll
    for (int s = 0; s < SI; ++s) {
      deliver_s(tc_int<0>(), 0, s); __builtin_amdgcn_sched_barrier(0);
      request_s(tc_int<0>(), s); __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
    int cn = 0, e_pos = 0;                          // the tile the multiplying waves are on (its epilogue is shared)
    int e_ot, e_tx, e_ty, e_ib;
    digits((unsigned)bx, e_ot, e_tx, e_ty, e_ib);
    TP_DECL(tp = 0, tp_all = 0, tp_setup = 0, tp_del = 0, tp_bar = 0, tp_blur = 0, tp_ebar = 0, tp_tab = 0);
    TP_NOW(tp); TP_NOW(tp_all);
    // interval n: chunk n + 1 (slot (n + 1) & 1, in flight for two intervals; its weights for one) -> LDS piece by piece, the
    // window of chunk n + 3 and the weights of chunk n + 2 requested behind it into the same registers (past the run:
    // harmless repeats)
    auto interval = [&](int n, auto tag) __attribute__((always_inline)) {
      tables(tag);
      TP_ADD(tp_tab, tp);
      __builtin_amdgcn_sched_barrier(0);
      // (the strip's noise BEFORE this interval's window requests: loads return in order, and the wait for the noise at the
      // blur must leave the younger window loads in flight)
      if (STAGERS_BLUR && cn == NC - 1) noise_request(e_ty, e_tx, e_ib);
      __builtin_amdgcn_sched_barrier(0);
      setup(tag);
      TP_ADD(tp_setup, tp);
#pragma unroll
      for (int s = 0; s < SI; ++s) {
        deliver_s(tag, (n + 1) & 1, s); __builtin_amdgcn_sched_barrier(0);
        request_s(tag, s); __builtin_amdgcn_sched_barrier(0);
      }
      TP_ADD(tp_del, tp);
      lds_barrier();
      TP_ADD(tp_bar, tp);
      if (++cn == NC) {                             // the tile's epilogue (the loads stay in flight)
        cn = 0;
        const int par = e_pos & 1;
        if (STAGERS_BLUR) blur(0, par, e_ot, e_tx, e_ty, e_ib);
        TP_ADD(tp_blur, tp);
        lds_barrier();                              // z is free: the multiplying waves write the other eight channels
        lds_barrier();
        TP_ADD(tp_ebar, tp);
        if (STAGERS_BLUR) blur(1, par, e_ot, e_tx, e_ty, e_ib);
        TP_ADD(tp_blur, tp);
        ++e_pos;
        advance(e_ot, e_tx, e_ty, e_ib);
      }
    };
    int n = 0;
    for (; n + 1 < N; n += 2) { interval(n, tc_int<1>()); interval(n + 1, tc_int<0>()); }
    if (n < N) interval(n, tc_int<1>());
#if TC_PROF
    if (wave == MW && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {
      unsigned long long* o = tc_prof + (blockIdx.x == 0 ? 0 : 32);
      o[8] = (unsigned long long)clock64() - tp_all; o[9] = tp_setup; o[10] = tp_del; o[11] = tp_bar; o[12] = tp_blur; o[13] = tp_ebar; o[14] = N;
      o[16] = tp_tab;
    }
#endif
    if (p.y_amax) rw_bound_store_wave(p.y_amax, rw_wave_max(ymax));
    return;
  }

  // =========================== multiplying waves ===========================
  const int lk = lane >> 4, lt = lane & 15;
  unsigned pb0[BPW], pb1[BPW];                     // operand addresses as in tconv_body
#pragma unroll
  for (int b = 0; b < BPW; ++b) {
    int q = 16 * (wave + MW * b) + lt;
    q = q < NPOS ? q : NPOS - 1;
    const int r = q / TC_PC, c = q - r * TC_PC;
    pb0[b] = (unsigned)(((r + 1) * TC_WC + c + 1) * 64 + ((lk ^ tc_swz(c + 1)) << 4));
    pb1[b] = (unsigned)(((r + 1) * TC_WC + c) * 64 + ((lk ^ tc_swz(c)) << 4));
  }
  tc_f32x4 acc[BPW][4];
#pragma unroll
  for (int b = 0; b < BPW; ++b)
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) acc[b][ph] = tc_f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int LB = BPW - 1;
  const bool last_ok = wave + MW * LB < NBLK;       // wave-uniform: only the last block of a wave can be missing
  // One chunk: the three tap groups of tconv_body's chunk() (at most five weight operands live), the pixel operands of a
  // block requested TWO blocks ahead -- a group gives a block four or five MFMAs (90 - 110 cycles with one multiplying
  // wave per SIMD), less than an LDS round trip beside the staging waves' traffic: 36 cycles per MFMA with the operands one
  // block ahead (profiles/r05p).  (All fourteen weight operands live and fourteen MFMAs per block would cover it too: 35
  // registers more than there are.)
  auto mma = [&](const unsigned char* lb, const unsigned char* wb) __attribute__((always_inline)) {
    {
      const tc_f16x8 u0 = TC_UH(0), u1 = TC_UH(1), u3 = TC_UH(3), u4 = TC_UH(4), l4 = tc_expand(TC_UL(4));
      tc_f16x8 pc = TC_PIX(pb0[0]), pd = TC_PIX(pb0[1]);
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        tc_f16x8 pn = pd;
        if (b + 2 < BPW) pn = TC_PIX(pb0[b + 2]);
        if (b < LB || last_ok) {
          TC_MFMA(3, pc, u4); TC_MFMA(0, pc, u0); TC_MFMA(1, pc, u1); TC_MFMA(2, pc, u3);
          TC_MFMA(3, pc, l4);
        }
        __builtin_amdgcn_sched_barrier(0);
        pc = pd; pd = pn;
      }
    }
    {
      const tc_f16x8 u2 = TC_UH(2), u5 = TC_UH(5), m02 = tc_pair(TC_UL(0), TC_UL(2)), m35 = tc_pair(TC_UL(3), TC_UL(5));
      tc_f32x2 hc = TC_VH(pb0[0]), hd = TC_VH(pb0[1]);
      tc_f16x8 qc = TC_PIX(pb1[0]), qd = TC_PIX(pb1[1]);
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        tc_f32x2 hn = hd;
        tc_f16x8 qn = qd;
        if (b + 2 < BPW) { hn = TC_VH(pb0[b + 2]); qn = TC_PIX(pb1[b + 2]); }
        if (b < LB || last_ok) {
          const tc_f16x8 M = tc_pair_hq(hc, qc);
          TC_MFMA(0, qc, u2); TC_MFMA(2, qc, u5);
          TC_MFMA(0, M, m02); TC_MFMA(2, M, m35);
        }
        __builtin_amdgcn_sched_barrier(0);
        hc = hd; qc = qd; hd = hn; qd = qn;
      }
    }
    {
      const tc_f16x8 u6 = TC_UH(6), u7 = TC_UH(7), u8 = TC_UH(8), m17 = tc_pair(TC_UL(1), TC_UL(7)),
                     m68 = tc_pair(TC_UL(6), TC_UL(8));
      tc_f32x2 hc = TC_VH(pb0[0]), hd = TC_VH(pb0[1]);
      tc_f16x8 qc = TC_PIX(pb0[0] - TC_WC * 64), rc = TC_PIX(pb1[0] - TC_WC * 64);
      tc_f16x8 qd = TC_PIX(pb0[1] - TC_WC * 64), rd = TC_PIX(pb1[1] - TC_WC * 64);
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        tc_f32x2 hn = hd;
        tc_f16x8 qn = qd, rn = rd;
        if (b + 2 < BPW) { hn = TC_VH(pb0[b + 2]); qn = TC_PIX(pb0[b + 2] - TC_WC * 64); rn = TC_PIX(pb1[b + 2] - TC_WC * 64); }
        if (b < LB || last_ok) {
          const tc_f16x8 M02 = tc_pair_hq(hc, qc), M23 = tc_pair(qc, rc);
          TC_MFMA(0, qc, u6); TC_MFMA(1, qc, u7);
          TC_MFMA(0, rc, u8); TC_MFMA(1, M02, m17);
          TC_MFMA(0, M23, m68);
        }
        __builtin_amdgcn_sched_barrier(0);
        hc = hd; qc = qd; rc = rd; hd = hn; qd = qn; rd = rn;
      }
    }
  };
  // eight of the sixteen channels -> the z tile (lanes lt >> 3 == pass hold them).  A lane's four positions of a block are
  // consecutive: one division per block, then + 2 floats per position and + 76 (a z row pair is 144, a position row 68)
  // where the run crosses the end of a position row
  auto zwrite = [&](int pass) __attribute__((always_inline)) {
    if ((lt >> 3) == pass) {
      // (the addresses are loop invariants: left to itself the compiler keeps all of them in registers for the whole run;
      // an opaque zero keeps their few instructions inside the loop)
      int opaque;
      asm volatile("v_mov_b32 %0, 0" : "=v"(opaque));
      float* zc = Zs + (lt & 7) * CHS + 4;
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        if (wave + MW * b >= NBLK) continue;
        const int q0 = 16 * (wave + MW * b) + 4 * lk + opaque;         // < 352: the spare rows take what lies past the window
        const int r0 = (q0 * 1928) >> 16, c0 = q0 - r0 * TC_PC;        // q0 / 34 for q0 < 400
        float* z0 = zc + (2 * r0) * TC_ZP + 2 * c0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float* zp = z0 + 2 * j + (c0 + j >= TC_PC ? 2 * TC_ZP - 2 * TC_PC : 0);
          // (four dword stores: the compiler pairs them into ds_write2_b32 -- two separate source registers each; as 8-byte
          // vectors every pair cost two moves to bring accumulators of different blocks side by side)
          zp[0] = acc[b][0][j];
          zp[1] = acc[b][1][j];
          zp[TC_ZP] = acc[b][2][j];
          zp[TC_ZP + 1] = acc[b][3][j];
        }
      }
    }
  };

  int pos = 0, c = 0;
  int ot, tx, ty, ib;
  digits((unsigned)bx, ot, tx, ty, ib);
  lds_barrier();                                    // chunk 0 and the FIR are in LDS
  TP_DECL(tp = 0, tp_all = 0, tp_mma = 0, tp_bar = 0, tp_zw = 0, tp_blur = 0, tp_ebar = 0);
  TP_NOW(tp); TP_NOW(tp_all);
  for (int n = 0; n < N; ++n) {
    const unsigned char* lb = Ls + (n & 1) * BUFB;
    const unsigned char* wb = Wl + (n & 1) * TC_WCH + lane * 8;
    mma(lb, wb);
    TP_ADD(tp_mma, tp);
    if (c + 1 < NC) { ++c; lds_barrier(); TP_ADD(tp_bar, tp); continue; }
    // ---- epilogue of the tile, eight channels at a time
    c = 0;
    const int par = pos & 1;
    noise_request(ty, tx, ib);                      // (behind the MFMAs: sixteen registers they need; the z write and a barrier cover it)
    zwrite(0);
    TP_ADD(tp_zw, tp);
    lds_barrier();                                  // (the barrier of the tile's last chunk)
    TP_ADD(tp_bar, tp);
    blur(0, par, ot, tx, ty, ib);
    TP_ADD(tp_blur, tp);
    lds_barrier();                                  // every strip of the first eight channels is read: z is free again
    TP_ADD(tp_ebar, tp);
    zwrite(1);
    TP_ADD(tp_zw, tp);
    lds_barrier();
    TP_ADD(tp_ebar, tp);
    blur(1, par, ot, tx, ty, ib);
    TP_ADD(tp_blur, tp);
#pragma unroll
    for (int b = 0; b < BPW; ++b)
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) acc[b][ph] = tc_f32x4{0.f, 0.f, 0.f, 0.f};
    ++pos;
    advance(ot, tx, ty, ib);
  }
#if TC_PROF
  if (wave == 0 && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {
    unsigned long long* o = tc_prof + (blockIdx.x == 0 ? 0 : 32);
    o[0] = (unsigned long long)clock64() - tp_all; o[1] = tp_mma; o[2] = tp_bar; o[3] = tp_zw; o[4] = tp_blur; o[5] = tp_ebar; o[6] = N;
  }
#endif
  if (p.y_amax) rw_bound_store_wave(p.y_amax, rw_wave_max(ymax));
}
__global__ void __launch_bounds__(512, 2) tconv_blur_ws_kernel(const TconvProblem p) { tconv_ws_body<4>(p); }
__global__ void __launch_bounds__(768, 1) tconv_blur_ws12_kernel(const TconvProblem p) { tconv_ws_body<8>(p); }

// ---------------------------------------------------------------------------------------
// Fourth form (round 6): the persistent kernel above with its two halves PIPELINED across tiles.  There the phases of a
// tile take turns -- all eight waves wait while four multiply, the matrix pipe idles while all eight blur (27 - 29 k cycles
// per tile on layer 17: multiplies 10 k, z write 4 k in two half-lane passes, blur 8 k of vector issue per SIMD, waits).
// Here the z tile holds all SIXTEEN channels (20 rows: the last position block tests its positions instead of writing
// into spare rows -- 92 KB, 161 KB of LDS with the windows and the weights) and
//   waves 0..3 multiply tile t, then write ITS z tile in one pass with every lane active (one barrier before: z is free;
//     one behind: z is ready), and go on to tile t + 1;
//   waves 4..7 stage as before AND blur tile t - 1 meanwhile: four passes of four channels (256 strips of four columns x
//     four rows each), spread over the chunk intervals of tile t, the noise of the tile requested at the switch.
// The vector work of a tile (staging conversion + blur) runs beside the matrix work of the next one; NC + 1 barriers per
// tile.  The staging waves' LDS addresses are computed once (the window's geometry does not change from chunk to chunk).
// Tables (demodulation x scales, bias, post scale) are kept for FOUR tiles (tile index & 3): the tile being requested is two
// ahead of the one being blurred.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 2) tconv_blur_pp_kernel(const TconvProblem p) {
  constexpr int TY = 8, MW = 4;
  constexpr int PR = TY + 2, NPOS = PR * TC_PC, NBLK = (NPOS + 15) / 16, BPW = (NBLK + MW - 1) / MW;
  constexpr int WR = TY + 3, NPIX = WR * TC_WC, BUFB = NPIX * 64;
  constexpr int IPR = TC_TX / 4 + 2, NITEM = WR * IPR, SI = (NITEM + 63) / 64;   // items: window columns 4 j - 2 .. 4 j + 1
  constexpr int ZR = 2 * PR, CHS = ZR * TC_ZP + 4;  // z rows 0 .. 2 PR - 1; position column c owns z columns 2 c + 4, 2 c + 5
  constexpr int SR = 4;                             // output rows of a strip
  static_assert(SI == 2 && BPW == 6 && 4 * SR == 2 * TY, "piece / block / strip counts the code below is written for");
  static_assert(2 * BUFB + 2 * TC_WCH + 16 * CHS * 4 + 1024 <= 163840, "LDS of a compute unit");
  __shared__ __attribute__((aligned(16))) unsigned char Ls[2 * BUFB];
  __shared__ __attribute__((aligned(16))) unsigned char Wl[2 * TC_WCH];
  __shared__ __attribute__((aligned(16))) float Zs[16 * CHS];
  __shared__ float Sc[4][16], Bs[4][16], Po[4][16], Kf[16], Ks[12];   // Ks: kh[4], kv[4], [8] = 1 when the FIR is kv x kh

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t hw = (int64_t)p.h * p.w;
  const int NC = p.in_ch >> 4, T = 9 * NC;
  const float gain = p.act ? 1.4142135623730951f : 1.f, slope = p.act ? 0.2f : 1.f;

  const int64_t total = (int64_t)p.batch * p.tiles_y * p.tiles_x * p.o_tiles;
  const int bx = tc_xcd_remap(blockIdx.x, gridDim.x);
  const int count = (int)((total - bx + gridDim.x - 1) / gridDim.x);
  if (count <= 0) {                                 // (every wave of the launch owns a slot of the bound: rw_common.h)
    if (p.y_amax) rw_bound_store_wave(p.y_amax, 0.f);
    return;
  }
  const int N = count * NC;                         // chunks of the run
  auto digits = [&](unsigned tile, int& ot, int& tx, int& ty, int& ib) __attribute__((always_inline)) {
    ot = (int)(tile % (unsigned)p.o_tiles);
    unsigned pg = tile / (unsigned)p.o_tiles;
    tx = (int)(pg % (unsigned)p.tiles_x); pg /= (unsigned)p.tiles_x;
    ty = (int)(pg % (unsigned)p.tiles_y);
    ib = (int)(pg / (unsigned)p.tiles_y);
  };
  int g_ot, g_tx, g_ty, g_ib;
  digits(gridDim.x, g_ot, g_tx, g_ty, g_ib);
  auto advance = [&](int& ot, int& tx, int& ty, int& ib) __attribute__((always_inline)) {
    ot += g_ot;
    int carry = ot >= p.o_tiles ? 1 : 0;
    ot -= carry ? p.o_tiles : 0;
    tx += g_tx + carry;
    carry = tx >= p.tiles_x ? 1 : 0;
    tx -= carry ? p.tiles_x : 0;
    ty += g_ty + carry;
    carry = ty >= p.tiles_y ? 1 : 0;
    ty -= carry ? p.tiles_y : 0;
    ib += g_ib + carry;
  };
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): vector loads and stores stay in flight across the barrier
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  if (tid < 16) {
    const int a = tid >> 2, c = tid & 3;
    Kf[tid] = p.k4[(3 - a) * 4 + (3 - c)];          // flipped, as upfirdn2d applies it (read after the first barrier)
  }
  if (tid == 0) {                                   // is the FIR an outer product kv x kh?  (tconv_body)
    float kf[16], kvv[4], kmax = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) { kf[t] = p.k4[(3 - (t >> 2)) * 4 + (3 - (t & 3))]; kmax = fmaxf(kmax, fabsf(kf[t])); }
    bool ok = kf[0] != 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) { kvv[a] = ok ? kf[4 * a] / kf[0] : 0.f; Ks[a] = kf[a]; Ks[4 + a] = kvv[a]; }
#pragma unroll
    for (int t = 0; t < 16; ++t) ok = ok && fabsf(kf[t] - kvv[t >> 2] * kf[t & 3]) <= 1e-6f * kmax;
    Ks[8] = ok ? 1.f : 0.f;
  }

  // ---- the blur (both roles): strip = four output columns x SR rows of one channel; wave (w & 3) owns the rows SR (w & 3) ..,
  // a pass covers four channels with the 256 threads of a role
  const int bw = wave & 3, lid = bw * 64 + lane;
  const float noise_wg = p.noise ? p.noise_w[0] * gain : 0.f;
  const int W2 = 2 * p.w;
  const int64_t hw2 = 4 * hw;
  const int s_og = lane & 15, s_chl = lane >> 4, s_oy0 = SR * bw;
  tc_f32x4 nzr[SR];
#pragma unroll
  for (int oy = 0; oy < SR; ++oy) nzr[oy] = tc_f32x4{0.f, 0.f, 0.f, 0.f};
  float ymax = 0.f;
  auto noise_request = [&](int ty, int tx, int ib) __attribute__((always_inline)) {
    if (p.noise) {
      const int64_t s_pix = (int64_t)(2 * ty * TY + s_oy0) * W2 + 2 * tx * TC_TX + 4 * s_og;
#pragma unroll
      for (int oy = 0; oy < SR; ++oy)
        nzr[oy] = *reinterpret_cast<const tc_f32x4*>(p.noise + (int64_t)ib * hw2 + s_pix + (int64_t)oy * W2);
    }
  };
  auto blur = [&](int pass, int par, int ot, int tx, int ty, int ib) __attribute__((always_inline)) {
    if (Ks[8] != 0.f) {
      const float kh[4] = {Ks[0], Ks[1], Ks[2], Ks[3]}, kv[4] = {Ks[4], Ks[5], Ks[6], Ks[7]};
      const int cl = 4 * pass + s_chl;            // channel within the workgroup's 16
      const int64_t s_pix = (int64_t)(2 * ty * TY + s_oy0) * W2 + 2 * tx * TC_TX + 4 * s_og;
      const float* zb = Zs + cl * CHS + (s_oy0 + 1) * TC_ZP + 4 * s_og + 4;
      const float sc = Sc[par][cl], bs = Bs[par][cl], post = Po[par][cl];
      float* yb = p.y + ((int64_t)ib * p.out_ch + 16 * ot + cl) * hw2 + s_pix;
      tc_f32x4 hrow[4];                           // the last four horizontally filtered rows
      tc_f32x4 lo = *reinterpret_cast<const tc_f32x4*>(zb), hi = *reinterpret_cast<const tc_f32x4*>(zb + 4);
#pragma unroll
      for (int zr = 0; zr < SR + 3; ++zr) {
        tc_f32x4 lon = lo, hin = hi;              // the next z row, requested before this one is filtered
        if (zr + 1 < SR + 3) {
          lon = *reinterpret_cast<const tc_f32x4*>(zb + (zr + 1) * TC_ZP);
          hin = *reinterpret_cast<const tc_f32x4*>(zb + (zr + 1) * TC_ZP + 4);
        }
        tc_f32x4 hsum = tc_f32x4{lo[1], lo[2], lo[3], hi[0]} * kh[0];
        hsum += tc_f32x4{lo[2], lo[3], hi[0], hi[1]} * kh[1];
        hsum += tc_f32x4{lo[3], hi[0], hi[1], hi[2]} * kh[2];
        hsum += hi * kh[3];
        hrow[zr & 3] = hsum;
        if (zr >= 3) {
          const int oy = zr - 3;                  // output row oy0 + oy: filtered rows zr - 3 .. zr
          tc_f32x4 res = hrow[(zr - 3) & 3] * kv[0];
          res += hrow[(zr - 2) & 3] * kv[1];
          res += hrow[(zr - 1) & 3] * kv[2];
          res += hrow[zr & 3] * kv[3];
          const tc_f32x4 nz = nzr[oy] * noise_wg;
          tc_f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float u = res[q] * sc + nz[q] + bs;
            v[q] = fmaxf(u, u * slope) * post;
            ymax = fmaxf(ymax, fabsf(v[q]));
          }
          *reinterpret_cast<tc_f32x4*>(yb + (int64_t)oy * W2) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
        lo = lon; hi = hin;
      }
    } else {
#pragma unroll 1
      for (int k = 0; k < 4; ++k) {
        const int gid = lid + 256 * k;            // 4 channels x 2 TY rows x 16 groups of four outputs
        const int og = gid & 15, oy = (gid >> 4) & (2 * TY - 1), ch = gid >> 8;
        const int cg = 4 * pass + ch;
        const float* zb = Zs + cg * CHS + (oy + 1) * TC_ZP + 4 * og + 4;
        float res[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const tc_f32x4 lo = *reinterpret_cast<const tc_f32x4*>(zb + a * TC_ZP);
          const tc_f32x4 hi = *reinterpret_cast<const tc_f32x4*>(zb + a * TC_ZP + 4);
          const float rowv[7] = {lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) res[q] += rowv[q + cc] * Kf[a * 4 + cc];
        }
        const int64_t pix = (int64_t)(2 * ty * TY + oy) * W2 + 2 * tx * TC_TX + 4 * og;
        tc_f32x4 nz = {0.f, 0.f, 0.f, 0.f};
        if (p.noise) nz = *reinterpret_cast<const tc_f32x4*>(p.noise + (int64_t)ib * hw2 + pix) * noise_wg;
        const float sc = Sc[par][cg], bs = Bs[par][cg], post = Po[par][cg];
        tc_f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float u = res[q] * sc + nz[q] + bs;
          v[q] = fmaxf(u, u * slope) * post;
          ymax = fmaxf(ymax, fabsf(v[q]));
        }
        *reinterpret_cast<tc_f32x4*>(p.y + ((int64_t)ib * p.out_ch + 16 * ot + cg) * hw2 + pix) = v;
      }
    }
  };

  if (wave >= MW) {
    // =========================== staging + blurring waves ===========================
#if TC_STAGE_PRIO
    __builtin_amdgcn_s_setprio(TC_STAGE_PRIO);
#endif
    const int g = wave - MW;
    const float xam = rw_bound_load(p.x_amax);
    const int hw4 = (int)hw * 4;
    int l_pos = 0, l_c = 0, l_ib = -1, l_ot, l_tx, l_ty, l_ibn;       // (l_ot, l_tx, l_ty, l_ibn): the tile being requested
    bool l_past = false;
    digits((unsigned)bx, l_ot, l_tx, l_ty, l_ibn);
    float in_scale = 1.f, out_scale = 1.f;
    int xoff[SI];
    // where a lane's pixels go in a window buffer: the same for every chunk and tile (-1: outside the window)
    int ldst[SI][4];
#pragma unroll
    for (int s = 0; s < SI; ++s) {
      const int it = 64 * s + lane;
      const int r = it / IPR, j = it - r * IPR;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int cc = 4 * j - 2 + e;               // window column of this pixel
        ldst[s][e] = (it < NITEM && cc >= 0 && cc < TC_WC) ? (r * TC_WC + cc) * 64 + ((g ^ tc_swz(cc)) << 4) : -1;
      }
    }
    __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0, 0x00020000);
    struct Flight {
      tc_f32x4 raw[SI][4];                          // [piece][channel]: four pixels
      float psv[4];
      float demod, bias, post, oscale, iscale;
      bool first;
      int par;
    };
    Flight fl[2];
    tc_f32x4 wraw[3];
    float sv[4];
    int l_s0 = 0;
    const unsigned char* l_wsrc = p.wp;
    const unsigned char* w_next = p.wp;
    auto setup = [&](auto tag) __attribute__((always_inline)) {   // the chunk to REQUEST: (l_pos, l_c); loads only, none used here
      Flight& F = fl[decltype(tag)::value];
      F.first = l_c == 0 && !l_past;                              // (past the run: the last chunk again, never read)
      if (F.first) {
        const int ib = (TC_ABL & 1024) ? 0 : l_ibn;                   // (timing ablation 1024: every tile reads image 0's first window -- cache hits)
        const int i0 = (TC_ABL & 1024) ? 0 : l_ty * TY, j0 = (TC_ABL & 1024) ? 0 : l_tx * TC_TX;
        if (ib != l_ib) {
          l_ib = ib;
          float smax = p.style ? 0.f : 1.f;
          if (p.style)
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
          const int iy = i0 - 2 + r, ix = j0 - 4 + 4 * j;            // the item lies inside the row or outside it as a whole
          const bool ok = it < NITEM && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
          xoff[s] = ok ? (iy * p.w + ix) * 4 : 0x7ffffff0;
        }
        F.par = l_pos & 3;
        F.oscale = out_scale;
        if (lid < 16) {
          const int o = 16 * l_ot + lid;
          F.demod = p.demod ? p.demod[(int64_t)l_ib * p.out_ch + o] : 1.f;
          F.bias = p.act ? p.bias[o] : 0.f;
          F.post = p.post ? p.post[(int64_t)l_ib * p.out_ch + o] : 1.f;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) F.psv[k] = p.style ? p.style[(int64_t)l_ib * p.in_ch + 16 * l_c + 4 * g + k] : 1.f;
      F.iscale = in_scale;
      l_s0 = (16 * l_c + 4 * g) * hw4;
      l_wsrc = w_next;                              // the weights of the chunk set up ONE call ago
      w_next = p.wp + ((int64_t)l_ot * T + 9 * l_c) * 1024 + lane * 16;
      if (++l_c == NC) {
        if (l_pos + 1 < count) { l_c = 0; ++l_pos; advance(l_ot, l_tx, l_ty, l_ibn); }
        else { l_c = NC - 1; l_past = true; }
      }
    };
    auto request_s = [&](auto tag, int s, bool with_w = true) __attribute__((always_inline)) {
      Flight& F = fl[decltype(tag)::value];
      if (with_w && s == 0) {                       // the weights FIRST (loads return in order)
        wraw[0] = *reinterpret_cast<const tc_f32x4*>(l_wsrc + g * 1024);
        wraw[1] = *reinterpret_cast<const tc_f32x4*>(l_wsrc + (g + 4) * 1024);
        if (g == 0) wraw[2] = *reinterpret_cast<const tc_f32x4*>(l_wsrc + 8 * 1024);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (TC_ABL & 1) F.raw[s][k] = tc_f32x4{1.f, 1.f, 1.f, 1.f};     // (timing ablation: no window loads; results wrong)
        else F.raw[s][k] = __builtin_bit_cast(tc_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xoff[s], l_s0 + k * hw4, 0));
      }
    };
    auto tables = [&](auto tag) __attribute__((always_inline)) {
      Flight& F = fl[decltype(tag)::value];
#pragma unroll
      for (int k = 0; k < 4; ++k) sv[k] = F.psv[k] * F.iscale;
      if (F.first && lid < 16) {
        Sc[F.par][lid] = F.demod * p.w_scale * F.oscale * gain;
        Bs[F.par][lid] = F.bias * gain;
        Po[F.par][lid] = F.post;
      }
    };
    auto deliver_s = [&](auto tag, int buf, int s) __attribute__((always_inline)) {
      Flight& F = fl[decltype(tag)::value];
      unsigned char* dst = Ls + buf * BUFB;
      unsigned char* wdst = Wl + buf * TC_WCH + lane * 16;
      if (s == 0) {
        *reinterpret_cast<tc_f32x4*>(wdst + g * 1024) = wraw[0];
        *reinterpret_cast<tc_f32x4*>(wdst + (g + 4) * 1024) = wraw[1];
        if (g == 0) *reinterpret_cast<tc_f32x4*>(wdst + 8 * 1024) = wraw[2];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v0 = F.raw[s][0][e] * sv[0], v1 = F.raw[s][1][e] * sv[1], v2 = F.raw[s][2][e] * sv[2],
                    v3 = F.raw[s][3][e] * sv[3];
        const tc_f16x2 h01 = __builtin_convertvector(tc_f32x2{v0, v1}, tc_f16x2);
        const tc_f16x2 h23 = __builtin_convertvector(tc_f32x2{v2, v3}, tc_f16x2);
        float r0, r1, r2, r3;                        // v - (float)h, exact
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h01), "v"(v0));
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h01), "v"(v1));
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h23), "v"(v2));
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h23), "v"(v3));
        const tc_f16x2 l01 = __builtin_convertvector(tc_f32x2{r0, r1}, tc_f16x2);
        const tc_f16x2 l23 = __builtin_convertvector(tc_f32x2{r2, r3}, tc_f16x2);
        tc_f16x8 word = {h01[0], h01[1], h23[0], h23[1], l01[0], l01[1], l23[0], l23[1]};
        if (TC_ABL & 256)                            // (timing ablation: no conversion arithmetic; results wrong)
          word = __builtin_bit_cast(tc_f16x8, tc_f32x4{F.raw[s][0][e], F.raw[s][1][e], F.raw[s][2][e], F.raw[s][3][e]});
        if (!(TC_ABL & 512) && ldst[s][e] >= 0) *reinterpret_cast<tc_f16x8*>(dst + ldst[s][e]) = word;
        if (TC_ABL & 512) asm volatile("" :: "v"(word));
      }
    };

    // chunks 0 and 1 requested (chunk 0's weights with chunk 1's window); chunk 0 written; chunk 2 requested
    setup(tc_int<0>());
#pragma unroll
    for (int s = 0; s < SI; ++s) request_s(tc_int<0>(), s, false);
    setup(tc_int<1>());
#pragma unroll
    for (int s = 0; s < SI; ++s) request_s(tc_int<1>(), s);
    tables(tc_int<0>());
    __builtin_amdgcn_sched_barrier(0);
    setup(tc_int<0>());
#pragma unroll
    for (int s = 0; s < SI; ++s) {
      deliver_s(tc_int<0>(), 0, s); __builtin_amdgcn_sched_barrier(0);
      request_s(tc_int<0>(), s); __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
    int cn = 0, e_pos = 0;                          // the tile the multiplying waves are on
    int e_ot, e_tx, e_ty, e_ib;
    digits((unsigned)bx, e_ot, e_tx, e_ty, e_ib);
    bool have_prev = false;                         // the tile whose z tile is in LDS, waiting to be blurred
    int b_ot = 0, b_tx = 0, b_ty = 0, b_ib = 0, b_par = 0;
    TP_DECL(tp = 0, tp_all = 0, tp_setup = 0, tp_del = 0, tp_bar = 0, tp_blur = 0, tp_zbar = 0);
    TP_NOW(tp); TP_NOW(tp_all);
    // interval n: chunk n + 1 -> LDS piece by piece, the window of chunk n + 3 and the weights of chunk n + 2 requested behind
    // it; then this interval's share of the previous tile's blur
    auto interval = [&](int n, auto tag) __attribute__((always_inline)) {
      tables(tag);
      __builtin_amdgcn_sched_barrier(0);
      setup(tag);
      TP_ADD(tp_setup, tp);
#pragma unroll
      for (int s = 0; s < SI; ++s) {
        deliver_s(tag, (n + 1) & 1, s); __builtin_amdgcn_sched_barrier(0);
        request_s(tag, s); __builtin_amdgcn_sched_barrier(0);
      }
      TP_ADD(tp_del, tp);
      if (have_prev) {
        // the staging waves' passes TC_PP_MQ .. 3, spread over the tile's intervals: pass MQ + k in interval [k NC / NS]
        constexpr int NS = 4 - TC_PP_MQ;
        const int p_lo = TC_PP_MQ + (NS * cn + NC - 1) / NC, p_hi = TC_PP_MQ + (NS * cn + NS + NC - 1) / NC;
#pragma unroll 1
        for (int ps = p_lo; ps < p_hi; ++ps) blur(ps, b_par, b_ot, b_tx, b_ty, b_ib);
      }
      TP_ADD(tp_blur, tp);
      lds_barrier();
      TP_ADD(tp_bar, tp);
      if (++cn == NC) {                             // the multiplying waves write the z tile of THEIR tile now
        cn = 0;
        b_ot = e_ot; b_tx = e_tx; b_ty = e_ty; b_ib = e_ib; b_par = e_pos & 3;
        have_prev = true;
        if (TC_PP_MQ < 4) noise_request(b_ty, b_tx, b_ib);       // (ahead of the next interval's window requests: loads return in order)
        ++e_pos;
        advance(e_ot, e_tx, e_ty, e_ib);
        lds_barrier();                              // z is ready
        TP_ADD(tp_zbar, tp);
      }
    };
    int n = 0;
    for (; n + 1 < N; n += 2) { interval(n, tc_int<1>()); interval(n + 1, tc_int<0>()); }
    if (n < N) interval(n, tc_int<1>());
    // the last tile's blur: nobody waits for it
#pragma unroll 1
    for (int ps = TC_PP_MQ; ps < 4; ++ps) blur(ps, b_par, b_ot, b_tx, b_ty, b_ib);
#if TC_PROF
    if (wave == MW && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {
      unsigned long long* o = tc_prof + (blockIdx.x == 0 ? 0 : 32);
      o[8] = (unsigned long long)clock64() - tp_all; o[9] = tp_setup; o[10] = tp_del; o[11] = tp_bar; o[12] = tp_blur; o[13] = tp_zbar; o[14] = N;
      o[16] = 0;
    }
#endif
    if (p.y_amax) rw_bound_store_wave(p.y_amax, rw_wave_max(ymax));
    return;
  }

  // =========================== multiplying waves ===========================
  const int lk = lane >> 4, lt = lane & 15;
  unsigned pb0[BPW], pb1[BPW];                     // operand addresses as in tconv_body
#pragma unroll
  for (int b = 0; b < BPW; ++b) {
    int q = 16 * (wave + MW * b) + lt;
    q = q < NPOS ? q : NPOS - 1;
    const int r = q / TC_PC, c = q - r * TC_PC;
    pb0[b] = (unsigned)(((r + 1) * TC_WC + c + 1) * 64 + ((lk ^ tc_swz(c + 1)) << 4));
    pb1[b] = (unsigned)(((r + 1) * TC_WC + c) * 64 + ((lk ^ tc_swz(c)) << 4));
  }
  tc_f32x4 acc[BPW][4];
#pragma unroll
  for (int b = 0; b < BPW; ++b)
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) acc[b][ph] = tc_f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int LB = BPW - 1;
  const bool last_ok = wave + MW * LB < NBLK;       // wave-uniform: only the last block of a wave can be missing
  auto mma = [&](const unsigned char* lb, const unsigned char* wb) __attribute__((always_inline)) {
    {
      const tc_f16x8 u0 = TC_UH(0), u1 = TC_UH(1), u3 = TC_UH(3), u4 = TC_UH(4), l4 = tc_expand(TC_UL(4));
      tc_f16x8 pc = TC_PIX(pb0[0]), pd = TC_PIX(pb0[1]);
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        tc_f16x8 pn = pd;
        if (b + 2 < BPW) pn = TC_PIX(pb0[b + 2]);
        if (b < LB || last_ok) {
          TC_MFMA(3, pc, u4); TC_MFMA(0, pc, u0); TC_MFMA(1, pc, u1); TC_MFMA(2, pc, u3);
          TC_MFMA(3, pc, l4);
        }
        __builtin_amdgcn_sched_barrier(0);
        pc = pd; pd = pn;
      }
    }
    {
      const tc_f16x8 u2 = TC_UH(2), u5 = TC_UH(5), m02 = tc_pair(TC_UL(0), TC_UL(2)), m35 = tc_pair(TC_UL(3), TC_UL(5));
      tc_f32x2 hc = TC_VH(pb0[0]), hd = TC_VH(pb0[1]);
      tc_f16x8 qc = TC_PIX(pb1[0]), qd = TC_PIX(pb1[1]);
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        tc_f32x2 hn = hd;
        tc_f16x8 qn = qd;
        if (b + 2 < BPW) { hn = TC_VH(pb0[b + 2]); qn = TC_PIX(pb1[b + 2]); }
        if (b < LB || last_ok) {
          const tc_f16x8 M = tc_pair_hq(hc, qc);
          TC_MFMA(0, qc, u2); TC_MFMA(2, qc, u5);
          TC_MFMA(0, M, m02); TC_MFMA(2, M, m35);
        }
        __builtin_amdgcn_sched_barrier(0);
        hc = hd; qc = qd; hd = hn; qd = qn;
      }
    }
    {
      const tc_f16x8 u6 = TC_UH(6), u7 = TC_UH(7), u8 = TC_UH(8), m17 = tc_pair(TC_UL(1), TC_UL(7)),
                     m68 = tc_pair(TC_UL(6), TC_UL(8));
      tc_f32x2 hc = TC_VH(pb0[0]), hd = TC_VH(pb0[1]);
      tc_f16x8 qc = TC_PIX(pb0[0] - TC_WC * 64), rc = TC_PIX(pb1[0] - TC_WC * 64);
      tc_f16x8 qd = TC_PIX(pb0[1] - TC_WC * 64), rd = TC_PIX(pb1[1] - TC_WC * 64);
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        tc_f32x2 hn = hd;
        tc_f16x8 qn = qd, rn = rd;
        if (b + 2 < BPW) { hn = TC_VH(pb0[b + 2]); qn = TC_PIX(pb0[b + 2] - TC_WC * 64); rn = TC_PIX(pb1[b + 2] - TC_WC * 64); }
        if (b < LB || last_ok) {
          const tc_f16x8 M02 = tc_pair_hq(hc, qc), M23 = tc_pair(qc, rc);
          TC_MFMA(0, qc, u6); TC_MFMA(1, qc, u7);
          TC_MFMA(0, rc, u8); TC_MFMA(1, M02, m17);
          TC_MFMA(0, M23, m68);
        }
        __builtin_amdgcn_sched_barrier(0);
        hc = hd; qc = qd; rc = rd; hd = hn; qd = qn; rd = rn;
      }
    }
  };
  // all sixteen channels -> the z tile: lane (lk, lt) holds channel lt of the positions 4 lk .. 4 lk + 3 of each block
  auto zwrite = [&]() __attribute__((always_inline)) {
    int opaque;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque));          // (keeps the address arithmetic inside the loop: tconv_blur_ws_kernel)
    float* zc = Zs + lt * CHS + 4;
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
      if (wave + MW * b >= NBLK) continue;
      const int q0 = 16 * (wave + MW * b) + 4 * lk + opaque;
      const int r0 = (q0 * 1928) >> 16, c0 = q0 - r0 * TC_PC;          // q0 / 34 for q0 < 400
      float* z0 = zc + (2 * r0) * TC_ZP + 2 * c0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (b == BPW - 1 && q0 + j >= NPOS) continue;                  // (the last block runs past the window: no spare rows here)
        float* zp = z0 + 2 * j + (c0 + j >= TC_PC ? 2 * TC_ZP - 2 * TC_PC : 0);
        zp[0] = acc[b][0][j];
        zp[1] = acc[b][1][j];
        zp[TC_ZP] = acc[b][2][j];
        zp[TC_ZP + 1] = acc[b][3][j];
      }
    }
  };

  int c = 0, pos = 0;
  int ot, tx, ty, ib;
  digits((unsigned)bx, ot, tx, ty, ib);
  lds_barrier();                                    // chunk 0 and the FIR are in LDS
  TP_DECL(tp = 0, tp_all = 0, tp_mma = 0, tp_bar = 0, tp_zw = 0, tp_zbar = 0, tp_blur = 0);
  TP_NOW(tp); TP_NOW(tp_all);
  for (int n = 0; n < N; ++n) {
    const unsigned char* lb = Ls + (n & 1) * BUFB;
    const unsigned char* wb = Wl + (n & 1) * TC_WCH + lane * 8;
    mma(lb, wb);
    TP_ADD(tp_mma, tp);
    lds_barrier();                                  // the chunk is consumed; at a tile's last chunk also: the previous tile is blurred, z is free
    TP_ADD(tp_bar, tp);
    if (c + 1 < NC) { ++c; continue; }
    c = 0;
    if (TC_PP_MQ > 0) noise_request(ty, tx, ib);    // (the z write and a barrier cover it)
    zwrite();
    TP_ADD(tp_zw, tp);
    lds_barrier();                                  // z is ready
    TP_ADD(tp_zbar, tp);
    // this role's share of the tile's blur (passes 0 .. TC_PP_MQ - 1), before the next tile's multiplies; the staging
    // waves take the rest beside them
#pragma unroll 1
    for (int ps = 0; ps < TC_PP_MQ; ++ps) blur(ps, pos & 3, ot, tx, ty, ib);
    TP_ADD(tp_blur, tp);
    ++pos;
    advance(ot, tx, ty, ib);
#pragma unroll
    for (int b = 0; b < BPW; ++b)
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) acc[b][ph] = tc_f32x4{0.f, 0.f, 0.f, 0.f};
  }
#if TC_PROF
  if (wave == 0 && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {
    unsigned long long* o = tc_prof + (blockIdx.x == 0 ? 0 : 32);
    o[0] = (unsigned long long)clock64() - tp_all; o[1] = tp_mma; o[2] = tp_bar; o[3] = tp_zw; o[4] = tp_blur; o[5] = tp_zbar; o[6] = N;
  }
#endif
  if (p.y_amax) rw_bound_store_wave(p.y_amax, rw_wave_max(ymax));
}

static bool tconv_shape_ok(int out_ch, int in_ch, int h, int w) {
  return out_ch > 0 && out_ch % 16 == 0 && in_ch >= 16 && in_ch % 16 == 0 && in_ch <= 512 && w % TC_TX == 0 && h % 16 == 0;
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
  // the form: RW_TCONV_TY = 0 the specialised persistent kernel, 8 / 16 the two shapes of tconv_body; unset: by input channels -- the persistent kernel where a tile has few chunks (<= 128
  // channels: its staging waves run ahead through the epilogue), the one-workgroup-per-CU shape where the MFMAs dominate
  // (profiles/r05q: layer 17 6.4 against 7.1 ms, layer 15 4.6 / 4.9, layer 13 3.7 / 3.7, layer 11 3.3 / 3.2, layer 9 1.9 / 1.7)
  const char* e = getenv("RW_TCONV_TY");
  // (RW_TCONV_PERSISTENT = 0 / 2: which of the two persistent kernels the automatic choice means)
  const char* pe = getenv("RW_TCONV_PERSISTENT");
  // (RW_TCONV_N32 = "lo:hi": the input-channel range the automatic choice gives to the 32-out-channel form.  Default: none --
  // stand-alone it ties the persistent form on layer 15 (4.2 against 4.3 ms), inside the forward the persistent form is ahead:
  // 1526 - 1528 against 1507 - 1518 img/s, same box, interleaved, profiles/r06ak)
  int n32_lo = 1, n32_hi = 0;
  if (const char* ne = getenv("RW_TCONV_N32")) { if (sscanf(ne, "%d:%d", &n32_lo, &n32_hi) != 2) { n32_lo = 1; n32_hi = 0; } }
  const int sel = e ? atoi(e)
                    : (out_ch % 32 == 0 && in_ch >= n32_lo && in_ch <= n32_hi ? 32
                       : (in_ch >= 32 && in_ch <= 128 ? (pe && atoi(pe) == 2 ? 2 : 0) : 16));
  if ((sel == 0 || sel == 2 || sel == 12) && in_ch >= 32) {      // 0: the specialised persistent kernel (one workgroup of eight waves per CU); 2: its pipelined form; 12: twelve waves
    p.tiles_x = w / TC_TX; p.tiles_y = h / 8; p.o_tiles = out_ch / 16;
    const int64_t tiles = (int64_t)batch * p.tiles_y * p.tiles_x * p.o_tiles;
    if (tiles <= 0 || tiles > 0x7fffffff) return RW_ERR_UNSUPPORTED;
    const char* ge = getenv("RW_TCONV_GRID");
    int64_t grid = ge ? atoi(ge) : rw_cu_count();       // one persistent workgroup per compute unit
    grid = grid < 1 ? 1 : (grid > tiles ? tiles : grid);
    const int wv = sel == 12 ? 12 : 8;
    if (y_amax && wv * grid > rw_bound_slot_capacity((int64_t)batch * out_ch * 4 * h * w)) return RW_ERR_UNSUPPORTED;
    if (sel == 2) hipLaunchKernelGGL(tconv_blur_pp_kernel, dim3((unsigned)grid), dim3(512), 0, rw_s(stream), p);
    else if (sel == 12) hipLaunchKernelGGL(tconv_blur_ws12_kernel, dim3((unsigned)grid), dim3(768), 0, rw_s(stream), p);
    else hipLaunchKernelGGL(tconv_blur_ws_kernel, dim3((unsigned)grid), dim3(512), 0, rw_s(stream), p);
    const int rc = RW_LAUNCH_RESULT();
    if (rc || !y_amax) return rc;
    return rw_bound_finish(y_amax, wv * grid, rw_s(stream));
  }
  const bool n32 = sel == 32 && out_ch % 32 == 0;   // 32 out-channels per workgroup (8 x 32 tile, eight waves)
  const int ty = sel == 16 ? 16 : 8;
  const int waves = ty == 16 || n32 ? 8 : 4;
  p.tiles_x = w / TC_TX; p.tiles_y = h / ty; p.o_tiles = out_ch / (n32 ? 32 : 16);
  const int64_t work = (int64_t)batch * p.tiles_y * p.tiles_x * p.o_tiles;
  if (work <= 0 || work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  if (y_amax && waves * work > rw_bound_slot_capacity((int64_t)batch * out_ch * 4 * h * w)) return RW_ERR_UNSUPPORTED;
  if (n32) hipLaunchKernelGGL(tconv_blur_n32_kernel, dim3((unsigned)work), dim3(512), 0, rw_s(stream), p);
  else if (ty == 16) hipLaunchKernelGGL(tconv_blur_t16_kernel, dim3((unsigned)work), dim3(512), 0, rw_s(stream), p);
  else hipLaunchKernelGGL(tconv_blur_t8_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  const int rc = RW_LAUNCH_RESULT();
  if (rc || !y_amax) return rc;
  return rw_bound_finish(y_amax, waves * work, rw_s(stream));
}
