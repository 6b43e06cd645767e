// Stride-1 3x3 modulated convolution (DemodulatedConv2dF, utils/stylegan2/models.py:313-329) by the Winograd
// minimal-filtering algorithm F(2x2, 3x3) in fp32, on the CDNA4 matrix cores.
//
// The direct implicit GEMM (rw_conv.hip) is bound by the fp32 MFMA rate (157 TFLOP/s = the fp32 vector rate): it
// sits at 86 % of that roof and there is no TF32 on gfx950.  F(2x2,3x3) computes a 2x2 output tile from a 4x4
// input tile with 16 multiplications per (in-channel, out-channel) pair instead of 36 -- 2.25x fewer matrix
// FLOPs for the same result in exact arithmetic; in fp32 the error is of the same class as the direct sum
// (transform coefficients are 0, +-1, +-1/2; measured 1.0 - 2.2x the direct kernel's error against float64,
// 5e-7 relative at 512 channels).  It is what cuDNN runs for fp32 3x3 stride-1 convolutions.
//
//   U[xi][o][i] = (G g G^T)[xi]      weights, once per weight version       (rw_pack_conv_weight_wino_f32)
//   V[xi][i][t] = (B^T d B)[xi]      input tiles d (4x4, stride 2), per workgroup, in LDS
//   M[xi][o][t] = sum_i U[xi][o][i] V[xi][i][t]          16 independent GEMMs -> v_mfma_f32_16x16x4_f32
//   Y[o][t]     = A^T M A            2x2 outputs, then the fused epilogue (demod, noise, bias, leaky-ReLU, ToRGB)
//
// Kernel design: see conv_wino16_kernel below.
#include "rw_common.h"

// Timing ablations for kernel work (build a second library with -DWN_ABL=<bits>; results are WRONG when a bit is
// set): 1 = no staging writes / transform, 2 = no barriers in the loop, 4 = no patch fetch, 8 = no B operand reads in
// the loop, 16 = no weight loads in the loop.  0 in the product build.
#ifndef WN_ABL
#define WN_ABL 0
#endif

#define WN_PC 34                // patch columns

struct WinoProblem {
  const float* x; const float* uf; float* y;
  const float* style; const float* demod; const float* noise; const float* noise_w; const float* bias;
  int batch, in_ch, out_ch, h, w;
  int groups_x, groups_y;
  int gpw;                      // tile groups along x per workgroup (divides groups_x)
  float w_scale;
  int act;
  const float* rgb_weight; const float* rgb_style; const float* rgb_bias; const float* rgb_skip; float* rgb_out;
  float rgb_scale;
};

__device__ __forceinline__ int wn_xcd_remap(int id, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, slot = id >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

template <int N> struct wn_int { static constexpr int value = N; };

// ---------------------------------------------------------------------------------------
// The GEMMs run on v_mfma_f32_16x16x4_f32 (32 cycles per instruction, 64 FLOP/clk/SIMD: the rate of 32x32x2).  A
// 16 x 16 accumulator tile is 4 registers, so ONE wave holds all sixteen transform points of 32 out-channels x 16
// tiles (one tile row of 16 tile columns = 2 x 32 output pixels) in 128 registers: two waves per SIMD / two
// workgroups per CU, and the output transform A^T M A is lane-local.  (A first version on 32x32x2 needed 256
// accumulator registers per wave or a partner wave and an exchange through LDS; it measured the same.)
//
// A workgroup (4 waves = WGM out-channel blocks x WGN tile rows) walks a RUN of `gpw` tile groups along x as ONE
// flattened pipeline of "virtual chunks" v = group * n_chunks + chunk:
//   * the raw patch is double buffered (Rs[2]) and a chunk needs ONE barrier: during the MFMAs of chunk v the
//     workgroup transforms chunk v+1 (Rs[(v+1)&1] -> V[(v+1)&1]), stores the fetched chunk v+2 (registers ->
//     Rs[v&1]) and fetches chunk v+3 -- across group boundaries, so only the first group of a run pays a prologue
//     (the 32- and 64-channel layers have only 4 - 16 chunks per group);
//   * the barrier sits right behind the last staging step, a few MFMAs before the end of the chunk, and the B
//     operands of the next chunk are read in the shadow of those MFMAs;
//   * operands are single buffered and streamed: the registers of a weight / V fragment are refilled for the next
//     k-quad right after the MFMAs that consumed them (28 MFMAs ~ 900 cycles ahead of their next use); the weight
//     stream wraps around at the end of a group (same out-channel tile for the whole run).
// LDS banks: V channel rows are padded (the 16x16x4 B operand reads 16 tiles of 4 consecutive channels: unpadded,
// channels k and k+1 share banks), and the raw patch has a row pitch of 48 floats (40 for the one-tile-row shape) so
// that the two 16-lane halves of a transform read (tile rows r, r+1, or channels c, c+1) fall on disjoint banks.
// Shapes: <4,1> 128 out-channels x 1 tile row, <2,2> 64 x 2, <1,4> 32 x 4; IC = 8 input channels per chunk (4 for
// <1,4>, whose V is twice as wide).  The more out-channels share one transformed patch, the less staging per MFMA.
//   weights: uf[o/32][i/4][xi/4][(o%32)/16][lane][xi%4], o = 32 (o/32) + 16 ((o%32)/16) + (lane & 15),
//            i = 4 (i/4) + (lane >> 4)                                     (rw_pack_conv_weight_wino_f32)
// ---------------------------------------------------------------------------------------
typedef float wn_f32x4 __attribute__((ext_vector_type(4)));

// NRW = 16: maps 16 pixels wide (the 16^2 layers: most of a key-statistics sweep at layer 8): a wave's "tile row" of 16
// tiles is then TWO map tile rows of 8 tiles -- tile lt sits at map tile row 2 wn + (lt >> 3), column lt & 7 -- the
// raw patch is (4 WGN + 2) rows x 18 columns, one group spans the map's width (groups_x = gpw = 1); everything else
// (chunk pipeline, operand streams, slot schedule: NRAW and NIT are those of the wide shapes) is unchanged.
// NRW = 8, 4: the 8^2 and 4^2 maps.  A wave's 16 tiles are ONE whole 8 x 8 image (4 x 4 tiles) or FOUR 4 x 4 images (2 x 2
// tiles each); a workgroup covers WGN resp. 4 WGN consecutive images of the batch, whose zero-bordered windows
// ((H + 2) x (W + 2) each) are stacked in the raw patch.  The demodulation factors are then per lane (registers) and
// the input arrives ALREADY multiplied by its style (the host side does that on these tiny maps).
template <int WGM, int WGN, int IC, bool RGB, int NRW = 0>
__global__ void __launch_bounds__(256, 2) conv_wino16_kernel(const WinoProblem p) {
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  static_assert(!(RGB && NRW), "the ToRGB epilogue is for the last (widest) layer");
  constexpr int KQ = IC / 4;                       // k-quads per chunk
  constexpr int NT = 16 * WGN;                     // tiles per workgroup: WGN tile rows x 16 tile columns
  constexpr int VP = NT == 16 ? 48 : NT + 16;      // padded channel-row pitch of V (see the bank notes above)
  // row pitch of the raw patch (NRW: 24 -- the two 8-lane halves of a transform read sit two rows apart, 48 floats =
  // 16 banks)
  constexpr bool WHOLE = NRW == 8 || NRW == 4;     // whole images per wave
  constexpr int IPW = NRW == 8 ? WGN : (NRW == 4 ? 4 * WGN : 1);       // images per workgroup
  constexpr int IROWS = NRW + 2;                   // WHOLE: window rows of one image
  constexpr int RS = NRW == 16 ? 24 : (NRW == 8 ? 12 : (NRW == 4 ? 8 : (NT == 16 ? 40 : 48)));
  constexpr int PC = NRW == 16 ? 18 : (WHOLE ? NRW + 2 : WN_PC);       // patch columns
  constexpr int PR = NRW == 16 ? 4 * WGN + 2 : (WHOLE ? IPW * IROWS : 2 * WGN + 2);   // patch rows
  constexpr int NPOS = PR * PC;
  constexpr int PSLOT = (NPOS + 255) / 256;
  constexpr int NRAW = PSLOT * IC;
  // transform items (tile, channel) per thread and chunk; with fewer items than threads (<4,1>: 16 tiles x 8
  // channels) only the first NT * IC threads transform
  constexpr int NIT = NT * IC >= 256 ? NT * IC / 256 : 1;
  constexpr bool XF_ALL = NT * IC >= 256;
  static_assert(NT * IC % 256 == 0 || NT * IC == 128, "transform items");
  constexpr int CH_STEP = 256 / NT;
  constexpr int SLOTS = 32 * KQ;                   // MFMA slots per chunk and wave
  constexpr int NXS = 1 + 4 * NIT;                 // transform slots: the reads, then two compute steps per slot
  constexpr int STRIDE = (SLOTS - 6) / (NXS + NRAW);           // staging step every STRIDE slots
  static_assert(STRIDE >= 1, "staging fits the slots of a chunk");
  constexpr int FETCH_SLOT = STRIDE * (NXS + NRAW);             // first slot after the last staging step
  constexpr int BAR_SLOT = FETCH_SLOT;                          // barrier right there; >= 5 MFMAs follow it
  static_assert(BAR_SLOT + 5 <= SLOTS, "room for the B prefetch behind the barrier");
  __shared__ __attribute__((aligned(16))) float Rs[2][IC][PR][RS];
  __shared__ __attribute__((aligned(16))) float Vs[2][16][IC][VP];
  // per-out-channel constants of the epilogue: [0] w_scale * demod, [1] bias, [2..4] ToRGB weights x style x scale
  __shared__ float Ct[RGB ? 5 : 2][32 * WGM];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int lk = lane >> 4, lt = lane & 15;        // k within the quad / tile column (A: out-channel row)

  const int local = wn_xcd_remap(blockIdx.x, gridDim.x);
  const int o_tiles = p.out_ch / (32 * WGM);
  const int runs_x = p.groups_x / p.gpw;
  const int ot = local % o_tiles;
  int pg = local / o_tiles;
  const int run = pg % runs_x; pg /= runs_x;
  const int gy = pg % p.groups_y;
  const int ib = pg / p.groups_y;
  const int o0 = ot * 32 * WGM;
  const int y0 = WHOLE ? 0 : gy * (NRW == 16 ? 4 : 2) * WGN, gx0 = run * p.gpw;
  const int64_t hw = (int64_t)p.h * p.w;
  const int img0 = ib * IPW;                       // first image of this workgroup
  const int nimg = min(IPW, p.batch - img0);
  const float* xb = p.x + (int64_t)img0 * p.in_ch * hw;
  const float* st = (p.style && !WHOLE) ? p.style + (int64_t)ib * p.in_ch : nullptr;
  const int NC = p.in_ch / IC;
  const int VT = p.gpw * NC;
  // map position of this lane's tile (tile row wn of the workgroup, tile lt): first pixel row relative to y0 / column
  // relative to the group's first column
  const int t_row = NRW == 16 ? 2 * (2 * wn + (lt >> 3)) : (NRW == 8 ? 2 * (lt >> 2) : (NRW == 4 ? 2 * ((lt >> 1) & 1) : 2 * wn));
  const int t_col = NRW == 16 ? 2 * (lt & 7) : (NRW == 8 ? 2 * (lt & 3) : (NRW == 4 ? 2 * (lt & 1) : 2 * lt));
  const int img_l = NRW == 8 ? wn : (NRW == 4 ? 4 * wn + (lt >> 2) : 0);       // this lane's image within the workgroup
  const bool img_ok = img_l < nimg;
  const int img = img0 + img_l;

  // The epilogue runs once per tile group, in the middle of the load stream: a global load there queues behind the
  // patch fetch just issued (vmcnt retires in order) and costs a full memory latency per group.  Everything it needs
  // per out-channel is therefore put into LDS once per workgroup (visible after the prologue's first barrier); the
  // per-pixel noise / running image are fetched at the start of a group's last chunk, ahead of that chunk's fetch.
  float scl[8];                                     // WHOLE: this lane's w_scale * demod, [half 2][j 4]
  if (WHOLE) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      scl[q] = (p.demod && img_ok) ? p.demod[(int64_t)img * p.out_ch + o0 + 32 * wm + 16 * (q >> 2) + 4 * lk + (q & 3)] * p.w_scale
                                   : p.w_scale;
  }
  if (tid < 32 * WGM) {
    const int o = o0 + tid;
    Ct[0][tid] = (p.demod && !WHOLE) ? p.demod[(int64_t)ib * p.out_ch + o] * p.w_scale : p.w_scale;
    Ct[1][tid] = p.act ? p.bias[o] : 0.f;
    if (RGB) {
      const float sr = p.rgb_scale * p.rgb_style[(int64_t)ib * p.out_ch + o];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) Ct[2 + cc][tid] = sr * p.rgb_weight[cc * p.out_ch + o];
    }
  }

  // ---- raw patch slots: position (r, c) = (pos / 34, pos % 34) of the patch.  The fetch is a BUFFER load
  // (descriptor = this image's feature maps, scalar channel offset, 32-bit lane offset): positions outside the
  // image get an out-of-range offset and the hardware returns 0 -- the zero padding costs no instruction, and
  // there is no 64-bit address arithmetic.  Lane offsets are recomputed when the fetch cursor enters a new group.
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)((int64_t)nimg * p.in_ch * hw * 4), 0x00020000);
  int xoff[PSLOT], xlds[PSLOT];
#pragma unroll
  for (int sl = 0; sl < PSLOT; ++sl) {
    const int pos = tid + 256 * sl;
    const int r = pos / PC, c = pos - r * PC;
    xlds[sl] = pos < NPOS ? r * RS + c : RS - 1;
  }
  auto set_group = [&](int g) __attribute__((always_inline)) {
    const int x0 = (gx0 + g) * 32;
#pragma unroll
    for (int sl = 0; sl < PSLOT; ++sl) {
      const int pos = tid + 256 * sl;
      const int r = pos / PC, c = pos - r * PC;
      const int m = WHOLE ? r / IROWS : 0;                        // image of this window row
      const int iy = y0 - 1 + (WHOLE ? r - m * IROWS : r), ix = x0 - 1 + c;
      const bool ok = pos < NPOS && m < nimg && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
      xoff[sl] = ok ? (m * p.in_ch * (int)hw + iy * p.w + ix) * 4 : 0x7fffffff;          // bytes; out of range -> 0
    }
  };
  float xreg[PSLOT][IC];
  float sty[IC];
  const int hw4 = (int)hw * 4;
  auto xfetch = [&](int i0) __attribute__((always_inline)) {
#pragma unroll
    for (int ic = 0; ic < IC; ++ic)
#pragma unroll
      for (int sl = 0; sl < PSLOT; ++sl)
        xreg[sl][ic] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xsrc, xoff[sl], (i0 + ic) * hw4, 0));
    if (st) {
#pragma unroll
      for (int ic = 0; ic < IC; ++ic) sty[ic] = st[i0 + ic];
    } else {
#pragma unroll
      for (int ic = 0; ic < IC; ++ic) sty[ic] = 1.0f;
    }
  };
  auto raw_step = [&](int rbuf, int j) __attribute__((always_inline)) {
    const int sl = j / IC, ic = j % IC;
    (&Rs[0][0][0][0])[rbuf * IC * PR * RS + ic * PR * RS + xlds[sl]] = xreg[sl][ic] * sty[ic];
  };
  // fetch of virtual chunk vf (uniform arguments, computed by the caller): chunk fc of group fg
  auto fetch_chunk = [&](int fg, int fc) __attribute__((always_inline)) {
    if (fc == 0) set_group(fg);
    xfetch(fc * IC);
  };

  // ---- transform items: tile tl of the workgroup (tile row tl >> 4, column tl & 15), channel tch + CH_STEP * item
  const int tl = tid % NT, tch = tid / NT;
  const float* rsrc = NRW == 16 ? &Rs[0][tch][2 * (2 * (tl >> 4) + ((tl & 15) >> 3))][2 * (tl & 7)]
                      : NRW == 8 ? &Rs[0][tch][(tl >> 4) * IROWS + 2 * ((tl & 15) >> 2)][2 * (tl & 3)]
                      : NRW == 4 ? &Rs[0][tch][(4 * (tl >> 4) + ((tl & 15) >> 2)) * IROWS + 2 * ((tl >> 1) & 1)][2 * (tl & 1)]
                                 : &Rs[0][tch][2 * (tl >> 4)][2 * (tl & 15)];
  float* vdst = &Vs[0][0][tch][tl];
  float2 drow[NIT][4][2];
  float e[4][4];
  constexpr int NXF = 9 * NIT;
  auto xform_step = [&](int buf, int s) __attribute__((always_inline)) {          // Rs[buf] -> Vs[buf]
    if (!XF_ALL && tid >= NT * IC) return;         // whole waves (NT * IC is a multiple of 64)
    if (s < NIT) {
      const float* src = rsrc + buf * IC * PR * RS + s * CH_STEP * PR * RS;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        drow[s][a][0] = *reinterpret_cast<const float2*>(src + a * RS);
        drow[s][a][1] = *reinterpret_cast<const float2*>(src + a * RS + 2);
      }
      return;
    }
    const int k = s - NIT, item = k / 8, ms = k % 8;
    if (ms < 4) {
      const int a = ms;
      const float d0 = drow[item][a][0].x, d1 = drow[item][a][0].y, d2 = drow[item][a][1].x, d3 = drow[item][a][1].y;
      e[a][0] = d0 - d2; e[a][1] = d1 + d2; e[a][2] = d2 - d1; e[a][3] = d1 - d3;
    } else {
      const int b = ms - 4;
      float* dst = vdst + buf * 16 * IC * VP + item * CH_STEP * VP;
      dst[(0 * 4 + b) * IC * VP] = e[0][b] - e[2][b];
      dst[(1 * 4 + b) * IC * VP] = e[1][b] + e[2][b];
      dst[(2 * 4 + b) * IC * VP] = e[2][b] - e[1][b];
      dst[(3 * 4 + b) * IC * VP] = e[1][b] - e[3][b];
    }
  };
  auto xform_slot = [&](int buf, int s2) __attribute__((always_inline)) {
    if (s2 == 0) {
#pragma unroll
      for (int i = 0; i < NIT; ++i) xform_step(buf, i);
    } else {
      const int k = NIT + 2 * (s2 - 1);
      if (k < NXF) xform_step(buf, k);
      if (k + 1 < NXF) xform_step(buf, k + 1);
    }
  };

  // ---- operands.  A: uniform base + 32-bit lane offset; one 16-byte load = four transform points of one
  // 16-channel half.  B: V[buf][xi][4 kq + lk][16 wn + lt].
  const int kq_total = p.in_ch >> 2;
  const float* ufs = p.uf + ((int64_t)((o0 >> 5) + wm) * kq_total) * 2048;     // uniform; + kq * 2048
  const int a_lane = lane * 4;
  wn_f32x4 areg[4][2];                              // [xi / 4][16-channel half] over xi % 4
  float breg[16];
  auto aload = [&](int xq, int kq) __attribute__((always_inline)) {
    const float* base = ufs + (int64_t)kq * 2048 + xq * 512;
    areg[xq][0] = *reinterpret_cast<const wn_f32x4*>(base + a_lane);
    areg[xq][1] = *reinterpret_cast<const wn_f32x4*>(base + 256 + a_lane);
  };
  const float* vsrc = &Vs[0][0][lk][wn * 16 + lt];
  auto bload = [&](int buf, int kq, int xi) __attribute__((always_inline)) {
    breg[xi] = vsrc[buf * 16 * IC * VP + xi * IC * VP + 4 * kq * VP];
  };

  wn_f32x4 acc[16][2];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[xi][h] = wn_f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: V[0] = chunk 0, Rs[1] = chunk 1, registers = chunk 2; operands of the first k-quad
  set_group(0);
  xfetch(0);
#pragma unroll
  for (int xq = 0; xq < 4; ++xq) aload(xq, 0);
#pragma unroll
  for (int j = 0; j < NRAW; ++j) raw_step(0, j);
  __syncthreads();
  if (VT > 1) fetch_chunk(1 / NC, 1 % NC);
#pragma unroll
  for (int s = 0; s < NXF; ++s) xform_step(0, s);
  if (VT > 1) {
#pragma unroll
    for (int j = 0; j < NRAW; ++j) raw_step(1, j);
    if (VT > 2) fetch_chunk(2 / NC, 2 % NC);
  }
  __syncthreads();
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) bload(0, 0, xi);

  // Epilogue of one group: lane-local output transform of this wave's 32 out-channels x 16 tiles.
  // acc[4a + b][half][j] = M[a][b] of out-channel o0 + 32 wm + 16 half + 4 lk + j, tile column lt.
  float2 pre_nz[2];                                 // noise of this lane's 2x2 pixels (rows oy, oy + 1)
  float2 pre_sk[RGB ? 3 : 1][2];                    // running RGB image, likewise
  const float rgb_b0 = RGB && p.rgb_bias ? p.rgb_bias[0] : 0.f, rgb_b1 = RGB && p.rgb_bias ? p.rgb_bias[1] : 0.f,
              rgb_b2 = RGB && p.rgb_bias ? p.rgb_bias[2] : 0.f;
  const float noise_w = p.noise ? p.noise_w[0] : 0.f;
  auto epilogue_prefetch = [&](int g) __attribute__((always_inline)) {
    const int64_t pix = (int64_t)(y0 + t_row) * p.w + (gx0 + g) * 32 + t_col;
    if (p.noise && (!WHOLE || img_ok)) {
      const float* np = p.noise + (int64_t)(WHOLE ? img : ib) * hw + pix;
      pre_nz[0] = *reinterpret_cast<const float2*>(np);
      pre_nz[1] = *reinterpret_cast<const float2*>(np + p.w);
    }
    if (RGB && p.rgb_skip) {
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const float* sp = p.rgb_skip + ((int64_t)ib * 3 + cc) * hw + pix;
        pre_sk[cc][0] = *reinterpret_cast<const float2*>(sp);
        pre_sk[cc][1] = *reinterpret_cast<const float2*>(sp + p.w);
      }
    }
  };
  auto group_epilogue = [&](int g) __attribute__((always_inline)) {
    const int x0 = (gx0 + g) * 32;
    const int oy = y0 + t_row, ox = x0 + t_col;
    const int o_first = o0 + 32 * wm + 4 * lk;            // + 16 half + j
    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.noise) {
      nz[0] = pre_nz[0].x * noise_w; nz[1] = pre_nz[0].y * noise_w;
      nz[2] = pre_nz[1].x * noise_w; nz[3] = pre_nz[1].y * noise_w;
    }
    float rgbp[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) rgbp[j][0] = rgbp[j][1] = rgbp[j][2] = 0.f;
    const float* ct = &Ct[0][32 * wm + 4 * lk];
    // neighbouring lanes (tile columns 2m, 2m+1) exchange halves: the even lane stores four consecutive pixels of
    // output row oy, the odd lane four of row oy + 1 -- one aligned 16-byte store per lane and channel
    const bool odd = lt & 1;
    // (WHOLE: a lane whose image lies past the batch writes nothing; its exchange partner shares the image)
    float* yb = (p.y && (!WHOLE || img_ok))
                    ? p.y + ((int64_t)(WHOLE ? img : ib) * p.out_ch + o_first) * hw + (int64_t)(oy + (odd ? 1 : 0)) * p.w + (ox & ~3)
                    : nullptr;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int half = q >> 2, j = q & 3;
      const int oc = 16 * half + j;
      const float scale = WHOLE ? scl[q] : ct[oc];
      const float bias = ct[32 * WGM + oc];
      float wr[3] = {0.f, 0.f, 0.f};
      if (RGB) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) wr[cc] = ct[(2 + cc) * 32 * WGM + oc];
      }
      float s0[4], s1[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float m0 = acc[b][half][j], m1 = acc[4 + b][half][j], m2 = acc[8 + b][half][j], m3 = acc[12 + b][half][j];
        s0[b] = m0 + m1 + m2;
        s1[b] = m1 - m2 - m3;
      }
      float v[4] = {s0[0] + s0[1] + s0[2], s0[1] - s0[2] - s0[3], s1[0] + s1[1] + s1[2], s1[1] - s1[2] - s1[3]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = v[k] * scale + nz[k];
        if (p.act) {
          t += bias;
          t = ((t > 0.f) ? t : t * 0.2f) * 1.4142135623730951f;
        }
        v[k] = t;
        if (RGB) {
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) rgbp[k][cc] += t * wr[cc];
        }
      }
      if (yb) {
        const float g0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(odd ? v[0] : v[2]), 0xB1, 0xf, 0xf, true));
        const float g1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(odd ? v[1] : v[3]), 0xB1, 0xf, 0xf, true));
        wn_f32x4 o4;
        o4[0] = odd ? g0 : v[0]; o4[1] = odd ? g1 : v[1];
        o4[2] = odd ? v[2] : g0; o4[3] = odd ? v[3] : g1;
        *reinterpret_cast<wn_f32x4*>(yb + (int64_t)(16 * half + j) * hw) = o4;
      }
    }
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
      for (int h = 0; h < 2; ++h) acc[xi][h] = wn_f32x4{0.f, 0.f, 0.f, 0.f};
    if (RGB) {
      // the four 16-lane groups of the wave hold the other out-channels of the same pixels
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          float t = rgbp[k][cc];
          t += __shfl_xor(t, 16, 64);
          t += __shfl_xor(t, 32, 64);
          rgbp[k][cc] = t;
        }
      if (lk == 0) {
        const float rb[3] = {rgb_b0, rgb_b1, rgb_b2};
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          const int64_t base = ((int64_t)ib * 3 + cc) * hw + (int64_t)oy * p.w + ox;
          float2 k0 = {0.f, 0.f}, k1 = {0.f, 0.f};
          if (p.rgb_skip) { k0 = pre_sk[cc][0]; k1 = pre_sk[cc][1]; }
          float2 r0 = {rgbp[0][cc] + rb[cc] + k0.x, rgbp[1][cc] + rb[cc] + k0.y};
          float2 r1 = {rgbp[2][cc] + rb[cc] + k1.x, rgbp[3][cc] + rb[cc] + k1.y};
          *reinterpret_cast<float2*>(p.rgb_out + base) = r0;
          *reinterpret_cast<float2*>(p.rgb_out + base + p.w) = r1;
        }
      }
    }
  };

  // One virtual chunk.  STAGE: 2 = transform v+1, store v+2, fetch v+3; 1 = transform v+1 only; 0 = nothing.
  // Slot = one MFMA; per k-quad the order is xi-quad, xi, 16-channel half (a B value feeds two MFMAs, an A
  // register four), and a quad's registers are refilled for the next k-quad right behind its last MFMA.
  auto chunk = [&](int v, int c, int fg, int fc, auto stage_tag) __attribute__((always_inline)) {
    constexpr int STAGE = decltype(stage_tag)::value;
    const int buf = v & 1;
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
      int kqn = c * KQ + kq + 1;                   // next k-quad of the weight stream (wraps at the end of a group)
      if (kqn >= kq_total) kqn = 0;
#pragma unroll
      for (int xq = 0; xq < 4; ++xq) {
#pragma unroll
        for (int xe = 0; xe < 4; ++xe) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int xi = 4 * xq + xe;
            const int slot = kq * 32 + xq * 8 + xe * 2 + h;
            acc[xi][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[xq][h][xe], breg[xi], acc[xi][h], 0, 0, 0);
            if (STAGE >= 1 && !(WN_ABL & 1) && slot % STRIDE == 0) {
              const int step = slot / STRIDE;
              if (step < NXS) xform_slot(buf ^ 1, step);
              else if (STAGE == 2 && step < NXS + NRAW) raw_step(buf, step - NXS);
            }
            if (h == 1 && !(WN_ABL & 8)) {
              // this B value is consumed: next k-quad of the same chunk now; next chunk only behind the barrier
              if (kq + 1 < KQ) bload(buf, kq + 1, xi);
              else if (STAGE >= 1 && slot > BAR_SLOT) bload(buf ^ 1, 0, xi);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (STAGE >= 1 && kq == KQ - 1 && slot == BAR_SLOT) {
              // unconditional: the last turn of this loop fetches one virtual chunk past the run (a legal address,
              // nobody reads it).  Behind a branch, the compiler must count vmcnt for the path WITHOUT the
              // fetch, and every weight wait that follows then drains the fetch on the path with it.
              if (STAGE == 2 && !(WN_ABL & 4)) fetch_chunk(fg, fc);
              if (!(WN_ABL & 2)) __syncthreads();  // V[buf^1] and Rs[buf] complete; nobody reads V[buf] any more
              if (!(WN_ABL & 8)) {
#pragma unroll
                for (int x2 = 0; x2 < 16; ++x2)
                  if (2 * x2 + 1 <= BAR_SLOT - 32 * (KQ - 1)) bload(buf ^ 1, 0, x2);   // consumed before the barrier
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        if (!(WN_ABL & 16)) aload(xq, kqn);        // weights of this xi-quad for the next k-quad
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  int v = 0, c = 0, g = 0;
  int fg = 3 / NC, fc = 3 % NC;                     // (group, chunk) of virtual chunk v + 3
  auto step = [&](auto stage_tag) __attribute__((always_inline)) {
    if (c == NC - 1) epilogue_prefetch(g);         // ahead of this chunk's patch fetch in the load queue
    chunk(v, c, fg, fc, stage_tag);
    if (c == NC - 1) { group_epilogue(g); c = 0; ++g; } else { ++c; }
    if (++fc == NC) { fc = 0; ++fg; }
    ++v;
  };
  for (; v + 2 < VT;) step(wn_int<2>());
  if (v + 1 < VT) step(wn_int<1>());
  step(wn_int<0>());
}

__global__ void __launch_bounds__(256) pack_wino16_kernel(const float* __restrict__ w, float* __restrict__ uf,
                                                          int out_ch, int in_ch) {
  const int64_t total = (int64_t)out_ch * in_ch;       // one thread: the 16 values of one (o, i)
  const int kqn = in_ch >> 2;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    int64_t rest = idx >> 6;
    const int half = (int)(rest & 1); rest >>= 1;
    const int kq = (int)(rest % kqn);
    const int ob = (int)(rest / kqn);
    const int o = 32 * ob + 16 * half + (lane & 15), i = 4 * kq + (lane >> 4);
    const float* g = w + ((int64_t)o * in_ch + i) * 9;
    float gg[4][3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float g0 = g[kx], g1 = g[3 + kx], g2 = g[6 + kx];
      gg[0][kx] = g0;
      gg[1][kx] = 0.5f * (g0 + g1 + g2);
      gg[2][kx] = 0.5f * (g0 - g1 + g2);
      gg[3][kx] = g2;
    }
    float* dst = uf + ((int64_t)ob * kqn + kq) * 2048 + half * 256 + lane * 4;
#pragma unroll
    for (int a = 0; a < 4; ++a)
      *reinterpret_cast<wn_f32x4*>(dst + a * 512) =
          wn_f32x4{gg[a][0], 0.5f * (gg[a][0] + gg[a][1] + gg[a][2]), 0.5f * (gg[a][0] - gg[a][1] + gg[a][2]), gg[a][2]};
  }
}

static bool wino_narrow(int h, int w) { return w == 16 && h % 16 == 0; }      // the NRW shapes (see conv_wino16_kernel)
static bool wino_whole(int h, int w) { return (w == 8 && h == 8) || (w == 4 && h == 4); }   // whole images per wave

static bool wino_shape_ok(int out_ch, int in_ch, int h, int w) {
  if (!(out_ch > 0 && in_ch > 0 && out_ch % 32 == 0 && in_ch % 8 == 0)) return false;
  return (w % 32 == 0 && w >= 32 && h % 8 == 0 && h >= 8) || wino_narrow(h, w) || wino_whole(h, w);
}

extern "C" int rw_conv3x3_wino_supported(int out_ch, int in_ch, int h, int w) {
  return wino_shape_ok(out_ch, in_ch, h, w) ? 1 : 0;
}

extern "C" long long rw_packed_conv_weight_wino_elems(int out_ch, int in_ch) {
  if (out_ch <= 0 || in_ch <= 0 || out_ch % 32 || in_ch % 2) return -1;
  return 16LL * out_ch * in_ch;
}

#include <stdlib.h>
static int wn_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

extern "C" int rw_pack_conv_weight_wino_f32(const float* w, float* uf, int out_ch, int in_ch, rw_stream_t stream) {
  RW_CHECK_ARG(w && uf && out_ch > 0 && in_ch > 0);
  if (out_ch % 32 || in_ch % 4) return RW_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)out_ch * in_ch;
  hipLaunchKernelGGL(pack_wino16_kernel, dim3(rw_stream_grid(total, 256)), dim3(256), 0, rw_s(stream), w, uf, out_ch,
                     in_ch);
  return RW_LAUNCH_RESULT();
}

// Workgroup shape by out-channel count: <4,1> 128 out-channels x 1 tile row (2 x 32 pixels), <2,2> 64 x 2 tile rows,
// <1,4> 32 x 4 tile rows; each walks runs of gpw tile groups along x (RW_WINO_GPW / RW_WINO_TILE: tuning overrides).
static int launch_wino(WinoProblem& p, bool rgb, hipStream_t s) {
  // (<4,1>: 128 out-channels x 1 tile row halves the staging per MFMA but gives each of the four waves its own weight
  // stream: measured 195-222 vs 240 TFLOP/s effective -- the weight stream binds first; kept selectable for tuning)
  int bm = p.out_ch % 64 == 0 ? 64 : 32;
  const int force = wn_env("RW_WINO_TILE", 0);
  if (force && p.out_ch % force == 0 && (force == 32 || force == 64 || force == 128)) bm = force;
  const bool narrow = wino_narrow(p.h, p.w), whole = wino_whole(p.h, p.w);
  if ((narrow || whole) && bm == 128) bm = 64;    // NRW exists for the <2,2> and <1,4> shapes
  const int wgn = 128 / bm;                       // tile rows per workgroup: 1, 2, 4
  if (whole && (rgb || p.style)) return RW_ERR_UNSUPPORTED;      // 8^2 / 4^2 maps arrive multiplied by their style
  if (!whole && (p.h % ((narrow ? 4 : 2) * wgn) || (narrow && rgb))) return RW_ERR_UNSUPPORTED;
  p.groups_x = (narrow || whole) ? 1 : p.w / 32;
  p.groups_y = whole ? 1 : p.h / ((narrow ? 4 : 2) * wgn);
  const int ipw = whole ? (p.w == 8 ? wgn : 4 * wgn) : 1;        // images per workgroup
  const int wg_batch = (p.batch + ipw - 1) / ipw;
  const int o_tiles = p.out_ch / bm;
  int gpw = wn_env("RW_WINO_GPW", 8);
  if (gpw < 1) gpw = 1;
  if (gpw > p.groups_x) gpw = p.groups_x;
  while (p.groups_x % gpw) --gpw;
  // short launches: keep at least ~4 workgroups per CU
  while (gpw > 1 && (int64_t)wg_batch * p.groups_y * (p.groups_x / gpw) * o_tiles < 1024) {
    --gpw;
    while (p.groups_x % gpw) --gpw;
  }
  p.gpw = gpw;
  const int64_t work = (int64_t)wg_batch * p.groups_y * (p.groups_x / gpw) * o_tiles;
  if (work <= 0 || work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)work), block(256);
  if (rgb && bm != 32) return RW_ERR_UNSUPPORTED;
  if (whole && p.w == 8 && bm == 64) hipLaunchKernelGGL((conv_wino16_kernel<2, 2, 8, false, 8>), grid, block, 0, s, p);
  else if (whole && p.w == 8) hipLaunchKernelGGL((conv_wino16_kernel<1, 4, 4, false, 8>), grid, block, 0, s, p);
  else if (whole && bm == 64) hipLaunchKernelGGL((conv_wino16_kernel<2, 2, 8, false, 4>), grid, block, 0, s, p);
  else if (whole) hipLaunchKernelGGL((conv_wino16_kernel<1, 4, 4, false, 4>), grid, block, 0, s, p);
  else if (narrow && bm == 64) hipLaunchKernelGGL((conv_wino16_kernel<2, 2, 8, false, 16>), grid, block, 0, s, p);
  else if (narrow) hipLaunchKernelGGL((conv_wino16_kernel<1, 4, 4, false, 16>), grid, block, 0, s, p);
  else if (bm == 128) hipLaunchKernelGGL((conv_wino16_kernel<4, 1, 8, false>), grid, block, 0, s, p);
  else if (bm == 64) hipLaunchKernelGGL((conv_wino16_kernel<2, 2, 8, false>), grid, block, 0, s, p);
  else if (rgb) hipLaunchKernelGGL((conv_wino16_kernel<1, 4, 4, true>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((conv_wino16_kernel<1, 4, 4, false>), grid, block, 0, s, p);
  return RW_LAUNCH_RESULT();
}

static void wino_fill(WinoProblem& p, const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch,
                      int h, int w, float w_scale, const rw_conv_epilogue* ep) {
  p.x = x; p.uf = uf; p.y = y;
  p.style = ep ? ep->style : nullptr; p.demod = ep ? ep->demod : nullptr; p.noise = ep ? ep->noise : nullptr;
  p.noise_w = ep ? ep->noise_w : nullptr; p.bias = ep ? ep->bias : nullptr; p.act = ep ? ep->act : 0;
  p.batch = batch; p.in_ch = in_ch; p.out_ch = out_ch; p.h = h; p.w = w; p.w_scale = w_scale;
  p.rgb_weight = nullptr; p.rgb_style = nullptr; p.rgb_bias = nullptr; p.rgb_skip = nullptr; p.rgb_out = nullptr;
  p.rgb_scale = 0.f;
}

extern "C" int rw_conv3x3_wino_f32(const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch, int h,
                                   int w, float w_scale, const rw_conv_epilogue* ep, rw_stream_t stream) {
  RW_CHECK_ARG(x && uf && y && batch > 0 && in_ch > 0 && out_ch > 0 && h > 0 && w > 0);
  RW_CHECK_ARG(!ep || ((!ep->noise || ep->noise_w) && (!ep->act || ep->bias)));
  if (!wino_shape_ok(out_ch, in_ch, h, w)) return RW_ERR_UNSUPPORTED;
  WinoProblem p;
  wino_fill(p, x, uf, y, batch, in_ch, out_ch, h, w, w_scale, ep);
  return launch_wino(p, false, rw_s(stream));
}

extern "C" int rw_conv3x3_wino_to_rgb_f32(const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch,
                                          int h, int w, float w_scale, const rw_conv_epilogue* ep,
                                          const rw_rgb_epilogue* rgb, rw_stream_t stream) {
  RW_CHECK_ARG(x && uf && rgb && rgb->weight && rgb->style && rgb->out && batch > 0 && in_ch > 0 && out_ch > 0);
  RW_CHECK_ARG(h > 0 && w > 0 && (!ep || ((!ep->noise || ep->noise_w) && (!ep->act || ep->bias))));
  if (out_ch != 32 || !wino_shape_ok(out_ch, in_ch, h, w)) return RW_ERR_UNSUPPORTED;   // one wave pair holds all channels
  WinoProblem p;
  wino_fill(p, x, uf, y, batch, in_ch, out_ch, h, w, w_scale, ep);
  p.rgb_weight = rgb->weight; p.rgb_style = rgb->style; p.rgb_bias = rgb->bias; p.rgb_skip = rgb->skip;
  p.rgb_out = rgb->out; p.rgb_scale = rgb->scale;
  return launch_wino(p, true, rw_s(stream));
}
