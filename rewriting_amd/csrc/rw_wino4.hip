// hipcc-flags: -fno-slp-vectorize
// (the SLP vectoriser pairs the scalar row pass of the input transform and pays for it in register moves)
// Stride-1 3x3 modulated convolution (DemodulatedConv2dF, utils/stylegan2/models.py:313-329) by Winograd
// F(4x4, 3x3) in fp32: 36 multiplications per (in-channel, out-channel) pair and 4x4 output tile instead of 144 --
// 4x fewer matrix FLOPs than the direct sum, 1.78x fewer than F(2x2,3x3) (rw_wino.hip).
//
// Accuracy: the transforms carry the constants 4, 5, 8 and 1/24, and fp32 rounding grows with them: measured 4e-6 to
// 9e-6 of the output range per layer (32 - 512 channels) against 2e-7 to 6e-7 for F(2x2,3x3) and the direct sum.
// That is inside the path's image tolerance (1e-3 L-inf) and outside what the key statistics and the solve are held
// to, so the host side selects these kernels for image generation only -- inside the un-hooked forward of a whole
// generator, or with RW_CONV_ALGO=winograd4 -- and never for a hooked or sliced model.
//
//   U[xi][o][i] = (G g G^T)[xi]      6x6 per (o, i), once per weight version      (rw_pack_conv_weight_wino4_f32)
//   V[xi][i][t] = (B^T d B)[xi]      6x6 input tile d (stride 4)
//   M[xi][o][t] = sum_i U[xi][o][i] V[xi][i][t]          36 independent GEMMs -> v_mfma_f32_16x16x4_f32
//   Y[o][t]     = A^T M A            4x4 outputs, then the fused epilogue (demod, noise, bias, leaky-ReLU)
//
// Kernel design -- no V in LDS.  The B operand of v_mfma_f32_16x16x4_f32 puts element (k, n) in lane 16 k + n: with
// k = input channel of the k-quad and n = tile, lane (k, n) needs V[all 36 xi][channel k][tile n] -- exactly the
// output of the input transform of ONE (tile, channel) item.  So every lane transforms its own item in registers
// (12 LDS reads of the raw patch, ~150 VALU instructions) and the results ARE the wave's B operands of that k-quad:
// no transformed buffer, no second barrier, no LDS traffic for B.  A wave holds all 36 points of 16 out-channels x
// 16 tiles (144 accumulator registers), so the output transform is lane-local too and a lane ends up with a 4x4
// pixel block of four channels: 16-byte stores, 256 contiguous bytes per 16 lanes.  The two waves of a SIMD (two
// workgroups per CU) alternate by themselves: while one transforms (VALU, LDS) the other issues its 36 MFMAs.
// Waves of a workgroup that differ only in out-channels (WGM) repeat the transform; the VALU is otherwise idle.
//
// A workgroup = WGM out-channel blocks of 16 x WGN tile rows of 16 tiles (4 x 64 pixels each) and walks a run of
// `gpw` 64-pixel groups along x as one pipeline of 4-channel intervals: interval v computes on Rs[v & 1], stores
// the patch of interval v + 1 (fetched during v - 1, registers) into Rs[(v + 1) & 1] and fetches v + 2; one barrier
// per interval.  Weights: uf[o / 16][i / 4][xi / 4][lane = 16 (i % 4) + o % 16][xi % 4], copied to LDS one interval
// ahead by global_load_lds and read from there (one ds_read_b128 per four points).
#include "rw_common.h"
#include <stdlib.h>

// Timing ablations (build a second library with -DW4_ABL=<bits>; results are WRONG when a bit is set): 1 = no input
// transform arithmetic, 2 = no patch fetch / staging, 4 = no weight copies, 8 = no output transform / stores,
// 16 = no barriers in the loop; H16 kernels: 32 = no operand split (the three conversions per value), 64 = no MFMAs,
// 128 = no duplication of the weight words.
#ifndef W4_ABL
#define W4_ABL 0
#endif
#ifndef W4_SETPRIO
#define W4_SETPRIO 0      // A/B builds: wave priority (1..3) while the 36 MFMAs of a k-quad are issued
#endif

typedef float w4_f32x4 __attribute__((ext_vector_type(4)));
template <int N> struct w4_int { static constexpr int value = N; };

// Order of the 36 transform points in the packed weights (position -> natural index xi = 6 row + column): rows 0, 1, 2
// first, then the second half in the order the point-split kernels want it -- positions 20..35 = rows 5, 3, 4 as far as
// sixteen floats go, positions 18, 19 = the last two columns of row 4 -- so that BOTH halves read four whole quads
// (at quad 5 WM) and one half quad (quad 4, components 2 WM, 2 WM + 1) and their local points 6 a + b mean (row, column)
// = ((0, 1, 2)[a], b) resp. ((5, 3, 4)[a], b): one instruction stream for both waves.  Everything that reads or writes
// the packed buffer goes through this function.
__host__ __device__ constexpr int w4_nat(int pos) { return pos < 18 ? pos : (pos < 26 ? pos + 10 : pos - 8); }

struct Wino4Problem {
  const float* x; const float* uf; float* y;
  const float* style; const float* demod; const float* noise; const float* noise_w; const float* bias;
  int batch, in_ch, out_ch, h, w;
  int groups_x, groups_y, gpw;
  float w_scale;
  int act;
  // ToRGB in the epilogue (conv_wino36_rgb_kernel, out_ch == 32): see rw_rgb_epilogue
  const float* rgb_weight; const float* rgb_style; const float* rgb_bias; const float* rgb_skip; float* rgb_out;
  float rgb_scale;
  const float* post;            // UP: (batch x real out_ch) factor on the result (the next layer's style), nullable
  // H16 kernels (operands split into f16 pairs, see conv_wino36b_body): the bound of the input map (RW_BOUND_LANES floats,
  // rw_common.h); nullable: the slots + bound of the result -- the next layer's x_amax; 1 / (the packed weights' scale)
  const float* x_amax; float* y_amax;
  float u_inv;
};

#define W4_PC 66                // patch columns: 64 + 2
#define W4_RS 72                // row pitch of the raw patch in LDS (floats)

__device__ __forceinline__ int w4_xcd_remap(int id, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, slot = id >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// 1-D input transform B^T (Lavin & Gray, points 0, +-1, +-2, inf), in place
__device__ __forceinline__ void w4_bt(float& d0, float& d1, float& d2, float& d3, float& d4, float& d5) {
  const float t0 = 4.f * d0 - 5.f * d2 + d4;
  const float p = d4 - 4.f * d2, q = d3 - 4.f * d1;
  const float r = d4 - d2, s = 2.f * (d3 - d1);
  const float t5 = 4.f * d1 - 5.f * d3 + d5;
  d0 = t0; d1 = p + q; d2 = p - q; d3 = r + s; d4 = r - s; d5 = t5;
}

// the same on two columns at once (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: adjacent columns of a patch row come out
// of the LDS reads as adjacent registers)
typedef float w4_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned w4_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned w4_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 w4_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 w4_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void w4_bt2(w4_f32x2& d0, w4_f32x2& d1, w4_f32x2& d2, w4_f32x2& d3, w4_f32x2& d4,
                                       w4_f32x2& d5) {
  const w4_f32x2 t0 = 4.f * d0 - 5.f * d2 + d4;
  const w4_f32x2 p = d4 - 4.f * d2, q = d3 - 4.f * d1;
  const w4_f32x2 r = d4 - d2, s = 2.f * (d3 - d1);
  const w4_f32x2 t5 = 4.f * d1 - 5.f * d3 + d5;
  d0 = t0; d1 = p + q; d2 = p - q; d3 = r + s; d4 = r - s; d5 = t5;
}

// The same on ONE row held as three column pairs P0 = (e0,e1), P1 = (e2,e3), P2 = (e4,e5): six packed instructions
// instead of thirteen scalar ones -- the packed operand selectors (op_sel) broadcast a half to both lanes, the
// constants ride in register pairs:
//   (t0,t5) = 4 P0 - 5 P1 + P2     (p,r) = e4 + (-4,-1) e2     (q,s) = e3 + (-4,-1) e1
//   (t1,t2) = p + (q,-q)           (t3,t4) = r + (2,-2) s
__device__ __forceinline__ void w4_bt_row(const w4_f32x2 P0, const w4_f32x2 P1, const w4_f32x2 P2, float (&t)[6]) {
  const w4_f32x2 K1 = {-4.f, -1.f}, K2 = {2.f, -2.f}, K3 = {1.f, -1.f};
  const w4_f32x2 t05 = 4.f * P0 + (P2 - 5.f * P1);
  const w4_f32x2 pr = w4_f32x2{P2[0], P2[0]} + K1 * w4_f32x2{P1[0], P1[0]};
  const w4_f32x2 qs = w4_f32x2{P1[1], P1[1]} + K1 * w4_f32x2{P0[1], P0[1]};
  const w4_f32x2 t12 = w4_f32x2{pr[0], pr[0]} + K3 * w4_f32x2{qs[0], qs[0]};
  const w4_f32x2 t34 = w4_f32x2{pr[1], pr[1]} + K2 * w4_f32x2{qs[1], qs[1]};
  t[0] = t05[0]; t[1] = t12[0]; t[2] = t12[1]; t[3] = t34[0]; t[4] = t34[1]; t[5] = t05[1];
}

template <int WGM, int WGN>
__global__ void __launch_bounds__(256, 2) conv_wino36_kernel(const Wino4Problem p) {
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  constexpr int IC = 4;                           // channels per interval = one k-quad
  constexpr int PR = 4 * WGN + 2;                 // patch rows
  constexpr int NPOS = PR * W4_PC;
  constexpr int PSLOT = (NPOS + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Rs[2][IC][PR][W4_RS];
  __shared__ __attribute__((aligned(16))) float Us[3][WGM][9 * 256];      // weights of one k-quad, A-fragment order
  __shared__ float Ct[2][16 * WGM];               // [0] w_scale * demod, [1] bias

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int lk = lane >> 4, lt = lane & 15;       // channel of the k-quad / tile column (A: k / out-channel row)

  const int local = w4_xcd_remap(blockIdx.x, gridDim.x);
  const int o_tiles = p.out_ch / (16 * WGM);
  const int runs_x = p.groups_x / p.gpw;
  const int ot = local % o_tiles;
  int pg = local / o_tiles;
  const int run = pg % runs_x; pg /= runs_x;
  const int gy = pg % p.groups_y;
  const int ib = pg / p.groups_y;
  const int o0 = ot * 16 * WGM;
  const int y0 = gy * 4 * WGN, gx0 = run * p.gpw;
  const int64_t hw = (int64_t)p.h * p.w;
  const float* xb = p.x + (int64_t)ib * p.in_ch * hw;
  const float* st = p.style ? p.style + (int64_t)ib * p.in_ch : nullptr;
  const int NC = p.in_ch / IC;
  const int VT = p.gpw * NC;

  if (tid < 16 * WGM) {
    const int o = o0 + tid;
    Ct[0][tid] = p.demod ? p.demod[(int64_t)ib * p.out_ch + o] * p.w_scale : p.w_scale;
    Ct[1][tid] = p.act ? p.bias[o] : 0.f;
  }

  // ---- raw patch: buffer loads (out-of-image positions read 0), style applied on the way into LDS
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)((int64_t)p.in_ch * hw * 4), 0x00020000);
  int xoff[PSLOT], xlds[PSLOT];
#pragma unroll
  for (int sl = 0; sl < PSLOT; ++sl) {
    const int pos = tid + 256 * sl;
    const int r = pos / W4_PC, c = pos - r * W4_PC;
    xlds[sl] = pos < NPOS ? r * W4_RS + c : W4_RS - 1;      // spare slots land in a padding column
  }
  auto set_group = [&](int g) __attribute__((always_inline)) {
    const int x0 = (gx0 + g) * 64;
#pragma unroll
    for (int sl = 0; sl < PSLOT; ++sl) {
      const int pos = tid + 256 * sl;
      const int r = pos / W4_PC, c = pos - r * W4_PC;
      const int iy = y0 - 1 + r, ix = x0 - 1 + c;
      const bool ok = pos < NPOS && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
      xoff[sl] = ok ? (iy * p.w + ix) * 4 : 0x7fffffff;
    }
  };
  float xreg[PSLOT][IC];
  float sty[IC];
  const int hw4 = (int)hw * 4;
  auto fetch = [&](int fg, int fc) __attribute__((always_inline)) {       // interval (group fg, chunk fc); uniform
    if (fc == 0) set_group(fg);
    const int i0 = fc * IC;
#pragma unroll
    for (int ic = 0; ic < IC; ++ic)
#pragma unroll
      for (int sl = 0; sl < PSLOT; ++sl)
        xreg[sl][ic] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xsrc, xoff[sl], (i0 + ic) * hw4, 0));
#pragma unroll
    for (int ic = 0; ic < IC; ++ic) sty[ic] = st ? st[(i0 + ic) < p.in_ch ? i0 + ic : 0] : 1.0f;
  };
  auto stash = [&](int rbuf) __attribute__((always_inline)) {
#pragma unroll
    for (int ic = 0; ic < IC; ++ic)
#pragma unroll
      for (int sl = 0; sl < PSLOT; ++sl)
        (&Rs[0][0][0][0])[(rbuf * IC + ic) * PR * W4_RS + xlds[sl]] = xreg[sl][ic] * sty[ic];
  };

  // ---- operands
  // Weights go global -> LDS directly (global_load_lds_dwordx4: no registers, and -- the point -- the A operands are
  // then read with ds_read, which waits on lgkmcnt: a weight load in registers would queue behind the patch fetch in
  // the in-order vmcnt FIFO and stall every interval on HBM latency).  9 KB per 16 out-channels and k-quad, the
  // 9 * WGM one-KB pieces dealt round-robin to the four waves; LDS image = global image.
  const int kq_total = p.in_ch >> 2;
  const int a_lane = lane * 4;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  auto uload = [&](int ubuf, int kq) __attribute__((always_inline)) {
    for (int j = wave; j < 9 * WGM; j += 4) {
      const int ob = j / 9, q = j - 9 * ob;
      const float* src = p.uf + ((int64_t)((o0 >> 4) + ob) * kq_total + kq) * (9 * 256) + q * 256 + a_lane;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&Us[ubuf][ob][q * 256], 16, 0, 0);
    }
  };
  // this lane's item: channel lk of the k-quad, tile (row wn, column lt): patch rows 4 wn .., columns 4 lt ..
  const float* item = &Rs[0][lk][4 * wn][4 * lt];

  w4_f32x4 acc[36];
#pragma unroll
  for (int xi = 0; xi < 36; ++xi) acc[xi] = w4_f32x4{0.f, 0.f, 0.f, 0.f};

  const float noise_w = p.noise ? p.noise_w[0] : 0.f;

  // one interval: transform this lane's item of Rs[buf], then the 36 MFMAs of k-quad kq
  auto compute = [&](int buf, int ubuf) __attribute__((always_inline)) {
    const float* base = &Us[ubuf][wm][a_lane];
    // column pass on column pairs (packed), row pass scalar on the halves of the pairs: no register shuffles
    w4_f32x2 c2[6][3];
    const float* src = item + buf * IC * PR * W4_RS;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const w4_f32x4 lo = *reinterpret_cast<const w4_f32x4*>(src + r * W4_RS);
      c2[r][0] = w4_f32x2{lo[0], lo[1]};
      c2[r][1] = w4_f32x2{lo[2], lo[3]};
      c2[r][2] = *reinterpret_cast<const w4_f32x2*>(src + r * W4_RS + 4);
    }
    float d[6][6];
    if (!(W4_ABL & 1)) {
#pragma unroll
      for (int cp = 0; cp < 3; ++cp) w4_bt2(c2[0][cp], c2[1][cp], c2[2][cp], c2[3][cp], c2[4][cp], c2[5][cp]);
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      if (!(W4_ABL & 1)) {
        w4_bt_row(c2[a][0], c2[a][1], c2[a][2], d[a]);                                            // rows: (B^T d) B
      } else {
        d[a][0] = c2[a][0][0]; d[a][1] = c2[a][0][1]; d[a][2] = c2[a][1][0]; d[a][3] = c2[a][1][1];
        d[a][4] = c2[a][2][0]; d[a][5] = c2[a][2][1];
      }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const w4_f32x4 a = *reinterpret_cast<const w4_f32x4*>(base + q * 256);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int xi = w4_nat(4 * q + e);
        acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], d[xi / 6][xi % 6], acc[xi], 0, 0, 0);
      }
    }
  };

  // epilogue of one group: lane-local output transform; acc[6 a + b][j] = M[a][b] of out-channel
  // o0 + 16 wm + 4 lk + j, tile (row wn, column lt) -> pixels (y0 + 4 wn .. +3, x0 + 4 lt .. +3)
  auto group_epilogue = [&](int g) __attribute__((always_inline)) {
    const int oy = y0 + 4 * wn, ox = (gx0 + g) * 64 + 4 * lt;
    w4_f32x4 nz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) nz[r] = w4_f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.noise) {
      const float* np = p.noise + (int64_t)ib * hw + (int64_t)oy * p.w + ox;
#pragma unroll
      for (int r = 0; r < 4; ++r) nz[r] = *reinterpret_cast<const w4_f32x4*>(np + (int64_t)r * p.w) * noise_w;
    }
    const float* ct = &Ct[0][16 * wm + 4 * lk];
    float* yb = p.y + ((int64_t)ib * p.out_ch + o0 + 16 * wm + 4 * lk) * hw + (int64_t)oy * p.w + ox;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float scale = ct[j], bias = ct[16 * WGM + j];
      // A^T M: columns b = 0..5 -> 4 rows
      float t[4][6];
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const float m0 = acc[b][j], m1 = acc[6 + b][j], m2 = acc[12 + b][j], m3 = acc[18 + b][j], m4 = acc[24 + b][j],
                    m5 = acc[30 + b][j];
        const float s1 = m1 + m2, s2 = m1 - m2, s3 = m3 + m4, s4 = m3 - m4;
        t[0][b] = m0 + s1 + s3;
        t[1][b] = s2 + 2.f * s4;
        t[2][b] = s1 + 4.f * s3;
        t[3][b] = s2 + 8.f * s4 + m5;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = t[r][1] + t[r][2], s2 = t[r][1] - t[r][2], s3 = t[r][3] + t[r][4], s4 = t[r][3] - t[r][4];
        w4_f32x4 v = {t[r][0] + s1 + s3, s2 + 2.f * s4, s1 + 4.f * s3, s2 + 8.f * s4 + t[r][5]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float u = v[k] * scale + nz[r][k];
          if (p.act) {
            u += bias;
            u = ((u > 0.f) ? u : u * 0.2f) * 1.4142135623730951f;
          }
          v[k] = u;
        }
        *reinterpret_cast<w4_f32x4*>(yb + (int64_t)j * hw + (int64_t)r * p.w) = v;
      }
    }
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) acc[xi] = w4_f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- prologue: Rs[0] = interval 0, registers = interval 1
  fetch(0, 0);
  uload(0, 0);
  uload(1, 1 % NC);
  stash(0);
  fetch(1 / NC, 1 % NC);
  __syncthreads();

  // Synchronisation of an interval: the staging writes of this wave (lgkmcnt) and a raw s_barrier.  The loads issued
  // in this interval -- the patch of v + 2 (registers) and the weights of v + 2 (LDS) -- stay in flight ACROSS the
  // barrier: the first is waited for by the staging of the next interval (which, with LDS-direct loads pending,
  // the compiler turns into vmcnt(0): it also retires this wave's weight pieces of v + 2, an interval before the
  // barrier that publishes them).  __syncthreads() would drain both at every barrier (measured: 33 % + 14 %).
  int c = 0, g = 0, ub = 0;                         // ub = v % 3
  int fg = 2 / NC, fc = 2 % NC;                     // (group, chunk) of interval v + 2
  for (int v = 0; v < VT; ++v) {
    const int buf = v & 1;
    const int ub2 = ub == 0 ? 2 : ub - 1;           // (v + 2) % 3
    if (!(W4_ABL & 2)) {
      stash(buf ^ 1);                               // interval v + 1 (the last one: a harmless refill)
      fetch(fg, fc);                                // interval v + 2; past the run: legal addresses, never read
    }
    if (!(W4_ABL & 4)) uload(ub2, c + 2 < NC ? c + 2 : c + 2 - NC);   // weights of interval v + 2 (same slices for every group)
    compute(buf, ub);
    if (c == NC - 1) {
      if (!(W4_ABL & 8) || acc[0][0] == 12345.f) group_epilogue(g);
      c = 0; ++g;
    } else { ++c; }
    if (++fc == NC) { fc = 0; ++fg; }
    ub = ub == 2 ? 0 : ub + 1;
    if (!(W4_ABL & 16)) {
      __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0), vmcnt / expcnt untouched
      __builtin_amdgcn_s_barrier();
    }
  }
}

// ---------------------------------------------------------------------------------------
// Second version: EVERYTHING the loop reads arrives by LDS-direct loads, two intervals ahead.
//
// The first version's interval time equals one memory latency: the patch of interval v + 1 is fetched into
// registers during interval v - 1, and with LDS-direct loads pending the compiler turns the wait for any register
// load into vmcnt(0) -- every interval drains the queue.  Worse, it orders every ds_read of an array that LDS-direct
// loads write behind ALL of them (vmcnt(0) in front of the first read of an interval), and with one array per ring
// slot it serialises the loads themselves.  So the loads are issued from inline assembly -- the compiler does not
// know they exist -- and every vmcnt wait in the loop is written by hand.  The loop has no ordinary vector load:
//   * patch: `buffer_load_dword ... lds` (out-of-image lanes write 0: the zero padding), the (4 WGN + 2) x 66 window
//     of a channel as a flat image of pitch 68, 20 one-wave pieces per channel; wave w copies channel w / 2, pieces
//     10 (w % 2) .. + 9.  The style factor is applied in the transform (packed multiplies of the 18 input pairs)
//     from a table in LDS;
//   * weights: `global_load_lds_dwordx4`, 18 one-KB pieces per interval dealt to the eight waves;
//   * noise of a group: four dword pieces per wave into a wave-private strip, issued FIRST in the group's second-to-last
//     interval, so the counted wait at the end of that interval retires them.
// Rings of three buffers (interval v computes on ring[v % 3] while v + 1 is complete or landing and v + 2 is being
// requested); synchronisation per interval = `s_waitcnt vmcnt(N) lgkmcnt(0)` with N = the pieces this wave issued in
// THIS interval (they stay in flight across the barrier) + raw `s_barrier`.  Stores of a group's epilogue are younger
// than the interval's pieces and count in vmcnt: + 16 in those intervals.
// One workgroup of 512 threads per CU (117 KB of LDS): WGM = 2 out-channel blocks of 16 x WGN = 4 tile rows -- the
// weights of an interval serve four waves, the patch halo is 2 rows in 18.
// ---------------------------------------------------------------------------------------
#define W4B_PITCH 68
#ifndef W4_NTS
#define W4_NTS 0          // non-temporal stores of the output map (A/B builds)
#endif
#if W4_NTS
#define W4_STORE(ptr, val) __builtin_nontemporal_store((val), reinterpret_cast<w4_f32x4*>(ptr))
#else
#define W4_STORE(ptr, val) (*reinterpret_cast<w4_f32x4*>(ptr) = (val))
#endif
#ifndef W4C_PREFETCH
#define W4C_PREFETCH 0        // raw window of the next interval read during the MFMAs (36 more live registers)
#endif
typedef int w4_i32x4 __attribute__((ext_vector_type(4)));
// LDS-direct loads as inline assembly (see the note on the compiler above): M0 = LDS byte address of the wave's 64 x
// size destination, lane L lands at + L * size.
__device__ __forceinline__ void w4_dma_buffer_b32(unsigned lds_addr, int voffset, w4_i32x4 rsrc, int soffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}
__device__ __forceinline__ void w4_dma_global_b128(unsigned lds_addr, const void* gptr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_addr), "v"(gptr) : "memory");
}
// the same with a scalar base and a 32-bit lane offset: no 64-bit vector address arithmetic per piece
__device__ __forceinline__ void w4_dma_global_b128_s(unsigned lds_addr, int voffset, const void* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds_addr), "v"(voffset), "s"(sbase)
               : "memory");
}
__device__ __forceinline__ void w4_dma_global_b32(unsigned lds_addr, const void* gptr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" :: "s"(lds_addr), "v"(gptr) : "memory");
}

// WGN tile rows per workgroup (2 WGN waves); UDEPTH = slots of the weight ring.  <4, 3>: one 512-thread workgroup per CU,
// everything two intervals ahead -- but its eight waves pass one barrier per interval, so the two waves of a SIMD run
// in lockstep (both transform, then both multiply).  <2, 2>: two 256-thread workgroups per CU (77 KB of LDS each)
// that drift apart, patch two intervals ahead, weights one (they come from L2).
//
// UP = the same machinery computing a stride-2 transposed 3x3 convolution FOLLOWED BY the 4x4 blur (pad 1,1), noise,
// bias and leaky ReLU of an upsampling StyledConv in one pass (conv_up_wino36_kernel).  conv_transpose (stride 2)
// then a 4-tap FIR is a stride-2 transposed convolution with the 6x6 kernel w (*) k, and each of its four
// output-parity phases is a 3x3 'same' convolution of the INPUT map: out[2m + p] = sum_a x[m - 1 + a] g[2 - 2a + p],
// g[t] = sum_b k'[b] w[t - 1 + b].  The four phases are virtual out-channels v = 4 o + 2 py + px of a stride-1
// F(4x4,3x3) problem (rw_pack_conv_transpose_blur_weight_wino4_f32 composes and transforms the weights); a lane's
// four accumulator components j are then the four phases of ONE channel, its 4x4 tile an 8x8 block of output pixels
// stored as 32-byte row segments.  The (2H+1) x (2W+1) map between transposed convolution and blur is never
// written or read (8.6 GB each way at 32 x 1024^2 x 64 images), there are no border strips (the composed phases are
// exact 'same' convolutions), and the blur pass disappears -- at 1.44x the matrix work of rw_upwino.hip's F(2,2),
// which pays where the blur's traffic outweighs it: few channels, large maps.
//
// MODE 2 (conv_wino36_rgb_kernel) = the last styled convolution of the generator with ToRGB in its epilogue
// (models.py:639-655; out_ch == 32 = one workgroup tile): every lane multiplies its activated outputs by the
// modulated 1x1 ToRGB weights of its four channels, the sums over the wave's 16 channels go by a reduce-scatter over
// the four 16-lane groups (lane (lk, lt) ends with row lk of its tile), the two waves that hold the other 16
// channels of the same pixels exchange halves through LDS, and the feature map -- which nothing else reads -- is
// never written.
// STYLE = false: the input map already carries this layer's style (its producer multiplied it in: `post` of the
// UP epilogue, rw_blur_noise_act_scaled_f32) -- 18 packed multiplies per item less in the loop.
// PS = the two out-channel waves of a tile row split the 36 POINTS instead of the 32 out-channels: wave wm computes rows
// 3 wm .. 3 wm + 2 of B^T d B -- half of the column pass (6 / 7 instead of 13 packed instructions per column pair), three
// of the six row passes: 36-39 instead of 75 packed instructions per k-quad, the input transform no longer done twice per
// workgroup -- and multiplies them with the weights of BOTH 16-channel blocks (the same 36 MFMAs and 144 accumulators).
// The output transform is linear in the points, so each wave transforms its 18 points of a block to a partial 4 x 4
// tile; per group and accumulator component j the waves swap the partial of the partner's block through the weight
// slot that has just been consumed (16 floats per lane, 16 KB per workgroup and round, two raw barriers per round)
// and add: wave wm ends, as before, with the tiles of block wm.
// The column pass of the three rows a wave of the point split computes, written ONCE for both waves (two copies of the
// loop behind a branch on the wave id cost the register allocator 90 spilled registers): e0, e2, e4 = patch rows WM,
// WM + 2, WM + 4 (the single row: 0 or 5), d1..d4 = patch rows 1..4 (the pair rows: 1, 2 or 3, 4) with the wave-uniform
// coefficients (cA, cB, cC) = (4, 1, 4) or (1, 2, 2): the same values as w4_bt2 computes (scaling by 2 commutes with
// rounding).  Local row order: (0, 1, 2) resp. (5, 3, 4) -- see w4_nat.
__device__ __forceinline__ void w4_bt2_rows(const w4_f32x2 e0, const w4_f32x2 e2, const w4_f32x2 e4, const w4_f32x2 d1,
                                            const w4_f32x2 d2, const w4_f32x2 d3, const w4_f32x2 d4, float cA, float cB,
                                            float cC, w4_f32x2& o0, w4_f32x2& o1, w4_f32x2& o2) {
  o0 = 4.f * e0 - 5.f * e2 + e4;
  const w4_f32x2 p = d4 - cA * d2, q = cB * d3 - cC * d1;
  o1 = p + q; o2 = p - q;
}
// one row of (A^T M) A: six values -> four
__device__ __forceinline__ void w4_at_row(const float (&t)[6], float (&v)[4]) {
  const float s1 = t[1] + t[2], s2 = t[1] - t[2], s3 = t[3] + t[4], s4 = t[3] - t[4];
  v[0] = t[0] + s1 + s3; v[1] = s2 + 2.f * s4; v[2] = s1 + 4.f * s3; v[3] = s2 + 8.f * s4 + t[5];
}

// ---------------------------------------------------------------------------------------
// H16 (round 4): the 36 GEMMs on the 16-bit matrix pipe, fp32-equivalent by an exact operand split.
//
// fp32 MFMA on gfx950 runs at the vector rate (1/16 of the f16 rate) and shares the vector lanes
// (scripts/probe/mfma_valu_probe.hip), so the fp32 kernel above spends half its time in the matrix instructions and
// cannot hide the input transform behind them.  Here every transformed operand is written as a sum of two f16 numbers,
//     V 2^eV = Vh + Vl,   U 2^eU = Uh + Ul      (round to nearest twice: |V 2^eV - Vh - Vl| <= 2^-22 |V 2^eV|),
// and v_mfma_f32_16x16x16_f16 accumulates all FOUR products in fp32: its k index within a lane group holds
// A = [Uh, Ul, Uh, Ul], B = [Vh, Vh, Vl, Vl] of ONE input channel -- the lane <-> (channel, tile) / (channel, out-channel)
// assignment of v_mfma_f32_16x16x4_f32 carries over unchanged, so does everything around the loop (rings, pieces,
// waits, epilogues).  The powers of two keep the operands inside f16's normal range: eU from max |U| at pack time (read
// back by the host once per weight version and passed by value), eV from the bound x_amax >= max |x| that the producer of
// the map left behind (or rw_absmax_f32), the style's largest factor and the transform's gain (<= 100): |V 2^eV| < 25600.  Both leave the result
// through the epilogue's per-channel factor (exact).  Values more than 2^17 below the map's maximum lose low bits of Vl
// (f16 denormals: absolute error <= 2^-25 2^-eV) -- no worse than what fp32 accumulation does to them beside the large terms.
// Per-product error <= 2^-21.x of the product against 2^-24 for an fp32 multiply; measured against float64 the kernel's
// total error is that of the fp32 F(4x4,3x3) kernel (the transforms' constants dominate both).
// Packed weights: uf[o / 16][i / 4][position 0..35 (w4_nat)][lane] of 32-bit words Uh | Ul << 16 -- the size and the
// pieces of rw_pack_conv_weight_wino4_f32, one word per (position, lane) so that consecutive lanes are 4 bytes apart --
// + 4 trailing floats [2^-eU, 2^eU, 0, 0] that no kernel reads: 2^-eU arrives BY VALUE (Wino4Problem::u_inv).
// What it costs: 3 VALU per transformed value (v_cvt_pk_f16_f32, v_fma_mix_f32, v_cvt_pk_f16_f32); the pair (w, w) of a
// weight word costs none -- ds_read2st64_b32 reads the word into both registers (inline assembly, waits tied to the
// operands: scripts/check_asm_loads.py) where two v_mov per word were a fifth of the loop's vector instructions.  What
// it buys: 36 MFMAs of ~17 cycles instead of 32, with vector instructions issuing
// beside them (profiles/r04a_mfma16_valu_probe.jsonl).  The legacy K = 16 instruction takes as long as the K = 32 one;
// the K = 32 form needs 8-channel intervals whose rings do not fit two workgroups' LDS.
// ---------------------------------------------------------------------------------------
// (h, h) and (l, l) of v: h = f16(v), l = f16(v - h)
__device__ __forceinline__ void w4_split16(float v, w4_f16x2& hh, w4_f16x2& ll) {
  if (W4_ABL & 32) { hh = __builtin_bit_cast(w4_f16x2, v); ll = hh; return; }
  hh = __builtin_convertvector(w4_f32x2{v, v}, w4_f16x2);
  float r;
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(v));     // v - (float)h, exact
  ll = __builtin_convertvector(w4_f32x2{r, r}, w4_f16x2);
}
__device__ __forceinline__ w4_f32x4 w4_mfma16(w4_u32x2 ww, w4_f16x2 hh, w4_f16x2 ll, w4_f32x4 acc) {
  const w4_f16x4 a = __builtin_bit_cast(w4_f16x4, ww);         // (w, w): the word of (position, lane) twice
  const w4_f16x4 b = {hh[0], hh[1], ll[0], ll[1]};
  if (W4_ABL & 64) {
    asm volatile("" :: "v"(a), "v"(b));
    return acc;
  }
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);
}
// the word Uh | Ul << 16 of u (already scaled)
__device__ __forceinline__ unsigned w4_pack16(float u) {
  const _Float16 h = (_Float16)u;
  const _Float16 l = (_Float16)(u - (float)h);
  return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}

template <int WGN, int UDEPTH, int MODE, bool STYLE, bool PS = false, bool H16 = false>
__device__ __forceinline__ void conv_wino36b_body(const Wino4Problem& p) {
  static_assert(!PS || (WGN == 2 && UDEPTH == 2), "the point split is written for the <2, 2> shape");
  constexpr bool UP = MODE == 1, RGB = MODE == 2;
  // floats of a wave's share of the noise strips (RGB: also the exchange buffer; UP: the 8 x 128 output-resolution
  // strip of a tile row is shared by the two out-channel waves, 512 floats each)
  constexpr int NSZ = RGB ? 384 : (UP ? 512 : 256);
  constexpr int NST = RGB ? 3 : 16;               // stores of a group's epilogue per wave
  constexpr int WGM = 2;
  constexpr int WAVES = WGM * WGN, THREADS = 64 * WAVES;
  constexpr int PR = 4 * WGN + 2;                 // patch rows
  constexpr int PIECES = (PR * W4B_PITCH + 63) / 64;      // 64-float pieces per channel
  constexpr int PPW = 4 * PIECES / WAVES;         // patch pieces per wave and interval
  static_assert(4 * PIECES % WAVES == 0, "patch pieces divide among the waves");
  constexpr int UPW = (9 * WGM + WAVES - 1) / WAVES;      // weight pieces per wave and interval (at most)
  constexpr int PSZ = 4 * PIECES * 64;            // floats per patch ring slot
  constexpr int USZ = WGM * 9 * 256;              // floats per weight ring slot
  __shared__ __attribute__((aligned(16))) float Ps[3 * PSZ];
  __shared__ __attribute__((aligned(16))) float Us[UDEPTH * USZ];
  __shared__ __attribute__((aligned(16))) float Ns[WAVES * NSZ];
  __shared__ float St[512];
  __shared__ float Ct[2][16 * WGM];
  __shared__ float Cr[3][16 * WGM];
  __shared__ float Red[WAVES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int lk = lane >> 4, lt = lane & 15;

  const int local = w4_xcd_remap(blockIdx.x, gridDim.x);
  const int o_tiles = p.out_ch / (16 * WGM);
  const int runs_x = p.groups_x / p.gpw;
  const int ot = local % o_tiles;
  int pg = local / o_tiles;
  const int run = pg % runs_x; pg /= runs_x;
  const int gy = pg % p.groups_y;
  const int ib = pg / p.groups_y;
  const int o0 = ot * 16 * WGM;
  const int y0 = gy * 4 * WGN, gx0 = run * p.gpw;
  const int64_t hw = (int64_t)p.h * p.w;
  const float* xb = p.x + (int64_t)ib * p.in_ch * hw;
  const int NC = p.in_ch >> 2;
  const int VT = p.gpw * NC;

  // H16: 2^eV rides in the style table (the transform multiplies by it), 2^-(eU + eV) in the per-channel factor
  float in_scale = 1.f, out_scale = 1.f;
  if (H16) {
    float smax = p.style ? 0.f : 1.f;
    if (p.style)
      for (int i = tid; i < p.in_ch; i += THREADS) smax = fmaxf(smax, fabsf(p.style[(int64_t)ib * p.in_ch + i]));
#pragma unroll
    for (int off = 32; off; off >>= 1) smax = fmaxf(smax, __shfl_xor(smax, off));
    if (lane == 0) Red[wave] = smax;
    __syncthreads();
    smax = Red[0];
#pragma unroll
    for (int wv = 1; wv < WAVES; ++wv) smax = fmaxf(smax, Red[wv]);
    const float am = rw_bound_load(p.x_amax) * smax;              // >= max |x style|; |B^T d B| <= 100 am < 2^(e + 7)
    int e = (int)((__float_as_uint(am) >> 23) & 0xff) - 126;      // am < 2^e
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    in_scale = __uint_as_float((unsigned)(127 + 8 - e) << 23);
    out_scale = __uint_as_float((unsigned)(127 + e - 8) << 23) * p.u_inv;
  }
  for (int i = tid; i < p.in_ch; i += THREADS) St[i] = (p.style ? p.style[(int64_t)ib * p.in_ch + i] : 1.0f) * in_scale;
  if (tid < 16 * WGM) {
    const int o = UP ? (o0 + tid) >> 2 : o0 + tid;          // UP: four phases per real channel
    const int real_ch = UP ? p.out_ch >> 2 : p.out_ch;
    Ct[0][tid] = (p.demod ? p.demod[(int64_t)ib * real_ch + o] * p.w_scale : p.w_scale) * out_scale;
    Ct[1][tid] = p.act ? p.bias[o] : 0.f;
    if (RGB) {
      const float sr = p.rgb_scale * p.rgb_style[(int64_t)ib * p.out_ch + o];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) Cr[cc][tid] = sr * p.rgb_weight[cc * p.out_ch + o];
    }
  }
  const float noise_w = p.noise ? p.noise_w[0] : 0.f;

  typedef __attribute__((address_space(3))) float* lds_f;
  const unsigned ps_base = (unsigned)(size_t)(lds_f)Ps, us_base = (unsigned)(size_t)(lds_f)Us,
                 ns_base = (unsigned)(size_t)(lds_f)Ns;
  const unsigned long long xaddr = (unsigned long long)xb;
  const w4_i32x4 xsrc = {(int)(unsigned)xaddr, (int)(unsigned)(xaddr >> 32), (int)((int64_t)p.in_ch * hw * 4),
                         0x00020000};
  const int hw4 = (int)hw * 4;
  // patch pieces of this wave: flat pieces [wave * PPW, + PPW) of the 4 * PIECES of a k-quad (never across a channel)
  static_assert(PIECES % PPW == 0, "a wave's pieces lie in one channel");
  const int pch = (wave * PPW) / PIECES, piece0 = (wave * PPW) % PIECES;
  int xoff[PPW];
  auto set_group = [&](int g) __attribute__((always_inline)) {
    const int x0 = (gx0 + g) * 64;
#pragma unroll
    for (int s = 0; s < PPW; ++s) {
      const int f = 64 * (piece0 + s) + lane;
      const int r = f / W4B_PITCH, c = f - r * W4B_PITCH;
      const int iy = y0 - 1 + r, ix = x0 - 1 + c;
      const bool ok = r < PR && c < W4_PC && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
      xoff[s] = ok ? (iy * p.w + ix) * 4 : 0x7fffffff;
    }
  };
  // the pieces of interval (group fg, k-quad fc) -> ring slot: all at once (prologue), or piece by piece from
  // inside the MFMA loop -- a wave that issues its eleven loads in one burst sits in the issue stage while the
  // address unit works them off, with its MFMAs waiting behind
  int p_soff = 0;
  unsigned p_dst = 0;
  auto pload_begin = [&](int ring, int fg, int fc) __attribute__((always_inline)) {
    if (fc == 0) set_group(fg);
    p_soff = (4 * fc + pch) * hw4;
    p_dst = ps_base + (unsigned)((ring * PSZ + pch * (PIECES * 64) + 64 * piece0) * 4);
  };
  auto pload_piece = [&](int s) __attribute__((always_inline)) { w4_dma_buffer_b32(p_dst + 256 * s, xoff[s], xsrc, p_soff); };
  auto pload = [&](int ring, int fg, int fc) __attribute__((always_inline)) {
    pload_begin(ring, fg, fc);
#pragma unroll
    for (int s = 0; s < PPW; ++s) pload_piece(s);
  };
  const int kq_total = p.in_ch >> 2;
  const int a_lane = lane * 4;
  auto uload = [&](int ring, int kq) __attribute__((always_inline)) {           // pieces wave, wave + WAVES, ...
#pragma unroll
    for (int t = 0; t < UPW; ++t) {
      const int j = wave + WAVES * t;
      if (j < 9 * WGM) {
        const int ob = j / 9, q = j - 9 * ob;
        const float* src = p.uf + ((int64_t)((o0 >> 4) + ob) * kq_total + kq) * (9 * 256) + q * 256;     // uniform
        w4_dma_global_b128_s(us_base + (unsigned)((ring * USZ + ob * (9 * 256) + q * 256) * 4), a_lane * 4, src);
      }
    }
  };
  // noise rows y0 + 4 wn .. + 3, columns x0 .. x0 + 63 of this image -> Ns[wave][64 r + c]
  auto nload = [&](int g) __attribute__((always_inline)) {
    if (UP) {
      // output rows 2 (y0 + 4 wn) .. + 7, columns 128 (gx0 + g) .. + 127 of this image -> Ns[wn][128 r + c]; wave
      // (wm, wn) copies rows 4 wm .. 4 wm + 3 as two 16-byte pieces of two rows each
      const float* sb = p.noise + (int64_t)ib * 4 * hw + (int64_t)(2 * (y0 + 4 * wn) + 4 * wm) * (2 * p.w)
                        + (gx0 + g) * 128;                              // wave-uniform: a scalar base
      const int vo = ((lane >> 5) * (2 * p.w) + 4 * (lane & 31)) * 4;   // bytes
#pragma unroll
      for (int j = 0; j < 2; ++j)
        w4_dma_global_b128_s(ns_base + (unsigned)((wn * 1024 + (4 * wm + 2 * j) * 128) * 4), vo,
                             sb + (int64_t)(2 * j) * (2 * p.w));
      return;
    }
    const float* np = p.noise + (int64_t)ib * hw + (int64_t)(y0 + 4 * wn) * p.w + (gx0 + g) * 64 + lane;
#pragma unroll
    for (int r = 0; r < 4; ++r) w4_dma_global_b32(ns_base + (unsigned)((wave * NSZ + 64 * r) * 4), np + (int64_t)r * p.w);
  };

  w4_f32x4 acc[36];
#pragma unroll
  for (int xi = 0; xi < 36; ++xi) acc[xi] = w4_f32x4{0.f, 0.f, 0.f, 0.f};
  float ymax = 0.f;                 // H16: max |result| of this lane -> p.y_amax

  const int item_off = lk * (PIECES * 64) + (4 * wn) * W4B_PITCH + 4 * lt;
  auto compute = [&](int ring, int uring, int kq, auto spread_tag) __attribute__((always_inline)) {
    constexpr bool SPREAD = decltype(spread_tag)::value != 0;
    const float* base = &Us[uring * USZ + wm * (9 * 256) + a_lane];
    const float* src = &Ps[ring * PSZ + item_off];
    const float sv = St[4 * kq + lk];
    w4_f32x4 a4[3];
    a4[0] = *reinterpret_cast<const w4_f32x4*>(base);
    a4[1] = *reinterpret_cast<const w4_f32x4*>(base + 256);
    a4[2] = *reinterpret_cast<const w4_f32x4*>(base + 512);
    w4_f32x2 c2[6][3];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const w4_f32x4 lo = *reinterpret_cast<const w4_f32x4*>(src + r * W4B_PITCH);
      if (STYLE) {
        c2[r][0] = w4_f32x2{lo[0], lo[1]} * sv;
        c2[r][1] = w4_f32x2{lo[2], lo[3]} * sv;
        c2[r][2] = *reinterpret_cast<const w4_f32x2*>(src + r * W4B_PITCH + 4) * sv;
      } else {
        c2[r][0] = w4_f32x2{lo[0], lo[1]};
        c2[r][1] = w4_f32x2{lo[2], lo[3]};
        c2[r][2] = *reinterpret_cast<const w4_f32x2*>(src + r * W4B_PITCH + 4);
      }
    }
    float d[6][6];
#pragma unroll
    for (int cp = 0; cp < 3; ++cp)
      if (!(W4_ABL & 1)) w4_bt2(c2[0][cp], c2[1][cp], c2[2][cp], c2[3][cp], c2[4][cp], c2[5][cp]);
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      if (!(W4_ABL & 1)) {
        w4_bt_row(c2[a][0], c2[a][1], c2[a][2], d[a]);
      } else {
        d[a][0] = c2[a][0][0]; d[a][1] = c2[a][0][1]; d[a][2] = c2[a][1][0]; d[a][3] = c2[a][1][1];
        d[a][4] = c2[a][2][0]; d[a][5] = c2[a][2][1];
      }
    }
    // weights two quads ahead (an LDS read issued right in front of its MFMAs costs the wave its latency nine times
    // per interval; two waves per SIMD do not hide that)
#if W4_SETPRIO
    __builtin_amdgcn_s_setprio(W4_SETPRIO);
#endif
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const w4_f32x4 a = a4[q % 3];
      if (q + 3 < 9) a4[q % 3] = *reinterpret_cast<const w4_f32x4*>(base + (q + 3) * 256);
      if (SPREAD && !(W4_ABL & 2)) {
        if (2 * q < PPW) pload_piece(2 * q);
        if (2 * q + 1 < PPW) pload_piece(2 * q + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int xi = w4_nat(4 * q + e);
        acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], d[xi / 6][xi % 6], acc[xi], 0, 0, 0);
      }
    }
#if W4_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };

  // H16: the same interval on v_mfma_f32_16x16x16_f16 (see the note above the body).  Points in the packed order
  // (w4_nat): a row of B^T d B is finished right before its first point, every value is split where it is used.  The
  // weight words come in batches of six, one batch ahead: ds_read2st64_b32 with both offsets equal puts the word of
  // (position, lane) into both registers of the operand.
  auto aread = [&](unsigned addr, auto pos_tag, w4_u32x2& dst) __attribute__((always_inline)) {
    constexpr int POS = decltype(pos_tag)::value;
    if (W4_ABL & 128) { dst = w4_u32x2{addr, (unsigned)POS}; return; }
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%2" : "=&v"(dst) : "v"(addr), "n"(POS));
  };
  auto compute16 = [&](int ring, int uring, int kq, auto spread_tag) __attribute__((always_inline)) {
    constexpr bool SPREAD = decltype(spread_tag)::value != 0;
    const unsigned ua = us_base + (unsigned)((uring * USZ + wm * (9 * 256) + lane) * 4);
    const float* src = &Ps[ring * PSZ + item_off];
    const float sv = St[4 * kq + lk];
    w4_f32x2 c2[6][3];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const w4_f32x4 lo = *reinterpret_cast<const w4_f32x4*>(src + r * W4B_PITCH);
      c2[r][0] = w4_f32x2{lo[0], lo[1]} * sv;
      c2[r][1] = w4_f32x2{lo[2], lo[3]} * sv;
      c2[r][2] = *reinterpret_cast<const w4_f32x2*>(src + r * W4B_PITCH + 4) * sv;
    }
    w4_u32x2 aq[2][6];
    aread(ua, w4_int<0>(), aq[0][0]); aread(ua, w4_int<1>(), aq[0][1]); aread(ua, w4_int<2>(), aq[0][2]);
    aread(ua, w4_int<3>(), aq[0][3]); aread(ua, w4_int<4>(), aq[0][4]); aread(ua, w4_int<5>(), aq[0][5]);
#pragma unroll
    for (int cp = 0; cp < 3; ++cp)
      if (!(W4_ABL & 1)) w4_bt2(c2[0][cp], c2[1][cp], c2[2][cp], c2[3][cp], c2[4][cp], c2[5][cp]);
    float d[6][6];
    auto batch = [&](auto bt_tag) __attribute__((always_inline)) {
      constexpr int BT = decltype(bt_tag)::value;
      w4_u32x2 (&cur)[6] = aq[BT & 1];
      if (BT < 5) {
        w4_u32x2 (&nxt)[6] = aq[(BT + 1) & 1];
        aread(ua, w4_int<6 * BT + 6>(), nxt[0]); aread(ua, w4_int<6 * BT + 7>(), nxt[1]);
        aread(ua, w4_int<6 * BT + 8>(), nxt[2]); aread(ua, w4_int<6 * BT + 9>(), nxt[3]);
        aread(ua, w4_int<6 * BT + 10>(), nxt[4]); aread(ua, w4_int<6 * BT + 11>(), nxt[5]);
        // this batch's reads are older than the six just issued
        if (!(W4_ABL & 128))
          asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]) :: "memory");
      } else if (!(W4_ABL & 128)) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]) :: "memory");
      }
      if (SPREAD && !(W4_ABL & 2)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (2 * BT + t < PPW) pload_piece(2 * BT + t);
      }
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const int pos = 6 * BT + e;
        const int xi = w4_nat(pos);
        const int ra = xi / 6;
        // first point of a row in the packed order: positions 0, 6, 12 (rows 0, 1, 2), 18 (row 4), 20 (row 5), 26 (row 3)
        if (pos == 0 || pos == 6 || pos == 12 || pos == 18 || pos == 20 || pos == 26) {
          if (!(W4_ABL & 1)) {
            w4_bt_row(c2[ra][0], c2[ra][1], c2[ra][2], d[ra]);
          } else {
            d[ra][0] = c2[ra][0][0]; d[ra][1] = c2[ra][0][1]; d[ra][2] = c2[ra][1][0]; d[ra][3] = c2[ra][1][1];
            d[ra][4] = c2[ra][2][0]; d[ra][5] = c2[ra][2][1];
          }
        }
        w4_f16x2 hh, ll;
        w4_split16(d[ra][xi % 6], hh, ll);
        acc[xi] = w4_mfma16(cur[e], hh, ll, acc[xi]);
      }
    };
    batch(w4_int<0>()); batch(w4_int<1>()); batch(w4_int<2>()); batch(w4_int<3>()); batch(w4_int<4>()); batch(w4_int<5>());
    static_assert(2 * 6 >= PPW, "two patch pieces per batch cover the wave's share");
  };

  // PS: this wave's 18 points against both 16-channel blocks: acc[18 ob + le], le = 6 a + b local (rows (0, 1, 2)[a] of
  // wave 0, (5, 3, 4)[a] of wave 1).  One instruction stream for both waves: everything that depends on wm is a scalar.
  const float ps_cA = wm ? 1.f : 4.f, ps_cB = wm ? 2.f : 1.f, ps_cC = wm ? 2.f : 4.f;
  const int ps_row = wm * W4B_PITCH;               // patch rows wm, wm + 2, wm + 4 feed the single row
  const int ps_quad = wm * (5 * 256), ps_half = 4 * 256 + 2 * wm;       // weight quads 5 wm .. + 3; half quad
  auto compute_ps = [&](int ring, int uring, int kq, auto spread_tag) __attribute__((always_inline)) {
    constexpr bool SPREAD = decltype(spread_tag)::value != 0;
    const float* base = &Us[uring * USZ + a_lane];                        // block 1: + 9 * 256
    const float* src = &Ps[ring * PSZ + item_off];
    const float sv = St[4 * kq + lk];
    w4_f32x4 a4[2][2];
    a4[0][0] = *reinterpret_cast<const w4_f32x4*>(base + ps_quad);
    a4[0][1] = *reinterpret_cast<const w4_f32x4*>(base + 9 * 256 + ps_quad);
    // rows: [0..2] = patch rows wm, wm + 2, wm + 4; [3..6] = patch rows 1..4
    w4_f32x2 c2[7][3];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const float* rp = r < 3 ? src + ps_row + 2 * r * W4B_PITCH : src + (r - 2) * W4B_PITCH;
      const w4_f32x4 lo = *reinterpret_cast<const w4_f32x4*>(rp);
      if (STYLE) {
        c2[r][0] = w4_f32x2{lo[0], lo[1]} * sv;
        c2[r][1] = w4_f32x2{lo[2], lo[3]} * sv;
        c2[r][2] = *reinterpret_cast<const w4_f32x2*>(rp + 4) * sv;
      } else {
        c2[r][0] = w4_f32x2{lo[0], lo[1]};
        c2[r][1] = w4_f32x2{lo[2], lo[3]};
        c2[r][2] = *reinterpret_cast<const w4_f32x2*>(rp + 4);
      }
    }
    w4_f32x2 hrow[3][3];
#pragma unroll
    for (int cp = 0; cp < 3; ++cp)
      w4_bt2_rows(c2[0][cp], c2[1][cp], c2[2][cp], c2[3][cp], c2[4][cp], c2[5][cp], c2[6][cp], ps_cA, ps_cB, ps_cC,
                  hrow[0][cp], hrow[1][cp], hrow[2][cp]);
    float d[3][6];
#pragma unroll
    for (int a = 0; a < 3; ++a) w4_bt_row(hrow[a][0], hrow[a][1], hrow[a][2], d[a]);
    // four whole quads (local points 0..15), then the half quad (16, 17); weights one quad ahead
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const w4_f32x4 a0 = a4[q & 1][0], a1 = a4[q & 1][1];
      if (q + 1 < 4) {
        a4[(q + 1) & 1][0] = *reinterpret_cast<const w4_f32x4*>(base + ps_quad + (q + 1) * 256);
        a4[(q + 1) & 1][1] = *reinterpret_cast<const w4_f32x4*>(base + 9 * 256 + ps_quad + (q + 1) * 256);
      } else if (q + 1 == 4) {
        const w4_f32x2 h0 = *reinterpret_cast<const w4_f32x2*>(base + ps_half);
        const w4_f32x2 h1 = *reinterpret_cast<const w4_f32x2*>(base + 9 * 256 + ps_half);
        a4[0][0] = w4_f32x4{h0[0], h0[1], 0.f, 0.f};
        a4[0][1] = w4_f32x4{h1[0], h1[1], 0.f, 0.f};
      }
      if (SPREAD && !(W4_ABL & 2)) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
          if (3 * q + t < PPW) pload_piece(3 * q + t);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < (q < 4 ? 4 : 2); ++e) {
        const int le = 4 * q + e;
        acc[le] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], d[le / 6][le % 6], acc[le], 0, 0, 0);
        acc[18 + le] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], d[le / 6][le % 6], acc[18 + le], 0, 0, 0);
      }
    }
  };

  // H16 + PS: as compute_ps; every value is split once and meets the weight words of both blocks.  Local point le of
  // wave wm sits at packed position 20 wm + le (le < 16) resp. 16 + 2 wm + (le - 16): two wave-uniform bases, the
  // position within them an immediate; block 1 is 36 positions further.
  auto compute16_ps = [&](int ring, int uring, int kq, auto spread_tag) __attribute__((always_inline)) {
    constexpr bool SPREAD = decltype(spread_tag)::value != 0;
    const unsigned ua = us_base + (unsigned)((uring * USZ + lane) * 4) + (unsigned)(wm * 20 * 256);
    const unsigned ub = us_base + (unsigned)((uring * USZ + lane) * 4) + (unsigned)(wm * 2 * 256);
    const float* src = &Ps[ring * PSZ + item_off];
    const float sv = St[4 * kq + lk];
    w4_u32x2 aq[2][6];                              // three points x two blocks per batch
    aread(ua, w4_int<0>(), aq[0][0]); aread(ua, w4_int<36>(), aq[0][1]); aread(ua, w4_int<1>(), aq[0][2]);
    aread(ua, w4_int<37>(), aq[0][3]); aread(ua, w4_int<2>(), aq[0][4]); aread(ua, w4_int<38>(), aq[0][5]);
    w4_f32x2 c2[7][3];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const float* rp = r < 3 ? src + ps_row + 2 * r * W4B_PITCH : src + (r - 2) * W4B_PITCH;
      const w4_f32x4 lo = *reinterpret_cast<const w4_f32x4*>(rp);
      c2[r][0] = w4_f32x2{lo[0], lo[1]} * sv;
      c2[r][1] = w4_f32x2{lo[2], lo[3]} * sv;
      c2[r][2] = *reinterpret_cast<const w4_f32x2*>(rp + 4) * sv;
    }
    w4_f32x2 hrow[3][3];
#pragma unroll
    for (int cp = 0; cp < 3; ++cp)
      w4_bt2_rows(c2[0][cp], c2[1][cp], c2[2][cp], c2[3][cp], c2[4][cp], c2[5][cp], c2[6][cp], ps_cA, ps_cB, ps_cC,
                  hrow[0][cp], hrow[1][cp], hrow[2][cp]);
    float d[3][6];
    auto batch = [&](auto bt_tag) __attribute__((always_inline)) {
      constexpr int BT = decltype(bt_tag)::value;
      w4_u32x2 (&cur)[6] = aq[BT & 1];
      if (BT < 5) {
        w4_u32x2 (&nxt)[6] = aq[(BT + 1) & 1];
        constexpr int L0 = 3 * BT + 3, L1 = 3 * BT + 4, L2 = 3 * BT + 5;      // the next batch's local points
        aread(L0 < 16 ? ua : ub, w4_int<L0>(), nxt[0]); aread(L0 < 16 ? ua : ub, w4_int<L0 + 36>(), nxt[1]);
        aread(L1 < 16 ? ua : ub, w4_int<L1>(), nxt[2]); aread(L1 < 16 ? ua : ub, w4_int<L1 + 36>(), nxt[3]);
        aread(L2 < 16 ? ua : ub, w4_int<L2>(), nxt[4]); aread(L2 < 16 ? ua : ub, w4_int<L2 + 36>(), nxt[5]);
        if (!(W4_ABL & 128))
          asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]) :: "memory");
      } else if (!(W4_ABL & 128)) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]) :: "memory");
      }
      if (SPREAD && !(W4_ABL & 2)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (2 * BT + t < PPW) pload_piece(2 * BT + t);
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int le = 3 * BT + e;
        if (le % 6 == 0) w4_bt_row(hrow[le / 6][0], hrow[le / 6][1], hrow[le / 6][2], d[le / 6]);
        w4_f16x2 hh, ll;
        w4_split16(d[le / 6][le % 6], hh, ll);
        acc[le] = w4_mfma16(cur[2 * e], hh, ll, acc[le]);
        acc[18 + le] = w4_mfma16(cur[2 * e + 1], hh, ll, acc[18 + le]);
      }
    };
    batch(w4_int<0>()); batch(w4_int<1>()); batch(w4_int<2>()); batch(w4_int<3>()); batch(w4_int<4>()); batch(w4_int<5>());
  };

  // ---- the output transform of one accumulator component j: Y[r][k] = (A^T M A)[r][k] of this wave's out-channel block
  // (16 wm + 4 lk + j) on its tile.  PS: see the note above the body; `xs` = the weight slot just consumed.
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): LDS-direct loads stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // partial tile of block ob from the three rows this wave holds: L0, L1, L2 = rows (0, 1, 2) resp. (5, 3, 4).  With
  // S = L1 + L2, D = L1 - L2 and R(.) the row pass:  wave 0: Y = (R(L0) + R(S), R(D), R(S), R(D));
  // wave 1: Y = (R(S), 2 R(D), 4 R(S), 8 R(D) + R(L0)) -- one stream, scalar coefficients (1 and 0 are exact).
  const float ps_k0 = wm ? 0.f : 1.f, ps_k1 = wm ? 2.f : 1.f, ps_k2 = wm ? 4.f : 1.f, ps_k3 = wm ? 8.f : 1.f,
              ps_k4 = wm ? 1.f : 0.f;
  auto partial_rows = [&](int j, int ob18, float (&Y)[4][4]) __attribute__((always_inline)) {
    float t0[6], ts[6], td[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const float l0 = acc[ob18 + b][j], l1 = acc[ob18 + 6 + b][j], l2 = acc[ob18 + 12 + b][j];
      t0[b] = l0; ts[b] = l1 + l2; td[b] = l1 - l2;
    }
    float r0[4], rs[4], rd[4];
    w4_at_row(t0, r0); w4_at_row(ts, rs); w4_at_row(td, rd);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      Y[0][k] = rs[k] + ps_k0 * r0[k];
      Y[1][k] = ps_k1 * rd[k];
      Y[2][k] = ps_k2 * rs[k];
      Y[3][k] = ps_k3 * rd[k] + ps_k4 * r0[k];
    }
  };
  auto tile = [&](int j, int xs, float (&Y)[4][4]) __attribute__((always_inline)) {
    if (!PS) {
      float t[4][6];
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const float m0 = acc[b][j], m1 = acc[6 + b][j], m2 = acc[12 + b][j], m3 = acc[18 + b][j], m4 = acc[24 + b][j],
                    m5 = acc[30 + b][j];
        const float s1 = m1 + m2, s2 = m1 - m2, s3 = m3 + m4, s4 = m3 - m4;
        t[0][b] = m0 + s1 + s3;
        t[1][b] = s2 + 2.f * s4;
        t[2][b] = s1 + 4.f * s3;
        t[3][b] = s2 + 8.f * s4 + m5;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) w4_at_row(t[r], Y[r]);
      return;
    }
    // both blocks' partials: block 0 and block 1; mine = block wm, the partner's = block 1 - wm
    float y0[4][4], y1[4][4];
    partial_rows(j, 0, y0);
    partial_rows(j, 18, y1);
    float* xw = &Us[xs * USZ + wave * 1024 + lane * 4];
    const float* xr = &Us[xs * USZ + (wave ^ WGN) * 1024 + lane * 4];
    lds_barrier();                                  // nobody reads what the slot held (weights / the previous round)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      w4_f32x4 o4;
#pragma unroll
      for (int k = 0; k < 4; ++k) { o4[k] = wm ? y0[r][k] : y1[r][k]; Y[r][k] = wm ? y1[r][k] : y0[r][k]; }
      *reinterpret_cast<w4_f32x4*>(xw + r * 256) = o4;
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const w4_f32x4 o4 = *reinterpret_cast<const w4_f32x4*>(xr + r * 256);
#pragma unroll
      for (int k = 0; k < 4; ++k) Y[r][k] += o4[k];
    }
  };

  // UP: lane (lk, lt) of wave (wm, wn) holds the four phases j = 2 py + px of channel o0 / 4 + 4 wm + lk on the tile
  // rows oy .. oy + 3, columns ox .. ox + 3 of the INPUT grid = output rows 2 oy .. + 7, columns 2 ox .. + 7.
  auto up_epilogue = [&](int g, int xs) __attribute__((always_inline)) {
    const int oy = y0 + 4 * wn, ox = (gx0 + g) * 64 + 4 * lt;
    const int W2 = 2 * p.w;
    const int64_t hw2 = 4 * hw;
    const int ch = (o0 >> 2) + 4 * wm + lk;
    float* yb = p.y + ((int64_t)ib * (p.out_ch >> 2) + ch) * hw2 + (int64_t)(2 * oy) * W2 + 2 * ox;
    // the noise of this tile row's 8 x 128 output pixels sits in LDS (nload, one interval earlier): a plain load here
    // would queue behind the patch pieces just issued for interval v + 2 and cost the group a memory latency
    const float* nb = &Ns[wn * 1024 + 8 * lt];
    // leaky ReLU and its gain as max(t, 0.2 t) on values that already carry the gain (act off: slope 1, gain 1)
    const float gain = p.act ? 1.4142135623730951f : 1.f, slope = p.act ? 0.2f : 1.f;
    const float scale = Ct[0][16 * wm + 4 * lk] * gain, bias = Ct[1][16 * wm + 4 * lk] * gain;
    const float nwg = noise_w * gain;
    const float post = p.post ? p.post[(int64_t)ib * (p.out_ch >> 2) + ch] : 1.f;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      float v[2][4][4];
#pragma unroll
      for (int px = 0; px < 2; ++px) tile(2 * py + px, xs, v[px]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t off = (int64_t)(2 * r + py) * W2;
        w4_f32x4 n0 = {0.f, 0.f, 0.f, 0.f}, n1 = n0;
        if (p.noise) {
          n0 = *reinterpret_cast<const w4_f32x4*>(nb + (2 * r + py) * 128) * nwg;
          n1 = *reinterpret_cast<const w4_f32x4*>(nb + (2 * r + py) * 128 + 4) * nwg;
        }
        w4_f32x4 q0 = {v[0][r][0], v[1][r][0], v[0][r][1], v[1][r][1]};
        w4_f32x4 q1 = {v[0][r][2], v[1][r][2], v[0][r][3], v[1][r][3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float u0 = q0[k] * scale + n0[k] + bias, u1 = q1[k] * scale + n1[k] + bias;
          q0[k] = fmaxf(u0, u0 * slope) * post; q1[k] = fmaxf(u1, u1 * slope) * post;
          if (H16) ymax = fmaxf(ymax, fmaxf(fabsf(q0[k]), fabsf(q1[k])));
        }
        W4_STORE(yb + off, q0);
        W4_STORE(yb + off + 4, q1);
      }
    }
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) acc[xi] = w4_f32x4{0.f, 0.f, 0.f, 0.f};
  };

  auto rgb_epilogue = [&](int g, int xs) __attribute__((always_inline)) {
    const int oy = y0 + 4 * wn, ox = (gx0 + g) * 64 + 4 * lt;
    const float gain = p.act ? 1.4142135623730951f : 1.f, slope = p.act ? 0.2f : 1.f;
    w4_f32x4 nz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      nz[r] = p.noise ? *reinterpret_cast<const w4_f32x4*>(&Ns[wave * NSZ + 64 * r + 4 * lt]) * (noise_w * gain)
                      : w4_f32x4{0.f, 0.f, 0.f, 0.f};
    float rp[4][4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int k = 0; k < 4; ++k) rp[r][k][0] = rp[r][k][1] = rp[r][k][2] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int oc = 16 * wm + 4 * lk + j;
      const float scale = Ct[0][oc] * gain, bias = Ct[1][oc] * gain;
      const float cr[3] = {Cr[0][oc], Cr[1][oc], Cr[2][oc]};
      float Y[4][4];
      tile(j, xs, Y);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float (&v)[4] = Y[r];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float u = v[k] * scale + nz[r][k] + bias;
          u = fmaxf(u, u * slope);
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) rp[r][k][cc] += u * cr[cc];
        }
      }
    }
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) acc[xi] = w4_f32x4{0.f, 0.f, 0.f, 0.f};
    // sum over the wave's 16 channels = over its four 16-lane groups, as a reduce-scatter: lane (lk, lt) ends with
    // row lk of the tile
    // (gfx950's v_permlane32_swap / v_permlane16_swap exchange the halves / the odd and even rows of TWO registers in
    // one instruction: A' + B' is then "what I keep" + "what my partner sent" for both sides at once -- 36 swaps and adds
    // where ds_bpermute needed 36 LDS round trips and 72 selects)
    typedef unsigned w4_u32x2 __attribute__((ext_vector_type(2)));
    float q[2][4][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          // lanes 0..31 (lk 0, 1) keep row a, lanes 32..63 keep row a + 2
          const w4_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(rp[a][k][cc]),
                                                              __float_as_uint(rp[a + 2][k][cc]), false, false);
          q[a][k][cc] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
    w4_f32x4 sum[3];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        // even 16-lane rows (lk 0, 2) keep q[0], odd rows keep q[1]
        const w4_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(q[0][k][cc]), __float_as_uint(q[1][k][cc]),
                                                            false, false);
        sum[cc][k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      }
    // the other 16 channels of the same pixels are in wave (1 - wm, wn): wave wm finishes rows 2 wm, 2 wm + 1 and
    // hands the other two over
    const bool mine = (lk >> 1) == wm;
    const int slot = ((lk & 1) * 16 + lt) * 4;
    asm volatile("" ::: "memory");
    if (!mine) {
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) *reinterpret_cast<w4_f32x4*>(&Ns[wave * NSZ + cc * 128 + slot]) = sum[cc];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (mine) {
      const int64_t pix = (int64_t)(oy + lk) * p.w + ox;
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const w4_f32x4 other = *reinterpret_cast<const w4_f32x4*>(&Ns[(wave ^ WGN) * NSZ + cc * 128 + slot]);
        const int64_t off = ((int64_t)ib * 3 + cc) * hw + pix;
        w4_f32x4 o4 = sum[cc] + other + (p.rgb_bias ? p.rgb_bias[cc] : 0.f);
        if (p.rgb_skip) o4 += *reinterpret_cast<const w4_f32x4*>(p.rgb_skip + off);
        *reinterpret_cast<w4_f32x4*>(p.rgb_out + off) = o4;
      }
    }
  };

  auto group_epilogue = [&](int g, int xs) __attribute__((always_inline)) {
    if (UP) { up_epilogue(g, xs); return; }
    if (RGB) { rgb_epilogue(g, xs); return; }
    const int oy = y0 + 4 * wn, ox = (gx0 + g) * 64 + 4 * lt;
    const float gain = p.act ? 1.4142135623730951f : 1.f, slope = p.act ? 0.2f : 1.f;
    w4_f32x4 nz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      nz[r] = p.noise ? *reinterpret_cast<const w4_f32x4*>(&Ns[wave * NSZ + 64 * r + 4 * lt]) * (noise_w * gain)
                      : w4_f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ct = &Ct[0][16 * wm + 4 * lk];
    float* yb = p.y + ((int64_t)ib * p.out_ch + o0 + 16 * wm + 4 * lk) * hw + (int64_t)oy * p.w + ox;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float scale = ct[j] * gain, bias = ct[16 * WGM + j] * gain;
      float Y[4][4];
      tile(j, xs, Y);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        w4_f32x4 v = {Y[r][0], Y[r][1], Y[r][2], Y[r][3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float u = v[k] * scale + nz[r][k] + bias;
          v[k] = fmaxf(u, u * slope);
        }
        if (H16) ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        W4_STORE(yb + (int64_t)j * hw + (int64_t)r * p.w, v);
      }
    }
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) acc[xi] = w4_f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // s_waitcnt immediates (gfx9): vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14; expcnt 7 = no wait
#define W4_WAIT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | 0x70)
  // Pieces that may stay in flight across the barrier of an interval = what this wave issued for interval v + 2:
  // UDEPTH 3: its patch and weight pieces (PPW + UPW or UPW - 1); UDEPTH 2: the patch pieces only (the weights of
  // v + 1 are issued FIRST in the interval and must have landed).  + 16 stores after a group's epilogue.
  const bool full_u = wave + WAVES * (UPW - 1) < 9 * WGM;
  auto sync_interval = [&](bool stores) __attribute__((always_inline)) {
    if (UDEPTH == 2) {
      if (stores) W4_WAIT(PPW + NST); else W4_WAIT(PPW);
    } else if (full_u) {
      if (stores) W4_WAIT(PPW + UPW + NST); else W4_WAIT(PPW + UPW);
    } else {
      if (stores) W4_WAIT(PPW + UPW - 1 + NST); else W4_WAIT(PPW + UPW - 1);
    }
    if (!(W4_ABL & 16)) __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: the tables (ordinary loads and LDS writes) are complete before any LDS-direct load is issued --
  // from here on the compiler does not know about the loads in flight and every vmcnt wait is one of the above
  __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0) lgkmcnt(0)
  if (UDEPTH == 2) {
    uload(0, 0);
    pload(0, 0, 0);
    pload(1, 1 / NC, 1 % NC);
  } else {
    pload(0, 0, 0);
    uload(0, 0);
    pload(1, 1 / NC, 1 % NC);
    uload(1, 1 % NC);
  }
  sync_interval(false);                             // slot 0 landed; slot 1 may still be in flight

  int c = 0, g = 0, ring = 0;
  int fg = 2 / NC, fc = 2 % NC;                     // (group, k-quad) of interval v + 2
  for (int v = 0; v < VT; ++v) {
    const int ring2 = ring == 0 ? 2 : ring - 1;     // (v + 2) % 3
    if (p.noise && c == NC - 2) nload(g);           // older than this interval's pieces: retired by its wait
    static_assert(2 * 9 >= PPW, "two patch pieces per weight quad cover the wave's share");
    if (UDEPTH == 2) {
      if (!(W4_ABL & 4)) uload((v + 1) & 1, c + 1 < NC ? c + 1 : 0);      // weights of interval v + 1
      pload_begin(ring2, fg, fc);                   // past the run: legal addresses, never read
      // ... issues the patch pieces of v + 2 between its MFMAs
      if (H16 && PS) compute16_ps(ring, v & 1, c, w4_int<1>());
      else if (H16) compute16(ring, v & 1, c, w4_int<1>());
      else if (!PS) compute(ring, v & 1, c, w4_int<1>());
      else compute_ps(ring, v & 1, c, w4_int<1>());
    } else {
      if (!(W4_ABL & 2)) pload(ring2, fg, fc);
      if (!(W4_ABL & 4)) uload(ring2, fc);
      if (H16) compute16(ring, ring, c, w4_int<0>());
      else compute(ring, ring, c, w4_int<0>());
    }
    const bool last = c == NC - 1;
    if (last) {
      if (!(W4_ABL & 8) || acc[0][0] == 12345.f) group_epilogue(g, v & 1);
      c = 0; ++g;
    } else { ++c; }
    if (++fc == NC) { fc = 0; ++fg; }
    ring = ring == 2 ? 0 : ring + 1;
    sync_interval(last);
  }
  __builtin_amdgcn_s_waitcnt(0x0070);               // nothing in flight into LDS when the workgroup retires
#undef W4_WAIT
  if (H16 && !RGB && p.y_amax) rw_bound_store_wave(p.y_amax, rw_wave_max(ymax));      // this wave's slot
}

template <int WGN, int UDEPTH>
__global__ void __launch_bounds__(128 * WGN, 4 / WGN) conv_wino36b_kernel(const Wino4Problem p) {
  conv_wino36b_body<WGN, UDEPTH, 0, true>(p);
}
// transposed convolution + blur + noise + bias + leaky ReLU (see UP above): <2, 2> only
__global__ void __launch_bounds__(256, 2) conv_up_wino36_kernel(const Wino4Problem p) {
  conv_wino36b_body<2, 2, 1, true>(p);
}
// the last styled convolution with ToRGB in its epilogue (MODE 2 above): <2, 2> only
__global__ void __launch_bounds__(256, 2) conv_wino36_rgb_kernel(const Wino4Problem p) {
  conv_wino36b_body<2, 2, 2, true>(p);
}
// the same three on an input map that already carries the style (STYLE = false; chosen when the style pointer is null)
__global__ void __launch_bounds__(256, 2) conv_wino36b_ns_kernel(const Wino4Problem p) {
  conv_wino36b_body<2, 2, 0, false>(p);
}
__global__ void __launch_bounds__(256, 2) conv_up_wino36_ns_kernel(const Wino4Problem p) {
  conv_wino36b_body<2, 2, 1, false>(p);
}
__global__ void __launch_bounds__(256, 2) conv_wino36_rgb_ns_kernel(const Wino4Problem p) {
  conv_wino36b_body<2, 2, 2, false>(p);
}
// H16: the six on the 16-bit matrix pipe (see the note above the body), and the point split for the stride-1 convolution
// and the upsampling layer (its vector work per interval halves; what it adds -- the partial tiles' swap per group --
// pays where a group has many intervals: in_ch >= 64, measured)
__global__ void __launch_bounds__(256, 2) conv_wino36h_kernel(const Wino4Problem p) { conv_wino36b_body<2, 2, 0, true, false, true>(p); }
__global__ void __launch_bounds__(256, 2) conv_up_wino36h_kernel(const Wino4Problem p) { conv_wino36b_body<2, 2, 1, true, false, true>(p); }
__global__ void __launch_bounds__(256, 2) conv_wino36h_rgb_kernel(const Wino4Problem p) { conv_wino36b_body<2, 2, 2, true, false, true>(p); }
__global__ void __launch_bounds__(256, 2) conv_wino36h_ps_kernel(const Wino4Problem p) { conv_wino36b_body<2, 2, 0, true, true, true>(p); }
__global__ void __launch_bounds__(256, 2) conv_up_wino36h_ps_kernel(const Wino4Problem p) { conv_wino36b_body<2, 2, 1, true, true, true>(p); }
__global__ void __launch_bounds__(256, 2) conv_wino36h_rgb_ps_kernel(const Wino4Problem p) { conv_wino36b_body<2, 2, 2, true, true, true>(p); }
// <4, 3>: one 512-thread workgroup per CU, 64 tiles per weight slice (RW_W4H_WG8=1; A/B builds)
__global__ void __launch_bounds__(512, 1) conv_wino36h_wg8_kernel(const Wino4Problem p) { conv_wino36b_body<4, 3, 0, true, false, true>(p); }
// point split: 0 = never, 1 = always, default = where in_ch >= 64 (RW_W4H_PS overrides)
static bool w4h_point_split(int in_ch) {
  const char* e = getenv("RW_W4H_PS");
  const int mode = e ? atoi(e) : -1;
  return mode < 0 ? in_ch >= 64 : mode != 0;
}

// The same six with the 36 points split between the two out-channel waves (PS above).  MEASURED FLAT (same box, batch
// 64, ms per 10 steps, split off / on: layers 10-16 173.3 / 172.8, layer 17 107.0 / 109.9, layer 18 + ToRGB 56.8 / 57.9;
// profiles/r03_w4_point_split.log): the ~220 vector cycles it saves per k-quad and wave are given back by what it adds -- a
// seventh patch row read, weights only one quad ahead (16 registers are gone), per group eight raw barriers and the swap
// -- and by what does not shrink: the LDS counters (profiles/r03_pmc_lds_summary_b64.json) show the LDS itself at 25 % of
// its bandwidth (32 B/clk per CU), so the interval is paced by its dependent chain (landed pieces -> barrier -> LDS reads
// -> transform -> MFMAs), not by a throughput that fewer instructions would relieve.  Built only with -DW4_PSPLIT=1
// (RW_W4_PSPLIT=1 then selects it at run time); the kernel tests pass in both modes.
#ifndef W4_PSPLIT
#define W4_PSPLIT 0
#endif
#if W4_PSPLIT
__global__ void __launch_bounds__(256, 2) conv_wino36b_ps_kernel(const Wino4Problem p) { conv_wino36b_body<2, 2, 0, true, true>(p); }
__global__ void __launch_bounds__(256, 2) conv_wino36b_ns_ps_kernel(const Wino4Problem p) { conv_wino36b_body<2, 2, 0, false, true>(p); }
__global__ void __launch_bounds__(256, 2) conv_up_wino36_ps_kernel(const Wino4Problem p) { conv_wino36b_body<2, 2, 1, true, true>(p); }
__global__ void __launch_bounds__(256, 2) conv_up_wino36_ns_ps_kernel(const Wino4Problem p) { conv_wino36b_body<2, 2, 1, false, true>(p); }
__global__ void __launch_bounds__(256, 2) conv_wino36_rgb_ps_kernel(const Wino4Problem p) { conv_wino36b_body<2, 2, 2, true, true>(p); }
__global__ void __launch_bounds__(256, 2) conv_wino36_rgb_ns_ps_kernel(const Wino4Problem p) { conv_wino36b_body<2, 2, 2, false, true>(p); }
static bool w4_point_split() {
  static const bool on = [] { const char* e = getenv("RW_W4_PSPLIT"); return e && e[0] == '1'; }();
  return on;
}
#endif

// G g G^T of one 3x3 kernel g[3 ky + kx]: the 36 values u[6 a + b]
__device__ __forceinline__ void w4_weight_points(const float* g, float (&u)[36]) {
    // G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
    float gg[6][3];                                   // G g

#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float g0 = g[kx], g1 = g[3 + kx], g2 = g[6 + kx];
      gg[0][kx] = 0.25f * g0;
      gg[1][kx] = (-1.f / 6.f) * (g0 + g1 + g2);
      gg[2][kx] = (-1.f / 6.f) * (g0 - g1 + g2);
      gg[3][kx] = (1.f / 24.f) * g0 + (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
      gg[4][kx] = (1.f / 24.f) * g0 - (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
      gg[5][kx] = g2;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const float g0 = gg[a][0], g1 = gg[a][1], g2 = gg[a][2];
      u[6 * a + 0] = 0.25f * g0;
      u[6 * a + 1] = (-1.f / 6.f) * (g0 + g1 + g2);
      u[6 * a + 2] = (-1.f / 6.f) * (g0 - g1 + g2);
      u[6 * a + 3] = (1.f / 24.f) * g0 + (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
      u[6 * a + 4] = (1.f / 24.f) * g0 - (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
      u[6 * a + 5] = g2;
    }
}
// PASS 0: the 36 fp32 values of lane `dst` (stride 256 floats per point quad); PASS 1: nothing is stored, the
// return value is max |u|; PASS 2: the f16 pair words of u * su in the same places (H16 kernels)
template <int PASS>
__device__ __forceinline__ float w4_pack_store(const float* g, float* dst, float su) {
    float u[36];
    w4_weight_points(g, u);
    float m = 0.f;
    if (PASS == 1) {
#pragma unroll
      for (int i = 0; i < 36; ++i) m = fmaxf(m, fabsf(u[i]));
      return m;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      if (PASS == 0) {
        *reinterpret_cast<w4_f32x4*>(dst + q * 256) =
            w4_f32x4{u[w4_nat(4 * q)], u[w4_nat(4 * q + 1)], u[w4_nat(4 * q + 2)], u[w4_nat(4 * q + 3)]};
      } else {
        // H16: one word per (position, lane), positions 256 bytes apart -- the kernels read a word into both halves of
        // an operand pair with ds_read2st64_b32 (dst = the lane's column of the block: + lane instead of + 4 lane)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          reinterpret_cast<unsigned*>(dst)[(4 * q + e) * 64] = w4_pack16(u[w4_nat(4 * q + e)] * su);
      }
    }
    return m;
}
// H16 packing runs as two entry points: rw_*_absmax_f32 (PASS 1) leaves max |U| as a bound (one slot per workgroup +
// rw_bound_finish); the host reads it ONCE per weight version, derives the power of two -- max |U| < 2^eu,
// su = 2^(15 - eu): rw_split_weight_scale -- and hands it BY VALUE to rw_pack_* (PASS 2) and, inverted, to every
// convolution launch.  (Round 4 kept max |U| and 2^-eU in device scalars behind the packed weights: zeroed by a memset,
// raised with atomics, read by the next launches -- the first forward after a pack occasionally read a stale one.)
// One thread: the 36 values of one (o, i).  uf[o / 16][i / 4][xi / 4][16 (i % 4) + o % 16][xi % 4]
template <int PASS>
__global__ void __launch_bounds__(256) pack_wino36_kernel(const float* __restrict__ w, float* __restrict__ uf,
                                                          int out_ch, int in_ch, float su, float* __restrict__ bound) {
  const int64_t total = (int64_t)out_ch * in_ch;
  const int kqn = in_ch >> 2;
  float m = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    int64_t rest = idx >> 6;
    const int kq = (int)(rest % kqn);
    const int ob = (int)(rest / kqn);
    const int o = 16 * ob + (lane & 15), i = 4 * kq + (lane >> 4);
    m = fmaxf(m, w4_pack_store<PASS>(w + ((int64_t)o * in_ch + i) * 9,
                                     uf + ((int64_t)ob * kqn + kq) * (9 * 256) + (PASS == 2 ? lane : lane * 4), su));
  }
  __shared__ float red[4];
  if (PASS == 1) rw_bound_store_block_256(bound, m, red);
  if (PASS == 2 && blockIdx.x == 0 && threadIdx.x < 4)       // for inspection only: no kernel reads the trailer
    uf[36 * total + threadIdx.x] = threadIdx.x == 0 ? 1.f / su : (threadIdx.x == 1 ? su : 0.f);
}

// The same for the transposed convolution + blur problem: virtual channel v = 4 o + 2 py + px carries the phase
// kernel h[a][b] = g6[2 - 2a + py][2 - 2b + px], g6[ty][tx] = sum_{c,d} k'[c][d] w[ty - 1 + c][tx - 1 + d] (t = -2..3),
// k' = the blur kernel as upfirdn2d applies it (flipped).  w[o][i][ky][kx] as rw_conv_transpose3x3s2_f32 sees it.
template <int PASS>
__global__ void __launch_bounds__(256) pack_up_wino36_kernel(const float* __restrict__ w, const float* __restrict__ k4,
                                                             float* __restrict__ uf, int out_ch, int in_ch, float su,
                                                             float* __restrict__ bound) {
  const int vch = 4 * out_ch;
  const int64_t total = (int64_t)vch * in_ch;
  const int kqn = in_ch >> 2;
  float m = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    int64_t rest = idx >> 6;
    const int kq = (int)(rest % kqn);
    const int ob = (int)(rest / kqn);
    const int v = 16 * ob + (lane & 15), i = 4 * kq + (lane >> 4);
    const int o = v >> 2, py = (v >> 1) & 1, px = v & 1;
    const float* g = w + ((int64_t)o * in_ch + i) * 9;
    float h[9];
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
    m = fmaxf(m, w4_pack_store<PASS>(h, uf + ((int64_t)ob * kqn + kq) * (9 * 256) + (PASS == 2 ? lane : lane * 4), su));
  }
  __shared__ float red[4];
  if (PASS == 1) rw_bound_store_block_256(bound, m, red);
  if (PASS == 2 && blockIdx.x == 0 && threadIdx.x < 4)
    uf[36 * total + threadIdx.x] = threadIdx.x == 0 ? 1.f / su : (threadIdx.x == 1 ? su : 0.f);
}

static bool wino4_shape_ok(int out_ch, int in_ch, int h, int w) {
  return out_ch > 0 && in_ch > 0 && out_ch % 32 == 0 && in_ch % 8 == 0 && w % 64 == 0 && h % 8 == 0;
}

extern "C" int rw_conv3x3_wino4_supported(int out_ch, int in_ch, int h, int w) {
  return wino4_shape_ok(out_ch, in_ch, h, w) ? 1 : 0;
}

extern "C" long long rw_packed_conv_weight_wino4_elems(int out_ch, int in_ch) {
  if (out_ch <= 0 || in_ch <= 0 || out_ch % 16 || in_ch % 4) return -1;
  return 36LL * out_ch * in_ch;
}

extern "C" int rw_pack_conv_weight_wino4_f32(const float* w, float* uf, int out_ch, int in_ch, rw_stream_t stream) {
  RW_CHECK_ARG(w && uf && out_ch > 0 && in_ch > 0);
  if (out_ch % 16 || in_ch % 4) return RW_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)out_ch * in_ch;
  hipLaunchKernelGGL(pack_wino36_kernel<0>, dim3(rw_stream_grid(total, 256)), dim3(256), 0, rw_s(stream), w, uf, out_ch,
                     in_ch, 1.f, (float*)nullptr);
  return RW_LAUNCH_RESULT();
}

static int wino4_gpw(const Wino4Problem& p, int batch, int o_tiles) {
  const char* e = getenv("RW_WINO4_GPW");
  int gpw = e ? atoi(e) : 16;         // whole rows where the launch still has >= 1024 workgroups: +1.5 % over 4
  if (gpw < 1) gpw = 1;
  if (gpw > p.groups_x) gpw = p.groups_x;
  while (p.groups_x % gpw) --gpw;
  while (gpw > 1 && (int64_t)batch * p.groups_y * (p.groups_x / gpw) * o_tiles < 1024) {
    --gpw;
    while (p.groups_x % gpw) --gpw;
  }
  return gpw;
}

// after an H16 launch whose waves stored their maxima: the bound of the result
static int w4h_finish(float* y_amax, int64_t nslots, rw_stream_t stream) {
  const int rc = RW_LAUNCH_RESULT();
  if (rc || !y_amax) return rc;
  return rw_bound_finish(y_amax, nslots, rw_s(stream));
}

static int conv3x3_wino4_launch(const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch, int h,
                                int w, float w_scale, const rw_conv_epilogue* ep, bool h16, float u_inv,
                                const float* x_amax, float* y_amax, rw_stream_t stream) {
  RW_CHECK_ARG(x && uf && y && batch > 0 && in_ch > 0 && out_ch > 0 && h > 0 && w > 0);
  RW_CHECK_ARG(!ep || ((!ep->noise || ep->noise_w) && (!ep->act || ep->bias)));
  RW_CHECK_ARG(!h16 || (x_amax && u_inv > 0.f));
  if (!wino4_shape_ok(out_ch, in_ch, h, w)) return RW_ERR_UNSUPPORTED;
  Wino4Problem p = {};
  p.x = x; p.uf = uf; p.y = y;
  p.style = ep ? ep->style : nullptr; p.demod = ep ? ep->demod : nullptr; p.noise = ep ? ep->noise : nullptr;
  p.noise_w = ep ? ep->noise_w : nullptr; p.bias = ep ? ep->bias : nullptr; p.act = ep ? ep->act : 0;
  p.batch = batch; p.in_ch = in_ch; p.out_ch = out_ch; p.h = h; p.w = w; p.w_scale = w_scale;
  p.x_amax = x_amax; p.y_amax = y_amax; p.u_inv = u_inv;
  // 32 out-channels x 2 tile rows (8 x 64 pixels) per workgroup
  p.groups_x = w / 64;
  p.groups_y = h / 8;
  const int o_tiles = out_ch / 32;
  const int gpw = wino4_gpw(p, batch, o_tiles);
  p.gpw = gpw;
  const int64_t work = (int64_t)batch * p.groups_y * (p.groups_x / gpw) * o_tiles;
  if (work <= 0 || work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  if (h16) {
    if (in_ch > 512) return RW_ERR_UNSUPPORTED;
    const int64_t n_out = (int64_t)batch * out_ch * h * w;
    const char* wg8 = getenv("RW_W4H_WG8");
    if (wg8 && wg8[0] == '1' && h % 16 == 0) {
      p.groups_y = h / 16;
      int g8 = 4;
      if (g8 > p.groups_x) g8 = p.groups_x;
      while (p.groups_x % g8) --g8;
      p.gpw = g8;
      const int64_t work8 = (int64_t)batch * p.groups_y * (p.groups_x / g8) * o_tiles;
      if (y_amax && 8 * work8 > rw_bound_slot_capacity(n_out)) return RW_ERR_UNSUPPORTED;
      hipLaunchKernelGGL(conv_wino36h_wg8_kernel, dim3((unsigned)work8), dim3(512), 0, rw_s(stream), p);
      return w4h_finish(y_amax, 8 * work8, stream);
    }
    if (y_amax && 4 * work > rw_bound_slot_capacity(n_out)) return RW_ERR_UNSUPPORTED;
    if (w4h_point_split(in_ch)) hipLaunchKernelGGL(conv_wino36h_ps_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
    else hipLaunchKernelGGL(conv_wino36h_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
    return w4h_finish(y_amax, 4 * work, stream);
  }
  // versions: 1 = registers / compiler-scheduled loads (256 threads); 2 = <4,3> 512-thread workgroups;
  // 3 (default) = <2,2> two 256-thread workgroups per CU.  RW_WINO4_V overrides for comparison.
  const char* ver = getenv("RW_WINO4_V");
  const int version = ver ? atoi(ver) : 3;
  if (version == 2 && h % 16 == 0 && in_ch <= 512) {
    const char* e = getenv("RW_WINO4_GPW");
    p.groups_y = h / 16;
    int gpw2 = e ? atoi(e) : 4;
    if (gpw2 < 1) gpw2 = 1;
    if (gpw2 > p.groups_x) gpw2 = p.groups_x;
    while (p.groups_x % gpw2) --gpw2;
    while (gpw2 > 1 && (int64_t)batch * p.groups_y * (p.groups_x / gpw2) * o_tiles < 512) {
      --gpw2;
      while (p.groups_x % gpw2) --gpw2;
    }
    p.gpw = gpw2;
    const int64_t work2 = (int64_t)batch * p.groups_y * (p.groups_x / gpw2) * o_tiles;
    if (work2 <= 0 || work2 > 0x7fffffff) return RW_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv_wino36b_kernel<4, 3>), dim3((unsigned)work2), dim3(512), 0, rw_s(stream), p);
    return RW_LAUNCH_RESULT();
  }
  if (version == 3 && in_ch <= 512) {
#if W4_PSPLIT
    if (w4_point_split()) {
      if (p.style) hipLaunchKernelGGL(conv_wino36b_ps_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
      else hipLaunchKernelGGL(conv_wino36b_ns_ps_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
    } else
#endif
    if (p.style) hipLaunchKernelGGL((conv_wino36b_kernel<2, 2>), dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
    else hipLaunchKernelGGL(conv_wino36b_ns_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
    return RW_LAUNCH_RESULT();
  }
  hipLaunchKernelGGL((conv_wino36_kernel<2, 2>), dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  return RW_LAUNCH_RESULT();
}

extern "C" int rw_conv3x3_wino4_f32(const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch, int h,
                                    int w, float w_scale, const rw_conv_epilogue* ep, rw_stream_t stream) {
  return conv3x3_wino4_launch(x, uf, y, batch, in_ch, out_ch, h, w, w_scale, ep, false, 1.f, nullptr, nullptr, stream);
}

// ---- H16: the same operations with the products on the 16-bit matrix pipe (exact operand split, fp32 accumulation)
extern "C" long long rw_packed_conv_weight_wino4h_elems(int out_ch, int in_ch) {
  if (out_ch <= 0 || in_ch <= 0 || out_ch % 16 || in_ch % 4) return -1;
  return 36LL * out_ch * in_ch + 4;
}

extern "C" int rw_conv_weight_wino4h_absmax_f32(const float* w, int out_ch, int in_ch, float* bound, rw_stream_t stream) {
  RW_CHECK_ARG(w && bound && out_ch > 0 && in_ch > 0);
  if (out_ch % 16 || in_ch % 4) return RW_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)out_ch * in_ch;
  const int grid = rw_stream_grid(total, 256);
  hipLaunchKernelGGL(pack_wino36_kernel<1>, dim3(grid), dim3(256), 0, rw_s(stream), w, (float*)nullptr, out_ch, in_ch, 1.f,
                     bound);
  return w4h_finish(bound, grid, stream);
}

extern "C" int rw_pack_conv_weight_wino4h_f32(const float* w, float* uf, int out_ch, int in_ch, float u_scale,
                                              rw_stream_t stream) {
  RW_CHECK_ARG(w && uf && out_ch > 0 && in_ch > 0 && u_scale > 0.f);
  if (out_ch % 16 || in_ch % 4) return RW_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)out_ch * in_ch;
  hipLaunchKernelGGL(pack_wino36_kernel<2>, dim3(rw_stream_grid(total, 256)), dim3(256), 0, rw_s(stream), w, uf, out_ch,
                     in_ch, u_scale, (float*)nullptr);
  return RW_LAUNCH_RESULT();
}

extern "C" int rw_conv3x3_wino4h_f32(const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch, int h,
                                     int w, float w_scale, const rw_conv_epilogue* ep, float u_inv, const float* x_amax,
                                     float* y_amax, rw_stream_t stream) {
  return conv3x3_wino4_launch(x, uf, y, batch, in_ch, out_ch, h, w, w_scale, ep, true, u_inv, x_amax, y_amax, stream);
}

// ---------------------------------------------------------------------------------------
// Transposed convolution + blur + noise + bias + leaky ReLU of an upsampling StyledConv in one pass
// (utils/stylegan2/models.py:313-329 F.conv_transpose2d(stride=2), then Blur(pad 1,1) :289-291, NoiseInjection
// :259-270, FusedLeakyReLU :232-257): y (B, out_ch, 2H, 2W).  See UP in conv_wino36b_body.
// ---------------------------------------------------------------------------------------
static bool up_wino4_shape_ok(int out_ch, int in_ch, int h, int w) {
  return out_ch > 0 && out_ch % 8 == 0 && in_ch >= 8 && in_ch <= 512 && in_ch % 8 == 0 && w % 64 == 0 && h % 8 == 0;
}

extern "C" int rw_conv_transpose_blur_wino4_supported(int out_ch, int in_ch, int h, int w) {
  return up_wino4_shape_ok(out_ch, in_ch, h, w) ? 1 : 0;
}

extern "C" long long rw_packed_conv_transpose_blur_wino4_elems(int out_ch, int in_ch) {
  if (out_ch <= 0 || in_ch <= 0 || out_ch % 4 || in_ch % 4) return -1;
  return 144LL * out_ch * in_ch;
}

extern "C" int rw_pack_conv_transpose_blur_weight_wino4_f32(const float* w, const float* k4, float* uf, int out_ch,
                                                            int in_ch, rw_stream_t stream) {
  RW_CHECK_ARG(w && k4 && uf && out_ch > 0 && in_ch > 0);
  if (out_ch % 4 || in_ch % 4) return RW_ERR_UNSUPPORTED;
  const int64_t total = 4LL * out_ch * in_ch;
  hipLaunchKernelGGL(pack_up_wino36_kernel<0>, dim3(rw_stream_grid(total, 256)), dim3(256), 0, rw_s(stream), w, k4, uf,
                     out_ch, in_ch, 1.f, (float*)nullptr);
  return RW_LAUNCH_RESULT();
}

static int up_wino4_launch(const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch, int h, int w,
                           float w_scale, const rw_conv_epilogue* ep, const float* post_scale, bool h16, float u_inv,
                           const float* x_amax, float* y_amax, rw_stream_t stream) {
  RW_CHECK_ARG(x && uf && y && batch > 0 && in_ch > 0 && out_ch > 0 && h > 0 && w > 0);
  RW_CHECK_ARG(!ep || ((!ep->noise || ep->noise_w) && (!ep->act || ep->bias)));
  RW_CHECK_ARG(!h16 || (x_amax && u_inv > 0.f));
  if (!up_wino4_shape_ok(out_ch, in_ch, h, w)) return RW_ERR_UNSUPPORTED;
  Wino4Problem p = {};
  p.x = x; p.uf = uf; p.y = y;
  p.style = ep ? ep->style : nullptr; p.demod = ep ? ep->demod : nullptr; p.noise = ep ? ep->noise : nullptr;
  p.noise_w = ep ? ep->noise_w : nullptr; p.bias = ep ? ep->bias : nullptr; p.act = ep ? ep->act : 0;
  p.post = post_scale;
  p.batch = batch; p.in_ch = in_ch; p.out_ch = 4 * out_ch; p.h = h; p.w = w; p.w_scale = w_scale;
  p.x_amax = x_amax; p.y_amax = y_amax; p.u_inv = u_inv;
  p.groups_x = w / 64;
  p.groups_y = h / 8;
  const int o_tiles = p.out_ch / 32;
  p.gpw = wino4_gpw(p, batch, o_tiles);
  const int64_t work = (int64_t)batch * p.groups_y * (p.groups_x / p.gpw) * o_tiles;
  if (work <= 0 || work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  if (h16) {
    if (y_amax && 4 * work > rw_bound_slot_capacity((int64_t)batch * out_ch * 4 * h * w)) return RW_ERR_UNSUPPORTED;
    if (w4h_point_split(in_ch)) hipLaunchKernelGGL(conv_up_wino36h_ps_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
    else hipLaunchKernelGGL(conv_up_wino36h_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
    return w4h_finish(y_amax, 4 * work, stream);
  }
#if W4_PSPLIT
  if (w4_point_split()) {
    if (p.style) hipLaunchKernelGGL(conv_up_wino36_ps_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
    else hipLaunchKernelGGL(conv_up_wino36_ns_ps_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  } else
#endif
  if (p.style) hipLaunchKernelGGL(conv_up_wino36_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  else hipLaunchKernelGGL(conv_up_wino36_ns_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  return RW_LAUNCH_RESULT();
}

extern "C" int rw_conv_transpose3x3s2_blur_wino4_f32(const float* x, const float* uf, float* y, int batch, int in_ch,
                                                     int out_ch, int h, int w, float w_scale,
                                                     const rw_conv_epilogue* ep, const float* post_scale,
                                                     rw_stream_t stream) {
  return up_wino4_launch(x, uf, y, batch, in_ch, out_ch, h, w, w_scale, ep, post_scale, false, 1.f, nullptr, nullptr, stream);
}

extern "C" long long rw_packed_conv_transpose_blur_wino4h_elems(int out_ch, int in_ch) {
  if (out_ch <= 0 || in_ch <= 0 || out_ch % 4 || in_ch % 4) return -1;
  return 144LL * out_ch * in_ch + 4;
}

extern "C" int rw_conv_transpose_blur_weight_wino4h_absmax_f32(const float* w, const float* k4, int out_ch, int in_ch,
                                                               float* bound, rw_stream_t stream) {
  RW_CHECK_ARG(w && k4 && bound && out_ch > 0 && in_ch > 0);
  if (out_ch % 4 || in_ch % 4) return RW_ERR_UNSUPPORTED;
  const int64_t total = 4LL * out_ch * in_ch;
  const int grid = rw_stream_grid(total, 256);
  hipLaunchKernelGGL(pack_up_wino36_kernel<1>, dim3(grid), dim3(256), 0, rw_s(stream), w, k4, (float*)nullptr, out_ch, in_ch,
                     1.f, bound);
  return w4h_finish(bound, grid, stream);
}

extern "C" int rw_pack_conv_transpose_blur_weight_wino4h_f32(const float* w, const float* k4, float* uf, int out_ch,
                                                             int in_ch, float u_scale, rw_stream_t stream) {
  RW_CHECK_ARG(w && k4 && uf && out_ch > 0 && in_ch > 0 && u_scale > 0.f);
  if (out_ch % 4 || in_ch % 4) return RW_ERR_UNSUPPORTED;
  const int64_t total = 4LL * out_ch * in_ch;
  hipLaunchKernelGGL(pack_up_wino36_kernel<2>, dim3(rw_stream_grid(total, 256)), dim3(256), 0, rw_s(stream), w, k4, uf,
                     out_ch, in_ch, u_scale, (float*)nullptr);
  return RW_LAUNCH_RESULT();
}

extern "C" int rw_conv_transpose3x3s2_blur_wino4h_f32(const float* x, const float* uf, float* y, int batch, int in_ch,
                                                      int out_ch, int h, int w, float w_scale,
                                                      const rw_conv_epilogue* ep, const float* post_scale, float u_inv,
                                                      const float* x_amax, float* y_amax, rw_stream_t stream) {
  return up_wino4_launch(x, uf, y, batch, in_ch, out_ch, h, w, w_scale, ep, post_scale, true, u_inv, x_amax, y_amax, stream);
}

// ---------------------------------------------------------------------------------------
// The last styled convolution of the generator with ToRGB in the epilogue (models.py:639-655), F(4x4,3x3):
// rgb->out (B, 3, H, W) = ToRGB(act(conv(x) + noise + bias)) + rgb bias + skip; the feature map is not written.
// ---------------------------------------------------------------------------------------
extern "C" int rw_conv3x3_wino4_to_rgb_supported(int out_ch, int in_ch, int h, int w) {
  return out_ch == 32 && in_ch <= 512 && wino4_shape_ok(out_ch, in_ch, h, w) ? 1 : 0;
}

static int wino4_to_rgb_launch(const float* x, const float* uf, int batch, int in_ch, int out_ch, int h, int w,
                               float w_scale, const rw_conv_epilogue* ep, const rw_rgb_epilogue* rgb, bool h16,
                               float u_inv, const float* x_amax, rw_stream_t stream) {
  RW_CHECK_ARG(x && uf && rgb && rgb->weight && rgb->style && rgb->out && batch > 0 && in_ch > 0 && out_ch > 0);
  RW_CHECK_ARG(!ep || ((!ep->noise || ep->noise_w) && (!ep->act || ep->bias)));
  RW_CHECK_ARG(!h16 || (x_amax && u_inv > 0.f));
  if (!rw_conv3x3_wino4_to_rgb_supported(out_ch, in_ch, h, w)) return RW_ERR_UNSUPPORTED;
  Wino4Problem p = {};
  p.x = x; p.uf = uf; p.y = nullptr;
  p.style = ep ? ep->style : nullptr; p.demod = ep ? ep->demod : nullptr; p.noise = ep ? ep->noise : nullptr;
  p.noise_w = ep ? ep->noise_w : nullptr; p.bias = ep ? ep->bias : nullptr; p.act = ep ? ep->act : 0;
  p.rgb_weight = rgb->weight; p.rgb_style = rgb->style; p.rgb_bias = rgb->bias; p.rgb_skip = rgb->skip;
  p.rgb_out = rgb->out; p.rgb_scale = rgb->scale;
  p.batch = batch; p.in_ch = in_ch; p.out_ch = out_ch; p.h = h; p.w = w; p.w_scale = w_scale;
  p.x_amax = x_amax; p.u_inv = u_inv;
  p.groups_x = w / 64;
  p.groups_y = h / 8;
  p.gpw = wino4_gpw(p, batch, 1);
  const int64_t work = (int64_t)batch * p.groups_y * (p.groups_x / p.gpw);
  if (work <= 0 || work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  if (h16) {
    if (w4h_point_split(in_ch)) hipLaunchKernelGGL(conv_wino36h_rgb_ps_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
    else hipLaunchKernelGGL(conv_wino36h_rgb_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
    return RW_LAUNCH_RESULT();
  }
#if W4_PSPLIT
  if (w4_point_split()) {
    if (p.style) hipLaunchKernelGGL(conv_wino36_rgb_ps_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
    else hipLaunchKernelGGL(conv_wino36_rgb_ns_ps_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  } else
#endif
  if (p.style) hipLaunchKernelGGL(conv_wino36_rgb_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  else hipLaunchKernelGGL(conv_wino36_rgb_ns_kernel, dim3((unsigned)work), dim3(256), 0, rw_s(stream), p);
  return RW_LAUNCH_RESULT();
}

extern "C" int rw_conv3x3_wino4_to_rgb_f32(const float* x, const float* uf, int batch, int in_ch, int out_ch, int h,
                                           int w, float w_scale, const rw_conv_epilogue* ep,
                                           const rw_rgb_epilogue* rgb, rw_stream_t stream) {
  return wino4_to_rgb_launch(x, uf, batch, in_ch, out_ch, h, w, w_scale, ep, rgb, false, 1.f, nullptr, stream);
}

extern "C" int rw_conv3x3_wino4h_to_rgb_f32(const float* x, const float* uf, int batch, int in_ch, int out_ch, int h,
                                            int w, float w_scale, const rw_conv_epilogue* ep,
                                            const rw_rgb_epilogue* rgb, float u_inv, const float* x_amax,
                                            rw_stream_t stream) {
  return wino4_to_rgb_launch(x, uf, batch, in_ch, out_ch, h, w, w_scale, ep, rgb, true, u_inv, x_amax, stream);
}
