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
//   M[xi][o][t] = sum_i U[xi][o][i] V[xi][i][t]          16 independent GEMMs -> v_mfma_f32_32x32x2_f32
//   Y[o][t]     = A^T M A            2x2 outputs, then the fused epilogue (demod, noise, bias, leaky-ReLU, ToRGB)
//
// Geometry.  An MFMA column tile is 32 Winograd tiles = 2 tile rows x 16 tile columns = 4 x 32 output pixels
// ("tile group").  A wave owns 32 out-channels x one tile group x EIGHT of the sixteen xi (128 accumulator
// registers: two waves per SIMD, two workgroups per CU, so one workgroup's prologue / transform / epilogue
// overlaps the other's MFMAs); its partner wave owns the other eight xi of the same outputs and hands its partial
// 2x2 outputs over through LDS at the end.  A workgroup is 4 waves: 2 xi-halves x (WGM out-channel blocks x
// WGN tile groups), WGM * WGN = 2.
//
// Per chunk of IC = 8 input channels the (4 WGN + 2) x 34 input patch is fetched once (coalesced NCHW row
// pieces, zero padding and the style multiply applied on the way into LDS), transformed cooperatively into
// V[2][16][IC][32 WGN] (double buffered) and consumed by 4 k-pairs x 8 MFMAs per wave.  Weight fragments come
// straight from L2 in MFMA A-fragment order (two 16-byte loads per lane and k-pair).  Fetch, staging and
// transform of chunk c+1 / c+2 ride between the MFMAs of chunk c.
#include "rw_common.h"

#define WN_IC 8                 // input channels per chunk
#define WN_KP (WN_IC / 2)       // k-pairs per chunk
#define WN_RS 48                // LDS row pitch of the raw patch (floats): 2 rows = 96 dwords = 32 mod 64 banks
#define WN_PC 34                // patch columns

struct WinoProblem {
  const float* x; const float* uf; float* y;
  const float* style; const float* demod; const float* noise; const float* noise_w; const float* bias;
  int batch, in_ch, out_ch, h, w;
  int groups_x, groups_y;
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

template <int WGM, int WGN, bool RGB>
__global__ void __launch_bounds__(256, 2) conv_wino_kernel(const WinoProblem p) {
  static_assert(WGM * WGN == 2, "two (out-channel block, tile group) pairs per workgroup");
  constexpr int IC = WN_IC, KP = WN_KP;
  constexpr int NT = 32 * WGN;                     // tiles per workgroup
  constexpr int PR = 4 * WGN + 2;                  // patch rows
  constexpr int NPOS = PR * WN_PC;
  constexpr int PSLOT = (NPOS + 255) / 256;
  constexpr int NRAW = PSLOT * IC;                 // raw elements per thread and chunk
  constexpr int NIT = NT * IC / 256;               // transform items (tile, channel) per thread and chunk
  static_assert(NIT >= 1 && NT * IC % 256 == 0, "transform items");
  constexpr int CH_STEP = 256 / NT;                // channel stride between a thread's items
  __shared__ __attribute__((aligned(16))) float Rs[IC][PR][WN_RS];
  __shared__ __attribute__((aligned(16))) float Vs[2][16][IC][NT];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xh = wave & 1;                         // which eight xi: rows a in {2 xh, 2 xh + 1} of the 4x4 transform
  const int pw = wave >> 1;                        // (out-channel block, tile group) pair
  const int wm = pw / WGN, wn = pw % WGN;
  const int frow = lane >> 5, fcol = lane & 31;

  int local = wn_xcd_remap(blockIdx.x, gridDim.x);
  const int o_tiles = p.out_ch / (32 * WGM);
  const int o0 = (local % o_tiles) * 32 * WGM; local /= o_tiles;
  const int gx = local % p.groups_x; local /= p.groups_x;
  const int gy = local % p.groups_y;
  const int ib = local / p.groups_y;
  const int y0 = gy * 4 * WGN, x0 = gx * 32;
  const int64_t hw = (int64_t)p.h * p.w;
  const float* xb = p.x + (int64_t)ib * p.in_ch * hw;
  const float* st = p.style ? p.style + (int64_t)ib * p.in_ch : nullptr;      // uniform: scalar loads
  const int n_chunks = p.in_ch / IC;

  // ---- raw patch: this thread owns up to PSLOT fixed positions (r, c) and walks the IC channels of a chunk.
  // Outside the image: a legal address, value multiplied by 0.  Slots past the patch: a padding column.
  int xoff[PSLOT], xlds[PSLOT];
  float xmask[PSLOT];
#pragma unroll
  for (int sl = 0; sl < PSLOT; ++sl) {
    const int pos = tid + 256 * sl;
    const int r = pos / WN_PC, c = pos - r * WN_PC;
    const int iy = y0 - 1 + r, ix = x0 - 1 + c;
    const bool ok = pos < NPOS && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
    xoff[sl] = ok ? iy * p.w + ix : 0;
    xmask[sl] = ok ? 1.0f : 0.0f;
    xlds[sl] = pos < NPOS ? r * WN_RS + c : WN_RS - 1;
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
  auto raw_step = [&](int j) {                     // j < NRAW
    const int sl = j / IC, ic = j % IC;
    (&Rs[0][0][0])[ic * PR * WN_RS + xlds[sl]] = xreg[sl][ic] * (xmask[sl] * sty[ic]);
  };

  // ---- transform items: tile tl of the workgroup, channel tch + CH_STEP * item
  const int tl = tid % NT, tch = tid / NT;
  const int tg = tl >> 5, ttr = (tl >> 4) & 1, ttc = tl & 15;
  const float* rsrc = &Rs[tch][4 * tg + 2 * ttr][2 * ttc];
  float* vdst = &Vs[0][0][tch][tl];
  float2 drow[NIT][4][2];                          // the 4x4 input tiles of this thread's items
  float e[4][4];
  // micro-steps of a chunk's transform: NIT reads (one per item), then per item four column transforms (row a)
  // and four row transforms + stores (column b)
  constexpr int NXF = 9 * NIT;
  auto xform_step = [&](int buf, int s) {
    if (s < NIT) {
      const float* src = rsrc + s * CH_STEP * PR * WN_RS;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        drow[s][a][0] = *reinterpret_cast<const float2*>(src + a * WN_RS);
        drow[s][a][1] = *reinterpret_cast<const float2*>(src + a * WN_RS + 2);
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
      float* dst = vdst + buf * 16 * IC * NT + item * CH_STEP * NT;
      dst[(0 * 4 + b) * IC * NT] = e[0][b] - e[2][b];
      dst[(1 * 4 + b) * IC * NT] = e[1][b] + e[2][b];
      dst[(2 * 4 + b) * IC * NT] = e[2][b] - e[1][b];
      dst[(3 * 4 + b) * IC * NT] = e[1][b] - e[3][b];
    }
  };
  // slot s2 of the second half of a chunk carries: s2 == 0 the reads, s2 >= 1 compute steps 2 (s2-1), 2 (s2-1) + 1
  auto xform_slot = [&](int buf, int s2) {
    if (s2 == 0) {
#pragma unroll
      for (int i = 0; i < NIT; ++i) xform_step(buf, i);
    } else {
      const int k = NIT + 2 * (s2 - 1);
      if (k < NXF) xform_step(buf, k);
      if (k + 1 < NXF) xform_step(buf, k + 1);
    }
  };

  // ---- operands.  A: U in fragment order uf[o/32][k-pair][xi half][2][lane][4]; B: V[buf][xi][2 kp + frow][tile]
  const int kpg_total = p.in_ch >> 1;
  const float* ufw = p.uf + (((int64_t)((o0 >> 5) + wm) * kpg_total) * 2 + xh) * 512 + lane * 4;    // + kpg * 1024
  rw_f32x4 areg[2][2];
  float breg[2][8];
  auto aload = [&](int slot, int kpg) {
    const float* base = ufw + (int64_t)kpg * 1024;
    areg[slot][0] = *reinterpret_cast<const rw_f32x4*>(base);
    areg[slot][1] = *reinterpret_cast<const rw_f32x4*>(base + 256);
  };
  const float* vsrc = &Vs[0][8 * xh][frow][wn * 32 + fcol];
  auto bload1 = [&](int slot, int buf, int kp, int q) {
    breg[slot][q] = vsrc[buf * 16 * IC * NT + q * IC * NT + 2 * kp * NT];
  };

  rw_f32x16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  // ---- prologue: chunk 0 transformed into V[0], chunk 1 fetched
  xfetch(0);
  aload(0, 0);
#pragma unroll
  for (int j = 0; j < NRAW; ++j) raw_step(j);
  __syncthreads();
  if (n_chunks > 1) xfetch(IC);
#pragma unroll
  for (int s = 0; s < NXF; ++s) xform_step(0, s);
  __syncthreads();

  constexpr int SLOTS = 8 * KP;                    // MFMA slots per chunk and wave
  constexpr int HALF = SLOTS / 2;
  static_assert(NRAW <= HALF && 1 + 4 * NIT <= HALF, "staging fits the slots of a chunk");
  // One chunk.  MORE: another chunk follows -- its raw patch goes into Rs during the first half of the slots,
  // barrier, the fetch of the chunk after it is issued, its transform into V[buf^1] rides on the second half.
  auto chunk = [&](int c, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value != 0;
    const int buf = c & 1;
    const int cn2 = c + 2 < n_chunks ? c + 2 : n_chunks - 1;
#pragma unroll
    for (int q = 0; q < 8; ++q) bload1(0, buf, 0, q);
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      const int cur = kp & 1, nxt = cur ^ 1;
      // next k-pair's weights (for the last k-pair: the first of the next chunk; global memory, no barrier involved)
      {
        int kpg = c * KP + kp + 1;
        if (kpg >= kpg_total) kpg = kpg_total - 1;
        aload(nxt, kpg);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int slot = kp * 8 + q;
        if (kp + 1 < KP) bload1(nxt, buf, kp + 1, q);          // B one k-pair ahead
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[cur][q >> 2][q & 3], breg[cur][q], acc[q], 0, 0, 0);
        if (MORE) {
          if (slot < HALF) {
            if (slot < NRAW) raw_step(slot);
          } else {
            xform_slot(buf ^ 1, slot - HALF);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MORE && slot == HALF - 1) {
          __syncthreads();                         // raw patch of chunk c+1 complete in Rs
          xfetch(cn2 * IC);                        // registers free again: chunk c+2 (a redundant refill at the end)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (MORE) __syncthreads();                     // V[buf^1] complete; Rs free
  };
  for (int c = 0; c + 1 < n_chunks; ++c) chunk(c, wn_int<1>());
  chunk(n_chunks - 1, wn_int<0>());

  // ---- output transform.  Row r of the accumulator tile is out-channel o_first + (r&3) + 8 (r>>2); this wave
  // holds M[a][b] for a in {2 xh, 2 xh + 1}: acc[4 (a - 2 xh) + b].  Y = A^T M A is linear in M, so each wave of
  // a pair forms its partial 2x2 outputs; the xh = 1 wave hands them over through LDS.
  float part[16][4];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float s0[4], s1[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float m0 = acc[b][r], m1 = acc[4 + b][r];
      if (xh == 0) { s0[b] = m0 + m1; s1[b] = m1; }            // rows a = 0, 1
      else { s0[b] = m0; s1[b] = -m0 - m1; }                   // rows a = 2, 3
    }
    part[r][0] = s0[0] + s0[1] + s0[2];
    part[r][1] = s0[1] - s0[2] - s0[3];
    part[r][2] = s1[0] + s1[1] + s1[2];
    part[r][3] = s1[1] - s1[2] - s1[3];
  }
  __syncthreads();                                 // every wave is done reading V
  float* xch = &Vs[0][0][0][0] + pw * 64 * 64 + lane;          // [pair][r * 4 + j][lane]
  if (xh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) xch[(r * 4 + j) * 64] = part[r][j];
  }
  __syncthreads();
  if (xh == 1) return;
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) part[r][j] += xch[(r * 4 + j) * 64];

  // ---- epilogue: this lane holds the 2x2 outputs of tile (wn, fcol) for 16 out-channels
  const int otr = (fcol >> 4) & 1, otc = fcol & 15;
  const int oy = y0 + 4 * wn + 2 * otr, ox = x0 + 2 * otc;
  const int o_first = o0 + 32 * wm + 4 * frow;
  float nz[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.noise) {
    const float nw = p.noise_w[0];
    const float* np = p.noise + (int64_t)ib * hw + (int64_t)oy * p.w + ox;
    const float2 n0 = *reinterpret_cast<const float2*>(np), n1 = *reinterpret_cast<const float2*>(np + p.w);
    nz[0] = n0.x * nw; nz[1] = n0.y * nw; nz[2] = n1.x * nw; nz[3] = n1.y * nw;
  }
  float scale[16], bias[16];
  if (p.demod) {
    const float* dm = p.demod + (int64_t)ib * p.out_ch + o_first;
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
    for (int r = 0; r < 16; ++r) bias[r] = p.bias[o_first + (r & 3) + 8 * (r >> 2)];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = 0.f;
  }
  float wr[3][16];
  float rgbp[4][3];
  if (RGB) {
    float sr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sr[r] = p.rgb_style[(int64_t)ib * p.out_ch + o_first + (r & 3) + 8 * (r >> 2)];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
#pragma unroll
      for (int r = 0; r < 16; ++r) wr[cc][r] = p.rgb_weight[cc * p.out_ch + o_first + (r & 3) + 8 * (r >> 2)];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
#pragma unroll
      for (int r = 0; r < 16; ++r) wr[cc][r] = p.rgb_scale * wr[cc][r] * sr[r];
#pragma unroll
    for (int j = 0; j < 4; ++j) rgbp[j][0] = rgbp[j][1] = rgbp[j][2] = 0.f;
  }
  // Neighbouring lanes (tile columns 2m, 2m+1) exchange halves so that the even lane stores four consecutive
  // pixels of output row oy and the odd lane four of row oy + 1: one aligned 16-byte store per lane and channel.
  const bool odd = fcol & 1;
  float* yb = p.y ? p.y + ((int64_t)ib * p.out_ch + o_first) * hw + (int64_t)(oy + (odd ? 1 : 0)) * p.w + (ox & ~3)
                  : nullptr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = part[r][j] * scale[r] + nz[j];
      if (p.act) {
        t += bias[r];
        t = ((t > 0.f) ? t : t * 0.2f) * 1.4142135623730951f;
      }
      v[j] = t;
      if (RGB) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) rgbp[j][cc] += t * wr[cc][r];
      }
    }
    if (yb) {
      // even lane keeps its row 0 (v0, v1) and takes the partner's row 0; odd lane keeps row 1, takes the partner's
      const float g0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(odd ? v[0] : v[2]), 0xB1, 0xf, 0xf, true));
      const float g1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(odd ? v[1] : v[3]), 0xB1, 0xf, 0xf, true));
      rw_f32x4 o4;
      o4[0] = odd ? g0 : v[0]; o4[1] = odd ? g1 : v[1];
      o4[2] = odd ? v[2] : g0; o4[3] = odd ? v[3] : g1;
      *reinterpret_cast<rw_f32x4*>(yb + (int64_t)((r & 3) + 8 * (r >> 2)) * hw) = o4;
    }
  }
  if (RGB) {
    // the partner half of the wave (frow) holds the other 16 of the 32 out-channels of the same pixels
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) rgbp[j][cc] += __shfl_xor(rgbp[j][cc], 32, 64);
    if (frow == 0) {
      float rb[3] = {0.f, 0.f, 0.f};
      if (p.rgb_bias) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) rb[cc] = p.rgb_bias[cc];
      }
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const int64_t base = ((int64_t)ib * 3 + cc) * hw + (int64_t)oy * p.w + ox;
        float2 k0 = {0.f, 0.f}, k1 = {0.f, 0.f};
        if (p.rgb_skip) {
          k0 = *reinterpret_cast<const float2*>(p.rgb_skip + base);
          k1 = *reinterpret_cast<const float2*>(p.rgb_skip + base + p.w);
        }
        float2 r0 = {rgbp[0][cc] + rb[cc] + k0.x, rgbp[1][cc] + rb[cc] + k0.y};
        float2 r1 = {rgbp[2][cc] + rb[cc] + k1.x, rgbp[3][cc] + rb[cc] + k1.y};
        *reinterpret_cast<float2*>(p.rgb_out + base) = r0;
        *reinterpret_cast<float2*>(p.rgb_out + base + p.w) = r1;
      }
    }
  }
}

// U = G g G^T in fragment order: uf[o/32][kpg][xi half][2][lane][4], value U[xi = 8 half + 4 q + e][o = 32 (o/32) +
// (lane & 31)][i = 2 kpg + (lane >> 5)].  The weight scale 1/sqrt(9 Cin) is NOT folded in.
__global__ void __launch_bounds__(256) pack_wino_kernel(const float* __restrict__ w, float* __restrict__ uf,
                                                        int out_ch, int in_ch) {
  const int64_t total = (int64_t)out_ch * in_ch;       // one thread: the 16 values of one (o, i)
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    const int64_t rest = idx >> 6;
    const int kpg = (int)(rest % (in_ch >> 1));
    const int ob = (int)(rest / (in_ch >> 1));
    const int o = 32 * ob + (lane & 31), i = 2 * kpg + (lane >> 5);
    const float* g = w + ((int64_t)o * in_ch + i) * 9;
    float gg[4][3];                                      // G g
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float g0 = g[kx], g1 = g[3 + kx], g2 = g[6 + kx];
      gg[0][kx] = g0;
      gg[1][kx] = 0.5f * (g0 + g1 + g2);
      gg[2][kx] = 0.5f * (g0 - g1 + g2);
      gg[3][kx] = g2;
    }
    float u[16];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      u[4 * a + 0] = gg[a][0];
      u[4 * a + 1] = 0.5f * (gg[a][0] + gg[a][1] + gg[a][2]);
      u[4 * a + 2] = 0.5f * (gg[a][0] - gg[a][1] + gg[a][2]);
      u[4 * a + 3] = gg[a][2];
    }
    float* dst = uf + ((int64_t)ob * (in_ch >> 1) + kpg) * 1024 + lane * 4;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        *reinterpret_cast<rw_f32x4*>(dst + (half * 2 + q) * 256) =
            rw_f32x4{u[8 * half + 4 * q], u[8 * half + 4 * q + 1], u[8 * half + 4 * q + 2], u[8 * half + 4 * q + 3]};
  }
}

static bool wino_shape_ok(int out_ch, int in_ch, int h, int w) {
  return out_ch > 0 && in_ch > 0 && out_ch % 32 == 0 && in_ch % WN_IC == 0 && w % 32 == 0 && w >= 32 && h >= 8 &&
         h % 8 == 0;
}

extern "C" int rw_conv3x3_wino_supported(int out_ch, int in_ch, int h, int w) {
  return wino_shape_ok(out_ch, in_ch, h, w) ? 1 : 0;
}

extern "C" long long rw_packed_conv_weight_wino_elems(int out_ch, int in_ch) {
  if (out_ch <= 0 || in_ch <= 0 || out_ch % 32 || in_ch % 2) return -1;
  return 16LL * out_ch * in_ch;
}

extern "C" int rw_pack_conv_weight_wino_f32(const float* w, float* uf, int out_ch, int in_ch, rw_stream_t stream) {
  RW_CHECK_ARG(w && uf && out_ch > 0 && in_ch > 0);
  if (out_ch % 32 || in_ch % 2) return RW_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)out_ch * in_ch;
  hipLaunchKernelGGL(pack_wino_kernel, dim3(rw_stream_grid(total, 256)), dim3(256), 0, rw_s(stream), w, uf, out_ch,
                     in_ch);
  return RW_LAUNCH_RESULT();
}

static int launch_wino(WinoProblem& p, bool rgb, hipStream_t s) {
  const bool two_blocks = p.out_ch % 64 == 0;            // <2,1>: 64 out-channels x one tile group
  const int wgn = two_blocks ? 1 : 2;
  p.groups_x = p.w / 32;
  p.groups_y = p.h / (4 * wgn);
  const int64_t work = (int64_t)p.batch * p.groups_x * p.groups_y * (p.out_ch / (two_blocks ? 64 : 32));
  if (work <= 0 || work > 0x7fffffff) return RW_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)work), block(256);
  if (two_blocks) {
    if (rgb) return RW_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv_wino_kernel<2, 1, false>), grid, block, 0, s, p);
  } else if (rgb) {
    hipLaunchKernelGGL((conv_wino_kernel<1, 2, true>), grid, block, 0, s, p);
  } else {
    hipLaunchKernelGGL((conv_wino_kernel<1, 2, false>), grid, block, 0, s, p);
  }
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
