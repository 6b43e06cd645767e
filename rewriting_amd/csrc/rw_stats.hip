// Key statistics on gfx950: the uncentred second moment C += a^T a
// (RunningSecondMoment.add, utils/runningstats.py:1086-1097) as a symmetric split-K GEMM on the
// fp32 matrix cores, and per-channel sums for RunningVariance (utils/runningstats.py:763-788).
//
// The reference feeds addbmm_ with rows x 1 x C outer products of a permuted COPY of the key map
// (rewrite/ganrewrite.py:90-93).  Here the NCHW key map is consumed as it lies in HBM (layout 1):
// a K-chunk is 16 consecutive pixels of 128 channels, read as 64-byte row pieces, transposed on
// the way into LDS, so the key map is read once and never rewritten.
#include "rw_common.h"

#define ST_KC 16

static int stats_tile(int channels) { return channels >= 128 ? 128 : 64; }

static int stats_ksplit(int channels, int64_t rows) {
  const int ts = stats_tile(channels);
  const int nt = (int)rw_cdiv(channels, ts);
  const int tri = nt * (nt + 1) / 2;
  const int64_t chunks = rw_cdiv(rows, ST_KC);
  // tri * ks workgroups, two per CU and NOT ONE MORE: the kernel is bound by the matrix pipe, so a CU that receives
  // a third workgroup finishes 1.5x later than the rest and the launch with it (512 channels: 10 tile pairs x 52
  // slices = 520 workgroups ran as long as 768 would have; 51 slices: -30 %).  Few tile pairs (128 channels: one) get
  // as many row slices as it takes to fill the chip -- the reduction reads ks x C^2 floats, microseconds.
  int64_t ks = 512 / tri;
  const int64_t max_by_work = chunks / 8 > 0 ? chunks / 8 : 1;
  if (ks > max_by_work) ks = max_by_work;
  if (ks > 512) ks = 512;
  if (ks < 1) ks = 1;
  return (int)ks;
}

extern "C" int64_t rw_second_moment_workspace_bytes(int channels, int64_t rows) {
  return (int64_t)stats_ksplit(channels, rows) * channels * channels * (int64_t)sizeof(float);
}

// TS = 32*T*2 (4 waves as 2x2, each T x T MFMA tiles)
template <int T, int LAYOUT>
__global__ void __launch_bounds__(256) second_moment_kernel(const float* __restrict__ a,
                                                            float* __restrict__ part, int64_t rows,
                                                            int channels, int64_t hw, int ntiles,
                                                            int ksplit) {
  constexpr int TS = 64 * T;
  constexpr int TSP = TS + 4;
  constexpr int VEC = (ST_KC * TS / 4) / 256;   // float4 per thread per operand: 2 (TS=128) / 1 (TS=64)
  __shared__ __attribute__((aligned(16))) float S1[2][ST_KC][TSP];
  __shared__ __attribute__((aligned(16))) float S2[2][ST_KC][TSP];

  // upper-triangular tile pair from blockIdx.x
  int t1 = 0, rem = blockIdx.x;
  while (rem >= ntiles - t1) { rem -= ntiles - t1; ++t1; }
  const int t2 = t1 + rem;
  const int c10 = t1 * TS, c20 = t2 * TS;
  const int ks = blockIdx.y;
  const int64_t chunks = (rows + ST_KC - 1) / ST_KC;
  const int64_t cbeg = chunks * ks / ksplit, cend = chunks * (ks + 1) / ksplit;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 32 * T, wn0 = (wave & 1) * 32 * T;
  const int frow = lane >> 5, fcol = lane & 31;

  float4 r1[VEC], r2[VEC];
  auto fetch = [&](int64_t chunk) {
    const int64_t row0 = chunk * ST_KC;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int q = tid + j * 256;
      float4 v1 = make_float4(0.f, 0.f, 0.f, 0.f), v2 = v1;
      if (LAYOUT == 0) {       // (rows, C): float4 along channels
        const int kk = q / (TS / 4), c4 = (q % (TS / 4)) * 4;
        const int64_t row = row0 + kk;
        if (row < rows) {
          if (c10 + c4 < channels) v1 = *reinterpret_cast<const float4*>(a + row * channels + c10 + c4);
          if (c20 + c4 < channels) v2 = *reinterpret_cast<const float4*>(a + row * channels + c20 + c4);
        }
      } else {                 // NCHW: float4 along pixels, 16 pixels of one image per chunk
        const int c = q >> 2, part4 = (q & 3) * 4;
        const int64_t img = row0 / hw, p0 = row0 - img * hw;
        if (row0 < rows) {
          if (c10 + c < channels)
            v1 = *reinterpret_cast<const float4*>(a + (img * channels + c10 + c) * hw + p0 + part4);
          if (c20 + c < channels)
            v2 = *reinterpret_cast<const float4*>(a + (img * channels + c20 + c) * hw + p0 + part4);
        }
      }
      r1[j] = v1; r2[j] = v2;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int q = tid + j * 256;
      if (LAYOUT == 0) {
        const int kk = q / (TS / 4), c4 = (q % (TS / 4)) * 4;
        *reinterpret_cast<float4*>(&S1[buf][kk][c4]) = r1[j];
        *reinterpret_cast<float4*>(&S2[buf][kk][c4]) = r2[j];
      } else {
        const int c = q >> 2, part4 = (q & 3) * 4;
        S1[buf][part4 + 0][c] = r1[j].x; S1[buf][part4 + 1][c] = r1[j].y;
        S1[buf][part4 + 2][c] = r1[j].z; S1[buf][part4 + 3][c] = r1[j].w;
        S2[buf][part4 + 0][c] = r2[j].x; S2[buf][part4 + 1][c] = r2[j].y;
        S2[buf][part4 + 2][c] = r2[j].z; S2[buf][part4 + 3][c] = r2[j].w;
      }
    }
  };

  rw_f32x16 acc[T][T];
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (cbeg < cend) {
    fetch(cbeg);
    stash(0);
  }
  __syncthreads();
  for (int64_t ch = cbeg; ch < cend; ++ch) {
    const int buf = (int)((ch - cbeg) & 1);
    if (ch + 1 < cend) fetch(ch + 1);
#pragma unroll
    for (int kp = 0; kp < ST_KC / 2; ++kp) {
      float af[T], bf[T];
#pragma unroll
      for (int i = 0; i < T; ++i) af[i] = S1[buf][2 * kp + frow][wm0 + 32 * i + fcol];
#pragma unroll
      for (int j = 0; j < T; ++j) bf[j] = S2[buf][2 * kp + frow][wn0 + 32 * j + fcol];
#pragma unroll
      for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (ch + 1 < cend) stash(buf ^ 1);
    __syncthreads();
  }

  float* slab = part + (int64_t)ks * channels * channels;
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c1 = c10 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * frow;
        const int c2 = c20 + wn0 + 32 * j + fcol;
        if (c1 < channels && c2 < channels) slab[(int64_t)c1 * channels + c2] = acc[i][j][r];
      }
}

// mom2[c1][c2] += sum_s part[s][min-tile-first]; the lower triangle mirrors the upper tiles.
__global__ void __launch_bounds__(256) second_moment_reduce_kernel(const float* __restrict__ part,
                                                                   float* __restrict__ mom2,
                                                                   int channels, int ts, int ksplit) {
  const int64_t total = (int64_t)channels * channels;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c1 = (int)(idx / channels), c2 = (int)(idx % channels);
    const int64_t src = (c1 / ts <= c2 / ts) ? idx : (int64_t)c2 * channels + c1;
    float acc = 0.f;
    for (int s = 0; s < ksplit; ++s) acc += part[(int64_t)s * total + src];
    mom2[idx] += acc;
  }
}

extern "C" int rw_second_moment_f32(const float* a, float* mom2, int64_t rows, int channels,
                                    int64_t hw, int layout, void* workspace, rw_stream_t stream) {
  RW_CHECK_ARG(a && mom2 && workspace && rows > 0 && channels > 0);
  RW_CHECK_ARG(layout == 0 || layout == 1);
  if (layout == 0 && (channels % 4)) return RW_ERR_UNSUPPORTED;
  if (layout == 1 && (hw <= 0 || hw % ST_KC || rows % hw)) return RW_ERR_UNSUPPORTED;
  const int ts = stats_tile(channels);
  const int nt = (int)rw_cdiv(channels, ts);
  const int tri = nt * (nt + 1) / 2;
  const int ksplit = stats_ksplit(channels, rows);
  float* part = (float*)workspace;
  const dim3 grid(tri, ksplit);
  hipStream_t s = rw_s(stream);
  if (ts == 128) {
    if (layout == 0)
      hipLaunchKernelGGL((second_moment_kernel<2, 0>), grid, dim3(256), 0, s, a, part, rows, channels, hw, nt, ksplit);
    else
      hipLaunchKernelGGL((second_moment_kernel<2, 1>), grid, dim3(256), 0, s, a, part, rows, channels, hw, nt, ksplit);
  } else {
    if (layout == 0)
      hipLaunchKernelGGL((second_moment_kernel<1, 0>), grid, dim3(256), 0, s, a, part, rows, channels, hw, nt, ksplit);
    else
      hipLaunchKernelGGL((second_moment_kernel<1, 1>), grid, dim3(256), 0, s, a, part, rows, channels, hw, nt, ksplit);
  }
  int rc = RW_LAUNCH_RESULT();
  if (rc) return rc;
  const int64_t total = (int64_t)channels * channels;
  hipLaunchKernelGGL(second_moment_reduce_kernel, dim3(rw_stream_grid(total, 256)), dim3(256), 0, s,
                     part, mom2, channels, ts, ksplit);
  return RW_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------
// per-channel sum / sum of squares
// ---------------------------------------------------------------------------------------
// NCHW: one workgroup per channel, deterministic.
__global__ void __launch_bounds__(256) channel_sums_nchw_kernel(const float* __restrict__ a,
                                                                float* __restrict__ sums,
                                                                int64_t batch, int channels,
                                                                int64_t hw, int square_input) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  float s1 = 0.f, s2 = 0.f;
  for (int64_t b = 0; b < batch; ++b) {
    const float* row = a + (b * channels + c) * hw;
    for (int64_t i = threadIdx.x; i < hw; i += 256) {
      float v = row[i];
      if (square_input) v = v * v;
      s1 += v; s2 += v * v;
    }
  }
  s1 = rw_block_sum_256(s1, red);
  s2 = rw_block_sum_256(s2, red);
  if (threadIdx.x == 0) { sums[c] = s1; sums[channels + c] = s2; }
}

// (rows, C): 64 channels x 4 row lanes per workgroup, row-split over blockIdx.y, fp32 atomics.
__global__ void __launch_bounds__(256) channel_sums_rows_kernel(const float* __restrict__ a,
                                                                float* __restrict__ sums,
                                                                int64_t rows, int channels,
                                                                int square_input) {
  __shared__ float l1[4][64], l2[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s1 = 0.f, s2 = 0.f;
  if (c < channels) {
    for (int64_t r = (int64_t)blockIdx.y * 4 + rl; r < rows; r += (int64_t)gridDim.y * 4) {
      float v = a[r * channels + c];
      if (square_input) v = v * v;
      s1 += v; s2 += v * v;
    }
  }
  l1[rl][cl] = s1; l2[rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < channels) {
    atomicAdd(&sums[c], l1[0][cl] + l1[1][cl] + l1[2][cl] + l1[3][cl]);
    atomicAdd(&sums[channels + c], l2[0][cl] + l2[1][cl] + l2[2][cl] + l2[3][cl]);
  }
}

extern "C" int rw_channel_sums_f32(const float* a, float* sums, int64_t rows, int channels,
                                   int64_t hw, int layout, int square_input, rw_stream_t stream) {
  RW_CHECK_ARG(a && sums && rows > 0 && channels > 0 && (layout == 0 || layout == 1));
  hipStream_t s = rw_s(stream);
  if (layout == 1) {
    RW_CHECK_ARG(hw > 0 && rows % hw == 0);
    hipLaunchKernelGGL(channel_sums_nchw_kernel, dim3(channels), dim3(256), 0, s, a, sums, rows / hw,
                       channels, hw, square_input);
    return RW_LAUNCH_RESULT();
  }
  hipError_t e = hipMemsetAsync(sums, 0, 2 * (size_t)channels * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  int64_t gy = rw_cdiv(rows, 4 * 64);
  if (gy > 512) gy = 512;
  hipLaunchKernelGGL(channel_sums_rows_kernel, dim3((unsigned)rw_cdiv(channels, 64), (unsigned)gy),
                     dim3(256), 0, s, a, sums, rows, channels, square_input);
  return RW_LAUNCH_RESULT();
}
