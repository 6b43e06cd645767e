/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's two CUDA kernels, used by
 * tests/ as a checker (never linked into or called by the product).
 *
 *   oracle_fused_bias_act  follows utils/stylegan2/op/fused_bias_act_kernel.cu:18-49
 *   oracle_upfirdn2d       follows utils/stylegan2/op/upfirdn2d_kernel.cu:52-137 (index algebra
 *                          of the tile loop, without the tiling) and :167-168 (output size)
 *
 * The reference's .cu files cannot be compiled here (they need nvcc and the ATen CUDA headers),
 * so this restatement is pinned against tests/golden/ops.npz, which was produced by the
 * reference's own pure-torch spec upfirdn2d_native (op/upfirdn2d.py:152-186).
 * Build: gcc -O2 -shared -fPIC oracle/native_ops.c -o oracle/_build/liboracle_native.so
 */
#include <stdint.h>

void oracle_fused_bias_act(const float* x, const float* b, const float* ref, float* y, int64_t n,
                           int64_t step_b, int64_t size_b, int act, int grad, float alpha,
                           float scale) {
  for (int64_t xi = 0; xi < n; ++xi) {
    float v = x[xi];
    if (b) v += b[(xi / step_b) % size_b];
    float r = ref ? ref[xi] : 0.0f;
    float out;
    switch (act * 10 + grad) {
      default:
      case 10: out = v; break;
      case 11: out = v; break;
      case 12: out = 0.0f; break;
      case 30: out = (v > 0.0f) ? v : v * alpha; break;
      case 31: out = (r > 0.0f) ? v : v * alpha; break;
      case 32: out = 0.0f; break;
    }
    y[xi] = out * scale;
  }
}

static int floor_div(int a, int b) {
  int c = a / b;
  if (c * b > a) c--;
  return c;
}

void oracle_upfirdn2d(const float* x, const float* k, float* y, int major, int in_h, int in_w,
                      int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                      int pad_x0, int pad_x1, int pad_y0, int pad_y1) {
  int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
  int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
  for (int ma = 0; ma < major; ++ma)
    for (int oy = 0; oy < out_h; ++oy)
      for (int ox = 0; ox < out_w; ++ox)
        for (int mi = 0; mi < minor; ++mi) {
          /* upfirdn2d_kernel.cu:86-89,112-121 */
          int mid_x = ox * down_x + up_x - 1 - pad_x0;
          int mid_y = oy * down_y + up_y - 1 - pad_y0;
          int in_x = floor_div(mid_x, up_x);
          int in_y = floor_div(mid_y, up_y);
          int kernel_x = (in_x + 1) * up_x - mid_x - 1;
          int kernel_y = (in_y + 1) * up_y - mid_y - 1;
          float v = 0.0f;
          for (int yy = 0; kernel_y + yy * up_y < kh; ++yy)
            for (int xx = 0; kernel_x + xx * up_x < kw; ++xx) {
              int sy = in_y + yy, sx = in_x + xx;
              int ky = kernel_y + yy * up_y, kx = kernel_x + xx * up_x;
              if (sx < 0 || sy < 0 || sx >= in_w || sy >= in_h) continue;
              /* sk holds the FLIPPED kernel (:71-81) */
              float tap = k[(kh - 1 - ky) * kw + (kw - 1 - kx)];
              v += x[((int64_t)(ma * in_h + sy) * in_w + sx) * minor + mi] * tap;
            }
          y[((int64_t)(ma * out_h + oy) * out_w + ox) * minor + mi] = v;
        }
}
