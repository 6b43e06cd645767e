/*
 * rewriting_hip.h -- C ABI of librewriting_hip.so, the MI355X (gfx950) kernels behind the
 * rule-editing hot path of davidbau/rewriting.
 *
 * Boundary contract (level B3):
 *   - every pointer is a DEVICE pointer to contiguous float32 unless stated otherwise;
 *   - the CALLER owns and allocates every buffer (the Python host passes
 *     torch.Tensor.data_ptr()); the library allocates nothing and keeps no global state;
 *   - every entry point enqueues on the stream it is given (a hipStream_t passed as void*,
 *     NULL = the default stream) and returns immediately: stream-ordered and re-entrant;
 *   - return value 0 = success, otherwise a hipError_t (or RW_ERR_* below); the host
 *     wrapper raises RuntimeError(rw_error_string(code)) -- the reference's ops surface
 *     failures as RuntimeError through TORCH_CHECK
 *     (utils/stylegan2/op/fused_bias_act.cpp:6-8, upfirdn2d.cpp:8-10);
 *   - "nullable" pointers may be NULL, mirroring the reference's "empty tensor means
 *     absent" convention (utils/stylegan2/op/fused_bias_act_kernel.cu:62-63).
 *
 * Each declaration cites the reference interface it replaces (paths relative to the
 * reference repository root).
 */
#ifndef REWRITING_HIP_H
#define REWRITING_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rw_stream_t; /* hipStream_t */

#define RW_ERR_BAD_ARGUMENT 10001
#define RW_ERR_UNSUPPORTED  10002

/* 7 since round 5: the split-operand ("H16") entry points take their bounds as RW_BOUND_LANES-float vectors written
 * by plain stores (no memset, no atomic, no device scalar that one launch raises and the next reads) and their weight
 * scale BY VALUE (rw_split_weight_scale); rw_publish_scalar_f32 is gone.  (3: rw_solve_run_*, the whole-image 8x8 / 4x4
 * shapes, the point order of the packed F(4x4,3x3) weights; 4: rw_*_wino4h_*; 5: rw_dconv*; 6: rw_publish_scalar_f32.)
 * Buffers packed by an older library do not fit this one: repack, as the Python side does per weight version. */
int rw_abi_version(void);
const char* rw_error_string(int code);

/* ---------------------------------------------------------------------------------------
 * L1 native ops -- replace the two pybind entry points of utils/stylegan2/op/
 *
 * fp32 ONLY.  The reference's modules dispatch float, double and half (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * fused_bias_act_kernel.cu:79, upfirdn2d_kernel.cu:225); every entry of this library takes `float` pointers -- the
 * path it accelerates runs in fp32 end to end (BASELINE.json: "images within 1e-3 L-inf fp32").  A model cast with
 * .double() or .half() does not reach these entries: the Python wrappers (rewriting_amd/hip.py, the ctypes stub of
 * INTEGRATION.md) refuse any other dtype with an error that names this limit (the status of the refusal is
 * RW_ERR_UNSUPPORTED: nothing was launched), they never convert silently.
 * ------------------------------------------------------------------------------------- */

/* fused_bias_act(input, bias, refer, act, grad, alpha, scale)
 *   pybind: utils/stylegan2/op/fused_bias_act.cpp:11-20
 *   kernel: utils/stylegan2/op/fused_bias_act_kernel.cu:18-49 (host wrapper :52-98)
 * y[i] = act'(x[i] + b[(i / step_b) % size_b]) * scale, act*10+grad in {10,11,12,30,31,32}.
 * b nullable (no bias), ref nullable (required for grad=1 of act=3). */
int rw_fused_bias_act_f32(const float* x, const float* b, const float* ref, float* y,
                          int64_t n, int64_t step_b, int64_t size_b,
                          int act, int grad, float alpha, float scale, rw_stream_t stream);

/* grad_bias = grad_input.sum(all dims but 1)   (utils/stylegan2/op/fused_act.py:32-39)
 * g viewed as (outer, channels, inner); gb[c] = sum_{o,i} g[o][c][i]. */
int rw_bias_grad_f32(const float* g, float* gb, int64_t outer, int64_t channels, int64_t inner,
                     rw_stream_t stream);

/* upfirdn2d(input[major,H,W,minor], kernel[kh,kw], up_x, up_y, down_x, down_y,
 *           pad_x0, pad_x1, pad_y0, pad_y1) -> out[major,out_h,out_w,minor]
 *   pybind: utils/stylegan2/op/upfirdn2d.cpp:12-22
 *   kernel: utils/stylegan2/op/upfirdn2d_kernel.cu:52-137 (host wrapper :140-271)
 * out_h = (in_h*up_y + pad_y0 + pad_y1 - kh + down_y) / down_y  (:167-168); the caller
 * allocates y with that shape.  Any up/down/pad/kernel size (the reference compiles only
 * six modes, :178-210). */
int rw_upfirdn2d_f32(const float* x, const float* k, float* y,
                     int major, int in_h, int in_w, int minor, int kh, int kw,
                     int up_x, int up_y, int down_x, int down_y,
                     int pad_x0, int pad_x1, int pad_y0, int pad_y1, rw_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Generator forward -- replaces the torch expressions of utils/stylegan2/models.py
 * ------------------------------------------------------------------------------------- */

/* PixelNormL: y = x * rsqrt(mean(x^2, dim=1) + eps)      (models.py:609-614) */
int rw_pixel_norm_f32(const float* x, float* y, int batch, int dim, float eps, rw_stream_t stream);

/* EqualLinear.forward (models.py:503-511): y[b][o] = sum_i x[b*x_stride + i] * (w[o][i]*w_scale)
 * + bias[o]*b_scale, then, if act != 0, fused lrelu(alpha) * act_scale
 * (op/fused_act.py:85-86).  x_stride lets the caller pass latent[:, index] un-copied
 * (PickLatent, models.py:585-595). */
int rw_equal_linear_f32(const float* x, const float* w, const float* bias, float* y,
                        int batch, int in_dim, int out_dim, int64_t x_stride,
                        float w_scale, float b_scale, int act, float alpha, float act_scale,
                        rw_stream_t stream);

/* AdjustLatent (models.py:570-583): out[b][l][:] = avg + psi*(w[b] - avg) for l < n_latent;
 * avg nullable (no truncation). */
int rw_adjust_latent_f32(const float* w, const float* avg, float* out, int batch, int n_latent,
                         int dim, float psi, rw_stream_t stream);

/* ApplyStyle (models.py:616-620): y[b][c][p] = style[b][c] * x[b][c][p].  This output is the
 * rewriter's KEY (rewrite/ganrewrite.py:662-665). */
int rw_style_mul_f32(const float* x, const float* style, float* y, int batch, int channels,
                     int64_t hw, rw_stream_t stream);

/* wsq[o][i] = sum_{ky,kx} (w_scale * W[o][i][ky][kx])^2   -- first half of the demodulation
 * factor of DemodulatedConv2dF.forward (models.py:320-328), cached per weight version. */
int rw_weight_sqsum_f32(const float* w, float* wsq, int out_ch, int in_ch, int taps, float w_scale,
                        rw_stream_t stream);

/* demod[b][o] = rsqrt(sum_i style[b][i]^2 * wsq[o][i] + eps)      (models.py:326-327) */
int rw_demod_f32(const float* wsq, const float* style, float* demod, int batch, int out_ch,
                 int in_ch, float eps, rw_stream_t stream);

/* Repack a conv weight W[o][i][3][3] for the implicit-GEMM kernels.  wp receives
 * rw_packed_conv_weight_elems(out_ch, in_ch, mode) floats: first 9*Cin*Cout in slab order
 *   mode 0 (stride-1 conv, models.py:318-319):       wp[tap][i][o] = W[o][i][tap]
 *   mode 1 (stride-2 transposed conv, models.py:315-316), grouped by output parity
 *          phase (py,px): wp = [phase(0,0): 4 taps][phase(0,1): 2][phase(1,0): 2][phase(1,1): 1],
 *          each tap a contiguous [i][o] slab,
 * then (when Cout % 32 == 0 and Cin % 16 == 0, the shapes the halo-tile kernels take) the same
 * values in MFMA A-fragment order, so that a wave fetches its operand with 16-byte loads of 1 KiB
 * of consecutive addresses (IC = 16 input channels per chunk; 8 for mode 0 with Cout % 64 != 0):
 *   mode 0: wf[tap][i / IC][o / 32][kp / 4][lane][kp % 4],   kp < IC/2
 *   mode 1: wf[i / 16][o / 32][kp]{[slab / 4][lane][slab % 4] for slab < 8, then [lane] for slab 8}
 *   with value W[32 (o/32) + (lane & 31)][IC (i/IC) + 2 kp + (lane >> 5)][tap].
 * The scale 1/sqrt(9*Cin) is NOT folded in. */
long long rw_packed_conv_weight_elems(int out_ch, int in_ch, int mode);
int rw_pack_conv_weight_f32(const float* w, float* wp, int out_ch, int in_ch, int mode,
                            rw_stream_t stream);

/* Epilogue description shared by the two conv entry points.  y = acc * w_scale, then
 *   if demod: y *= demod[b][o]                                   (models.py:328)
 *   if noise: y += noise_w[0] * noise[b][pixel]                  (NoiseInjectionF :539-546)
 *   if act:   y = lrelu(y + bias[o], 0.2) * sqrt(2)              (FusedLeakyReLUF :622-626)
 * noise/bias/act are only legal for the stride-1 conv (for up layers the blur sits between). */
typedef struct rw_conv_epilogue {
  const float* style;    /* nullable: multiply the input by style[b][i] while loading (ApplyStyle fused) */
  const float* demod;    /* nullable */
  const float* noise;    /* nullable, (batch, out_h*out_w) */
  const float* noise_w;  /* device scalar, required iff noise */
  const float* bias;     /* nullable, (out_ch) */
  int act;               /* 0 / 1 */
} rw_conv_epilogue;

/* F.conv2d(x, scale*W, padding=1) [* demod]           (DemodulatedConv2dF, models.py:318-329)
 * x (B,Cin,H,W) -> y (B,Cout,H,W), wp from rw_pack_conv_weight_f32 mode 0.
 * impl: 0 = auto (fp32 MFMA implicit GEMM, v_mfma_f32_32x32x2_f32: halo-tile kernel when the map
 * is >= 24 wide (or 5..16 wide: column tiles of 2 x 16 / 4 x 8 pixels), else the im2col kernel, which splits K across the four waves of a workgroup when
 * the launch would otherwise leave most CUs idle), 1 = direct VALU kernel (cross-check), 2 = force
 * the im2col MFMA kernel, 3 = force the halo-tile MFMA kernel (RW_ERR_UNSUPPORTED if not applicable),
 * 4 = (transposed conv only) per-phase halo tiles, 5 = im2col MFMA kernel without split-K,
 * 6 = im2col MFMA kernel with split-K forced; transposed conv only: 7 = the quad tiles of the halo
 * kernel without output row 2H / column 2W (RW_ERR_UNSUPPORTED where the halo kernel does not apply), 8 = that row
 * and column only (disjoint writes: the two may be issued on different streams) -- three GEMMs over the last input row /
 * column of all images, any map size, in_ch % 16 == 0 and out_ch % 32 == 0. */
int rw_conv3x3_f32(const float* x, const float* wp, float* y, int batch, int in_ch, int out_ch,
                   int h, int w, float w_scale, const rw_conv_epilogue* ep, int impl,
                   rw_stream_t stream);

/* ToRGB (ToRGBF.forward, models.py:639-655) fused into the epilogue of the styled convolution that feeds it:
 *   rgb[b][c] = sum_o (scale * weight[c][o] * style[b][o]) * out[b][o] + bias[c] + skip[b][c]
 * computed from the activated outputs while they are still in registers.  y may be NULL: the feature map is
 * then not stored at all (the last layer of the generator, which only ToRGB consumes).
 * RW_ERR_UNSUPPORTED unless out_ch is 32 or 64 (the tile shapes in which one wave holds all out-channels of
 * its pixels), w >= 24 and in_ch % 16 == 0; callers then run rw_conv3x3_f32 and rw_to_rgb_f32. */
typedef struct rw_rgb_epilogue {
  const float* weight;   /* (3, out_ch) ToRGB conv weight */
  const float* style;    /* (batch, out_ch) ToRGB modulation */
  const float* bias;     /* (3) nullable */
  const float* skip;     /* (batch, 3, h, w) nullable: the upsampled running image */
  float* out;            /* (batch, 3, h, w) */
  float scale;           /* 1/sqrt(out_ch) */
} rw_rgb_epilogue;
int rw_conv3x3_to_rgb_f32(const float* x, const float* wp, float* y, int batch, int in_ch, int out_ch,
                          int h, int w, float w_scale, const rw_conv_epilogue* ep,
                          const rw_rgb_epilogue* rgb, rw_stream_t stream);

/* OPT-IN split-precision stride-1 convolution ("bf16x6"): same operation and epilogue as rw_conv3x3_f32,
 * computed on the bf16 matrix pipe with every fp32 operand split exactly into three bf16 pieces and the
 * six leading piece products accumulated in fp32 (relative error of a product < 2^-22; no range loss).
 * wb from rw_pack_conv_weight_bf16x3 (rw_packed_conv_weight_bf16x3_bytes bytes).  RW_ERR_UNSUPPORTED
 * unless w >= 24, in_ch % 16 == 0 and out_ch % 64 == 0 -- callers then use rw_conv3x3_f32, which is
 * also the default everywhere: nothing selects this path implicitly. */
long long rw_packed_conv_weight_bf16x3_bytes(int out_ch, int in_ch);
int rw_pack_conv_weight_bf16x3(const float* w, void* wb, int out_ch, int in_ch, rw_stream_t stream);
int rw_conv3x3_bf16x6_f32(const float* x, const void* wb, float* y, int batch, int in_ch, int out_ch,
                          int h, int w, float w_scale, const rw_conv_epilogue* ep, rw_stream_t stream);

/* The same stride-1 convolution (and the same fused epilogues) by the Winograd minimal-filtering algorithm
 * F(2x2, 3x3) in fp32: 16 instead of 36 multiplications per 2x2 output tile and channel pair, i.e. 2.25x fewer
 * matrix FLOPs for a result that is identical in exact arithmetic and of the direct kernel's error class in
 * fp32 (transform coefficients 0, +-1, +-1/2; fp32 MFMA accumulation).  rw_conv3x3_wino_supported() says which
 * shapes it takes (out_ch % 32 == 0, in_ch % 8 == 0, and w % 32 == 0 with h % 8 == 0, w == 16 with h % 16 == 0, or the
 * whole 8 x 8 / 4 x 4 maps -- several images per workgroup; these take ep->style == NULL, i.e. an input that already
 * carries its style, RW_ERR_UNSUPPORTED otherwise); elsewhere, and whenever the caller prefers the direct sum,
 * rw_conv3x3_f32 is the kernel.
 *   uf: rw_packed_conv_weight_wino_elems(out_ch, in_ch) = 16*out_ch*in_ch floats from rw_pack_conv_weight_wino_f32:
 *       U = G g G^T of every (o, i) filter in the A-fragment order of v_mfma_f32_16x16x4_f32
 *       uf[o / 32][i / 4][xi / 4][(o % 32) / 16][lane][xi % 4],  o = 32 (o/32) + 16 ((o%32)/16) + (lane & 15),
 *       i = 4 (i/4) + (lane >> 4), xi = 4 a + b the transform point (row a, column b of G g G^T).
 * rw_conv3x3_wino_to_rgb_f32: ToRGB in the epilogue as rw_conv3x3_to_rgb_f32 (out_ch == 32 only). */
int rw_conv3x3_wino_supported(int out_ch, int in_ch, int h, int w);
long long rw_packed_conv_weight_wino_elems(int out_ch, int in_ch);
int rw_pack_conv_weight_wino_f32(const float* w, float* uf, int out_ch, int in_ch, rw_stream_t stream);
int rw_conv3x3_wino_f32(const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch, int h,
                        int w, float w_scale, const rw_conv_epilogue* ep, rw_stream_t stream);
int rw_conv3x3_wino_to_rgb_f32(const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch,
                               int h, int w, float w_scale, const rw_conv_epilogue* ep,
                               const rw_rgb_epilogue* rgb, rw_stream_t stream);

/* The same convolution by Winograd F(4x4, 3x3) in fp32: 36 multiplications per 4x4 output tile and channel pair
 * (4x fewer matrix FLOPs than the direct sum, 1.78x fewer than F(2x2,3x3)).  Its transforms carry the constants
 * 4, 5, 8, 1/24: measured fp32 error 4e-6 .. 9e-6 of the output range per layer against 2e-7 .. 6e-7 for the two
 * kernels above.  The Python host makes it the DEFAULT inside the un-hooked forward of the whole generator (image
 * generation: the image tolerance of the path is 1e-3 L-inf, the measured deviation from the reference image 3e-5)
 * and nowhere else: a hooked or sliced model -- the statistics sweeps, goal maps, the solve's context and its
 * rendering -- runs F(2x2,3x3), so the same weights run hooked and un-hooked differ by 1e-5 .. 1e-4 on the image.
 * Shapes: out_ch % 32 == 0, in_ch % 8 == 0, w % 64 == 0, h % 8 == 0.
 *   uf: rw_packed_conv_weight_wino4_elems(out_ch, in_ch) = 36*out_ch*in_ch floats from
 *       rw_pack_conv_weight_wino4_f32: uf[o / 16][i / 4][xi / 4][lane][xi % 4], o = 16 (o/16) + (lane & 15),
 *       i = 4 (i/4) + (lane >> 4), xi = 6 a + b the transform point (row a, column b of G g G^T). */
int rw_conv3x3_wino4_supported(int out_ch, int in_ch, int h, int w);
long long rw_packed_conv_weight_wino4_elems(int out_ch, int in_ch);
int rw_pack_conv_weight_wino4_f32(const float* w, float* uf, int out_ch, int in_ch, rw_stream_t stream);
int rw_conv3x3_wino4_f32(const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch, int h,
                         int w, float w_scale, const rw_conv_epilogue* ep, rw_stream_t stream);

/* rw_conv3x3_wino4_f32 with ToRGB in its epilogue, as rw_conv3x3_wino_to_rgb_f32 (models.py:639-655): the last styled
 * convolution of the generator; rgb->out = ToRGB(act(conv + noise + bias)) + rgb bias + skip, the feature map is not
 * written.  Shapes: out_ch == 32, in_ch % 8 == 0, in_ch <= 512, w % 64 == 0, h % 8 == 0; uf from
 * rw_pack_conv_weight_wino4_f32. */
int rw_conv3x3_wino4_to_rgb_supported(int out_ch, int in_ch, int h, int w);
int rw_conv3x3_wino4_to_rgb_f32(const float* x, const float* uf, int batch, int in_ch, int out_ch, int h, int w,
                                float w_scale, const rw_conv_epilogue* ep, const rw_rgb_epilogue* rgb,
                                rw_stream_t stream);

/* F.conv_transpose2d(x, scale*W^T, stride=2, padding=0) [* demod]     (models.py:315-316,328)
 * x (B,Cin,H,W) -> y (B,Cout,2H+1,2W+1), wp from rw_pack_conv_weight_f32 mode 1. */
int rw_conv_transpose3x3s2_f32(const float* x, const float* wp, float* y, int batch, int in_ch,
                               int out_ch, int h, int w, float w_scale, const rw_conv_epilogue* ep,
                               int impl, rw_stream_t stream);

/* The quads y < H, x < W of the same transposed convolution (everything but output row 2H and column 2W, which
 * rw_conv_transpose3x3s2_f32 impl 8 writes) by the minimal-filtering algorithm F(2,2) in fp32: 25 instead of 36
 * multiplications per 2x2 block of quads and channel pair; coefficients 0, +-1 (the direct sum's error class).
 * Shapes: out_ch % 32 == 0, 16 <= in_ch <= 512, in_ch % 8 == 0, and w % 32 == 0 with h % 4 == 0, w == 16 with h % 8 == 0,
 * or the whole 8 x 8 / 4 x 4 maps (two / eight images per workgroup; these take style == NULL, i.e. an input that
 * already carries its style, RW_ERR_UNSUPPORTED otherwise).
 *   uf: rw_packed_conv_transpose_wino_elems(out_ch, in_ch) = 28*out_ch*in_ch floats from
 *       rw_pack_conv_transpose_wino_f32 (w = the (1,out_ch,in_ch,3,3) parameter as rw_pack_conv_weight_f32 takes it):
 *       uf[o / 16][i / 4][q][lane][xi % 4], xi = 4 q + e < 25 the point (rw_upwino.hip lists them), 25..27 zero.
 *   style (batch x in_ch) and demod (batch x out_ch) as in rw_conv_epilogue, either may be NULL. */
int rw_conv_transpose3x3s2_wino_supported(int out_ch, int in_ch, int h, int w);
long long rw_packed_conv_transpose_wino_elems(int out_ch, int in_ch);
int rw_pack_conv_transpose_wino_f32(const float* w, float* uf, int out_ch, int in_ch, rw_stream_t stream);
int rw_conv_transpose3x3s2_wino_f32(const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch,
                                    int h, int w, float w_scale, const float* style, const float* demod,
                                    rw_stream_t stream);

/* An upsampling StyledConv in ONE pass: F.conv_transpose2d(x, scale*W^T, stride=2) [* demod] -> Blur(pad 1,1) ->
 * NoiseInjectionF -> FusedLeakyReLUF  (models.py:315-316,328; 277-281; 539-546; 622-626):
 * x (B,Cin,H,W) -> y (B,Cout,2H,2W).  A stride-2 transposed 3x3 convolution followed by the 4x4 FIR is a stride-2
 * transposed convolution with their 6x6 composition, and each of its four output-parity phases is a 3x3 'same'
 * convolution of x: the phases run as 4*Cout virtual channels of the F(4x4,3x3) kernel above (its error class; like
 * it the host uses it inside the un-hooked whole-generator forward only), the (2H+1)x(2W+1) map is never written, there are no border strips.
 * Shapes: out_ch % 8 == 0, 8 <= in_ch <= 512, in_ch % 8 == 0, w % 64 == 0, h % 8 == 0.
 *   uf: rw_packed_conv_transpose_blur_wino4_elems(out_ch, in_ch) = 144*out_ch*in_ch floats from
 *       rw_pack_conv_transpose_blur_weight_wino4_f32(w, k4): w = the (1,out_ch,in_ch,3,3) parameter, k4 = the 4x4
 *       FIR buffer of the layer's Blur (already multiplied by 4); layout of rw_pack_conv_weight_wino4_f32 with
 *       virtual channel 4 o + 2 py + px.
 *   ep: style / demod / noise (B x 2H x 2W) + noise_w / bias + act as in rw_conv_epilogue.
 *   post_scale (batch x out_ch, nullable): a factor on the finished result -- the style of the convolution that
 *       consumes y, which then runs with style == NULL (the F(4x4,3x3) kernels skip the multiply in their loop). */
int rw_conv_transpose_blur_wino4_supported(int out_ch, int in_ch, int h, int w);
long long rw_packed_conv_transpose_blur_wino4_elems(int out_ch, int in_ch);
int rw_pack_conv_transpose_blur_weight_wino4_f32(const float* w, const float* k4, float* uf, int out_ch, int in_ch,
                                                 rw_stream_t stream);
int rw_conv_transpose3x3s2_blur_wino4_f32(const float* x, const float* uf, float* y, int batch, int in_ch,
                                          int out_ch, int h, int w, float w_scale, const rw_conv_epilogue* ep,
                                          const float* post_scale, rw_stream_t stream);

/* The three F(4x4,3x3) operations above with their 36 GEMMs on the 16-bit matrix pipe, fp32-equivalent by an EXACT
 * operand split ("H16", rw_wino4.hip): every transformed input V and weight U is written as the sum of two f16 numbers
 * (round to nearest twice: representation error <= 2^-22 relative), all four piece products are accumulated in fp32 by
 * v_mfma_f32_16x16x16_f16.  Per-product error <= 2^-21 of the product (fp32 multiply: 2^-24); accumulation, transforms
 * and epilogue are the fp32 kernels' -- same shapes, same results within the F(4x4,3x3) error class (tested at the same
 * bars), 36 MFMAs of ~17 cycles per k-quad instead of 32-cycle fp32 MFMAs that block the vector lanes.
 * Powers of two keep the pieces inside f16's normal range.  Neither of them is ever handed from one launch to the next
 * through a device scalar (round 4 did: a 4-byte value zeroed by a memset, raised with atomics and read back by the next
 * launch was occasionally read stale -- images 0.01 - 0.05 off; the forward of the reference is a pure function of its
 * inputs, utils/stylegan2/models.py:126-141, and so is this one now):
 *   uf: rw_packed_*_wino4h_elems floats from rw_pack_*_wino4h_f32: the wino4 layout with every float replaced by the
 *       32-bit word Uh | Ul << 16 of U * u_scale, + 4 trailing floats [1 / u_scale, u_scale, 0, 0] that NO kernel reads
 *       (they tell the format from the fp32 packing by size and let a host inspect what a buffer was packed with);
 *   u_scale (BY VALUE, pack entry points) / u_inv = 1 / u_scale (BY VALUE, convolution entry points): the power of two
 *       that rw_split_weight_scale() derives from max |U|; max |U| comes from rw_*_absmax_f32 (a bound, below), read
 *       back by the host ONCE per weight version;
 *   a BOUND on a map: RW_BOUND_LANES floats whose maximum is >= max |x| -- every wave of a consumer loads the vector
 *       (one float per lane, an ordinary vector load of data the previous launch stored plainly) and reduces it;
 *   x_amax (required): the bound of the input map (without the on-load style; the kernel multiplies by the style's
 *       largest factor itself) -- what the producer of x left in its y_amax, or rw_absmax_f32(x).  A bound that is too
 *       small overflows f16 (inf/NaN in the result); a bound that is 2^k too large costs k low bits of the smallest
 *       values only;
 *   y_amax (nullable): rw_bound_floats(elements of y) floats.  Every wave (or workgroup) of the producer stores ITS
 *       maximum into its own slot behind the first RW_BOUND_LANES floats -- plain stores, nothing is zeroed first,
 *       every slot that is read is written by the same launch -- and one 64-workgroup launch reduces the slots into
 *       y_amax[0 .. RW_BOUND_LANES): the bound of y.  Kernel boundaries order all of it like any feature map.
 * Everything else as in the fp32 entry points. */
#define RW_BOUND_LANES 64
/* floats of a y_amax buffer for a result of n_elems floats (64 + slots: every producer needs at most
 * 2048 + n_elems / 1024 + 1 of them; one that would need more measures its result with rw_absmax_f32 instead) */
long long rw_bound_floats(long long n_elems);
/* out (rw_bound_floats(n) floats) <- the bound of x[0 .. n) */
int rw_absmax_f32(const float* x, long long n, float* out, rw_stream_t stream);
/* u_scale = 2^(15 - e) with max |U| < 2^e (1 for max |U| == 0): |U u_scale| < 2^15 */
float rw_split_weight_scale(float u_absmax);
/* bound (rw_bound_floats(0) floats) <- max |U| over the transformed weights the matching rw_pack_* would store */
int rw_conv_weight_wino4h_absmax_f32(const float* w, int out_ch, int in_ch, float* bound, rw_stream_t stream);
int rw_conv_transpose_blur_weight_wino4h_absmax_f32(const float* w, const float* k4, int out_ch, int in_ch, float* bound,
                                                    rw_stream_t stream);
int rw_conv_transpose_weight_winoh_absmax_f32(const float* w, int out_ch, int in_ch, float* bound, rw_stream_t stream);
int rw_dconv_weight_absmax_f32(const float* w, int out_ch, int in_ch, float* bound, rw_stream_t stream);
int rw_dconv_transpose_blur_weight_absmax_f32(const float* w, const float* k4, int out_ch, int in_ch, float* bound,
                                              rw_stream_t stream);
long long rw_packed_conv_weight_wino4h_elems(int out_ch, int in_ch);
int rw_pack_conv_weight_wino4h_f32(const float* w, float* uf, int out_ch, int in_ch, float u_scale, rw_stream_t stream);
int rw_conv3x3_wino4h_f32(const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch, int h,
                          int w, float w_scale, const rw_conv_epilogue* ep, float u_inv, const float* x_amax,
                          float* y_amax, rw_stream_t stream);
int rw_conv3x3_wino4h_to_rgb_f32(const float* x, const float* uf, int batch, int in_ch, int out_ch, int h, int w,
                                 float w_scale, const rw_conv_epilogue* ep, const rw_rgb_epilogue* rgb, float u_inv,
                                 const float* x_amax, rw_stream_t stream);
long long rw_packed_conv_transpose_blur_wino4h_elems(int out_ch, int in_ch);
int rw_pack_conv_transpose_blur_weight_wino4h_f32(const float* w, const float* k4, float* uf, int out_ch, int in_ch,
                                                  float u_scale, rw_stream_t stream);
int rw_conv_transpose3x3s2_blur_wino4h_f32(const float* x, const float* uf, float* y, int batch, int in_ch,
                                           int out_ch, int h, int w, float w_scale, const rw_conv_epilogue* ep,
                                           const float* post_scale, float u_inv, const float* x_amax,
                                           float* y_amax, rw_stream_t stream);

/* The same three operations as DIRECT sums on the 16-bit matrix pipe (rw_dconv.hip, round 4): no transform at all -- an
 * input value is scaled by style 2^eV and split into its f16 pair ONCE per workgroup while it is staged into LDS, and read
 * nine times as a ready operand of v_mfma_f32_16x16x32_f16 (a lane's eight k values = [Vh c0..c3, Vl c0..c3] of four input
 * channels against [Uh c0..c3] twice, then [Ul c0..c3] twice: all four piece products, fp32 accumulation).  4x the
 * multiplies of F(4x4,3x3) and none of its transforms.  Error class: the direct fp32 kernels' (per-product 2^-21, no transform constants).
 * Shapes: in_ch % 16 == 0 (<= 512), w % 32 == 0; rw_dconv3x3: out_ch % 32 == 0, h % 16 == 0; the transposed form:
 * out_ch % 16 == 0, h % 8 == 0; to_rgb: out_ch == 32.
 *   wp: rw_packed_dconv_*_elems floats from rw_pack_dconv_*_f32: wp[o / 16][9 (i / 16) + tap][Uh | Ul][lane = 16 ((i % 16) / 4)
 *       + o % 16][4 halves: channels 4 ((i % 16) / 4) + (0..3)] (the kernels double them into the operand) + 4 trailing
 *       floats [1 / u_scale, u_scale, 0, 0] (not read by any kernel).  The transposed form packs the four output-parity phases of conv_transpose (*) blur as
 *       blocks of 16 virtual channels: block 4 (o / 16) + 2 py + px.
 * Two kernel families behind the same entry points: one-role workgroups (two per CU), and -- where in_ch >= 32, the
 * epilogue carries a style, w % 64 == 0 and (rw_dconv3x3_f32) out_ch % 64 == 0, h % 8 == 0 -- workgroups of eight
 * multiplying and four staging waves (RW_DCONV_V=1 forces the first).  Measured on MI355X: as fast as the split F(4x4,3x3)
 * kernels on the 512^2 / 1024^2 layers, not faster (the 16-bit pipe retires an MFMA per 20 cycles and SIMD at ~1.7 GHz
 * under this load: DESIGN.md section 4.4) -- the Python host leaves them opt-in (RW_MM_DIRECT16=1).
 * u_scale / u_inv / x_amax / y_amax / ep / post_scale / rgb: as in the wino4h entry points. */
int rw_dconv3x3_supported(int out_ch, int in_ch, int h, int w);
long long rw_packed_dconv_weight_elems(int out_ch, int in_ch);
int rw_pack_dconv_weight_f32(const float* w, float* wp, int out_ch, int in_ch, float u_scale, rw_stream_t stream);
int rw_dconv3x3_f32(const float* x, const float* wp, float* y, int batch, int in_ch, int out_ch, int h, int w,
                    float w_scale, const rw_conv_epilogue* ep, float u_inv, const float* x_amax, float* y_amax,
                    rw_stream_t stream);
/* rw_dconv3x3_f32 that ALSO leaves the channel sums of the ToRGB which reads its result (ToRGBF.forward, models.py:639-655;
 * the reference's generator applies it to the output of every second styled convolution, :126-131): rgb->out receives
 * rw_dconv3x3_rgb_partials(out_ch) = out_ch / 32 partial images, layout (partial, batch, 3, h, w) -- one per wave's 32
 * out-channels, summed while the activated values are in registers; rgb->weight (3, out_ch), rgb->style (batch, out_ch),
 * rgb->scale as in rw_rgb_epilogue, rgb->bias / rgb->skip are NOT used here.  rw_rgb_combine_f32 finishes the ToRGB:
 *   out[b][c][p] = sum_k partial[k][b][c][p] + bias[c] + skip[b][c][p]      (bias, skip nullable; hw % 4 == 0).
 * The feature map y is written as by rw_dconv3x3_f32 (the next layer reads it); the second pass over it (rw_to_rgb_f32)
 * disappears.  Same shapes as rw_dconv3x3_f32, and the same choice of kernel: with a style on load (ep->style), in_ch >= 32,
 * 64 | out_ch, 8 | h, 64 | w the specialised persistent kernel (twelve waves per compute unit), the one-role kernels otherwise
 * (RW_DCONV_V=1: always) -- y and the partial images are bit-identical to rw_dconv3x3_f32's y + a float32 channel sum. */
int rw_dconv3x3_rgb_partials(int out_ch);
int rw_dconv3x3_rgb_partial_f32(const float* x, const float* wp, float* y, int batch, int in_ch, int out_ch, int h, int w,
                                float w_scale, const rw_conv_epilogue* ep, const rw_rgb_epilogue* rgb, float u_inv,
                                const float* x_amax, float* y_amax, rw_stream_t stream);
int rw_rgb_combine_f32(const float* partials, int n_part, const float* bias, const float* skip, float* y, int batch,
                       int64_t hw, rw_stream_t stream);
int rw_dconv3x3_to_rgb_supported(int out_ch, int in_ch, int h, int w);
int rw_dconv3x3_to_rgb_f32(const float* x, const float* wp, int batch, int in_ch, int out_ch, int h, int w,
                           float w_scale, const rw_conv_epilogue* ep, const rw_rgb_epilogue* rgb, float u_inv,
                           const float* x_amax, rw_stream_t stream);
int rw_dconv_transpose_blur_supported(int out_ch, int in_ch, int h, int w);
long long rw_packed_dconv_transpose_blur_weight_elems(int out_ch, int in_ch);
int rw_pack_dconv_transpose_blur_weight_f32(const float* w, const float* k4, float* wp, int out_ch, int in_ch,
                                            float u_scale, rw_stream_t stream);
int rw_dconv_transpose3x3s2_blur_f32(const float* x, const float* wp, float* y, int batch, int in_ch, int out_ch, int h,
                                     int w, float w_scale, const rw_conv_epilogue* ep, const float* post_scale,
                                     float u_inv, const float* x_amax, float* y_amax, rw_stream_t stream);

/* The upsampling StyledConv in one pass at the transposed convolution's OWN multiply count (rw_tconv.hip, round 5):
 * utils/stylegan2/models.py:313-316 F.conv_transpose2d(stride=2) as a direct sum on the 16-bit matrix pipe (exact f16 operand
 * split, fp32 accumulation; 9 multiplies per input position and channel pair where the phase kernels above spend 36), its
 * (2H+1) x (2W+1) result kept in LDS, the 4x4 FIR of Blur(pad 1,1) (:275-281) read from there, NoiseInjection (:539-546),
 * FusedLeakyReLU (:232-257) and post_scale in the same epilogue: x (B, in_ch, H, W) -> y (B, out_ch, 2H, 2W).
 *   wp: rw_packed_dconv_weight_elems floats from rw_pack_dconv_weight_f32 -- the PLAIN packing of dconv.weight (no
 *       composition with the blur), u_inv its scale; k4: the 4x4 FIR buffer (already multiplied by 4);
 *   shapes: in_ch % 16 == 0 (<= 512), out_ch % 16 == 0, h % 16 == 0, w % 32 == 0;
 *   ep / post_scale / x_amax / y_amax: as in rw_dconv_transpose3x3s2_blur_f32;
 *   the kernel's form is picked by in_ch (32 .. 128: one persistent workgroup per CU with specialised waves; otherwise one
 *   8-wave workgroup per CU and tile); environment RW_TCONV_TY = 0 / 16 / 8 forces a form, RW_TCONV_GRID the persistent
 *   form's workgroup count (diagnostics). */
int rw_tconv_blur_supported(int out_ch, int in_ch, int h, int w);
int rw_tconv_blur_f32(const float* x, const float* wp, const float* k4, float* y, int batch, int in_ch, int out_ch, int h,
                      int w, float w_scale, const rw_conv_epilogue* ep, const float* post_scale, float u_inv,
                      const float* x_amax, float* y_amax, rw_stream_t stream);

/* rw_conv_transpose3x3s2_wino_f32 (the F(2,2) quads of the stride-2 transposed convolution) with its 25 GEMMs on the
 * 16-bit matrix pipe and the exact f16 operand split of the wino4h entry points (rw_upwino.hip): one
 * v_mfma_f32_16x16x32_f16 per point takes the two k-quads of an 8-channel interval (25 MFMAs of ~17 cycles per 8
 * channels against 50 fp32 MFMAs of 32 that block the vector lanes).  Coefficients 0, +-1 as before: the direct sum's
 * error class plus <= 2^-21 per product (tested at the fp32 kernel's bars).  Wide maps only: w % 32 == 0, h % 4 == 0,
 * 16 <= in_ch <= 512, in_ch % 8 == 0, out_ch % 32 == 0.
 *   uf: rw_packed_conv_transpose_winoh_elems = 16*out_ch*in_ch + 4 floats from rw_pack_conv_transpose_winoh_f32 (the
 *       25 points carry 16 distinct weights): uf[o / 16][i / 8][wi = 0..15][lane = 16 ((i % 8) % 4) + o % 16]
 *       [{0: Uh, 1: Ul}], each 32-bit word the f16 pair of channels (i, i + 4) of the interval; trailer
 *       [1 / u_scale, u_scale, 0, 0] (not read by any kernel);
 *   u_scale / u_inv / x_amax (the bound of x before the style): as for rw_conv3x3_wino4h_f32. */
int rw_conv_transpose3x3s2_winoh_supported(int out_ch, int in_ch, int h, int w);
long long rw_packed_conv_transpose_winoh_elems(int out_ch, int in_ch);
int rw_pack_conv_transpose_winoh_f32(const float* w, float* uf, int out_ch, int in_ch, float u_scale, rw_stream_t stream);
int rw_conv_transpose3x3s2_winoh_f32(const float* x, const float* uf, float* y, int batch, int in_ch, int out_ch,
                                     int h, int w, float w_scale, const float* style, const float* demod, float u_inv,
                                     const float* x_amax, rw_stream_t stream);

/* NoiseInjectionF (models.py:535-546): y[b][c][p] = x[b][c][p] + noise_w[0] * noise[b][p] */
int rw_noise_add_f32(const float* x, const float* noise, const float* noise_w, float* y,
                     int batch, int channels, int64_t hw, rw_stream_t stream);

/* BlurF(pad=(1,1)) -> NoiseInjectionF -> FusedLeakyReLUF in one pass for upsampling layers
 * (models.py:277-281,481-485,539-546,622-626): x (B,C,2H+1,2W+1) -> y (B,C,2H,2W);
 * k4 = the 4x4 FIR buffer (already multiplied by 4). */
int rw_blur_noise_act_f32(const float* x, const float* k4, const float* noise, const float* noise_w,
                          const float* bias, float* y, int batch, int channels, int out_h, int out_w,
                          rw_stream_t stream);
/* The same with a per (image, channel) factor on the result (batch x channels, nullable): the style of the
 * convolution that consumes y, handed over pre-multiplied. */
int rw_blur_noise_act_scaled_f32(const float* x, const float* k4, const float* noise, const float* noise_w,
                                 const float* bias, const float* post_scale, float* y, int batch, int channels,
                                 int out_h, int out_w, rw_stream_t stream);
/* ... and y_amax (nullable, rw_bound_floats(batch * channels * out_h * out_w) floats) receives the bound of the result,
 * post_scale included: the x_amax of a split-operand (wino4h / winoh) convolution that reads y. */
int rw_blur_noise_act_amax_f32(const float* x, const float* k4, const float* noise, const float* noise_w,
                               const float* bias, const float* post_scale, float* y, int batch, int channels,
                               int out_h, int out_w, float* y_amax, rw_stream_t stream);

/* ToRGBF (models.py:628-655): y[b][c][p] = sum_i (W[c][i]*style[b][i]*w_scale) x[b][i][p]
 * + bias[c] + skip[b][c][p]; out channels = 3, skip nullable. */
int rw_to_rgb_f32(const float* x, const float* w, const float* style, const float* bias,
                  const float* skip, float* y, int batch, int in_ch, int64_t hw, float w_scale,
                  rw_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Key statistics -- replaces RunningSecondMoment.add (utils/runningstats.py:1086-1097,
 * progress_addbmm :1181-1190) and RunningVariance.add's reductions (:763-788)
 * ------------------------------------------------------------------------------------- */

/* mom2 (C,C) += a^T a on fp32 MFMA.
 *   layout 0: a is (rows, C) row-major, as the reference passes it
 *             (rewrite/ganrewrite.py:90-93);
 *   layout 1: a is (batch, C, hw) NCHW, rows = batch*hw -- the key map as the generator
 *             produced it (no permute copy).
 * workspace: >= rw_second_moment_workspace_bytes(C, rows) bytes of device scratch. */
int64_t rw_second_moment_workspace_bytes(int channels, int64_t rows);
int rw_second_moment_f32(const float* a, float* mom2, int64_t rows, int channels, int64_t hw,
                         int layout, void* workspace, rw_stream_t stream);

/* per-channel sum and sum of squares over rows (same layouts); sums (2,C) are OVERWRITTEN. */
int rw_channel_sums_f32(const float* a, float* sums, int64_t rows, int channels, int64_t hw,
                        int layout, int square_input, rw_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * The rank-constrained solve -- replaces the body of ProgressiveGanRewriter.insert
 * (rewrite/ganrewrite.py:254-298) and of linear_insert (:201-252) for a SeqStyleGAN2 layer
 * (stride-1 or upsampling): forward of target_model, L1 loss, backward to dconv.weight,
 * torch.optim.Adam step, projection.
 * Arithmetic: SURVEY.md section 10.
 * ------------------------------------------------------------------------------------- */

typedef struct rw_solve_problem {
  /* shapes */
  int out_ch, in_ch, h, w, rank;
  /* constants of the layer / goal (device) */
  const float* key;      /* (in_ch, h, w)   goal_in.fmap  = adain output crop            */
  const float* style;    /* (in_ch)         goal_in.style                               */
  const float* val;      /* (out_ch, h, w)  goal_out.fmap                               */
  const float* bias;     /* (out_ch)        activate.bias; NULL = the target is the demodulated
                          *                 convolution alone (SeqTinyStyleGanRewriter, ganrewrite.py:731-738):
                          *                 no blur / noise / bias / activation, val is (out_ch, conv map) */
  const float* noise;    /* (h*w)           RandomState(0).randn(1,h*w)  (quirk Q1)     */
  const float* noise_w;  /* device scalar   noise.weight                                */
  const float* context;  /* (rank, in_ch)   orthonormal rows                            */
  const float* ortho;    /* (out_ch,in_ch,9) W0 - P(W0), nullable when !low_rank_insert  */
  /* state (device, updated in place) */
  float* weight;         /* (out_ch,in_ch,3,3) dconv.weight[0]                          */
  float* exp_avg;        /* Adam m */
  float* exp_avg_sq;     /* Adam v */
  /* per-step tables (device, length niter), computed on the host in double like torch.optim.Adam */
  const float* step_size;   /* lr / (1 - beta1^t)      */
  const float* bc2_sqrt;    /* sqrt(1 - beta2^t)       */
  int32_t* step_counter;    /* device int: index of the LAST step taken; the library pre-increments it, so it
                             * starts at -1 and step t reads step_size[t], bc2_sqrt[t], writes losses[t] */
  float* losses;            /* (niter) l1 loss of every step, written by the library        */
  /* scratch (device); element counts from rw_solve_scratch_elems().  P64 = ceil64(positions of the map the
   * convolution writes): h*w, or (2h+1)*(2w+1) for an upsampling target */
  float* conv;           /* (ksplit, out_ch, P64)  partial conv sums                      */
  float* wsq;            /* (ksplit, out_ch)                                              */
  float* gd;             /* (out_ch, P64)  g_pre*demod, zero padded                       */
  float* c2;             /* (2*out_ch)     s^2 * demod^3 * sum_p g_pre*conv | per-channel loss */
  float* grad;           /* (out_ch,in_ch,9) only for low_rank_gradient, else nullable     */
  int ksplit;
  float beta1, beta2, eps, w_scale;
  int low_rank_gradient;
  float one_minus_beta1, one_minus_beta2;   /* (1 - beta) evaluated in double on the host */
  /* upsampling (odd) layers: target = conv_transpose2d stride 2 -> blur -> noise -> activate
   * (utils/stylegan2/models.py:315-316,277-281).  key is (in_ch,h,w); val is (out_ch,2h,2w);
   * noise is (2h*2w); conv/gd rows are padded counts of the (2h+1)x(2w+1) pre-blur map. */
  int upsample;
  const float* blur_k;   /* (4,4) FIR buffer of the layer's BlurF, required iff upsample */
  /* linear_insert (rewrite/ganrewrite.py:201-252): optimise Lambda (out_ch,rank,3,3) with
   * weight = ortho(=W0) + Lambda.context; exp_avg / exp_avg_sq then hold Lambda's Adam state. */
  int linear_insert;
  float* lambda;
} rw_solve_problem;

/* split-K factor for a convolution-output map of h x w positions ((2h+1) x (2w+1) of the key for upsampling) */
int rw_solve_ksplit(int out_ch, int in_ch, int h, int w);
/* 1 when rw_solve_step_f32 takes this shape (h, w of the key crop), 0 otherwise -- like every other *_supported
 * entry point: out_ch % 64, in_ch % 16, <= 64 KB of LDS for the blur staging of an upsampling target (plain == 0)
 * and for the rank-r projection (constrained != 0).  rw_solve_step_f32 runs the same check before its first launch
 * and returns RW_ERR_UNSUPPORTED / RW_ERR_BAD_ARGUMENT. */
int rw_solve_supported(int out_ch, int in_ch, int h, int w, int upsample, int plain, int constrained);
/* sizes[0..4] = element counts of conv, wsq, gd, c2, grad for this shape; sizes[5] = the split-K factor they were
 * sized for, i.e. the value rw_solve_problem.ksplit must carry (the caller passes long long sizes[6]) */
int rw_solve_scratch_elems(int out_ch, int in_ch, int h, int w, int upsample, long long* sizes);
/* one iteration `it` (loss, gradient, Adam); project != 0 also applies W <- ortho + P(W) */
int rw_solve_step_f32(const rw_solve_problem* p, int project, rw_stream_t stream);
/* The same solve as ONE launch for iterations [it_begin, it_end) of niter (rewrite/ganrewrite.py:271-294 incl. the
 * projection rule `it % piter == 0 or it == niter - 1` when low_rank_insert != 0): every workgroup owns two out-channels
 * for the whole run, weight and Adam moments stay in registers, the key crop in LDS; no scratch of rw_solve_problem but
 * lpart (niter * out_ch floats) is used, the step counter is not touched, losses[it] is written for the iterations
 * run, state is read from / written back to weight, exp_avg, exp_avg_sq (consecutive calls continue each other).
 * rw_solve_run_supported: 1 for stride-1 targets (with or without bias), in_ch % 64 == 0, in_ch <= 512,
 * out_ch % 2 == 0, rank <= 8, no linear_insert, and either in_ch * ((h+2)(w+1) + 3 | 1) floats + scratch within 160 KB of
 * LDS (the crop resident) or w <= 16 (the crop streamed, see below); else 0 and rw_solve_step_f32 is the way. */
int rw_solve_run_supported(int out_ch, int in_ch, int h, int w, int rank, int upsample, int linear_insert);
/* Round 4: crops that do not fit the LDS (w <= 16: the watermark erase's whole 16 x 16 maps of 512 channels) run in one
 * launch too -- every thread streams ITS channel's crop row by row from a position-major copy (built in `lpart`'s tail),
 * and with low_rank_gradient the gradient phase needs no key at all (it correlates g with the one-channel maps
 * d_r^T key).  rw_solve_run_scratch_elems: floats of `lpart` = niter * out_ch (rounded up to 4) + that copy where the
 * crop is streamed. */
long long rw_solve_run_scratch_elems(int out_ch, int in_ch, int h, int w, int niter);
int rw_solve_run_f32(const rw_solve_problem* p, int it_begin, int it_end, int niter, int piter, int low_rank_insert,
                     float* lpart, rw_stream_t stream);
/* W <- W - P(W) + amount*P(1)  (zero(), ganrewrite.py:190-195) and ortho = W - P(W) helpers */
int rw_project_weight_f32(const float* w, const float* context, const float* base, float* out,
                          int out_ch, int in_ch, int taps, int rank, float scale_w, rw_stream_t stream);

/* ---- gradients of the demodulated 3x3 convolution (the autograd path of `insert` on targets the fused solver does
 * not restate: rewrite/ganrewrite.py:265-283 through utils/stylegan2/models.py:313-329).  Backward-to-input needs
 * no entry point of its own: it is rw_conv3x3_f32 on the transposed (stride 1: and flipped) weights.
 *
 * rw_conv_wgrad_f32: dw[o][i][tap] = scale * sum_{b, p} (g[b][o][p] * gscale[b][o]) * (xcol_b[(i, tap)][p] * xscale[b][i])
 *   g  (batch, out_ch, CH, CW)  gradient w.r.t. the map the convolution writes: (h, w), or (2h+1, 2w+1) when
 *                               upsample != 0 (the stride-2 transposed convolution)
 *   x  (batch, in_ch, h, w)     the convolution's input map;  gscale (batch, out_ch), xscale (batch, in_ch): nullable
 *   scratch  rw_conv_wgrad_ksplit(...) * out_ch * in_ch * 9 floats (split-K partial sums, reduced in a fixed order)
 *   dw (out_ch, in_ch, 3, 3) is overwritten. */
int rw_conv_wgrad_ksplit(int batch, int in_ch, int out_ch, int h, int w, int upsample);
int rw_conv_wgrad_f32(const float* g, const float* x, const float* gscale, const float* xscale, float* scratch,
                      float* dw, int batch, int in_ch, int out_ch, int h, int w, int upsample, float scale,
                      rw_stream_t stream);
/* out[r] = sum_j a[r][j] * b[r][j] for `rows` rows of length n (per-(image, channel) sums over a feature map) */
int rw_rowdot_f32(const float* a, const float* b, float* out, long long rows, long long n, rw_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* REWRITING_HIP_H */
