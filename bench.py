"""Benchmark of the rule-editing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--batch B]

Default workload = BASELINE.json's metric: images/sec of the StyleGANv2-1024 sequential
generator forward (SeqStyleGAN2 size 1024, mconv='seq', truncation 0.5, synthetic seeded
weights, z = standard_z_sample seed 1), one "step" = one forward pass over a batch of B seeds per
GPU (default 64) with inputs resident in HBM.  For N>1 either launch it under torch.distributed.run (one rank
per GPU, RCCL) or just run `python bench.py --gpus N`: without RANK/WORLD_SIZE in the environment it re-executes
itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`.  Seeds are
partitioned rank-wise (no data-path collective, weak scaling); the timed region is bracketed by barrier +
synchronize and the MAX over ranks is reported.

Rank 0 prints ONE JSON line with `roofline` (the dominant conv kernel: the matrix FLOPs its launches ISSUE /
their HIP-event time, against the 157.3 TFLOP/s fp32-MFMA peak; the direct-sum figure beside it) and
`cpu_baseline` (the reference's own files when /root/reference is present, else the oracle restatement, timed
on the host cores on a bounded sample at the best of several thread counts; N=1 only) and `extra`: the other
half of BASELINE.json's metric (seconds per rank-1 edit = 1000-seed key statistics + 2001-step solve), the
256^2 forward (configs[1]), the key-statistics sweeps at layers 8/10/14 (configs[3]; each with `context_layers`, the
roofline of every layer's convolution kernel), a one-rank RCCL self-check (`rccl_one_rank`) and the five-variant
watermark job (configs[4]), each timed in this process; `step` (the whole step against the MFMA-issue and HBM
roofs) and `parity` (the output of the last timed step against the reference-generated fixture).

Other workloads (parity-test configurations of BASELINE.json, not the headline line):
  ffhq256   StyleGANv2-256 forward, batch 64         (configs[1])
  edit      horse->hat rank-1 edit at layer 8 of the 256 model: 1000-seed key statistics +
            2001-step solve, seconds per edit       (configs[2])
  sweep     key-statistics sweep only (seeds/s), sharded over ranks with one all-reduce (configs[3]);
            --layer 8|10|14 --seeds 10000
  watermark the five watermark.sh variants (erase edits, configs[4]) as replicas, one per rank, + sample sets
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
F16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: dense f16 / bf16 MFMA (the split-operand kernels' pipe)
HBM_PEAK_GBS = 8000.0


def w4h_point_split(in_ch):
    """rw_wino4.hip's w4h_point_split: which of the two split-operand F(4x4,3x3) kernels a launch picks."""
    e = os.environ.get('RW_W4H_PS')
    return in_ch >= 64 if e is None else int(e) != 0


def conv_flops(model_size, channel_multiplier=2):
    """2*MACs of every 3x3 styled conv per image."""
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
          256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
    total = 2 * 9 * 512 * 512 * 16                           # layer2 at 4x4
    cin, res = 512, 4
    while res < model_size:
        cout = ch[res * 2]
        total += 2 * 9 * cin * cout * res * res              # stride-2 transposed conv (input res)
        res *= 2
        total += 2 * 9 * cout * cout * res * res             # stride-1 conv
        cin = cout
    return total


def conv_kernel_name(out_ch, in_ch, width, upsample):
    """Which kernel rw_conv3x3_f32 / rw_conv_transpose3x3s2_f32 dispatch to (impl 0), mirroring
    launch_halo / launch_up_halo / launch_batch in rewriting_amd/csrc/rw_conv.hip."""
    ok = in_ch % 16 == 0 and in_ch <= 1024 and out_ch % 32 == 0
    if upsample:
        if ok and 5 <= width <= 8 and out_ch % 128 == 0:
            return 'conv_up_halo_kernel<4, 1, 16, 8>'
        if ok and (width >= 24 or 9 <= width <= 16):
            tw = 32 if width >= 24 else 16
            return ('conv_up_halo_kernel<2, 2, 16, %d>' if out_ch % 64 == 0 else 'conv_up_halo_kernel<1, 4, 16, %d>') % tw
        return 'conv_mfma(+ksplit)_kernel [4 phases]'
    if ok and 5 <= width <= 8 and out_ch % 128 == 0:
        return 'conv_halo_kernel<2, 1, 2, 2, 16, true, 8>'
    if ok and (width >= 24 or 9 <= width <= 16):
        tw = 32 if width >= 24 else 16
        if out_ch % 128 == 0:
            return 'conv_halo_kernel<2, 2, 2, 2, 16, true, %d>' % tw
        return ('conv_halo_kernel<2, 2, 1, 4, 16, true, %d>' if out_ch % 64 == 0
                else 'conv_halo_kernel<1, 4, 1, 4, 8, true, %d>') % tw
    return 'conv_mfma(+ksplit)_kernel'


def tconv_auto_form(in_ch, out_ch):
    """The form rw_tconv_blur_f32 picks when RW_TCONV_TY is unset (csrc/rw_tconv.hip, the launcher): the 32-out-channel
    kernel for the input-channel range RW_TCONV_N32 = "lo:hi" (default: none), a persistent kernel for 32 .. 128 input
    channels otherwise, one eight-wave workgroup per CU above."""
    lo, hi = 1, 0
    spec = os.environ.get('RW_TCONV_N32')
    if spec:
        try:
            lo, hi = (int(v) for v in spec.split(':'))
        except ValueError:
            lo, hi = 1, 0
    if out_ch % 32 == 0 and lo <= in_ch <= hi:
        return '32'
    if 32 <= in_ch <= 128:
        return '2' if os.environ.get('RW_TCONV_PERSISTENT') == '2' else '0'
    return '16'


class ConvTimer:
    """HIP events around every implicit-GEMM conv call, on the stream the kernels are launched on
    (torch's current stream), attributed to the kernel the call dispatches to.  Installed for
    the timed region only."""

    def __init__(self):
        self.calls = []          # (kernel name, start event, end event, flops)

    def install(self):
        from rewriting_amd import hip
        self._orig = (hip.conv3x3, hip.conv_transpose3x3s2, hip.conv3x3_bf16x6, hip.conv3x3_to_rgb,
                      hip.conv3x3_wino, hip.conv3x3_wino_to_rgb, hip.conv3x3_wino4, hip.conv_transpose3x3s2_wino,
                      hip.conv_transpose3x3s2_blur_wino4, hip.conv3x3_wino4_to_rgb,
                      hip.conv3x3_direct16, hip.conv_transpose3x3s2_blur_direct16, hip.conv3x3_direct16_to_rgb)
        self._orig_fused = hip.conv_transpose3x3s2_blur_fused
        self._orig_rgbp = hip.conv3x3_direct16_rgb_partial
        timer = self

        def fused(x, wp, k4, out_ch, w_scale, *a, **k):
            # csrc/rw_tconv.hip: the launcher's form by input channels (RW_TCONV_TY overrides)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = timer._orig_fused(x, wp, k4, out_ch, w_scale, *a, **k)
            e.record()
            b, i, h, w = x.shape
            form = os.environ.get('RW_TCONV_TY') or tconv_auto_form(i, out_ch)
            name = {'0': 'tconv_blur_ws_kernel' if i >= 32 else 'tconv_blur_t8_kernel', '2': 'tconv_blur_pp_kernel',
                    '16': 'tconv_blur_t16_kernel',
                    '32': 'tconv_blur_n32_kernel' if out_ch % 32 == 0 else 'tconv_blur_t8_kernel'}.get(form, 'tconv_blur_t8_kernel')
            timer.calls.append((name, s, e, 2.0 * 9 * i * out_ch * h * w * b, 4.0 * (b * i * h * w + b * out_ch * 4 * h * w)))
            return y
        hip.conv_transpose3x3s2_blur_fused = fused

        def wrap(fn, upsample, split=False, wino=None):
            def inner(x, wp, out_ch, w_scale, *a, **k):
                if k.get('impl') == 8:        # border strips of an up layer, issued on a third stream beside
                    return fn(x, wp, out_ch, w_scale, *a, **k)      # the tiles call that carries the layer's FLOPs
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                y = fn(x, wp, out_ch, w_scale, *a, **k)
                e.record()
                b, i, h, w = x.shape
                lib = hip.lib()
                if wino == 'd16p':            # the direct sum that also leaves the ToRGB's channel sums
                    ws = (k.get('style') is not None and i >= 32 and w % 64 == 0 and out_ch % 64 == 0 and h % 8 == 0
                          and os.environ.get('RW_DCONV_V') != '1')
                    name = 'dconv_ws_w2_rgbp_kernel' if ws else \
                        'dconv_w%d_rgbp_kernel' % (4 if out_ch % 128 == 0 else 2 if out_ch % 64 == 0 else 1)
                elif wino in ('d16', 'd16up', 'd16rgb'):
                    # rw_dconv.hip: the specialised kernels take a style on load, >= 32 channels, maps 64 columns wide
                    ws = k.get('style') is not None and i >= 32 and w % 64 == 0 and os.environ.get('RW_DCONV_V') != '1'
                    if wino == 'd16up':
                        name = 'dconv_ws_up_kernel' if ws else 'dconv_up_kernel'
                    elif wino == 'd16rgb':
                        name = 'dconv_rgb_kernel'
                    elif ws and out_ch % 64 == 0 and h % 8 == 0:
                        name = 'dconv_ws_w2_kernel'
                    else:
                        name = 'dconv_w%d_kernel' % (4 if out_ch % 128 == 0 else 2 if out_ch % 64 == 0 else 1)
                elif wino == 'up':
                    if wp.numel() == lib.rw_packed_conv_transpose_winoh_elems(out_ch, i):
                        name = 'conv_up_winoh_kernel'             # operands split into f16 pairs (16-bit matrix pipe)
                    else:
                        name = {16: 'conv_up_wino_narrow_kernel', 8: 'conv_up_wino_8x8_kernel',
                                4: 'conv_up_wino_4x4_kernel'}.get(w, 'conv_up_wino_kernel')
                elif wino in ('up4', 'f4rgb', 'f4'):
                    # rw_wino4.hip picks the no-style variants when the input map already carries the style
                    ns = k.get('style') is None
                    h16 = wp.numel() == (lib.rw_packed_conv_transpose_blur_wino4h_elems(out_ch, i) if wino == 'up4'
                                         else lib.rw_packed_conv_weight_wino4h_elems(out_ch, i))
                    ps = '_ps' if h16 and w4h_point_split(i) else ''
                    if wino == 'up4':         # transposed conv + blur + noise + activation in one pass
                        name = ('conv_up_wino36h%s_kernel' % ps if h16 else
                                'conv_up_wino36_ns_kernel' if ns else 'conv_up_wino36_kernel')
                    elif wino == 'f4rgb':
                        name = ('conv_wino36h_rgb%s_kernel' % ps if h16 else
                                'conv_wino36_rgb_ns_kernel' if ns else 'conv_wino36_rgb_kernel')
                    elif h16:
                        name = 'conv_wino36h%s_kernel' % ps
                    elif i > 512:
                        name = 'conv_wino36_kernel<2, 2>'
                    else:
                        name = 'conv_wino36b_ns_kernel' if ns else 'conv_wino36b_kernel<2, 2>'
                elif wino is not None:
                    # last template argument: 16 / 8 / 4 = the shapes for maps that narrow (rw_wino.hip)
                    name = 'conv_wino16_kernel<%s, %s%s>' % ('2, 2, 8' if out_ch % 64 == 0 else '1, 4, 4', wino,
                                                             ', %d' % w if w <= 16 else '')
                else:
                    name = ('conv_halo_bf16x6_kernel<2, 2, %s>' % ('2, 2' if out_ch % 128 == 0 else '1, 4') if split
                            else conv_kernel_name(out_ch, i, w, upsample))
                # algorithmic HBM bytes of the call: the input map once, what it writes once (the (2H+1)^2 map of a
                # transposed convolution, the (2H)^2 result of the one-pass layer, the RGB image of the fused last layer)
                if wino in ('up4', 'd16up'):
                    out_elems = b * out_ch * 4 * h * w
                elif wino in ('f4rgb', 'd16rgb') or fn is self._orig[3] or fn is self._orig[5]:
                    out_elems = b * 3 * h * w * 2                  # running image read and written; no feature map
                elif upsample:
                    out_elems = b * out_ch * (2 * h + 1) * (2 * w + 1)
                elif wino == 'd16p':
                    out_elems = b * out_ch * h * w + (out_ch // 32) * b * 3 * h * w      # + the partial images
                else:
                    out_elems = b * out_ch * h * w
                timer.calls.append((name, s, e, 2.0 * 9 * i * out_ch * h * w * b, 4.0 * (b * i * h * w + out_elems)))
                return y
            return inner
        hip.conv3x3 = wrap(self._orig[0], False)
        hip.conv_transpose3x3s2 = wrap(self._orig[1], True)
        hip.conv3x3_bf16x6 = wrap(self._orig[2], False, split=True)
        hip.conv3x3_to_rgb = wrap(self._orig[3], False)         # same kernel, ToRGB in the epilogue
        hip.conv3x3_wino = wrap(self._orig[4], False, wino='false')
        hip.conv3x3_wino_to_rgb = wrap(self._orig[5], False, wino='true')
        hip.conv3x3_wino4 = wrap(self._orig[6], False, wino='f4')
        hip.conv_transpose3x3s2_wino = wrap(self._orig[7], True, wino='up')
        hip.conv_transpose3x3s2_blur_wino4 = wrap(self._orig[8], True, wino='up4')
        hip.conv3x3_wino4_to_rgb = wrap(self._orig[9], False, wino='f4rgb')
        hip.conv3x3_direct16 = wrap(self._orig[10], False, wino='d16')
        hip.conv_transpose3x3s2_blur_direct16 = wrap(self._orig[11], True, wino='d16up')
        hip.conv3x3_direct16_to_rgb = wrap(self._orig[12], False, wino='d16rgb')
        hip.conv3x3_direct16_rgb_partial = wrap(self._orig_rgbp, False, wino='d16p')

    def remove(self):
        from rewriting_amd import hip
        (hip.conv3x3, hip.conv_transpose3x3s2, hip.conv3x3_bf16x6, hip.conv3x3_to_rgb, hip.conv3x3_wino,
         hip.conv3x3_wino_to_rgb, hip.conv3x3_wino4, hip.conv_transpose3x3s2_wino,
         hip.conv_transpose3x3s2_blur_wino4, hip.conv3x3_wino4_to_rgb, hip.conv3x3_direct16,
         hip.conv_transpose3x3s2_blur_direct16, hip.conv3x3_direct16_to_rgb) = self._orig
        hip.conv_transpose3x3s2_blur_fused = self._orig_fused
        hip.conv3x3_direct16_rgb_partial = self._orig_rgbp

    def result(self):
        per = {}
        for name, s, e, fl, by in self.calls:
            d = per.setdefault(name, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            d['ms'] += s.elapsed_time(e)
            d['flops'] += fl
            d['bytes'] += by
            d['launches'] += 1
        if not per:
            return None

        def rec(n, v):
            factor, algorithm, pipe, peak = issued_fraction(n)
            sec = v['ms'] * 1e-3
            issued = v['flops'] * factor / sec / 1e12
            return dict(effective_tflops=round(v['flops'] / sec / 1e12, 1), issued_tflops=round(issued, 1), pipe=pipe,
                        issued_frac=round(issued / peak, 3), hbm_gbs=round(v['bytes'] / sec / 1e9, 1),
                        hbm_frac=round(v['bytes'] / sec / 1e9 / HBM_PEAK_GBS, 3), ms=round(v['ms'], 2),
                        launches=v['launches'])
        dom = max(per, key=lambda n: per[n]['ms'])
        d = per[dom]
        factor, algorithm, pipe, peak = issued_fraction(dom)
        effective = d['flops'] / (d['ms'] * 1e-3) / 1e12          # direct-sum FLOPs per second
        issued = effective * factor                                  # what the matrix pipe executes
        tot_ms = sum(v['ms'] for v in per.values())
        tot_fl = sum(v['flops'] for v in per.values())
        # time the matrix pipes would need at their peaks for what all conv launches issue (fp32 and f16 pipes mixed)
        pipe_s = sum(v['flops'] * issued_fraction(n)[0] / (issued_fraction(n)[3] * 1e12) for n, v in per.items())
        out = dict(bound='mfma', achieved=round(issued, 2), peak=peak,
                   unit='TFLOP/s', frac=round(issued / peak, 4), traffic=None,
                   kernel=dom, launches=d['launches'], pipe=pipe,
                   avg_launch_us=round(d['ms'] * 1e3 / d['launches'], 2),
                   algorithm=algorithm, mfma_flops_issued_over_direct_sum=round(factor, 4),
                   flops_per_launch=round(d['flops'] / d['launches'] * factor),
                   direct_sum_flops_per_launch=round(d['flops'] / d['launches']),
                   effective_tflops=round(effective, 2),
                   effective_over_fp32_direct_roof=round(effective / FP32_MFMA_PEAK_TFLOPS, 4),
                   hbm=dict(algorithmic_bytes_per_launch=round(d['bytes'] / d['launches']),
                            achieved_gbs=round(d['bytes'] / (d['ms'] * 1e-3) / 1e9, 1), peak_gbs=HBM_PEAK_GBS,
                            frac=round(d['bytes'] / (d['ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)),
                   note='`achieved` = matrix FLOPs the kernel ISSUES per second on ITS pipe (direct-sum FLOPs of SURVEY 8d x '
                        'the algorithm\'s multiply count / the direct sum\'s, x 4 piece products where the operands are '
                        'split into f16 pairs) against that pipe\'s dense peak; `effective_tflops` = the direct sum\'s '
                        'FLOPs per second; `hbm` = the launch\'s algorithmic bytes against the HBM roof.  Neither roof binds '
                        'these kernels: see DESIGN.md section 4 (ablations, SQ counters)',
                   all_conv_kernels=dict(matrix_pipe_time_frac=round(pipe_s / (tot_ms * 1e-3), 4),
                                         effective_tflops=round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                                         effective_over_fp32_direct_roof=round(
                                             tot_fl / (tot_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                                         ms_per_step=None))
        # the roof the dominant kernel is NEARER to is the one `bound` / `achieved` / `peak` / `frac` name (the fused upsampling
        # kernels of the layers with few channels move more bytes per FLOP than they issue MFMAs); the other stays beside it
        if out['hbm']['frac'] > out['frac']:
            out['mfma'] = dict(achieved_tflops=out['achieved'], peak_tflops=out['peak'], frac=out['frac'])
            out.update(bound='hbm', achieved=out['hbm']['achieved_gbs'], peak=HBM_PEAK_GBS, unit='GB/s', frac=out['hbm']['frac'])
            out['note'] = ('`achieved` = the launch\'s ALGORITHMIC bytes (input map once + result once, SURVEY 8d) per second of its '
                           'average duration against the HBM roof -- the nearer of this kernel\'s two roofs; `mfma` = the matrix '
                           'FLOPs it ISSUES per second (direct-sum FLOPs x the algorithm\'s multiply count / the direct sum\'s, x '
                           'the f16 piece products per multiply) against the 16-bit pipe\'s dense peak; `effective_tflops` = the '
                           'direct sum\'s FLOPs per second.  Neither roof binds it: DESIGN.md section 4.5 (cycle counters, '
                           'ablations)')
        out['per_kernel'] = {n: rec(n, v) for n, v in sorted(per.items(), key=lambda kv: -kv[1]['ms'])}
        out['_tot_ms'] = tot_ms
        out['_pipe_s'] = pipe_s
        return out


def issued_fraction(kernel):
    """(matrix FLOPs a kernel issues / direct-sum FLOPs of the convolution it computes, what the algorithm is, the
    pipe it issues them on, that pipe's dense peak in TFLOP/s).  The minimal-filtering kernels multiply less than the
    direct sum; the split-operand kernels (…h…) issue FOUR f16 piece products per multiply on the 16-bit pipe
    (rw_wino.hip, rw_wino4.hip, rw_upwino.hip headers)."""
    f32 = ('fp32 MFMA', FP32_MFMA_PEAK_TFLOPS)
    f16 = ('f16 MFMA, 4 piece products per multiply (exact operand split), fp32 accumulate', F16_MFMA_PEAK_TFLOPS)
    # rw_dconv.hip, round 5 (DC_PRODUCTS == 3): 5 MFMAs per kernel column and output block where 6 carried four piece
    # products -- the matrix FLOPs ISSUED are 4 x 5/6 = 3.33 per multiply
    f16d = ('f16 MFMA, 3 piece products for the taps ky = 0 / 1 and 4 for ky = 2 (exact operand split, 10 MFMA-halves per 3 '
            'taps), fp32 accumulate', F16_MFMA_PEAK_TFLOPS)
    if kernel.startswith('dconv') and '_up_' in kernel:
        return (16.0 * 5 / 6, 'transposed conv (*) blur as four DIRECT 3x3 phase convolutions: 4x the transposed conv\'s '
                'direct-sum multiplies, 3.33 f16 piece products issued per multiply') + f16d
    if kernel.startswith('dconv'):
        return (4.0 * 5 / 6, 'direct sum, 3.33 f16 piece products issued per multiply') + f16d
    if kernel.startswith('tconv_blur'):
        halo = 18.0 * 34 / (16 * 32) if 't16' in kernel else 10.0 * 34 / (8 * 32)     # positions computed / positions kept (t16 | 8-row tiles)
        return (14.0 / 18.0 * 4 * halo, 'transposed conv as a direct sum at its own multiply count (14 MFMAs per block and '
                'chunk where four piece products take 18; x %.2f halo positions), its (2H+1)^2 map in LDS, blur from there'
                % halo) + f16d
    if kernel.startswith('conv_up_wino36h'):
        return (4.0, 'transposed conv (*) blur as four F(4x4,3x3) phase convolutions: the transposed conv\'s direct-sum '
                'multiply count, each as 4 f16 piece products') + f16
    if kernel.startswith('conv_up_winoh'):
        return (4 * 25.0 / 36.0, 'transposed conv by F(2,2) on the four output-parity phases: 25 multiplies per 2x2 block '
                'of quads where the direct sum has 36, each as 4 f16 piece products') + f16
    if kernel.startswith('conv_wino36h'):
        return (1.0, 'winograd F(4x4,3x3): 36 multiplies per 4x4 output tile where the direct sum has 144, each as 4 f16 '
                'piece products') + f16
    if kernel.startswith('conv_up_wino36'):
        return (1.0, 'transposed conv (*) blur as four F(4x4,3x3) phase convolutions, fp32: issues the transposed '
                'conv\'s direct-sum FLOP count') + f32
    if kernel.startswith('conv_up_wino'):
        return (25.0 / 36.0, 'transposed conv by F(2,2) on the four output-parity phases, fp32: 25 multiplies per 2x2 '
                'block of quads where the direct sum has 36') + f32
    if kernel.startswith('conv_wino36'):
        return (0.25, 'winograd F(4x4,3x3), fp32: 36 multiplies per 4x4 output tile where the direct sum has 144') + f32
    if kernel.startswith('conv_wino16'):
        return (1.0 / 2.25, 'winograd F(2x2,3x3), fp32: 16 multiplies per 2x2 output tile where the direct sum has 36') + f32
    return (1.0, 'direct implicit GEMM, fp32 MFMA') + f32


def attach_pmc_traffic(roof, workload, batch):
    """`traffic` (HBM bytes per launch of the roofline kernel) comes from separate rocprofv3 --pmc
    passes (FETCH_SIZE, WRITE_SIZE; corrected as MI355X_MICROARCH.md prescribes), whose summary is
    committed under profiles/.  Filled in only if that summary matches this kernel and workload."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if not os.path.isfile(path):
        return
    with open(path) as f:
        rec = json.load(f)
    for r in rec.get('entries', []):
        if r.get('kernel') == roof['kernel'] and r.get('workload') == workload and r.get('batch') == batch:
            roof['traffic'] = r['hbm_bytes_per_launch']
            roof['traffic_source'] = r.get('source')
            roof['algorithmic_bytes_per_launch'] = r.get('algorithmic_bytes_per_launch')


def parity_of_timed_output(img, size, z, world):
    """Compares the image batch the LAST TIMED step produced with tests/golden/gen_s<size>_full.npz -- the digest of
    the reference's own CPU forward (oracle/make_golden.py executing /root/reference/utils/stylegan2/models.py:41-141)
    of the first rows of the same z stream.  Row j of a batch gets row j of RandomState(0).randn(batch, H*W) as noise
    whatever the batch size (quirk Q1) and standard_z_sample fills row-major, so rows 0..b-1 of a 64-seed batch are
    exactly the fixture's b seeds.  Fields: linf over the strided sub-sample and the four full-resolution crops,
    row/column-sum deviation (every pixel of the rows enters one of each), the rows compared."""
    import numpy
    path = os.path.join(ROOT, 'tests', 'golden', 'gen_s%d_full.npz' % size)
    if not os.path.isfile(path):
        return None
    g = numpy.load(path)
    zg = torch.from_numpy(g['z'])
    rows = [j for j in range(min(zg.shape[0], z.shape[0])) if torch.equal(z[j].cpu(), zg[j])]
    if not rows:
        return dict(fixture=os.path.relpath(path, ROOT), rows=[], note='no row of this rank\'s batch is a fixture seed')
    stride = int(g['image/stride'])
    got = img[rows].detach()
    want = torch.from_numpy(g['image/strided'])[rows]
    linf = (got[:, :, ::stride, ::stride].cpu() - want).abs().max().item()
    crops = g['image/crops']
    c = crops.shape[-1]
    for k, (y, x) in enumerate(g['image/crop_origin']):
        linf = max(linf, (got[:, :, y:y + c, x:x + c].cpu() - torch.from_numpy(crops[k])[rows]).abs().max().item())
    rs = (got.double().sum(3).cpu().numpy() - g['image/rowsum'][rows])
    cs = (got.double().sum(2).cpu().numpy() - g['image/colsum'][rows])
    return dict(fixture=os.path.relpath(path, ROOT), rows=rows, linf=float('%.3e' % linf),
                max_abs_image=float('%.3f' % want.abs().max().item()),
                rowsum_dev=float('%.3e' % max(abs(rs).max(), abs(cs).max())), bar_linf=1e-3,
                ok=bool(linf < 1e-3 and torch.isfinite(got).all().item()),
                what='output of the last timed step, batch rows %s, vs the reference CPU forward of the same seeds '
                     '(reference-generated fixture; north_star bar 1e-3 L-inf)' % rows)


def attach_pmc_step(step, workload, batch):
    """HBM bytes of ONE whole step, summed over every kernel of the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (profiles/pmc_traffic.json 'steps'), if a committed summary matches this workload."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if not os.path.isfile(path):
        return
    with open(path) as f:
        rec = json.load(f)
    for r in rec.get('steps', []):
        if r.get('workload') == workload and r.get('batch') == batch:
            step['hbm_bytes_pmc'] = r['hbm_bytes_per_step']
            step['hbm_pmc_over_algorithmic'] = round(r['hbm_bytes_per_step'] / step['hbm_bytes_algorithmic'], 3)
            step['hbm_pmc_source'] = r.get('source')


def build_generator(size, device):
    from rewriting_amd import synthetic
    from rewriting_amd.utils.stylegan2 import models
    g = models.SeqStyleGAN2(size, 512, 8, truncation=0.5, mconv='seq')
    synthetic.randomize_(g, seed=0)
    return g.eval().to(device)


def _cpu_forward_fn(size):
    """(callable(z) -> images, kind): the reference's own SeqStyleGAN2 through oracle/reference_shim.py when
    /root/reference exists (build container), else the oracle restatement (a port; the GPU box)."""
    from rewriting_amd import synthetic
    from oracle import reference_shim
    if reference_shim.available():
        ref = reference_shim.load()
        g = ref.models.SeqStyleGAN2(size, 512, 8, truncation=0.5, mconv='seq')
        synthetic.randomize_(g, seed=0)
        g.eval()
        return (lambda z: g(z)), 'reference'
    from rewriting_amd.utils.stylegan2 import models
    from oracle import restatement as R
    g = models.SeqStyleGAN2(size, 512, 8, truncation=0.5, mconv='seq')
    synthetic.randomize_(g, seed=0)
    sd = {k: v.detach() for k, v in g.state_dict().items()}
    return (lambda z: R.generator_forward(sd, z, size, truncation=0.5)), 'port'


def cpu_baseline_forward(size, images=8):
    """The same forward on the host cores, at the best of a small sweep: one image at each thread count (torch's
    CPU kernels stop scaling well before 128 threads and lose to oversubscription beyond), then batch 4 at the
    best count (larger batches are SLOWER per image on the CPU: the working set leaves the caches), then
    `images` images in the winning configuration.  ~25 s of CPU work."""
    from rewriting_amd.utils import zdataset
    fwd, kind = _cpu_forward_fn(size)
    z = zdataset.standard_z_sample(max(images, 4), 512, seed=1)
    ncpu = os.cpu_count() or 1
    counts = sorted({min(c, ncpu) for c in (8, 16, 32, 64, 128)})
    saved = torch.get_num_threads()
    probe = {}
    try:
        with torch.no_grad():
            torch.set_num_threads(counts[0])
            fwd(z[:1])                                          # warm-up (allocator, thread pool)
            for c in counts:
                torch.set_num_threads(c)
                t0 = time.perf_counter()
                fwd(z[:1])
                probe['%dt_b1' % c] = 1 / (time.perf_counter() - t0)
            best = max(counts, key=lambda c: probe['%dt_b1' % c])
            torch.set_num_threads(best)
            t0 = time.perf_counter()
            fwd(z[:4])
            probe['%dt_b4' % best] = 4 / (time.perf_counter() - t0)
            batch = 4 if probe['%dt_b4' % best] > probe['%dt_b1' % best] else 1
            t0 = time.perf_counter()
            n = 0
            while n < images:
                fwd(z[n:n + batch])
                n += batch
            dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(saved)
    return dict(value=round(n / dt, 4), unit='images/sec', cores=best, threads=best, host_cores=ncpu, kind=kind,
                sample_short='%d images, batches of %d, %.1f s at %d of %d threads' % (n, batch, dt, best, ncpu),
                sample='%d images of the stylegan2-%d forward in batches of %d through %s (torch %s CPU kernels), '
                       '%.1f s at %d threads; probe img/s by (threads, batch): %s; host has %d logical cpus'
                       % (n, size, batch, "the reference's own utils/stylegan2/models.py (oracle/reference_shim.py)"
                          if kind == 'reference' else 'oracle/restatement.py', torch.__version__.split('+')[0], dt,
                          best, json.dumps({k: round(v, 3) for k, v in probe.items()}), ncpu))


def cpu_baseline_edit(shape, seeds=30, steps=150):
    """The other half of BASELINE.json's metric on the host cores: key collection (context forward of the 256^2
    generator up to layer 8 in batches of 10 + the second moment) and the 2001-step rank-1 solve, through the oracle
    restatement (a port: /root/reference does not exist on the GPU box) -- on a BOUNDED sample (`seeds` of the 1000
    seeds, `steps` of the 2001 iterations, at the solve's own shapes) extrapolated linearly, at the best of a few thread
    counts."""
    from rewriting_amd import synthetic
    from rewriting_amd.utils import zdataset
    from rewriting_amd.utils.stylegan2 import models
    from oracle import restatement as R
    g = models.SeqStyleGAN2(256, 512, 8, truncation=0.5, mconv='seq')
    synthetic.randomize_(g, seed=0)
    sd = {k: v.detach() for k, v in g.state_dict().items()}
    zds = zdataset.z_dataset_for_model(g, size=seeds)
    zb = [torch.stack([zds[j][0] for j in range(i, min(i + 10, seeds))]) for i in range(0, seeds, 10)]
    O, I, h, w = shape.get('out_ch', 512), shape.get('in_ch', 512), shape.get('h', 5), shape.get('w', 8)
    gen = torch.Generator().manual_seed(0)
    W0 = torch.randn(1, O, I, 3, 3, generator=gen)
    key, style = torch.randn(1, I, h, w, generator=gen), 1 + 0.3 * torch.randn(1, I, generator=gen)
    val, bias = torch.randn(1, O, h, w, generator=gen), torch.zeros(O)
    ctx = torch.nn.functional.normalize(torch.randn(1, I, generator=gen), dim=1)
    ncpu = os.cpu_count() or 1
    saved = torch.get_num_threads()
    best = None
    try:
        with torch.no_grad():
            for c in sorted({min(t, ncpu) for t in (8, 32, 64)}):
                torch.set_num_threads(c)
                R.context_forward(sd, zb[0][:2], 256, 8, truncation=0.5)        # warm-up
                t0 = time.perf_counter()
                R.second_moment(R.context_forward(sd, b, 256, 8, truncation=0.5)[0] for b in zb)
                t_stats = (time.perf_counter() - t0) * 1000.0 / seeds
                t0 = time.perf_counter()
                R.insert_explicit(W0, key, style, val, bias, 0.1, ctx, steps)
                t_solve = (time.perf_counter() - t0) * 2001.0 / steps
                if best is None or t_stats + t_solve < best[0] + best[1]:
                    best = (t_stats, t_solve, c)
    finally:
        torch.set_num_threads(saved)
    return dict(value=round(best[0] + best[1], 2), unit='s per edit', cores=best[2], threads=best[2], host_cores=ncpu,
                kind='port', key_collect_s=round(best[0], 2), solve_s=round(best[1], 2),
                sample='oracle/restatement.py (a port; the reference itself is not on this box): context_forward + '
                       'second_moment on %d of the 1000 seeds in batches of 10, insert_explicit for %d of the 2001 '
                       'iterations on a %d x %d x %d x %d problem (random values, the solve\'s shapes); both extrapolated '
                       'linearly; best of 8 / 32 / 64 threads (torch %s CPU kernels)' % (seeds, steps, O, I, h, w,
                                                                                        torch.__version__.split('+')[0]))


def bench_device(local):
    """This rank's MI355X.  (tests/bench_cpu_driver.py -- test infrastructure -- substitutes CPU tensors and the
    kernel stand-ins of tests/hip_emulation.py here to drive the launcher and the rank plumbing without a GPU.)"""
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the HIP kernels have no CPU path)')
    torch.cuda.set_device(local)
    return torch.device('cuda', local)


def device_sync():
    torch.cuda.synchronize()


def timed(fn, steps, warmup, world):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
    import torch.distributed as dist
    from rewriting_amd import parallel
    for _ in range(warmup):
        fn()
    if world > 1:
        dist.barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    device_sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=parallel.collective_device(), dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt


def run_forward(args, rank, world, device, size, batch, name, cpu=True):
    from rewriting_amd.utils import zdataset
    g = build_generator(size, device)
    # seed i -> rank i mod world; every rank holds its own `batch` seeds, resident in HBM
    zall = zdataset.standard_z_sample(batch * world, 512, seed=1)
    z = zall[rank::world].contiguous().to(device)

    last = [None]

    def step():
        with torch.no_grad():
            last[0] = g(z)
    timed(step, 1, args.warmup, world)                      # warm-up incl. weight repack caches
    dt = timed(step, args.steps, 0, world)                  # THE number: nothing installed around the kernels
    from rewriting_amd.utils.stylegan2 import models as sg_models
    split = sg_models.matrix_mode_of_image_path() == 'split'
    images = batch * world * args.steps
    out = dict(metric='images/sec StyleGANv2-%d fwd' % size, value=round(images / dt, 2), unit='images/sec',
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
               higher_is_better=True, scaling='weak', vs_baseline=None,
               dtype=('f32 (conv operands split into exact f16 pairs from 32x32 up, f32 accumulate)' if split and
                      args.precision == 'f32' else
                      'f32' if args.precision == 'f32' else 'f32 via bf16x6 split (stride-1 convs), f32 elsewhere'),
               data='synthetic',
               config=dict(workload=name, batch_per_gpu=batch, truncation=0.5, mconv='seq',
                           weights='synthetic seed 0', parallelism='seeds partitioned per rank, no collective',
                           conv_gflop_per_image=round(conv_flops(size) / 1e9, 2),
                           matrix_mode='split' if split else 'f32',
                           direct_sums=os.environ.get('RW_MM_DIRECT16', 'auto') if split else None))
    # second pass, NOT the headline: HIP events around every convolution call (on torch's current stream, the one the
    # kernels are launched on) -> the per-kernel table and the dominant kernel's roofline
    timer = ConvTimer()
    timer.install()
    inst_steps = max(1, min(3, args.steps))
    dt_inst = timed(step, inst_steps, 0, world)
    timer.remove()
    roof = timer.result()
    tot_ms = roof.pop('_tot_ms')
    pipe_s = roof.pop('_pipe_s')
    roof['all_conv_kernels']['ms_per_step'] = round(tot_ms / inst_steps, 3)
    roof['instrumented_pass'] = dict(steps=inst_steps, ms_per_step=round(dt_inst / inst_steps * 1e3, 3))
    attach_pmc_traffic(roof, 'ffhq%d' % size, batch)
    out['roofline'] = roof
    # the whole step against both roofs: time the matrix pipes would need at peak for what all conv launches issue /
    # wall time, and SURVEY 8d's algorithmic bytes (a perfectly block-fused forward) / (wall time x HBM peak); PMC
    # bytes beside them
    alg_bytes = {256: 276.3e6, 1024: 1217.7e6}.get(size, 0) * batch
    step = dict(matrix_pipe_time_frac=round(pipe_s / inst_steps / (dt / args.steps), 4),
                direct_sum_tflops=round(conv_flops(size) * batch * args.steps / dt / 1e12, 2),
                direct_sum_over_fp32_roof=round(conv_flops(size) * batch * args.steps / dt / 1e12
                                                / FP32_MFMA_PEAK_TFLOPS, 4),
                hbm_frac=round(alg_bytes * args.steps / dt / 1e9 / HBM_PEAK_GBS, 4),
                hbm_algorithmic_gbs=round(alg_bytes * args.steps / dt / 1e9, 1),
                hbm_bytes_algorithmic=round(alg_bytes), hbm_bytes_pmc=None,
                note='per GPU and step.  north_star\'s ">= 60 % of the HBM roofline": the step moves its algorithmic '
                     'bytes at hbm_frac of 8 TB/s; its convolution kernels are bound by neither roof but by their '
                     'load -> barrier -> transform -> multiply chains (DESIGN.md section 4)')
    attach_pmc_step(step, 'ffhq%d' % size, batch)
    out['step'] = step
    out['parity'] = parity_of_timed_output(last[0], size, z, world)
    del g, z
    last[0] = None
    if cpu and rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline_forward(size)
    return out


def context_flops(model_size, layer, channel_multiplier=2):
    """Conv FLOPs per seed of the context model of `layer` (styled convs of layers 2..layer-1)
    and (key channels, key height) of that layer."""
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
          256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
    total, cin, res, idx = 0, 512, 4, 2
    layers = [(2, 512, 512, 4, False)]
    while res < model_size:
        cout = ch[res * 2]
        layers.append((idx + 1, cin, cout, res, True))
        res *= 2
        layers.append((idx + 2, cout, cout, res, False))
        cin, idx = cout, idx + 2
    key = None
    for n, ci, co, r, up in layers:
        if n == layer:
            key = (ci, r)
            break
        total += 2 * 9 * ci * co * r * r
    return total, key


def measure_edit(device, reps, warmup, cpu=False):
    """configs[2]: horse->hat rank-1 edit at layer 8 of the 256 model.  Every repetition builds a fresh
    rewriter (1000-seed key statistics in batches of 10 + ZCA) and runs apply_edit (goal, context direction,
    2001-step solve)."""
    from rewriting_amd.rewrite import ganrewrite
    from rewriting_amd.utils import zdataset
    g = build_generator(256, device)
    zds = zdataset.z_dataset_for_model(g, size=1000)
    with open(os.path.join(ROOT, 'tests', 'golden', 'masks', 'recorded_horse_hat.json')) as f:
        req = json.load(f)
    times = dict(stats=[], edit=[], solve=[], total=[])
    for i in range(warmup + reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gw = ganrewrite.SeqStyleGanRewriter(g, zds, 8)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        box, insert = [], gw.insert

        def timed_insert(*a, **k):              # the reference's own timing hook (ganrewrite.py:261-263,295-298)
            box.append(insert(*a, return_timing=True, **k))
            return box[-1]
        gw.insert = timed_insert
        gw.apply_edit(req, rank=1, niter=2001, piter=10, lr=0.05)
        solve_ms = box[0] if box else None
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if i >= warmup:
            times['stats'].append(t1 - t0)
            times['edit'].append(t2 - t1)
            times['solve'].append((solve_ms or 0.0) / 1e3)
            times['total'].append(t2 - t0)
    med = lambda v: sorted(v)[len(v) // 2]
    solve_s = med(times['solve']) or med(times['edit'])
    from rewriting_amd.rewrite import hipsolve
    last = dict(hipsolve.LAST)
    if last.get('one_launch'):
        # rw_solve_run_f32: the weight stays in registers, the key crop in LDS; no HBM traffic to price.  Its work is
        # the two 3x3 correlations of an iteration (forward and weight gradient), 2 x 2 x O x I x 9 x h x w FLOPs, on
        # the packed fp32 VALU (v_pk_fma_f32: the same 157.3 TFLOP/s as the fp32 MFMA pipe); the kernel is bound by
        # instruction issue (reductions over channels, Adam's IEEE sqrt and divisions).
        flops = 4.0 * last['out_ch'] * last['in_ch'] * 9 * last['h'] * last['w'] * last['niter']
        roof = dict(bound='valu', achieved=round(flops / solve_s / 1e12, 2), peak=FP32_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                    frac=round(flops / solve_s / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                    note='one-launch solver, %d x %d channels on a %d x %d key crop: 4 x O x I x 9 x h x w FLOPs per '
                         'iteration (forward + weight gradient); %.1f us per iteration'
                         % (last['out_ch'], last['in_ch'], last['h'], last['w'], solve_s * 1e6 / last['niter']))
    else:
        solve_bytes = 7 * 512 * 512 * 9 * 4 * 2001
        roof = dict(bound='hbm', achieved=round(solve_bytes / solve_s / 1e9, 1), peak=HBM_PEAK_GBS,
                    unit='GB/s', frac=round(solve_bytes / solve_s / 1e9 / HBM_PEAK_GBS, 4),
                    note='7 x |W| x 4 B per step algorithmic (SURVEY.md 8d); latency-bound')
    # the same apply_edit as the reference's drivers call it: with an update_callback on every iteration
    # (rewrite/rewriteapp.py:517-521 prints loss.item() every 50th; metrics/make_watermark_images.py:66-72 ticks a bar) --
    # unmarked = the reference's contract (the callback runs between the steps and may render the stepped weight: one
    # launch per iteration), and marked ganrewrite.loss_only (it declares it reads (it, loss) only: the solve is not
    # interrupted)
    def ui_callback(it, loss):
        if it % 50 == 0 or it == 2000:
            loss.item()
    with_cb = {}
    for label, cb in (('reference_contract', ui_callback), ('loss_only', ganrewrite.loss_only(lambda it, loss: ui_callback(it, loss)))):
        gw = ganrewrite.SeqStyleGanRewriter(g, zds, 8)
        ts = []
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gw.apply_edit(req, rank=1, niter=2001, piter=10, lr=0.05, update_callback=cb)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        with_cb[label] = round(min(ts), 4)
    cpu_rec = cpu_baseline_edit(dict(last)) if cpu else None
    return dict(seconds_per_edit=round(med(times['total']), 4), seconds_per_edit_min=round(min(times['total']), 4),
                key_collect_s=round(med(times['stats']), 4),
                apply_edit_s=round(med(times['edit']), 4), solve_s=round(solve_s, 4), reps=reps,
                apply_edit_with_update_callback_s=with_cb, cpu_baseline=cpu_rec,
                workload='stylegan2-256 layer 8, recorded_horse_hat.json: 1000-seed key statistics + ZCA, goal, '
                         'context direction, 2001-step rank-1 solve',
                solve_path='rw_solve_run_f32' if last.get('one_launch') else 'rw_solve_step_f32',
                solve_roofline=roof)


def run_edit(args, rank, world, device):
    e = measure_edit(device, args.steps, args.warmup)
    return dict(metric='wall-clock per rank-1 edit (1000-seed key collect + 2001-step solve)',
                value=e['seconds_per_edit'], unit='s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                ms_per_step=round(e['seconds_per_edit'] * 1e3, 2), higher_is_better=False, scaling='weak',
                vs_baseline=None, dtype='f32', data='synthetic',
                config=dict(workload=e['workload'], key_collect_s=e['key_collect_s'], edit_s=e['apply_edit_s'],
                            solve_s=e['solve_s'], solve_path=e['solve_path'], replicas=world),
                roofline=dict(e['solve_roofline'], traffic=None))


def measure_sweep(device, size, layer, nseeds, steps, warmup, world, g=None):
    """configs[3]: the key second-moment sweep as the rewriters run it -- launches of k x 10 seeds, every seed
    with the noise row of its reference batch of 10 (noise_batch_period), launches dealt round-robin to the
    ranks, ONE all-reduce of (mom2, count)."""
    from rewriting_amd import parallel
    from rewriting_amd.utils import tally, zdataset, nethook
    from rewriting_amd.utils.stylegan2.models import noise_batch_period
    g = g if g is not None else build_generator(size, device)
    ctx = nethook.subsequence(g, upto_layer='layer%d.sconv.mconv.dconv' % layer, share_weights=True)
    zds = zdataset.z_dataset_for_model(g, size=nseeds)
    flops_ctx, (cin, res) = context_flops(size, layer)
    # launch size: as many reference batches as fit in ~2 GB of key map, at most 510 seeds, >= one launch per rank
    per_seed_bytes = cin * res * res * 4
    cap = int(os.environ.get('RW_SWEEP_LAUNCH', '510'))      # the rewriters' sweep_batch (whole rounds of 512 workgroups)
    launch = parallel.balanced_batch(nseeds, min(cap, max(10, (2 << 30) // per_seed_bytes // 10 * 10)), world)

    def step():
        with torch.no_grad(), noise_batch_period(10):
            return tally.tally_second_moment(lambda zb: ctx(zb.to(device)).fmap, zds, shard=parallel.shard(),
                                             nchw=True, batch_size=launch)
    dt = timed(step, steps, warmup, world)
    flops = (flops_ctx + 2.0 * res * res * cin * cin) * nseeds * steps
    effective = flops / dt / 1e12 / world
    # the path's one collective, timed on its own: the all-reduce of (mom2, count) in float64 as tally_second_moment
    # issues it (world == 1: nothing to reduce)
    launches_total = len(range(0, nseeds, launch))
    my_launches = len(parallel.batches_for_rank(launches_total, *parallel.shard())) if parallel.shard() else launches_total
    allreduce_ms = None
    if world > 1:
        import torch.distributed as dist
        buf = torch.zeros(cin * cin + 1, dtype=torch.float64, device=parallel.collective_device())
        dist.all_reduce(buf)
        device_sync()
        t0 = time.perf_counter()
        for _ in range(5):
            dist.all_reduce(buf)
        device_sync()
        allreduce_ms = round((time.perf_counter() - t0) / 5 * 1e3, 4)
    layers = None
    issued_frac = None
    if world == 1 and torch.device(device).type == 'cuda':
        # one more context forward of one launch with HIP events around every convolution: the roofline of each layer's
        # kernel (matrix FLOPs issued / its pipe's peak), as the headline line gives it for the generator's
        timer = ConvTimer()
        timer.install()
        try:
            with torch.no_grad(), noise_batch_period(10):
                zb = torch.stack([zds[j][0] for j in range(launch)]).to(device)
                ctx(zb)
                torch.cuda.synchronize()
        finally:
            timer.remove()
        r = timer.result()
        if r is not None:
            issued_frac = r['all_conv_kernels']['matrix_pipe_time_frac']
            layers = dict(seeds=launch, conv_ms=round(r['_tot_ms'], 3),
                          conv_issued_frac=issued_frac, per_kernel=r['per_kernel'],
                          note='HIP events around each convolution of ONE %d-seed context forward (border strips '
                               'and streaming kernels not included); kernel -> layer: conv_wino16<.., 4 / 8 / 16> = '
                               'layers 2 / 4 / 6, conv_up_wino_4x4 / _8x8 / _narrow = layers 3 / 5 / 7, the un-suffixed '
                               'kernels = the layers from 32 x 32 up' % launch)
    return dict(seeds_per_s=round(nseeds * steps / dt, 1), ms_per_sweep=round(dt / steps * 1e3, 2), size=size,
                layer=layer, seeds=nseeds, launch=launch, launches=launches_total, launches_this_rank=my_launches,
                allreduce_ms=allreduce_ms, key_map='%d x %d x %d' % (cin, res, res),
                gflop_per_seed=round((flops_ctx + 2.0 * res * res * cin * cin) / 1e9, 3), context_layers=layers,
                roofline=dict(bound='mfma', achieved=None if issued_frac is None else round(issued_frac * FP32_MFMA_PEAK_TFLOPS, 2),
                              peak=FP32_MFMA_PEAK_TFLOPS, unit='TFLOP/s', frac=issued_frac, traffic=None,
                              effective_tflops=round(effective, 2),
                              effective_over_fp32_direct_roof=round(effective / FP32_MFMA_PEAK_TFLOPS, 4),
                              note='per GPU.  `frac` = the time the matrix pipes would need AT THEIR DENSE PEAKS (fp32 MFMAs for '
                                   'the layers below 32^2, the 16-bit pipe for the split-operand kernels from 32^2 up) for '
                                   'what the context forward\'s convolution kernels issue / those kernels\' own time (one '
                                   'launch under HIP events; null when not measured: N > 1); `achieved` = that fraction of '
                                   'the fp32 MFMA peak.  `effective_*` = direct-sum conv FLOPs of '
                                   'layers 2..%d + 2 H W C^2 of a^T a per seed (SURVEY.md 8d) / wall time -- the minimal-'
                                   'filtering kernels issue fewer; the key map itself (%.1f MB per seed) is read once'
                                   % (layer - 1, per_seed_bytes / 1e6)))


def run_sweep(args, rank, world, device):
    m = measure_sweep(device, args.size, args.layer, args.seeds, args.steps, args.warmup, world)
    return dict(metric='key-statistics sweep seeds/sec (layer %d of stylegan2-%d)' % (args.layer, args.size),
                value=m['seeds_per_s'], unit='seeds/sec', n_gpus=world, steps=args.steps,
                warmup=args.warmup, ms_per_step=m['ms_per_sweep'], higher_is_better=True,
                scaling='strong', vs_baseline=None,
                dtype='f32 (conv operands split into exact f16 pairs from 32x32 up, f32 accumulate)',
                data='synthetic',
                config=dict(workload='%d-seed second-moment sweep, launches of %d seeds (reference batches of 10 '
                                     'inside) dealt round-robin, one all-reduce' % (args.seeds, m['launch']),
                            key_map=m['key_map'], gflop_per_seed=m['gflop_per_seed'],
                            launches=m['launches'], launches_this_rank=m['launches_this_rank'],
                            allreduce_ms=m['allreduce_ms'], context_layers=m['context_layers']),
                roofline=m['roofline'])


def run_watermark(args, rank, world, device):
    from rewriting_amd import workloads
    return workloads.watermark_bench(args, rank, world, device, timed)


def extras(args, rank, world, device):
    """The rest of BASELINE.json's metric and configs, timed in this process after the headline workload (whose
    buffers are released first).  Sweeps: every rank takes part (they contain the path's one collective); the
    watermark job deals its five variants to the ranks as replicas; the edit and the 256^2 forward are per-GPU
    quantities and are measured on one-GPU runs only."""
    out = {}
    torch.cuda.empty_cache()
    g = build_generator(1024, device)
    for layer in (8, 10, 14):      # configs[3]: the three sweep layers of SURVEY 8d, on its own 10 000 seeds at every N
        out['sweep_ffhq1024_layer%d' % layer] = measure_sweep(device, 1024, layer, 10000, 1 if layer == 14 else 2, 1, world,
                                                              g=g)
        torch.cuda.empty_cache()
    if world == 1:
        # the headline's forward on the fp32 matrix pipe (RW_MM=f32), same process, same box: what the operand split buys
        from rewriting_amd.utils import zdataset
        z = zdataset.standard_z_sample(64, 512, seed=1).to(device)

        def fwd():
            with torch.no_grad():
                g(z)
        rates = {}
        modes = {'f32': ('f32', '0'), 'split, F(4x4,3x3) on layers 10-18': ('split', '0'),
                 'split, direct sums on layers 10-17 (the default)': ('split', 'auto'),
                 'split, direct sums on layers 10-18': ('split', '1')}
        saved = {k: os.environ.get(k) for k in ('RW_MM', 'RW_MM_DIRECT16')}
        for mm, (pipe, d16) in modes.items():
            os.environ['RW_MM'], os.environ['RW_MM_DIRECT16'] = pipe, d16
            timed(fwd, 1, 1, world)
            rates[mm] = round(64 * 5 / timed(fwd, 5, 0, world), 2)
        for key, val in saved.items():
            if val is None:
                del os.environ[key]
            else:
                os.environ[key] = val
        out['forward_ffhq1024_by_matrix_mode'] = dict(
            images_per_s=rates, batch=64, steps=5,
            note='f32 = every product on fp32 MFMAs (round 3\'s kernels); split = f16 operand pairs on the 16-bit pipe (four piece '
                 'products per multiply, fp32 accumulate) in the F(2,2) kernels of the transposed convolutions and, by row: the '
                 'F(4x4,3x3) kernels on every stride-1 layer from 64^2 up and the one-pass upsampling layer (RW_MM_DIRECT16=0) / '
                 'DIRECT sums (csrc/rw_dconv.hip) on layers 10-17 with F(4x4,3x3) + ToRGB on the last one (auto: the default) / '
                 'direct sums on the last layer too (1).  No forward drains a stream: round 4\'s RW_FORWARD_DRAIN workaround is '
                 'gone with the device scalars it covered')
        del z
    del g
    if world == 1:
        out['edit_horse256_layer8'] = measure_edit(device, 7, 1, cpu=not args.no_cpu_baseline)
        torch.cuda.empty_cache()
        saved = (args.steps, args.warmup)
        args.steps, args.warmup = 5, 2
        f = run_forward(args, rank, world, device, 256, 64, 'stylegan2-256 generator forward, 64-seed batch', cpu=False)
        args.steps, args.warmup = saved
        out['forward_ffhq256_b64'] = dict(images_per_s=f['value'], ms_per_step=f['ms_per_step'],
                                          all_conv_kernels=f['roofline']['all_conv_kernels'],
                                          step=f['step'], parity=f['parity'])
        torch.cuda.empty_cache()
    if world == 1:
        out['rccl_one_rank'] = rccl_selfcheck()
    # configs[4]: the five watermark.sh variants (statistics, erase solves, sample sets), one warm-up job
    saved = (args.steps, args.warmup, args.seeds)
    args.steps, args.warmup, args.seeds = 1, 1, 1000
    w = run_watermark(args, rank, world, device)
    args.steps, args.warmup, args.seeds = saved
    out['watermark_church256'] = dict(seconds_per_job=w['value'], images_per_s=w['config']['images_per_s'],
                                      variants=w['config']['variants'], scaling=w['scaling'],
                                      workload=w['config']['workload'],
                                      update_callback='the driver\'s progress-bar hook on every iteration, marked '
                                                      'ganrewrite.loss_only (it reads `it` only); the variants share '
                                                      'one statistics cache, as watermark.sh\'s runs share their '
                                                      'results directory')
    if world == 1:
        # the erase solves of the first variant (2 x 2001 steps) under the three callback contracts
        from rewriting_amd import workloads
        req = workloads.fold_request(workloads.load_request(), 1000)
        modes = {}
        for mode in ('none', 'loss_only', 'reference'):
            t, _, _ = workloads.run_watermark_variant(workloads.WATERMARK_VARIANTS[0], device, req, sample_size=1000,
                                                      callback=mode)
            modes[mode] = round(t['edit_s'], 4)
        out['watermark_church256']['edit_s_of_variant_0_by_update_callback'] = dict(
            modes, note='none = no callback; loss_only = the marked hook (callbacks delivered after the solve); '
                        'reference = the unmarked hook: called between the steps as rewrite/ganrewrite.py:288-289 does, '
                        'one launch per iteration')
    return out


def _num(v, nd=4):
    """Numbers only (rounded); anything else becomes None."""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, (int, float)):
        return v if isinstance(v, int) else (round(v, nd) if math.isfinite(v) else None)
    return None


def compact_line(out):
    """The ONE line the driver parses: the contract keys only, numbers and short names, < 4 KB.  Everything else the
    run measured (per-kernel tables, notes, samples, sources) is in bench_detail.json (write_detail)."""
    short = lambda v, n=80: v if not isinstance(v, str) or len(v) <= n else v[:n - 1].rstrip() + '~'
    c = {k: out.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                 'scaling', 'vs_baseline', 'dtype', 'data')}
    c['dtype'] = short(c['dtype'])
    c['config'] = {k: short(v, 64) for k, v in (out.get('config') or {}).items()
                   if isinstance(v, (str, int, float, bool)) or v is None}
    r = out.get('roofline')
    c['roofline'] = None                 # always present; null = this workload is a job time, no single kernel priced
    if r:
        c['roofline'] = {k: (short(r.get(k), 48) if k in ('bound', 'kernel', 'unit') else _num(r.get(k)))
                         for k in ('bound', 'kernel', 'launches', 'avg_launch_us', 'achieved', 'peak', 'unit', 'frac',
                                   'traffic')}
        alg = r.get('algorithmic_bytes_per_launch') or (r.get('hbm') or {}).get('algorithmic_bytes_per_launch')
        if alg:
            c['roofline']['algorithmic_bytes_per_launch'] = alg
        if isinstance(r.get('mfma'), dict):
            c['roofline']['mfma_frac'] = _num(r['mfma'].get('frac'))
    b = out.get('cpu_baseline')
    if b:
        c['cpu_baseline'] = dict(value=_num(b.get('value')), unit=short(b.get('unit'), 24), cores=b.get('cores'),
                                 kind=b.get('kind'), sample=short(b.get('sample_short') or b.get('sample'), 96))
    st = out.get('step')
    if st:
        c['step'] = {k: _num(st.get(k)) for k in ('hbm_frac', 'hbm_bytes_algorithmic', 'hbm_bytes_pmc',
                                                   'matrix_pipe_time_frac')}
    pa = out.get('parity')
    if pa:
        c['parity'] = dict(linf=_num(pa.get('linf'), 9), ok=pa.get('ok'), bar=pa.get('bar_linf'))
    if out.get('rccl'):
        c['rccl'] = {k: out['rccl'].get(k) for k in ('backend', 'world_size')}
    ex = out.get('extra')
    if ex:
        e = {}
        for k, v in ex.items():
            if k.startswith('sweep_') and isinstance(v, dict):
                e[k + '_seeds_per_s'] = _num(v.get('seeds_per_s'))
                if v.get('allreduce_ms') is not None:
                    e[k + '_allreduce_ms'] = _num(v.get('allreduce_ms'))
                    e[k + '_launches_this_rank'] = v.get('launches_this_rank')
        ed = ex.get('edit_horse256_layer8')
        if ed:
            e['edit_s'] = _num(ed.get('seconds_per_edit'))
            e['edit_key_collect_s'] = _num(ed.get('key_collect_s'))
            e['edit_apply_s'] = _num(ed.get('apply_edit_s'))
            e['edit_solve_s'] = _num(ed.get('solve_s'))
            cb = ed.get('apply_edit_with_update_callback_s') or {}
            e['edit_apply_reference_callback_s'] = _num(cb.get('reference_contract'))
            if ed.get('cpu_baseline'):
                e['edit_cpu_s'] = _num(ed['cpu_baseline'].get('value'))
                e['edit_cpu_cores'] = ed['cpu_baseline'].get('cores')
        f = ex.get('forward_ffhq256_b64')
        if f:
            e['ffhq256_b64_images_per_s'] = _num(f.get('images_per_s'))
            e['ffhq256_parity_linf'] = _num((f.get('parity') or {}).get('linf'), 9)
            e['ffhq256_hbm_frac'] = _num((f.get('step') or {}).get('hbm_frac'))
        m = ex.get('forward_ffhq1024_by_matrix_mode')
        if m:
            e['ffhq1024_f32_pipe_images_per_s'] = _num((m.get('images_per_s') or {}).get('f32'))
        w = ex.get('watermark_church256')
        if w:
            e['watermark_job_s'] = _num(w.get('seconds_per_job'))
            e['watermark_images_per_s'] = _num(w.get('images_per_s'))
        rc = ex.get('rccl_one_rank')
        if rc:
            e['rccl_one_rank_ok'] = bool(rc.get('ok'))
            e['rccl_one_rank_allreduce_us'] = _num(rc.get('allreduce_us'))
        c['extra'] = e
    c['detail'] = out.get('detail')
    return c


def write_detail(out):
    """The long form (per-kernel tables, notes, samples, counter sources): bench_detail.json next to this file and,
    when the directory exists (a gpurun call), under gpurun_out/ so that it travels back; RW_BENCH_DETAIL=<path> names
    the one file to write instead.  Returns the path written first, relative to the repository."""
    text = json.dumps(out, indent=1)
    written = None
    where = ([os.environ['RW_BENCH_DETAIL']] if os.environ.get('RW_BENCH_DETAIL') else
             [os.path.join(d, 'bench_detail.json') for d in (ROOT, os.path.join(ROOT, 'gpurun_out'))])
    for path in where:
        if os.path.isdir(os.path.dirname(os.path.abspath(path))):
            try:
                with open(path, 'w') as f:
                    f.write(text)
                written = written or os.path.relpath(path, ROOT)
            except OSError:
                pass
    return written


def rccl_selfcheck():
    """scripts/rccl_selfcheck.py in a child process (its own process group, bounded by a timeout): RCCL initialises a
    one-rank communicator on this box's GPU and runs the sweep's all-reduce.  A failure is reported, never raised."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'rccl_selfcheck.py')], capture_output=True,
                           text=True, timeout=180)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith('{'):
                return json.loads(line)
        return dict(ok=False, error=('rc %d: ' % r.returncode) + (r.stderr or r.stdout)[-300:])
    except Exception as e:                                  # timeout, missing interpreter, unparsable output
        return dict(ok=False, error='%s: %s' % (type(e).__name__, e))


def self_launch(argv, gpus):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU
    (backend nccl = RCCL), rendezvous on 127.0.0.1; rank 0's JSON line is this process's output."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    # RW_BENCH_ENTRY: the script the ranks execute (default: this file).  tests/test_bench_launcher.py points it at
    # tests/bench_cpu_driver.py, which calls this file's main() on emulated kernels.
    entry = os.environ.get('RW_BENCH_ENTRY') or os.path.abspath(__file__)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), entry] + argv
    # torch.distributed.run exits non-zero as soon as ANY rank does (it then terminates the others): that code is
    # this process's exit code, so a failure on a rank other than 0 cannot look like success
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='ffhq1024', choices=['ffhq1024', 'ffhq256', 'edit', 'sweep', 'watermark'])
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--layer', type=int, default=8)
    ap.add_argument('--seeds', type=int, default=1000)
    ap.add_argument('--niters', type=int, default=2001, help='watermark workload: solver steps per erase (reference: 2001)')
    ap.add_argument('--wm-size', type=int, default=256, help='watermark workload: generator resolution (reference: 256)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='headline workload only')
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16x6'],
                    help='bf16x6: opt-in split-precision stride-1 convolutions (fp32-product accuracy); '
                         'the default and the headline number are exact fp32 MFMA')
    ap.add_argument('--conv-algo', default=None, choices=['direct', 'winograd', 'winograd4'],
                    help='stride-1 3x3 convolutions: winograd F(2x2,3x3) in fp32 (default where it applies) or '
                         'direct implicit GEMM')
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    if args.precision != 'f32':
        os.environ['RW_CONV_PRECISION'] = args.precision
    if args.conv_algo:
        os.environ['RW_CONV_ALGO'] = args.conv_algo
    from rewriting_amd import parallel
    rank, world, local = parallel.init_from_env()
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but the launcher started %d ranks' % (args.gpus, world))
    device = bench_device(local)
    if args.workload == 'ffhq1024':
        out = run_forward(args, rank, world, device, 1024, args.batch or 64,
                          'stylegan2-1024 generator forward (FFHQ-1024 architecture)')
        if not args.no_extra:
            out['extra'] = extras(args, rank, world, device)
    elif args.workload == 'ffhq256':
        out = run_forward(args, rank, world, device, 256, args.batch or 64,
                          'stylegan2-256 generator forward, 64-seed batch (FFHQ-256 architecture)')
    elif args.workload == 'edit':
        out = run_edit(args, rank, world, device)
    elif args.workload == 'sweep':
        out = run_sweep(args, rank, world, device)
    else:
        out = run_watermark(args, rank, world, device)
    if world > 1:
        import torch.distributed as dist
        if dist.get_world_size() != args.gpus:
            raise SystemExit('bench.py: process group of %d ranks for --gpus %d' % (dist.get_world_size(), args.gpus))
        out['rccl'] = dict(backend=dist.get_backend(), world_size=dist.get_world_size(),
                           note='seeds / launches / variants are partitioned per rank; the only data-path collective is '
                                'the sweeps\' all-reduce of (mom2, count) (extra.sweep_*: allreduce_ms, launches, '
                                'launches_this_rank = rank 0\'s share)')
    if rank == 0:
        # stdout: a pointer to the long form, then THE line (compact, contract keys, < 4 KB) as the last line
        out['detail'] = write_detail(out)
        print('# bench detail (per-kernel tables, notes, sources): %s' % out['detail'], flush=True)
        line = json.dumps(compact_line(out), allow_nan=False, separators=(',', ':'))
        assert len(line) < 4096, len(line)
        print(line, flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
