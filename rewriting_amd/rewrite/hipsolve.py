"""Driver of the fused HIP solver (``rw_solve_run_f32`` / ``rw_solve_step_f32``) behind
``SeqStyleGanRewriter.insert``.

Reference loop: rewrite/ganrewrite.py:271-294 -- 2001 iterations of {L1 loss through
dconv+demod, noise, bias+lrelu; backward to W; Adam; optional callback; projection when
``it % piter == 0 or it == niter-1``}.  The Adam bias corrections come from per-step tables
computed on the host in double precision exactly as torch.optim.Adam computes them.  A stride-1
target whose key crop fits the LDS (``hip.solve_run_supported``: every edit the reference's
notebooks and metrics make at layers 5-14) is solved in ONE launch -- a workgroup per pair of
out-channels keeps its weights and Adam moments in registers for all iterations; with an
``update_callback`` the same kernel is launched per iteration.  Any other target (upsampling
layers, ``linear_insert``, crops beyond the LDS) takes the step path: three kernels (+ one
projection kernel) per iteration, blocks of ``piter`` iterations captured once in a HIP graph
and replayed.  The host never synchronises unless the caller's ``update_callback`` reads a loss.
"""
import ctypes
import math
import os

import torch

from .. import hip
from ..utils.stylegan2.models import bump_weight_epoch, reference_noise


LAST = {}      # shape and path of the most recent solve (read by bench.py and the tests; diagnostics only)


def _ptr(t):
    return t.data_ptr() if t is not None else 0


class Solver:
    def __init__(self, weight, key, style, val, bias, noise_w, context, niter, piter, lr,
                 low_rank_insert, low_rank_gradient, blur_kernel=None, linear=False, upsample=None):
        """``bias is None``: the target is the demodulated convolution alone (no blur, noise, bias,
        activation; SeqTinyStyleGanRewriter) and ``upsample`` says whether it is the transposed one."""
        dev = weight.device
        plain = bias is None
        if upsample is None:
            upsample = blur_kernel is not None
        assert plain or upsample == (blur_kernel is not None)
        if linear:
            low_rank_insert = low_rank_gradient = False
        self.weight = weight                       # (1,O,I,3,3) parameter, updated in place
        w = weight.detach()
        assert w.is_contiguous() and w.dtype == torch.float32
        _, O, I, kh, kw = w.shape
        assert (kh, kw) == (3, 3)
        _, _, h, wd = key.shape
        constrained = low_rank_insert or low_rank_gradient or linear
        # the shape limits first: before any allocation, and before project_weight meets the same LDS limit
        if not hip.solve_supported(O, I, h, wd, upsample, plain, constrained):
            raise NotImplementedError(
                'the fused HIP solver takes out_ch %% 64 == 0, in_ch %% 16 == 0, in_ch <= 910 with a context '
                'direction, and for an upsampling target a key crop with (2h+1)(2w+1) + 4hw <= 16384 '
                '(got %d -> %d channels, key crop %d x %d, upsample=%s)' % (I, O, h, wd, bool(upsample)))
        self.niter, self.piter = niter, piter
        self.low_rank_insert, self.low_rank_gradient = low_rank_insert, low_rank_gradient
        f32 = dict(device=dev, dtype=torch.float32)
        self.key = key.detach().reshape(I, h, wd).contiguous().float()
        self.style = style.detach().reshape(I).contiguous().float()
        ch, cw = (2 * h + 1, 2 * wd + 1) if upsample else (h, wd)  # map the convolution writes
        oh, ow = (2 * h, 2 * wd) if (upsample and not plain) else (ch, cw)   # value map of the target
        assert tuple(val.shape[-2:]) == (oh, ow), (val.shape, (oh, ow))
        self.val = val.detach().reshape(O, oh, ow).contiguous().float()
        self.bias = None if plain else bias.detach().contiguous().float()
        self.noise = None if plain else reference_noise(1, oh * ow, dev).reshape(-1).contiguous()
        self.blur_k = blur_kernel.detach().contiguous().float() if (upsample and not plain) else None
        self.linear = linear
        self.noise_w = None if plain else noise_w.detach().reshape(1).contiguous().float()
        self.context = context.detach().contiguous().float().to(dev) if constrained else None
        self.ortho = None
        self.lam = None
        if linear:
            self.ortho = w.clone()                              # W0: weight = W0 + Lambda . context
            self.lam = torch.zeros(O, self.context.shape[0], 9, **f32)
        elif constrained:
            self.ortho = (w - hip.project_weight(w, self.context).view(w.shape)).contiguous()
        self.exp_avg = torch.zeros_like(w)
        self.exp_avg_sq = torch.zeros_like(w)
        steps = range(1, niter + 1)                # torch.optim.Adam: python doubles, then fp32 use
        self.step_size = torch.tensor([lr / (1 - 0.9 ** t) for t in steps], **f32)
        self.bc2_sqrt = torch.tensor([math.sqrt(1 - 0.999 ** t) for t in steps], **f32)
        self.counter = torch.full((1,), -1, device=dev, dtype=torch.int32)
        self.losses = torch.zeros(niter, **f32)
        n = hip.solve_scratch_elems(O, I, h, wd, upsample)      # sizes AND split-K factor as the library states them
        ks = n['ksplit']
        assert n['wsq'] == ks * O
        self.conv = torch.empty(n['conv'], **f32)
        self.wsq = torch.empty(n['wsq'], **f32)
        self.gd = torch.empty(n['gd'], **f32)
        self.c2 = torch.empty(n['c2'], **f32)
        self.grad = torch.empty(n['grad'], **f32) if (low_rank_gradient or linear) else None
        p = hip.SolveProblem()
        p.out_ch, p.in_ch, p.h, p.w = O, I, h, wd
        p.rank = self.context.shape[0] if constrained else 0
        p.key, p.style, p.val, p.bias = _ptr(self.key), _ptr(self.style), _ptr(self.val), _ptr(self.bias)
        p.noise, p.noise_w = _ptr(self.noise), _ptr(self.noise_w)
        p.context, p.ortho = _ptr(self.context), _ptr(self.ortho)
        p.weight, p.exp_avg, p.exp_avg_sq = _ptr(w), _ptr(self.exp_avg), _ptr(self.exp_avg_sq)
        p.step_size, p.bc2_sqrt = _ptr(self.step_size), _ptr(self.bc2_sqrt)
        p.step_counter, p.losses = _ptr(self.counter), _ptr(self.losses)
        p.conv, p.wsq, p.gd, p.c2, p.grad = (_ptr(self.conv), _ptr(self.wsq), _ptr(self.gd),
                                             _ptr(self.c2), _ptr(self.grad))
        p.ksplit = ks
        p.beta1, p.beta2, p.eps = 0.9, 0.999, 1e-8
        p.one_minus_beta1, p.one_minus_beta2 = 1 - 0.9, 1 - 0.999     # python doubles, rounded once
        p.w_scale = 1 / math.sqrt(I * 9)
        p.low_rank_gradient = int(low_rank_gradient)
        p.upsample = int(upsample)
        p.blur_k = _ptr(self.blur_k)
        p.linear_insert = int(linear)
        p.lambda_ = _ptr(self.lam)
        self.problem = p
        self._w = w
        # the one-launch path: decided once per solve, so the step counter and the tables stay consistent
        self.one_launch = (os.environ.get('RW_SOLVE_ONE_LAUNCH', '1') != '0'
                           and hip.solve_run_supported(O, I, h, wd, p.rank, upsample, linear))
        self.lpart = torch.empty(hip.solve_run_scratch_elems(O, I, h, wd, niter), **f32) if self.one_launch else None
        LAST.clear()
        LAST.update(one_launch=bool(self.one_launch), out_ch=O, in_ch=I, h=h, w=wd, niter=niter,
                    upsample=bool(upsample), linear=bool(linear))

    def projects(self, it):
        return self.low_rank_insert and (it % self.piter == 0 or it == self.niter - 1)

    def step(self, it, project=None):
        hip.solve_step(self.problem, self.projects(it) if project is None else project)

    def run_range(self, it0, it1, project=True):
        """Iterations [it0, it1) in one launch (one_launch targets only)."""
        hip.solve_run(self.problem, it0, it1, self.niter, self.piter, self.low_rank_insert and project, self.lpart)

    def project_now(self):
        hip.project_weight(self._w, self.context, base=self.ortho, out=self._w)

    def run(self, update_callback=None):
        """The kernels write the parameter through its raw pointer, which torch's version counter cannot
        see: every cache of tensors derived from it (packed weights, squared sums) is invalidated before each
        callback -- a callback that renders through the model sees the stepped weight, as the reference's
        does (rewrite/ganrewrite.py:288-289) -- and, whatever happens (an exception in the callback,
        KeyboardInterrupt in the middle of 2001 steps), once more on the way out."""
        try:
            self._run(update_callback)
        finally:
            bump_weight_epoch()

    def _run(self, update_callback):
        niter, piter = self.niter, self.piter
        if update_callback is not None and getattr(update_callback, 'loss_only', False):
            # a callback that DECLARES it looks at (it, loss) only -- a progress bar, a loss printout: the solve runs as
            # if there were none (one launch, or the replayed graph) and the callbacks are delivered afterwards, in
            # order, against the loss buffer.  Callbacks without the mark keep the reference's contract below.
            self._run(None)
            for it in range(niter):
                update_callback(it, self.losses[it])
            return
        if self.one_launch and update_callback is None:
            self.run_range(0, niter)
            return
        if update_callback is not None:
            # reference order: step, callback (sees the stepped, not yet projected weight), projection
            for it in range(niter):
                if self.one_launch:
                    self.run_range(it, it + 1, project=False)
                else:
                    self.step(it, project=False)
                bump_weight_epoch()
                update_callback(it, self.losses[it])
                if self.projects(it):
                    self.project_now()
            return
        use_graph = os.environ.get('RW_SOLVE_GRAPH', '1') != '0' and niter >= 3 * piter + 1 and piter > 1
        if not use_graph:
            for it in range(niter):
                self.step(it)
            return
        self.step(0)                                # eager first step: loads every code object
        blocks = (niter - 1) // piter
        # iterations 1..piter: the projection falls on the last one of the block
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for j in range(1, piter + 1):
                self.step(j, project=self.low_rank_insert and j == piter)
        # the capture itself did not execute; replay once per block
        for _ in range(blocks):
            graph.replay()
        for it in range(1 + blocks * piter, niter):
            self.step(it)


def run(weight, key, style, val, bias, noise_w, context, niter=2001, piter=10, lr=0.05,
        low_rank_insert=True, low_rank_gradient=False, update_callback=None, blur_kernel=None, upsample=None,
        linear=False):
    solver = Solver(weight, key, style, val, bias, noise_w, context, niter, piter, lr,
                    low_rank_insert, low_rank_gradient, blur_kernel=blur_kernel, linear=linear,
                    upsample=upsample)
    solver.run(update_callback)
    return solver
