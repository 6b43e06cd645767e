"""Rewriting a layer of a generator as a rank-constrained associative-memory update.

Drop-in for rewrite/ganrewrite.py: ``ProgressiveGanRewriter`` (:24) and its StyleGAN
subclasses (:658, :732, :742) keep their constructor arguments, attributes and methods
(level B1), so ``rewriteapp.GanRewriteApp`` and the ``metrics/`` drivers
call them unchanged.  Underneath:

* the three sub-models (context | target | rendering) are ``nethook.subsequence`` views sharing
  the parameters of a private deep copy of the model, as in the reference (:47-58);
* the key statistics sweep feeds the NCHW key map straight into the fp32-MFMA second-moment
  kernel, optionally sharded over ranks with one RCCL all-reduce (``parallel``);
* ``insert`` on the targets the reference's StyleGAN rewriters define runs the fused HIP solver (four kernels
  per iteration, ten iterations per HIP graph, no host synchronisation unless the caller's
  ``update_callback`` asks for one); any other target takes the autograd path the reference describes --
  ProgGAN's plain ``nn.Conv2d`` on torch ops, a SeqStyleGAN2 target of several layers / with a hooked module /
  with a goal batch > 1 through the adjoints of utils/stylegan2/grad.py on the HIP kernels;
* the tiny dense factorizations (fp64 ``eigh`` for ZCA, ``svd``/``qr`` of the ~100 x 512 key
  matrix) run in LAPACK on the host, as in the reference's CPU configuration.
"""
import copy
import json
import math
import os
import random
import time
import warnings
from collections import OrderedDict

import torch

from ..utils import nethook, pbar, renormalize, tally
from .. import hip, parallel

# Debug globals the reference's notebooks peek at (rewrite/ganrewrite.py:13-14); kept assigned.
(all_obs, all_weight, all_CinvK, all_kCinvK, e_val, e_vec, kbasis, row_dirs, q) = (None,) * 9


class ProgressiveGanRewriter(object):
    def __init__(self, model, zds, layernum, cachedir=None,
                 low_rank_insert=True,       # restrict the update to the context subspace
                 low_rank_gradient=False,    # also project every gradient
                 use_linear_insert=False,    # optimise Lambda with W = W0 + Lambda D
                 tight_paste=True,           # optimise over the pasted crop, not the whole map
                 alpha_area=True,            # composite with the painted area, not its bounding box
                 key_method='zca'):          # or 'svd', 'mean', 'gandissect'
        self.firstlayer, self.lastlayer = self.maplayers(layernum)
        self.cachedir = cachedir
        self.tight_paste = tight_paste
        self.alpha_area = alpha_area
        self.key_method = key_method
        self.unit_rq = None
        self.unit_rs = None
        self.cad_rq = None
        self.low_rank_insert = low_rank_insert
        self.low_rank_gradient = low_rank_gradient
        self.use_linear_insert = use_linear_insert
        self.device = next(model.parameters()).device
        self.zds = zds
        self.model = copy.deepcopy(model)
        self.context_model = nethook.subsequence(
            self.model, upto_layer=self.firstlayer, share_weights=True)
        self.target_model = nethook.subsequence(
            self.model, first_layer=self.firstlayer, last_layer=self.lastlayer, share_weights=True)
        self.rendering_model = nethook.subsequence(
            self.model, after_layer=self.lastlayer, share_weights=True)
        with torch.no_grad():
            k = self.context_model(self.get_z(0))
            v = self.target_model(k)
            x = self.rendering_model(v)
        self.k_shape = self.context_acts(k).shape
        self.v_shape = self.target_acts(v).shape
        self.x_shape = self.rendered_image(x).shape
        self.c_matrix = self.collect_2nd_moment().to(self.device)
        self.zca_matrix = zca_from_cov(self.c_matrix)

    # ---- model plumbing -------------------------------------------------------------------
    def maplayers(self, layernum):
        return 'layer%d.conv' % layernum, 'layer%d.conv' % layernum

    def model_state_dict(self):
        parts = [m.state_dict() for m in (self.context_model, self.target_model, self.rendering_model)]
        merged = {}
        for part in parts:
            merged.update(part)
        assert len(merged) == sum(len(p) for p in parts)
        return merged

    def context_acts(self, context_out):
        return context_out

    def target_acts(self, target_out):
        return target_out

    def rendered_image(self, rendered_out):
        return rendered_out

    def detach(self, v):
        return v.detach()

    def merge_target_output(self, target_out, new_acts, crop_bounds):
        return new_acts

    def get_z(self, imgnum):
        return self.zds[imgnum][0][None].to(self.device)

    def sample_image_from_latent(self, z):
        return self.rendering_model(self.target_model(self.context_model(z)))

    def target_weights(self):
        for name, param in self.target_model.named_parameters():
            if 'weight' in name:
                return param
        raise ValueError('target model has no weight')

    def rf(self, fn):
        return None if self.cachedir is None else os.path.join(self.cachedir, fn)

    def _kernels(self):
        """True when this rewriter's tensors live on a HIP device (kernels mandatory)."""
        return hip.on_device(next(self.model.parameters()))

    # ---- key statistics -------------------------------------------------------------------
    def collect_2nd_moment(self):
        """Uncentred second moment C of the keys over the seed sweep (cached in r2m.npz)."""
        with torch.no_grad(), pbar.quiet():
            on_gpu = self._kernels()

            def key_rows(zbatch):
                acts = self.context_acts(self.context_model(zbatch.to(self.device)))
                if on_gpu:
                    return acts          # NCHW straight into the MFMA kernel
                return acts.permute(0, 2, 3, 1).reshape(-1, acts.shape[1])
            if on_gpu and self._noise_periodic():
                # many reference batches of 10 per launch; each seed keeps its reference noise row
                from ..utils.stylegan2.models import noise_batch_period
                with noise_batch_period(10):
                    r2m = tally.tally_second_moment(key_rows, self.zds, batch_size=self._sweep_batch(),
                                                    cachefile=self.rf('r2m.npz'),
                                                    shard=parallel.shard(), nchw=True)
            else:
                r2m = tally.tally_second_moment(key_rows, self.zds, cachefile=self.rf('r2m.npz'),
                                                shard=parallel.shard(), nchw=on_gpu)
            return r2m.moment()

    # seeds per launch of the statistics sweeps on the GPU (a multiple of the reference's batch of 10), at most.  510 and
    # not 500: the kernels of the context forward launch 16 / 8 / 4 / 2 / 1 workgroups per image or pair of images, and
    # the chip runs 512 at a time -- 510 seeds make whole rounds of it (15.94 of 16) where 500 leave the last round 60 %
    # full (layer-8 sweep of the 1024 model: +2.7 %, same box)
    sweep_batch = 510
    sweep_bytes = 2 << 30     # ... and at most this many bytes of key map per launch

    def _sweep_batch(self):
        """Seeds per launch: large launches amortise the host side and the tails of ~50 kernel launches per batch
        (layer 8 of the 1024 model: 31.0 k seeds/s at 500 per launch against 29.9 k at 250), but the key map of a launch
        stays within sweep_bytes, and a sharded sweep is cut into a multiple of `world` launches (they are dealt
        round-robin: parallel.balanced_batch), never fewer than there are ranks."""
        sh = parallel.shard()
        world = sh[1] if sh else 1
        cap = self.sweep_batch
        k_shape = getattr(self, 'k_shape', None)
        if k_shape is not None:
            per_seed = 4
            for n in tuple(k_shape)[1:]:
                per_seed *= int(n)
            cap = min(cap, max(10, self.sweep_bytes // per_seed // 10 * 10))
        return parallel.balanced_batch(len(self.zds), cap, world)

    def _noise_periodic(self):
        """Large sweep launches are only equivalent to the reference's batches of 10 if the dataset
        length keeps every launch a multiple of 10 (the last reference batch may be ragged)."""
        return False

    def square_scales_for_units(self):
        if self.unit_rs is None:
            with pbar.quiet(), torch.no_grad():
                on_gpu = self._kernels()

                def squared_units(zbatch):
                    acts = self.context_acts(self.context_model(zbatch.to(self.device))).detach()
                    if on_gpu:
                        return acts
                    return acts.permute(0, 2, 3, 1).reshape(-1, acts.shape[1]).pow(2)
                if on_gpu and self._noise_periodic():
                    from ..utils.stylegan2.models import noise_batch_period
                    with noise_batch_period(10):
                        rv = tally.tally_mean(squared_units, self.zds, batch_size=self._sweep_batch(),
                                              cachefile=self.rf('unit_rs.npz'), nchw=True, square_input=True,
                                              shard=parallel.shard())
                else:
                    rv = tally.tally_mean(squared_units, self.zds, cachefile=self.rf('unit_rs.npz'),
                                          nchw=on_gpu, square_input=on_gpu, shard=parallel.shard())
                self.unit_rs = rv.mean()
        return self.unit_rs

    def covariance_adjusted_query_key(self, k):
        """C^-1 k by least squares (the reference's torch.lstsq, :101-105), in float32 LAPACK on the
        host like the reference's CPU configuration: C is ill-conditioned, so the precision of this
        solve is part of the result -- and so is the driver: torch 1.x lstsq was gels (QR, no rank
        truncation); torch.linalg.lstsq's default gelsy would drop the directions of C below
        eps * 512 * lambda_max and return a different key."""
        c = self.c_matrix.cpu()
        rhs = (k[:, None] if k.dim() == 1 else k.permute(1, 0)).cpu()
        with host_linalg_threads():
            sol = torch.linalg.lstsq(c, rhs, driver='gels').solution
        sol = sol.to(k.dtype).to(k.device)
        return sol[:, 0] if k.dim() == 1 else sol.permute(1, 0)

    def covariance_adjusted_key(self, k, kout):
        return self.covariance_adjusted_query_key(k)

    def zca_whitened_query_key(self, k):
        if k.dim() == 1:
            return torch.mm(self.zca_matrix, k[:, None])[:, 0]
        return torch.mm(self.zca_matrix, k.permute(1, 0)).permute(1, 0)

    # ---- requests -------------------------------------------------------------------------
    def apply_edit(self, request, rank=1, niter=2001, piter=10, lr=0.05, update_callback=None,
                   single_key=-1):
        """Applies an edit request as saved by the UI: {'object': [imgnum, mask], 'paste': [...],
        'key': [[imgnum, mask], ...]}."""
        o_imgnum, o_mask = request['object']
        p_imgnum, p_mask = request['paste']
        key_examples = request.get('key', [(p_imgnum, p_mask)])
        if single_key >= 0:
            print('Using only key', single_key, 'out of a total', len(key_examples))
            key_examples = [key_examples[single_key]]
        obj_acts, _, obj_area, _ = self.object_from_selection(o_imgnum, o_mask)
        goal_in, goal_out, _, _ = self.paste_from_selection(p_imgnum, p_mask, obj_acts, obj_area)
        mkey = self.multi_key_from_selection(key_examples, rank=rank)
        return self.insert(goal_in, goal_out, mkey, update_callback=update_callback,
                           niter=niter, piter=piter, lr=lr)

    def apply_erase(self, request, rank=1, drank=30, niter=2001, piter=10, lr=0.05,
                    update_callback=None):
        p_imgnum, p_mask = request['paste']
        key_examples = request.get('key', [(p_imgnum, p_mask)])
        goal_in, goal_out = self.erase_from_selection(p_imgnum, p_mask, key_examples, drank)
        mkey = self.multi_key_from_selection(key_examples, rank=rank)
        self.insert(goal_in, goal_out, mkey, update_callback=update_callback,
                    niter=niter, piter=piter, lr=lr)

    def apply_overfit(self, request, niter=20001, lr=0.01, update_callback=None):
        raise NotImplementedError(
            'apply_overfit trains every weight against a VGG16 perceptual loss '
            '(rewrite/ganrewrite.py:171-181,300-331); torchvision weights are unavailable and the '
            'path is outside the rule-editing hot path (SURVEY.md section 8a row d5)')

    def zero(self, context, amount=0.0):
        """W <- W - P(W) + amount * P(1)   (:190-195)"""
        weight = self.target_weights()
        with torch.no_grad():
            pw = projected_conv(weight, context)
            weight[...] = weight - pw
            if amount != 0.0:
                weight[...] = weight + amount * projected_conv(torch.ones_like(weight), context)
        _weights_changed()

    # ---- the solve ------------------------------------------------------------------------
    def insert(self, key, val, context=None, update_callback=None, niter=2001, piter=10, lr=0.05,
               return_timing=False):
        if self.use_linear_insert:
            return self.linear_insert(key, val, context, update_callback=update_callback,
                                      niter=niter, lr=lr, return_timing=return_timing)
        sync = (lambda: torch.cuda.synchronize()) if self.device.type == 'cuda' else (lambda: None)
        if return_timing:
            sync()
            started = time.time()
        key, val = self.detach(key), self.detach(val)
        self._run_insert(key, val, context, update_callback, niter, piter, lr)
        if return_timing:
            sync()
            return (time.time() - started) * 1000

    def _run_insert(self, key, val, context, update_callback, niter, piter, lr):
        """Projected-gradient Adam through torch autograd (targets made of torch ops)."""
        weight = self.target_weights()
        constrained = self.low_rank_insert or self.low_rank_gradient
        if constrained:
            with torch.no_grad():
                ortho = weight - projected_conv(weight, context)
        optimizer = torch.optim.Adam([weight], lr=lr)
        goal = self.target_acts(val)
        for it in range(niter):
            with torch.enable_grad():
                loss = torch.nn.functional.l1_loss(goal, self.target_acts(self.target_model(key)))
                optimizer.zero_grad()
                loss.backward()
                if self.low_rank_gradient:
                    weight.grad[...] = projected_conv(weight.grad, context)
                optimizer.step()
                if update_callback is not None:
                    update_callback(it, loss)
                if self.low_rank_insert and (it % piter == 0 or it == niter - 1):
                    with torch.no_grad():
                        weight[...] = ortho + projected_conv(weight, context)
        _weights_changed()

    def linear_insert(self, key, val, context=None, update_callback=None, niter=2001, lr=0.05,
                      return_timing=False):
        """weight = W0 + Lambda . context with Adam on Lambda alone (rewrite/ganrewrite.py:201-252), through torch
        autograd for targets made of torch ops: the module that owns the target weight gets a forward that
        rebuilds its weight from Lambda on every call and is restored afterwards, the frozen parameters stay
        frozen (:206), the learned update is written into the original parameter (:246-249).

        The reference sizes Lambda as (ws[0], ws[1], rank, ws[3], ws[4]) and therefore raises IndexError on
        ProgGAN's 4-d nn.Conv2d weight (use_linear_insert=True never worked on its config 1); here a 4-d weight
        (O, I, ky, kx) gets Lambda (O, rank, ky, kx), a 5-d one (1, O, I, ky, kx) the reference's shape."""
        sync = (lambda: torch.cuda.synchronize()) if self.device.type == 'cuda' else (lambda: None)
        if return_timing:
            sync()
            started = time.time()
        nethook.set_requires_grad(False, self.model)
        key, val = self.detach(key), self.detach(val)
        weight = self.target_weights()
        owner = [m for m in self.target_model.modules() if getattr(m, 'weight', None) is weight][0]
        five = weight.dim() == 5
        rule = 'godyx,di->goiyx' if five else 'odyx,di->oiyx'
        lam = torch.zeros(tuple(weight.shape[:2 if five else 1]) + (context.shape[0],) + tuple(weight.shape[-2:]),
                          device=weight.device, dtype=weight.dtype, requires_grad=True)
        hooked_before = owner.__dict__.get('forward')        # an instance-level forward (a nethook edit), if any
        plain_forward = owner.forward
        del owner._parameters['weight']

        from ..utils.stylegan2 import models as sg

        def forward_with_lambda(*args, **kwargs):
            owner.weight = weight + torch.einsum(rule, lam, context)
            sg.bump_weight_epoch()      # a fresh tensor every call: derived (packed) weights must never be looked up by
            #                             its address and version alone
            return plain_forward(*args, **kwargs)
        owner.forward = forward_with_lambda
        try:
            optimizer = torch.optim.Adam([lam], lr=lr)
            goal = self.target_acts(val)
            for it in range(niter):
                with torch.enable_grad():
                    loss = torch.nn.functional.l1_loss(goal, self.target_acts(self.target_model(key)))
                    optimizer.zero_grad()
                    loss.backward()
                    optimizer.step()
                    if update_callback is not None:
                        update_callback(it, loss)
        finally:
            with torch.no_grad():
                weight[...] = weight + torch.einsum(rule, lam.detach(), context)
            if 'weight' in owner.__dict__:
                del owner.weight
            owner.register_parameter('weight', weight)
            if hooked_before is not None:
                owner.forward = hooked_before
            else:
                del owner.forward         # the instance attribute; the class's forward is back
        _weights_changed()
        if return_timing:
            sync()
            return (time.time() - started) * 1000

    def all_weights_insert(self, *args, **kwargs):
        raise NotImplementedError('see apply_overfit')

    # ---- context direction ----------------------------------------------------------------
    def _key_observations(self, imgnum_mask_pairs):
        """[(rows (H*W, C), context output, mask weights (H*W, 1))] for every key example."""
        observed = []
        for imgnum, mask in imgnum_mask_pairs:
            k_outs = self.context_model(self.get_z(imgnum))
            k_acts = self.context_acts(k_outs)
            area = renormalize.from_url(mask, target='pt', size=self.k_shape[2:])[0]
            observed.append((k_acts.permute(0, 2, 3, 1).reshape(-1, k_acts.shape[1]), k_outs,
                             area.reshape(-1)[:, None].to(k_acts.device)))
        return observed

    def multi_key_from_selection(self, imgnum_mask_pairs, rank=1, key_method=None):
        """Orthonormal rows (rank, C) spanning the context subspace d.  'zca' (:339-374): whiten
        the selected keys with Z = C^-1/2, take the top right-singular vectors, whiten again,
        orthogonalise, and orient each along the mean selected key."""
        global all_obs, all_weight, row_dirs, q
        key_method = key_method or self.key_method
        with torch.no_grad():
            if key_method == 'zca':
                observed = self._key_observations(imgnum_mask_pairs)
                sel = [(w > 0).nonzero()[:, 0] for _, _, w in observed]
                all_obs = torch.cat([obs[s] for (obs, _, _), s in zip(observed, sel)])
                all_weight = torch.cat([w[w > 0] for _, _, w in observed])
                all_zca_k = torch.cat([(w * self.zca_whitened_query_key(obs))[s]
                                       for (obs, _, w), s in zip(observed, sel)])
                zk = all_zca_k.cpu()
                zca = self.zca_matrix.cpu()
                with host_linalg_threads():
                    _, _, vh = torch.linalg.svd(zk, full_matrices=False)
                    top = vh.t()[:, :rank]
                    row_dirs = torch.mm(zca, top).t()
                    qmat, _ = torch.linalg.qr(row_dirs.t())
                signs = (qmat * zk.sum(0)[:, None]).sum(0).sign()
                q = qmat * signs[None, :]
                return q.t().contiguous().to(self.device)
            if key_method == 'gandissect':
                # one-hot rows for the units whose activations under the masks are the most unusual:
                # mean of -log(1 - quantile rank) (:375-400)
                observed = self._key_observations(imgnum_mask_pairs)
                all_obs = torch.cat([obs for obs, _, _ in observed])
                all_weight = torch.cat([w for _, _, w in observed])
                rq = self.quantiles_for_units()
                logscore = -torch.log(1.0 - rq.normalize(all_obs.permute(1, 0))).permute(1, 0)
                logscore = logscore.to(all_obs.device)
                mean_logscore = (logscore * all_weight).sum(0) / all_weight.sum()
                # rank 1.0 (an activation that IS the sample maximum) gives inf * 0 = NaN; torch.sort
                # documents NaN as the greatest value, which a radix sort on the device does not honour
                # for negative NaNs: make the reference's ordering explicit and the tie order stable
                mean_logscore = torch.where(torch.isnan(mean_logscore),
                                            torch.full_like(mean_logscore, float('inf')), mean_logscore)
                top = mean_logscore.sort(descending=True, stable=True)[1][:rank]
                result = torch.zeros(rank, all_obs.shape[1], device=all_obs.device)
                result[torch.arange(rank), top] = 1.0
                return result
            assert key_method in ['svd', 'mean']
            gathered = []
            for imgnum, mask in imgnum_mask_pairs:
                k_outs = self.context_model(self.get_z(imgnum))
                k_acts = self.context_acts(k_outs)
                area = renormalize.from_url(mask, target='pt', size=self.k_shape[2:])[0]
                weighted = (k_acts[0] * area[None].to(self.device)).permute(1, 2, 0).reshape(
                    -1, k_acts.shape[1])
                gathered.append((weighted[weighted.norm(2, dim=1) > 0], k_outs))
            all_k = torch.cat([self.covariance_adjusted_key(nk, ko) for nk, ko in gathered])
            just_avg = all_k.mean(0)
            if key_method == 'mean':
                assert rank == 1
                return just_avg[None, :] / just_avg.norm()
            with host_linalg_threads():
                u, _, _ = torch.linalg.svd(all_k.permute(1, 0).cpu(), full_matrices=True)
            u = u.to(self.device)
            if (just_avg * u[:, 0]).sum() < 0:
                u[:, 0] = -u[:, 0]
            assert u.shape[1] >= rank
            return u.permute(1, 0)[:rank].contiguous()

    def query_key_from_selection(self, imgnum, mask):
        area = renormalize.from_url(mask, target='pt', size=self.k_shape[2:])[0]
        with torch.no_grad():
            k_acts = self.context_acts(self.context_model(self.get_z(imgnum)))
            mean = (k_acts[0] * area[None].to(self.device)).sum(2).sum(1) / (1e-10 + area.sum())
        k = self.covariance_adjusted_query_key(mean)
        return k / (1e-10 + k.norm(2))

    def is_empty_mask(self, mask):
        return renormalize.from_url(mask, target='pt')[0].sum() == 0.0

    # ---- goals ----------------------------------------------------------------------------
    # Host-side glue around three renderings (rewrite/ganrewrite.py:442-520 of the reference: same names, same return
    # tuples).  Nothing here is differentiated -- `insert` detaches key and value (:265) -- so every rendering runs under
    # no_grad and the sub-models keep their fused kernels (under grad mode a styled convolution runs module by module,
    # utils/stylegan2/grad.py).
    def _mask_on(self, mask, shape):
        """The mask (a data URL from the UI) as a 0/1 map at a layer's resolution."""
        return renormalize.from_url(mask, target='pt', size=shape[2:])[0]

    def _render_to_target(self, imgnum):
        """(bag in front of the edited layer, bag behind it) of image `imgnum`, unedited."""
        with torch.no_grad():
            before = self.context_model(self.get_z(imgnum))
            return before, self.target_model(before)

    def _goal_pair(self, before, keys, key_box, behind, values, value_box):
        """(goal_in, goal_out): the two bags with their key / value maps swapped in, cropped to the boxes (None: whole map)."""
        return (self.merge_target_output(before, keys, key_box), self.merge_target_output(behind, values, value_box))

    def object_from_selection(self, imgnum, mask):
        """The value patch to copy: activations of the target layer under the mask's bounding box."""
        where = self._mask_on(mask, self.v_shape)
        _, behind = self._render_to_target(imgnum)
        top, left, bottom, right = box = positive_bounding_box(where)
        patch = self.target_acts(behind)[:, :, top:bottom, left:right]
        return patch, behind, where[top:bottom, left:right], box

    def paste_from_selection(self, imgnum, mask, obj_acts, obj_area):
        before, behind = self._render_to_target(imgnum)
        keys = self.context_acts(before)
        pasted, bounds = paste_clip_at_center(self.target_acts(behind), obj_acts,
                                              centered_location(self._mask_on(mask, self.v_shape)),
                                              obj_area if self.alpha_area else None)
        viz_out = self.merge_target_output(behind, pasted, None)        # the whole pasted map, for the UI
        key_box = value_box = None
        values = pasted
        if self.tight_paste:
            keys, values, key_box, value_box = crop_clip_to_bounds(keys, pasted, bounds)
        goal_in, goal_out = self._goal_pair(before, keys, key_box, behind, values, value_box)
        return goal_in, goal_out, viz_out, bounds

    def normdissect_units(self, imgnum_mask_pairs, rank):
        """Units whose squared activation, relative to its sweep mean, is largest under the masks."""
        with torch.no_grad():
            seen = self._key_observations(imgnum_mask_pairs)
            acts = torch.cat([o for o, _, _ in seen])
            weight = torch.cat([w for _, _, w in seen])
            per_unit = self.square_scales_for_units().to(acts.device)
            score = ((acts.pow(2) / per_unit[None, :]) * weight).sum(0) / weight.sum()
            return score.sort(descending=True)[1][:rank]

    def erase_from_selection(self, imgnum, mask, context_mask_pairs, rank):
        before, behind = self._render_to_target(imgnum)
        keys = self.context_acts(before)
        with torch.no_grad():
            silenced = keys.clone()
            silenced[:, self.normdissect_units(context_mask_pairs, rank)] = 0.0
            values = self.target_acts(self.target_model(self.merge_target_output(before, silenced, None)))
        key_box = value_box = None
        if self.tight_paste:
            key_box = positive_bounding_box(self._mask_on(mask, self.k_shape))
            value_box = positive_bounding_box(self._mask_on(mask, self.v_shape))
        return self._goal_pair(before, keys, key_box, behind, values, value_box)

    def rgb_from_selection(self, imgnum, mask):
        where = self._mask_on(mask, self.x_shape)
        with torch.no_grad():
            image = self.model(self.get_z(imgnum))
        top, left, bottom, right = box = positive_bounding_box(where)
        return image[:, :, top:bottom, left:right], image, where[top:bottom, left:right], box

    def rgbpaste_from_selection(self, imgnum, mask, obj_rgb, obj_area):
        z = self.get_z(imgnum)
        with torch.no_grad():
            changed, bounds = paste_clip_at_center(self.model(z), obj_rgb,
                                                   centered_location(self._mask_on(mask, self.x_shape)), obj_area)
        return z, changed, bounds

    # ---- UI conveniences ------------------------------------------------------------------
    def _flat_units(self, zbatch):
        acts = self.context_acts(self.context_model(zbatch.to(self.device))).detach()
        return acts.permute(0, 2, 3, 1).reshape(-1, acts.shape[1])

    def _sweep(self, fn, *args, **kwargs):
        """Runs a tally over self.zds; on the GPU in launches of sweep_batch seeds with the reference's
        batch-of-10 noise rows (see collect_2nd_moment)."""
        with pbar.quiet(), torch.no_grad():
            if self._kernels() and self._noise_periodic():
                from ..utils.stylegan2.models import noise_batch_period
                with noise_batch_period(10):
                    return fn(*args, batch_size=self._sweep_batch(), **kwargs)
            return fn(*args, **kwargs)

    def quantiles_for_units(self):
        if self.unit_rq is None:
            self.unit_rq = self._sweep(tally.tally_quantile, self._flat_units, self.zds,
                                       cachefile=self.rf('unit_rq.npz'))
        return self.unit_rq

    def quantiles_for_covariance_adjusted_directions(self):
        if self.cad_rq is None:
            def adjusted(zbatch):
                outs = self.context_model(zbatch.to(self.device))
                acts = self.context_acts(outs)
                flat = acts.permute(0, 2, 3, 1).reshape(-1, acts.shape[1])
                return self.covariance_adjusted_key(flat, outs)
            self.cad_rq = self._sweep(tally.tally_quantile, adjusted, self.zds,
                                      cachefile=self.rf('unit_cad.npz'))
        return self.cad_rq

    def ranking_for_key(self, key, k=12):
        """Seeds whose key map responds most to ``key`` + the quantile statistics of the response
        (the UI's "Search"; rewrite/ganrewrite.py:582-594)."""
        tensorkey = key.to(self.device)[None, :, None, None]

        def image_max_sel(zbatch):
            acts = self.context_acts(self.context_model(zbatch.to(self.device)))
            heatmap = (acts * tensorkey).sum(dim=1)
            return heatmap.reshape(heatmap.shape[0], -1).max(1)[0], heatmap.reshape(-1)[:, None]
        topk, rq = self._sweep(tally.tally_topk_and_quantile, image_max_sel, self.zds, k=k)
        return topk.result()[1], rq

    def _overlay(self):
        """utils.imgviz (heat-map / mask overlays, outside the hot path): this package's own if it has one,
        else the reference's through install_reference_aliases(reference_root=...)."""
        import importlib
        for name in ('rewriting_amd.utils.imgviz', 'utils.imgviz'):
            try:
                return importlib.import_module(name)
            except ImportError:
                continue
        raise NotImplementedError('heat-map / mask overlays need utils.imgviz (visualisation is outside the hot '
                                  'path; install_reference_aliases(reference_root=...) provides the '
                                  "reference's); plain rendering works")

    def render_object(self, target_output, obj_area=None, box=None):
        with torch.no_grad():
            imgdata = self.rendered_image(self.rendering_model(target_output))
        if box is None:
            return renormalize.as_image(imgdata[0])
        t, l, b, r = box
        lowres = torch.zeros(self.v_shape[2:])
        lowres[t:b, l:r] = 1
        iv = self._overlay().ImageVisualizer(imgdata.shape[2:])
        return iv.masked_image(imgdata, activations=lowres, level=0.0, border_color=[255, 0, 0],
                               thickness=3)

    def render_image(self, imgnum, key=None, level=None, mask=None, **kwargs):
        with torch.no_grad():
            z = self.get_z(imgnum)
            imgdata = self.rendered_image(self.sample_image_from_latent(z))
        if key is not None and level is not None:
            with torch.no_grad():
                acts = self.context_acts(self.context_model(z))
            heatmap = (acts * key.to(self.device)[None, :, None, None]).sum(dim=1)[0]
            iv = self._overlay().ImageVisualizer(imgdata.shape[2:])
            return iv.masked_image(imgdata, heatmap, level=level, **kwargs)
        if mask is not None:
            iv = self._overlay().ImageVisualizer(imgdata.shape[2:])
            return iv.masked_image(imgdata, mask=mask, **kwargs)
        return renormalize.as_image(imgdata[0])

    def render_image_batch(self, imgnums, key=None, level=None, **kwargs):
        """Batches of three, as the reference renders them (rewrite/ganrewrite.py:626-650): an image's noise row is
        its position in its batch of three (quirk Q1), for the picture and for the heat map alike."""
        results = []
        for i in range(0, len(imgnums), 3):
            with torch.no_grad():
                zb = torch.cat([self.get_z(n) for n in imgnums[i:i + 3]])
                batch = self.rendered_image(self.sample_image_from_latent(zb))
                if key is not None and level is not None:
                    acts = self.context_acts(self.context_model(zb))
                    heatmap = (acts * key.to(self.device)[None, :, None, None]).sum(dim=1)
            if key is not None and level is not None:
                iv = self._overlay().ImageVisualizer(batch.shape[2:])
                results.extend(iv.masked_image(img, heatmap[j], level=level, **kwargs) for j, img in enumerate(batch))
            else:
                results.extend(renormalize.as_image(img) for img in batch)
        return results


class SeqStyleGanRewriter(ProgressiveGanRewriter):
    """Rewrites ``layerN.sconv.mconv.dconv`` of a SeqStyleGAN2; the key is the style-modulated
    input of that convolution, the value the output of ``layerN.sconv.activate`` (:658-729)."""

    def __init__(self, model, zds, layernum, **kwargs):
        super().__init__(model, zds, layernum, **kwargs)

    def maplayers(self, layernum):
        return 'layer%d.sconv.mconv.dconv' % layernum, 'layer%d.sconv.activate' % layernum

    def detach(self, v):
        if isinstance(v, dict):
            return type(v)({k: d.detach() for k, d in v.items()})
        return v.detach()

    def context_acts(self, context_out):
        return context_out.fmap

    def target_acts(self, target_out):
        return target_out.fmap

    def _noise_periodic(self):
        n = len(self.zds)
        return n % 10 == 0 or n < 10

    def merge_target_output(self, target_out, new_acts, crop_bounds):
        merged = type(target_out)({k: d.detach() for k, d in target_out.items()})
        if crop_bounds is not None:
            t, l, b, r = crop_bounds
            merged.output = merged.output[:, :, t:b, l:r]
        merged.fmap = new_acts
        return merged

    def sample_image_patch(self, z, act_crop_size, seed=(None, None), act=False, size=None):
        out = self.context_model(z)
        fmap, img = out['fmap'], out['output']
        assert act_crop_size <= fmap.size(2)
        if seed[0] is not None:
            xi, yi = seed
        else:
            xi = random.randint(0, fmap.shape[2] - act_crop_size)
            yi = random.randint(0, fmap.shape[3] - act_crop_size)
        xf, yf = xi + act_crop_size, yi + act_crop_size
        ratio = 1 if fmap.shape[2:] == img.shape[2:] else 2
        out['output'] = img[:, :, ratio * xi:ratio * xf, ratio * yi:ratio * yf]
        out['fmap'] = fmap[:, :, xi:xf, yi:yf]
        result = self.rendering_model(self.target_model(out))
        if not act:
            return result
        top = out['fmap'].max(3)[0].max(2)[0].max(1)[1].item()
        iv = self._overlay().ImageVisualizer((size, size))
        return result, iv.heatmap(out['fmap'][0, top], mode='nearest')

    # ---- fused HIP solve ------------------------------------------------------------------
    def _hip_solvable(self, key):
        """The module chains the fused solver restates: [adain] dconv [blur] noise activate
        (SeqStyleGanRewriter, SeqPreStyleGanRewriter) and dconv alone (SeqTinyStyleGanRewriter).
        Returns (adain, dconv, blur, noise, act) with None for absent stages, or None."""
        from ..utils.stylegan2 import models as sg
        mods = list(self.target_model.modules())
        leaves = [m for m in mods if len(list(m.children())) == 0]
        if not self._kernels() or not leaves:
            return None
        adain = None
        if isinstance(leaves[0], sg.ApplyStyle):
            adain, leaves = leaves[0], leaves[1:]
        if not leaves or not isinstance(leaves[0], sg.DemodulatedConv2dF):
            return None
        dconv = leaves[0]
        blur = noise = act = None
        if len(leaves) > 1:
            if len(leaves) not in (3, 4):
                return None
            noise, act = leaves[-2], leaves[-1]
            blur = leaves[1] if len(leaves) == 4 else None
            if not (isinstance(noise, sg.NoiseInjectionF) and isinstance(act, sg.FusedLeakyReLUF)):
                return None
            if dconv.upsample != (blur is not None) or (blur is not None and not isinstance(blur, sg.BlurF)):
                return None
            if blur is not None and (tuple(blur.kernel.shape) != (4, 4) or tuple(blur.pad) != (1, 1)):
                return None
        if not dconv.demodulate or key.fmap.shape[0] != 1:
            return None
        if any('forward' in m.__dict__ for m in mods):
            return None                       # someone hooked the target: keep module semantics
        return adain, dconv, blur, noise, act

    def _run_insert(self, key, val, context, update_callback, niter, piter, lr, linear=False):
        parts = self._hip_solvable(key) if isinstance(key, dict) else None
        if parts is None:
            # any other target -- several layers, a hooked module, a goal batch > 1: the reference's own loop
            # (loss.backward() through the module chain + torch.optim.Adam), every module's forward and adjoint on
            # the HIP kernels (utils/stylegan2/grad.py)
            if linear:
                return ProgressiveGanRewriter.linear_insert(self, key, val, context, update_callback=update_callback,
                                                            niter=niter, lr=lr)
            return super()._run_insert(key, val, context, update_callback, niter, piter, lr)
        from . import hipsolve
        adain, dconv, blur, noise, act = parts
        # a target that starts at adain sees the un-modulated map: ApplyStyle (models.py:616-620) first
        fmap = key.fmap if adain is None else hip.style_mul(key.fmap.contiguous(), key.style)
        hipsolve.run(dconv.weight, fmap, key.style, val.fmap,
                     None if act is None else act.bias, None if noise is None else noise.weight, context,
                     niter=niter, piter=piter, lr=lr,
                     low_rank_insert=self.low_rank_insert, low_rank_gradient=self.low_rank_gradient,
                     update_callback=update_callback,
                     blur_kernel=None if blur is None else blur.kernel, upsample=bool(dconv.upsample),
                     linear=linear)
        _weights_changed()

    def linear_insert(self, key, val, context=None, update_callback=None, niter=2001, lr=0.05,
                      return_timing=False):
        """Optimise Lambda with weight = W0 + Lambda . context (rewrite/ganrewrite.py:201-252): Adam on
        the (Cout, rank, 3, 3) coefficients only; the weight stays on the rank-r affine subspace."""
        if return_timing:
            torch.cuda.synchronize()
            started = time.time()
        key, val = self.detach(key), self.detach(val)
        self._run_insert(key, val, context, update_callback, niter, 10, lr, linear=True)
        if return_timing:
            torch.cuda.synchronize()
            return (time.time() - started) * 1000


class SeqTinyStyleGanRewriter(SeqStyleGanRewriter):
    def maplayers(self, layernum):
        name = 'layer%d.sconv.mconv.dconv' % layernum
        return name, name


class SeqPreStyleGanRewriter(SeqStyleGanRewriter):
    def maplayers(self, layernum):
        return 'layer%d.sconv.mconv.adain' % layernum, 'layer%d.sconv.activate' % layernum

    def covariance_adjusted_key(self, k, kout):
        assert 'adain' in self.firstlayer
        assert kout.style.shape[0] == 1
        cs = (self.c_matrix * kout.style[0][None, :]).cpu()
        rhs = (k[:, None] if k.dim() == 1 else k.permute(1, 0)).cpu()
        with host_linalg_threads():
            sol = torch.linalg.lstsq(cs, rhs, driver='gels').solution
        sol = sol.to(k.dtype).to(k.device)
        return sol[:, 0] if k.dim() == 1 else sol.permute(1, 0)


# ------------------------------------------------------------------------------------------
# utilities
# ------------------------------------------------------------------------------------------

def _weights_changed():
    from ..utils.stylegan2 import models as sg
    sg.bump_weight_epoch()


def positive_bounding_box(data):
    """(top, left, bottom, right) of the positive entries of a 2-d map; zeros if none."""
    pos = data > 0
    if pos.sum() == 0:
        return 0, 0, 0, 0
    cols = pos.any(0).nonzero()
    rows = pos.any(1).nonzero()
    return rows.min().item(), cols.min().item(), rows.max().item() + 1, cols.max().item() + 1


def centered_location(data):
    t, l, b, r = positive_bounding_box(data)
    return (t + b) // 2, (l + r) // 2


def paste_clip_at_center(source, clip, center, area=None):
    """Copy of ``source`` with ``clip`` pasted (alpha-blended by ``area``) centred at ``center``,
    shifted to stay inside; returns (target, (t, l, b, r))."""
    target = source.clone()
    t, l = (max(0, min(extent - size, c - size // 2))
            for size, c, extent in zip(clip.shape[2:], center, source.shape[2:]))
    b, r = t + clip.shape[2], l + clip.shape[3]
    if area is None:
        target[:, :, t:b, l:r] = clip
    else:
        a = area[None, None, :, :].to(target.device)
        target[:, :, t:b, l:r] = (1 - a) * target[:, :, t:b, l:r] + a * clip
    return target, (t, l, b, r)


def crop_clip_to_bounds(source, target, bounds):
    """Crop key (source) and value (target) maps to the paste bounds, rounding outwards to the
    key grid when the value map has twice its resolution."""
    t, l, b, r = bounds
    vr, hr = [ts // ss for ts, ss in zip(target.shape[2:], source.shape[2:])]
    st, sl, sb, sr = t // vr, l // hr, -(-b // vr), -(-r // hr)
    tt, tl, tb, tr = st * vr, sl * hr, sb * vr, sr * hr
    return (source[:, :, st:sb, sl:sr], target[:, :, tt:tb, tl:tr],
            (st, sl, sb, sr), (tt, tl, tb, tr))


def loss_only(callback):
    """Marks an `update_callback(it, loss)` as one that looks at its two arguments only (a progress bar, a loss
    printout -- metrics/make_watermark_images.py:66-72, rewrite/rewriteapp.py:517-521 are of this kind) and never at
    the model: the solvers then run without interruption and deliver the callbacks afterwards, in order, with the loss
    of every iteration.  An unmarked callback is called after each step and sees the stepped, not yet projected weight,
    as in the reference (rewrite/ganrewrite.py:288-289) -- at one kernel launch per iteration."""
    callback.loss_only = True
    return callback


def projected_conv(weight, direction):
    """P(W)[.., o, i, y, x] = sum_d (sum_j W[.., o, j, y, x] d[d, j]) d[d, i]   (:806-813)"""
    if hip.on_device(weight) and weight.dtype == torch.float32 and weight.dim() in (4, 5) \
            and weight.shape[-1] * weight.shape[-2] <= 9:
        return hip.project_weight(weight, direction).view(weight.shape)
    if weight.dim() == 5:
        cos = torch.einsum('goiyx,di->godyx', weight, direction)
        return torch.einsum('godyx,di->goiyx', cos, direction)
    cos = torch.einsum('oiyx,di->odyx', weight, direction)
    return torch.einsum('odyx,di->oiyx', cos, direction)


def rank_one_conv(weight, direction):
    cos = (weight * direction[None, :, None, None]).sum(1, keepdim=True)
    return cos * direction[None, :, None, None]


class host_linalg_threads:
    """The host LAPACK calls on this path are small (a 512 x 512 eigh, a ~100 x 512 svd, a lstsq): on a
    128-thread box the default OpenMP team makes them slower and leaves spinning workers that starve
    the single-threaded PIL mask rasterisation that follows (measured: 16 ms instead of 0.2 ms per
    mask).  Inside this context torch's intra-op parallelism is capped; restored on exit."""

    def __init__(self, n=8):
        self.n = n

    def __enter__(self):
        self.old = torch.get_num_threads()
        if self.old > self.n:
            torch.set_num_threads(self.n)
        return self

    def __exit__(self, *exc):
        if torch.get_num_threads() != self.old:
            torch.set_num_threads(self.old)


def zca_from_cov(cov):
    """Z = V diag(1/sqrt(lambda)) V^T in float64 -> cov.dtype (:821-826).  LAPACK on the host;
    the reference's symeig read the upper triangle."""
    with host_linalg_threads():
        evals, evecs = torch.linalg.eigh(cov.double().cpu(), UPLO='U')
        zca = evecs @ torch.diag(evals.sqrt().clamp(1e-20).reciprocal()) @ evecs.t()
    return zca.to(cov.dtype).to(cov.device)
