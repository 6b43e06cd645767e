"""Sample-set generation and Frechet statistics -- the "next" rows SURVEY.md section 8f.3 marks
after the hot path: sharded per-seed generator passes (metrics/sample.py:32-37,
metrics/sample_edited.py:55-59) and the reduction + distance of metrics/fid.py:40-62,137-187.
The feature extractor stays pluggable (the reference's TF Inception graph is downloaded at run
time and is unavailable here): any callable images -> (B, F) features.
"""
import warnings

import numpy
import torch

from . import hip, parallel
from .utils import zdataset
from .utils.stylegan2.models import noise_batch_period


def seed_latents(model, seeds, offset=0):
    """The reference draws image n from ``z_sample_for_model(model, size=1, seed=n + offset)``
    (metrics/sample.py:33): one independent stream per image."""
    return torch.cat([zdataset.z_sample_for_model(model, size=1, seed=int(s) + offset) for s in seeds])


def generate(model_fn, model, seeds, batch=32, offset=0, device=None, shard=None):
    """Yields (seed list, images) with seed i handled by rank i mod world (no collective).  The
    reference generates these sets at batch 1, where every image takes noise row 0 (quirk Q1);
    ``noise_batch_period(1)`` keeps that row for every image of a larger launch."""
    shard = parallel.shard() if shard is None else shard
    seeds = list(seeds)
    if shard is not None:
        seeds = seeds[shard[0]::shard[1]]
    device = device or next(model.parameters()).device
    with torch.no_grad(), noise_batch_period(1):
        for i in range(0, len(seeds), batch):
            chunk = seeds[i:i + batch]
            z = seed_latents(model, chunk, offset).to(device)
            yield chunk, model_fn(z)


class FeatureStatistics:
    """Running sum f and sum f f^T of feature rows, kept in float64; ``allreduce_`` pools the raw sums
    of all ranks with one RCCL all-reduce (F=2048: 33.5 MB)."""

    def __init__(self, dim=None):
        self.count = 0
        self.sum = None
        self.outer = None
        self._dim = dim

    def add(self, feats):
        feats = feats.detach()
        if feats.dim() != 2:
            feats = feats.reshape(feats.shape[0], -1)
        f = feats.shape[1]
        if self.sum is None:
            self.sum = torch.zeros(f, dtype=torch.float64, device=feats.device)
            self.outer = torch.zeros(f, f, dtype=torch.float64, device=feats.device)
        if hip.on_device(feats) and feats.dtype == torch.float32 and f % 4 == 0:
            block = torch.zeros(f, f, dtype=torch.float32, device=feats.device)
            hip.second_moment_accumulate(block, feats.contiguous(), nchw=False)    # fp32 MFMA per batch
            self.outer += block.double()                                            # fp64 across batches
        else:
            d = feats.double()
            self.outer += d.t() @ d
        self.sum += feats.double().sum(0)
        self.count += feats.shape[0]

    def allreduce_(self):
        import torch.distributed as dist
        if parallel.shard() is None:
            return self
        dev = self.sum.device
        if dist.get_backend() == 'gloo':
            dev = torch.device('cpu')
        packed = torch.cat([self.sum.to(dev), self.outer.reshape(-1).to(dev),
                            torch.tensor([float(self.count)], dtype=torch.float64, device=dev)])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        f = self.sum.numel()
        self.sum = packed[:f].to(self.sum.device)
        self.outer = packed[f:f + f * f].reshape(f, f).to(self.outer.device)
        self.count = int(round(packed[-1].item()))
        return self

    def mean_cov(self):
        """(mu, sigma) as numpy float64, sigma unbiased like numpy.cov(act, rowvar=False)
        (metrics/fid.py:60-61)."""
        n = self.count
        mu = self.sum / n
        sigma = (self.outer - n * torch.outer(mu, mu)) / (n - 1)
        return mu.cpu().numpy(), sigma.cpu().numpy()


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """d^2 = |mu1-mu2|^2 + Tr(S1 + S2 - 2 sqrt(S1 S2)) with the reference's singular-product retry and
    imaginary-part check (metrics/fid.py:137-187)."""
    from scipy import linalg
    mu1, mu2 = numpy.atleast_1d(mu1), numpy.atleast_1d(mu2)
    sigma1, sigma2 = numpy.atleast_2d(sigma1), numpy.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not numpy.isfinite(covmean).all():
        warnings.warn('fid calculation produces singular product; adding %s to diagonal of cov estimates' % eps)
        offset = numpy.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if numpy.iscomplexobj(covmean):
        if not numpy.allclose(numpy.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError('Imaginary component {}'.format(numpy.max(numpy.abs(covmean.imag))))
        covmean = covmean.real
    return diff.dot(diff) + numpy.trace(sigma1) + numpy.trace(sigma2) - 2 * numpy.trace(covmean)
