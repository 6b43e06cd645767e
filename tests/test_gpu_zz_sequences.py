"""Reproducibility and stress of the un-hooked forward -- collected LAST (the file name sorts behind every other test
file), so that a failure here cannot hide another row's tests behind `pytest -x`.

The reference's forward is a pure function of its inputs (utils/stylegan2/models.py:126-141: an nn.Sequential walk).
Round 4's split-operand path was not: max |x| bounds and the packed weights' 2^-eU went from launch to launch through
4-byte device scalars (memset + atomics + system-scope loads) and were occasionally read stale -- 0.0415 on the image in
GPUTEST_r04.  Round 5 removed the mechanism (per-wave slots stored plainly + one reduction launch + per-lane vector
loads; weight scales by value; no stream drain): these tests hold the forward to the reference's contract."""
import os

import pytest
import torch

from tests.conftest import build_stylegan, load_golden, golden_meta

pytestmark = pytest.mark.gpu
DEV = 'cuda'

CONFIGS = [{}, {'RW_PRESCALE': '0'}, {'RW_PRESCALE': '0', 'RW_UP_FUSED': '0', 'RW_RGB_F4': '0'},
           {'RW_PRESCALE': '0', 'RW_UP_FUSED': '0', 'RW_RGB_F4': '0', 'RW_CONV_ALGO': 'winograd'}]


def _run(model, z, monkeypatch, sync, configs=CONFIGS):
    outs = []
    with torch.no_grad():
        for env in configs:
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            outs.append(model(z))
            if sync:
                torch.cuda.synchronize()
            for k in env:
                monkeypatch.delenv(k)
    torch.cuda.synchronize()
    return outs


@pytest.mark.parametrize('direct16', [pytest.param(None, id='default'), pytest.param('0', id='f4x4-everywhere'),
                                      pytest.param('1', id='direct16-all'),
                                      pytest.param('unfused-up', id='two-pass-upsampling')])
@pytest.mark.parametrize('size,batch,reps', [(256, 4, 200), (1024, 2, 200)])
def test_forwards_issued_back_to_back_equal_their_synced_twins(monkeypatch, size, batch, reps, direct16):
    """200 sequences of four differently configured forwards, issued WITHOUT a host sync in between (a forward starts
    while the previous one still runs, the RGB branch on its second stream), against the same four with a sync after
    each: bit-identical, at 256^2 and 1024^2, for the default kernel selection (direct sums on layers 10 - 17), with the
    split F(4x4,3x3) kernels everywhere (RW_MM_DIRECT16=0), with direct sums also on the last layer (=1), and without the
    fused upsampling kernel (RW_UP_FUSED2=0: the routes that were the default before it).  The default INCLUDES that kernel --
    the one whose first forms' neighbours on the RGB stream came back wrong until the streaming kernels lost their packed
    fp32 FMAs: its sequences are held to the same bit-identity."""
    if direct16 == 'unfused-up':      # without csrc/rw_tconv.hip's kernel: F(2,2) transposed convolutions + blur passes, phase kernels
        monkeypatch.setenv('RW_UP_FUSED2', '0')
    elif direct16:
        monkeypatch.setenv('RW_MM_DIRECT16', direct16)
    assert 'RW_FORWARD_DRAIN' not in os.environ          # the round-4 workaround is gone, not switched off
    model = build_stylegan(size, 0.7, device=DEV)
    z = torch.randn(batch, 512, generator=torch.Generator().manual_seed(9)).to(DEV)
    ref = _run(model, z, monkeypatch, True)
    again = _run(model, z, monkeypatch, True)
    for i, (a, b) in enumerate(zip(again, ref)):         # the very first forwards (they pack the weights) are not special
        assert torch.equal(a, b), ('synced twice', i, (a - b).abs().max().item())
    for rep in range(reps):
        got = _run(model, z, monkeypatch, False)
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.equal(a, b), (rep, i, (a - b).abs().max().item())


@pytest.mark.parametrize('name', ['gen_s256_full', 'gen_s1024_full'])
def test_edit_repack_first_forward_equals_second_and_the_reference(name):
    """What the rule-editing use case does: rewrite a layer's weight -> the derived (packed, split) weights are rebuilt ->
    render.  The FIRST forward after a re-pack equals the second bit for bit, an un-synced one equals both, and with the
    original weights restored the image is the reference fixture's again.  (Round 4: the first forward after a pack
    occasionally read a stale 2^-eU behind the packed weights.)"""
    g = load_golden(name)
    meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'], meta['channel_multiplier'], device=DEV)
    z = torch.from_numpy(g['z']).to(DEV)
    layers = [n for n, _ in model.named_children() if n.startswith('layer')]
    targets = [getattr(model, n).sconv.mconv.dconv for n in layers[-6:]]       # the split-operand layers
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        base = model(z)
        torch.cuda.synchronize()
        for rep in range(6):
            dconv = targets[rep % len(targets)]
            saved = dconv.weight.detach().clone()
            # an edit that moves max |U| across binades: the by-value scale must follow the weight version
            dconv.weight.mul_(float(2.0 ** (rep - 2)))
            dconv.weight.add_(0.05 * torch.randn(dconv.weight.shape, generator=gen).to(DEV))
            first = model(z)                     # re-packs, then renders -- no sync in between
            second = model(z)
            torch.cuda.synchronize()
            third = model(z)
            assert torch.isfinite(first).all()
            assert torch.equal(first, second), (rep, (first - second).abs().max().item())
            assert torch.equal(first, third), (rep, (first - third).abs().max().item())
            assert not torch.equal(first, base)
            dconv.weight.copy_(saved)
            back = model(z)
            assert torch.equal(back, base), (rep, (back - base).abs().max().item())
    from tests.test_gpu_fullsize import check_image_digest
    check_image_digest(g, 'image/', base)            # ... and that image is the reference's (1e-3 L-inf, 1e-4 of the range)


def test_a_fresh_process_packs_and_renders_the_same_image_every_time():
    """Twelve fresh models (each packs its split weights in its first forward, scales by value) render the same image:
    the pack path has no run-to-run state."""
    z = torch.randn(3, 512, generator=torch.Generator().manual_seed(5)).to(DEV)
    ref = None
    for rep in range(12):
        model = build_stylegan(256, 0.7, device=DEV)
        with torch.no_grad():
            img = model(z)
        if ref is None:
            ref = img
        assert torch.equal(img, ref), (rep, (img - ref).abs().max().item())
