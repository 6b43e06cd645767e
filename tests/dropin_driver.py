"""TEST INFRASTRUCTURE: runs the REFERENCE's own driver code (metrics/make_watermark_images.py main(), and the
rewriter construction + sampling loop of metrics/sample_edited.py) either on this package through
``rewriting_amd.install_reference_aliases()`` (mode "ours": the kernels replaced by tests/hip_emulation.py on CPU,
or the real ones with --device cuda) or on the reference itself through oracle/reference_shim.py (mode
"reference"), and leaves the PNGs the drivers wrote plus a small JSON report in --out.

    python tests/dropin_driver.py --mode ours|reference --out DIR [--device cpu|cuda]

What is NOT the driver's code and is supplied here, identically in both modes: the checkpoint provider
(``load_seq_stylegan`` builds a 64^2 generator with the seeded synthetic weights, there being no network), a
``torchvision.transforms.ToPILImage`` (torchvision is not installed), ``.cuda()`` as the identity on a CPU run,
and request files whose seed indices are folded into the 20-seed sample the watermark run is sized to."""
import argparse
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REFERENCE = os.environ.get('RW_REFERENCE_ROOT', '/root/reference')


def torchvision_stub():
    import numpy
    import PIL.Image
    import torch
    tv = types.ModuleType('torchvision')
    tr = types.ModuleType('torchvision.transforms')
    fn = types.ModuleType('torchvision.transforms.functional')
    md = types.ModuleType('torchvision.models')

    def to_tensor(pic):
        arr = numpy.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(numpy.ascontiguousarray(arr.transpose(2, 0, 1)))
        return t.float().div(255) if t.dtype == torch.uint8 else t

    class ToPILImage:
        def __call__(self, t):
            arr = t.detach().mul(255).byte().permute(1, 2, 0).cpu().numpy()    # torchvision: mul(255).byte()
            return PIL.Image.fromarray(arr)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std
    fn.to_tensor = to_tensor
    tr.functional, tr.Normalize, tr.ToPILImage = fn, Normalize, ToPILImage
    tv.transforms, tv.models = tr, md
    return {'torchvision': tv, 'torchvision.transforms': tr, 'torchvision.transforms.functional': fn,
            'torchvision.models': md}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', required=True, choices=['ours', 'reference'])
    ap.add_argument('--out', required=True)
    ap.add_argument('--device', default='cpu')
    ap.add_argument('--skip-sample-edited', action='store_true')
    opts = ap.parse_args()
    import torch
    torch.set_num_threads(8)
    out = os.path.abspath(opts.out)
    os.makedirs(out, exist_ok=True)
    from rewriting_amd import synthetic

    if opts.mode == 'ours':
        if opts.device == 'cpu':
            import pytest
            from tests import hip_emulation
            hip_emulation.install(pytest.MonkeyPatch())
        sys.modules.update(torchvision_stub())
        import rewriting_amd
        rewriting_amd.install_reference_aliases(reference_root=REFERENCE)
    else:
        from oracle import reference_shim
        reference_shim.load()
        sys.modules.update(torchvision_stub())           # the shim's stub has no working ToPILImage
        sys.path.append(REFERENCE)
    if opts.device == 'cpu':
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self

    import utils.stylegan2 as sg                       # ours (aliased) or the reference's
    from utils.stylegan2.models import SeqStyleGAN2

    def load_seq_stylegan(category, truncation=1.0, **kwargs):
        g = SeqStyleGAN2(64, style_dim=512, n_mlp=8, truncation=truncation, **kwargs)
        synthetic.randomize_(g, seed=0)
        return g.to(opts.device) if opts.device != 'cpu' else g
    sg.load_seq_stylegan = load_seq_stylegan
    report = {'mode': opts.mode}

    # ---- request files with seed indices folded into the 20-seed sample
    reqdir = os.path.join(out, 'masks')
    os.makedirs(os.path.join(reqdir, 'stylegan', 'church'), exist_ok=True)
    with open(os.path.join(REFERENCE, 'notebooks/masks/stylegan/church/multikey_markandbottom.json')) as f:
        req = json.load(f)
    folded = {k: ([[n % 20, m] for n, m in v] if k == 'key' else [v[0] % 20, v[1]]) for k, v in req.items()}
    with open(os.path.join(reqdir, 'stylegan', 'church', 'multikey_markandbottom.json'), 'w') as f:
        json.dump(folded, f)

    os.chdir(REFERENCE)                                 # the drivers copy 'utils/lightbox.html' relative to cwd

    # ---- metrics/make_watermark_images.py, main() as it stands
    import metrics.make_watermark_images as wm          # imports `from utils import pidfile, zdataset, ...`
    wm.load_seq_stylegan = load_seq_stylegan            # `from utils.stylegan2 import load_seq_stylegan` bound at import
    for method, extra in (('ours', ['--nreps', '1']), ('gandissect', [])):
        sys.argv = ['make_watermark_images', '--outdir', os.path.join(out, 'watermark'), '--requestdir', reqdir,
                    '--sample_size', '20', '--layer', '6', '--niters', '11', '--drank', '30', '--rank', '1',
                    '--erasemethod', method] + extra
        wm.main()
    dirs = sorted(os.listdir(os.path.join(out, 'watermark')))
    report['watermark_dirs'] = dirs

    # ---- metrics/sample_edited.py: module-level script; its rewriter construction (lines 38-47) and its
    # sampling loop (lines 53-61) are executed verbatim from the file.  Its apply_edit call (line 51, default
    # niter = 2001) is issued here with niter=11 to bound CPU time.
    if not opts.skip_sample_edited:
        with open(os.path.join(REFERENCE, 'metrics', 'sample_edited.py')) as f:
            lines = f.read().split('\n')
        construct = '\n'.join(lines[37:47])             # zds = ... ; writer = ... ; gw = writer(...)
        loop = '\n'.join(lines[52:61])                  # saver = ... ; for imgnum in tqdm(range(N)) ... ; rd.done()
        assert construct.lstrip().startswith('zds = zdataset.z_dataset_for_model(model, size=1000)'), construct
        assert loop.lstrip().startswith('saver = SaveImagePool()') and 'rd.done()' in loop, loop
        import utils.pidfile
        from utils import zdataset
        from utils.imgsave import SaveImagePool
        from rewrite import ganrewrite
        from torchvision.transforms import ToPILImage
        from tqdm import tqdm
        model = load_seq_stylegan('church', mconv='seq', truncation=0.5)
        model.eval()
        rd = utils.pidfile.reserve_dir(os.path.join(out, 'samples', 'dome2spire'))
        ns = dict(zdataset=zdataset, ganrewrite=ganrewrite, model=model, layernum=8, dataset='church',
                  args=types.SimpleNamespace(full_rank=False, no_tight_paste=False, single_context=-1),
                  SaveImagePool=SaveImagePool, ToPILImage=ToPILImage, tqdm=tqdm, torch=torch, rd=rd, N=4, os=os,
                  json=json)
        os.chdir(out)                                   # cachedir 'results/rewrite/...' is relative
        exec(construct, ns)
        gw = ns['gw']
        with open(os.path.join(REFERENCE, 'notebooks/masks/stylegan/church/dome2spire.json')) as f:
            gw.apply_edit(json.load(f), rank=1, single_key=-1, niter=11)
        exec(loop, ns)
        report['sample_edited_files'] = sorted(os.listdir(os.path.join(out, 'samples', 'dome2spire')))
        report['r2m_cache'] = os.path.isfile(os.path.join(out, 'results/rewrite/stylegan/church/layer8/r2m.npz'))
    with open(os.path.join(out, 'report.json'), 'w') as f:
        json.dump(report, f)
    print('dropin driver done:', json.dumps(report)[:300])


if __name__ == '__main__':
    main()
