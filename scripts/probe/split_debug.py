"""Debug: image of seed 0 alone vs inside a batch, under the matrix modes and with parts of the split path disabled."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common_checks import build_stylegan          # noqa
from rewriting_amd.utils import zdataset
size = int(os.environ.get('SIZE', '256'))
model = build_stylegan(size, 0.5, device='cuda')
z = zdataset.standard_z_sample(4, 512, seed=1).cuda()
def run(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with torch.no_grad():
            a = model(z); b = model(z[:1])
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    return a, b
ref4, ref1 = run({'RW_MM': 'f32'})
print('f32: batch-of-4 row 0 vs alone', (ref4[:1] - ref1).abs().max().item(), 'range', ref4.abs().max().item())
for env in ({}, {'RW_W4H_PS': '0'}, {'RW_MM_PARTS': 'w4'}, {'RW_MM_PARTS': 'up'}, {'RW_MM_PARTS': 'up1'}, {'RW_MM_PARTS': 'w4,up,up1', 'RW_MM_NO_HANDOVER': '1'}):
    a, b = run(env)
    print(env, 'row0 vs alone', (a[:1] - b).abs().max().item(), '| batch vs f32', (a - ref4).abs().max().item(), '| alone vs f32', (b - ref1).abs().max().item())

# check every hand-over bound against the map it describes
from rewriting_amd.utils.stylegan2 import models as M
orig = M.StyledConvSeq.forward
def checked(self, d):
    out = orig(self, d)
    e = out.get('amax')
    if e is not None and out.fmap is not None:
        true = out.fmap.abs().max().item()
        per = out.fmap.abs().amax(dim=(1, 2, 3)).tolist()
        if e[0].item() != true or os.environ.get('VERBOSE'):
            print('layer %dx%d->%d up=%s: bound %.6g true %.6g per-image %s' % (self.mconv.dconv.in_channel, out.fmap.shape[-1], self.mconv.dconv.out_channel,
                  self.mconv.upsample, e[0].item(), true, ['%.4g' % v for v in per]))
    return out
M.StyledConvSeq.forward = checked
for ps in ('1', '0', '0', '0'):
    os.environ['RW_W4H_PS'] = ps
    print('== RW_W4H_PS', ps)
    with torch.no_grad():
        img = model(z)
    print('   vs f32', (img - ref4).abs().max().item())
