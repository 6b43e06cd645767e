"""Cycle counts from inside the specialised direct-16 kernel (library built with -DDC_PROF=1): where the multiplying wave
and the staging wave of workgroups 0 and 100 spend their time."""
import ctypes, json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rewriting_amd import hip, _lib  # noqa: E402
DEV = 'cuda:0'
batch, cin, cout, res = 64, 64, 64, 512
x = torch.randn(batch, cin, res, res, device=DEV)
g = torch.Generator().manual_seed(0)
wt = torch.randn(1, cout, cin, 3, 3, generator=g).to(DEV)
style = (1 + 0.3 * torch.randn(batch, cin, generator=g)).to(DEV)
s = 1 / math.sqrt(cin * 9)
dm = hip.demod(hip.weight_sqsum(wt, s), style)
bias = torch.randn(cout, generator=g).to(DEV)
noise = torch.randn(batch, res * res, device=DEV)
nw = torch.tensor([0.1], device=DEV)
pk = hip.pack_conv_weight_direct16(wt)
amax = hip.absmax(x)
args = dict(style=style, demod=dm, noise=noise, noise_w=nw, bias=bias, act=True, x_amax=amax)
for _ in range(3):
    hip.conv3x3_direct16(x, pk, cout, s, **args)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
lib = _lib.load()
lib.rw_dconv_prof.argtypes = [ctypes.c_void_p]
lib.rw_dconv_prof.restype = ctypes.c_int
assert lib.rw_dconv_prof(buf) == 0
for wg, o in ((0, 0), (100, 16)):
    v = list(buf[o:o + 16])
    print(json.dumps(dict(wg=wg, mfma_wave=dict(total=v[0], compute=v[1], barrier=v[2], epilogue=v[3], chunks=v[4]),
                          staging_wave=dict(total=v[8], deliver=v[9], request=v[10], barrier=v[11], chunks=v[12]))))
