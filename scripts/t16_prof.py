"""Cycle counts from inside the one-workgroup-per-CU form of rw_tconv.hip (library built with -DTC_PROF=1, RW_HIP_LIB): where
waves 0 and 7 of two workgroups in the middle of the launch spend a tile -- the three tap groups' MFMAs, the two window
halves' wait + conversion + LDS write, the chunk barrier, the epilogue.  RW_LAYERS picks the layers."""
import ctypes, json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('RW_TCONV_TY', '16')
from rewriting_amd import hip, _lib  # noqa: E402
DEV = 'cuda:0'
batch = int(os.environ.get('RW_BATCH', '64'))
layers = dict(layer9=(512, 512, 32), layer11=(512, 256, 64), layer13=(256, 128, 128), layer15=(128, 64, 256), layer17=(64, 32, 512))
lib = _lib.load()
lib.rw_tconv_prof.argtypes = [ctypes.c_void_p]
lib.rw_tconv_prof.restype = ctypes.c_int
for name in os.environ.get('RW_LAYERS', 'layer13,layer11').split(','):
    cin, cout, res = layers[name]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(batch, cin, res, res, device=DEV)
    wt = torch.randn(1, cout, cin, 3, 3, generator=g).to(DEV)
    style = (1 + 0.3 * torch.randn(batch, cin, generator=g)).to(DEV)
    s = 1 / math.sqrt(cin * 9)
    dm = hip.demod(hip.weight_sqsum(wt, s), style)
    bias = torch.randn(cout, generator=g).to(DEV)
    noise = torch.randn(batch, 1, 2 * res, 2 * res, device=DEV)
    nw = torch.tensor([0.1], device=DEV)
    k1 = torch.tensor([1., 3., 3., 1.])
    k4 = (k1[:, None] * k1[None, :])
    k4 = (k4 / k4.sum() * 4).to(DEV)
    pk = hip.pack_conv_weight_direct16(wt)
    amax = hip.absmax(x)
    ymax = hip.new_bound(batch * cout * 4 * res * res, DEV)
    args = dict(style=style, demod=dm, noise=noise, noise_w=nw, bias=bias, act=True, x_amax=amax, y_amax=ymax)
    for _ in range(3):
        hip.conv_transpose3x3s2_blur_fused(x, pk, k4, cout, s, **args)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        hip.conv_transpose3x3s2_blur_fused(x, pk, k4, cout, s, **args)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    buf = (ctypes.c_ulonglong * 64)()
    assert lib.rw_tconv_prof(buf) == 0
    keys = ('tile', 'prologue', 'group0', 'half0_wait_convert_store', 'group1', 'group2', 'half1_wait_convert_store', 'chunk_barrier',
            'epilogue', 'chunks')
    for wg, o in (('mid', 0), ('mid+777', 32)):
        for wave, oo in ((0, 0), (7, 16)):
            v = list(buf[o + oo:o + oo + 10])
            print(json.dumps(dict(layer=name, ms=round(ms, 3), wg=wg, wave=wave, **dict(zip(keys, v)))), flush=True)
    del x, noise
    torch.cuda.empty_cache()
