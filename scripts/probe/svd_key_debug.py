import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, numpy
from tests.common_checks import load_golden, golden_meta, _rewriter, load_mask_request
from rewriting_amd.rewrite import ganrewrite
g = load_golden('rw_s64_l8_extras'); sc = load_golden('rw_s64_l8_keyscatter')
meta = golden_meta(g)
gw = _rewriter(meta, 'cuda')
keys = load_mask_request(meta['mask'], meta['nseeds'])['key']
Cx = torch.from_numpy(sc['c_exact']).double()
def pc(a, b, weight=None):
    a, b = a.double().cpu().t(), b.double().cpu().t()
    if weight is not None: a, b = weight @ a, weight @ b
    return torch.linalg.svdvals(torch.linalg.qr(a)[0].t() @ torch.linalg.qr(b)[0])
got = gw.multi_key_from_selection(keys, rank=4, key_method='svd')
ex = torch.from_numpy(sc['svd_exact']); ref = torch.from_numpy(g['mkey_svd'])
print('C gpu vs exact rel', ((gw.c_matrix.double().cpu()-Cx).norm()/Cx.norm()).item())
print('ours r2 vs exact thru C', pc(got[:2], ex, Cx), 'raw', pc(got[:2], ex))
print('ours r2 vs refgolden thru C', pc(got[:2], ref, Cx))
print('refgolden vs exact thru C', pc(ref, ex, Cx))
print('ours r4 vs exact r2 thru C', pc(got[:4], ex, Cx))
print('lead: ours vs exact thru C', pc(got[:1], ex[:1], Cx), 'ours2 vs exact2', pc(got[1:2], ex[1:2], Cx))
# singular values
gathered = []
from rewriting_amd.utils import renormalize
for imgnum, mask in keys:
    k_outs = gw.context_model(gw.get_z(imgnum)); k_acts = gw.context_acts(k_outs)
    area = renormalize.from_url(mask, target='pt', size=gw.k_shape[2:])[0]
    w = (k_acts[0] * area[None].to(gw.device)).permute(1, 2, 0).reshape(-1, k_acts.shape[1])
    gathered.append(w[w.norm(2, dim=1) > 0])
rows = torch.cat(gathered)
all_k = gw.covariance_adjusted_query_key(rows)
print('rows', rows.shape, 'sv fp32 lstsq', torch.linalg.svdvals(all_k.double().cpu())[:5])
all_k64 = torch.linalg.solve(Cx, rows.double().cpu().t()).t()
print('sv exact C fp64 solve', torch.linalg.svdvals(all_k64)[:5])
u = torch.linalg.svd(all_k64.t(), full_matrices=False)[0][:, :2].t()
print('fp64 solve of GPU rows vs exact thru C', pc(u, ex, Cx))
