import sys, os, torch
sys.path.insert(0, os.getcwd())
from tests.conftest import *
from tests.test_gpu_model import *
def run():
    from rewriting_amd.rewrite import ganrewrite
    from rewriting_amd.utils import zdataset
    from rewriting_amd.utils.stylegan2.models import DataBag
    g = load_golden('rw_s64_l6_erase'); meta = golden_meta(g)
    model = build_stylegan(meta['size'], meta['truncation'], device=DEV)
    zds = zdataset.z_dataset_for_model(model, size=meta['nseeds'])
    gw = ganrewrite.SeqStyleGanRewriter(model, zds, meta['layernum'], low_rank_insert=True, low_rank_gradient=True)
    req = load_mask_request(meta['mask'], meta['nseeds'])
    with torch.no_grad():
        goal_in, goal_out = gw.erase_from_selection(req['paste'][0], req['paste'][1], req['key'], meta['drank'])
    mkey = torch.from_numpy(g['mkey']).to(DEV)
    gin = DataBag(goal_in, fmap=torch.from_numpy(g['goal_in_fmap']).to(DEV), style=torch.from_numpy(g['goal_in_style']).to(DEV))
    gout = DataBag(goal_out, fmap=torch.from_numpy(g['goal_out_fmap']).to(DEV))
    W0 = gw.target_weights().detach().clone()
    for n in (1, 11):
        gw.target_weights().data.copy_(W0) if hasattr(gw.target_weights(), 'data') else None
        gw.insert(gin, gout, mkey, niter=n, piter=10, lr=0.05)
        dW = (gw.target_weights().detach() - W0)[0]
        if n == 11:
            r = ((torch.einsum('oiyx,di->odyx', dW, mkey).cpu() - torch.from_numpy(g['dW_11_cos'])).norm() / float(g['dW_11_norm'])).item()
            print('lib', os.environ.get('RW_HIP_LIB', 'product'), 'r11', r)
run()
