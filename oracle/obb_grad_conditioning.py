"""TEST INFRASTRUCTURE ONLY: fp32 conditioning of the oriented iteration (profiles/r03/obb_grad_conditioning.txt): the oracle iteration on
the fixture of tests/golden/ref_iter_obb.npz evaluated in float32 and float64.  python oracle/obb_grad_conditioning.py"""
import sys, torch, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden
import test_reference_iteration_obb as T
from oracle import ref_model as M, ref_obb as O
G = load_golden('ref_iter_obb')
torch.set_num_threads(8)
def run(dtype):
    torch.set_default_dtype(torch.float32)
    sd_s, sd_t = T._states(G)
    img, boxes, labels, _ = T._inputs(G)
    torch.set_default_dtype(dtype)
    cast = lambda d: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in d.items()}
    sd_s, sd_t = cast(sd_s), cast(sd_t)
    img = img.to(dtype); boxes = [b.to(dtype) for b in boxes]
    points = [b[:, :2] for b in boxes]
    out = {}
    for it in range(2):
        inj = T._inject(G, it)
        inj['neg0'] = inj['neg0'].to(dtype)
        if 'syn' in inj:
            inj['syn'] = [{k: (v.to(dtype) if v.is_floating_point() else v) for k, v in d.items()} for d in inj['syn']]
        inj['stats'] = {}
        sd_t = M.ema(sd_t, sd_s)
        params = {k: (v.clone().requires_grad_(True) if O.trainable_obb(k) else v) for k, v in sd_s.items()}
        step = O.forward_train_step1 if it == 0 else O.forward_train_step2
        ref, points = step(params, sd_t, img, boxes, labels, points, dict(O.MODEL_CFG), inj)
        loss = M.total_loss(ref)
        live = [k for k in T.GRAD_KEYS if float(G[f'it{it}_gradnorm_{k}']) > 0]
        gr = dict(zip(live, torch.autograd.grad(loss, [params[k] for k in live], allow_unused=True)))
        out[it] = ({k: float(v) for k, v in ref.items()}, gr)
        points = [p.detach() for p in points]
    return out
a = run(torch.float32)
try:
    b = run(torch.float64)
except Exception as e:
    import traceback; traceback.print_exc(); sys.exit()
for it in range(2):
    print('iteration', it)
    for k in a[it][1]:
        g32, g64 = a[it][1][k].double(), b[it][1][k].double()
        print(f'  {k:45s} fp32 vs fp64 {float((g32-g64).norm()/g64.norm()):.2e}')
    print('  losses max rel', max(abs(a[it][0][k]-b[it][0][k])/max(abs(b[it][0][k]),1e-2) for k in a[it][0]))
