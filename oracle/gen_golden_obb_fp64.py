"""TEST INFRASTRUCTURE ONLY.  float64 gradients of the oriented iteration for the parameters whose fp32 gradients are ill
conditioned (tests/test_reference_iteration_obb.py ILL_CONDITIONED): the oracle iteration (oracle/ref_obb.py, pinned to the
reference's own `RotatedFCOS_TS.forward_train` by tests/golden/ref_iter_obb.npz) on that fixture's seeds, weights and draws,
evaluated in float64 -> tests/golden/ref_iter_obb_fp64.npz.  The product test compares ITS fp32 gradients with these and requires
them to be no further away than the reference's own fp32 gradients (the golden) are (round-3 verdict P2).
python oracle/gen_golden_obb_fp64.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden  # noqa: E402
import test_reference_iteration_obb as T  # noqa: E402
from oracle import ref_model as M, ref_obb as O  # noqa: E402

G = load_golden('ref_iter_obb')
torch.set_num_threads(8)
dtype = torch.float64
torch.set_default_dtype(torch.float32)
sd_s, sd_t = T._states(G)
img, boxes, labels, _ = T._inputs(G)
torch.set_default_dtype(dtype)
cast = lambda d: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in d.items()}  # noqa: E731
sd_s, sd_t = cast(sd_s), cast(sd_t)
img = img.to(dtype)
boxes = [b.to(dtype) for b in boxes]
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
    live = [k for k in T.ILL_CONDITIONED if float(G[f'it{it}_gradnorm_{k}']) > 0]
    for k, g in zip(live, torch.autograd.grad(loss, [params[k] for k in live], allow_unused=True)):
        out[f'it{it}_grad64_{k}'] = T._sample(g.detach()).numpy()
        out[f'it{it}_gradnorm64_{k}'] = np.float64(g.norm())
        g32 = G.t(f'it{it}_grad_{k}').double()
        print(it, k, 'fp32 reference golden vs float64:', float((g32 - T._sample(g.detach())).norm() / T._sample(g.detach()).norm()))
    for k, v in ref.items():
        out[f'it{it}_loss64_{k}'] = np.float64(v)
    points = [p.detach() for p in points]
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'ref_iter_obb_fp64.npz'), **out)
print('written', len(out), 'arrays')
