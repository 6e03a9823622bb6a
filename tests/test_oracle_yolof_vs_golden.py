"""The YOLOF oracle (oracle/ref_yolof.py: implicit objectness, ctr-clamped decode, UniformAssigner, focal + DIoU loss) against
the outputs of the reference's own YOLOFHead / UniformAssigner (tests/golden/yolof_baseline.npz, oracle/gen_golden_yolof.py) -
CPU.  Groundwork for the fourth baseline of row N4: there is no YOLOF product yet."""
import os

import numpy as np
import torch

from oracle import ref_retina as RR
from oracle import ref_yolof as Y

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _g():
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'yolof_baseline.npz'))
    return {k: torch.from_numpy(np.asarray(d[k])) for k in d.files}


def test_objectness_decode_and_assigner():
    g = _g()
    torch.testing.assert_close(Y.normalized_cls(g['in_cls'], g['in_obj'], 8), g['out_normalized_cls'], rtol=1e-6, atol=1e-6)
    anchors = RR.grid_anchors([tuple(g['in_cls'].shape[-2:])], [8], [0.5, 1, 2], [1.0])[0]
    torch.testing.assert_close(anchors, g['out_anchors'], rtol=0, atol=0)
    for b in range(2):
        dec = Y.delta2bbox_ctr_clamp(anchors, g['in_reg'][b].permute(1, 2, 0).reshape(-1, 4))
        torch.testing.assert_close(dec, g[f'out_decoded{b}'], rtol=1e-5, atol=1e-5)
        r = Y.uniform_assign(g[f'out_decoded{b}'], anchors, g[f'in_gt_bboxes{b}'], g[f'in_gt_labels{b}'])
        assert torch.equal(r['gt_inds'], g[f'out_assign{b}']) and torch.equal(r['labels'], g[f'out_assign_labels{b}'])
        assert torch.equal(r['pos_idx'], g[f'out_pos_idx{b}'])
        torch.testing.assert_close(r['pos_pred'], g[f'out_pos_pred{b}'], rtol=0, atol=0)
        torch.testing.assert_close(r['target'], g[f'out_pos_target{b}'], rtol=0, atol=0)
        assert int((r['gt_inds'] > 0).sum()) > 0 and int((r['gt_inds'] == -1).sum()) > 0


def test_loss_and_gradients():
    g = _g()
    cs = g['out_normalized_cls'].clone().requires_grad_(True)
    rg = g['in_reg'].clone().requires_grad_(True)
    lc, lb = Y.yolof_loss(cs, rg, [g['in_gt_bboxes0'], g['in_gt_bboxes1']], [g['in_gt_labels0'], g['in_gt_labels1']], 8)
    torch.testing.assert_close(lc, g['out_loss_cls'], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(lb, g['out_loss_bbox'], rtol=1e-5, atol=1e-7)
    gc, gr = torch.autograd.grad(lc + lb, (cs, rg))
    torch.testing.assert_close(gc, g['out_grad_cls'], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(gr, g['out_grad_reg'], rtol=1e-4, atol=1e-8)
