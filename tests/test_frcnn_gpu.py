"""The Faster R-CNN baseline on the GPU (-m gpu): the same comparisons as tests/test_frcnn_host_logic.py, now through the real
kernels (pt_max_iou_assign, pt_nms_sorted, pt_delta2bbox) against the outputs of the reference's own RPNHead / RandomSampler /
BBoxHead (tests/golden/frcnn_baseline.npz), then training iterations and inference of the baseline config (multi-level
RoIAlign, two-stage losses through the flat-parameter Trainer)."""
import os

import numpy as np
import pytest
import torch

import test_frcnn_host_logic as H
from oracle import ref_frcnn as FR

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rpn_and_rcnn_vs_reference_on_gpu():
    dev = torch.device('cuda:0')
    g = {k: v.to(dev) for k, v in H._g().items()}
    rpn, roi = H._model()
    rpn, roi = rpn.to(dev), roi.to(dev)
    cpu = lambda k: g[k].cpu()
    # ---- RPN loss, gradients, proposals
    rpn.sampler.randperm = FR.PermFeed([cpu(f'in_rpn_perm{i}') for i in range(int(g['in_rpn_nperm']))])
    cls = [g[f'in_cls{i}'].clone().requires_grad_(True) for i in range(3)]
    reg = [g[f'in_reg{i}'].clone().requires_grad_(True) for i in range(3)]
    boxes, labels = [g['in_gt_bboxes0'], g['in_gt_bboxes1']], [g['in_gt_labels0'], g['in_gt_labels1']]
    Hh, Ww = [int(v) for v in g['in_img_hw']]
    metas = [dict(img_shape=(Hh, Ww, 3), pad_shape=(Hh, Ww, 3)) for _ in boxes]
    losses = rpn.loss(cls, reg, boxes, metas)
    assert rpn.sampler.randperm.i == int(g['in_rpn_nperm'])
    for i in range(3):
        torch.testing.assert_close(losses['loss_rpn_cls'][i], g[f'out_loss_rpn_cls{i}'], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(losses['loss_rpn_bbox'][i], g[f'out_loss_rpn_bbox{i}'], rtol=1e-4, atol=1e-6)
    grads = torch.autograd.grad(sum(losses['loss_rpn_cls']) + sum(losses['loss_rpn_bbox']), cls + reg)
    for i in range(3):
        torch.testing.assert_close(grads[i], g[f'out_grad_cls{i}'], rtol=1e-3, atol=1e-7)
        torch.testing.assert_close(grads[3 + i], g[f'out_grad_reg{i}'], rtol=1e-3, atol=1e-7)
    props = rpn.get_bboxes([g[f'in_cls{i}'] for i in range(3)], [g[f'in_reg{i}'] for i in range(3)], metas)
    for b in range(2):
        assert props[b].shape == g[f'out_proposals{b}'].shape
        torch.testing.assert_close(props[b], g[f'out_proposals{b}'], rtol=1e-4, atol=1e-3)
    # ---- R-CNN stage
    roi.bbox_sampler.randperm = FR.PermFeed([cpu(f'in_rcnn_perm{i}') for i in range(int(g['in_rcnn_nperm']))])
    results = roi.sample([g['in_rcnn_proposals0'], g['in_rcnn_proposals1']], boxes, labels)
    for b, r in enumerate(results):
        assert torch.equal(r.pos_inds, g[f'out_rcnn_pos_inds{b}']) and torch.equal(r.neg_inds, g[f'out_rcnn_neg_inds{b}'])
        assert torch.equal(r.pos_assigned_gt_inds, g[f'out_rcnn_pos_assigned{b}'])
    from point_teacher_amd.core import bbox2roi
    rois = bbox2roi([r.bboxes for r in results])
    torch.testing.assert_close(rois, g['out_rois'], rtol=0, atol=0)
    lab, lw, bt, bw = roi.bbox_head.get_targets(results, roi.train_cfg)
    assert torch.equal(lab, g['out_rcnn_labels']) and torch.equal(lw, g['out_rcnn_label_weights'])
    torch.testing.assert_close(bt, g['out_rcnn_bbox_targets'], rtol=1e-4, atol=1e-5)
    ls = roi.bbox_head.loss(g['in_cls_score'], g['in_bbox_pred'], rois, lab, lw, bt, bw)
    torch.testing.assert_close(ls['loss_cls'], g['out_rcnn_loss_cls'], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(ls['loss_bbox'], g['out_rcnn_loss_bbox'], rtol=1e-4, atol=1e-6)


def test_faster_rcnn_trains_and_infers():
    import point_teacher_amd as pta
    import test_train_step_parity as T
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'baselines', 'aitodv2_faster_rcnn_r50_1x.py'))
    model = pta.build_detector(cfg.model).to(dev)
    model.train()
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config)
    img, boxes, labels, metas = T._data(dev)
    img = (img - 110.0) / 58.0                                             # the config normalises its input
    data = dict(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in boxes], gt_labels=[l.to(dev) for l in labels])
    hist = []
    for it in range(6):
        lv = trainer.step(data)['log_vars'].materialize()
        assert {'loss_rpn_cls', 'loss_rpn_bbox', 'loss_cls', 'loss_bbox', 'acc', 'loss'} <= set(lv), lv
        assert all(np.isfinite(v) for v in lv.values()), lv
        hist.append(lv['loss'])
    assert min(hist[2:]) < hist[0], hist            # (the sampled RoIs change from iteration to iteration: not monotonic)
    gs = dict(model.named_parameters())
    for n in ('rpn_head.rpn_conv.weight', 'roi_head.bbox_head.shared_fcs.0.weight', 'roi_head.bbox_head.fc_reg.weight',
              'neck.lateral_convs.0.conv.weight', 'backbone.layer2.0.conv1.weight'):
        assert gs[n].grad is not None and float(gs[n].grad.abs().sum()) > 0, n
    model.eval()
    with torch.no_grad():
        res = model(return_loss=False, rescale=True, img=[data['img']], img_metas=[metas])
    assert len(res) == 2 and len(res[0]) == 8 and all(r.shape[1] == 5 for r in res[0])
