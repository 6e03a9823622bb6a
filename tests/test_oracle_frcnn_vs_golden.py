"""The Faster R-CNN oracle (oracle/ref_frcnn.py) against the outputs of the reference's own RPNHead / RandomSampler / BBoxHead
(tests/golden/frcnn_baseline.npz, oracle/gen_golden_frcnn.py) - CPU."""
import os

import numpy as np
import torch

from oracle import ref_frcnn as FR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRIDES, SCALES, RATIOS = [4, 8, 16], [8], [0.5, 1.0, 2.0]
RPN_ASG = dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True)
RCNN_ASG = dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False)


def _g():
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'frcnn_baseline.npz'))
    return {k: torch.from_numpy(np.asarray(d[k])) for k in d.files}


def test_rpn_targets_loss_and_grads():
    g = _g()
    cls = [g[f'in_cls{i}'].clone().requires_grad_(True) for i in range(3)]
    reg = [g[f'in_reg{i}'].clone().requires_grad_(True) for i in range(3)]
    boxes = [g['in_gt_bboxes0'], g['in_gt_bboxes1']]
    feed = FR.PermFeed([g[f'in_rpn_perm{i}'] for i in range(int(g['in_rpn_nperm']))])
    lc, lb, tg, npos, nneg = FR.rpn_loss(cls, reg, boxes, STRIDES, SCALES, RATIOS, RPN_ASG, 256, 0.5, feed)
    assert feed.i == int(g['in_rpn_nperm'])
    assert npos == int(g['out_num_total_pos']) and nneg == int(g['out_num_total_neg'])
    for i in range(3):
        assert torch.equal(tg[i][0], g[f'out_labels{i}']) and torch.equal(tg[i][1], g[f'out_label_weights{i}'])
        torch.testing.assert_close(tg[i][2], g[f'out_bbox_targets{i}'], rtol=1e-5, atol=1e-6)
        assert torch.equal(tg[i][3], g[f'out_bbox_weights{i}'])
        torch.testing.assert_close(lc[i], g[f'out_loss_rpn_cls{i}'], rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(lb[i], g[f'out_loss_rpn_bbox{i}'], rtol=1e-5, atol=1e-7)
    grads = torch.autograd.grad(sum(lc) + sum(lb), cls + reg)
    for i in range(3):
        torch.testing.assert_close(grads[i], g[f'out_grad_cls{i}'], rtol=1e-4, atol=1e-8)
        torch.testing.assert_close(grads[3 + i], g[f'out_grad_reg{i}'], rtol=1e-4, atol=1e-8)


def test_rpn_proposals():
    g = _g()
    H, W = [int(v) for v in g['in_img_hw']]
    for b in range(2):
        p = FR.rpn_proposals([g[f'in_cls{i}'][b] for i in range(3)], [g[f'in_reg{i}'][b] for i in range(3)], (H, W, 3), STRIDES,
                             SCALES, RATIOS, nms_pre=120, max_per_img=90, iou_thr=0.7)
        torch.testing.assert_close(p, g[f'out_proposals{b}'], rtol=1e-5, atol=1e-5)


def test_rcnn_sampling_targets_loss():
    g = _g()
    boxes, labels = [g['in_gt_bboxes0'], g['in_gt_bboxes1']], [g['in_gt_labels0'], g['in_gt_labels1']]
    feed = FR.PermFeed([g[f'in_rcnn_perm{i}'] for i in range(int(g['in_rcnn_nperm']))])
    samples = [FR.rcnn_sample(g[f'in_rcnn_proposals{b}'], boxes[b], labels[b], RCNN_ASG, 64, 0.25, feed) for b in range(2)]
    assert feed.i == int(g['in_rcnn_nperm'])
    for b, s in enumerate(samples):
        assert torch.equal(s['pos_inds'], g[f'out_rcnn_pos_inds{b}']) and torch.equal(s['neg_inds'], g[f'out_rcnn_neg_inds{b}'])
        assert torch.equal(s['pos_assigned'], g[f'out_rcnn_pos_assigned{b}'])
        torch.testing.assert_close(s['bboxes'], g[f'out_rcnn_bboxes{b}'], rtol=0, atol=0)
    torch.testing.assert_close(FR.bbox2roi([s['bboxes'] for s in samples]), g['out_rois'], rtol=0, atol=0)
    lab, lw, bt, bw = FR.rcnn_targets(samples, 8)
    assert torch.equal(lab, g['out_rcnn_labels']) and torch.equal(lw, g['out_rcnn_label_weights']) and torch.equal(bw, g['out_rcnn_bbox_weights'])
    torch.testing.assert_close(bt, g['out_rcnn_bbox_targets'], rtol=1e-5, atol=1e-6)
    cs, bp = g['in_cls_score'].clone().requires_grad_(True), g['in_bbox_pred'].clone().requires_grad_(True)
    lc, lb, acc = FR.rcnn_loss(cs, bp, lab, lw, bt, bw, 8)
    torch.testing.assert_close(lc, g['out_rcnn_loss_cls'], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(lb, g['out_rcnn_loss_bbox'], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(acc.reshape(-1), g['out_rcnn_acc'].reshape(-1), rtol=1e-5, atol=1e-5)
    gc, gb = torch.autograd.grad(lc + lb, (cs, bp))
    torch.testing.assert_close(gc, g['out_rcnn_grad_cls'], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(gb, g['out_rcnn_grad_bbox'], rtol=1e-4, atol=1e-8)
