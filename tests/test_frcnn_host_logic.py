"""The Faster R-CNN baseline's host logic (point_teacher_amd/faster_rcnn.py: RandomSampler, RPN targets / loss / proposal
selection, R-CNN sampling / targets / loss) against the outputs of the REFERENCE'S OWN classes
(tests/golden/frcnn_baseline.npz), on the CPU: the four HIP entry points it calls (pt_max_iou_assign, pt_nms_sorted,
pt_delta2bbox; RoIAlign is not on these code paths) are replaced by the oracle's functions for the duration of a test, the
permutations the reference drew are fed to the sampler.  (The kernels themselves are parity-tested in test_hip_ops.py /
test_retina_baseline.py; the assembled detector runs on the GPU in test_frcnn_gpu.py.)"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_frcnn as FR
from oracle import ref_ops as R
from oracle import ref_retina as RR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _g():
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'frcnn_baseline.npz'))
    return {k: torch.from_numpy(np.asarray(d[k])) for k in d.files}


@pytest.fixture
def cpu_kernels(monkeypatch):
    from point_teacher_amd import functional as F

    def max_iou_assign(anchors, gt_boxes, off, B, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, match_low_quality=True, gt_max_assign_all=True):
        inds, mo = [], []
        for b in range(B):
            a, m = RR.max_iou_assign(anchors, gt_boxes[int(off[b]):int(off[b + 1])], pos_iou_thr=pos_iou_thr, neg_iou_thr=neg_iou_thr,
                                     min_pos_iou=min_pos_iou, gt_max_assign_all=gt_max_assign_all, match_low_quality=match_low_quality)
            inds.append(a.int()); mo.append(m)
        return torch.stack(inds), torch.stack(mo)

    def batched_nms(boxes, scores, idxs, nms_cfg):
        keep = R.batched_nms(boxes, scores, idxs, float(nms_cfg['iou_threshold']))
        return torch.cat([boxes[keep], scores[keep, None]], -1), keep
    monkeypatch.setattr(F, 'max_iou_assign', max_iou_assign)
    monkeypatch.setattr(F, 'batched_nms', batched_nms)
    monkeypatch.setattr(F, 'delta2bbox', lambda rois, deltas, max_shape=None, wh_ratio_clip=16 / 1000: FR.delta2bbox(rois, deltas, max_shape=max_shape))
    return F


def _model():
    import point_teacher_amd as pta
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'baselines', 'aitodv2_faster_rcnn_r50_1x.py'))
    m = dict(cfg.model)
    rpn = dict(m['rpn_head'])
    rpn['anchor_generator'] = dict(rpn['anchor_generator'], strides=[4, 8, 16])
    rpn.update(train_cfg=m['train_cfg']['rpn'], test_cfg=dict(nms_pre=120, max_per_img=90, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0))
    roi = dict(m['roi_head'])
    rc = dict(m['train_cfg']['rcnn'])
    rc['sampler'] = dict(rc['sampler'], num=64)
    roi.update(train_cfg=rc, test_cfg=m['test_cfg']['rcnn'])
    return pta.build_from_cfg(rpn, pta.HEADS), pta.build_from_cfg(roi, pta.HEADS)


def test_rpn_loss_matches_reference(cpu_kernels):
    g = _g()
    rpn, _ = _model()
    rpn.sampler.randperm = FR.PermFeed([g[f'in_rpn_perm{i}'] for i in range(int(g['in_rpn_nperm']))])
    cls = [g[f'in_cls{i}'].clone().requires_grad_(True) for i in range(3)]
    reg = [g[f'in_reg{i}'].clone().requires_grad_(True) for i in range(3)]
    boxes = [g['in_gt_bboxes0'], g['in_gt_bboxes1']]
    H, W = [int(v) for v in g['in_img_hw']]
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3)) for _ in boxes]
    losses = rpn.loss(cls, reg, boxes, metas)
    assert rpn.sampler.randperm.i == int(g['in_rpn_nperm'])
    for i in range(3):
        torch.testing.assert_close(losses['loss_rpn_cls'][i], g[f'out_loss_rpn_cls{i}'], rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(losses['loss_rpn_bbox'][i], g[f'out_loss_rpn_bbox{i}'], rtol=1e-5, atol=1e-7)
    grads = torch.autograd.grad(sum(losses['loss_rpn_cls']) + sum(losses['loss_rpn_bbox']), cls + reg)
    for i in range(3):
        torch.testing.assert_close(grads[i], g[f'out_grad_cls{i}'], rtol=1e-4, atol=1e-8)
        torch.testing.assert_close(grads[3 + i], g[f'out_grad_reg{i}'], rtol=1e-4, atol=1e-8)
    # the targets themselves
    rpn.sampler.randperm = FR.PermFeed([g[f'in_rpn_perm{i}'] for i in range(int(g['in_rpn_nperm']))])
    anchors = torch.cat(rpn.anchor_generator.grid_anchors([c.shape[-2:] for c in cls], 'cpu'))
    lab, lw, bt, bw, npos, nneg = rpn.get_targets(anchors, boxes)
    assert npos == int(g['out_num_total_pos']) and nneg == int(g['out_num_total_neg'])
    o = 0
    for i in range(3):
        n = g[f'out_labels{i}'].shape[1]
        assert torch.equal(lab[:, o:o + n], g[f'out_labels{i}']) and torch.equal(lw[:, o:o + n], g[f'out_label_weights{i}'])
        torch.testing.assert_close(bt[:, o:o + n], g[f'out_bbox_targets{i}'], rtol=1e-5, atol=1e-6)
        assert torch.equal(bw[:, o:o + n], g[f'out_bbox_weights{i}'])
        o += n


def test_rpn_proposals_match_reference(cpu_kernels):
    g = _g()
    rpn, _ = _model()
    H, W = [int(v) for v in g['in_img_hw']]
    metas = [dict(img_shape=(H, W, 3)) for _ in range(2)]
    props = rpn.get_bboxes([g[f'in_cls{i}'] for i in range(3)], [g[f'in_reg{i}'] for i in range(3)], metas)
    for b in range(2):
        torch.testing.assert_close(props[b], g[f'out_proposals{b}'], rtol=1e-5, atol=1e-5)


def test_rcnn_sampling_targets_loss_match_reference(cpu_kernels):
    g = _g()
    _, roi = _model()
    roi.bbox_sampler.randperm = FR.PermFeed([g[f'in_rcnn_perm{i}'] for i in range(int(g['in_rcnn_nperm']))])
    boxes, labels = [g['in_gt_bboxes0'], g['in_gt_bboxes1']], [g['in_gt_labels0'], g['in_gt_labels1']]
    results = roi.sample([g['in_rcnn_proposals0'], g['in_rcnn_proposals1']], boxes, labels)
    assert roi.bbox_sampler.randperm.i == int(g['in_rcnn_nperm'])
    for b, r in enumerate(results):
        assert torch.equal(r.pos_inds, g[f'out_rcnn_pos_inds{b}']) and torch.equal(r.neg_inds, g[f'out_rcnn_neg_inds{b}'])
        assert torch.equal(r.pos_assigned_gt_inds, g[f'out_rcnn_pos_assigned{b}'])
        torch.testing.assert_close(r.bboxes, g[f'out_rcnn_bboxes{b}'], rtol=0, atol=0)
    from point_teacher_amd.core import bbox2roi
    rois = bbox2roi([r.bboxes for r in results])
    torch.testing.assert_close(rois, g['out_rois'], rtol=0, atol=0)
    lab, lw, bt, bw = roi.bbox_head.get_targets(results, roi.train_cfg)
    assert torch.equal(lab, g['out_rcnn_labels']) and torch.equal(lw, g['out_rcnn_label_weights']) and torch.equal(bw, g['out_rcnn_bbox_weights'])
    torch.testing.assert_close(bt, g['out_rcnn_bbox_targets'], rtol=1e-5, atol=1e-6)
    cs, bp = g['in_cls_score'].clone().requires_grad_(True), g['in_bbox_pred'].clone().requires_grad_(True)
    ls = roi.bbox_head.loss(cs, bp, rois, lab, lw, bt, bw)
    torch.testing.assert_close(ls['loss_cls'], g['out_rcnn_loss_cls'], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(ls['loss_bbox'], g['out_rcnn_loss_bbox'], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(ls['acc'].reshape(-1), g['out_rcnn_acc'].reshape(-1), rtol=1e-5, atol=1e-5)
    gc, gb = torch.autograd.grad(ls['loss_cls'] + ls['loss_bbox'], (cs, bp))
    torch.testing.assert_close(gc, g['out_rcnn_grad_cls'], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(gb, g['out_rcnn_grad_bbox'], rtol=1e-4, atol=1e-8)


def test_config_builds_with_mmdet_names():
    import point_teacher_amd as pta
    import warnings
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'baselines', 'aitodv2_faster_rcnn_r50_1x.py'))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = pta.build_detector(cfg.model)
    names = {n for n, _ in m.named_parameters()}
    for n in ('rpn_head.rpn_conv.weight', 'rpn_head.rpn_cls.bias', 'rpn_head.rpn_reg.weight', 'roi_head.bbox_head.shared_fcs.0.weight',
              'roi_head.bbox_head.shared_fcs.1.bias', 'roi_head.bbox_head.fc_cls.weight', 'roi_head.bbox_head.fc_reg.bias',
              'neck.lateral_convs.0.conv.weight', 'backbone.layer4.2.conv3.weight'):
        assert n in names, n
    assert m.roi_head.bbox_head.fc_cls.out_features == 9 and m.roi_head.bbox_head.fc_reg.out_features == 32
    assert m.rpn_head.num_anchors == 3 and len(m.roi_head.bbox_roi_extractor.roi_layers) == 4
    assert sum(p.numel() for p in m.parameters()) == 41_384_056


def test_images_without_boxes_and_without_proposals(cpu_kernels):
    """An image without annotations contributes negatives only (RPN and R-CNN); an image whose RPN returns nothing yields an
    empty result; the losses stay finite and differentiable."""
    g = _g()
    rpn, roi = _model()
    torch.manual_seed(0)
    cls = [g[f'in_cls{i}'].clone().requires_grad_(True) for i in range(3)]
    reg = [g[f'in_reg{i}'].clone().requires_grad_(True) for i in range(3)]
    empty = torch.zeros((0, 4))
    boxes, labels = [g['in_gt_bboxes0'], empty], [g['in_gt_labels0'], torch.zeros((0,), dtype=torch.long)]
    H, W = [int(v) for v in g['in_img_hw']]
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=np.ones(4, np.float32)) for _ in boxes]
    losses = rpn.loss(cls, reg, boxes, metas)
    total = sum(losses['loss_rpn_cls']) + sum(losses['loss_rpn_bbox'])
    assert torch.isfinite(total)
    total.backward()
    anchors = torch.cat(rpn.anchor_generator.grid_anchors([c.shape[-2:] for c in cls], 'cpu'))
    lab, lw, bt, bw, npos, nneg = rpn.get_targets(anchors, boxes)
    assert int((lab[1] == 0).sum()) == 0 and float(bw[1].sum()) == 0 and int(lw[1].sum()) == 256      # second image: 256 negatives
    props = rpn.get_bboxes([c.detach() for c in cls], [r.detach() for r in reg], metas)
    results = roi.sample(props, boxes, labels)
    assert results[1].pos_inds.numel() == 0 and results[1].neg_inds.numel() == min(64, props[1].shape[0])
    lab, lw, bt, bw = roi.bbox_head.get_targets(results, roi.train_cfg)
    n = lab.shape[0]
    cs, bp = torch.randn(n, 9, requires_grad=True), torch.randn(n, 32, requires_grad=True)
    from point_teacher_amd.core import bbox2roi
    ls = roi.bbox_head.loss(cs, bp, bbox2roi([r.bboxes for r in results]), lab, lw, bt, bw)
    assert torch.isfinite(ls['loss_cls']) and torch.isfinite(ls['loss_bbox'])
    (ls['loss_cls'] + ls['loss_bbox']).backward()
    # no annotation anywhere: the box loss is the empty sum
    results = roi.sample(props, [empty, empty], [labels[1], labels[1]])
    lab, lw, bt, bw = roi.bbox_head.get_targets(results, roi.train_cfg)
    ls = roi.bbox_head.loss(torch.randn(lab.shape[0], 9), torch.randn(lab.shape[0], 32), None, lab, lw, bt, bw)
    assert float(ls['loss_bbox']) == 0.0 and bool((lab == 8).all())
    # inference with no proposals at all
    out = roi.simple_test([torch.zeros(2, 256, 8, 8)] * 4, [torch.zeros((0, 5)), torch.zeros((0, 5))], metas)
    assert all(d.shape == (0, 5) and l.shape == (0,) for d, l in out)
