#!/usr/bin/env python3
"""Golden vectors for the YOLOF baseline's training-side logic (fourth entry: groundwork for the product) -
TEST INFRASTRUCTURE ONLY.

    python oracle/gen_golden_yolof.py   ->  tests/golden/yolof_baseline.npz

Runs the REFERENCE's own classes, loaded by path (/root/reference/HBB_TOD/mmdet), with the configuration of
configs/baselines/aitodv2_yolof_r50_1x.py on seeded head-shaped tensors:
  * core/bbox/assigners/uniform_assigner.py UniformAssigner(pos_ignore_thr .15, neg_ignore_thr .7, match_times 4).assign on
    decoded predictions + anchors (the L1 cdist top-k of predictions AND of anchors, the two ignore rules, the duplicate-index
    scatter whose last write wins);
  * models/dense_heads/yolof_head.py YOLOFHead.loss (:136-222) with get_targets / _get_targets_single (:223-414),
    DeltaXYWHBBoxCoder(add_ctr_clamp, ctr_clamp 32).decode, FocalLoss and DIoULoss on the decoded boxes, and the implicit-
    objectness normalisation of forward_single (:118-134) on seeded cls / objectness logits.
Build container only."""
import importlib
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader as L  # noqa: E402
import gen_golden as G  # noqa: E402
import gen_golden_retina as GR  # noqa: E402

SIZE = (10, 12)            # one level (the head asserts it), stride 8 as the config says
A, C = 3, 8


def install():
    ag, mi, ps, ah, rh = GR.install()
    imp = importlib.import_module
    core = sys.modules['mmdet.core']
    core.reduce_mean = lambda t: t
    cnn = sys.modules['mmcv.cnn']
    for n in ('bias_init_with_prob', 'constant_init', 'is_norm', 'normal_init'):
        setattr(cnn, n, None)                          # only the (skipped) constructor / init_weights call them
    ua = imp('mmdet.core.bbox.assigners.uniform_assigner')
    yh = imp('mmdet.models.dense_heads.yolof_head')
    return ag, ps, ua, yh


def inputs(seed=37, B=2):
    g = torch.Generator().manual_seed(seed)
    H, W = SIZE
    cls = torch.randn(B, A * C, H, W, generator=g) * 1.2 - 2.0
    obj = torch.randn(B, A, H, W, generator=g) * 1.0 - 1.0
    reg = torch.randn(B, A * 4, H, W, generator=g) * 0.4
    boxes, labels = [], []
    for n in (6, 3):
        c = torch.rand(n, 2, generator=g) * torch.tensor([W * 8 - 20.0, H * 8 - 20.0]) + 10
        wh = torch.exp(torch.rand(n, 2, generator=g) * (np.log(40.0) - np.log(6.0)) + np.log(6.0))
        boxes.append(torch.cat([c - wh / 2, c + wh / 2], 1))
        labels.append(torch.randint(0, C, (n,), generator=g))
    return cls, obj, reg, boxes, labels, (H * 8, W * 8)


def stable_topk(x, k, dim=-1, largest=True, sorted=True):
    """`torch.topk` may return tied elements in any order, and the UniformAssigner's anchor costs tie STRUCTURALLY: the 8- and the
    16-pixel anchor of one cell are equally far (L1 over cx, cy, w, h) from every box whose sides lie between 8 and 16 px.  The
    order decides which of the duplicate writes of `assigned_gt_inds[indexes] = ...` comes last (uniform_assigner.py:113-131), so the
    assignment is not a function of its inputs (CPU and GPU kernels disagree).  Served here by a stable top-k, lowest index first
    among ties - one of the results torch may legally return, and the rule the oracle and the product implement."""
    v, i = torch.sort(x, dim=dim, descending=largest, stable=True)
    return v.narrow(dim, 0, k), i.narrow(dim, 0, k)


def main():
    ag, ps, ua, yh = install()
    ua.torch = types.SimpleNamespace(**{n: getattr(torch, n) for n in dir(torch) if not n.startswith('__')})
    ua.torch.topk = stable_topk
    dc = L.ref('core.bbox.coder.delta_xywh_bbox_coder')
    fl = L.ref('models.losses.focal_loss')
    il = L.ref('models.losses.iou_loss')
    f = yh.YOLOFHead.loss
    if hasattr(f, '__wrapped__'):
        yh.YOLOFHead.loss = f.__wrapped__
    out = {}
    cls, obj, reg, boxes, labels, (H, W) = inputs()
    h = object.__new__(yh.YOLOFHead)
    torch.nn.Module.__init__(h)
    h.num_classes = h.cls_out_channels = C
    h.use_sigmoid_cls, h.sampling, h.reg_decoded_bbox = True, False, True
    h.rf_based, h.decay = False, 1
    h.anchor_generator = ag.AnchorGenerator(ratios=[1.0], scales=[0.5, 1, 2], strides=[8])
    h.num_anchors = h.anchor_generator.num_base_anchors[0]
    h.assigner = ua.UniformAssigner(pos_ignore_thr=0.15, neg_ignore_thr=0.7)
    h.sampler = ps.PseudoSampler()
    h.bbox_coder = dc.DeltaXYWHBBoxCoder(target_means=[.0, .0, .0, .0], target_stds=[1., 1., 1., 1.], add_ctr_clamp=True, ctr_clamp=32)
    h.loss_cls = fl.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0)
    h.loss_bbox = il.DIoULoss(loss_weight=1.0)
    h.train_cfg = types.SimpleNamespace(allowed_border=-1, pos_weight=-1, debug=False)
    # forward_single's implicit objectness (:124-134) on the seeded logits
    N = cls.shape[0]
    cs = cls.view(N, -1, C, *SIZE)
    ob = obj.view(N, -1, 1, *SIZE)
    norm = (cs + ob - torch.log(1. + torch.clamp(cs.exp(), max=yh.INF) + torch.clamp(ob.exp(), max=yh.INF))).view(N, -1, *SIZE)
    out['in_cls'], out['in_obj'], out['in_reg'], out['out_normalized_cls'] = cls, obj, reg, norm
    ncls = norm.detach().clone().requires_grad_(True)
    reg = reg.clone().requires_grad_(True)
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3)) for _ in boxes]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        losses = h.loss([ncls], [reg], boxes, labels, metas)
        gc, gr = torch.autograd.grad(losses['loss_cls'] + losses['loss_bbox'], (ncls, reg))
        # the assigner alone, per image, on the decoded predictions
        anchors = h.anchor_generator.grid_anchors([SIZE], device='cpu')[0]
        for b in range(len(boxes)):
            bp = reg.detach()[b].permute(1, 2, 0).reshape(-1, 4)
            dec = h.bbox_coder.decode(anchors, bp)
            r = h.assigner.assign(dec, anchors, boxes[b], None, labels[b])
            out[f'out_decoded{b}'] = dec
            out[f'out_assign{b}'], out[f'out_assign_labels{b}'] = r.gt_inds, r.labels
            out[f'out_pos_idx{b}'] = r.get_extra_property('pos_idx')
            out[f'out_pos_pred{b}'], out[f'out_pos_target{b}'] = r.get_extra_property('pos_predicted_boxes'), r.get_extra_property('target_boxes')
            out[f'in_gt_bboxes{b}'], out[f'in_gt_labels{b}'] = boxes[b], labels[b]
    out['out_anchors'] = anchors
    out['out_loss_cls'], out['out_loss_bbox'] = losses['loss_cls'], losses['loss_bbox']
    out['out_grad_cls'], out['out_grad_reg'] = gc, gr
    out['in_img_hw'] = np.array([H, W])
    G.save('yolof_baseline', **out)


if __name__ == '__main__':
    main()
