#!/usr/bin/env python3
"""Golden vectors for the supervised FCOS baseline - TEST INFRASTRUCTURE ONLY.

    python oracle/gen_golden_fcos.py   ->  tests/golden/fcos_baseline.npz

Runs the REFERENCE's own FCOSHead methods (/root/reference/HBB_TOD/mmdet/models/dense_heads/fcos_head.py: get_targets,
_get_target_single, centerness_target, loss; anchor_free_head.py get_points) on seeded head-shaped tensors, with the
configuration of configs/baselines/aitodv2_fcos_r50_1x.py (center_sampling, norm_on_bbox, DIoULoss) on three levels.
Build container only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader as L  # noqa: E402
import gen_golden as G  # noqa: E402

STRIDES = [8, 16, 32]
RANGES = ((-1, 32), (32, 64), (64, 1e8))
SIZES = [(20, 24), (10, 12), (5, 6)]


def inputs(seed=91, B=3, C=8):
    g = torch.Generator().manual_seed(seed)
    cls = [torch.randn(B, C, h, w, generator=g) * 1.5 - 2.0 for h, w in SIZES]
    reg = [(torch.randn(B, 4, h, w, generator=g) * 0.8 + 1.2).clamp(min=0) for h, w in SIZES]
    ctr = [torch.randn(B, 1, h, w, generator=g) for h, w in SIZES]
    H, W = SIZES[0][0] * 8, SIZES[0][1] * 8
    boxes, labels = [], []
    for n in (14, 0, 9):                                             # the middle image has no object
        c = torch.rand(n, 2, generator=g) * torch.tensor([W - 30.0, H - 30.0]) + 15
        wh = torch.exp(torch.rand(n, 2, generator=g) * (np.log(120.0) - np.log(6.0)) + np.log(6.0))     # 6 .. 120 px: all three levels
        b = torch.cat([c - wh / 2, c + wh / 2], 1)
        b[:, 0::2] = b[:, 0::2].clamp(0, W)
        b[:, 1::2] = b[:, 1::2].clamp(0, H)
        boxes.append(b)
        labels.append(torch.randint(0, C, (n,), generator=g))
    if len(boxes[0]) > 2:                                            # nested boxes: the smallest-area rule decides
        boxes[0][1] = boxes[0][0] + torch.tensor([3.0, 3.0, -3.0, -3.0])
    return cls, reg, ctr, boxes, labels


def main():
    L.install()
    hd = L.ref('models.dense_heads.fcos_head')
    af = L.ref('models.dense_heads.anchor_free_head')
    fl = L.ref('models.losses.focal_loss')
    il = L.ref('models.losses.iou_loss')
    ce = L.ref('models.losses.cross_entropy_loss')

    # a FCOSHead instance without running its constructor (which builds conv layers and needs mmcv's ConvModule): the
    # methods, including the super() calls into AnchorFreeHead, are the reference's own
    h = object.__new__(hd.FCOSHead)
    torch.nn.Module.__init__(h)
    for n in ['loss']:                                # undo the identity-decorated force_fp32 wrapper if one is present
        f = getattr(hd.FCOSHead, n)
        if hasattr(f, '__wrapped__'):
            setattr(hd.FCOSHead, n, f.__wrapped__)
    h.num_classes = h.cls_out_channels = 8
    h.feat_channels = 4
    h.strides, h.regress_ranges = STRIDES, RANGES
    h.center_sampling, h.center_sample_radius, h.norm_on_bbox = True, 1.5, True
    h.loss_cls = fl.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0)
    h.loss_bbox = il.DIoULoss(loss_weight=1.0)
    h.loss_centerness = ce.CrossEntropyLoss(use_sigmoid=True, loss_weight=1.0)
    cls, reg, ctr, boxes, labels = inputs()
    for t in cls + reg + ctr:
        t.requires_grad_(True)
    feats = [torch.zeros(c.shape[0], h.feat_channels, *c.shape[-2:]) for c in cls]
    pts = h.get_points([c.shape[-2:] for c in cls], torch.float32, 'cpu')
    out = {}
    for tag, cs in (('cs', True), ('plain', False)):
        h.center_sampling = cs
        lab, tg = h.get_targets(pts, boxes, labels)
        for i in range(len(SIZES)):
            out[f'out_{tag}_labels{i}'], out[f'out_{tag}_targets{i}'] = lab[i], tg[i]
    h.center_sampling = True
    losses = h.loss(cls, reg, ctr, feats, boxes, labels, [dict()] * cls[0].shape[0])
    total = losses['loss_cls'] + losses['loss_bbox'] + losses['loss_centerness']
    grads = torch.autograd.grad(total, cls + reg + ctr)
    for k, v in losses.items():
        out['out_' + k] = v
    nl = len(SIZES)
    for i in range(nl):
        out[f'in_cls{i}'], out[f'in_reg{i}'], out[f'in_ctr{i}'] = cls[i], reg[i], ctr[i]
        out[f'out_points{i}'] = pts[i]
        out[f'out_grad_cls{i}'], out[f'out_grad_reg{i}'], out[f'out_grad_ctr{i}'] = grads[i], grads[nl + i], grads[2 * nl + i]
    for b in range(len(boxes)):
        out[f'in_gt_bboxes{b}'], out[f'in_gt_labels{b}'] = boxes[b], labels[b]
    # an image set without any object at all: the `else` branch of the loss (:462-464)
    empty = h.loss(cls, reg, ctr, feats, [torch.zeros(0, 4)] * 3, [torch.zeros(0, dtype=torch.long)] * 3, [dict()] * 3)
    for k, v in empty.items():
        out['out_empty_' + k] = v
    G.save('fcos_baseline', **out)


if __name__ == '__main__':
    main()
