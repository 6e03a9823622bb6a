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


if __name__ == '__main__' and len(sys.argv) == 1:
    main()


def gen_eval_decode():
    """fcos_head_p2b_ts.py get_bboxes / _get_bboxes (:797-1005) with `with_nms=False` (NMS is an mmcv op): per-level top
    `nms_pre` by score x centerness, distance2bbox with clipping to the image, rescale by 1/scale_factor, the background
    score column -> tests/golden/eval_decode.npz.  The decode half of the Point-Teacher eval path (row a21)."""
    L.install()
    import importlib
    import types
    hd = L.ref('models.dense_heads.fcos_head_p2b_ts')
    md = os.path.join(L.HBB, 'mmdet')
    if 'mmdet.core.export' not in sys.modules:
        ex = L._pkg('mmdet.core.export')
        ex.get_k_for_topk = lambda k, size: int(k) if 0 < int(k) < size else -1      # core/export/onnx_helper.py outside ONNX export
        ex.add_dummy_nms_for_onnx = None
    h = object.__new__(hd.TS_P2BFCOSHead)
    torch.nn.Module.__init__(h)
    h.num_classes = h.cls_out_channels = 8
    h.strides = [8, 16, 32]
    for n in ['get_bboxes']:
        f = getattr(hd.TS_P2BFCOSHead, n)
        if hasattr(f, '__wrapped__'):
            setattr(hd.TS_P2BFCOSHead, n, f.__wrapped__)
    g = torch.Generator().manual_seed(5)
    B, sizes, strides = 2, [(20, 24), (10, 12), (5, 6)], [8, 16, 32]
    cls = [torch.randn(B, 8, hh, ww, generator=g) * 2 - 1 for hh, ww in sizes]
    reg = [(torch.randn(B, 4, hh, ww, generator=g).abs() * 3 + 0.5) * s for (hh, ww), s in zip(sizes, strides)]
    ctr = [torch.randn(B, 1, hh, ww, generator=g) for hh, ww in sizes]
    pts = []
    for (hh, ww), s in zip(sizes, strides):
        ys, xs = torch.meshgrid(torch.arange(hh, dtype=torch.float32), torch.arange(ww, dtype=torch.float32), indexing='ij')
        pts.append(torch.stack((xs.reshape(-1) * s, ys.reshape(-1) * s), dim=-1) + s // 2)
    metas = [dict(img_shape=(150, 180, 3), scale_factor=np.array([1.5, 1.25, 1.5, 1.25], np.float32)),
             dict(img_shape=(160, 192, 3), scale_factor=np.array([1.0, 1.0, 1.0, 1.0], np.float32))]
    cfg = types.SimpleNamespace(score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100)
    cfg.get = lambda k, d=None: dict(nms_pre=60).get(k, d)
    out = {}
    for tag, rescale in (('plain', False), ('rescale', True)):
        res = h.get_bboxes(cls, reg, ctr, pts, metas, cfg=cfg, rescale=rescale, with_nms=False)
        for b in range(B):
            out[f'out_{tag}_bboxes{b}'], out[f'out_{tag}_scores{b}'], out[f'out_{tag}_ctr{b}'] = res[b]
    for i in range(len(sizes)):
        out[f'in_cls{i}'], out[f'in_reg{i}'], out[f'in_ctr{i}'], out[f'in_points{i}'] = cls[i], reg[i], ctr[i], pts[i]
    G.save('eval_decode', **out)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'eval_decode':
    gen_eval_decode()
