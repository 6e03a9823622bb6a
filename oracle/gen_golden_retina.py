#!/usr/bin/env python3
"""Golden vectors for the RetinaNet baseline (second entry) - TEST INFRASTRUCTURE ONLY.

    python oracle/gen_golden_retina.py   ->  tests/golden/retina_baseline.npz

Runs the REFERENCE's own classes, loaded by path (/root/reference/HBB_TOD/mmdet): core/anchor/anchor_generator.py
(AnchorGenerator.grid_anchors / valid_flags), core/bbox/assigners/max_iou_assigner.py (MaxIoUAssigner.assign, both
`gt_max_assign_all` modes and a (lo, hi) negative range), core/bbox/samplers/pseudo_sampler.py, core/bbox/coder
(bbox2delta) and models/dense_heads/anchor_head.py (get_targets, loss) through a RetinaHead instance whose constructor is
skipped (it builds conv layers), on seeded head-shaped tensors with the configuration of
configs/baselines/aitodv2_retinanet_r50_1x.py on three levels.  Build container only."""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader as L  # noqa: E402
import gen_golden as G  # noqa: E402

STRIDES = [8, 16, 32]
SIZES = [(20, 24), (10, 12), (5, 6)]


def install():
    L.install()
    imp = importlib.import_module
    md = os.path.join(L.HBB, 'mmdet')
    core = sys.modules['mmdet.core']
    sys.modules['mmcv'].is_tuple_of = lambda seq, typ: isinstance(seq, tuple) and all(isinstance(x, typ) for x in seq)
    L._pkg('mmdet.core.anchor', os.path.join(md, 'core', 'anchor'))
    ab = imp('mmdet.core.anchor.builder')
    ag = imp('mmdet.core.anchor.anchor_generator')
    au = imp('mmdet.core.anchor.utils')
    L._pkg('mmdet.core.bbox.samplers', os.path.join(md, 'core', 'bbox', 'samplers'))
    imp('mmdet.core.bbox.samplers.sampling_result')
    imp('mmdet.core.bbox.samplers.base_sampler')
    ps = imp('mmdet.core.bbox.samplers.pseudo_sampler')
    mi = imp('mmdet.core.bbox.assigners.max_iou_assigner')
    misc = imp('mmdet.core.utils.misc')
    bld = imp('mmdet.core.bbox.builder')
    core.anchor_inside_flags, core.images_to_levels = au.anchor_inside_flags, au.images_to_levels
    core.build_anchor_generator, core.build_prior_generator = ab.build_anchor_generator, ab.build_prior_generator
    core.build_sampler, core.build_bbox_coder, core.build_assigner = bld.build_sampler, bld.build_bbox_coder, bld.build_assigner
    core.unmap, core.multiclass_nms = misc.unmap, None
    for n in ('mmdet.models.dense_heads.base_dense_head', 'mmdet.models.dense_heads.dense_test_mixins'):
        if n not in sys.modules:
            try:
                imp(n)
            except Exception:
                stub = L._pkg(n, stub=True)
                stub.BaseDenseHead = torch.nn.Module
                stub.BBoxTestMixin = object
    ah = imp('mmdet.models.dense_heads.anchor_head')
    rh = imp('mmdet.models.dense_heads.retina_head')
    return ag, mi, ps, ah, rh


def inputs(seed=17, B=3, C=8, A=9):
    """(for B != 3 only the head-shaped tensors are meaningful)"""
    g = torch.Generator().manual_seed(seed)
    cls = [torch.randn(B, A * C, h, w, generator=g) * 1.5 - 2.0 for h, w in SIZES]
    reg = [torch.randn(B, A * 4, h, w, generator=g) * 0.3 for h, w in SIZES]
    H, W = SIZES[0][0] * 8, SIZES[0][1] * 8
    boxes, labels = [], []
    for n in (12, 0, 7):
        c = torch.rand(n, 2, generator=g) * torch.tensor([W - 30.0, H - 30.0]) + 15
        wh = torch.exp(torch.rand(n, 2, generator=g) * (np.log(150.0) - np.log(10.0)) + np.log(10.0))
        b = torch.cat([c - wh / 2, c + wh / 2], 1)
        boxes.append(b)
        labels.append(torch.randint(0, C, (n,), generator=g))
    boxes[0][3] = torch.tensor([500.0, 500.0, 503.0, 502.0])          # far outside the anchors of this 192x160 image: best IoU 0
    return cls, reg, boxes, labels, (H, W)


def main():
    ag, mi, ps, ah, rh = install()
    out = {}
    gen = ag.AnchorGenerator(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=STRIDES)
    cls, reg, boxes, labels, (H, W) = inputs()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        anchors = gen.grid_anchors(SIZES, device='cpu')
    flags = gen.valid_flags(SIZES, (H - 20, W - 40, 3), device='cpu')            # a padded shape smaller than the maps
    for i in range(len(SIZES)):
        out[f'out_anchors{i}'], out[f'out_flags{i}'] = anchors[i], flags[i]
        out[f'out_base{i}'] = gen.base_anchors[i]
    flat = torch.cat(anchors)
    for tag, kw in (('all', dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1)),
                    ('first', dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.2, gt_max_assign_all=False, ignore_iof_thr=-1)),
                    ('range', dict(pos_iou_thr=0.6, neg_iou_thr=(0.1, 0.4), min_pos_iou=0.3, ignore_iof_thr=-1)),
                    ('nolow', dict(pos_iou_thr=0.5, neg_iou_thr=0.4, match_low_quality=False, ignore_iof_thr=-1))):
        asg = mi.MaxIoUAssigner(**kw)
        for b in range(len(boxes)):
            r = asg.assign(flat, boxes[b], None, None)
            out[f'out_assign_{tag}_{b}'], out[f'out_maxov_{tag}_{b}'] = r.gt_inds, r.max_overlaps
    # targets + loss through the reference's AnchorHead code
    fl = L.ref('models.losses.focal_loss')
    sl = L.ref('models.losses.smooth_l1_loss')
    dc = L.ref('core.bbox.coder.delta_xywh_bbox_coder')
    for n in ['loss']:
        f = getattr(ah.AnchorHead, n)
        if hasattr(f, '__wrapped__'):
            setattr(ah.AnchorHead, n, f.__wrapped__)
    h = object.__new__(rh.RetinaHead)
    torch.nn.Module.__init__(h)
    h.num_classes = h.cls_out_channels = 8
    h.use_sigmoid_cls, h.sampling, h.reg_decoded_bbox = True, False, False
    h.rf_based, h.decay = False, 1                     # the RFLA fork option of this tree (anchor_head.py:45), off in the baseline
    h.anchor_generator = gen
    h.num_anchors = gen.num_base_anchors[0]
    h.assigner = mi.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1)
    h.sampler = ps.PseudoSampler()
    h.bbox_coder = dc.DeltaXYWHBBoxCoder(target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0])
    h.loss_cls = fl.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0)
    h.loss_bbox = sl.L1Loss(loss_weight=1.0)
    from types import SimpleNamespace
    h.train_cfg = SimpleNamespace(allowed_border=-1, pos_weight=-1, debug=False)
    for t in cls + reg:
        t.requires_grad_(True)
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3)) for _ in boxes]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        losses = h.loss(cls, reg, boxes, labels, metas)
        alist, vlist = h.get_anchors(SIZES, metas, device='cpu')
        tg = h.get_targets(alist, vlist, boxes, metas, gt_labels_list=labels, label_channels=8)
    total = sum(losses['loss_cls']) + sum(losses['loss_bbox'])
    grads = torch.autograd.grad(total, cls + reg)
    nl = len(SIZES)
    for i in range(nl):
        out[f'in_cls{i}'], out[f'in_reg{i}'] = cls[i], reg[i]
        out[f'out_loss_cls{i}'], out[f'out_loss_bbox{i}'] = losses['loss_cls'][i], losses['loss_bbox'][i]
        out[f'out_grad_cls{i}'], out[f'out_grad_reg{i}'] = grads[i], grads[nl + i]
        out[f'out_labels{i}'], out[f'out_label_weights{i}'] = tg[0][i], tg[1][i]
        out[f'out_bbox_targets{i}'], out[f'out_bbox_weights{i}'] = tg[2][i], tg[3][i]
    out['out_num_total_pos'] = np.int64(tg[4])
    for b in range(len(boxes)):
        out[f'in_gt_bboxes{b}'], out[f'in_gt_labels{b}'] = boxes[b], labels[b]
    out['in_pad_hw'] = np.array([H - 20, W - 40])
    G.save('retina_baseline', **out)


def gen_decode():
    """anchor_head.py get_bboxes / _get_bboxes (:503-739) with `with_nms=False`: per level top `nms_pre` by the best class
    score, DeltaXYWHBBoxCoder.decode with clipping to each image, rescale, background column
    -> tests/golden/retina_eval_decode.npz"""
    ag, mi, ps, ah, rh = install()
    import types
    import warnings
    dc = L.ref('core.bbox.coder.delta_xywh_bbox_coder')
    ex = L._pkg('mmdet.core.export')
    ex.get_k_for_topk = lambda k, size: int(k) if 0 < int(k) < size else -1      # core/export/onnx_helper.py outside ONNX export
    f = ah.AnchorHead.get_bboxes
    if hasattr(f, '__wrapped__'):
        ah.AnchorHead.get_bboxes = f.__wrapped__
    h = object.__new__(rh.RetinaHead)
    torch.nn.Module.__init__(h)
    h.num_classes = h.cls_out_channels = 8
    h.use_sigmoid_cls, h.rf_based, h.decay = True, False, 1
    h.anchor_generator = ag.AnchorGenerator(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=STRIDES)
    h.bbox_coder = dc.DeltaXYWHBBoxCoder(target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0])
    cls, reg, _, _, (H, W) = inputs(seed=29, B=2)
    metas = [dict(img_shape=(150, 180, 3), scale_factor=np.array([1.5, 1.25, 1.5, 1.25], np.float32)),
             dict(img_shape=(H, W, 3), scale_factor=np.array([1.0, 1.0, 1.0, 1.0], np.float32))]
    cfg = types.SimpleNamespace(score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100)
    cfg.get = lambda k, d=None: dict(nms_pre=200).get(k, d)
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for tag, rescale in (('plain', False), ('rescale', True)):
            res = h.get_bboxes(cls, reg, metas, cfg=cfg, rescale=rescale, with_nms=False)
            for b in range(2):
                out[f'out_{tag}_bboxes{b}'], out[f'out_{tag}_scores{b}'] = res[b]
    for i in range(len(SIZES)):
        out[f'in_cls{i}'], out[f'in_reg{i}'] = cls[i], reg[i]
    G.save('retina_eval_decode', **out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'decode':
        gen_decode()
    else:
        main()
