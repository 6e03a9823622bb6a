#!/usr/bin/env python3
"""Golden vectors for the Faster R-CNN baseline (third entry) - TEST INFRASTRUCTURE ONLY.

    python oracle/gen_golden_frcnn.py   ->  tests/golden/frcnn_baseline.npz

Runs the REFERENCE's own classes, loaded by path (/root/reference/HBB_TOD/mmdet), with the configuration of
configs/baselines/aitodv2_faster_rcnn_r50_1x.py on three levels of seeded head-shaped tensors:
  * models/dense_heads/rpn_head.py RPNHead.loss (-> anchor_head.py get_targets / loss with MaxIoUAssigner(.7, .3, .3) and
    core/bbox/samplers/random_sampler.py RandomSampler(256, .5)) - the permutations `torch.randperm` returned are recorded so
    that a restatement can be fed the same draws;
  * RPNHead.get_bboxes (per-level top-k, decode + clip, min size, level-aware NMS, max_per_img); the one mmcv op on this
    path, `batched_nms`, is bound to the oracle's greedy NMS (oracle/ref_ops.py - the recipe of gen_golden_obb.install_obb_eval);
  * the R-CNN stage: MaxIoUAssigner(.5, .5, .5, match_low_quality=False) + RandomSampler(512, .25, add_gt_as_proposals=True)
    over proposals, roi_heads/bbox_heads/bbox_head.py get_targets and loss (CrossEntropyLoss, L1Loss, accuracy) of a
    Shared2FCBBoxHead-shaped output (num_classes 8, class-specific regression, stds .1 .1 .2 .2).
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
import ref_ops as R  # noqa: E402

STRIDES = [4, 8, 16]
SIZES = [(24, 28), (12, 14), (6, 7)]
A = 3


class PermLog:
    """Records what torch.randperm returned (the only random draw of RandomSampler.random_choice)."""

    def __init__(self):
        self.perms = []
        self._orig = torch.randperm

    def __enter__(self):
        def logged(n, *a, **k):
            p = self._orig(n, *a, **k)
            self.perms.append(p.clone())
            return p
        torch.randperm = logged
        return self

    def __exit__(self, *exc):
        torch.randperm = self._orig


def install():
    ag, mi, ps, ah, rh = GR.install()
    imp = importlib.import_module
    md = os.path.join(L.HBB, 'mmdet')
    bs = imp('mmdet.core.bbox.samplers.base_sampler')
    rs_path = os.path.join(md, 'core', 'bbox', 'samplers', 'random_sampler.py')
    rs = imp('mmdet.core.bbox.samplers.random_sampler')
    # mmcv.ops.batched_nms(boxes, scores, idxs, nms_cfg) -> (dets [M,5], keep): the oracle's greedy NMS in its place
    def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
        keep = R.batched_nms(boxes.detach(), scores.detach(), idxs, float(nms_cfg['iou_threshold']))
        return torch.cat([boxes[keep], scores[keep, None]], -1), keep
    sys.modules['mmcv.ops'].batched_nms = batched_nms
    sys.modules['mmcv.cnn'].build_conv_layer = None          # only the (skipped) constructor of RPNHead calls it
    rpn = imp('mmdet.models.dense_heads.rpn_head')
    rpn.batched_nms = batched_nms
    # bbox head
    core = sys.modules['mmdet.core']
    misc = imp('mmdet.core.utils.misc')
    core.multi_apply = misc.multi_apply
    ls = sys.modules['mmdet.models.losses']
    acc = imp('mmdet.models.losses.accuracy')
    ls.accuracy = acc.accuracy
    mu = sys.modules['mmdet.models.utils']
    mu.build_linear_layer = lambda cfg, *a, **k: torch.nn.Linear(*a, **k)
    L._pkg('mmdet.models.roi_heads', os.path.join(md, 'models', 'roi_heads'))
    L._pkg('mmdet.models.roi_heads.bbox_heads', os.path.join(md, 'models', 'roi_heads', 'bbox_heads'))
    bh = imp('mmdet.models.roi_heads.bbox_heads.bbox_head')
    return ag, mi, ah, rpn, bs, rs, bh


def sampler(rs, bs, num, pos_fraction, add_gt):
    s = object.__new__(rs.RandomSampler)                 # the constructor only adds an (unused) numpy rng from demodata
    bs.BaseSampler.__init__(s, num, pos_fraction, -1, add_gt)
    return s


def inputs(seed=23, B=2):
    g = torch.Generator().manual_seed(seed)
    cls = [torch.randn(B, A, h, w, generator=g) * 1.5 - 1.0 for h, w in SIZES]
    reg = [torch.randn(B, A * 4, h, w, generator=g) * 0.2 for h, w in SIZES]
    H, W = SIZES[0][0] * 4, SIZES[0][1] * 4
    boxes, labels = [], []
    for n in (9, 5):
        c = torch.rand(n, 2, generator=g) * torch.tensor([W - 24.0, H - 24.0]) + 12
        wh = torch.exp(torch.rand(n, 2, generator=g) * (np.log(60.0) - np.log(14.0)) + np.log(14.0))
        boxes.append(torch.cat([c - wh / 2, c + wh / 2], 1))
        labels.append(torch.randint(0, 8, (n,), generator=g))
    return cls, reg, boxes, labels, (H, W)


def main():
    ag, mi, ah, rpn, bs, rs, bh = install()
    dc = L.ref('core.bbox.coder.delta_xywh_bbox_coder')
    ce = L.ref('models.losses.cross_entropy_loss')
    sl = L.ref('models.losses.smooth_l1_loss')
    for cls_, n in ((ah.AnchorHead, 'loss'), (rpn.RPNHead, 'get_bboxes'), (bh.BBoxHead, 'loss')):
        f = getattr(cls_, n)
        if hasattr(f, '__wrapped__'):
            setattr(cls_, n, f.__wrapped__)
    out = {}
    cls, reg, boxes, labels, (H, W) = inputs()
    # ------------------------------------------------------------------------------------------------ RPN loss --
    gen = ag.AnchorGenerator(scales=[8], ratios=[0.5, 1.0, 2.0], strides=STRIDES)
    h = object.__new__(rpn.RPNHead)
    torch.nn.Module.__init__(h)
    h.num_classes, h.cls_out_channels = 1, 1
    h.use_sigmoid_cls, h.sampling, h.reg_decoded_bbox = True, True, False
    h.rf_based, h.decay = False, 1
    h.anchor_generator = gen
    h.num_anchors = gen.num_base_anchors[0]
    h.assigner = mi.MaxIoUAssigner(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True, ignore_iof_thr=-1)
    h.sampler = sampler(rs, bs, 256, 0.5, False)
    h.bbox_coder = dc.DeltaXYWHBBoxCoder(target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0])
    h.loss_cls = ce.CrossEntropyLoss(use_sigmoid=True, loss_weight=1.0)
    h.loss_bbox = sl.L1Loss(loss_weight=1.0)
    h.train_cfg = types.SimpleNamespace(allowed_border=-1, pos_weight=-1, debug=False)
    for t in cls + reg:
        t.requires_grad_(True)
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=np.ones(4, np.float32)) for _ in boxes]
    torch.manual_seed(5)
    with warnings.catch_warnings(), PermLog() as pl:
        warnings.simplefilter('ignore')
        losses = h.loss(cls, reg, boxes, metas)
    for i, p in enumerate(pl.perms):
        out[f'in_rpn_perm{i}'] = p
    out['in_rpn_nperm'] = np.int64(len(pl.perms))
    total = sum(losses['loss_rpn_cls']) + sum(losses['loss_rpn_bbox'])
    grads = torch.autograd.grad(total, cls + reg)
    torch.manual_seed(5)
    with warnings.catch_warnings(), PermLog():
        warnings.simplefilter('ignore')
        alist, vlist = h.get_anchors(SIZES, metas, device='cpu')
        tg = h.get_targets(alist, vlist, boxes, metas, gt_labels_list=None, label_channels=1)
    nl = len(SIZES)
    for i in range(nl):
        out[f'in_cls{i}'], out[f'in_reg{i}'] = cls[i], reg[i]
        out[f'out_loss_rpn_cls{i}'], out[f'out_loss_rpn_bbox{i}'] = losses['loss_rpn_cls'][i], losses['loss_rpn_bbox'][i]
        out[f'out_grad_cls{i}'], out[f'out_grad_reg{i}'] = grads[i], grads[nl + i]
        out[f'out_labels{i}'], out[f'out_label_weights{i}'] = tg[0][i], tg[1][i]
        out[f'out_bbox_targets{i}'], out[f'out_bbox_weights{i}'] = tg[2][i], tg[3][i]
    out['out_num_total_pos'], out['out_num_total_neg'] = np.int64(tg[4]), np.int64(tg[5])
    for b in range(len(boxes)):
        out[f'in_gt_bboxes{b}'], out[f'in_gt_labels{b}'] = boxes[b], labels[b]
    out['in_img_hw'] = np.array([H, W])
    # ----------------------------------------------------------------------------------------------- proposals --
    cfg = types.SimpleNamespace(nms_pre=120, max_per_img=90, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        props = h.get_bboxes([c.detach() for c in cls], [r.detach() for r in reg], metas, cfg=cfg)
    for b, p in enumerate(props):
        out[f'out_proposals{b}'] = p
    # ------------------------------------------------------------------------------------------------- R-CNN stage --
    asg = mi.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False, ignore_iof_thr=-1)
    smp = sampler(rs, bs, 64, 0.25, True)                  # (512 in the config; 64 so that both classes are really sub-sampled)
    g = torch.Generator().manual_seed(31)
    # proposals of a half-trained RPN: the RPN's own proposals plus jittered copies of the boxes (so that positives exist)
    plist = []
    for b in range(len(boxes)):
        jit = boxes[b].repeat(4, 1) + torch.randn(boxes[b].shape[0] * 4, 4, generator=g) * 2.5
        plist.append(torch.cat([props[b][:, :4], jit]))
    results = []
    torch.manual_seed(7)
    with PermLog() as pl2:
        for b in range(len(boxes)):
            ar = asg.assign(plist[b], boxes[b], None, labels[b])
            results.append(smp.sample(ar, plist[b], boxes[b], labels[b]))
    for i, p in enumerate(pl2.perms):
        out[f'in_rcnn_perm{i}'] = p
    out['in_rcnn_nperm'] = np.int64(len(pl2.perms))
    hd = object.__new__(bh.BBoxHead)
    torch.nn.Module.__init__(hd)
    hd.num_classes, hd.reg_class_agnostic, hd.reg_decoded_bbox = 8, False, False
    hd.bbox_coder = dc.DeltaXYWHBBoxCoder(target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2])
    hd.loss_cls = ce.CrossEntropyLoss(use_sigmoid=False, loss_weight=1.0)
    hd.loss_bbox = sl.L1Loss(loss_weight=1.0)
    rcfg = types.SimpleNamespace(pos_weight=-1)
    lab, lw, bt, bw = hd.get_targets(results, boxes, labels, rcfg)
    n = lab.shape[0]
    cls_score = (torch.randn(n, 9, generator=g) * 1.2).requires_grad_(True)
    bbox_pred = (torch.randn(n, 32, generator=g) * 0.5).requires_grad_(True)
    tr = L.ref('core.bbox.transforms')
    rois = tr.bbox2roi([r.bboxes for r in results])
    ls = hd.loss(cls_score, bbox_pred, rois, lab, lw, bt, bw)
    gc, gb = torch.autograd.grad(ls['loss_cls'] + ls['loss_bbox'], (cls_score, bbox_pred))
    for b in range(len(boxes)):
        out[f'in_rcnn_proposals{b}'] = plist[b]
        out[f'out_rcnn_pos_inds{b}'], out[f'out_rcnn_neg_inds{b}'] = results[b].pos_inds, results[b].neg_inds
        out[f'out_rcnn_pos_assigned{b}'] = results[b].pos_assigned_gt_inds
        out[f'out_rcnn_bboxes{b}'] = results[b].bboxes
    out['out_rois'] = rois
    out['out_rcnn_labels'], out['out_rcnn_label_weights'], out['out_rcnn_bbox_targets'], out['out_rcnn_bbox_weights'] = lab, lw, bt, bw
    out['in_cls_score'], out['in_bbox_pred'] = cls_score, bbox_pred
    out['out_rcnn_loss_cls'], out['out_rcnn_loss_bbox'], out['out_rcnn_acc'] = ls['loss_cls'], ls['loss_bbox'], ls['acc']
    out['out_rcnn_grad_cls'], out['out_rcnn_grad_bbox'] = gc, gb
    G.save('frcnn_baseline', **out)


if __name__ == '__main__':
    main()
