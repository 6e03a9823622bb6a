#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE
(/root/reference, loaded file-by-file through oracle/_ref_loader.py) on seeded
inputs.  Runs only in the build container; the fixtures (inputs + the
reference's outputs, plain numpy arrays) are committed, this script is the
committed recipe that made them.  TEST INFRASTRUCTURE - never imported by the
product, by ``-m gpu`` tests, by smoke() or by bench.py.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

Every group below names the reference function (file:line under
/root/reference/HBB_TOD/mmdet) whose outputs it records.
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader as L  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(4)


def npy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    sz = os.path.getsize(path)
    print(f'  wrote {name}.npz  ({sz / 1024:.1f} KiB, {len(arrays)} arrays)')


def grid_points(h, w, stride=8):
    # fcos_head_p2b_ts.py:1007-1017
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(),
                            indexing='ij')
    return torch.stack((xs.reshape(-1) * stride, ys.reshape(-1) * stride), -1) + stride // 2


def rand_boxes(g, n, size, smin=2., smax=48.):
    cx = torch.rand(n, generator=g) * (size - 32) + 16
    cy = torch.rand(n, generator=g) * (size - 32) + 16
    w = torch.exp(torch.randn(n, generator=g) * 0.5 + np.log(12.)).clamp(smin, smax)
    h = torch.exp(torch.randn(n, generator=g) * 0.5 + np.log(12.)).clamp(smin, smax)
    return torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)


def tie_free_mask(points, gts_cxcywh, k):
    """Per gt: True when the float32 L1 distances (computed exactly as PointCost does,
    match_cost.py:206-210) of the k-th and (k+1)-th nearest grid point differ, i.e. the
    reference's ``torch.topk`` result is a well-defined SET.  On a regular grid the 5th
    and 6th nearest points tie structurally in exact arithmetic (8+a+b is reached from
    two neighbours), so about half of all random gts are ambiguous at k=5 in float32 and
    torch.topk's choice among them is unspecified (and differs CPU<->CUDA).  Goldens only
    contain unambiguous gts; the build's own rule for ties (lowest point index) is tested
    against the oracle, not against the reference."""
    d = torch.sum(torch.abs(points[:, None, :2] - gts_cxcywh[None, :, :2]), dim=2)
    ok = torch.ones(gts_cxcywh.shape[0], dtype=torch.bool)
    for kk in sorted(set([1, 3, k])):
        v, _ = torch.topk(d, kk + 1, dim=0, largest=False)
        ok &= v[kk - 1] < v[kk]
    return ok


def tie_free(points, gts_cxcywh, k):
    return bool(tie_free_mask(points, gts_cxcywh, k).all())


def sample_gt_boxes(g, G, size, points, k=5, center_fn=None):
    """G random xyxy boxes whose centres are all tie-free w.r.t. ``points``."""
    out = torch.zeros(0, 4)
    while out.shape[0] < G:
        b = rand_boxes(g, max(2 * (G - out.shape[0]), 8), size)
        if center_fn is not None:
            b = center_fn(b)
        c = torch.cat([(b[:, :2] + b[:, 2:]) / 2, b[:, 2:] - b[:, :2]], 1)
        # the centre the reference sees is recomputed from xyxy: test exactly that value
        out = torch.cat([out, b[tie_free_mask(points, c, k)]])
    return out[:G].contiguous()


# ---------------------------------------------------------------------------
def build_fake_head():
    """A stand-in for ``self`` carrying only configuration (assigners, loss
    modules, num_classes), with the reference's own methods attached."""
    hd = L.ref('models.dense_heads.fcos_head_p2b_ts')
    bld = L.ref('core.bbox.builder')
    fl = L.ref('models.losses.focal_loss')
    il = L.ref('models.losses.iou_loss')
    ce = L.ref('models.losses.cross_entropy_loss')
    dc = L.ref('core.bbox.coder.delta_xywh_bbox_coder')

    class FakeHead:
        pass
    for n in ['loss_pseudo', 'loss', 'get_targets', '_get_target_single',
              'get_target_pseudo', '_get_target_pseudo_single', 'gnerate_pseudo',
              '_gnerate_pseudo_single', 'get_pseudo_bbox', 'centerness_target',
              'concat_per_img', 'concat_prediction', 'gfocal_loss',
              'mil_bag_selection_single', 'mil_bag_selection', 'mil_bag_training',
              '_get_points_single']:
        setattr(FakeHead, n, getattr(hd.TS_P2BFCOSHead, n))
    h = FakeHead()
    h.num_classes = 8
    h.cls_out_channels = 8
    h.strides = [8]
    # aitodv2_point_teacher_0%.py:103-127
    h.assigner = bld.build_assigner(dict(
        type='TopkAssigner', num_pre=1, topk=1,
        cls_cost=dict(type='FocalLossCost', weight=1.0),
        reg_cost=dict(type='PointCost', mode='L1', weight=1.0)))
    h.pseudo_assigner = bld.build_assigner(dict(
        type='TopkAssigner', num_pre=3, topk=3,
        cls_cost=dict(type='FocalLossCost', weight=0.0),
        reg_cost=dict(type='PointCost', mode='L1', weight=1.0)))
    h.syn_assigner = bld.build_assigner(dict(
        type='TopkAssigner', num_pre=3, topk=3,
        cls_cost=dict(type='FocalLossCost', weight=0.0),
        reg_cost=dict(type='PointCost', mode='L1', weight=1.0)))
    h.fuse_assigner = bld.build_assigner(dict(
        type='FUSETopkAssigner', num_pre=5, topk=3,
        cls_cost=dict(type='FocalLossCost', weight=1.0),
        reg_cost=dict(type='PointCost', mode='L1', weight=1.0),
        location_cost=dict(type='InsiderCost', weight=1.0)))
    h.loss_cls = fl.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0)
    h.loss_bbox = il.DIoULoss(loss_weight=1.0)
    h.loss_bbox_burn2 = il.DN_DIoULoss(loss_weight=1.0, hyper=0.1)
    h.loss_bbox_denosing = il.DN_DIoULoss(loss_weight=1.0, hyper=0.2)
    h.loss_centerness = ce.CrossEntropyLoss(use_sigmoid=True, loss_weight=1.0)
    h.mil_bbox_decoder = dc.DeltaXYWHBBoxCoder(target_means=[0., 0., 0., 0.],
                                               target_stds=[1., 1., 1., 1.])
    h.beta = 0.25
    h.topk = 1
    return h


def head_outputs(g, B, H, W, C=8):
    cls = torch.randn(B, C, H, W, generator=g) * 1.5 - 2.0
    reg = (torch.randn(B, 4, H, W, generator=g) * 0.8 + 1.0).clamp(min=0) * 8
    ctr = torch.randn(B, 1, H, W, generator=g)
    return cls, reg, ctr


# ---------------------------------------------------------------------------
def gen_costs():
    # match_cost.py:54-99 (FocalLossCost), :188-214 (PointCost), :217-252 (InsiderCost)
    mc = L.ref('core.bbox.match_costs.match_cost')
    g = torch.Generator().manual_seed(11)
    P, G = 96, 7
    pts = grid_points(8, 12)
    boxes = torch.cat([pts + torch.randn(P, 2, generator=g) * 3,
                       torch.rand(P, 2, generator=g) * 30 + 1], 1)      # cxcywh
    gts = torch.cat([torch.rand(G, 2, generator=g) * 90, torch.rand(G, 2, generator=g) * 20], 1)
    cls = torch.randn(P, 8, generator=g) * 2
    labels = torch.randint(0, 8, (G,), generator=g)
    save('costs',
         in_boxes_cxcywh=boxes, in_gt_cxcywh=gts, in_cls=cls, in_labels=labels,
         out_point_l1=mc.PointCost('L1', 1.0)(boxes, gts),
         out_point_l2=mc.PointCost('L2', 2.0)(boxes, gts),
         out_focal=mc.FocalLossCost(weight=1.0)(cls, labels),
         out_insider=mc.InsiderCost(weight=1.0)(boxes, gts))


def gen_assigners():
    # topk_assigner.py:54-147, fuse_topk_assigner.py:56-121
    h = build_fake_head()
    tr = L.ref('core.bbox.transforms')
    cases = [('g0', 20, 20, 0, 1), ('g1', 20, 20, 1, 2), ('g7', 20, 20, 7, 3),
             ('g40', 32, 32, 40, 4), ('clustered', 32, 32, 60, 5),
             ('full', 100, 100, 300, 6)]
    for name, H, W, G, seed in cases:
        g = torch.Generator().manual_seed(seed)
        pts = grid_points(H, W)
        P = pts.shape[0]
        size = 8 * H
        cfn = None
        if name == 'clustered':              # collision-heavy: gts piled on a few spots
            ctr = torch.rand(6, 2, generator=g) * (size - 64) + 32

            def cfn(b):
                n = b.shape[0]
                pick = torch.randint(0, 6, (n,), generator=g)
                c = ctr[pick] + torch.randn(n, 2, generator=g) * 5
                wh = (b[:, 2:] - b[:, :2])
                return torch.cat([c - wh / 2, c + wh / 2], 1)
        gtb = sample_gt_boxes(g, G, size, pts, 5, cfn) if G else torch.zeros(0, 4)
        gcx = tr.bbox_xyxy_to_cxcywh(gtb) if G else gtb.new_zeros((0, 4))
        assert G == 0 or tie_free(pts, gcx, 5)
        labels = torch.randint(0, 8, (G,), generator=g)
        cls = torch.randn(P, 8, generator=g) * 1.5 - 2.0
        reg = (torch.randn(P, 4, generator=g) * 0.8 + 1.0).clamp(min=0) * 8
        ctrn = torch.randn(P, 1, generator=g)
        gt_points = gcx[:, :2].contiguous()
        out = {}
        for tag, asg in (('a11', h.assigner), ('a33', h.pseudo_assigner)):
            r = asg.assign(pts, cls, gcx, labels)
            out[f'out_{tag}_gt_inds'] = r.gt_inds
            out[f'out_{tag}_labels'] = r.labels
        # the (1,1) assigner is also called with bare [G,2] points (fcos_head_p2b_ts.py:665)
        r = h.assigner.assign(pts, cls, gt_points, labels)
        out['out_a11pts_gt_inds'] = r.gt_inds
        dec = tr.distance2bbox(pts, reg)
        dcx = tr.bbox_xyxy_to_cxcywh(dec)
        r = h.fuse_assigner.assign(dcx, pts, cls, ctrn, gt_points, labels)
        out['out_fuse_gt_inds'] = r.gt_inds
        out['out_fuse_labels'] = r.labels
        save(f'assign_{name}', in_points=pts, in_cls=cls, in_reg=reg, in_ctr=ctrn,
             in_gt_bboxes=gtb, in_gt_labels=labels, hw=np.array([H, W]), **out)


def gen_pseudo_and_targets():
    # fcos_head_p2b_ts.py:736-794 (_gnerate_pseudo_single), :658-708, :570-605
    h = build_fake_head()
    tr = L.ref('core.bbox.transforms')
    for name, H, W, G, seed in [('small', 20, 20, 9, 21), ('mid', 50, 50, 80, 22)]:
        g = torch.Generator().manual_seed(seed)
        pts = grid_points(H, W)
        P = pts.shape[0]
        size = 8 * H
        gtb = sample_gt_boxes(g, G, size, pts, 5)
        gcx = tr.bbox_xyxy_to_cxcywh(gtb)
        labels = torch.randint(0, 8, (G,), generator=g)
        cls = torch.randn(P, 8, generator=g) * 1.5 - 2.0
        reg = (torch.randn(P, 4, generator=g) * 0.8 + 1.0).clamp(min=0) * 8
        ctrn = torch.randn(P, 1, generator=g)
        gt_points = gcx[:, :2].contiguous()
        meta = dict(ori_filename='x.png', img_shape=(size, size, 3))
        pb, pp, pl, miou, valid = h._gnerate_pseudo_single(
            gt_points, labels, gtb, cls, reg, ctrn, meta, None,
            filter_scores=0.0, points=pts, num_points_per_lvl=[P])
        # pseudo targets from those pseudo boxes (step-2 loss path)
        lr, bt, lab, wts = h._get_target_pseudo_single(
            gt_points, labels, pp, pl, pb, cls, reg, ctrn, meta, None, None,
            points=pts, num_points_per_lvl=[P], burn_in_step1=False)
        # synthetic-box targets (step-1 loss path)
        sl, sbt = h._get_target_single(gtb, cls, reg, ctrn, points=pts,
                                       num_points_per_lvl=[P])
        save(f'pseudo_{name}', in_points=pts, in_cls=cls, in_reg=reg, in_ctr=ctrn,
             in_gt_points=gt_points, in_gt_labels=labels, in_gt_bboxes=gtb,
             out_pseudo_bboxes=pb, out_pseudo_points=pp, out_mean_iou=miou,
             out_valid_inds=np.sort(npy(valid)),
             out_labels_reg=lr, out_bbox_targets=bt, out_labels=lab, out_weights=wts,
             out_syn_labels=sl, out_syn_bbox_targets=sbt,
             out_ctr_target=h.centerness_target(bt[lr < 8]))


def gen_losses():
    il = L.ref('models.losses.iou_loss')
    fl = L.ref('models.losses.focal_loss')
    ce = L.ref('models.losses.cross_entropy_loss')
    g = torch.Generator().manual_seed(31)
    N = 257
    tgt = rand_boxes(g, N, 400)
    pred = (tgt + torch.randn(N, 4, generator=g) * 4).requires_grad_(True)
    # a few degenerate rows: no overlap, identical, inverted
    with torch.no_grad():
        pred[0] = tgt[0]
        pred[1] = tgt[1] + 200.
        pred[2, 2:] = pred[2, :2] - 1.
    w = torch.rand(N, generator=g)
    w[5:20] = 0
    out = {}
    # iou_loss.py:139-189 diou_loss (through @weighted_loss, losses/utils.py:57-98)
    l_none = il.diou_loss(pred, tgt, reduction='none', eps=1e-6)
    out['out_diou_none'] = l_none
    loss = il.DIoULoss(loss_weight=1.0)(pred, tgt, weight=w, avg_factor=37.5)
    gd, = torch.autograd.grad(loss, pred)
    out['out_diou_loss'] = loss
    out['out_diou_grad'] = gd
    # iou_loss.py:398-465 DN_diou_loss
    for hyper in (0.1, 0.2):
        ln = il.DN_diou_loss(pred, tgt, reduction='none', hyper=hyper, eps=1e-6)
        loss = il.DN_DIoULoss(loss_weight=1.0, hyper=hyper)(pred, tgt, weight=w, avg_factor=37.5)
        gd, = torch.autograd.grad(loss, pred)
        tag = str(hyper).replace('.', '')
        out[f'out_dn{tag}_none'] = ln
        out[f'out_dn{tag}_loss'] = loss
        out[f'out_dn{tag}_grad'] = gd
    # zero-weight shortcut iou_loss.py:858-861
    out['out_dn_zero_w'] = il.DN_DIoULoss(hyper=0.1)(pred, tgt, weight=torch.zeros(N), avg_factor=1.0)
    save('loss_diou', in_pred=pred, in_target=tgt, in_weight=w, avg_factor=np.float32(37.5), **out)

    # focal_loss.py:11-56 py_sigmoid_focal_loss via FocalLoss.forward CPU branch :159-176
    M, C = 600, 8
    logits = (torch.randn(M, C, generator=g) * 2.5).requires_grad_(True)
    labels = torch.randint(0, C + 1, (M,), generator=g)
    labels[:400] = C                      # mostly background as in training
    weight = torch.ones(M)
    weight[::7] = 0.5
    loss = fl.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25)(logits, labels, weight=weight, avg_factor=23.0)
    gd, = torch.autograd.grad(loss, logits)
    onehot = torch.nn.functional.one_hot(labels, C + 1)[:, :C]
    elem = fl.py_sigmoid_focal_loss(logits, onehot, reduction='none')
    save('loss_focal', in_logits=logits, in_labels=labels, in_weight=weight,
         avg_factor=np.float32(23.0), out_loss=loss, out_grad=gd, out_elem=elem)

    # cross_entropy_loss.py:62-100 binary_cross_entropy on 1-D logits with soft targets
    K = 301
    x = (torch.randn(K, generator=g) * 2).requires_grad_(True)
    t = torch.rand(K, generator=g)
    loss = ce.CrossEntropyLoss(use_sigmoid=True, loss_weight=1.0)(x, t, avg_factor=77.0)
    gd, = torch.autograd.grad(loss, x)
    save('loss_bce', in_logits=x, in_target=t, avg_factor=np.float32(77.0), out_loss=loss, out_grad=gd)


def gen_bbox_math():
    tr = L.ref('core.bbox.transforms')
    iou = L.ref('core.bbox.iou_calculators.iou2d_calculator')
    dc = L.ref('core.bbox.coder.delta_xywh_bbox_coder')
    dab = L.ref('models.detectors.data_augument_bank')
    syn = L.ref('models.detectors.syn_images_generator_v2')
    g = torch.Generator().manual_seed(41)
    a = rand_boxes(g, 64, 300)
    b = rand_boxes(g, 40, 300)
    b2 = a + torch.randn(64, 4, generator=g) * 3
    out = {}
    for mode in ('iou', 'iof', 'giou'):
        out[f'out_pair_{mode}'] = iou.bbox_overlaps(a, b, mode=mode)
        out[f'out_align_{mode}'] = iou.bbox_overlaps(a, b2, mode=mode, is_aligned=True)
    # known answers the reference's own tests hold
    # tests/test_metrics/test_box_overlap.py:87-105
    k1 = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]])
    k2 = torch.FloatTensor([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]])
    out['known_giou_aligned'] = iou.bbox_overlaps(k1, k2, 'giou', is_aligned=True)
    # tests/test_utils/test_misc.py:51-76 distance2bbox
    point = torch.Tensor([[74., 61.], [-29., 106.], [138., 61.], [29., 170.]])
    dist = torch.Tensor([[0., 0, 1., 1.], [1., 2., 10., 6.], [22., -29., 138., 61.], [54., -29., 170., 0.]])
    out['known_d2b'] = tr.distance2bbox(point, dist, max_shape=(110, 200))
    # tests/test_utils/test_coder.py:26-75 delta2bbox
    rois = torch.Tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    deltas = torch.Tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
    out['known_delta2bbox'] = dc.delta2bbox(rois, deltas, max_shape=(32, 32))
    pts = torch.rand(50, 2, generator=g) * 200
    d4 = torch.randn(50, 4, generator=g) * 20
    out['out_d2b'] = tr.distance2bbox(pts, d4)
    out['out_d2b_clip'] = tr.distance2bbox(pts, d4, max_shape=(200, 160, 3))
    dl = torch.randn(64, 4, generator=g) * 0.8
    dl[3, 2] = 9.0
    dl[4, 3] = -9.0          # exercise the wh_ratio_clip
    coder = dc.DeltaXYWHBBoxCoder(target_means=[0., 0., 0., 0.], target_stds=[1., 1., 1., 1.])
    out['out_delta_decode'] = coder.decode(a, dl, max_shape=(300, 300, 3))
    out['out_xyxy2cxcywh'] = tr.bbox_xyxy_to_cxcywh(a)
    out['out_cxcywh2xyxy'] = tr.bbox_cxcywh_to_xyxy(tr.bbox_xyxy_to_cxcywh(a))
    out['out_bbox2roi'] = tr.bbox2roi([a[:5], a[:0], b[:3]])
    # rotated helpers used by the step-1 generator
    obb = torch.cat([torch.rand(30, 2, generator=g) * 300, torch.rand(30, 2, generator=g) * 40 + 2,
                     (torch.rand(30, 1, generator=g) - 0.5) * np.pi], 1)
    out['out_obb2poly'] = dab.obb2poly_le90(obb)
    out['out_obb2xyxy'] = syn.obb2xyxy(obb)
    # random_point_in_quadrilateral with injected uniforms (data_augument_bank.py:1471-1493)
    u = torch.rand(2, 64, generator=g)
    seq = [u[0], u[1]]
    real_rand = torch.rand
    torch.rand = lambda *a_, **k_: seq.pop(0)
    try:
        for pos in (0.0, 0.3, 1.0):
            seq[:] = [u[0], u[1]]
            out[f'out_rpq_{int(pos * 100)}'] = dab.random_point_in_quadrilateral(a, pos)
    finally:
        torch.rand = real_rand
    save('bbox_math', in_a=a, in_b=b, in_b2=b2, in_pts=pts, in_d4=d4, in_deltas=dl,
         in_obb=obb, in_u=u, **out)


FINE_CFGS = [   # aitodv2_point_teacher_0%.py:128-167 and _100%.py:128-167
    dict(gen_mode='refine', gen_proposal_mode='fix_gen', cut_mode=None, shake_ratio=None,
         base_ratios=[1.0], min_scale=0, gen_num_neg=200),
    dict(gen_mode='refine', gen_proposal_mode='fix_gen', cut_mode=None, shake_ratio=None,
         base_ratios=[1.0, 1.2, 1.3, 0.8, 0.7], min_scale=4, gen_num_neg=0),
    dict(gen_mode='refine', gen_proposal_mode='fix_gen', cut_mode=None, shake_ratio=[0.1],
         base_ratios=[1.0, 1.2, 1.3, 0.8, 0.7], min_scale=16, gen_num_neg=0),
    dict(gen_mode='refine', gen_proposal_mode='fix_gen', cut_mode=None, shake_ratio=None,
         base_ratios=[1.0, 1.3, 0.8], min_scale=0, gen_num_neg=200),
    dict(gen_mode='refine', gen_proposal_mode='fix_gen', cut_mode=None, shake_ratio=[0.1],
         base_ratios=[1.0, 1.3, 0.7], min_scale=4, gen_num_neg=0),
]


def gen_proposals():
    # syn_images_generator_v2.py:134-145, :262-324, :234-259
    syn = L.ref('models.detectors.syn_images_generator_v2')
    tr = L.ref('core.bbox.transforms')
    g = torch.Generator().manual_seed(51)
    metas = [dict(img_shape=(320, 320, 3)), dict(img_shape=(320, 320, 3))]
    boxes = [rand_boxes(g, 11, 320), rand_boxes(g, 6, 320)]
    # push some boxes over the border so the IoF validity mask has zeros
    boxes[0][0] = torch.tensor([-20., 5., 6., 30.])
    boxes[0][1] = torch.tensor([300., 300., 340., 350.])
    boxes[1][0] = torch.tensor([310., -8., 330., 4.])
    gts = [b + torch.randn_like(b) for b in boxes]
    pts = [tr.bbox_xyxy_to_cxcywh(b)[:, :2] for b in boxes]
    out = {}
    for ci, cfg in enumerate(FINE_CFGS):
        pl, vl, rl, gl = syn.MIL_gen_proposals_from_cfg(pts, boxes, cfg, gts, metas)
        for i in range(2):
            out[f'out_c{ci}_prop{i}'] = pl[i]
            out[f'out_c{ci}_valid{i}'] = vl[i]
            out[f'out_c{ci}_ref{i}'] = rl[i]
            out[f'out_c{ci}_real{i}'] = gl[i]
    # negatives with injected uniforms: torch.rand is called 4x per image (:247-250)
    u = torch.rand(2, 4, 200, generator=g)
    seq = []
    real_rand = torch.rand
    torch.rand = lambda *a_, **k_: seq.pop(0)
    try:
        seq[:] = [u[i, j] for i in range(2) for j in range(4)]
        pl, _, _, _ = syn.MIL_gen_proposals_from_cfg(pts, boxes, FINE_CFGS[0], gts, metas)
        nl, nw = syn.gen_negative_proposals(pts, FINE_CFGS[0], pl, metas)
    finally:
        torch.rand = real_rand
    for i in range(2):
        out[f'out_neg{i}'] = nl[i]
        out[f'out_negw{i}'] = nw[i]
    save('proposals', in_boxes0=boxes[0], in_boxes1=boxes[1], in_gt0=gts[0], in_gt1=gts[1],
         in_u=u, img_hw=np.array([320, 320]), **out)


def gen_mil_bags():
    # fcos_head_p2b_ts.py:1147-1180 (mil_bag_training), :1112-1145/:1092-1110 (selection)
    h = build_fake_head()
    g = torch.Generator().manual_seed(61)
    for name, ngt_per_img, U1, U2, k in [('u1x25_k1', [13, 9], 1, 25, 1),
                                         ('u9x45_k3', [5, 4], 9, 45, 3),
                                         ('full_k1', [100, 100], 1, 25, 1)]:
        N = sum(ngt_per_img)
        C = 8
        cls = (torch.randn(N, U1, U2, C, generator=g) * 2).requires_grad_(True)
        ins = (torch.randn(N, U1, U2, C, generator=g) * 2).requires_grad_(True)
        valid = (torch.rand(N * U1 * U2, 1, generator=g) > 0.15)
        valid.view(N, U1, U2)[1, 0, :] = False         # an entirely invalid bag
        labels = torch.randint(0, C, (N,), generator=g)
        nneg = 400
        negc = (torch.randn(nneg, C, generator=g) * 2).requires_grad_(True)
        negw = torch.rand(nneg, generator=g) > 0.3
        bags = rand_boxes(g, N * U1 * U2, 800)
        pseudo = rand_boxes(g, N, 800)
        # split per image exactly as the head does
        vs, bs, ls, ps, ws = [], [], [], [], []
        o = 0
        for n_i in ngt_per_img:
            vs.append(valid[o * U1 * U2:(o + n_i) * U1 * U2])
            bs.append(bags[o * U1 * U2:(o + n_i) * U1 * U2])
            ls.append(labels[o:o + n_i])
            ps.append(pseudo[o:o + n_i])
            o += n_i
        ws = [negw[:200], negw[200:]]
        res = dict(cls_score=cls, ins_score=ins, iou_target=None, extensive_bags_valid=vs,
                   extensive_bags=bs, neg_cls_score=negc)
        loss = h.mil_bag_training(res, ls, ws)
        gc, gi, gn = torch.autograd.grad(loss, [cls, ins, negc])
        h.topk = k
        metas = [dict(img_shape=(800, 800, 3))] * len(ngt_per_img)
        merged = h.mil_bag_selection(res, metas, ps, ls)
        save(f'mil_{name}', in_cls=cls, in_ins=ins, in_valid=valid, in_labels=labels,
             in_neg_cls=negc, in_neg_w=negw, in_bags=bags, in_pseudo=pseudo,
             ngt=np.array(ngt_per_img), topk=np.array(k), beta=np.float32(h.beta),
             out_loss=loss, out_grad_cls=gc, out_grad_ins=gi, out_grad_neg=gn,
             out_merged=torch.cat(list(merged)))
    h.topk = 1


def gen_strong_aug():
    # syn_images_generator_v2.py:24-132 with the python/numpy draws injected
    syn = L.ref('models.detectors.syn_images_generator_v2')
    g = torch.Generator().manual_seed(71)
    B, H, W = 4, 64, 64
    img = torch.round(torch.rand(B, 3, H, W, generator=g) * 255)
    flips = ['horizontal', 'vertical', 'diagonal', 'None']
    scales = [0.8, 1.2, 1.0, 0.9]
    gt_pts = [torch.rand(7, 2, generator=g) * 64 for _ in range(B)]
    gt_lab = [torch.randint(0, 8, (7,), generator=g) for _ in range(B)]
    pb = [rand_boxes(g, 7, 64, 2, 12) for _ in range(B)]
    pp = [(b[:, :2] + b[:, 2:]) / 2 for b in pb]
    pl = [l.clone() for l in gt_lab]
    fq, sq = list(flips), list(scales)
    real_choice, real_unif = random.choice, np.random.uniform
    random.choice = lambda seq_: fq.pop(0)
    np.random.uniform = lambda a_, b_: sq.pop(0)
    try:
        ia, _, agp, agl, app, apl, apb = syn.strong_augmentation(img, gt_pts, gt_lab, pp, pl, pb)
    finally:
        random.choice, np.random.uniform = real_choice, real_unif
    out = dict(out_img=ia)
    for i in range(B):
        out[f'out_gt_points{i}'] = agp[i]
        out[f'out_gt_labels{i}'] = agl[i]
        out[f'out_pseudo_points{i}'] = app[i]
        out[f'out_pseudo_labels{i}'] = apl[i]
        out[f'out_pseudo_bboxes{i}'] = apb[i]
        out[f'in_gt_points{i}'] = gt_pts[i]
        out[f'in_gt_labels{i}'] = gt_lab[i]
        out[f'in_pseudo_bboxes{i}'] = pb[i]
    save('strong_aug', in_img=img, flips=np.array(flips), scales=np.array(scales), **out)


def gen_head_losses():
    """loss_pseudo (fcos_head_p2b_ts.py:380-465), loss (:470-534) and
    get_pseudo_bbox (:357-375) end to end on head-shaped tensors, with grads."""
    h = build_fake_head()
    tr = L.ref('core.bbox.transforms')
    for name, B, H, W, Gs, seed in [('small', 2, 20, 20, [9, 5], 81), ('mid', 2, 40, 40, [50, 33], 82)]:
        g = torch.Generator().manual_seed(seed)
        size = 8 * H
        pts = grid_points(H, W)
        cls, reg, ctr = head_outputs(g, B, H, W)
        cls.requires_grad_(True); reg.requires_grad_(True); ctr.requires_grad_(True)
        gtb, gpts, glab = [], [], []
        for G in Gs:
            b = sample_gt_boxes(g, G, size, pts, 5)
            c = tr.bbox_xyxy_to_cxcywh(b)
            gtb.append(b); gpts.append(c[:, :2].contiguous())
            glab.append(torch.randint(0, 8, (G,), generator=g))
        metas = [dict(ori_filename=f'i{i}.png', img_shape=(size, size, 3)) for i in range(B)]
        # teacher side: pseudo boxes from (detached) predictions
        pb, pp, pl, miou, valid = h.get_pseudo_bbox([cls.detach()], [reg.detach()], [ctr.detach()], [pts],
                                                    gpts, glab, gtb, 0.0, metas, [None] * B)
        # student side: loss on the same maps (pseudo boxes jittered so the loss is not trivial)
        pb2 = [b + torch.randn(b.shape, generator=g) * 1.5 for b in pb]
        pp2 = [tr.bbox_xyxy_to_cxcywh(b)[:, :2] for b in pb2]
        lc, lb, lct = h.loss_pseudo([cls], [reg], [ctr], [pts], gpts, glab, pp2, pl, pb2,
                                    [None] * B, metas, [None] * B, False)
        gr = torch.autograd.grad(lc + lb + lct, [cls, reg, ctr])
        sb, sct = h.loss([cls], [reg], [ctr], [pts], gtb, metas)
        gs = torch.autograd.grad(sb + sct, [reg, ctr])
        out = dict(out_mean_iou=miou, out_loss_cls=lc, out_loss_bbox=lb, out_loss_ctr=lct,
                   out_grad_cls=gr[0], out_grad_reg=gr[1], out_grad_ctr=gr[2],
                   out_syn_loss_bbox=sb, out_syn_loss_ctr=sct, out_syn_grad_reg=gs[0],
                   out_syn_grad_ctr=gs[1])
        for i in range(B):
            out[f'in_gt_bboxes{i}'] = gtb[i]
            out[f'in_gt_points{i}'] = gpts[i]
            out[f'in_gt_labels{i}'] = glab[i]
            out[f'out_pseudo_bboxes{i}'] = pb[i]
            out[f'out_pseudo_points{i}'] = pp[i]
            out[f'in_pseudo_bboxes_jit{i}'] = pb2[i]
        save(f'head_{name}', in_cls=cls, in_reg=reg, in_ctr=ctr, in_points=pts, **out)


def gen_ema():
    # fcos_p2b_teacher_student.py:254-257
    g = torch.Generator().manual_seed(91)
    t = torch.randn(1000, generator=g)
    s = torch.randn(1000, generator=g)
    t2 = t.clone()
    t2.mul_(0.999).add_(s, alpha=1 - 0.999)
    save('ema', in_teacher=t, in_student=s, out_teacher=t2, alpha=np.float32(0.999))


if __name__ == '__main__':
    L.install()
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    for fn in [gen_costs, gen_assigners, gen_pseudo_and_targets, gen_losses, gen_bbox_math,
               gen_proposals, gen_mil_bags, gen_strong_aug, gen_head_losses, gen_ema]:
        print(fn.__name__)
        fn()
    print('done')
