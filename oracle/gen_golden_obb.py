#!/usr/bin/env python3
"""Generate the ORIENTED-BOX golden fixtures tests/golden/obb_*.npz by RUNNING THE REFERENCE
(/root/reference/OBB_TOD/mmrotate, loaded file-by-file through oracle/_ref_loader.install_obb)
on seeded inputs.  Build container only; the fixtures are data (inputs + the reference's
outputs), this script is the committed recipe.  TEST INFRASTRUCTURE - never imported by the
product, by `-m gpu` tests, by smoke() or by bench.py.

    python oracle/gen_golden_obb.py

What the reference cannot compute here is replaced by a recorded no-op, and the affected
outputs are NOT stored: torchvision's TF.rotate (image pixels of strong_augmentation) and
mmcv's rbbox_overlaps (the mean-IoU log value of _gnerate_pseudo_single).
Citations are file:line under /root/reference/OBB_TOD/mmrotate.
"""
import math
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader as L  # noqa: E402
import gen_golden as G  # noqa: E402  (helpers: save, grid_points, tie-free sampling)

torch.set_num_threads(4)


def rand_rboxes(g, n, size, smin=4.0, smax=40.0):
    c = torch.rand(n, 2, generator=g) * (size - 48) + 24
    wh = torch.exp(torch.randn(n, 2, generator=g) * 0.5 + math.log(14.0)).clamp(smin, smax)
    a = torch.rand(n, generator=g) * math.pi - math.pi / 2
    return torch.stack([c[:, 0], c[:, 1], torch.max(wh[:, 0], wh[:, 1]), torch.min(wh[:, 0], wh[:, 1]), a], 1)


def tie_free_rboxes(g, n, size, points, k=5):
    """Oriented boxes whose centres have no exact fp32 L1 tie among their k+1 nearest grid points."""
    out = []
    while len(out) < n:
        b = rand_rboxes(g, 2 * n + 8, size)
        ok = G.tie_free_mask(points, b[:, :4], k)
        out.extend(list(b[ok]))
    return torch.stack(out[:n])


def gen_transforms():
    # core/bbox/transforms.py:301-331 (poly2obb_le90), :474-499 (obb2poly_le90), :850-867 (norm_angle)
    tr = L.ref_obb('core.bbox.transforms')
    g = torch.Generator().manual_seed(101)
    rb = rand_rboxes(g, 64, 400)
    rb[0, 2:4] = 10.0                                   # a square: edge1 == edge2 branch
    poly = tr.obb2poly_le90(rb)
    back = tr.poly2obb_le90(poly)
    jit = poly + torch.randn(poly.shape, generator=g) * 2.0       # general quadrilaterals
    ang = (torch.rand(50, generator=g) - 0.5) * 12
    G.save('obb_transforms', in_rboxes=rb, out_poly=poly, out_back=back, in_quads=jit, out_quads_obb=tr.poly2obb_le90(jit),
           in_angles=ang, out_norm_le90=tr.norm_angle(ang, 'le90'), out_obb2xyxy=tr.obb2xyxy(rb, 'le90'))


def gen_coder():
    # core/bbox/coder/distance_angle_point_coder.py:9-111
    cd = L.ref_obb('core.bbox.coder.distance_angle_point_coder').DistanceAnglePointCoder(angle_version='le90')
    g = torch.Generator().manual_seed(102)
    pts = torch.rand(200, 2, generator=g) * 400
    dist = torch.cat([torch.rand(200, 4, generator=g) * 40, (torch.rand(200, 1, generator=g) - 0.5) * 4], 1)
    dec = cd.decode(pts, dist)
    gt = rand_rboxes(g, 200, 400)
    G.save('obb_coder', in_points=pts, in_distance=dist, out_decode=dec, in_gt=gt, out_encode=cd.encode(pts, gt),
           out_encode_clamped=cd.encode(pts, gt, max_dis=16.0, eps=0.1))


def gen_proposals():
    # detectors/syn_images_generator_v2.py:24-39 (MIL_gen_proposals_from_cfg) over :159-215
    syn = L.ref_obb('models.detectors.syn_images_generator_v2')
    g = torch.Generator().manual_seed(103)
    size = 256
    metas = [dict(img_shape=(size, size, 3))] * 2
    boxes = [rand_rboxes(g, 9, size), rand_rboxes(g, 5, size)]
    boxes[0][0, :2] = torch.tensor([3.0, 250.0])        # a bag that leaves the image -> invalid entries
    real = [rand_rboxes(g, 9, size), rand_rboxes(g, 5, size)]
    out = {}
    for tag, cfg in (('coarse', dict(gen_mode='refine', gen_proposal_mode='fix_gen', cut_mode=None, shake_ratio=None,
                                     base_ratios=[1.0], min_scale=0)),
                     ('ext', dict(gen_mode='refine', gen_proposal_mode='fix_gen', cut_mode=None, shake_ratio=None,
                                  base_ratios=[1.0, 1.2, 1.3, 0.8, 0.6], min_scale=4))):
        pr, pv, pref, preal = syn.MIL_gen_proposals_from_cfg([b[:, :2] for b in boxes], [b.clone() for b in boxes], cfg,
                                                            [r.clone() for r in real], metas)
        for i in range(2):
            out[f'out_{tag}_props{i}'] = pr[i]
            out[f'out_{tag}_valid{i}'] = pv[i]
            out[f'out_{tag}_ref{i}'] = pref[i]
            out[f'out_{tag}_real{i}'] = preal[i]
    G.save('obb_proposals', in_boxes0=boxes[0], in_boxes1=boxes[1], in_real0=real[0], in_real1=real[1], **out)


def gen_strong_aug():
    # detectors/syn_images_generator_v2.py:217-343, draws injected, TF.rotate -> recorded no-op (pixels not stored)
    syn = L.ref_obb('models.detectors.syn_images_generator_v2')
    g = torch.Generator().manual_seed(104)
    B, H, W = 4, 96, 96
    img = torch.round(torch.randn(B, 3, H, W, generator=g) * 30)
    flips = ['horizontal', 'vertical', 'diagonal', 'None']
    angles = [3, 19, 11, 7]
    scales = [0.8, 1.2, 1.0, 1.1]
    pb = [rand_rboxes(g, 8, 96, 4, 20) for _ in range(B)]
    pp = [b[:, :2].clone() + torch.randn(8, 2, generator=g) for b in pb]
    gt_pts = [torch.rand(8, 2, generator=g) * 96 for _ in range(B)]
    lab = [torch.randint(0, 9, (8,), generator=g) for _ in range(B)]
    fq, aq, sq = list(flips), list(angles), list(scales)
    seen = []

    class _TF:
        @staticmethod
        def rotate(im, angle, fill=0):
            seen.append(int(angle))
            return im
    real = (random.choice, np.random.randint, np.random.uniform, syn.TF)
    random.choice = lambda seq_: fq.pop(0)
    np.random.randint = lambda a_, b_: aq.pop(0)
    np.random.uniform = lambda a_, b_: sq.pop(0)
    syn.TF = _TF
    try:
        _, _, agp, agl, app, apl, apb = syn.strong_augmentation(img, gt_pts, lab, pp, [l.clone() for l in lab], pb, 'le90')
    finally:
        random.choice, np.random.randint, np.random.uniform, syn.TF = real
    assert seen == angles
    out = {}
    for i in range(B):
        out[f'in_gt_points{i}'], out[f'in_labels{i}'] = gt_pts[i], lab[i]
        out[f'in_pseudo_points{i}'], out[f'in_pseudo_bboxes{i}'] = pp[i], pb[i]
        out[f'out_gt_points{i}'], out[f'out_gt_labels{i}'] = agp[i], agl[i]
        out[f'out_pseudo_points{i}'], out[f'out_pseudo_labels{i}'], out[f'out_pseudo_bboxes{i}'] = app[i], apl[i], apb[i]
    G.save('obb_strong_aug', flips=np.array(flips), angles=np.array(angles), scales=np.array(scales), hw=np.array([H, W]), **out)


def build_fake_head():
    """`self` stand-in with the configuration of sodaa_fcos_pointteacher_1x.py:55-93,107-135 and the
    reference's own methods attached."""
    hd = L.ref_obb('models.dense_heads.rotated_fcos_head_p2rb_ts')
    bld = L.ref('core.bbox.builder')
    cd = L.ref_obb('core.bbox.coder.distance_angle_point_coder')

    class FakeHead:
        pass
    for n in ['_gnerate_pseudo_single', '_get_target_single', '_get_target_pseudo_single', 'centerness_target',
              'gfocal_loss', 'mil_bag_selection_single', 'mil_bag_selection', 'mil_bag_training']:
        setattr(FakeHead, n, getattr(hd.TS_P2RBRotatedFCOSHead, n))
    h = FakeHead()
    h.num_classes = h.cls_out_channels = 9

    def topk(n, w):
        return bld.build_assigner(dict(type='TopkAssigner', num_pre=n, topk=n, cls_cost=dict(type='FocalLossCost', weight=w),
                                       reg_cost=dict(type='PointCost', mode='L1', weight=1.0)))
    h.assigner, h.pseudo_assigner, h.syn_assigner = topk(1, 1.0), topk(3, 0.0), topk(3, 0.0)
    h.fuse_assigner = bld.build_assigner(dict(type='FUSETopkAssigner', num_pre=5, topk=3,
                                              cls_cost=dict(type='FocalLossCost', weight=1.0),
                                              reg_cost=dict(type='PointCost', mode='L1', weight=1.0),
                                              location_cost=dict(type='InsiderCost', weight=1.0)))
    h.bbox_coder = cd.DistanceAnglePointCoder(angle_version='le90')
    h.beta, h.topk = 0.25, 3
    return h, hd


def gen_head():
    h, hd = build_fake_head()
    real_iou = hd.rbbox_overlaps
    hd.rbbox_overlaps = lambda a, b, mode='iou', is_aligned=False: a.new_zeros(a.shape[0])   # mmcv op: log value only
    try:
        for name, H, W, n_gt, seed in (('small', 20, 20, 7, 111), ('mid', 40, 40, 40, 112)):
            g = torch.Generator().manual_seed(seed)
            size = 8 * H
            pts = G.grid_points(H, W)
            P = pts.shape[0]
            cls = torch.randn(P, 9, generator=g) * 2 - 2
            reg = torch.cat([torch.rand(P, 4, generator=g) * 20, torch.randn(P, 1, generator=g) * 0.6], 1)
            ctr = torch.randn(P, 1, generator=g)
            gtb = tie_free_rboxes(g, n_gt, size, pts)
            lab = torch.randint(0, 9, (n_gt,), generator=g)
            gp = gtb[:, :2].contiguous()
            # :871-929
            pb, pp, pl, _, valid = h._gnerate_pseudo_single(gp, lab, gtb, cls, reg, ctr, dict(), None, 0.0, pts, [P])
            # :671-716 (synthetic boxes) and :781-843 (pseudo boxes, jittered so targets are not trivial)
            l_syn, t_syn, a_syn = h._get_target_single(gtb, cls, reg, ctr, pts, [P])
            pb2 = tie_free_rboxes(g, n_gt, size, pts)
            lr, tr_, ar, lc, wc = h._get_target_pseudo_single(gp, lab, pb2[:, :2].contiguous(), lab, pb2, cls, reg, ctr,
                                                             dict(ori_filename='x'), None, None, pts, [P], False)
            pos = lr < 9
            G.save(f'obb_head_{name}', in_points=pts, in_cls=cls, in_reg=reg, in_gt_bboxes=gtb, in_gt_labels=lab,
                   out_pseudo_bboxes=pb, out_pseudo_points=pp, out_valid=valid,
                   out_syn_labels=l_syn, out_syn_bbox_targets=t_syn, out_syn_angle_targets=a_syn,
                   in_pseudo_bboxes2=pb2, out_labels_reg=lr, out_bbox_targets=tr_, out_angle_targets=ar, out_labels_cls=lc,
                   out_centerness=h.centerness_target(tr_[pos]))
    finally:
        hd.rbbox_overlaps = real_iou
    # :1198-1283 (selection + bag loss with the 0.25 / 0.75 weights).  The OBB tree was written against an
    # mmdet whose _expand_onehot_labels takes `ignore_index` (:1270); the HBB fork's copy (the only mmdet
    # in /root/reference) builds the same one-hot without that keyword - adapt the call, not the arithmetic.
    hbb_onehot = hd._expand_onehot_labels
    hd._expand_onehot_labels = lambda l, w, c, ignore_index=None: hbb_onehot(l, w, c)
    g = torch.Generator().manual_seed(113)
    counts, U1, U2, C = [9, 6], 1, 25, 9
    N = sum(counts)
    cls = torch.randn(N, U1, U2, C, generator=g, requires_grad=True)
    ins = torch.randn(N, U1, U2, C, generator=g, requires_grad=True)
    valid = (torch.rand(N * U1 * U2, 1, generator=g) > 0.15)
    valid[:U2] = False                                   # one bag with no valid member
    labels = [torch.randint(0, C, (c,), generator=g) for c in counts]
    pseudo = [rand_rboxes(g, c, 200) for c in counts]
    bags = [p[:, None].repeat(1, U1 * U2, 1).reshape(-1, 5) + torch.randn(p.shape[0] * U1 * U2, 5, generator=g) for p in pseudo]
    bags[0][U2:2 * U2, 0] = 260.0                        # clamp (w = 240, h = 200)
    bags[0][2 * U2:3 * U2, 1] = -4.0
    neg_cls = torch.randn(80, C, generator=g, requires_grad=True)
    neg_w = [torch.rand(40, generator=g) > 0.3, torch.rand(40, generator=g) > 0.3]
    res = dict(cls_score=cls, ins_score=ins, extensive_bags=bags,
               extensive_bags_valid=list(torch.split(valid, [c * U1 * U2 for c in counts])), neg_cls_score=neg_cls)
    metas = [dict(img_shape=(200, 240, 3))] * 2
    merged = h.mil_bag_selection(res, metas, pseudo, labels)
    loss = h.mil_bag_training(res, labels, neg_w)
    gr = torch.autograd.grad(loss, [cls, ins, neg_cls])
    loss_pos = h.mil_bag_training(res, labels, None)
    hd._expand_onehot_labels = hbb_onehot
    G.save('obb_mil', in_cls=cls, in_ins=ins, in_valid=valid, in_labels=torch.cat(labels), in_pseudo=torch.cat(pseudo),
           in_bags=torch.cat(bags), in_neg_cls=neg_cls, in_neg_w=torch.cat(neg_w), counts=np.array(counts),
           out_merged=torch.cat(merged), out_loss=loss, out_loss_pos_only=loss_pos, out_grad_cls=gr[0], out_grad_ins=gr[1],
           out_grad_neg=gr[2])


def gen_eval_map():
    """core/evaluation/eval_map.py eval_rbbox_map (:126-246) + tpfp_default (:12-94) + mmdet average_precision, run on a
    seeded scene.  mmcv's box_iou_rotated is replaced by the fp64 polygon-clip IoU of oracle/ref_ops.py (unpinned)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import ref_ops as R
    # tpfp_default hands the (m, 6) detection rows (score included) to the op; the stand-in reads the five box columns
    em = L.install_obb_eval(lambda a, b: R.box_iou_rotated(a[:, :5], b[:, :5]))
    rng = np.random.RandomState(7)
    K, n_img = 3, 7
    dets, anns, flat = [], [], {}
    for i in range(n_img):
        NG = rng.randint(0, 10)
        c = rng.rand(NG, 2) * 300 + 30
        wh = np.exp(rng.randn(NG, 2) * 0.5 + np.log(20.0)).clip(4, 90)
        a = rng.rand(NG, 1) * np.pi - np.pi / 2
        gb = np.concatenate([c, wh, a], 1).astype(np.float32)
        lab = rng.randint(0, K, NG)
        ign = rng.rand(NG) < 0.15
        ann = dict(bboxes=gb[~ign], labels=lab[~ign], bboxes_ignore=gb[ign], labels_ignore=lab[ign])
        anns.append(ann)
        per = []
        for k in range(K):
            g = gb[lab == k]
            keep = rng.rand(len(g)) < 0.8
            d = g[keep] + rng.randn(int(keep.sum()), 5).astype(np.float32) * np.array([1.5, 1.5, 1.5, 1.5, 0.05], np.float32)
            d = np.concatenate([d, d[: len(d) // 2] + 0.5], 0)
            nf = rng.randint(0, 4)
            f = np.concatenate([rng.rand(nf, 2) * 300 + 30, np.exp(rng.randn(nf, 2) * 0.5 + np.log(20.0)).clip(4, 90),
                                rng.rand(nf, 1) * np.pi - np.pi / 2], 1)
            d = np.concatenate([d, f], 0).astype(np.float32)
            sc = (rng.rand(len(d), 1) * 0.9 + 0.05).astype(np.float32)      # distinct scores (ties would be ordered by numpy's quicksort)
            per.append(np.concatenate([d, sc], 1).astype(np.float32))
        dets.append(per)
    out = {}
    for thr in (0.5, 0.25):
        mean_ap, res = em.eval_rbbox_map(dets, anns, iou_thr=thr, nproc=1)
        out[f'out_map_{int(thr * 100)}'] = np.float64(mean_ap)
        for k in range(K):
            out[f'out_ap_{int(thr * 100)}_{k}'] = np.float64(res[k]['ap'])
            out[f'out_recall_{int(thr * 100)}_{k}'] = np.asarray(res[k]['recall'], np.float64)
            out[f'out_precision_{int(thr * 100)}_{k}'] = np.asarray(res[k]['precision'], np.float64)
            out[f'out_num_gts_{int(thr * 100)}_{k}'] = np.int64(res[k]['num_gts'])
    for i in range(n_img):
        for key in ('bboxes', 'labels', 'bboxes_ignore', 'labels_ignore'):
            out[f'in_ann{i}_{key}'] = anns[i][key]
        for k in range(K):
            out[f'in_det{i}_{k}'] = dets[i][k]
    allsc = np.concatenate([d[:, -1] for per in dets for d in per])
    assert len(np.unique(allsc)) == len(allsc), 'score ties: the reference would order them with an unstable sort'
    G.save('obb_eval_map', n_img=np.int64(n_img), num_classes=np.int64(K), **out)


def gen_sodaa_eval():
    """datasets/sodaa_eval/sodaa_eval.py SODAAeval (evaluate -> accumulate -> summarize, the class behind
    SODAADataset.evaluate, sodaa.py:377-470) run on a seeded scene of whole images.  mmcv's box_iou_rotated is replaced by the
    fp64 polygon-clip IoU of oracle/ref_ops.py rounded to float32 (what the fp32 op returns); `np.float`, removed from
    numpy, is aliased to float for the two `astype(dtype=np.float)` calls.  Scene: object sizes spread over the four
    SODA area bins, jittered / duplicated / spurious detections with distinct scores, one image without objects and one
    category that never occurs; image 0's first annotation and the first detection exercise the id-0 quirk."""
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import ref_ops as R
    L.install_obb()
    import importlib
    import types
    if not hasattr(np, 'float'):
        np.float = float
    sys.modules['mmcv.ops'].box_iou_rotated = lambda a, b: torch.from_numpy(
        np.asarray(R.box_iou_rotated(a.numpy().astype(np.float64), b.numpy().astype(np.float64)), np.float64).astype(np.float32).reshape(len(a), len(b)))
    pkg = types.ModuleType('mmrotate.datasets.sodaa_eval')
    pkg.__path__ = [os.path.join(L.OBB, 'mmrotate', 'datasets', 'sodaa_eval')]
    sys.modules['mmrotate.datasets.sodaa_eval'] = pkg
    if 'mmrotate.datasets' not in sys.modules:
        L._pkg('mmrotate.datasets', os.path.join(L.OBB, 'mmrotate', 'datasets'))
    se = importlib.import_module('mmrotate.datasets.sodaa_eval.sodaa_eval')
    rng = np.random.RandomState(11)
    K, n_img = 4, 6
    anns, dets = [], []
    for i in range(n_img):
        NG = 0 if i == 3 else rng.randint(6, 16)
        c = rng.rand(NG, 2) * 500 + 40
        side = np.exp(rng.uniform(np.log(5.0), np.log(46.0), (NG, 1)))
        wh = side * np.exp(rng.randn(NG, 2) * 0.2)
        a = rng.rand(NG, 1) * np.pi - np.pi / 2
        gb = np.concatenate([c, wh, a], 1).astype(np.float32)
        lab = rng.randint(0, K - 1, NG)                                      # category K-1 never occurs
        anns.append(dict(bboxes=gb, labels=lab.astype(np.int64)))
        per = []
        for k in range(K):
            g = gb[lab == k]
            keep = rng.rand(len(g)) < 0.85
            d = g[keep] + rng.randn(int(keep.sum()), 5).astype(np.float32) * np.array([0.8, 0.8, 0.8, 0.8, 0.04], np.float32)
            d = np.concatenate([d, d[: len(d) // 2] + 0.4], 0)
            nf = rng.randint(0, 4)
            f = np.concatenate([rng.rand(nf, 2) * 500 + 40, np.exp(rng.uniform(np.log(5.0), np.log(46.0), (nf, 2))),
                                rng.rand(nf, 1) * np.pi - np.pi / 2], 1)
            d = np.concatenate([d, f], 0).astype(np.float32)
            sc = (rng.rand(len(d), 1) * 0.9 + 0.05).astype(np.float32)
            per.append(np.concatenate([d, sc], 1).astype(np.float32))
        dets.append(per)
    allsc = np.concatenate([d[:, -1] for per in dets for d in per])
    assert len(np.unique(allsc)) == len(allsc)
    assert len(dets[0][0]) > 0 and len(anns[0]['labels']) > 0
    out = {}
    for tag, thrs in (('default', None), ('t25', np.array([0.25]))):
        ev = se.SODAAeval(anns, dets, numCats=K, nproc=0)
        if thrs is not None:
            ev.params.iouThrs = thrs
        ev.evaluate()
        ev.accumulate()
        ev.summarize()
        out[f'out_{tag}_stats'] = np.asarray(ev.stats, np.float64)
        out[f'out_{tag}_precision'] = ev.eval['precision']
        out[f'out_{tag}_recall'] = ev.eval['recall']
    for i in range(n_img):
        out[f'in_ann{i}_bboxes'], out[f'in_ann{i}_labels'] = anns[i]['bboxes'], anns[i]['labels']
        for k in range(K):
            out[f'in_det{i}_{k}'] = dets[i][k]
    G.save('obb_sodaa_eval', n_img=np.int64(n_img), num_classes=np.int64(K), **out)


def gen_aitod_eval_cocofork():
    """Pins the AI-TOD evaluator configuration of row N1 (5 AI-TOD area ranges, maxDets 100/300/1500, IoU 0.50:0.95) through the
    COCOeval fork that IS vendored in the reference: datasets/sodaa_eval/sodaa_eval.py run with those parameters on
    axis-aligned boxes written as (cx, cy, w, h, 0) - the IoU of two oriented boxes with angle 0 is the axis-aligned IoU
    (the stand-in for mmcv's op computes exactly that, in float32).  The scene keeps the fork's id-0 quirk silent (annotation
    0 is a far-away box no detection touches, detection 0 is a false positive), so its arrays are what pycocotools' COCOeval
    (the algorithm `aitodpycocotools` forks, absent here) gives for non-crowd data.  Coordinates are multiples of 1/4 so that
    the (cx, cy, w, h) <-> (x1, y1, x2, y2) conversion is exact in float32."""
    L.install_obb()
    import importlib
    import types
    if not hasattr(np, 'float'):
        np.float = float

    def hbb_iou(a, b):
        a, b = a.numpy().astype(np.float32), b.numpy().astype(np.float32)
        ax1, ay1, ax2, ay2 = a[:, 0] - a[:, 2] / 2, a[:, 1] - a[:, 3] / 2, a[:, 0] + a[:, 2] / 2, a[:, 1] + a[:, 3] / 2
        bx1, by1, bx2, by2 = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2
        w = np.minimum(ax2[:, None], bx2[None]) - np.maximum(ax1[:, None], bx1[None])
        h = np.minimum(ay2[:, None], by2[None]) - np.maximum(ay1[:, None], by1[None])
        inter = np.where((w <= 0) | (h <= 0), np.float32(0), w * h).astype(np.float32)
        ua = ((ax2 - ax1) * (ay2 - ay1))[:, None] + ((bx2 - bx1) * (by2 - by1))[None] - inter
        return torch.from_numpy(np.where(inter > 0, inter / ua, np.float32(0)).astype(np.float32))
    sys.modules['mmcv.ops'].box_iou_rotated = hbb_iou
    if 'mmrotate.datasets' not in sys.modules:
        L._pkg('mmrotate.datasets', os.path.join(L.OBB, 'mmrotate', 'datasets'))
    pkg = types.ModuleType('mmrotate.datasets.sodaa_eval')
    pkg.__path__ = [os.path.join(L.OBB, 'mmrotate', 'datasets', 'sodaa_eval')]
    sys.modules['mmrotate.datasets.sodaa_eval'] = pkg
    sys.modules.pop('mmrotate.datasets.sodaa_eval.sodaa_eval', None)
    se = importlib.import_module('mmrotate.datasets.sodaa_eval.sodaa_eval')
    rng = np.random.RandomState(23)
    K, n_img = 3, 5
    q = lambda v: np.round(np.asarray(v) * 4) / 4                           # noqa: E731
    anns, dets = [], []
    for i in range(n_img):
        NG = rng.randint(8, 30)
        c = q(rng.rand(NG, 2) * 600 + 60)
        wh = np.round(np.exp(rng.uniform(np.log(3.0), np.log(60.0), (NG, 2))))          # verytiny .. medium
        lab = rng.randint(0, K, NG)
        if i == 0:
            c[0], wh[0], lab[0] = [5000.0, 5000.0], [10.0, 10.0], 0                      # annotation id 0: out of reach
        gb = np.concatenate([c, wh, np.zeros((NG, 1))], 1).astype(np.float32)
        anns.append(dict(bboxes=gb, labels=lab.astype(np.int64)))
        per = []
        for k in range(K):
            g = gb[lab == k]
            g = g[g[:, 0] < 4000]
            keep = rng.rand(len(g)) < 0.85
            d = g[keep].copy()
            d[:, :2] += q(rng.randn(len(d), 2) * 1.0)
            d[:, 2:4] = np.maximum(d[:, 2:4] + np.round(rng.randn(len(d), 2) * 0.8), 1)
            nf = 170 if (i == 1 and k == 0) else rng.randint(0, 6)                      # > 100 detections in one segment
            f = np.concatenate([q(rng.rand(nf, 2) * 600 + 60), np.round(np.exp(rng.uniform(np.log(3.0), np.log(60.0), (nf, 2)))),
                                np.zeros((nf, 1))], 1)
            d = np.concatenate([d, d[: len(d) // 2] + np.array([0.5, 0.25, 0, 0, 0]), f], 0).astype(np.float32)
            if i == 0 and k == 0:
                d = np.concatenate([np.array([[3000.0, 3000.0, 9.0, 9.0, 0.0]], np.float32), d], 0)   # detection id 0: a false positive
            per.append(d)
        dets.append(per)
    n_tot = sum(len(d) for per in dets for d in per)
    sc = (rng.permutation(n_tot).astype(np.float32) + 1) / np.float32(n_tot + 1)        # distinct scores
    o = 0
    for per in dets:
        for k in range(K):
            n = len(per[k])
            per[k] = np.concatenate([per[k], sc[o:o + n, None]], 1).astype(np.float32)
            o += n
    ev = se.SODAAeval(anns, dets, numCats=K, nproc=0)
    ev.params.areaRng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 8 ** 2], [8 ** 2, 16 ** 2], [16 ** 2, 32 ** 2], [32 ** 2, 1e5 ** 2]]
    ev.params.areaRngLbl = ['all', 'verytiny', 'tiny', 'small', 'medium']
    ev.params.maxDets = [100, 300, 1500]
    ev.evaluate()
    ev.accumulate()
    out = dict(out_precision=ev.eval['precision'], out_recall=ev.eval['recall'])
    for i in range(n_img):
        b = anns[i]['bboxes']
        out[f'in_gt{i}_xyxy'] = np.stack([b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2], 1)
        out[f'in_gt{i}_labels'] = anns[i]['labels']
        for k in range(K):
            d = dets[i][k]
            out[f'in_det{i}_{k}'] = np.concatenate([np.stack([d[:, 0] - d[:, 2] / 2, d[:, 1] - d[:, 3] / 2, d[:, 0] + d[:, 2] / 2,
                                                               d[:, 1] + d[:, 3] / 2], 1), d[:, 5:6]], 1).astype(np.float32)
    G.save('aitod_eval_cocofork', n_img=np.int64(n_img), num_classes=np.int64(K), **out)


def gen_eval_decode():
    """rotated_fcos_head_p2rb_ts.py get_bboxes / _get_bboxes_single (:931-1083) up to the rotated NMS (an mmcv op): per level
    top `nms_pre` by the maximum class score, DistanceAnglePointCoder.decode with clipping, rescale of the first four
    columns, background column.  `multiclass_nms_rotated` is replaced by a recorder that stores its arguments (its result
    is not used); `prior_generator.grid_priors` - the tree expects a newer mmdet's signature - hands back the level points
    (x * s + s // 2), the same points the training path uses.  -> tests/golden/obb_eval_decode.npz"""
    import types
    L.install_obb()
    hd = L.ref_obb('models.dense_heads.rotated_fcos_head_p2rb_ts')
    cd = L.ref_obb('core.bbox.coder.distance_angle_point_coder')
    rec = []
    hd.multiclass_nms_rotated = lambda b, s, thr, nms, mx, score_factors=None: (rec.append((b.clone(), s.clone(), score_factors.clone())), (b, s))[1]
    f = hd.TS_P2RBRotatedFCOSHead.get_bboxes
    if hasattr(f, '__wrapped__'):
        hd.TS_P2RBRotatedFCOSHead.get_bboxes = f.__wrapped__
    h = object.__new__(hd.TS_P2RBRotatedFCOSHead)
    torch.nn.Module.__init__(h)
    h.num_classes = h.cls_out_channels = 9
    h.bbox_coder = cd.DistanceAnglePointCoder(angle_version='le90')
    g = torch.Generator().manual_seed(6)
    B, sizes, strides = 2, [(20, 24), (10, 12), (5, 6)], [8, 16, 32]
    pts = []
    for (hh, ww), s in zip(sizes, strides):
        ys, xs = torch.meshgrid(torch.arange(hh, dtype=torch.float32), torch.arange(ww, dtype=torch.float32), indexing='ij')
        pts.append(torch.stack((xs.reshape(-1) * s, ys.reshape(-1) * s), dim=-1) + s // 2)
    h.prior_generator = types.SimpleNamespace(grid_priors=lambda sizes_, dtype, device: pts)
    cls = [torch.randn(B, 9, hh, ww, generator=g) * 2 - 1 for hh, ww in sizes]
    reg = [(torch.randn(B, 4, hh, ww, generator=g).abs() * 3 + 0.5) * s for (hh, ww), s in zip(sizes, strides)]
    ang = [torch.randn(B, 1, hh, ww, generator=g) * 0.6 for hh, ww in sizes]
    ctr = [torch.randn(B, 1, hh, ww, generator=g) for hh, ww in sizes]
    metas = [dict(img_shape=(150, 180, 3), scale_factor=np.array([1.5, 1.25, 1.5, 1.25], np.float32)),
             dict(img_shape=(160, 192, 3), scale_factor=np.array([1.0, 1.0, 1.0, 1.0], np.float32))]
    cfg = types.SimpleNamespace(score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=100)
    cfg.get = lambda k, d=None: dict(nms_pre=60).get(k, d)
    out = {}
    for tag, rescale in (('plain', False), ('rescale', True)):
        rec.clear()
        h.get_bboxes(cls, reg, ang, ctr, pts, metas, cfg=cfg, rescale=rescale)
        for b in range(B):
            out[f'out_{tag}_bboxes{b}'], out[f'out_{tag}_scores{b}'], out[f'out_{tag}_factors{b}'] = rec[b]
    for i in range(len(sizes)):
        out[f'in_cls{i}'], out[f'in_reg{i}'], out[f'in_ang{i}'], out[f'in_ctr{i}'], out[f'in_points{i}'] = cls[i], reg[i], ang[i], ctr[i], pts[i]
    G.save('obb_eval_decode', **out)


if __name__ == '__main__':
    L.install_obb()
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    if len(sys.argv) > 1:                       # regenerate selected fixtures only, e.g. `gen_golden_obb.py gen_sodaa_eval`
        for name in sys.argv[1:]:
            print(name)
            globals()[name]()
        print('done')
        sys.exit(0)
    for fn in [gen_transforms, gen_coder, gen_proposals, gen_strong_aug, gen_head, gen_eval_map, gen_sodaa_eval]:
        print(fn.__name__)
        fn()
    print('done')
