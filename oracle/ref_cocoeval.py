"""CPU ORACLE of the AI-TOD COCO-style detection evaluator - TEST INFRASTRUCTURE ONLY.

The reference evaluates through `aitodpycocotools.cocoeval.COCOeval`
(/root/reference/HBB_TOD/mmdet/datasets/aitod.py:10, :100-135), an un-vendored fork of pycocotools that is
not installed here (neither is pycocotools).  PINNED for non-crowd data through the COCOeval fork that IS vendored in the
reference (OBB_TOD/mmrotate/datasets/sodaa_eval/sodaa_eval.py): run with the AI-TOD parameters on axis-aligned boxes at
angle 0 it produced tests/golden/aitod_eval_cocofork.npz (oracle/gen_golden_obb.py gen_aitod_eval_cocofork), which
`evaluate` below reproduces to 1e-12; the crowd / ignore handling and the fork's oLRP extras stay PARITY UNPINNED.  This file restates the published COCOeval
algorithm for boxes (pycocotools/cocoeval.py: evaluateImg, accumulate, summarize) with the AI-TOD
parameterisation the reference's metric table names (aitod.py:117-135):

    area ranges   all [0, 1e5^2], verytiny [0, 8^2], tiny [8^2, 16^2], small [16^2, 32^2], medium [32^2, 1e5^2]
    maxDets       (100, 300, 1500)   (`proposal_nums`, aitod.py:30)
    IoU thrs      0.50:0.05:0.95 by default (the fork's `evaluate` overwrites the argument with [0.25] at
                  aitod.py:64 - a debugging leftover; pass iou_thrs=[0.25] to reproduce it)
    stats         mAP, mAP_25, mAP_50, mAP_75, mAP_vt, mAP_t, mAP_s, mAP_m, AR@100, AR@300, AR@1500,
                  AR_vt@1500, AR_t@1500, AR_s@1500, AR_m@1500          (oLRP of the fork: not restated)

Plain python / numpy loops, one (image, category, area range) at a time, exactly in the order pycocotools
walks them.  Only tests import this file.
"""
import numpy as np

AREAS = (('all', 0.0, 1e5 ** 2), ('verytiny', 0.0, 8.0 ** 2), ('tiny', 8.0 ** 2, 16.0 ** 2), ('small', 16.0 ** 2, 32.0 ** 2),
         ('medium', 32.0 ** 2, 1e5 ** 2))
MAX_DETS = (100, 300, 1500)
REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)


def default_iou_thrs():
    return np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)


def iou_xyxy(d, g, iscrowd):
    """maskApi bbIou on xywh boxes = continuous areas (no +1); crowd gts use the detection's area as the union."""
    D, G = d.shape[0], g.shape[0]
    out = np.zeros((D, G))
    for i in range(D):
        for j in range(G):
            w = min(d[i, 2], g[j, 2]) - max(d[i, 0], g[j, 0])
            h = min(d[i, 3], g[j, 3]) - max(d[i, 1], g[j, 1])
            if w <= 0 or h <= 0:
                continue
            inter = w * h
            da = (d[i, 2] - d[i, 0]) * (d[i, 3] - d[i, 1])
            ga = (g[j, 2] - g[j, 0]) * (g[j, 3] - g[j, 1])
            out[i, j] = inter / (da if iscrowd[j] else da + ga - inter)
    return out


def evaluate_img(dets, gts, gt_area, gt_crowd, gt_ignore, lo, hi, max_det, iou_thrs):
    """cocoeval.py evaluateImg for one (image, category, area range).  dets [D,5] = xyxy + score."""
    G, D = gts.shape[0], dets.shape[0]
    if G == 0 and D == 0:
        return None
    g_ig = (gt_ignore.astype(bool) | gt_crowd.astype(bool) | (gt_area < lo) | (gt_area > hi)).astype(int)
    gtind = np.argsort(g_ig, kind='mergesort')
    dtind = np.argsort(-dets[:, 4], kind='mergesort')[:max_det]
    d, g = dets[dtind], gts[gtind]
    g_ig, crowd = g_ig[gtind], gt_crowd[gtind].astype(bool)
    ious = iou_xyxy(d[:, :4], g, crowd)
    T, Dk = len(iou_thrs), d.shape[0]
    gtm = -np.ones((T, G), int)
    dtm = -np.ones((T, Dk), int)
    dt_ig = np.zeros((T, Dk), bool)
    for ti, t in enumerate(iou_thrs):
        for di in range(Dk):
            iou = min(t, 1 - 1e-10)
            m = -1
            for gi in range(G):
                if gtm[ti, gi] >= 0 and not crowd[gi]:
                    continue
                if m > -1 and g_ig[m] == 0 and g_ig[gi] == 1:
                    break
                if ious[di, gi] < iou:
                    continue
                iou = ious[di, gi]
                m = gi
            if m == -1:
                continue
            dt_ig[ti, di] = bool(g_ig[m])
            dtm[ti, di] = gtind[m]
            gtm[ti, m] = di
    d_area = (d[:, 2] - d[:, 0]) * (d[:, 3] - d[:, 1])
    out_rng = (d_area < lo) | (d_area > hi)
    dt_ig = dt_ig | ((dtm == -1) & out_rng[None, :])
    return dict(scores=d[:, 4], dtm=dtm, dt_ig=dt_ig, g_ig=g_ig)


def evaluate(results, gts, num_classes, iou_thrs=None, max_dets=MAX_DETS, areas=AREAS):
    """results[i][k] = ndarray [n,5] (xyxy, score) of image i, class k (mmdet bbox2result layout);
    gts[i] = dict(bboxes [G,4] xyxy, labels [G], optional area [G], iscrowd [G], ignore [G]).
    Returns (stats dict, precision [T,R,K,A,M], recall [T,K,A,M])."""
    iou_thrs = default_iou_thrs() if iou_thrs is None else np.asarray(iou_thrs, float)
    T, R, K, A, M = len(iou_thrs), len(REC_THRS), num_classes, len(areas), len(max_dets)
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    for k in range(K):
        for a, (_, lo, hi) in enumerate(areas):
            per_img = []
            for i, g in enumerate(gts):
                sel = g['labels'] == k
                gb = g['bboxes'][sel]
                area = g['area'][sel] if 'area' in g else (gb[:, 2] - gb[:, 0]) * (gb[:, 3] - gb[:, 1])
                crowd = g['iscrowd'][sel] if 'iscrowd' in g else np.zeros(len(gb), int)
                ign = g['ignore'][sel] if 'ignore' in g else np.zeros(len(gb), int)
                e = evaluate_img(np.asarray(results[i][k], float).reshape(-1, 5), gb, area, crowd, ign, lo, hi, max_dets[-1],
                                 iou_thrs)
                if e is not None:
                    per_img.append(e)
            if not per_img:
                continue
            for m, md in enumerate(max_dets):
                scores = np.concatenate([e['scores'][:md] for e in per_img])
                inds = np.argsort(-scores, kind='mergesort')
                dtm = np.concatenate([e['dtm'][:, :md] for e in per_img], axis=1)[:, inds]
                dt_ig = np.concatenate([e['dt_ig'][:, :md] for e in per_img], axis=1)[:, inds]
                g_ig = np.concatenate([e['g_ig'] for e in per_img])
                npig = np.count_nonzero(g_ig == 0)
                if npig == 0:
                    continue
                tps = np.logical_and(dtm >= 0, np.logical_not(dt_ig))
                fps = np.logical_and(dtm < 0, np.logical_not(dt_ig))
                tp_sum = np.cumsum(tps, axis=1).astype(float)
                fp_sum = np.cumsum(fps, axis=1).astype(float)
                for t in range(T):
                    tp, fp = tp_sum[t], fp_sum[t]
                    nd = len(tp)
                    rc = tp / npig
                    pr = tp / (fp + tp + np.spacing(1))
                    q = np.zeros((R,))
                    recall[t, k, a, m] = rc[-1] if nd else 0
                    pr = pr.tolist()
                    for j in range(nd - 1, 0, -1):
                        if pr[j] > pr[j - 1]:
                            pr[j - 1] = pr[j]
                    idx = np.searchsorted(rc, REC_THRS, side='left')
                    for ri, pi in enumerate(idx):
                        if pi < nd:
                            q[ri] = pr[pi]
                    precision[t, :, k, a, m] = q

    def ap(thr=None, area=0, md=M - 1):
        s = precision[:, :, :, area, md]
        if thr is not None:
            w = np.where(np.isclose(iou_thrs, thr))[0]
            if len(w) == 0:
                return -1.0
            s = s[w]
        s = s[s > -1]
        return float(np.mean(s)) if s.size else -1.0

    def ar(area=0, md=M - 1):
        s = recall[:, :, area, md]
        s = s[s > -1]
        return float(np.mean(s)) if s.size else -1.0
    stats = {'mAP': ap(), 'mAP_25': ap(.25), 'mAP_50': ap(.5), 'mAP_75': ap(.75), 'mAP_vt': ap(area=1), 'mAP_t': ap(area=2),
             'mAP_s': ap(area=3), 'mAP_m': ap(area=4), f'AR@{max_dets[0]}': ar(md=0), f'AR@{max_dets[1]}': ar(md=1),
             f'AR@{max_dets[2]}': ar(md=2), f'AR_vt@{max_dets[2]}': ar(1), f'AR_t@{max_dets[2]}': ar(2),
             f'AR_s@{max_dets[2]}': ar(3), f'AR_m@{max_dets[2]}': ar(4)}
    return stats, precision, recall


# ------------------------------------------------------------------------------------------------
# DOTA-style mAP of the oriented tree (config 5: `evaluation = dict(metric='mAP')`):
# OBB_TOD/mmrotate/core/evaluation/eval_map.py eval_rbbox_map (:126-246) + tpfp_default (:12-94) +
# HBB_TOD/mmdet/core/evaluation/mean_ap.py average_precision (mode '11points' by default, 'area' on request).  Pinned by
# tests/golden/obb_eval_map.npz, captured from the reference's own file (only mmcv's rotated IoU replaced).
# ------------------------------------------------------------------------------------------------
def average_precision_area(recalls, precisions):
    mrec = np.hstack(([0.0], recalls, [1.0]))
    mpre = np.hstack(([0.0], precisions, [0.0]))
    for i in range(len(mpre) - 1, 0, -1):
        mpre[i - 1] = max(mpre[i - 1], mpre[i])
    ind = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[ind + 1] - mrec[ind]) * mpre[ind + 1]))


def average_precision_11points(recalls, precisions):
    """mean_ap.py average_precision mode '11points' (VOC07) - the default of eval_rbbox_map (use_07_metric=True)."""
    ap = 0.0
    for thr in np.arange(0, 1 + 1e-3, 0.1):
        precs = precisions[recalls >= thr]
        ap += precs.max() if precs.size > 0 else 0
    return float(ap / 11)


def eval_rbbox_map(det_results, annotations, iou_thr=0.5, iou_fn=None, use_07_metric=True):
    """det_results[i][k] = [n,6] (cx,cy,w,h,a,score); annotations[i] = dict(bboxes [G,5], labels [G],
    bboxes_ignore, labels_ignore).  Returns (mAP, per-class list of dict(ap, recall, precision, num_gts))."""
    from . import ref_ops as R
    iou_fn = iou_fn or (lambda a, b: R.box_iou_rotated(a, b).numpy())
    import torch
    K = len(det_results[0])
    out = []
    for k in range(K):
        tps, fps, scores, num_gts = [], [], [], 0
        for dets, ann in zip(det_results, annotations):
            d = np.asarray(dets[k], np.float32).reshape(-1, 6)
            g = np.asarray(ann['bboxes'], np.float32).reshape(-1, 5)[np.asarray(ann['labels']) == k]
            gi = np.zeros((0, 5), np.float32)
            if ann.get('labels_ignore', None) is not None:
                gi = np.asarray(ann['bboxes_ignore'], np.float32).reshape(-1, 5)[np.asarray(ann['labels_ignore']) == k]
            num_gts += g.shape[0]
            ignore = np.concatenate([np.zeros(len(g), bool), np.ones(len(gi), bool)])
            allg = np.vstack([g, gi])
            tp, fp = np.zeros(len(d), np.float32), np.zeros(len(d), np.float32)
            if allg.shape[0] == 0:
                fp[:] = 1
            elif len(d):
                ious = np.asarray(iou_fn(torch.from_numpy(d[:, :5]).float(), torch.from_numpy(allg).float()))
                imax, iarg = ious.max(1), ious.argmax(1)
                covered = np.zeros(len(allg), bool)
                for i in np.argsort(-d[:, -1]):
                    if imax[i] >= iou_thr:
                        m = iarg[i]
                        if not ignore[m]:
                            if not covered[m]:
                                covered[m] = True
                                tp[i] = 1
                            else:
                                fp[i] = 1
                    else:
                        fp[i] = 1
            tps.append(tp); fps.append(fp); scores.append(d[:, -1])
        sc = np.concatenate(scores)
        order = np.argsort(-sc)
        tp = np.cumsum(np.concatenate(tps)[order])
        fp = np.cumsum(np.concatenate(fps)[order])
        eps = np.finfo(np.float32).eps
        # dtypes as in the reference: tp/fp are float32, num_gts an int64 array -> recalls float64, precisions float32
        # (recall values such as 3/10 sit exactly on the 11-point thresholds, so the dtype decides `>=`)
        rec = (tp / np.maximum(np.array([num_gts], dtype=int)[:, np.newaxis], eps))[0]
        prec = tp / np.maximum(tp + fp, eps)
        ap = average_precision_11points(rec, prec) if use_07_metric else average_precision_area(rec, prec)
        out.append(dict(ap=ap, recall=rec, precision=prec, num_gts=num_gts))
    aps = [r['ap'] for r in out if r['num_gts'] > 0]
    return (float(np.mean(aps)) if aps else 0.0), out
