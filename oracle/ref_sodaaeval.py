"""CPU ORACLE of the SODA-A evaluation protocol - TEST INFRASTRUCTURE ONLY (only tests import this file).

Restates /root/reference/OBB_TOD/mmrotate/datasets/sodaa_eval/sodaa_eval.py (SODAAeval: _prepare :104-150,
computeIoU :246-273, evaluateImg :348-437, accumulate :440-559, summarize :561-655, SODAAParams :662-693), the evaluator
behind `SODAADataset.evaluate` (sodaa.py:377-470) of config 5.  PINNED: tests/golden/obb_sodaa_eval.npz holds the
precision / recall arrays and the 12 statistics the reference's own class produced on seeded inputs
(oracle/gen_golden_obb.py `sodaa_eval`; only mmcv's `box_iou_rotated`, absent here, is replaced by the fp64 polygon-clip
IoU of oracle/ref_obb.py rounded to float32).

Kept exactly, because they change numbers:
* instance ids start at 0 for ground truths AND detections (:108, :121) while the matching code treats 0 as "none"
  (`gtm[tind, gind] > 0`, `dtm == 0`, `np.logical_and(dtm, ...)`): a detection matched to the dataset's first ground
  truth counts as unmatched, and the dataset's first detection never blocks the ground truth it takes;
* areas are w*h of the float32 box arrays; there are no crowd / ignore flags; maxDets = [20000].
"""
import numpy as np

AREAS = (('Small', 0 ** 2, 32 ** 2), ('eS', 0 ** 2, 12 ** 2), ('rS', 12 ** 2, 20 ** 2), ('gS', 20 ** 2, 32 ** 2), ('Normal', 32 ** 2, 40 * 50))
MAX_DETS = (20000,)
REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
NAMES = ('AP', 'AP_50', 'AP_75', 'AP_eS', 'AP_rS', 'AP_gS', 'AP_Normal', 'AR@20000', 'AR_eS@20000', 'AR_rS@20000', 'AR_gS@20000',
         'AR_Normal@20000')


def default_iou_thrs():
    return np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)


def evaluate(annotations, results, num_cats, iou_fn, iou_thrs=None, areas=AREAS, max_dets=MAX_DETS):
    """annotations[i] = dict(bboxes [G,5] float32, labels [G]); results[i][k] = ndarray [n,6] (cx,cy,w,h,a,score);
    iou_fn(d [D,5], g [G,5]) -> [D,G].  Returns (stats [12], precision [T,R,K,A,M], recall [T,K,A,M])."""
    iou_thrs = default_iou_thrs() if iou_thrs is None else np.asarray(iou_thrs, float)
    gts, dts = {}, {}
    ins = 0
    for i, ann in enumerate(annotations):
        for j in range(len(ann['labels'])):
            b = ann['bboxes'][j]
            gts.setdefault((i, int(ann['labels'][j])), []).append(dict(bbox=b, area=b[2] * b[3], id=ins))
            ins += 1
    ins = 0
    for i, res in enumerate(results):
        for k, cat in enumerate(res):
            for r in range(len(cat)):
                b = cat[r, :5]
                dts.setdefault((i, k), []).append(dict(bbox=b, score=cat[r, -1], area=b[2] * b[3], id=ins))
                ins += 1
    I, K, A, T, R, M = len(annotations), num_cats, len(areas), len(iou_thrs), len(REC_THRS), len(max_dets)
    max_det = max_dets[-1]
    ious = {}
    for i in range(I):
        for k in range(K):
            gt, dt = gts.get((i, k), []), dts.get((i, k), [])
            if not gt and not dt:
                ious[i, k] = []
                continue
            order = np.argsort([-d['score'] for d in dt], kind='mergesort')
            dt = [dt[o] for o in order][:max_det]
            ious[i, k] = iou_fn(np.array([d['bbox'] for d in dt], np.float32).reshape(-1, 5),
                                np.array([g['bbox'] for g in gt], np.float32).reshape(-1, 5))

    def evaluate_img(i, k, rng):
        gt, dt = gts.get((i, k), []), dts.get((i, k), [])
        if not gt and not dt:
            return None
        ig = [1 if (g['area'] < rng[0] or g['area'] > rng[1]) else 0 for g in gt]
        gtind = np.argsort(ig, kind='mergesort')
        gt = [gt[o] for o in gtind]
        dtind = np.argsort([-d['score'] for d in dt], kind='mergesort')
        dt = [dt[o] for o in dtind[:max_det]]
        iou_m = ious[i, k][:, gtind] if len(ious[i, k]) > 0 else ious[i, k]
        G, D = len(gt), len(dt)
        gtm, dtm = np.zeros((T, G)), np.zeros((T, D))
        gt_ig = np.array([ig[o] for o in gtind])
        dt_ig = np.zeros((T, D))
        if not len(iou_m) == 0:
            for ti, t in enumerate(iou_thrs):
                for di, d in enumerate(dt):
                    iou = min([t, 1 - 1e-10])
                    m = -1
                    for gi in range(G):
                        if gtm[ti, gi] > 0:                     # id 0 never blocks
                            continue
                        if m > -1 and gt_ig[m] == 0 and gt_ig[gi] == 1:
                            break
                        if iou_m[di, gi] < iou:
                            continue
                        iou = iou_m[di, gi]
                        m = gi
                    if m == -1:
                        continue
                    dt_ig[ti, di] = gt_ig[m]
                    dtm[ti, di] = gt[m]['id']
                    gtm[ti, m] = d['id']
        a = np.array([d['area'] < rng[0] or d['area'] > rng[1] for d in dt]).reshape((1, len(dt)))
        dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(a, T, 0)))
        return dict(scores=[d['score'] for d in dt], dtm=dtm, dt_ig=dt_ig, gt_ig=gt_ig)

    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    for k in range(K):
        for a, (_, lo, hi) in enumerate(areas):
            E = [e for e in (evaluate_img(i, k, (lo, hi)) for i in range(I)) if e is not None]
            if not E:
                continue
            for m, md in enumerate(max_dets):
                scores = np.concatenate([e['scores'][0:md] for e in E])
                inds = np.argsort(-scores, kind='mergesort')
                dtm = np.concatenate([e['dtm'][:, 0:md] for e in E], axis=1)[:, inds]
                dt_ig = np.concatenate([e['dt_ig'][:, 0:md] for e in E], axis=1)[:, inds]
                gt_ig = np.concatenate([e['gt_ig'] for e in E])
                npig = np.count_nonzero(gt_ig == 0)
                if npig == 0:
                    continue
                tps = np.logical_and(dtm, np.logical_not(dt_ig))
                fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                tp_sum, fp_sum = np.cumsum(tps, axis=1).astype(float), np.cumsum(fps, axis=1).astype(float)
                for t in range(T):
                    tp, fp = tp_sum[t], fp_sum[t]
                    nd = len(tp)
                    rc = tp / npig
                    pr = (tp / (fp + tp + np.spacing(1))).tolist()
                    recall[t, k, a, m] = rc[-1] if nd else 0
                    for j in range(nd - 1, 0, -1):
                        if pr[j] > pr[j - 1]:
                            pr[j - 1] = pr[j]
                    q = np.zeros((R,))
                    for ri, pi in enumerate(np.searchsorted(rc, REC_THRS, side='left')):
                        if pi >= nd:
                            break
                        q[ri] = pr[pi]
                    precision[t, :, k, a, m] = q

    def summ(ap, thr=None, area=0):
        s = precision if ap else recall
        if thr is not None:
            s = s[np.where(thr == iou_thrs)[0]]
        s = s[:, :, :, area, 0] if ap else s[:, :, area, 0]
        return -1 if len(s[s > -1]) == 0 else np.mean(s[s > -1])
    stats = np.array([summ(1), summ(1, .50), summ(1, .75), summ(1, area=1), summ(1, area=2), summ(1, area=3), summ(1, area=4),
                      summ(0), summ(0, area=1), summ(0, area=2), summ(0, area=3), summ(0, area=4)], dtype=float)
    return stats, precision, recall
