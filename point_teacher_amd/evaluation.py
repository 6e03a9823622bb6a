"""AI-TOD / COCO-style box evaluator on the GPU: the counterpart of
`AITODDataset.evaluate(metric='bbox')` (/root/reference/HBB_TOD/mmdet/datasets/aitod.py:23-246), which hands the
detections to `aitodpycocotools.cocoeval.COCOeval` (un-vendored).  Same inputs as the reference's method (the
per-image, per-class arrays `simple_test` returns), same metric names (`bbox_mAP`, `bbox_mAP_50`, `bbox_mAP_vt`, ...).

Matching (`evaluateImg`) is one HIP launch for the whole dataset (`pt_coco_match`: a wavefront per (image, category),
a lane per (area range, IoU threshold)); accumulation (`accumulate`: global score sort, cumulative TP/FP, precision
envelope, 101-point recall sampling) and `summarize` are a few dozen batched torch ops on the device.

Protocol: area ranges all / verytiny [0,8^2] / tiny [8^2,16^2] / small [16^2,32^2] / medium [32^2,inf), maxDets
(100, 300, 1500) (aitod.py:30), IoU thresholds 0.50:0.05:0.95 unless given.  The fork's `evaluate` overwrites its
`iou_thrs` argument with [0.25] (aitod.py:64, a debugging leftover): pass `iou_thrs=[0.25]` to reproduce that.
The fork's oLRP statistics are not computed.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import hip

AREAS = (('all', 0.0, 1e5 ** 2), ('verytiny', 0.0, 8.0 ** 2), ('tiny', 8.0 ** 2, 16.0 ** 2), ('small', 16.0 ** 2, 32.0 ** 2),
         ('medium', 32.0 ** 2, 1e5 ** 2))
MAX_DETS = (100, 300, 1500)


def _default_iou_thrs():
    return np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)


class AITODEvaluator:
    box_dim = 4                      # xyxy; the SODA-A subclass evaluates (cx, cy, w, h, a)

    @staticmethod
    def _box_area(b):
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def __init__(self, gts, num_classes, device='cuda', areas=AREAS, max_dets=MAX_DETS, iou_thrs=None):
        """gts[i] = dict(bboxes [G,4] xyxy, labels [G], optional area [G], iscrowd [G], ignore [G]) for image i."""
        self.K, self.areas, self.max_dets = num_classes, areas, tuple(max_dets)
        self.iou_thrs = np.asarray(_default_iou_thrs() if iou_thrs is None else iou_thrs, dtype=np.float64)
        self.device = torch.device(device)
        assert len(areas) * len(self.iou_thrs) <= 64, 'area ranges x IoU thresholds must fit one wavefront'
        box, img, lab, area, flags = [], [], [], [], []
        for i, g in enumerate(gts):
            b = np.asarray(g['bboxes'], np.float32).reshape(-1, self.box_dim)
            n = b.shape[0]
            box.append(b)
            img.append(np.full(n, i, np.int64))
            lab.append(np.asarray(g['labels'], np.int64).reshape(-1))
            area.append(np.asarray(g['area'], np.float32).reshape(-1) if 'area' in g else self._box_area(b))
            crowd = np.asarray(g['iscrowd']).reshape(-1).astype(bool) if 'iscrowd' in g else np.zeros(n, bool)
            ign = np.asarray(g['ignore']).reshape(-1).astype(bool) if 'ignore' in g else np.zeros(n, bool)
            flags.append((ign | crowd).astype(np.uint8) | (crowd.astype(np.uint8) << 1))
        self.n_img = len(gts)
        cat = np.concatenate(lab) if lab else np.zeros(0, np.int64)
        key = (np.concatenate(img) if img else np.zeros(0, np.int64)) * self.K + cat
        order = np.argsort(key, kind='stable')                              # segment-major, original order inside
        dev = self.device
        self.gt_key = torch.from_numpy(key[order]).to(dev)
        self.gt_box = torch.from_numpy(np.concatenate(box)[order] if box else np.zeros((0, self.box_dim), np.float32)).to(dev)
        self.gt_area = torch.from_numpy(np.concatenate(area)[order] if area else np.zeros(0, np.float32)).to(dev)
        self.gt_flags = torch.from_numpy(np.concatenate(flags)[order] if flags else np.zeros(0, np.uint8)).to(dev)
        self.gt_cat = torch.from_numpy(cat[order]).to(dev)

    # ------------------------------------------------------------------------------
    def _flatten(self, results):
        """mmdet bbox2result layout -> (boxes, scores, segment key) sorted by (key, -score), stable."""
        box, score, key = [], [], []
        assert len(results) == self.n_img, (len(results), self.n_img)
        for i, per_cls in enumerate(results):
            assert len(per_cls) == self.K
            for k, arr in enumerate(per_cls):
                D = self.box_dim
                arr = np.asarray(arr, np.float32).reshape(-1, D + 1)
                if arr.shape[0]:
                    o = np.argsort(-arr[:, D], kind='mergesort')
                    box.append(arr[o, :D]); score.append(arr[o, D]); key.append(np.full(arr.shape[0], i * self.K + k, np.int64))
        if not box:
            return (torch.zeros((0, self.box_dim), device=self.device), torch.zeros(0, device=self.device),
                    torch.zeros(0, dtype=torch.long, device=self.device))
        dev = self.device
        return (torch.from_numpy(np.concatenate(box)).to(dev), torch.from_numpy(np.concatenate(score)).to(dev),
                torch.from_numpy(np.concatenate(key)).to(dev))

    def evaluate(self, results):
        """-> OrderedDict of the reference's `bbox_*` metrics (+ `precision` [T,R,K,A,M] and `recall` [T,K,A,M] tensors)."""
        dev = self.device
        A, T, K, M = len(self.areas), len(self.iou_thrs), self.K, len(self.max_dets)
        R = 101
        dbox, dscore, dkey = self._flatten(results)
        Nd, Ng = dbox.shape[0], self.gt_box.shape[0]
        # segments = every (image, category) that has detections or ground truths
        keys = torch.unique(torch.cat([dkey, self.gt_key]))
        S = int(keys.numel())
        det_off = torch.searchsorted(dkey, torch.cat([keys, keys[-1:] + 1]) if S else keys).to(torch.int32)
        gt_off = torch.searchsorted(self.gt_key, torch.cat([keys, keys[-1:] + 1]) if S else keys).to(torch.int32)
        L = A * T
        dtm = torch.full((Nd, L), -1, dtype=torch.int32, device=dev)
        dt_ig = torch.zeros((Nd, L), dtype=torch.uint8, device=dev)
        if S and Nd:
            lo = torch.tensor([a[1] for a in self.areas], dtype=torch.float32, device=dev)
            hi = torch.tensor([a[2] for a in self.areas], dtype=torch.float32, device=dev)
            thr = torch.tensor(self.iou_thrs, dtype=torch.float32, device=dev)
            scratch = torch.zeros((max(Ng, 1) * 64,), dtype=torch.uint8, device=dev)
            self._match(dbox.contiguous(), det_off, gt_off, S, lo, hi, A, thr, T, int(self.max_dets[-1]), scratch, dtm, dt_ig)
        # rank of every detection inside its segment (for the maxDets prefixes) and its category
        seg_of = torch.searchsorted(keys, dkey) if Nd else dkey
        rank = torch.arange(Nd, device=dev) - det_off[:-1].long()[seg_of] if Nd else dkey
        dcat = dkey % K
        precision = -torch.ones((T, R, K, A, M), dtype=torch.float64, device=dev)
        recall = -torch.ones((T, K, A, M), dtype=torch.float64, device=dev)
        # numpy's linspace, bit for bit: recall values such as 7/10 tie with the thresholds (cocoeval.py Params.recThrs)
        rec_thrs = torch.from_numpy(np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)).to(dev)
        lo_t = torch.tensor([a[1] for a in self.areas], dtype=torch.float32, device=dev)
        hi_t = torch.tensor([a[2] for a in self.areas], dtype=torch.float32, device=dev)
        g_ign = ((self.gt_flags & 1).bool()[:, None] | (self.gt_area[:, None] < lo_t[None]) | (self.gt_area[:, None] > hi_t[None]))
        eps = float(np.spacing(1))
        for k in range(K):
            gsel = self.gt_cat == k
            npig_a = (~g_ign[gsel]).sum(0)                                    # [A] non-ignored gts of this category
            dsel_k = dcat == k
            if not bool(gsel.any()) and not bool(dsel_k.any()):
                continue
            for m, md in enumerate(self.max_dets):
                sel = (dsel_k & (rank < md)).nonzero().reshape(-1)
                sc = dscore[sel]
                order = torch.sort(sc, descending=True, stable=True)[1]
                sel = sel[order]
                nd = int(sel.numel())
                mt = dtm[sel].reshape(nd, A, T) >= 0                          # [nd, A, T]
                ig = dt_ig[sel].reshape(nd, A, T).bool()
                tp = torch.cumsum((mt & ~ig).double(), 0)
                fp = torch.cumsum((~mt & ~ig).double(), 0)
                for a in range(A):
                    npig = int(npig_a[a])
                    if npig == 0:
                        continue
                    if nd == 0:
                        recall[:, k, a, m] = 0
                        precision[:, :, k, a, m] = 0
                        continue
                    # divide by a TENSOR: torch turns `x / python_scalar` into `x * (1 / scalar)`, and 7 * (1/10) is one ulp
                    # above 7/10 - enough to move a recall value across the 0.70 threshold of the 101-point sampling
                    rc = (tp[:, a, :] / npig_a[a].double()).t().contiguous()  # [T, nd] non-decreasing
                    pr = (tp[:, a, :] / (fp[:, a, :] + tp[:, a, :] + eps)).t().contiguous()
                    recall[:, k, a, m] = rc[:, -1]
                    pr = torch.flip(torch.cummax(torch.flip(pr, [1]), 1)[0], [1])      # precision envelope
                    idx = torch.searchsorted(rc, rec_thrs[None, :].expand(T, R).contiguous(), right=False)
                    q = torch.where(idx < nd, torch.gather(pr, 1, idx.clamp(max=nd - 1)), torch.zeros((), dtype=torch.float64, device=dev))
                    precision[:, :, k, a, m] = q
        return self._summarize(precision, recall)

    def _match(self, dbox, det_off, gt_off, S, lo, hi, A, thr, T, max_det, scratch, dtm, dt_ig):
        hip.call('pt_coco_match', dbox, det_off, self.gt_box, self.gt_area, self.gt_flags, gt_off, S, lo, hi, A, thr, T, max_det,
                 scratch, dtm, dt_ig)

    def _summarize(self, precision, recall):
        M = len(self.max_dets)

        def mean_valid(s):
            s = s[s > -1]
            return float(s.mean()) if s.numel() else -1.0

        def ap(thr=None, area=0, md=M - 1):
            s = precision[:, :, :, area, md]
            if thr is not None:
                w = np.where(np.isclose(self.iou_thrs, thr))[0]
                if len(w) == 0:
                    return -1.0
                s = s[torch.as_tensor(w, device=s.device)]
            return mean_valid(s)

        def ar(area=0, md=M - 1):
            return mean_valid(recall[:, :, area, md])
        md = self.max_dets
        out = OrderedDict()
        for name, v in (('mAP', ap()), ('mAP_25', ap(.25)), ('mAP_50', ap(.5)), ('mAP_75', ap(.75)), ('mAP_vt', ap(area=1)),
                        ('mAP_t', ap(area=2)), ('mAP_s', ap(area=3)), ('mAP_m', ap(area=4)), (f'AR@{md[0]}', ar(md=0)),
                        (f'AR@{md[1]}', ar(md=1)), (f'AR@{md[2]}', ar(md=2)), (f'AR_vt@{md[2]}', ar(1)), (f'AR_t@{md[2]}', ar(2)),
                        (f'AR_s@{md[2]}', ar(3)), (f'AR_m@{md[2]}', ar(4))):
            out['bbox_' + name] = v
        out['precision'], out['recall'] = precision, recall
        return out


SODA_AREAS = (('Small', 0.0 ** 2, 32.0 ** 2), ('eS', 0.0 ** 2, 12.0 ** 2), ('rS', 12.0 ** 2, 20.0 ** 2), ('gS', 20.0 ** 2, 32.0 ** 2),
              ('Normal', 32.0 ** 2, 40.0 * 50.0))


class SODAAEvaluator(AITODEvaluator):
    """The SODA-A protocol of config 5 (`evaluation = dict(metric='mAP')` on SODAADataset):
    /root/reference/OBB_TOD/mmrotate/datasets/sodaa_eval/sodaa_eval.py - COCOeval with the IoU between oriented boxes
    (`box_iou_rotated`), area = w*h, the five SODA area bins (Small / eS / rS / gS / Normal), maxDets = [20000], no crowd
    and no ignore flags.  `annotations[i] = dict(bboxes [G,5], labels [G])` per whole image, `results[i][k] = ndarray
    [n,6]` (cx, cy, w, h, a, score) after the patch merge.  The IoU matrices of all (image, category) segments are
    filled by `pt_segment_iou_rotated` (chunks of <= 2^28 pairs), the greedy matching is `pt_coco_match_iou`: one
    wavefront per segment, one lane per (area bin, IoU threshold) - 5 x 10 = 50 lanes."""
    box_dim = 5
    PAIR_CHUNK = 1 << 28

    @staticmethod
    def _box_area(b):
        return b[:, 2] * b[:, 3]

    def __init__(self, annotations, num_classes=9, device='cuda', iou_thrs=None, max_dets=(20000,), areas=SODA_AREAS,
                 reference_ids=True):
        """reference_ids: keep SODAAeval's instance numbering from 0, which its matching code reads as "none"
        (sodaa_eval.py:108,121 vs :392,415,509): a detection matched to the dataset's first annotation counts as unmatched
        and the dataset's first detection does not block the ground truth it takes.  False = the intended COCO protocol."""
        super().__init__(annotations, num_classes, device=device, areas=areas, max_dets=max_dets, iou_thrs=iou_thrs)
        self.reference_ids = reference_ids
        self._gt_zero = self._det_zero = -1
        if reference_ids:                      # row of annotation 0 of the first image that has one, after the (image, category) sort
            for i, g in enumerate(annotations):
                lab = np.asarray(g['labels']).reshape(-1)
                if lab.size:
                    self._gt_zero = int((self.gt_key < i * self.K + int(lab[0])).sum())
                    break

    def _flatten(self, results):
        out = super()._flatten(results)
        self._det_zero = -1
        if self.reference_ids:                 # row of the first detection in (image, category) order, after the score sort
            seen = 0
            for per_cls in results:
                for arr in per_cls:
                    arr = np.asarray(arr, np.float32).reshape(-1, 6)
                    if arr.shape[0]:
                        self._det_zero = seen + int(np.where(np.argsort(-arr[:, 5], kind='mergesort') == 0)[0][0])
                        return out
                    seen += arr.shape[0]
        return out

    def _match(self, dbox, det_off, gt_off, S, lo, hi, A, thr, T, max_det, scratch, dtm, dt_ig):
        dev = dbox.device
        darea = (dbox[:, 2] * dbox[:, 3]).contiguous()
        nd = (det_off[1:] - det_off[:-1]).clamp(max=max_det).long()
        pairs = nd * (gt_off[1:] - gt_off[:-1]).long()
        cum = torch.cumsum(pairs, 0).cpu()
        pairs_c = pairs.cpu()
        s0 = 0
        while s0 < S:                                   # segment ranges whose IoU matrices fit the chunk budget
            base = int(cum[s0 - 1]) if s0 else 0
            s1 = s0 + 1
            while s1 < min(S, s0 + 65535) and int(cum[s1]) - base <= self.PAIR_CHUNK:
                s1 += 1
            n = int(cum[s1 - 1]) - base
            off = torch.zeros(s1 - s0 + 1, dtype=torch.int64, device=dev)
            off[1:] = torch.cumsum(pairs[s0:s1], 0)
            iou = torch.empty((max(n, 1),), dtype=torch.float32, device=dev)
            do, go = det_off[s0:s1 + 1].contiguous(), gt_off[s0:s1 + 1].contiguous()
            hip.call('pt_segment_iou_rotated', dbox, do, self.gt_box, go, s1 - s0, max_det, off, int(pairs_c[s0:s1].max()), iou)
            hip.call('pt_coco_match_iou', darea, do, iou, off, self.gt_area, self.gt_flags, go, s1 - s0, lo, hi, A, thr, T, max_det,
                     self._gt_zero, self._det_zero, scratch, dtm, dt_ig)
            s0 = s1

    NAMES = ('AP', 'AP_50', 'AP_75', 'AP_eS', 'AP_rS', 'AP_gS', 'AP_Normal', 'AR@20000', 'AR_eS@20000', 'AR_rS@20000', 'AR_gS@20000',
             'AR_Normal@20000')

    def _summarize(self, precision, recall):
        """sodaa_eval.py:560-650 `_summarizeDets`: 12 statistics at maxDets[0]; an IoU threshold is selected by exact equality
        with the linspace values, as the reference does."""
        def mean_valid(s):
            s = s[s > -1]
            return float(s.mean()) if s.numel() else -1.0

        def pick(s, thr):
            if thr is None:
                return s
            w = np.where(thr == self.iou_thrs)[0]
            return s[torch.as_tensor(w, device=s.device, dtype=torch.long)]
        stats = [mean_valid(pick(precision, None)[:, :, :, 0, 0]), mean_valid(pick(precision, .50)[:, :, :, 0, 0]),
                 mean_valid(pick(precision, .75)[:, :, :, 0, 0])]
        stats += [mean_valid(precision[:, :, :, a, 0]) for a in (1, 2, 3, 4)]
        stats += [mean_valid(recall[:, :, a, 0]) for a in (0, 1, 2, 3, 4)]
        out = OrderedDict((n, v) for n, v in zip(self.NAMES, stats))
        out['stats'] = np.array(stats)
        out['precision'], out['recall'] = precision, recall
        return out


def collect_results(result_part, size):
    """apis/test.py:105-171 (collect_results_cpu / _gpu): rank r holds the results of samples r, r + world, ... (the
    DistributedSampler of the test loader); rank 0 gets them interleaved back into dataset order and cut to `size` (the
    sampler pads by wrap-around), the other ranks get None.  One all_gather_object instead of a temp dir or padded byte
    tensors."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(result_part)[:size]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, list(result_part))
    if dist.get_rank() != 0:
        return None
    ordered = []
    for res in zip(*parts):
        ordered.extend(list(res))
    return ordered[:size]


def multi_gpu_test(model, loader, size):
    """apis/test.py:68-103: every rank runs its shard of the test loader (`build_dataloader(dist=True, shuffle=False)`), rank 0
    returns the whole dataset's results."""
    it = iter(loader)
    return collect_results(single_gpu_test(model, lambda _: next(it), len(loader)), size)


def single_gpu_test(model, batches, n_batches):
    """apis/test.py:16-66 for this path: `model(return_loss=False, rescale=True, **data)` over the data, results
    collected in dataset order (one list of per-class arrays per image)."""
    model.eval()
    results = []
    with torch.no_grad():
        for it in range(n_batches):
            data = batches(it)
            results.extend(model(return_loss=False, rescale=True, img=data['img'], img_metas=data['img_metas']))
    return results


# ------------------------------------------------------------------------------------------------
# Oriented tree (config 5, `evaluation = dict(metric='mAP')`): DOTA-style mAP with rotated IoU
# ------------------------------------------------------------------------------------------------
def eval_rbbox_map(det_results, annotations, iou_thr=0.5, use_07_metric=True, device='cuda'):
    """/root/reference/OBB_TOD/mmrotate/core/evaluation/eval_map.py:126-246 (eval_rbbox_map, tpfp_default, VOC07
    11-point average precision by default as in the reference, `area` with use_07_metric=False) on the GPU.  det_results[i][k] = ndarray [n,6] (cx,cy,w,h,a,score) as `rbbox2result` returns
    them; annotations[i] = dict(bboxes [G,5], labels [G], optional bboxes_ignore / labels_ignore).
    Returns (mAP, per-class list of dict(ap, recall, precision, num_gts, num_dets)).

    A detection is a true positive iff its best-IoU ground truth (IoU >= iou_thr, `pt_box_iou_rotated`) is not an
    ignored one and no higher-scored detection of the image claimed that ground truth first; detections whose best
    match is an ignored box are neither TP nor FP.  The "first claimant" rule needs no sequential loop: it is the
    first occurrence of each (image, gt) key in score order.  The reference's dataset wrapper overwrites `iou_thr`
    with 0.25 (sodaa_rewrite.py:206, a debugging leftover); pass iou_thr=0.25 to reproduce it."""
    from . import functional as F
    dev = torch.device(device)
    K = len(det_results[0])
    eps = float(np.finfo(np.float32).eps)
    out = []
    for k in range(K):
        tp_l, fp_l, sc_l, num_gts = [], [], [], 0
        gt_base = 0
        for dets, ann in zip(det_results, annotations):
            d = torch.as_tensor(np.asarray(dets[k], np.float32).reshape(-1, 6), device=dev)
            lab = np.asarray(ann['labels'])
            g = np.asarray(ann['bboxes'], np.float32).reshape(-1, 5)[lab == k]
            gi = np.zeros((0, 5), np.float32)
            if ann.get('labels_ignore', None) is not None:
                gi = np.asarray(ann['bboxes_ignore'], np.float32).reshape(-1, 5)[np.asarray(ann['labels_ignore']) == k]
            num_gts += g.shape[0]
            n, G = d.shape[0], g.shape[0] + gi.shape[0]
            tp = torch.zeros(n, device=dev)
            fp = torch.zeros(n, device=dev)
            if n and G == 0:
                fp[:] = 1
            elif n:
                allg = torch.as_tensor(np.vstack([g, gi]), device=dev)
                ious = F.box_iou_rotated(d[:, :5].contiguous(), allg)
                imax, iarg = ious.max(1)
                hit = imax >= iou_thr
                ign = iarg >= g.shape[0]
                order = torch.sort(d[:, 5], descending=True, stable=True)[1]
                # first detection (in score order) of every matched, non-ignored gt
                key = torch.where(hit & ~ign, iarg, torch.full_like(iarg, G))[order]
                first = torch.zeros(G + 1, dtype=torch.long, device=dev).scatter_reduce(
                    0, key, torch.arange(n, device=dev), reduce='amin', include_self=False)
                is_first = torch.zeros(n, dtype=torch.bool, device=dev)
                is_first[order] = (first[key] == torch.arange(n, device=dev)) & (key < G)
                tp = is_first.float()
                fp = ((hit & ~ign & ~is_first) | ~hit).float()
            tp_l.append(tp); fp_l.append(fp); sc_l.append(d[:, 5])
            gt_base += G
        sc = torch.cat(sc_l)
        order = torch.sort(sc, descending=True, stable=True)[1]
        # the per-class curves are short: finish on the host with the reference's dtypes (tp/fp float32, recalls
        # float64 because num_gts is an int64 array, precisions float32) - recall values such as 3/10 sit exactly on
        # the 11-point thresholds, so the rounding of this division decides `>=`
        tp = torch.cumsum(torch.cat(tp_l)[order], 0).cpu().numpy().astype(np.float32)
        fp = torch.cumsum(torch.cat(fp_l)[order], 0).cpu().numpy().astype(np.float32)
        feps = np.finfo(np.float32).eps
        rec = (tp / np.maximum(np.array([num_gts], dtype=int)[:, np.newaxis], feps))[0]
        prec = tp / np.maximum(tp + fp, feps)
        if use_07_metric:   # average_precision(mode='11points'): max precision at recall >= 0, .1, ..., 1
            ap = np.float32(0)
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                precs = prec[rec >= thr]
                ap += precs.max() if precs.size > 0 else 0
            ap = float(ap / 11)
        else:               # mode='area': envelope from the right, sum over recall steps
            mrec = np.hstack(([0.0], rec, [1.0]))
            mpre = np.hstack(([0.0], prec, [0.0]))
            for i in range(len(mpre) - 1, 0, -1):
                mpre[i - 1] = max(mpre[i - 1], mpre[i])
            ind = np.where(mrec[1:] != mrec[:-1])[0]
            ap = float(np.sum((mrec[ind + 1] - mrec[ind]) * mpre[ind + 1]))
        out.append(dict(ap=ap, recall=rec, precision=prec, num_gts=num_gts, num_dets=int(sc.numel())))
    aps = [r['ap'] for r in out if r['num_gts'] > 0]
    return (float(np.mean(aps)) if aps else 0.0), out
