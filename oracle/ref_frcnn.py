"""CPU ORACLE of the Faster R-CNN baseline's training-side logic - TEST INFRASTRUCTURE ONLY.

Restates, in plain torch fp32 on the CPU, of /root/reference/HBB_TOD/mmdet:
  core/bbox/samplers/base_sampler.py sample :34-101 + random_sampler.py random_choice / _sample_pos / _sample_neg :31-81
      (the permutations torch.randperm returned are an INPUT: `perms`, consumed in call order);
  core/bbox/samplers/sampling_result.py :24-60, core/bbox/assigners/assign_result.py add_gt_ :196-212;
  models/dense_heads/anchor_head.py _get_targets_single :181-278 / loss_single / loss :382-500 as RPNHead uses them
      (models/dense_heads/rpn_head.py loss :66-98: one class, sigmoid cross entropy, sampling);
  models/dense_heads/rpn_head.py _get_bboxes_single :159-249 (per-level top-k by sort, DeltaXYWHBBoxCoder.decode with clipping,
      level-aware NMS, max_per_img);
  models/roi_heads/bbox_heads/bbox_head.py _get_target_single / get_targets :122-255 and loss :257-314
      (models/losses/cross_entropy_loss.py cross_entropy, smooth_l1_loss.py L1Loss, accuracy.py top-1).
PINNED by tests/golden/frcnn_baseline.npz (outputs of the reference's own classes, oracle/gen_golden_frcnn.py); only tests
import this file."""
import numpy as np
import torch
import torch.nn.functional as F

from . import ref_ops as R
from . import ref_retina as RR


class PermFeed:
    """Hands out the recorded torch.randperm results in call order (and checks their length)."""

    def __init__(self, perms):
        self.perms, self.i = list(perms), 0

    def __call__(self, n):
        p = torch.as_tensor(self.perms[self.i]).long()
        self.i += 1
        assert p.numel() == n, (p.numel(), n)
        return p


def random_sample(gt_inds, num, pos_fraction, randperm, num_gts_added=0):
    """BaseSampler.sample for a RandomSampler: (pos_inds, neg_inds), each sorted ascending (`.unique()`).
    `gt_inds` already has the `add_gt_as_proposals` rows in front when num_gts_added > 0."""
    def choice(gallery, k):
        return gallery[randperm(gallery.numel())[:k]]
    pos = (gt_inds > 0).nonzero().reshape(-1)
    n_pos = int(num * pos_fraction)
    if pos.numel() > n_pos:
        pos = choice(pos, n_pos)
    pos = pos.unique()
    neg = (gt_inds == 0).nonzero().reshape(-1)
    n_neg = num - pos.numel()
    if neg.numel() > n_neg:
        neg = choice(neg, n_neg)
    return pos, neg.unique()


# --------------------------------------------------------------------------------------------------------- RPN --
def rpn_targets(anchors, gt_bboxes, assigner_kw, num, pos_fraction, randperm):
    """anchor_head._get_targets_single for an RPN (no labels: foreground = 0, background = 1; every anchor valid)."""
    asg, _ = RR.max_iou_assign(anchors, gt_bboxes, **assigner_kw)
    pos, neg = random_sample(asg, num, pos_fraction, randperm)
    n = anchors.shape[0]
    lab, lw = torch.full((n,), 1, dtype=torch.long), torch.zeros(n)
    bt, bw = torch.zeros_like(anchors), torch.zeros_like(anchors)
    if pos.numel():
        bt[pos] = RR.bbox2delta(anchors[pos], gt_bboxes[asg[pos] - 1])
        bw[pos] = 1.0
        lab[pos] = 0
        lw[pos] = 1.0
    lw[neg] = 1.0
    return lab, lw, bt, bw, pos.numel(), neg.numel()


def rpn_loss(cls_scores, bbox_preds, gt_bboxes, strides, scales, ratios, assigner_kw, num, pos_fraction, randperm):
    """-> (loss_rpn_cls per level, loss_rpn_bbox per level, per-level targets, num_total_pos, num_total_neg)."""
    sizes = [tuple(c.shape[-2:]) for c in cls_scores]
    lvl = RR.grid_anchors(sizes, strides, scales, ratios)
    flat = torch.cat(lvl)
    per_img = [rpn_targets(flat, g, assigner_kw, num, pos_fraction, randperm) for g in gt_bboxes]
    npos, nneg = sum(t[4] for t in per_img), sum(t[5] for t in per_img)
    avg = float(npos + nneg)                                      # sampling: num_total_samples = pos + neg (anchor_head.py:470)
    counts = [a.shape[0] for a in lvl]
    lc, lb, tg = [], [], []
    o = 0
    for i, n in enumerate(counts):
        lab = torch.stack([t[0][o:o + n] for t in per_img]).reshape(-1)
        lw = torch.stack([t[1][o:o + n] for t in per_img]).reshape(-1)
        bt = torch.stack([t[2][o:o + n] for t in per_img]).reshape(-1, 4)
        bw = torch.stack([t[3][o:o + n] for t in per_img]).reshape(-1, 4)
        cs = cls_scores[i].permute(0, 2, 3, 1).reshape(-1)
        bp = bbox_preds[i].permute(0, 2, 3, 1).reshape(-1, 4)
        onehot = (lab == 0).float()                              # _expand_onehot_labels with one channel
        lc.append((F.binary_cross_entropy_with_logits(cs, onehot, reduction='none') * lw).sum() / avg)
        lb.append(((bp - bt).abs() * bw).sum() / avg)
        tg.append((lab.view(len(per_img), -1), lw.view(len(per_img), -1), bt.view(len(per_img), -1, 4), bw.view(len(per_img), -1, 4)))
        o += n
    return lc, lb, tg, npos, nneg


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None, wh_ratio_clip=16 / 1000):
    """delta_xywh_bbox_coder.py delta2bbox :144-260 (clip_border, no add_ctr_clamp)."""
    d = deltas * deltas.new_tensor(stds) + deltas.new_tensor(means)
    mr = abs(float(np.log(wh_ratio_clip)))
    dw, dh = d[:, 2].clamp(-mr, mr), d[:, 3].clamp(-mr, mr)
    px, py = (rois[:, 0] + rois[:, 2]) * 0.5, (rois[:, 1] + rois[:, 3]) * 0.5
    pw, ph = rois[:, 2] - rois[:, 0], rois[:, 3] - rois[:, 1]
    gw, gh = pw * dw.exp(), ph * dh.exp()
    gx, gy = px + pw * d[:, 0], py + ph * d[:, 1]
    x1, y1, x2, y2 = gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5
    if max_shape is not None:
        x1, x2 = x1.clamp(0, max_shape[1]), x2.clamp(0, max_shape[1])
        y1, y2 = y1.clamp(0, max_shape[0]), y2.clamp(0, max_shape[0])
    return torch.stack([x1, y1, x2, y2], -1)


def rpn_proposals(cls_scores, bbox_preds, img_hw, strides, scales, ratios, nms_pre, max_per_img, iou_thr, min_bbox_size=0):
    """rpn_head._get_bboxes_single for ONE image (tensors [A*1,H,W] / [A*4,H,W] per level) -> [n, 5]."""
    sizes = [tuple(c.shape[-2:]) for c in cls_scores]
    lvl = RR.grid_anchors(sizes, strides, scales, ratios)
    S, D, AN, ID = [], [], [], []
    for i, (c, r) in enumerate(zip(cls_scores, bbox_preds)):
        s = c.permute(1, 2, 0).reshape(-1).sigmoid()
        d = r.permute(1, 2, 0).reshape(-1, 4)
        a = lvl[i]
        if 0 < nms_pre < s.shape[0]:
            rs, ri = s.sort(descending=True)
            s, d, a = rs[:nms_pre], d[ri[:nms_pre]], a[ri[:nms_pre]]
        S.append(s); D.append(d); AN.append(a); ID.append(torch.full((s.shape[0],), i, dtype=torch.long))
    s, d, a, ids = torch.cat(S), torch.cat(D), torch.cat(AN), torch.cat(ID)
    p = delta2bbox(a, d, max_shape=img_hw)
    if min_bbox_size > 0:
        ok = ((p[:, 2] - p[:, 0]) >= min_bbox_size) & ((p[:, 3] - p[:, 1]) >= min_bbox_size)
        p, s, ids = p[ok], s[ok], ids[ok]
    if p.numel() == 0:
        return p.new_zeros(0, 5)
    keep = R.batched_nms(p, s, ids, iou_thr)
    return torch.cat([p[keep], s[keep, None]], -1)[:max_per_img]


# ------------------------------------------------------------------------------------------------------- R-CNN --
def rcnn_sample(proposals, gt_bboxes, gt_labels, assigner_kw, num, pos_fraction, randperm):
    """MaxIoUAssigner.assign + RandomSampler.sample(add_gt_as_proposals=True) for one image ->
    dict(pos_inds, neg_inds, bboxes [pos | neg], pos_gt_bboxes, pos_gt_labels, pos_assigned)."""
    asg, _ = RR.max_iou_assign(proposals, gt_bboxes, **assigner_kw)
    G = gt_bboxes.shape[0]
    boxes = proposals
    if G > 0:
        boxes = torch.cat([gt_bboxes, proposals])
        asg = torch.cat([torch.arange(1, G + 1), asg])
    pos, neg = random_sample(asg, num, pos_fraction, randperm)
    pa = asg[pos] - 1
    return dict(pos_inds=pos, neg_inds=neg, bboxes=torch.cat([boxes[pos], boxes[neg]]), pos_bboxes=boxes[pos], neg_bboxes=boxes[neg],
                pos_assigned=pa, pos_gt_bboxes=gt_bboxes[pa] if G else gt_bboxes.view(-1, 4), pos_gt_labels=gt_labels[pa] if G else gt_labels)


def rcnn_targets(samples, num_classes, stds=(0.1, 0.1, 0.2, 0.2)):
    """bbox_head.get_targets(concat=True) -> (labels, label_weights, bbox_targets, bbox_weights)."""
    L, LW, BT, BW = [], [], [], []
    for s in samples:
        npos, nneg = s['pos_bboxes'].shape[0], s['neg_bboxes'].shape[0]
        lab = torch.full((npos + nneg,), num_classes, dtype=torch.long)
        lw, bt, bw = torch.zeros(npos + nneg), torch.zeros(npos + nneg, 4), torch.zeros(npos + nneg, 4)
        if npos:
            lab[:npos] = s['pos_gt_labels']
            lw[:npos] = 1.0
            bt[:npos] = RR.bbox2delta(s['pos_bboxes'], s['pos_gt_bboxes']) / torch.tensor(stds)
            bw[:npos] = 1.0
        if nneg:
            lw[-nneg:] = 1.0
        L.append(lab); LW.append(lw); BT.append(bt); BW.append(bw)
    return torch.cat(L), torch.cat(LW), torch.cat(BT), torch.cat(BW)


def rcnn_loss(cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, num_classes):
    """bbox_head.loss: softmax cross entropy / #(weight > 0), class-specific L1 on the positives / #samples, top-1 accuracy (%)."""
    avg = max(float((label_weights > 0).sum()), 1.0)
    loss_cls = (F.cross_entropy(cls_score, labels, reduction='none') * label_weights).sum() / avg
    acc = (cls_score.argmax(1) == labels).float().mean() * 100.0 if labels.numel() else cls_score.new_zeros(())
    pos = (labels >= 0) & (labels < num_classes)
    if pos.any():
        pp = bbox_pred.view(bbox_pred.shape[0], -1, 4)[pos, labels[pos]]
        loss_bbox = ((pp - bbox_targets[pos]).abs() * bbox_weights[pos]).sum() / float(bbox_targets.shape[0])
    else:
        loss_bbox = bbox_pred[pos].sum()
    return loss_cls, loss_bbox, acc


def bbox2roi(bbox_list):
    """core/bbox/transforms.py:58-78"""
    out = []
    for i, b in enumerate(bbox_list):
        out.append(torch.cat([b.new_full((b.shape[0], 1), i), b[:, :4]], -1) if b.shape[0] else b.new_zeros((0, 5)))
    return torch.cat(out, 0)
