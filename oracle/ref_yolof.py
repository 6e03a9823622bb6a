"""CPU ORACLE of the YOLOF baseline's training-side logic (product: point_teacher_amd/yolof_baseline.py) - TEST INFRASTRUCTURE ONLY.

Restates, in plain torch fp32 on the CPU, of /root/reference/HBB_TOD/mmdet:
  models/dense_heads/yolof_head.py forward_single :118-134 (implicit objectness), loss :136-222, get_targets :223-309,
      _get_targets_single :310-414;
  core/bbox/assigners/uniform_assigner.py assign :38-134;
  core/bbox/coder/delta_xywh_bbox_coder.py delta2bbox :144-262 with add_ctr_clamp;
  models/losses/focal_loss.py (through ref_ops), iou_loss.py diou_loss :139-189 / DIoULoss :783-850.
PINNED by tests/golden/yolof_baseline.npz (outputs of the reference's own classes, oracle/gen_golden_yolof.py); only tests
import this file."""
import numpy as np
import torch

from . import ref_ops as R
from . import ref_retina as RR

INF = 1e8


def normalized_cls(cls, obj, num_classes):
    """[N, A*C, H, W], [N, A, H, W] -> [N, A*C, H, W]: cls + obj - log(1 + exp(cls) + exp(obj))."""
    N, _, H, W = cls.shape
    c, o = cls.view(N, -1, num_classes, H, W), obj.view(N, -1, 1, H, W)
    return (c + o - torch.log(1. + torch.clamp(c.exp(), max=INF) + torch.clamp(o.exp(), max=INF))).view(N, -1, H, W)


def delta2bbox_ctr_clamp(rois, deltas, ctr_clamp=32, wh_ratio_clip=16 / 1000):
    """means 0 / stds 1; the centre shift is clamped to +-ctr_clamp pixels and the size factor only from above."""
    px, py = (rois[:, 0] + rois[:, 2]) * 0.5, (rois[:, 1] + rois[:, 3]) * 0.5
    pw, ph = rois[:, 2] - rois[:, 0], rois[:, 3] - rois[:, 1]
    mr = abs(float(np.log(wh_ratio_clip)))
    dxw = (pw * deltas[:, 0]).clamp(min=-ctr_clamp, max=ctr_clamp)
    dyh = (ph * deltas[:, 1]).clamp(min=-ctr_clamp, max=ctr_clamp)
    gw, gh = pw * deltas[:, 2].clamp(max=mr).exp(), ph * deltas[:, 3].clamp(max=mr).exp()
    gx, gy = px + dxw, py + dyh
    return torch.stack([gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5], -1)


def _cxcywh(b):
    return torch.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], -1)


def uniform_assign(pred, anchors, gts, gt_labels, pos_ignore_thr=0.15, neg_ignore_thr=0.7, match_times=4):
    """-> dict(gt_inds [n] (-1 ignore, 0 negative, i + 1), labels [n] (-1 = none), pos_idx [2*k*G] bool, pos_pred [2*k*G, 4],
    target [2*k*G, 4]).  The duplicate indices of the scatter are resolved as the reference's CPU indexing does: the LAST write
    wins."""
    n, G = pred.shape[0], gts.shape[0]
    gt_inds = torch.zeros(n, dtype=torch.long)
    if G == 0 or n == 0:
        return dict(gt_inds=gt_inds, labels=torch.full((n,), -1, dtype=torch.long), pos_idx=torch.zeros(0, dtype=torch.bool),
                    pos_pred=pred.new_zeros((0, 4)), target=pred.new_zeros((0, 4)))
    cost = torch.cdist(_cxcywh(pred), _cxcywh(gts), p=1)
    cost_a = torch.cdist(_cxcywh(anchors), _cxcywh(gts), p=1)
    # stable top-k, lowest index first among ties: the anchor costs tie structurally (gen_golden_yolof.stable_topk)
    idx = torch.sort(cost, dim=0, stable=True)[1][:match_times]
    idx_a = torch.sort(cost_a, dim=0, stable=True)[1][:match_times]
    indexes = torch.cat((idx, idx_a), dim=1).reshape(-1)                    # row-major over [k, 2G]
    pred_ov = R.bbox_overlaps(pred, gts)
    anchor_ov = R.bbox_overlaps(anchors, gts)
    gt_inds[pred_ov.max(dim=1)[0] > neg_ignore_thr] = -1
    pos_gt = torch.arange(0, G).repeat(match_times * 2)
    pos_ignore = anchor_ov[indexes, pos_gt] < pos_ignore_thr
    val = pos_gt + 1
    val[pos_ignore] = -1
    for i, v in zip(indexes.tolist(), val.tolist()):                        # sequential: last write wins
        gt_inds[i] = v
    labels = torch.full((n,), -1, dtype=torch.long)
    pos = gt_inds > 0
    labels[pos] = gt_labels[gt_inds[pos] - 1]
    return dict(gt_inds=gt_inds, labels=labels, pos_idx=~pos_ignore, pos_pred=pred[indexes], target=gts[pos_gt])


def yolof_loss(cls_score, bbox_pred, gt_bboxes, gt_labels, num_classes, stride=8, scales=(0.5, 1, 2), ratios=(1.0,)):
    """cls_score [N, A*C, H, W] (already normalised), bbox_pred [N, A*4, H, W] -> (loss_cls, loss_bbox)."""
    N, _, H, W = cls_score.shape
    anchors = RR.grid_anchors([(H, W)], [stride], list(scales), list(ratios))[0]
    labels, weights, npos, pp, tt, ww = [], [], 0, [], [], []
    for b in range(N):
        bp = bbox_pred[b].permute(1, 2, 0).reshape(-1, 4)
        dec = delta2bbox_ctr_clamp(anchors, bp)
        r = uniform_assign(dec, anchors, gt_bboxes[b], gt_labels[b])
        n = anchors.shape[0]
        lab, lw = torch.full((n,), num_classes, dtype=torch.long), torch.zeros(n)
        pos, neg = (r['gt_inds'] > 0).nonzero().reshape(-1), (r['gt_inds'] == 0).nonzero().reshape(-1)
        lab[pos] = gt_labels[b][r['gt_inds'][pos] - 1]
        lw[pos] = 1.0
        lw[neg] = 1.0
        labels.append(lab); weights.append(lw)
        npos += max(pos.numel(), 1)
        pp.append(r['pos_pred']); tt.append(r['target']); ww.append(r['pos_idx'])
    avg = max(float(npos), 1.0)
    cs = cls_score.permute(0, 2, 3, 1).reshape(-1, num_classes)
    loss_cls = R.sigmoid_focal_loss(cs, torch.cat(labels), torch.cat(weights), avg_factor=avg)
    pp, tt, ww = torch.cat(pp), torch.cat(tt), torch.cat(ww).float()
    if pp.shape[0] == 0:
        return loss_cls, pp.sum() * 0
    return loss_cls, (R.diou_loss_elem(pp, tt) * ww).sum() / avg
