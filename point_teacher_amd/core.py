"""Host-side mirror of the reference's `mmdet.core` names used on the Point-Teacher path:
box codecs, match costs, the two assigners, the delta coder, NMS wrappers, misc helpers.
Same names, argument meaning and return conventions as the reference (file:line cited per
item, relative to /root/reference/HBB_TOD/mmdet/); the arithmetic of the hot ones runs in
libpt_hip.so through `functional`."""
from functools import partial

import numpy as np
import torch
import torch.distributed as dist

from . import functional as F
from .registry import (BBOX_ASSIGNERS, BBOX_CODERS, IOU_CALCULATORS, MATCH_COST, build_iou_calculator,
                       build_match_cost)

bbox_overlaps = F.bbox_overlaps          # core/bbox/iou_calculators/iou2d_calculator.py:74


# ------------------------------------------------------------------ misc helpers --
def multi_apply(func, *args, **kwargs):
    """core/utils/misc.py:10-29"""
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def reduce_mean(tensor):
    """core/utils/dist_utils.py:63-69"""
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor


_ZEROS = {}


def zero_like_scalar(t):
    """A cached 0-dim zero of t's dtype on t's device: `torch.where(mask, x, zero_like_scalar(x))` is ONE launch (a python scalar
    there costs a `scalar_tensor` fill per call, `zeros_like(x)` a full-size one)."""
    k = (t.device, t.dtype)
    z = _ZEROS.get(k)
    if z is None:
        z = _ZEROS[k] = torch.zeros((), dtype=t.dtype, device=t.device)
    return z


def mean0(t):
    """t.mean() that is 0 (not NaN) for an empty tensor: a batch without any object logs zeros (the reference
    never sees one: its datasets filter empty images)."""
    return t.mean() if t.numel() else t.new_zeros(())          # (one launch; sum / n was two)


def reduce_mean_many(*scalars):
    """The normalisers of one loss call (num_pos of the classification branch, num_pos and the centerness
    sum of the regression branch) in ONE coalesced all-reduce instead of one tiny collective each
    (the reference issues them separately: fcos_head_p2b_ts.py:425, :436, :447)."""
    if not (dist.is_available() and dist.is_initialized()):
        return tuple(s.float() for s in scalars)
    return reduce_mean(torch.stack([s.float().reshape(()) for s in scalars])).unbind(0)


# -------------------------------------------------------------------- box codecs --
def _fusable(bbox):
    return bbox.is_cuda and bbox.dim() == 2 and bbox.shape[1] == 4 and bbox.dtype == torch.float32 and not bbox.requires_grad


def bbox_xyxy_to_cxcywh(bbox):
    """core/bbox/transforms.py:250-262 (one launch through pt_box_convert when no gradient is needed)"""
    if _fusable(bbox):
        return F.box_convert(bbox, 0)
    x1, y1, x2, y2 = bbox.split((1, 1, 1, 1), dim=-1)
    return torch.cat([(x1 + x2) / 2, (y1 + y2) / 2, (x2 - x1), (y2 - y1)], dim=-1)


def bbox_cxcywh_to_xyxy(bbox):
    """core/bbox/transforms.py:236-247"""
    if _fusable(bbox):
        return F.box_convert(bbox, 1)
    cx, cy, w, h = bbox.split((1, 1, 1, 1), dim=-1)
    return torch.cat([(cx - 0.5 * w), (cy - 0.5 * h), (cx + 0.5 * w), (cy + 0.5 * h)], dim=-1)


def distance2bbox(points, distance, max_shape=None):
    """core/bbox/transforms.py:134-177"""
    # (the same four differences / sums as the reference's per-coordinate form, two launches + one cat instead of four + a stack)
    bboxes = torch.cat((points[..., :2] - distance[..., :2], points[..., :2] + distance[..., 2:4]), -1)
    x1 = bboxes[..., 0]
    if max_shape is not None:
        if not isinstance(max_shape, torch.Tensor):
            max_shape = x1.new_tensor(max_shape)
        max_shape = max_shape[..., :2].type_as(x1)
        if max_shape.ndim == 2:
            assert bboxes.ndim == 3 and max_shape.size(0) == bboxes.size(0)
        min_xy = x1.new_tensor(0)
        max_xy = torch.cat([max_shape, max_shape], dim=-1).flip(-1).unsqueeze(-2)
        bboxes = torch.where(bboxes < min_xy, min_xy, bboxes)
        bboxes = torch.where(bboxes > max_xy, max_xy, bboxes)
    return bboxes


def bbox2roi(bbox_list):
    """core/bbox/transforms.py:58-78.  The batch-index column comes from the host-known list lengths (one asynchronous
    upload from the pinned ring) instead of a new_full + cat per image."""
    counts = [int(b.size(0)) for b in bbox_list]
    if sum(counts) == 0:
        return bbox_list[0].new_zeros((0, 5))
    boxes = torch.cat([b[:, :4] for b in bbox_list], 0)
    if not boxes.is_cuda:
        ids = torch.cat([b.new_full((b.size(0), 1), i) for i, b in enumerate(bbox_list)], 0)
        return torch.cat([ids, boxes], dim=-1)
    ids = F.upload_f32(np.repeat(np.arange(len(counts)), counts), boxes.device).to(boxes.dtype)
    return torch.cat([ids[:, None], boxes], dim=-1)


def bbox2result(bboxes, labels, num_classes):
    """core/bbox/transforms.py bbox2result: list of per-class ndarray[n,5]."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    bboxes = bboxes.detach().cpu().numpy()
    labels = labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


@IOU_CALCULATORS.register_module()
class BboxOverlaps2D:
    """iou2d_calculator.py:18-60"""

    def __init__(self, scale=1., dtype=None):
        self.scale = scale
        self.dtype = dtype

    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        return bbox_overlaps(bboxes1[..., :4], bboxes2[..., :4], mode, is_aligned)


# ------------------------------------------------------------------- match costs --
@MATCH_COST.register_module()
class FocalLossCost:
    """match_costs/match_cost.py:54-99 (dense [P,G] form; the assigners below never build it)."""

    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        cls_pred = cls_pred.sigmoid()
        neg_cost = -(1 - cls_pred + self.eps).log() * (1 - self.alpha) * cls_pred.pow(self.gamma)
        pos_cost = -(cls_pred + self.eps).log() * self.alpha * (1 - cls_pred).pow(self.gamma)
        return (pos_cost[:, gt_labels] - neg_cost[:, gt_labels]) * self.weight


@MATCH_COST.register_module()
class PointCost:
    """match_costs/match_cost.py:188-214"""

    def __init__(self, mode='L1', weight=1.):
        assert mode in ['L1', 'L2']
        self.weight, self.mode = weight, mode

    def __call__(self, bboxes, gt_bboxes):
        d = bboxes[:, None, :2] - gt_bboxes[None, :, :2]
        if self.mode == 'L1':
            return torch.sum(torch.abs(d), dim=2) * self.weight
        return torch.sqrt(torch.sum(d ** 2, dim=2)) * self.weight


@MATCH_COST.register_module()
class InsiderCost:
    """match_costs/match_cost.py:217-252"""

    def __init__(self, weight=1.):
        self.weight = weight

    def __call__(self, bboxes, gt_bboxes):
        b = bboxes[:, :4]
        x1, y1 = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2
        x2, y2 = b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2
        gx, gy = gt_bboxes[None, :, 0], gt_bboxes[None, :, 1]
        inside = (gx >= x1[:, None]) & (gx <= x2[:, None]) & (gy >= y1[:, None]) & (gy <= y2[:, None])
        return (~inside) * self.weight


# --------------------------------------------------------------------- assigners --
class AssignResult:
    """core/bbox/assigners/assign_result.py: gt_inds 0 = background, i+1 = gt i; labels -1 = none."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels

    @property
    def num_preds(self):
        return len(self.gt_inds)


def _labels_from_inds(gt_inds, gt_labels):
    lab = gt_labels.long()[(gt_inds - 1).clamp(min=0)]
    return torch.where(gt_inds > 0, lab, torch.full_like(lab, -1))


def _off1(n, device):
    return torch.tensor([0, n], dtype=torch.int32, device=device)


@BBOX_ASSIGNERS.register_module()
class TopkAssigner:
    """core/bbox/assigners/topk_assigner.py:13-147.  Per gt the `num_pre` L1-nearest
    predictions; if num_pre > topk the `topk` cheapest (class cost, over every column) of
    them; later gts overwrite earlier ones.  Distance ties -> lowest prediction index."""

    def __init__(self, num_pre=7, topk=3, cls_cost=dict(type='FocalLossCost', weight=2.0),
                 reg_cost=dict(type='PointCost', mode='L1', weight=5.0)):
        self.cls_cost = build_match_cost(cls_cost)
        self.reg_cost = build_match_cost(reg_cost)
        self.topk, self.num_pre = topk, num_pre
        self.iou_calculator = build_iou_calculator(dict(type='BboxOverlaps2D'))
        assert self.reg_cost.mode == 'L1', 'only the L1 PointCost of the shipped configs runs on the HIP path'

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, gt_bboxes_ignore=None, eps=1e-7):
        assert gt_bboxes_ignore is None, 'Only case when gt_bboxes_ignore is None is supported.'
        num_bboxes = bbox_pred.size(0)
        if gt_bboxes is None or gt_bboxes.size(0) == 0 or num_bboxes == 0:
            gt_inds = bbox_pred.new_full((num_bboxes,), 0 if (gt_bboxes is None or gt_bboxes.size(0) == 0) else -1,
                                         dtype=torch.long)
            return AssignResult(0 if gt_bboxes is None else gt_bboxes.size(0), gt_inds, None,
                                labels=bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long))
        G = gt_bboxes.size(0)
        off = _off1(G, bbox_pred.device)
        pts = bbox_pred[:, :2].contiguous()
        if self.num_pre <= self.topk:
            gi = F.topk_assign(pts, gt_bboxes, off, 1, self.num_pre)
        else:   # stage 2 on the class cost alone == the fused kernel with the location term off
            gi, _ = F.fuse_assign(pts, torch.zeros((num_bboxes, 4), device=pts.device), cls_pred, gt_bboxes,
                                  gt_labels, off, 1, self.num_pre, self.topk, self.cls_cost.weight,
                                  self.reg_cost.weight, 0.0)
        gi = gi.reshape(-1).long()
        return AssignResult(G, gi, None, labels=_labels_from_inds(gi, gt_labels))


@BBOX_ASSIGNERS.register_module()
class FUSETopkAssigner:
    """core/bbox/assigners/fuse_topk_assigner.py:13-121 (stage 1 on PointCost(points), stage 2
    on FocalLossCost + InsiderCost of the decoded boxes)."""

    def __init__(self, num_pre=7, topk=5, cls_cost=dict(type='FocalLossCost', weight=2.0),
                 reg_cost=dict(type='PointCost', mode='L1', weight=1.0),
                 location_cost=dict(type='InsiderCost', weight=2.0)):
        self.cls_cost = build_match_cost(cls_cost)
        self.reg_cost = build_match_cost(reg_cost)
        self.location_cost = build_match_cost(location_cost)
        self.topk, self.num_pre = topk, num_pre
        self.iou_calculator = build_iou_calculator(dict(type='BboxOverlaps2D'))

    def assign(self, bbox_pred, points, cls_pred, centerness, gt_bboxes, gt_labels, gt_bboxes_ignore=None, eps=1e-7):
        """bbox_pred: decoded boxes as (cx,cy,w,h) [P,4]; points [P,2]; gt_bboxes: gt points [G,>=2]."""
        assert gt_bboxes_ignore is None
        num_bboxes = bbox_pred.size(0)
        if gt_bboxes is None or gt_bboxes.size(0) == 0 or num_bboxes == 0:
            gt_inds = bbox_pred.new_full((num_bboxes,), 0 if (gt_bboxes is None or gt_bboxes.size(0) == 0) else -1,
                                         dtype=torch.long)
            return AssignResult(0 if gt_bboxes is None else gt_bboxes.size(0), gt_inds, None,
                                labels=bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long))
        G = gt_bboxes.size(0)
        # the kernel takes (l,t,r,b) distances; recover them from the cxcywh boxes the caller decoded
        xyxy = bbox_cxcywh_to_xyxy(bbox_pred)
        reg = torch.stack([points[:, 0] - xyxy[:, 0], points[:, 1] - xyxy[:, 1], xyxy[:, 2] - points[:, 0],
                           xyxy[:, 3] - points[:, 1]], -1)
        gi, _ = F.fuse_assign(points, reg, cls_pred, gt_bboxes, gt_labels, _off1(G, points.device), 1, self.num_pre,
                              self.topk, self.cls_cost.weight, self.reg_cost.weight, self.location_cost.weight)
        gi = gi.reshape(-1).long()
        return AssignResult(G, gi, None, labels=_labels_from_inds(gi, gt_labels))


# ------------------------------------------------------------------------ coders --
@BBOX_CODERS.register_module()
class DeltaXYWHBBoxCoder:
    """core/bbox/coder/delta_xywh_bbox_coder.py:9-100 (decode only is on the path; means 0/stds 1)."""

    def __init__(self, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.), clip_border=True,
                 add_ctr_clamp=False, ctr_clamp=32):
        self.means, self.stds = tuple(target_means), tuple(target_stds)
        self.clip_border, self.add_ctr_clamp, self.ctr_clamp = clip_border, add_ctr_clamp, ctr_clamp
        assert not add_ctr_clamp, 'add_ctr_clamp (YOLOF) is not on the Point-Teacher path'

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        d = pred_bboxes
        if self.means != (0., 0., 0., 0.) or self.stds != (1., 1., 1., 1.):
            d = d * d.new_tensor(self.stds) + d.new_tensor(self.means)
        return F.delta2bbox(bboxes, d, max_shape if self.clip_border else None, wh_ratio_clip)


# --------------------------------------------------------------- post-processing --
def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None,
                   return_inds=False):
    """core/post_processing/bbox_nms.py:5-91"""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 4)
    scores = multi_scores[:, :-1]
    labels = torch.arange(num_classes, dtype=torch.long, device=scores.device).view(1, -1).expand_as(scores)
    bboxes, scores, labels = bboxes.reshape(-1, 4), scores.reshape(-1), labels.reshape(-1)
    valid_mask = scores > score_thr
    if score_factors is not None:
        scores = scores * score_factors.view(-1, 1).expand(multi_scores.size(0), num_classes).reshape(-1)
    inds = valid_mask.nonzero(as_tuple=False).squeeze(1)
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    if bboxes.numel() == 0:
        dets = torch.cat([bboxes, scores[:, None]], -1)
        return (dets, labels, inds) if return_inds else (dets, labels)
    dets, keep = F.batched_nms(bboxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return (dets, labels[keep], inds[keep]) if return_inds else (dets, labels[keep])
