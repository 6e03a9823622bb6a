"""Oriented-box (OBB) building blocks of the path - the level-1 pieces of config 5
(`OBB_TOD/configs/point teacher/sodaa_fcos_pointteacher_1x.py`), mirroring
/root/reference/OBB_TOD/mmrotate/{core/bbox,models/losses,models/roi_heads/roi_extractors}.
The rotated head / detector classes that compose them are the next row to build (DESIGN.md
section 2); everything here runs on the HIP kernels of csrc/rotated.hip and csrc/nms.hip.
Citations are relative to /root/reference/OBB_TOD/mmrotate/.
"""

import numpy as np
import torch
import torch.nn as nn

from . import functional as F
from .losses import _reduce
from .registry import BBOX_CODERS, IOU_CALCULATORS, LOSSES, ROI_EXTRACTORS


# ------------------------------------------------------------------ transforms --
def norm_angle(angle, angle_range):
    """core/bbox/transforms.py:850-867"""
    if angle_range == 'oc':
        return angle
    if angle_range == 'le135':
        return (angle + np.pi / 4) % np.pi - np.pi / 4
    if angle_range == 'le90':
        return (angle + np.pi / 2) % np.pi - np.pi / 2
    raise NotImplementedError(angle_range)


def rbbox2roi(bbox_list):
    """core/bbox/transforms.py:73-92 -> [n,6] = (batch, cx, cy, w, h, a)"""
    out = []
    for i, b in enumerate(bbox_list):
        out.append(torch.cat([b.new_full((b.size(0), 1), i), b[:, :5]], dim=-1) if b.size(0) > 0 else b.new_zeros((0, 6)))
    return torch.cat(out, 0)


def obb2poly_le90(rboxes):
    """core/bbox/transforms.py obb2poly_le90 (same arithmetic as the HBB fork's copy)."""
    from .proposals import obb2poly_le90 as _f
    return _f(rboxes)


def rbbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False):
    """core/bbox/iou_calculators/rotate_iou2d_calculator.py rbbox_overlaps -> mmcv.ops.box_iou_rotated."""
    assert mode == 'iou', "only mode='iou' is used on the Point-Teacher path"
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    if rows * cols == 0:
        return bboxes1.new_zeros((rows,)) if is_aligned else bboxes1.new_zeros((rows, cols))
    return F.box_iou_rotated(bboxes1[:, :5], bboxes2[:, :5], aligned=is_aligned)


@IOU_CALCULATORS.register_module()
class RBboxOverlaps2D:
    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        return rbbox_overlaps(bboxes1[..., :5].contiguous(), bboxes2[..., :5].contiguous(), mode, is_aligned)


@BBOX_CODERS.register_module()
class DistanceAnglePointCoder:
    """core/bbox/coder/distance_angle_point_coder.py:9-111"""

    def __init__(self, clip_border=True, angle_version='oc'):
        self.clip_border, self.angle_version = clip_border, angle_version

    def encode(self, points, gt_bboxes, max_dis=None, eps=0.1):
        return self.obb2distance(points, gt_bboxes, max_dis, eps)

    def decode(self, points, pred_bboxes, max_shape=None):
        assert points.size(0) == pred_bboxes.size(0) and points.size(-1) == 2 and pred_bboxes.size(-1) == 5
        return self.distance2obb(points, pred_bboxes, max_shape if self.clip_border else None, self.angle_version)

    def obb2distance(self, points, distance, max_dis=None, eps=None):
        """:73-91: (l, t, r, b, angle) of a point w.r.t. an oriented box."""
        ctr, wh, angle = torch.split(distance, [2, 2, 1], dim=1)
        c, s = torch.cos(angle), torch.sin(angle)
        off = points - ctr
        ox = c[:, 0] * off[:, 0] + s[:, 0] * off[:, 1]
        oy = -s[:, 0] * off[:, 0] + c[:, 0] * off[:, 1]
        w, h = wh[:, 0], wh[:, 1]
        left, right, top, bottom = w / 2 + ox, w / 2 - ox, h / 2 + oy, h / 2 - oy
        if max_dis is not None:
            left, top = left.clamp(min=0, max=max_dis - eps), top.clamp(min=0, max=max_dis - eps)
            right, bottom = right.clamp(min=0, max=max_dis - eps), bottom.clamp(min=0, max=max_dis - eps)
        return torch.stack((left, top, right, bottom, angle[:, 0]), -1)

    def distance2obb(self, points, distance, max_shape=None, angle_version='oc'):
        """:93-111"""
        d, angle = distance.split([4, 1], dim=1)
        c, s = torch.cos(angle), torch.sin(angle)
        wh = d[:, :2] + d[:, 2:]
        ot = (d[:, 2:] - d[:, :2]) / 2
        off = torch.cat([c * ot[:, 0:1] - s * ot[:, 1:2], s * ot[:, 0:1] + c * ot[:, 1:2]], dim=1)
        return torch.cat([points + off, wh, norm_angle(angle, angle_version)], dim=-1)


def rotated_fcos_targets(points, gt_inds, boxes5):
    """rotated_fcos_head_p2rb_ts.py:671-716: (l,t,r,b) in the frame of the ASSIGNED oriented box and its
    angle, for every point (box 0 for unassigned points, as `inds * 0` does)."""
    idx = (gt_inds.long() - 1).clamp(min=0)
    b = boxes5[idx]
    c, s = torch.cos(b[:, 4]), torch.sin(b[:, 4])
    off = points - b[:, :2]
    ox = c * off[:, 0] + s * off[:, 1]
    oy = -s * off[:, 0] + c * off[:, 1]
    w, h = b[:, 2], b[:, 3]
    return torch.stack([w / 2 + ox, h / 2 + oy, w / 2 - ox, h / 2 - oy], -1), b[:, 4:5]


# ---------------------------------------------------------------------- losses --
def _iou_to_loss(ious, mode, eps):
    ious = ious.clamp(min=eps)
    if mode == 'linear':
        return 1 - ious
    if mode == 'square':
        return 1 - ious ** 2
    return -ious.log()


def rotated_iou_loss_elem(pred, target, mode='log', eps=1e-6):
    """models/losses/rotated_iou_loss.py:17-60 (element-wise)."""
    return _iou_to_loss(F.diff_iou_rotated_2d(pred.unsqueeze(0), target.unsqueeze(0)).squeeze(0), mode, eps)


@LOSSES.register_module()
class RotatedIoULoss(nn.Module):
    """models/losses/rotated_iou_loss.py:149-226"""

    def __init__(self, linear=False, eps=1e-6, reduction='mean', loss_weight=1.0, mode='log'):
        super().__init__()
        assert mode in ['linear', 'square', 'log']
        self.mode = 'linear' if linear else mode
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def _elem(self, pred, target):
        return rotated_iou_loss_elem(pred, target, self.mode, self.eps)

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None and weight.dim() > 1:
            assert weight.shape == pred.shape
            weight = weight.mean(-1)
        return self.loss_weight * _reduce(self._elem(pred, target), weight, reduction, avg_factor)


@LOSSES.register_module()
class DN_IoULoss(RotatedIoULoss):
    """models/losses/rotated_iou_loss.py:229-300 over DN_iou_loss :105-147: min over the 9 targets
    whose (w, h) are shrunk/grown by hyper/2, averaged with the plain loss.  Unlike the HBB
    DN_diou_loss the base term is per-element (undecorated helper, :131)."""

    def __init__(self, linear=False, eps=1e-6, reduction='mean', loss_weight=1.0, mode='log', hyper=0.2):
        super().__init__(linear, eps, reduction, loss_weight, mode)
        self.hyper = hyper

    def _elem(self, pred, target):
        base = rotated_iou_loss_elem(pred, target, self.mode, self.eps)
        anx = self.hyper / 2
        w, h = target[:, 2], target[:, 3]
        bank = []
        for i in (-1, 0, 1):
            for j in (-1, 0, 1):
                t = torch.stack([target[:, 0], target[:, 1], w - anx * w * i, h - anx * h * j, target[:, 4]], 1)
                bank.append(rotated_iou_loss_elem(pred, t, self.mode, self.eps))
        return (base + torch.stack(bank, 1).min(1)[0]) / 2


# -------------------------------------------------------------- RoI extraction --
class RoIAlignRotated(nn.Module):
    """mmcv.ops.RoIAlignRotated(out_size, spatial_scale, sample_num=0, aligned=True, clockwise=False)."""

    def __init__(self, out_size, spatial_scale, sample_num=0, aligned=True, clockwise=False):
        super().__init__()
        self.out_size = out_size if isinstance(out_size, int) else out_size[0]
        self.spatial_scale, self.sample_num = float(spatial_scale), int(sample_num)
        self.aligned, self.clockwise = aligned, clockwise

    def forward(self, features, rois):
        return F.roi_align_rotated(features, rois, self.out_size, self.spatial_scale, self.sample_num, self.aligned,
                                   self.clockwise)


@ROI_EXTRACTORS.register_module()
class RotatedSingleRoIExtractor(nn.Module):
    """models/roi_heads/roi_extractors/rotate_single_level_roi_extractor.py:13-126 (single level)."""

    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56, init_cfg=None):
        super().__init__()
        cfg = dict(roi_layer)
        assert cfg.pop('type') == 'RoIAlignRotated'
        self.roi_layers = nn.ModuleList([RoIAlignRotated(spatial_scale=1 / s, **cfg) for s in featmap_strides])
        self.out_channels, self.featmap_strides, self.finest_scale = out_channels, featmap_strides, finest_scale

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def forward(self, feats, rois, roi_scale_factor=None):
        assert len(feats) == 1 and roi_scale_factor is None, 'single-level extraction is what the path uses'
        if rois.shape[0] == 0:
            o = self.roi_layers[0].out_size
            return feats[0].new_zeros(0, self.out_channels, o, o)
        return self.roi_layers[0](feats[0], rois)


# -------------------------------------------------------------- post-processing --
def multiclass_nms_rotated(multi_bboxes, multi_scores, score_thr, nms, max_num=-1, score_factors=None, return_inds=False):
    """core/post_processing/bbox_nms_rotated.py:6-91"""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 5:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 5)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 5)
    scores = multi_scores[:, :-1]
    labels = torch.arange(num_classes, dtype=torch.long, device=scores.device).view(1, -1).expand_as(scores)
    bboxes, scores, labels = bboxes.reshape(-1, 5), scores.reshape(-1), labels.reshape(-1)
    valid = scores > score_thr
    if score_factors is not None:
        scores = scores * score_factors.view(-1, 1).expand(multi_scores.size(0), num_classes).reshape(-1)
    inds = valid.nonzero(as_tuple=False).squeeze(1)
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    if bboxes.numel() == 0:
        dets = torch.cat([bboxes, scores[:, None]], -1)
        return (dets, labels, inds) if return_inds else (dets, labels)
    max_coordinate = bboxes[:, :2].max() + bboxes[:, 2:4].max()
    nb = bboxes.clone()
    nb[:, :2] = nb[:, :2] + (labels.to(bboxes) * (max_coordinate + 1))[:, None]
    _, keep = F.nms_rotated(nb, scores, nms['iou_thr'] if 'iou_thr' in nms else nms.get('iou_threshold'))
    if max_num > 0:
        keep = keep[:max_num]
    dets = torch.cat([bboxes[keep], scores[keep, None]], 1)
    return (dets, labels[keep], keep) if return_inds else (dets, labels[keep])
