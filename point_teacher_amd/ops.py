"""The mmcv 1.x op surface of the Point-Teacher path, name for name and argument for argument, over libpt_hip.so:
a maintainer of the reference swaps `from mmcv.ops import X` for `from point_teacher_amd.ops import X`
("op surface").  Signatures follow mmcv 1.3 - 1.7 (the range HBB_TOD/mmdet/__init__.py:18-19 and
OBB_TOD/mmrotate/__init__.py:23-24 accept; mmcv itself is un-vendored and absent here, so they are restated from its
public API).  Call sites in the reference:

  sigmoid_focal_loss      HBB_TOD/mmdet/models/losses/focal_loss.py:4, :85
  RoIAlign / roi_align    HBB_TOD/mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:53-58
  nms / batched_nms       HBB_TOD/mmdet/core/post_processing/bbox_nms.py:2, :76
  nms_rotated             HBB_TOD/mmdet/models/detectors/syn_images_generator_v2.py:8, :667; OBB_TOD/.../bbox_nms_rotated.py:80
  box_iou_rotated         OBB_TOD/mmrotate/core/bbox/iou_calculators/rotate_iou2d_calculator.py
  diff_iou_rotated_2d     OBB_TOD/mmrotate/models/losses/rotated_iou_loss.py:47, :90
  RoIAlignRotated         OBB_TOD/mmrotate/models/roi_heads/roi_extractors/rotate_single_level_roi_extractor.py:126
  (Modulated)DeformConv2d(Pack)   HBB_TOD/mmdet/models/dense_heads/anchor_free_head.py:101-102 (`conv_cfg=dict(type='DCNv2')`)

Everything runs on the HIP device; there is no CPU path (non-CUDA tensors are refused)."""
import torch
import torch.nn as nn

from . import functional as F
from .nn_modules import DeformConv2dPack, ModulatedDeformConv2dPack, RoIAlign as _RoIAlignModule

__all__ = ['sigmoid_focal_loss', 'SigmoidFocalLoss', 'roi_align', 'RoIAlign', 'nms', 'batched_nms', 'nms_rotated',
           'box_iou_rotated', 'diff_iou_rotated_2d', 'roi_align_rotated', 'RoIAlignRotated', 'deform_conv2d', 'DeformConv2d',
           'DeformConv2dPack', 'modulated_deform_conv2d', 'ModulatedDeformConv2d', 'ModulatedDeformConv2dPack']

NMS_MAX_CANDIDATES = 32768          # csrc/nms.hip NMS_MAXN: one launch holds that many candidates' suppression bitmasks


# ----------------------------------------------------------------------------------------------- focal loss --
def sigmoid_focal_loss(input, target, gamma=2.0, alpha=0.25, weight=None, reduction='mean'):
    """mmcv.ops.sigmoid_focal_loss(input [N,C] logits, target [N] int64 in [0, C] (C = background), gamma, alpha,
    weight [C] or None, reduction).  As in mmcv, `weight` is a PER-CLASS weight; 'none' returns [N,C]."""
    assert reduction in ('none', 'mean', 'sum')
    assert input.dim() == 2 and target.dim() == 1 and input.size(0) == target.size(0)
    N, C = input.shape
    if reduction == 'none' or weight is not None:
        elem = _FocalElem.apply(input, target, float(gamma), float(alpha))
        if weight is not None:
            elem = elem * weight.view(1, -1)
        if reduction == 'none':
            return elem
        return elem.sum() / N if reduction == 'mean' else elem.sum()
    total = F.sigmoid_focal_loss_sum(input, target, None, gamma, alpha)
    return total / N if reduction == 'mean' else total


class _FocalElem(torch.autograd.Function):
    """Element-wise focal loss with its gradient (the summed form the training loop uses is functional._FocalSum)."""

    @staticmethod
    def forward(ctx, logits, labels, gamma, alpha):
        ctx.save_for_backward(logits, labels)
        ctx.ga = (gamma, alpha)
        return F.sigmoid_focal_loss_elem(logits, labels, gamma, alpha)

    @staticmethod
    def backward(ctx, g):
        logits, labels = ctx.saved_tensors
        N, C = logits.shape
        one = torch.ones(1, dtype=torch.float32, device=logits.device)
        grad = torch.empty_like(logits, dtype=torch.float32)
        F.hip.call('pt_sigmoid_focal_loss_bwd', F._f(logits), labels.to(torch.int32).contiguous(), None, one, N, C,
                   ctx.ga[0], ctx.ga[1], grad)
        return grad * g, None, None, None


class SigmoidFocalLoss(nn.Module):
    def __init__(self, gamma, alpha, weight=None, reduction='mean'):
        super().__init__()
        self.gamma, self.alpha, self.reduction = gamma, alpha, reduction
        self.register_buffer('weight', weight)

    def forward(self, input, target):
        return sigmoid_focal_loss(input, target, self.gamma, self.alpha, self.weight, self.reduction)


# ------------------------------------------------------------------------------------------------- RoIAlign --
def roi_align(input, rois, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True):
    """mmcv.ops.roi_align(input [B,C,H,W], rois [K,5] = (batch, x1, y1, x2, y2), ...) -> [K,C,oh,ow]."""
    assert pool_mode == 'avg', "pool_mode='max' is not on the Point-Teacher path"
    o = output_size if isinstance(output_size, int) else output_size[0]
    assert isinstance(output_size, int) or output_size[0] == output_size[1]
    return F.roi_align(input, rois, o, spatial_scale, sampling_ratio, aligned)


class RoIAlign(_RoIAlignModule):
    """mmcv.ops.RoIAlign(output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True,
    use_torchvision=False); forward(input, rois)."""


# ------------------------------------------------------------------------------------------------------ NMS --
def nms(boxes, scores, iou_threshold, offset=0, score_threshold=0, max_num=-1):
    """mmcv.ops.nms -> (dets [M,5] = boxes + score, keep [M] int64), descending score.  offset in {0, 1} is mmcv's
    legacy "+1 pixel" box size.  Above NMS_MAX_CANDIDATES the greedy pass runs on the best NMS_MAX_CANDIDATES by
    score - exact whenever the suppressed tail cannot be kept, which is the case for every test_cfg of the path
    (`nms_pre` x classes <= 27 000) and otherwise a documented limit."""
    assert offset in (0, 1)
    assert boxes.size(1) == 4 and boxes.size(0) == scores.size(0)
    if score_threshold > 0:
        sel = (scores > score_threshold).nonzero(as_tuple=False).squeeze(1)
        dets, keep = nms(boxes[sel], scores[sel], iou_threshold, offset, 0, max_num)
        return dets, sel[keep]
    b = boxes
    if offset == 1:
        b = boxes.clone()
        b[:, 2:] += 1
    sel = None
    if b.size(0) > NMS_MAX_CANDIDATES:
        sel = torch.topk(scores, NMS_MAX_CANDIDATES, sorted=False)[1]
        b, scores_ = b[sel], scores[sel]
    else:
        scores_ = scores
    _, keep = F.nms(b, scores_, iou_threshold)
    if sel is not None:
        keep = sel[keep]
    if max_num > 0:
        keep = keep[:max_num]
    return torch.cat([boxes[keep], scores[keep, None]], 1), keep


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv.ops.batched_nms: class-aware NMS -> (dets [M,5], keep [M]); nms_cfg = dict(type='nms', iou_threshold=...,
    [max_num, split_thr]).  Candidates above NMS_MAX_CANDIDATES are processed class by class, as mmcv itself does
    above `split_thr`."""
    cfg = dict(nms_cfg)
    class_agnostic = cfg.pop('class_agnostic', class_agnostic)
    assert cfg.pop('type', 'nms') == 'nms'
    thr = cfg.pop('iou_threshold', cfg.pop('iou_thr', 0.5))
    max_num = cfg.pop('max_num', -1)
    if boxes.size(0) == 0:
        return torch.cat([boxes, scores[:, None]], 1), torch.empty((0,), dtype=torch.long, device=boxes.device)
    if class_agnostic:
        dets, keep = nms(boxes, scores, thr)
    elif boxes.size(0) <= NMS_MAX_CANDIDATES:
        dets, keep = F.nms(boxes, scores, thr, class_ids=idxs)
    else:
        keeps = []
        for c in torch.unique(idxs).tolist():
            sel = (idxs == c).nonzero(as_tuple=False).squeeze(1)
            keeps.append(sel[nms(boxes[sel], scores[sel], thr)[1]])
        keep = torch.cat(keeps)
        keep = keep[scores[keep].argsort(descending=True, stable=True)]
        dets = torch.cat([boxes[keep], scores[keep, None]], 1)
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return dets, keep


def nms_rotated(dets, scores, iou_threshold, labels=None, clockwise=True):
    """mmcv.ops.nms_rotated(dets [N,5] = (cx, cy, w, h, angle), scores, iou_threshold, labels=None) ->
    (dets [M,6], keep [M]).  With `labels`, boxes of different labels never suppress each other (mmcv's
    multi_label path); `clockwise=False` flips the sign of the angle (mmcv >= 1.6)."""
    d = dets
    if not clockwise:
        d = dets.clone()
        d[:, 4] = -d[:, 4]
    if labels is None:
        out, keep = F.nms_rotated(d, scores, iou_threshold)
    else:                      # shift every class to its own far-away region: the IoU across classes is exactly 0
        span = (d[:, :2].abs().max() + d[:, 2:4].abs().max() + 1.0) * 2
        shifted = d.clone()
        shifted[:, 0] += labels.to(d.dtype) * span
        _, keep = F.nms_rotated(shifted, scores, iou_threshold)
    return torch.cat([dets[keep], scores[keep, None]], 1), keep


# --------------------------------------------------------------------------------------------- rotated boxes --
def box_iou_rotated(bboxes1, bboxes2, mode='iou', aligned=False, clockwise=True):
    """mmcv.ops.box_iou_rotated(bboxes1 [M,5], bboxes2 [N,5], mode, aligned) -> [M,N] (or [M] when aligned)."""
    assert mode in ('iou', 'iof')
    if not clockwise:
        bboxes1, bboxes2 = bboxes1.clone(), bboxes2.clone()
        bboxes1[:, 4], bboxes2[:, 4] = -bboxes1[:, 4], -bboxes2[:, 4]
    iou = F.box_iou_rotated(bboxes1, bboxes2, aligned)
    if mode == 'iou':
        return iou
    a1 = bboxes1[:, 2] * bboxes1[:, 3]                         # iof = inter / area1, inter = iou * (a1 + a2) / (1 + iou)
    a2 = bboxes2[:, 2] * bboxes2[:, 3]
    tot = (a1 + a2) if aligned else (a1[:, None] + a2[None, :])
    inter = iou * tot / (1 + iou)
    return inter / (a1 if aligned else a1[:, None]).clamp(min=1e-12)


def diff_iou_rotated_2d(box1, box2):
    """mmcv.ops.diff_iou_rotated_2d(box1 [B,N,5], box2 [B,N,5]) -> IoU [B,N], differentiable."""
    return F.diff_iou_rotated_2d(box1, box2)


def roi_align_rotated(input, rois, output_size, spatial_scale, sampling_ratio=0, aligned=True, clockwise=False):
    """mmcv.ops.roi_align_rotated(input, rois [K,6] = (batch, cx, cy, w, h, theta), output_size, spatial_scale,
    sampling_ratio, aligned, clockwise)."""
    o = output_size if isinstance(output_size, int) else output_size[0]
    return F.roi_align_rotated(input, rois, o, spatial_scale, sampling_ratio, aligned, clockwise)


class RoIAlignRotated(nn.Module):
    """mmcv.ops.RoIAlignRotated(output_size, spatial_scale, sampling_ratio=0, aligned=True, clockwise=False); the
    pre-1.5 keyword names `out_size` / `sample_num` the OBB config uses are accepted as mmcv does."""

    def __init__(self, output_size=None, spatial_scale=1.0, sampling_ratio=0, aligned=True, clockwise=False, out_size=None,
                 sample_num=None):
        super().__init__()
        output_size = out_size if output_size is None else output_size
        sampling_ratio = sample_num if sample_num is not None else sampling_ratio
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale, self.sampling_ratio = float(spatial_scale), int(sampling_ratio)
        self.aligned, self.clockwise = aligned, clockwise

    def forward(self, input, rois):
        return roi_align_rotated(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned,
                                 self.clockwise)


# -------------------------------------------------------------------------------------- deformable convolution --
def deform_conv2d(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deform_groups=1, bias=False,
                  im2col_step=32):
    """mmcv.ops.deform_conv2d (groups = 1, no bias - as DeformConv2d in mmcv)."""
    assert not bias
    return F.deform_conv2d(input, offset, weight, stride, padding, dilation, groups, deform_groups)


def modulated_deform_conv2d(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                            deform_groups=1):
    """mmcv.ops.modulated_deform_conv2d (DCNv2)."""
    return F.modulated_deform_conv2d(input, offset, mask, weight, bias, stride, padding, dilation, groups, deform_groups)


class DeformConv2d(nn.Module):
    """mmcv.ops.DeformConv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deform_groups,
    bias=False); forward(x, offset)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deform_groups=1,
                 bias=False, im2col_step=32):
        super().__init__()
        assert not bias and groups == 1
        k = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.stride, self.padding, self.dilation, self.deform_groups = stride, padding, dilation, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *k))
        nn.init.kaiming_uniform_(self.weight, nonlinearity='relu')

    def forward(self, x, offset):
        return deform_conv2d(x, offset, self.weight, self.stride, self.padding, self.dilation, 1, self.deform_groups)


class ModulatedDeformConv2d(nn.Module):
    """mmcv.ops.ModulatedDeformConv2d(...); forward(x, offset, mask)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deform_groups=1,
                 bias=True):
        super().__init__()
        assert groups == 1
        k = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.stride, self.padding, self.dilation, self.deform_groups = stride, padding, dilation, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *k))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, nonlinearity='relu')

    def forward(self, x, offset, mask):
        return modulated_deform_conv2d(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation, 1,
                                       self.deform_groups)
