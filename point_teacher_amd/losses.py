"""Loss modules with the reference's registry names and forward signatures
(/root/reference/HBB_TOD/mmdet/models/losses/{focal_loss,iou_loss,cross_entropy_loss,
smooth_l1_loss}.py), computed by the HIP kernels behind `functional`.

All reductions keep `avg_factor` as a tensor when it is one, so no loss forces a host
synchronisation (the reference calls len(pos_inds)/.item() several times per iteration)."""
import torch
import torch.nn as nn
import torch.nn.functional as TF

from .core import zero_like_scalar
from . import functional as F
from .registry import LOSSES


def _reduce(loss, weight, reduction, avg_factor):
    """losses/utils.py:28-54 weight_reduce_loss"""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'mean':
            return loss.mean()
        if reduction == 'sum':
            return loss.sum()
        return loss
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


@LOSSES.register_module()
class FocalLoss(nn.Module):
    """focal_loss.py:106-181.  `target` holds class indices in [0, C] (C = background)."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if reduction == 'none':
            loss = F.sigmoid_focal_loss_elem(pred, target, self.gamma, self.alpha)
            if weight is not None:
                loss = loss * weight.view(-1, 1)
            return self.loss_weight * loss
        if weight is not None and weight.dim() > 1:
            weight = weight.reshape(pred.size(0), -1)[:, 0]
        total = F.sigmoid_focal_loss_sum(pred, target, weight, self.gamma, self.alpha)
        if avg_factor is None:
            total = total / pred.numel() if reduction == 'mean' else total
        else:
            if reduction != 'mean':
                raise ValueError('avg_factor can not be used with reduction="sum"')
            total = total / avg_factor
        return self.loss_weight * total


class _IoULossBase(nn.Module):
    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def _elem(self, pred, target):
        raise NotImplementedError

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None and weight.dim() > 1:
            assert weight.shape == pred.shape
            weight = weight.mean(-1)
        # The reference returns (pred * weight).sum() == 0 when no weight is positive
        # (iou_loss.py:826-829 / :858-861).  That needs a host sync; the arithmetic below gives
        # the same 0 (and zero gradients) in that case because every term is multiplied by weight.
        return self.loss_weight * _reduce(self._elem(pred, target), weight, reduction, avg_factor)


@LOSSES.register_module()
class DIoULoss(_IoULossBase):
    """iou_loss.py:803-840 over diou_loss :139-189"""

    def _elem(self, pred, target):
        return F.diou_loss_elem(pred, target, self.eps)


@LOSSES.register_module()
class DN_DIoULoss(_IoULossBase):
    """iou_loss.py:842-880 over DN_diou_loss :398-465.  Quirk kept (:412): the `base_loss`
    term is the MEAN of the plain DIoU over all N rows, added to every row's min-over-9."""

    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0, hyper=0.2):
        super().__init__(eps, reduction, loss_weight)
        self.hyper = hyper

    def _elem(self, pred, target, base_mask=None):
        diou, dnmin = F.dn_diou_pair(pred, target, self.hyper, self.eps)
        if base_mask is None:
            base = diou.mean()
        else:   # dense-with-mask form used by the head: mean over the rows that exist in the reference
            base = torch.where(base_mask, diou, zero_like_scalar(diou)).sum() / base_mask.sum().clamp(min=1)   # (a cached 0-dim `other`: no fill launch)
        return (base + dnmin) / 2

    def forward_masked(self, pred, target, mask, weight, avg_factor):
        """Same value/gradients as forward(pred[mask], target[mask], weight[mask], avg_factor) but
        without the boolean gather (no host sync): rows outside `mask` contribute exactly 0."""
        elem = self._elem(pred, target, base_mask=mask)
        elem = torch.where(mask, elem * weight, zero_like_scalar(elem))
        return self.loss_weight * elem.sum() / avg_factor


def diou_forward_masked(loss_mod, pred, target, mask, weight, avg_factor):
    elem = loss_mod._elem(pred, target)
    elem = torch.where(mask, elem * weight, zero_like_scalar(elem))
    return loss_mod.loss_weight * elem.sum() / avg_factor


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """cross_entropy_loss.py:142-214.  Only the paths the head uses are provided natively:
    use_sigmoid=True on 1-D logits with float targets (centerness)."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        assert (use_sigmoid is False) or (use_mask is False)
        self.use_sigmoid, self.use_mask, self.reduction = use_sigmoid, use_mask, reduction
        self.loss_weight, self.class_weight = loss_weight, class_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if self.use_sigmoid:
            if cls_score.dim() != label.dim():
                label = TF.one_hot(label.clamp(min=0), cls_score.size(-1) + 1)[..., :cls_score.size(-1)]
                if weight is not None:
                    weight = weight.view(-1, 1).expand(weight.size(0), cls_score.size(-1))
            loss = TF.binary_cross_entropy_with_logits(cls_score, label.float(), reduction='none')
        elif self.use_mask:
            raise NotImplementedError('mask cross entropy is not on the Point-Teacher path')
        else:
            loss = TF.cross_entropy(cls_score, label, reduction='none')
        if weight is not None:
            weight = weight.float()
        return self.loss_weight * _reduce(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    """smooth_l1_loss.py - constructed by the head (fcos_head_p2b_ts.py:181,187) but never called."""

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        reduction = reduction_override if reduction_override else self.reduction
        loss = TF.smooth_l1_loss(pred, target, beta=self.beta, reduction='none')
        return self.loss_weight * _reduce(loss, weight, reduction, avg_factor)
