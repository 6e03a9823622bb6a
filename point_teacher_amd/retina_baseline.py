"""The anchor-based RetinaNet baseline of the comparison (second entry):
`configs/baselines/aitodv2_retinanet_r50_1x.py` = `RetinaNet` + `RetinaHead` + `AnchorGenerator` + `MaxIoUAssigner`.

    AnchorGenerator   /root/reference/HBB_TOD/mmdet/core/anchor/anchor_generator.py:12-510
    MaxIoUAssigner    /root/reference/HBB_TOD/mmdet/core/bbox/assigners/max_iou_assigner.py:10-212
    AnchorHead        /root/reference/HBB_TOD/mmdet/models/dense_heads/anchor_head.py:16-760   (targets, loss, get_bboxes)
    RetinaHead        /root/reference/HBB_TOD/mmdet/models/dense_heads/retina_head.py:9-114
    bbox2delta        /root/reference/HBB_TOD/mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:98-140

The reference builds, per image, a [G, A] IoU matrix over ~120 000 anchors (144 MB at 300 boxes) and walks it with a python
loop over the boxes; `pt_max_iou_assign` assigns the whole batch in two launches without storing it.  Targets and losses
stay dense over all B*A anchors with masks (no `nonzero`, no host synchronisation)."""
import numpy as np
import torch
import torch.nn as nn

from . import functional as F
from .core import multiclass_nms
from .detectors import Student_FCOS
from .nn_modules import ConvModule
from .registry import BBOX_ASSIGNERS, DETECTORS, HEADS, LOSSES, build_assigner, build_loss


@LOSSES.register_module()
class L1Loss(nn.Module):
    """models/losses/smooth_l1_loss.py L1Loss: |pred - target| * weight, summed / avg_factor."""

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        loss = torch.abs(pred - target)
        if weight is not None:
            loss = loss * weight
        reduction = reduction_override or self.reduction
        if avg_factor is None:
            loss = loss.mean() if reduction == 'mean' else (loss.sum() if reduction == 'sum' else loss)
        else:
            assert reduction == 'mean'
            loss = loss.sum() / avg_factor
        return self.loss_weight * loss


class AnchorGenerator:
    """anchor_generator.py: octave scales x ratios base anchors (scale_major), shifted over every feature map."""

    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True, octave_base_scale=None,
                 scales_per_octave=None, centers=None, center_offset=0., **kwargs):
        assert centers is None and 0 <= center_offset <= 1
        self.strides = [(s, s) if isinstance(s, int) else tuple(s) for s in strides]
        self.base_sizes = [min(s) for s in self.strides] if base_sizes is None else base_sizes
        assert ((octave_base_scale is not None and scales_per_octave is not None) ^ (scales is not None))
        if scales is not None:
            self.scales = torch.Tensor(scales)
        else:
            octave_scales = np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)])
            self.scales = torch.Tensor(octave_scales * octave_base_scale)
        self.ratios = torch.Tensor(ratios)
        self.scale_major, self.center_offset = scale_major, center_offset
        self.base_anchors = [self.gen_single_level_base_anchors(b, self.scales, self.ratios) for b in self.base_sizes]
        self._cache = {}

    @property
    def num_base_anchors(self):
        return [b.size(0) for b in self.base_anchors]

    @property
    def num_levels(self):
        return len(self.strides)

    def gen_single_level_base_anchors(self, base_size, scales, ratios, center=None):
        w = h = base_size
        x_center, y_center = self.center_offset * w, self.center_offset * h
        h_ratios = torch.sqrt(ratios)
        w_ratios = 1 / h_ratios
        if self.scale_major:
            ws = (w * w_ratios[:, None] * scales[None, :]).view(-1)
            hs = (h * h_ratios[:, None] * scales[None, :]).view(-1)
        else:
            ws = (w * scales[:, None] * w_ratios[None, :]).view(-1)
            hs = (h * scales[:, None] * h_ratios[None, :]).view(-1)
        return torch.stack([x_center - 0.5 * ws, y_center - 0.5 * hs, x_center + 0.5 * ws, y_center + 0.5 * hs], dim=-1)

    def grid_anchors(self, featmap_sizes, device='cuda'):
        """-> per-level [H*W*A, 4], order (y, x, base anchor); cached per (sizes, device)."""
        key = (tuple((int(h), int(w)) for h, w in featmap_sizes), str(device))
        if key not in self._cache:
            out = []
            for (fh, fw), base, stride in zip(featmap_sizes, self.base_anchors, self.strides):
                sx = torch.arange(0, fw, device=device) * stride[0]
                sy = torch.arange(0, fh, device=device) * stride[1]
                xx, yy = sx.repeat(len(sy)), sy.view(-1, 1).repeat(1, len(sx)).view(-1)
                base = base.to(device)
                shifts = torch.stack([xx, yy, xx, yy], dim=-1).to(base.dtype)
                out.append((base[None, :, :] + shifts[:, None, :]).view(-1, 4))
            self._cache[key] = out
        return self._cache[key]

    def valid_flags(self, featmap_sizes, pad_shape, device='cuda'):
        flags = []
        for i, (fh, fw) in enumerate(featmap_sizes):
            s = self.strides[i]
            h, w = pad_shape[:2]
            vh, vw = min(int(np.ceil(h / s[1])), fh), min(int(np.ceil(w / s[0])), fw)
            vx = torch.zeros(fw, dtype=torch.bool, device=device)
            vy = torch.zeros(fh, dtype=torch.bool, device=device)
            vx[:vw] = 1
            vy[:vh] = 1
            v = vx.repeat(fh) & vy.view(-1, 1).repeat(1, fw).view(-1)
            flags.append(v[:, None].expand(v.size(0), self.num_base_anchors[i]).contiguous().view(-1))
        return flags


@BBOX_ASSIGNERS.register_module()
class MaxIoUAssigner:
    """max_iou_assigner.py through `pt_max_iou_assign`; `assign_batch` is the whole-batch form the head uses."""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, gpu_assign_thr=-1, iou_calculator=dict(type='BboxOverlaps2D')):
        if ignore_iof_thr > 0:
            raise NotImplementedError('ignore_iof_thr > 0 (crowd regions) is not used by the AI-TOD configs')
        assert iou_calculator.get('type', 'BboxOverlaps2D') == 'BboxOverlaps2D'
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all, self.match_low_quality = gt_max_assign_all, match_low_quality
        self.gpu_assign_thr = gpu_assign_thr            # the reference moves images with more boxes to the CPU; nothing to do here

    def assign_batch(self, anchors, gt_bboxes_list):
        """-> (assigned_gt_inds int32 [B, A], max_overlaps [B, A], off int32 [B+1], concatenated boxes)"""
        off, _ = F.make_offsets([int(b.shape[0]) for b in gt_bboxes_list], anchors.device)
        gb = torch.cat(gt_bboxes_list).float()
        inds, mo = F.max_iou_assign(anchors, gb, off, len(gt_bboxes_list), self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou,
                                    self.match_low_quality, self.gt_max_assign_all)
        return inds, mo, off, gb

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        inds, mo, _, _ = self.assign_batch(bboxes, [gt_bboxes])
        return inds[0].long(), mo[0]


def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    """delta_xywh_bbox_coder.py:98-140"""
    px, py = (proposals[..., 0] + proposals[..., 2]) * 0.5, (proposals[..., 1] + proposals[..., 3]) * 0.5
    pw, ph = proposals[..., 2] - proposals[..., 0], proposals[..., 3] - proposals[..., 1]
    gx, gy = (gt[..., 0] + gt[..., 2]) * 0.5, (gt[..., 1] + gt[..., 3]) * 0.5
    gw, gh = gt[..., 2] - gt[..., 0], gt[..., 3] - gt[..., 1]
    deltas = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)
    return deltas.sub_(deltas.new_tensor(means).unsqueeze(0)).div_(deltas.new_tensor(stds).unsqueeze(0))


@HEADS.register_module()
class RetinaHead(nn.Module):
    """retina_head.py + anchor_head.py with mmdet's parameter names (cls_convs / reg_convs / retina_cls / retina_reg)."""

    def __init__(self, num_classes, in_channels, stacked_convs=4, feat_channels=256, conv_cfg=None, norm_cfg=None,
                 anchor_generator=dict(type='AnchorGenerator', octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0],
                                       strides=[8, 16, 32, 64, 128]),
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=True, target_means=(.0, .0, .0, .0),
                                 target_stds=(1.0, 1.0, 1.0, 1.0)),
                 reg_decoded_bbox=False,
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox=dict(type='L1Loss', loss_weight=1.0), train_cfg=None, test_cfg=None, init_cfg=None, **kwargs):
        super().__init__()
        assert not reg_decoded_bbox and conv_cfg is None and norm_cfg is None
        assert bbox_coder.get('type') == 'DeltaXYWHBBoxCoder'
        self.num_classes = self.cls_out_channels = num_classes
        self.in_channels, self.feat_channels, self.stacked_convs = in_channels, feat_channels, stacked_convs
        ag = dict(anchor_generator)
        assert ag.pop('type') == 'AnchorGenerator'
        self.anchor_generator = AnchorGenerator(**ag)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self.means, self.stds = tuple(bbox_coder.get('target_means', (0.,) * 4)), tuple(bbox_coder.get('target_stds', (1.,) * 4))
        self.loss_cls, self.loss_bbox = build_loss(loss_cls), build_loss(loss_bbox)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.sampling = False                                   # focal loss: PseudoSampler, every anchor contributes
        self._flag_cache = {}
        if train_cfg:
            self.assigner = build_assigner(train_cfg['assigner'])
            assert train_cfg.get('allowed_border', -1) < 0, 'allowed_border >= 0 is not used by the AI-TOD configs'
            self.pos_weight = train_cfg.get('pos_weight', -1)
        self.cls_convs = nn.ModuleList([ConvModule(in_channels if i == 0 else feat_channels, feat_channels, 3, padding=1)
                                        for i in range(stacked_convs)])
        self.reg_convs = nn.ModuleList([ConvModule(in_channels if i == 0 else feat_channels, feat_channels, 3, padding=1)
                                        for i in range(stacked_convs)])
        self.retina_cls = nn.Conv2d(feat_channels, self.num_anchors * self.cls_out_channels, 3, padding=1)
        self.retina_reg = nn.Conv2d(feat_channels, self.num_anchors * 4, 3, padding=1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.retina_cls.bias, float(-np.log((1 - 0.01) / 0.01)))

    def forward_single(self, x):
        cls_feat = reg_feat = x
        for l in self.cls_convs:
            cls_feat = l(cls_feat)
        for l in self.reg_convs:
            reg_feat = l(reg_feat)
        return self.retina_cls(cls_feat), self.retina_reg(reg_feat)

    def forward(self, feats):
        outs = [self.forward_single(x) for x in feats]
        return [o[0] for o in outs], [o[1] for o in outs]

    # ------------------------------------------------------------------------------------- targets --
    def _valid_flags(self, featmap_sizes, pad_shape, dev):
        key = (tuple((int(h), int(w)) for h, w in featmap_sizes), tuple(int(v) for v in pad_shape[:2]), str(dev))
        if key not in self._flag_cache:
            self._flag_cache[key] = torch.cat(self.anchor_generator.valid_flags(featmap_sizes, pad_shape, dev))
        return self._flag_cache[key]

    def get_targets(self, anchors, gt_bboxes, gt_labels, img_metas, featmap_sizes):
        """anchor_head.py:181-380 for the whole batch -> labels [B, A] (num_classes = background), label_weights [B, A],
        bbox_targets [B, A, 4], bbox_weights [B, A, 1] (float masks) and num_total_pos (device scalar)."""
        B, A = len(gt_bboxes), anchors.shape[0]
        dev = anchors.device
        inds, _, off, gb = self.assigner.assign_batch(anchors, gt_bboxes)
        # anchors beyond the padded image are invalid (anchor_inside_flags with allowed_border < 0 keeps every valid one)
        valid = torch.stack([self._valid_flags(featmap_sizes, m.get('pad_shape', m['img_shape']), dev) for m in img_metas])
        pos = (inds > 0) & valid
        neg = (inds == 0) & valid
        gl = torch.cat(gt_labels)
        if gb.shape[0]:
            gi = (inds.long() - 1).clamp(min=0) + off[:-1].long()[:, None]                   # row of the assigned box
            gi = gi.clamp(max=gb.shape[0] - 1)
            labels = torch.where(pos, gl[gi], torch.full_like(gi, self.num_classes))
            tg = bbox2delta(anchors[None].expand(B, A, 4), gb[gi], self.means, self.stds)
            tg = torch.where(pos[..., None], tg, torch.zeros_like(tg))
        else:
            labels = torch.full((B, A), self.num_classes, dtype=torch.long, device=dev)
            tg = torch.zeros((B, A, 4), dtype=torch.float32, device=dev)
        pw = 1.0 if self.pos_weight <= 0 else float(self.pos_weight)
        label_weights = pos.float() * pw + neg.float()
        num_total_pos = pos.sum(1).clamp(min=1).sum()                                         # sum_i max(#pos_i, 1)
        return labels, label_weights, tg, pos.float()[..., None], num_total_pos

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
        """anchor_head.py:382-500: one (loss_cls, loss_bbox) pair per level, normalised by the batch's positive count."""
        sizes = [f.size()[-2:] for f in cls_scores]
        dev = cls_scores[0].device
        lvl_anchors = self.anchor_generator.grid_anchors(sizes, dev)
        anchors = torch.cat(lvl_anchors)
        labels, lw, tg, bw, num_pos = self.get_targets(anchors, gt_bboxes, gt_labels, img_metas, sizes)
        num_pos = num_pos.float()
        losses_cls, losses_bbox = [], []
        o = 0
        for cs, bp, la in zip(cls_scores, bbox_preds, lvl_anchors):
            n = la.shape[0]
            cs = cs.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
            bp = bp.permute(0, 2, 3, 1).reshape(-1, 4)
            losses_cls.append(self.loss_cls(cs, labels[:, o:o + n].reshape(-1), lw[:, o:o + n].reshape(-1), avg_factor=num_pos))
            losses_bbox.append(self.loss_bbox(bp, tg[:, o:o + n].reshape(-1, 4), bw[:, o:o + n].reshape(-1, 1), avg_factor=num_pos))
            o += n
        return dict(loss_cls=losses_cls, loss_bbox=losses_bbox)

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, **kwargs):
        return self.loss(*self(x), gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=gt_bboxes_ignore)

    # ----------------------------------------------------------------------------------- inference --
    def simple_test(self, feats, img_metas, rescale=False):
        return self.get_bboxes(*self(feats), img_metas, rescale=rescale)

    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg=None, rescale=False, with_nms=True):
        """anchor_head.py:503-739 (batched `_get_bboxes`, sigmoid scores, nms_pre per level by the best class score)."""
        cfg = self.test_cfg if cfg is None else cfg
        B = cls_scores[0].shape[0]
        sizes = [f.size()[-2:] for f in cls_scores]
        lvl_anchors = self.anchor_generator.grid_anchors(sizes, cls_scores[0].device)
        nms_pre = cfg.get('nms_pre', -1)
        mb, ms = [], []
        for cs, bp, anchors in zip(cls_scores, bbox_preds, lvl_anchors):
            scores = cs.detach().permute(0, 2, 3, 1).reshape(B, -1, self.cls_out_channels).sigmoid()
            bp = bp.detach().permute(0, 2, 3, 1).reshape(B, -1, 4)
            anchors = anchors[None].expand(B, -1, 4)
            if 0 < nms_pre < scores.shape[1]:
                _, topk = scores.max(-1)[0].topk(nms_pre)
                bi = torch.arange(B, device=scores.device).view(-1, 1).expand_as(topk)
                anchors, bp, scores = anchors[bi, topk], bp[bi, topk], scores[bi, topk]
            mb.append(torch.stack([F.delta2bbox(anchors[b].contiguous(), bp[b].contiguous(), max_shape=img_metas[b]['img_shape'])
                                   for b in range(B)]))
            ms.append(scores)
        bb, sc = torch.cat(mb, 1), torch.cat(ms, 1)
        if rescale:
            bb = bb / bb.new_tensor(np.stack([np.asarray(m['scale_factor'], np.float32) for m in img_metas])).unsqueeze(1)
        sc = torch.cat([sc, sc.new_zeros(B, sc.shape[1], 1)], dim=-1)
        if not with_nms:
            return list(zip(bb, sc))
        return [multiclass_nms(b, s, cfg['score_thr'], cfg['nms'], cfg['max_per_img']) for b, s in zip(bb, sc)]


@DETECTORS.register_module()
class RetinaNet(Student_FCOS):
    """detectors/retinanet.py over single_stage.py:80-104."""

    def __init__(self, backbone, neck=None, neck_agg=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None,
                 init_cfg=None):
        super().__init__(backbone, neck, neck_agg, bbox_head, None, train_cfg, test_cfg, pretrained, init_cfg)

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
        x = self.extract_feat(img.to(torch.float))
        return self.bbox_head.forward_train(x, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore)
