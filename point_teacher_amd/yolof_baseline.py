"""The YOLOF baseline of the comparison (fourth entry): `configs/baselines/aitodv2_yolof_r50_1x.py` =
`YOLOF` + `DilatedEncoder` + `YOLOFHead` + `UniformAssigner`, trained with `paramwise_cfg=dict(norm_decay_mult=0.,
custom_keys={'backbone': dict(lr_mult=1. / 3)})` (three parameter groups of the flat optimizer, runtime.FlatParams).

    YOLOF            /root/reference/HBB_TOD/mmdet/models/detectors/yolof.py:5-19
    DilatedEncoder   /root/reference/HBB_TOD/mmdet/models/necks/dilated_encoder.py:9-104   (C5 -> 512 channels, 4 dilated residual blocks)
    YOLOFHead        /root/reference/HBB_TOD/mmdet/models/dense_heads/yolof_head.py:44-414 (implicit objectness, uniform matching)
    UniformAssigner  /root/reference/HBB_TOD/mmdet/core/bbox/assigners/uniform_assigner.py:10-134
    delta2bbox       /root/reference/HBB_TOD/mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:144-262 with add_ctr_clamp

The reference loops over the images in python, gathers the positives with `nonzero`, resolves the duplicate targets of its scatter
by whatever the indexing kernel does (on the CPU: the last write wins) and reads the positive count back with `.item()`
(yolof_head.py:196).  Here the assignment is a handful of device ops per image with a DETERMINISTIC last-write-wins scatter, the
regression loss keeps every candidate with a 0 / 1 weight, and nothing synchronises the host.  Kept quirk: the config pairs the
stride-32 C5 map with `strides=[8]` anchors (4 / 8 / 16 px on a stride-8 grid over a 25 x 25 map)."""
import numpy as np
import torch
import torch.nn as nn

from . import functional as F
from .detectors import Student_FCOS
from .nn_modules import ConvModule
from .registry import BBOX_ASSIGNERS, DETECTORS, HEADS, NECKS
from .retina_baseline import RetinaHead

INF = 1e8


class _EncoderBottleneck(nn.Module):
    """dilated_encoder.py:9-47: 1x1 -> dilated 3x3 -> 1x1, each conv + BatchNorm (training mode) + ReLU, + identity."""

    def __init__(self, in_channels, mid_channels, dilation, norm_cfg=dict(type='BN', requires_grad=True)):
        super().__init__()
        self.conv1 = ConvModule(in_channels, mid_channels, 1, bias='auto', norm_cfg=norm_cfg)
        self.conv2 = ConvModule(mid_channels, mid_channels, 3, padding=dilation, dilation=dilation, bias='auto', norm_cfg=norm_cfg)
        self.conv3 = ConvModule(mid_channels, in_channels, 1, bias='auto', norm_cfg=norm_cfg)

    def forward(self, x):
        return self.conv3(self.conv2(self.conv1(x))) + x


@NECKS.register_module()
class DilatedEncoder(nn.Module):
    """dilated_encoder.py:50-104; parameter names as mmdet's (lateral_conv / lateral_norm / fpn_conv / fpn_norm /
    dilated_encoder_blocks.N.convM.{conv,bn})."""

    def __init__(self, in_channels, out_channels, block_mid_channels, num_residual_blocks):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.block_dilations = [2, 4, 6, 8]
        self.lateral_conv = nn.Conv2d(in_channels, out_channels, 1)
        self.lateral_norm = nn.BatchNorm2d(out_channels)
        self.fpn_conv = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.fpn_norm = nn.BatchNorm2d(out_channels)
        self.dilated_encoder_blocks = nn.Sequential(*[_EncoderBottleneck(out_channels, block_mid_channels, self.block_dilations[i])
                                                      for i in range(num_residual_blocks)])
        self.init_weights()

    def init_weights(self):
        """:89-99: caffe2 xavier (kaiming_uniform, a = 1) for the two projection convs, N(0, 0.01) inside the blocks."""
        for m in (self.lateral_conv, self.fpn_conv):
            nn.init.kaiming_uniform_(m.weight, a=1, mode='fan_in', nonlinearity='leaky_relu')
            nn.init.constant_(m.bias, 0)
        for m in self.dilated_encoder_blocks.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, feature):
        out = self.lateral_norm(self.lateral_conv(feature[-1]))
        out = self.fpn_norm(self.fpn_conv(out))
        return self.dilated_encoder_blocks(out),


def _cxcywh(b):
    return torch.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], -1)


def delta2bbox_ctr_clamp(rois, deltas, ctr_clamp=32, wh_ratio_clip=16 / 1000):
    """delta_xywh_bbox_coder.py:144-262 with means 0 / stds 1, add_ctr_clamp=True, no max_shape: the centre shift is clamped
    to +-ctr_clamp pixels, the size factor only from above."""
    px, py = (rois[:, 0] + rois[:, 2]) * 0.5, (rois[:, 1] + rois[:, 3]) * 0.5
    pw, ph = rois[:, 2] - rois[:, 0], rois[:, 3] - rois[:, 1]
    mr = abs(float(np.log(wh_ratio_clip)))
    dxw = (pw * deltas[:, 0]).clamp(min=-ctr_clamp, max=ctr_clamp)
    dyh = (ph * deltas[:, 1]).clamp(min=-ctr_clamp, max=ctr_clamp)
    gw, gh = pw * deltas[:, 2].clamp(max=mr).exp(), ph * deltas[:, 3].clamp(max=mr).exp()
    gx, gy = px + dxw, py + dyh
    return torch.stack([gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5], -1)


@BBOX_ASSIGNERS.register_module()
class UniformAssigner:
    """uniform_assigner.py:10-134: every box takes its `match_times` L1-nearest decoded predictions AND its `match_times`
    L1-nearest anchors ((cx, cy, w, h) distance) as positives; a candidate whose ANCHOR overlaps its box by less than
    pos_ignore_thr is ignored, a prediction that overlaps any box by more than neg_ignore_thr is ignored as a negative."""

    def __init__(self, pos_ignore_thr, neg_ignore_thr, match_times=4, iou_calculator=dict(type='BboxOverlaps2D')):
        self.pos_ignore_thr, self.neg_ignore_thr, self.match_times = pos_ignore_thr, neg_ignore_thr, match_times

    def assign(self, bbox_pred, anchor, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        """-> dict(gt_inds [n] (-1 ignore, 0 negative, i + 1 positive), labels [n] (-1 = none), pos_idx [2 k G] bool,
        pos_predicted_boxes [2 k G, 4], target_boxes [2 k G, 4]) - the AssignResult and its extra properties."""
        n, G, k = bbox_pred.shape[0], gt_bboxes.shape[0], self.match_times
        dev = bbox_pred.device
        gt_inds = torch.zeros(n, dtype=torch.long, device=dev)
        if G == 0 or n == 0:
            return dict(gt_inds=gt_inds, labels=torch.full((n,), -1, dtype=torch.long, device=dev),
                        pos_idx=torch.zeros(0, dtype=torch.bool, device=dev), pos_predicted_boxes=bbox_pred.new_zeros((0, 4)),
                        target_boxes=bbox_pred.new_zeros((0, 4)))
        gt_c = _cxcywh(gt_bboxes)
        cost = torch.cdist(_cxcywh(bbox_pred.detach()), gt_c, p=1)
        cost_a = torch.cdist(_cxcywh(anchor), gt_c, p=1)
        # Stable top-k, lowest index first among ties: the 8- and the 16-pixel anchor of one cell are EXACTLY equally far from every
        # box whose sides lie between 8 and 16 px, `torch.topk` orders such ties differently on CPU and GPU, and the order decides
        # which duplicate write below comes last (tests/golden/yolof_baseline.npz is generated under the same rule).
        idx = torch.sort(cost, dim=0, stable=True)[1][:k]                       # [k, G]
        idx_a = torch.sort(cost_a, dim=0, stable=True)[1][:k]
        indexes = torch.cat((idx, idx_a), dim=1).reshape(-1)                    # row-major over [k, 2 G]
        pred_ov = F.bbox_overlaps(bbox_pred.detach(), gt_bboxes)
        anchor_ov = F.bbox_overlaps(anchor, gt_bboxes)
        gt_inds = torch.where(pred_ov.max(dim=1)[0] > self.neg_ignore_thr, torch.full_like(gt_inds, -1), gt_inds)
        pos_gt = torch.arange(0, G, device=dev).repeat(k * 2)
        pos_ignore = anchor_ov[indexes, pos_gt] < self.pos_ignore_thr
        val = torch.where(pos_ignore, torch.full_like(pos_gt, -1), pos_gt + 1)
        # `assigned_gt_inds[indexes] = val` with duplicate indices: the reference's CPU kernel lets the LAST write win; an
        # index_put on the device would pick any.  The winner of every target is the write with the largest position.
        # Every write to a target carries the value of that target's LAST write, so the order the device applies them in is
        # immaterial.
        order = torch.arange(indexes.numel(), device=dev)
        last = torch.zeros(n, dtype=torch.long, device=dev).scatter_reduce(0, indexes, order, 'amax', include_self=False)
        gt_inds = gt_inds.scatter(0, indexes, val[last[indexes]])
        labels = torch.full((n,), -1, dtype=torch.long, device=dev)
        if gt_labels is not None:
            labels = torch.where(gt_inds > 0, gt_labels[(gt_inds - 1).clamp(min=0)], labels)
        return dict(gt_inds=gt_inds, labels=labels, pos_idx=~pos_ignore, pos_predicted_boxes=bbox_pred[indexes],
                    target_boxes=gt_bboxes[pos_gt])


@HEADS.register_module()
class YOLOFHead(RetinaHead):
    """yolof_head.py:44-414 over anchor_head.py (inference path shared with RetinaHead); parameter names as mmdet's
    (cls_subnet.N.{conv,bn} / bbox_subnet / cls_score / bbox_pred / object_pred)."""

    def __init__(self, num_classes, in_channels, num_cls_convs=2, num_reg_convs=4, norm_cfg=dict(type='BN', requires_grad=True),
                 anchor_generator=dict(type='AnchorGenerator', ratios=[1.0], scales=[0.5, 1, 2], strides=[8]),
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1., 1., 1., 1.],
                                 add_ctr_clamp=True, ctr_clamp=32),
                 reg_decoded_bbox=True,
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox=dict(type='DIoULoss', loss_weight=1.0), train_cfg=None, test_cfg=None, init_cfg=None, **kwargs):
        assert reg_decoded_bbox and bbox_coder.get('add_ctr_clamp', False), 'YOLOF decodes before the loss, with the centre clamp'
        coder = {k: v for k, v in bbox_coder.items() if k not in ('add_ctr_clamp', 'ctr_clamp')}
        super().__init__(num_classes, in_channels, stacked_convs=0, feat_channels=in_channels, anchor_generator=anchor_generator,
                         bbox_coder=coder, reg_decoded_bbox=False, loss_cls=loss_cls, loss_bbox=loss_bbox, train_cfg=train_cfg,
                         test_cfg=test_cfg)
        del self.cls_convs, self.reg_convs, self.retina_cls, self.retina_reg
        self.ctr_clamp = bbox_coder.get('ctr_clamp', 32)
        self.cls_subnet = nn.Sequential(*[ConvModule(in_channels, in_channels, 3, padding=1, bias='auto', norm_cfg=norm_cfg)
                                          for _ in range(num_cls_convs)])
        self.bbox_subnet = nn.Sequential(*[ConvModule(in_channels, in_channels, 3, padding=1, bias='auto', norm_cfg=norm_cfg)
                                           for _ in range(num_reg_convs)])
        self.cls_score = nn.Conv2d(in_channels, self.num_anchors * num_classes, 3, padding=1)
        self.bbox_pred = nn.Conv2d(in_channels, self.num_anchors * 4, 3, padding=1)
        self.object_pred = nn.Conv2d(in_channels, self.num_anchors, 3, padding=1)
        for m in self.modules():                                                # :107-116
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.cls_score.bias, float(-np.log((1 - 0.01) / 0.01)))

    def forward_single(self, feature):
        """:118-134: the objectness is folded into the class logits ("implicit objectness")."""
        cls_score = self.cls_score(self.cls_subnet(feature))
        N, _, H, W = cls_score.shape
        cls_score = cls_score.view(N, -1, self.num_classes, H, W)
        reg_feat = self.bbox_subnet(feature)
        bbox_reg = self.bbox_pred(reg_feat)
        return self.implicit_objectness(cls_score, self.object_pred(reg_feat)), bbox_reg

    @staticmethod
    def implicit_objectness(cls_score, objectness):
        """[N, A, C, H, W] class logits and [N, A, H, W] objectness -> [N, A*C, H, W]: log-odds of sigmoid(cls) * sigmoid(obj)."""
        N, A, C, H, W = cls_score.shape
        objectness = objectness.view(N, -1, 1, H, W)
        norm = cls_score + objectness - torch.log(1. + torch.clamp(cls_score.exp(), max=INF) + torch.clamp(objectness.exp(), max=INF))
        return norm.view(N, -1, H, W)

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
        """:136-222 (+ get_targets / _get_targets_single :223-414) without `nonzero`, `.item()` or filtered lists."""
        assert len(cls_scores) == 1 and self.anchor_generator.num_levels == 1
        sizes = [f.size()[-2:] for f in cls_scores]
        dev = cls_scores[0].device
        anchors = self.anchor_generator.grid_anchors(sizes, dev)[0]
        B, A = cls_scores[0].shape[0], anchors.shape[0]
        bp = bbox_preds[0].permute(0, 2, 3, 1).reshape(B, -1, 4)
        labels, weights, pp, tt, ww, npos = [], [], [], [], [], 0
        for b in range(B):
            # (allowed_border < 0: every anchor of the padded map is inside, anchor_head.py anchor_inside_flags)
            dec = delta2bbox_ctr_clamp(anchors, bp[b], self.ctr_clamp)
            r = self.assigner.assign(dec, anchors, gt_bboxes[b], None, gt_labels[b])
            pos, neg = r['gt_inds'] > 0, r['gt_inds'] == 0
            labels.append(torch.where(pos, r['labels'], torch.full_like(r['labels'], self.num_classes)))
            pw = 1.0 if self.pos_weight <= 0 else float(self.pos_weight)
            weights.append(pos.float() * pw + neg.float())
            npos = npos + pos.sum().clamp(min=1)
            pp.append(r['pos_predicted_boxes']); tt.append(r['target_boxes']); ww.append(r['pos_idx'].float())
        from .core import reduce_mean
        avg = reduce_mean(npos.float()).clamp(min=1.0)
        cs = cls_scores[0].permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
        loss_cls = self.loss_cls(cs, torch.cat(labels), torch.cat(weights), avg_factor=avg)
        pp, tt, ww = torch.cat(pp), torch.cat(tt), torch.cat(ww)
        loss_bbox = self.loss_bbox(pp, tt, ww, avg_factor=avg) if pp.shape[0] else pp.sum() * 0
        return dict(loss_cls=loss_cls, loss_bbox=loss_bbox)

    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg=None, rescale=False, with_nms=True):
        """anchor_head.py:503-739 with this head's decoder (centre clamp, clipped to the image)."""
        from .core import multiclass_nms
        cfg = self.test_cfg if cfg is None else cfg
        B = cls_scores[0].shape[0]
        anchors = self.anchor_generator.grid_anchors([f.size()[-2:] for f in cls_scores], cls_scores[0].device)[0]
        scores = cls_scores[0].detach().permute(0, 2, 3, 1).reshape(B, -1, self.cls_out_channels).sigmoid()
        bp = bbox_preds[0].detach().permute(0, 2, 3, 1).reshape(B, -1, 4)
        nms_pre = cfg.get('nms_pre', -1)
        out = []
        for b in range(B):
            sc, bb, an = scores[b], bp[b], anchors
            if 0 < nms_pre < sc.shape[0]:
                _, topk = sc.max(-1)[0].topk(nms_pre)
                sc, bb, an = sc[topk], bb[topk], an[topk]
            box = delta2bbox_ctr_clamp(an, bb, self.ctr_clamp)
            h, w = img_metas[b]['img_shape'][:2]
            box = torch.stack([box[:, 0].clamp(0, w), box[:, 1].clamp(0, h), box[:, 2].clamp(0, w), box[:, 3].clamp(0, h)], -1)
            if rescale:
                box = box / box.new_tensor(np.asarray(img_metas[b]['scale_factor'], np.float32))
            sc = torch.cat([sc, sc.new_zeros(sc.shape[0], 1)], dim=-1)
            out.append((box, sc) if not with_nms else multiclass_nms(box, sc, cfg['score_thr'], cfg['nms'], cfg['max_per_img']))
        return out


@DETECTORS.register_module()
class YOLOF(Student_FCOS):
    """detectors/yolof.py over single_stage.py:80-104."""

    def __init__(self, backbone, neck, neck_agg=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__(backbone, neck, neck_agg, bbox_head, None, train_cfg, test_cfg, pretrained, init_cfg)

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
        x = self.extract_feat(img.to(torch.float))
        return self.bbox_head.forward_train(x, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore)
