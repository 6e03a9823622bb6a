"""The two-stage Faster R-CNN baseline of the comparison (third entry):
`configs/baselines/aitodv2_faster_rcnn_r50_1x.py` = `FasterRCNN` + `RPNHead` + `StandardRoIHead` / `Shared2FCBBoxHead` +
`MaxIoUAssigner` + `RandomSampler`, with mmdet's parameter names (`rpn_head.rpn_conv / rpn_cls / rpn_reg`,
`roi_head.bbox_head.shared_fcs / fc_cls / fc_reg`).

    RandomSampler     /root/reference/HBB_TOD/mmdet/core/bbox/samplers/base_sampler.py:34-101, random_sampler.py:31-81
    RPNHead           /root/reference/HBB_TOD/mmdet/models/dense_heads/rpn_head.py:14-249 over anchor_head.py:181-500
    BBoxHead          /root/reference/HBB_TOD/mmdet/models/roi_heads/bbox_heads/bbox_head.py:122-380, convfc_bbox_head.py:27-237
    StandardRoIHead   /root/reference/HBB_TOD/mmdet/models/roi_heads/standard_roi_head.py:55-160, test_mixins.py (simple_test_bboxes)
    TwoStageDetector  /root/reference/HBB_TOD/mmdet/models/detectors/two_stage.py:96-190

The kernels are the path's own: `pt_max_iou_assign` (anchors / proposals x boxes without the IoU matrix), `pt_nms_sorted`
(level- / class-aware), `pt_delta2bbox`, `pt_roi_align_*` per pyramid level.  The random sub-sampling keeps the reference's
form (nonzero -> randperm -> index; the index sets are data dependent, so each image synchronises the host twice) - this model
is a comparison baseline, not the training hot path."""
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import functional as F
from .core import bbox2result, bbox2roi, multiclass_nms
from .detectors import Student_FCOS
from .registry import BBOX_SAMPLERS, DETECTORS, HEADS, build_assigner, build_from_cfg, build_head, build_loss, build_roi_extractor
from .retina_baseline import AnchorGenerator, bbox2delta


@BBOX_SAMPLERS.register_module()
class RandomSampler:
    """base_sampler.py sample + random_sampler.py: at most num * pos_fraction positives, the rest negatives, each drawn
    with `randperm` (replaceable: the parity tests feed the permutations the reference drew) and returned sorted."""
    randperm = staticmethod(torch.randperm)

    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        self.num, self.pos_fraction, self.neg_pos_ub, self.add_gt_as_proposals = num, pos_fraction, neg_pos_ub, add_gt_as_proposals

    def random_choice(self, gallery, num):
        perm = self.randperm(gallery.numel())[:num].to(device=gallery.device)
        return gallery[perm]

    def sample(self, gt_inds, bboxes, gt_bboxes, gt_labels=None):
        """gt_inds: long [N] (-1 ignore, 0 negative, i + 1 = box i) for `bboxes` [N, >= 4] -> the fields of SamplingResult."""
        bboxes = bboxes[:, :4]
        gt_flags = bboxes.new_zeros((bboxes.shape[0],), dtype=torch.uint8)
        G = gt_bboxes.shape[0]
        if self.add_gt_as_proposals and G > 0:
            assert gt_labels is not None, 'gt_labels must be given when add_gt_as_proposals is True'
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
            gt_inds = torch.cat([torch.arange(1, G + 1, dtype=gt_inds.dtype, device=gt_inds.device), gt_inds])   # AssignResult.add_gt_
            gt_flags = torch.cat([bboxes.new_ones(G, dtype=torch.uint8), gt_flags])
        num_pos = int(self.num * self.pos_fraction)
        pos = torch.nonzero(gt_inds > 0, as_tuple=False).reshape(-1)
        if pos.numel() > num_pos:
            pos = self.random_choice(pos, num_pos)
        pos = pos.unique()
        num_neg = self.num - pos.numel()
        if self.neg_pos_ub >= 0:
            num_neg = min(num_neg, int(self.neg_pos_ub * max(1, pos.numel())))
        neg = torch.nonzero(gt_inds == 0, as_tuple=False).reshape(-1)
        if neg.numel() > num_neg:
            neg = self.random_choice(neg, num_neg)
        neg = neg.unique()
        pa = gt_inds[pos] - 1
        r = types.SimpleNamespace(pos_inds=pos, neg_inds=neg, pos_bboxes=bboxes[pos], neg_bboxes=bboxes[neg], pos_is_gt=gt_flags[pos],
                                  num_gts=G, pos_assigned_gt_inds=pa,
                                  pos_gt_bboxes=gt_bboxes[pa] if G else gt_bboxes.reshape(-1, 4),
                                  pos_gt_labels=gt_labels[pa] if (gt_labels is not None and G) else None)
        r.bboxes = torch.cat([r.pos_bboxes, r.neg_bboxes])
        return r


def delta2bbox_stds(rois, deltas, stds, max_shape=None):
    """DeltaXYWHBBoxCoder.decode with target_means 0 and `stds` through pt_delta2bbox (which takes unit stds)."""
    return F.delta2bbox(rois.contiguous(), (deltas * deltas.new_tensor(stds)).contiguous(), max_shape=max_shape)


@HEADS.register_module()
class RPNHead(nn.Module):
    """rpn_head.py: 3x3 conv + ReLU, then one objectness logit and four deltas per anchor."""

    def __init__(self, in_channels, feat_channels=256,
                 anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32, 64]),
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]),
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_bbox=dict(type='L1Loss', loss_weight=1.0), train_cfg=None, test_cfg=None, init_cfg=None, **kwargs):
        super().__init__()
        assert bbox_coder.get('type') == 'DeltaXYWHBBoxCoder' and loss_cls.get('use_sigmoid', False)
        ag = dict(anchor_generator)
        assert ag.pop('type') == 'AnchorGenerator'
        self.anchor_generator = AnchorGenerator(**ag)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self.num_classes = self.cls_out_channels = 1
        self.means, self.stds = tuple(bbox_coder.get('target_means', (0.,) * 4)), tuple(bbox_coder.get('target_stds', (1.,) * 4))
        assert self.means == (0., 0., 0., 0.)
        self.loss_cls, self.loss_bbox = build_loss(loss_cls), build_loss(loss_bbox)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if train_cfg:
            self.assigner = build_assigner(train_cfg['assigner'])
            self.sampler = build_from_cfg(train_cfg['sampler'], BBOX_SAMPLERS)
            assert train_cfg.get('allowed_border', -1) < 0, 'allowed_border >= 0 is not used by the AI-TOD configs'
            self.pos_weight = train_cfg.get('pos_weight', -1)
        self.rpn_conv = nn.Conv2d(in_channels, feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(feat_channels, self.num_anchors * self.cls_out_channels, 1)
        self.rpn_reg = nn.Conv2d(feat_channels, self.num_anchors * 4, 1)
        for m in (self.rpn_conv, self.rpn_cls, self.rpn_reg):                       # init_cfg: Normal(std 0.01)
            nn.init.normal_(m.weight, 0, 0.01)
            nn.init.constant_(m.bias, 0)

    def forward_single(self, x):
        x = TF.relu(self.rpn_conv(x), inplace=True)
        return self.rpn_cls(x), self.rpn_reg(x)

    def forward(self, feats):
        outs = [self.forward_single(x) for x in feats]
        return [o[0] for o in outs], [o[1] for o in outs]

    # --------------------------------------------------------------------------------------- training --
    def get_targets(self, anchors, gt_bboxes):
        """anchor_head.py:181-380 with sampling -> labels [B, A] (0 = object, 1 = background), label_weights [B, A],
        bbox_targets [B, A, 4], bbox_weights [B, A, 4], num_total_pos, num_total_neg (host ints: the sampler synchronises)."""
        B, A = len(gt_bboxes), anchors.shape[0]
        inds, _, _, _ = self.assigner.assign_batch(anchors, gt_bboxes)
        labels = torch.full((B, A), self.num_classes, dtype=torch.long, device=anchors.device)
        lw = anchors.new_zeros((B, A))
        bt, bw = anchors.new_zeros((B, A, 4)), anchors.new_zeros((B, A, 4))
        npos = nneg = 0
        pw = 1.0 if self.pos_weight <= 0 else float(self.pos_weight)
        for b in range(B):
            r = self.sampler.sample(inds[b].long(), anchors, gt_bboxes[b].float())
            if r.pos_inds.numel():
                bt[b, r.pos_inds] = bbox2delta(r.pos_bboxes, r.pos_gt_bboxes, self.means, self.stds)
                bw[b, r.pos_inds] = 1.0
                labels[b, r.pos_inds] = 0
                lw[b, r.pos_inds] = pw
            lw[b, r.neg_inds] = 1.0
            npos, nneg = npos + r.pos_inds.numel(), nneg + r.neg_inds.numel()
        return labels, lw, bt, bw, npos, nneg

    def loss(self, cls_scores, bbox_preds, gt_bboxes, img_metas, gt_bboxes_ignore=None):
        """rpn_head.py:66-98 -> dict(loss_rpn_cls=[per level], loss_rpn_bbox=[per level]), normalised by the sampled count."""
        sizes = [f.size()[-2:] for f in cls_scores]
        lvl_anchors = self.anchor_generator.grid_anchors(sizes, cls_scores[0].device)
        labels, lw, bt, bw, npos, nneg = self.get_targets(torch.cat(lvl_anchors), gt_bboxes)
        avg = float(npos + nneg)
        lc, lb, o = [], [], 0
        for cs, bp, la in zip(cls_scores, bbox_preds, lvl_anchors):
            n = la.shape[0]
            cs = cs.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
            bp = bp.permute(0, 2, 3, 1).reshape(-1, 4)
            lc.append(self.loss_cls(cs, labels[:, o:o + n].reshape(-1), lw[:, o:o + n].reshape(-1), avg_factor=avg))
            lb.append(self.loss_bbox(bp, bt[:, o:o + n].reshape(-1, 4), bw[:, o:o + n].reshape(-1, 4), avg_factor=avg))
            o += n
        return dict(loss_rpn_cls=lc, loss_rpn_bbox=lb)

    # -------------------------------------------------------------------------------------- proposals --
    @torch.no_grad()
    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg=None, rescale=False):
        """rpn_head.py:100-249 -> one [n, 5] tensor (x1, y1, x2, y2, score) per image."""
        cfg = self.test_cfg if cfg is None else cfg
        sizes = [f.size()[-2:] for f in cls_scores]
        lvl_anchors = self.anchor_generator.grid_anchors(sizes, cls_scores[0].device)
        nms_pre, out = cfg.get('nms_pre', -1), []
        for b, meta in enumerate(img_metas):
            S, D, AN, ID = [], [], [], []
            for i, (cs, bp, a) in enumerate(zip(cls_scores, bbox_preds, lvl_anchors)):
                s = cs[b].permute(1, 2, 0).reshape(-1).sigmoid()
                d = bp[b].permute(1, 2, 0).reshape(-1, 4)
                if 0 < nms_pre < s.shape[0]:
                    rs, ri = s.sort(descending=True)
                    s, d, a = rs[:nms_pre], d[ri[:nms_pre]], a[ri[:nms_pre]]
                S.append(s); D.append(d); AN.append(a)
                ID.append(s.new_full((s.shape[0],), i, dtype=torch.long))
            s, d, a, ids = torch.cat(S), torch.cat(D), torch.cat(AN), torch.cat(ID)
            p = delta2bbox_stds(a, d, self.stds, max_shape=meta['img_shape'])
            mb = cfg.get('min_bbox_size', 0)
            if mb > 0:
                ok = ((p[:, 2] - p[:, 0]) >= mb) & ((p[:, 3] - p[:, 1]) >= mb)
                p, s, ids = p[ok], s[ok], ids[ok]
            if p.numel() == 0:
                out.append(p.new_zeros(0, 5))
                continue
            dets, _ = F.batched_nms(p, s, ids, cfg['nms'])
            out.append(dets[:cfg['max_per_img']])
        return out

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None):
        """base_dense_head.py forward_train: (losses, proposal_list)."""
        outs = self(x)
        losses = self.loss(*outs, gt_bboxes, img_metas, gt_bboxes_ignore)
        return losses, self.get_bboxes(*outs, img_metas, cfg=proposal_cfg)

    def simple_test_rpn(self, x, img_metas):
        return self.get_bboxes(*self(x), img_metas)


@HEADS.register_module()
class Shared2FCBBoxHead(nn.Module):
    """convfc_bbox_head.py Shared2FCBBoxHead over bbox_head.py: two shared FCs, class logits (+ background) and class-specific
    deltas."""

    def __init__(self, in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=80,
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2]),
                 reg_class_agnostic=False, reg_decoded_bbox=False,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                 loss_bbox=dict(type='L1Loss', loss_weight=1.0), init_cfg=None, **kwargs):
        super().__init__()
        assert not reg_decoded_bbox and not loss_cls.get('use_sigmoid', False) and bbox_coder.get('type') == 'DeltaXYWHBBoxCoder'
        self.num_classes, self.reg_class_agnostic = num_classes, reg_class_agnostic
        self.means, self.stds = tuple(bbox_coder.get('target_means', (0.,) * 4)), tuple(bbox_coder.get('target_stds', (.1, .1, .2, .2)))
        assert self.means == (0., 0., 0., 0.)
        self.loss_cls, self.loss_bbox = build_loss(loss_cls), build_loss(loss_bbox)
        d = in_channels * roi_feat_size * roi_feat_size
        self.shared_fcs = nn.ModuleList([nn.Linear(d, fc_out_channels), nn.Linear(fc_out_channels, fc_out_channels)])
        self.fc_cls = nn.Linear(fc_out_channels, num_classes + 1)
        self.fc_reg = nn.Linear(fc_out_channels, 4 if reg_class_agnostic else 4 * num_classes)
        nn.init.normal_(self.fc_cls.weight, 0, 0.01); nn.init.constant_(self.fc_cls.bias, 0)          # bbox_head.py init_cfg
        nn.init.normal_(self.fc_reg.weight, 0, 0.001); nn.init.constant_(self.fc_reg.bias, 0)
        for fc in self.shared_fcs:                                                                     # convfc_bbox_head.py: Xavier
            nn.init.xavier_uniform_(fc.weight); nn.init.constant_(fc.bias, 0)

    def forward(self, x):
        x = x.flatten(1)
        for fc in self.shared_fcs:
            x = TF.relu(fc(x), inplace=True)
        return self.fc_cls(x), self.fc_reg(x)

    def get_targets(self, sampling_results, rcnn_train_cfg=None):
        """bbox_head.py:122-255 (concat=True): rows ordered [positives | negatives] per image."""
        pw = 1.0 if (rcnn_train_cfg or {}).get('pos_weight', -1) <= 0 else float(rcnn_train_cfg['pos_weight'])
        L, LW, BT, BW = [], [], [], []
        for r in sampling_results:
            npos, nneg = r.pos_bboxes.shape[0], r.neg_bboxes.shape[0]
            lab = r.pos_bboxes.new_full((npos + nneg,), self.num_classes, dtype=torch.long)
            lw, bt, bw = r.pos_bboxes.new_zeros(npos + nneg), r.pos_bboxes.new_zeros(npos + nneg, 4), r.pos_bboxes.new_zeros(npos + nneg, 4)
            if npos:
                lab[:npos] = r.pos_gt_labels
                lw[:npos] = pw
                bt[:npos] = bbox2delta(r.pos_bboxes, r.pos_gt_bboxes, self.means, self.stds)
                bw[:npos] = 1.0
            if nneg:
                lw[-nneg:] = 1.0
            L.append(lab); LW.append(lw); BT.append(bt); BW.append(bw)
        return torch.cat(L), torch.cat(LW), torch.cat(BT), torch.cat(BW)

    def loss(self, cls_score, bbox_pred, rois, labels, label_weights, bbox_targets, bbox_weights):
        """bbox_head.py:257-314"""
        losses = {}
        avg = (label_weights > 0).sum().float().clamp(min=1.0)
        if cls_score.numel():
            losses['loss_cls'] = self.loss_cls(cls_score, labels, label_weights, avg_factor=avg)
            losses['acc'] = (cls_score.argmax(1) == labels).float().mean() * 100.0
        pos = (labels >= 0) & (labels < self.num_classes)
        if bool(pos.any()):
            if self.reg_class_agnostic:
                pp = bbox_pred.view(bbox_pred.size(0), 4)[pos]
            else:
                pp = bbox_pred.view(bbox_pred.size(0), -1, 4)[pos, labels[pos]]
            losses['loss_bbox'] = self.loss_bbox(pp, bbox_targets[pos], bbox_weights[pos], avg_factor=bbox_targets.size(0))
        else:
            losses['loss_bbox'] = bbox_pred[pos].sum()
        return losses

    @torch.no_grad()
    def get_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False, cfg=None):
        """bbox_head.py:316-379: softmax scores, per-class decode with clipping, rescale, multiclass NMS."""
        scores = TF.softmax(cls_score, dim=-1)
        n = rois.shape[0]
        if self.reg_class_agnostic:
            boxes = delta2bbox_stds(rois[:, 1:], bbox_pred, self.stds, max_shape=img_shape)
        else:
            r = rois[:, None, 1:].expand(n, self.num_classes, 4).reshape(-1, 4)
            boxes = delta2bbox_stds(r, bbox_pred.reshape(-1, 4), self.stds, max_shape=img_shape).view(n, -1)
        if rescale and n:
            sf = boxes.new_tensor(np.asarray(scale_factor, np.float32))
            boxes = (boxes.view(n, -1, 4) / sf).view(n, -1)
        if cfg is None:
            return boxes, scores
        return multiclass_nms(boxes, scores, cfg['score_thr'], cfg['nms'], cfg['max_per_img'])


@HEADS.register_module()
class StandardRoIHead(nn.Module):
    """standard_roi_head.py (bbox branch; the AI-TOD config has no mask head)."""

    def __init__(self, bbox_roi_extractor=None, bbox_head=None, train_cfg=None, test_cfg=None, init_cfg=None, **kwargs):
        super().__init__()
        self.bbox_roi_extractor = build_roi_extractor(bbox_roi_extractor)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if train_cfg:
            self.bbox_assigner = build_assigner(train_cfg['assigner'])
            self.bbox_sampler = build_from_cfg(train_cfg['sampler'], BBOX_SAMPLERS)

    def _bbox_forward(self, x, rois):
        feats = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois)
        return self.bbox_head(feats)

    def sample(self, proposal_list, gt_bboxes, gt_labels):
        """standard_roi_head.py:70-88: assign + sample per image."""
        out = []
        for p, gb, gl in zip(proposal_list, gt_bboxes, gt_labels):
            inds, _ = self.bbox_assigner.assign(p[:, :4].contiguous(), gb.float())
            out.append(self.bbox_sampler.sample(inds, p, gb.float(), gl))
        return out

    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore=None, **kwargs):
        results = self.sample(proposal_list, gt_bboxes, gt_labels)
        rois = bbox2roi([r.bboxes for r in results])
        cls_score, bbox_pred = self._bbox_forward(x, rois)
        targets = self.bbox_head.get_targets(results, self.train_cfg)
        return self.bbox_head.loss(cls_score, bbox_pred, rois, *targets)

    @torch.no_grad()
    def simple_test(self, x, proposal_list, img_metas, rescale=False):
        """test_mixins.py simple_test_bboxes -> [(dets [n, 5], labels [n])] per image."""
        rois = bbox2roi([p[:, :4] for p in proposal_list])
        if rois.shape[0] == 0:
            z = x[0].new_zeros((0, 5))
            return [(z, z.new_zeros((0,), dtype=torch.long)) for _ in proposal_list]
        cls_score, bbox_pred = self._bbox_forward(x, rois)
        counts = [p.shape[0] for p in proposal_list]
        out = []
        for r, cs, bp, meta in zip(rois.split(counts), cls_score.split(counts), bbox_pred.split(counts), img_metas):
            if r.shape[0] == 0:
                out.append((r.new_zeros((0, 5)), r.new_zeros((0,), dtype=torch.long)))
                continue
            out.append(self.bbox_head.get_bboxes(r, cs, bp, meta['img_shape'], meta.get('scale_factor', np.ones(4, np.float32)),
                                                 rescale=rescale, cfg=self.test_cfg))
        return out


@DETECTORS.register_module()
class FasterRCNN(Student_FCOS):
    """detectors/faster_rcnn.py over two_stage.py:96-190."""

    def __init__(self, backbone, neck=None, rpn_head=None, roi_head=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        nn.Module.__init__(self)
        self.fp16_enabled = False                        # (BaseDetector.__init__)
        from .registry import build_backbone, build_neck
        if pretrained is not None and isinstance(backbone, dict) and 'init_cfg' not in backbone:
            backbone = dict(backbone, init_cfg=dict(type='Pretrained', checkpoint=pretrained))
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck) if neck is not None else None
        rpn = dict(rpn_head)
        rpn.update(train_cfg=(train_cfg or {}).get('rpn'), test_cfg=(test_cfg or {}).get('rpn'))
        self.rpn_head = build_head(rpn)
        roi = dict(roi_head)
        roi.update(train_cfg=(train_cfg or {}).get('rcnn'), test_cfg=(test_cfg or {}).get('rcnn'))
        self.roi_head = build_head(roi)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    with_neck_agg = False

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, proposals=None, **kwargs):
        x = self.extract_feat(img.to(torch.float))
        losses = {}
        proposal_cfg = (self.train_cfg or {}).get('rpn_proposal', (self.test_cfg or {}).get('rpn'))
        rpn_losses, proposal_list = self.rpn_head.forward_train(x, img_metas, gt_bboxes, gt_bboxes_ignore=gt_bboxes_ignore,
                                                                proposal_cfg=proposal_cfg)
        losses.update(rpn_losses)
        losses.update(self.roi_head.forward_train(x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore))
        return losses

    def simple_test(self, img, img_metas, proposals=None, rescale=False):
        x = self.extract_feat(img.to(torch.float))
        proposal_list = self.rpn_head.simple_test_rpn(x, img_metas) if proposals is None else proposals
        res = self.roi_head.simple_test(x, proposal_list, img_metas, rescale=rescale)
        return [bbox2result(b, l, self.roi_head.bbox_head.num_classes) for b, l in res]
