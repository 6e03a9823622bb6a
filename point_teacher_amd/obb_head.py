"""TS_P2RBRotatedFCOSHead - the oriented twin of TS_P2BFCOSHead (config 5,
`OBB_TOD/configs/point teacher/sodaa_fcos_pointteacher_1x.py`), mirroring
/root/reference/OBB_TOD/mmrotate/models/dense_heads/rotated_fcos_head_p2rb_ts.py (+ its parent
rotated_anchor_free_head.py) method for method.  Citations are relative to that file.

What changes against the HBB head: an angle branch (`conv_angle` + `scale_angle`), boxes are
(cx,cy,w,h,a) decoded by DistanceAnglePointCoder, targets live in the frame of the assigned
oriented box, the regression loss is RotatedIoULoss (differentiable rotated IoU kernel), bags are
pooled with RoIAlignRotated and the bag loss is weighted 0.25 / 0.75.  As in the HBB head, the
whole batch goes through ONE launch per op and nothing synchronises the host.
"""
import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import functional as F
from .core import bbox_cxcywh_to_xyxy, bbox_xyxy_to_cxcywh, mean0, multi_apply, reduce_mean_many
from .head import TS_P2BFCOSHead, _cat
from .nn_modules import Scale
from .obb import multiclass_nms_rotated, rbbox2roi, rbbox_overlaps
from .obb_proposals import MIL_gen_proposals_from_cfg
from .registry import HEADS, build_bbox_coder

INF = 1e8


@HEADS.register_module()
class TS_P2RBRotatedFCOSHead(TS_P2BFCOSHead):
    """Constructor keywords as :96-216 (RotatedAnchorFreeHead defaults for the rest)."""

    def __init__(self, num_classes, in_channels, regress_ranges=((-1, INF),), center_sampling=False,
                 center_sample_radius=1.5, norm_on_bbox=False, centerness_on_reg=False, separate_angle=False,
                 scale_angle=True, h_bbox_coder=dict(type='DistancePointBBoxCoder'),
                 bbox_coder=dict(type='DistanceAnglePointCoder', angle_version='le90'), mil_stack_conv=1, beta=0.25,
                 top_k=3, num_stages=2, angle_version='le90',
                 bbox_roi_extractor=dict(type='RotatedSingleRoIExtractor',
                                         roi_layer=dict(type='RoIAlignRotated', out_size=7, sample_num=2, clockwise=True),
                                         out_channels=256, featmap_strides=[8]),
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_angle=dict(type='L1Loss', loss_weight=1.0),
                 loss_bbox_burn1=dict(type='RotatedIoULoss', loss_weight=1.0),
                 loss_bbox_burn2=dict(type='DN_IoULoss', loss_weight=1.0, hyper=0.1),
                 loss_bbox_denosing=dict(type='DN_DIoULoss', loss_weight=1.0, hyper=0.3),
                 loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True), **kwargs):
        assert not separate_angle, 'separate_angle=False in the Point-Teacher OBB config'
        self.separate_angle, self.is_scale_angle = separate_angle, scale_angle
        self.angle_version = angle_version
        self.h_bbox_coder_cfg, self.loss_angle_cfg = h_bbox_coder, loss_angle     # built but never called (:189-190)
        super().__init__(num_classes, in_channels, center_sampling=center_sampling,
                         center_sample_radius=center_sample_radius, norm_on_bbox=norm_on_bbox,
                         centerness_on_reg=centerness_on_reg, mil_stack_conv=mil_stack_conv, beta=beta, top_k=top_k,
                         num_stages=num_stages, bbox_roi_extractor=bbox_roi_extractor, loss_cls=loss_cls,
                         loss_bbox_burn1=loss_bbox_burn1, loss_bbox_burn2=loss_bbox_burn2,
                         loss_bbox_denosing=loss_bbox_denosing, loss_centerness=loss_centerness, norm_cfg=norm_cfg,
                         regress_ranges=regress_ranges, **kwargs)
        self.bbox_coder = build_bbox_coder(bbox_coder)

    # ------------------------------------------------------------------ layers --
    def _init_layers(self):
        """:217-241: the HBB layers + the angle branch."""
        super()._init_layers()
        del self.fc_iou            # the oriented head builds no IoU branch (:243-283) - the HBB head does (fcos_head_p2b_ts.py:181) - so its
        #                            checkpoints carry no `fc_iou.*` keys (found by building the reference's real head, oracle/gen_golden_obb_iter.py)
        self.conv_angle = nn.Conv2d(self.feat_channels, 1, 3, padding=1)
        if self.is_scale_angle:
            self.scale_angle = Scale(1.0)

    # ----------------------------------------------------------------- forward --
    def forward_single(self, x, scale, stride, points, branches='both'):
        """:356-391 -> (cls_score, bbox_pred, angle_pred, centerness, points); `branches`: see TS_P2BFCOSHead.forward."""
        cls_feat, reg_feat = self._towers(x, branches)
        cls_score = self.conv_cls(cls_feat) if branches != 'reg' else None
        if reg_feat is None:
            return cls_score, None, None, None, points
        bbox_pred = self.conv_reg(reg_feat)
        centerness = self.conv_centerness(reg_feat if self.centerness_on_reg else cls_feat)
        bbox_pred = scale(bbox_pred).float()
        if self.norm_on_bbox:
            bbox_pred = bbox_pred.clamp(min=0) * stride
        else:
            bbox_pred = bbox_pred.exp()
        angle_pred = self.conv_angle(reg_feat)
        if self.is_scale_angle:
            angle_pred = self.scale_angle(angle_pred).float()
        return cls_score, bbox_pred, angle_pred, centerness, points

    def _flat5(self, cls_scores, bbox_preds, angle_preds, centernesses):
        """concat_per_img (:1140-1161): [B,P,C], [B,P,5] = (l,t,r,b,angle), [B,P]; None for a branch that was not evaluated."""
        B = (cls_scores if cls_scores[0] is not None else bbox_preds)[0].size(0)
        fc = fr = ft = None
        if cls_scores[0] is not None:
            fc = torch.cat([c.permute(0, 2, 3, 1).reshape(B, -1, self.cls_out_channels) for c in cls_scores], 1).float().contiguous()
        if bbox_preds[0] is not None:
            fr = torch.cat([torch.cat([r.permute(0, 2, 3, 1).reshape(B, -1, 4), a.permute(0, 2, 3, 1).reshape(B, -1, 1).float()], -1)
                            for r, a in zip(bbox_preds, angle_preds)], 1).float().contiguous()
            ft = torch.cat([t.permute(0, 2, 3, 1).reshape(B, -1) for t in centernesses], 1).float().contiguous()
        return fc, fr, ft

    # -------------------------------------------------------------- pseudo boxes --
    def get_pseudo_bbox(self, cls_scores, bbox_preds, angle_preds, centernesses, all_level_points, gt_points, gt_labels,
                        gt_bboxes, filter_scores, img_metas, img_list, gt_bboxes_ignore=None):
        """:392-411 + gnerate_pseudo(_single) :844-929."""
        assert len(cls_scores) == len(bbox_preds) == len(angle_preds) == len(centernesses)
        fc, fr, _ = self._flat5(cls_scores, bbox_preds, angle_preds, centernesses)
        B, P = fc.shape[:2]
        points = torch.cat(all_level_points, 0)
        counts = [int(l.shape[0]) for l in gt_labels]
        off, _ = F.make_offsets(counts, fc.device)
        a = self.fuse_assigner
        gp, gl, gb = torch.cat(gt_points), torch.cat(gt_labels), torch.cat(gt_bboxes)
        dec = self.bbox_coder.decode(points.repeat(B, 1), fr.reshape(-1, 5))
        cls = fc.reshape(-1, self.cls_out_channels)
        gi, cand = F.fuse_assign_obb(points, dec, cls, gp, gl, off, B, a.num_pre, a.topk, a.cls_cost.weight,
                                     a.reg_cost.weight, a.location_cost.weight)
        ps = F.pseudo_boxes_obb(P, dec, cls, gp, gl, off, B, gi, cand)
        nz = ps['nassigned'] > 0
        valid = nz & (ps['scores'] >= filter_scores)
        iou = torch.where(nz, rbbox_overlaps(ps['bboxes'], gb, is_aligned=True), torch.zeros_like(ps['scores']))
        ious, nzs = torch.split(iou, counts), torch.split(nz, counts)
        mean_iou = sum((i.sum() / z.sum()) if c else i.new_zeros(()) for i, z, c in zip(ious, nzs, counts)) / B
        return (list(torch.split(ps['bboxes'], counts)), list(torch.split(ps['points'], counts)), list(gt_labels),
                mean_iou, list(torch.split(valid, counts)))

    # --------------------------------------------------------------------- losses --
    def _reg_branch_obb(self, points, B, fr, ft, gi_reg, boxes5, box_labels, off, cls_pos=None):
        """Shared tail of loss_pseudo (:483-512) and loss (:583-624): oriented targets, centerness-weighted
        RotatedIoULoss and centerness BCE, dense over all B*P points with the positive mask.  `cls_pos`: see
        TS_P2BFCOSHead._reg_branch (one coalesced all-reduce for the normalisers of a loss call)."""
        labels_reg, tg, ang, ctr_t = F.fcos_targets_obb(points, gi_reg, boxes5, box_labels, off, B, self.num_classes)
        pos = labels_reg < self.num_classes
        norms = reduce_mean_many(pos.sum(), ctr_t.sum().detach(), *([cls_pos] if cls_pos is not None else []))
        num_pos, ctr_den = norms[0].clamp(min=1.0), norms[1].clamp(min=1e-6)
        if fr is None:                                 # regression branch not evaluated (burn-in step 1, augmented pass)
            return None, None, norms[2].clamp(min=1.0)
        pts = points.repeat(B, 1)
        pred = self.bbox_coder.decode(pts, fr.reshape(-1, 5))
        tgt = self.bbox_coder.decode(pts, torch.cat([tg, ang], dim=-1))
        # rows outside `pos` do not exist in the reference: give them a harmless unit box on both sides
        dummy = pred.new_zeros(5)                 # (0, 0, 1, 1, 0) built on the device: new_tensor(list) is a blocking upload
        dummy[2:4] = 1.0
        pred = torch.where(pos[:, None], pred, dummy)
        tgt = torch.where(pos[:, None], tgt, dummy)
        elem = self.loss_bbox._elem(pred, tgt)
        loss_bbox = self.loss_bbox.loss_weight * torch.where(pos, elem * ctr_t, torch.zeros_like(elem)).sum() / ctr_den
        bce = TF.binary_cross_entropy_with_logits(ft.reshape(-1), ctr_t, reduction='none')
        loss_ctr = self.loss_centerness.loss_weight * torch.where(pos, bce, torch.zeros_like(bce)).sum() / num_pos
        if cls_pos is not None:
            return loss_bbox, loss_ctr, norms[2].clamp(min=1.0)
        return loss_bbox, loss_ctr

    def loss_pseudo(self, cls_scores, bbox_preds, angle_preds, centernesses, all_level_points, gt_points, gt_labels,
                    pseudo_points, pseudo_labels, pseudo_bboxes, gt_augument_ignore, img_metas, img_list, burn_in_step1,
                    gt_bboxes_ignore=None, gt_valid=None, pseudo_valid=None):
        """:417-529"""
        assert len(cls_scores) == len(bbox_preds) == len(angle_preds) == len(centernesses)
        fc, fr, ft = self._flat5(cls_scores, bbox_preds, angle_preds, centernesses)
        B = fc.shape[0]
        points = torch.cat(all_level_points, 0)
        dev = fc.device
        gcounts = [int(p.shape[0]) for p in gt_points]
        goff, _ = F.make_offsets(gcounts, dev)
        gi_cls = F.topk_assign(points, torch.cat(gt_points), goff, B, self.assigner.num_pre,
                               gt_valid=_cat(gt_valid) if gt_valid is not None else None)
        labels, _, _ = F.fcos_targets(points, gi_cls, None, torch.cat(gt_labels), goff, B, self.num_classes)
        pcounts = [int(b.shape[0]) for b in pseudo_bboxes]
        poff, _ = F.make_offsets(pcounts, dev)
        pb = torch.cat(pseudo_bboxes)
        gi_reg = F.topk_assign(points, pb, poff, B, self.pseudo_assigner.num_pre,
                               gt_valid=_cat(pseudo_valid) if pseudo_valid is not None else None)
        loss_bbox, loss_ctr, num_pos = self._reg_branch_obb(points, B, fr, ft, gi_reg, pb, torch.cat(pseudo_labels), poff,
                                                            cls_pos=(labels < self.num_classes).sum())
        loss_cls = self.loss_cls(fc.reshape(-1, self.cls_out_channels), labels, weight=None, avg_factor=num_pos)
        return loss_cls, loss_bbox, loss_ctr

    def loss(self, cls_scores, bbox_preds, angle_preds, centernesses, all_level_points, gt_bboxes, img_metas,
             gt_bboxes_ignore=None, gt_valid=None):
        """:535-626 (burn-in step 1: synthetic oriented rectangles, syn_assigner)."""
        _, fr, ft = self._flat5(cls_scores, bbox_preds, angle_preds, centernesses)
        B = fr.shape[0]
        points = torch.cat(all_level_points, 0)
        counts = [int(b.shape[0]) for b in gt_bboxes]
        off, _ = F.make_offsets(counts, fr.device)
        gb = torch.cat(gt_bboxes)
        gi = F.topk_assign(points, gb, off, B, self.syn_assigner.num_pre,
                           gt_valid=_cat(gt_valid) if gt_valid is not None else None)
        return self._reg_branch_obb(points, B, fr, ft, gi, gb, None, off)

    # ----------------------------------------------------------------- MIL head --
    def mil_bag_extensive(self, num_gt, num_gt_pre_image, x, img_metas, proposals_list, proposals_valid_list,
                          proposals_reference_list, proposals_real_list, bbox_results, fine_proposal_cfg, stage,
                          bag_weight=None):
        """:1285-1343.  The refinement regresses the axis-aligned (cx,cy,w,h) part of each bag with the
        HBB delta coder and DN-DIoU; the angle of the bag is carried through unchanged."""
        U1 = max(sum(p.shape[0] for p in proposals_list) // max(num_gt, 1), 1)
        bbox_results['base_shaking_num'] = U1
        points_list = [p[:, :2] for p in proposals_list]
        ext, ext_valid, _, ext_ref = MIL_gen_proposals_from_cfg(points_list, proposals_list, fine_proposal_cfg,
                                                               proposals_reference_list, img_metas)
        num_aug = len(fine_proposal_cfg['base_ratios']) ** 2 * (1 + 4 * len(fine_proposal_cfg['shake_ratio'] or []))
        ext_real = [r.unsqueeze(1).repeat(1, num_aug, 1).reshape(-1, 5) for r in proposals_real_list]
        bbox_results['base_bags'], bbox_results['base_bags_valid'] = proposals_list, proposals_valid_list
        bags, real, ref = torch.cat(ext), torch.cat(ext_real), torch.cat(ext_ref)
        valid = torch.cat(ext_valid).reshape(-1)
        K = bags.shape[0]
        ci = rbbox_overlaps(bags, real, is_aligned=True)
        if bag_weight is None:
            bbox_results['coarse_bags_iou'] = mean0(ci)
            wgt, avg, wsum = valid.float(), float(max(K, 1)), None
        else:
            bw = torch.cat(bag_weight)
            wsum = bw.sum().clamp(min=1.0)
            bbox_results['coarse_bags_iou'] = (ci * bw).sum() / wsum
            wgt, avg = valid.float() * bw, wsum
        bbox_results['extensive_shaking_num'] = num_aug
        feats = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rbbox2roi(ext)).flatten(1)
        bbox_pred = self.fc_reg[stage](self._fc_stack(self.shared_fcs_reg[stage], feats))
        del feats
        bbox_pred = self.mil_bbox_decoder.decode(bbox_cxcywh_to_xyxy(bags[:, :4]), bbox_pred,
                                                 max_shape=img_metas[0]['img_shape'])
        pred_d = bbox_pred.detach()
        target = bbox_cxcywh_to_xyxy(ref[:, :4])
        if bag_weight is None:
            bbox_results['loss_mil_bbox'] = self.loss_bbox_denosing(bbox_pred, target, weight=wgt, avg_factor=avg)
        else:
            bbox_results['loss_mil_bbox'] = self.loss_bbox_denosing.forward_masked(
                bbox_pred, target, torch.cat(bag_weight) > 0, valid.float(), avg)
        refined = torch.cat([bbox_xyxy_to_cxcywh(pred_d), bags[:, 4:5]], dim=1)
        ri = rbbox_overlaps(refined, real, is_aligned=True)
        bbox_results['refine_bags_iou'] = mean0(ri) if wsum is None else (ri * torch.cat(bag_weight)).sum() / wsum
        sizes = [e.shape[0] for e in ext]
        bbox_results['extensive_bags'] = list(torch.split(refined, sizes))
        bbox_results['extensive_bags_valid'] = ext_valid
        bbox_results['extensive_bags_reference'] = ext_ref
        bbox_results['extensive_bags_real'] = ext_real

    def mil_bag_classifier(self, num_gt, x, bbox_results, stage):
        """:1347-1363"""
        rois = rbbox2roi(bbox_results['extensive_bags'])
        U1, U2 = bbox_results['base_shaking_num'], bbox_results['extensive_shaking_num']
        feats = self._fc_stack(self.shared_fcs_bag[stage],
                               self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois).flatten(1))
        bbox_results['cls_score'] = self.fc_cls[stage](feats).view(num_gt, U1, U2, self.num_classes)
        bbox_results['ins_score'] = self.fc_ins[stage](feats).view(num_gt, U1, U2, self.num_classes)

    def forward_mil_head(self, num_gt, num_gt_pre_image, x, proposals_list, proposals_valid_list,
                         proposals_reference_list, proposals_real_list, img_metas, fine_proposal_cfg, stage,
                         neg_proposal_list=None, neg_weight_list=None, bag_weight=None, need_classifier=True):
        """:1366-1384"""
        bbox_results = {}
        x = [f.float().contiguous(memory_format=torch.channels_last) for f in x[:self.bbox_roi_extractor.num_inputs]]
        self.mil_bag_extensive(num_gt, num_gt_pre_image, x, img_metas, proposals_list, proposals_valid_list,
                               proposals_reference_list, proposals_real_list, bbox_results, fine_proposal_cfg, stage,
                               bag_weight=bag_weight)
        if need_classifier:
            self.mil_bag_classifier(num_gt, x, bbox_results, stage)
        if neg_proposal_list is not None:
            feats = self._fc_stack(self.shared_fcs_bag[stage],
                                   self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs],
                                                           rbbox2roi(neg_proposal_list)).flatten(1))
            bbox_results['neg_cls_score'] = self.fc_cls[stage](feats)
        return bbox_results

    def mil_bag_training(self, bbox_results, gt_labels, neg_weight_list):
        """:1252-1283: as the HBB head with the positive bags weighted 0.25 and the negatives 0.75."""
        cls, ins = bbox_results['cls_score'], bbox_results['ins_score']
        N, U1, U2, C = cls.shape
        labels = torch.cat(gt_labels).unsqueeze(1).repeat(1, U1).reshape(-1)
        valid = torch.cat(bbox_results['extensive_bags_valid'], 0).reshape(-1)
        total, nvalid = F.mil_bag_loss_sum(cls.reshape(N * U1, U2, C), ins.reshape(N * U1, U2, C), valid, labels)
        num_sample = nvalid.clamp(min=1.0)
        loss = 0.25 * total / num_sample
        if neg_weight_list is not None:
            loss = loss + 0.75 * F.mil_neg_loss_sum(bbox_results['neg_cls_score'], torch.cat(neg_weight_list)) / num_sample
        return loss

    # ------------------------------------------------------------------ inference --
    def get_bboxes(self, cls_scores, bbox_preds, angle_preds, centernesses, all_level_points, img_metas, cfg=None,
                   rescale=None, with_nms=True):
        """:931-1083: per image, top `nms_pre` by the maximum class score (centerness is NOT applied:
        score_factors are ones, :1076-1082), decode, rotated multiclass NMS."""
        cfg = self.test_cfg if cfg is None else cfg
        out = []
        for i in range(len(img_metas)):
            mb, ms = [], []
            for cls_score, bbox_pred, angle_pred, points in zip(cls_scores, bbox_preds, angle_preds, all_level_points):
                scores = cls_score[i].detach().permute(1, 2, 0).reshape(-1, self.cls_out_channels).sigmoid()
                bp = torch.cat([bbox_pred[i].detach().permute(1, 2, 0).reshape(-1, 4),
                                angle_pred[i].detach().permute(1, 2, 0).reshape(-1, 1).float()], dim=1)
                nms_pre = cfg.get('nms_pre', -1)
                if 0 < nms_pre < scores.shape[0]:
                    _, topk_inds = scores.max(dim=1)[0].topk(nms_pre)
                    points, bp, scores = points[topk_inds], bp[topk_inds], scores[topk_inds]
                mb.append(self.bbox_coder.decode(points, bp, max_shape=img_metas[i]['img_shape']))
                ms.append(scores)
            bb, sc = torch.cat(mb), torch.cat(ms)
            if rescale:
                bb = torch.cat([bb[:, :4] / bb.new_tensor(img_metas[i]['scale_factor']), bb[:, 4:]], dim=1)
            sc = torch.cat([sc, sc.new_zeros(sc.shape[0], 1)], dim=1)
            if not with_nms:
                out.append((bb, sc))
                continue
            out.append(multiclass_nms_rotated(bb, sc, cfg['score_thr'], cfg['nms'], cfg['max_per_img'],
                                              score_factors=torch.ones_like(sc[:, 0])))
        return out
