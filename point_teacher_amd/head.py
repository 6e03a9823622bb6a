"""TS_P2BFCOSHead - the FCOS-style dense head with the MIL refinement branch, mirroring
/root/reference/HBB_TOD/mmdet/models/dense_heads/fcos_head_p2b_ts.py (+ anchor_free_head.py)
method for method.  Label assignment, target build, pseudo-box fusion, focal / DN-DIoU
losses, RoIAlign, bag scoring and bag selection all run in libpt_hip.so; images of a batch
are processed in ONE launch per op and no method synchronises the host (the reference
syncs on every `.nonzero()`, `len(pos_inds)`, `.item()` and `.tolist()`).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import functional as F
from . import planes as PL
from .core import (bbox2roi, bbox_cxcywh_to_xyxy, bbox_overlaps, bbox_xyxy_to_cxcywh, distance2bbox, mean0, multi_apply, zero_like_scalar,
                   multiclass_nms, reduce_mean, reduce_mean_many)
from .losses import diou_forward_masked
from .nn_modules import ConvModule, Scale
from .proposals import MIL_gen_proposals_from_cfg
from .registry import HEADS, build_assigner, build_bbox_coder, build_loss, build_roi_extractor

_SPLIT_GEMM = os.environ.get('PT_SPLIT_GEMM', '1') != '0'
_PLANE_FC = _SPLIT_GEMM and os.environ.get('PT_PLANE_FC', '1') != '0'       # 0: the round-3 routing (fp32 RoI blocks, blocked-plane GEMMs)
_MERGE_NEG = os.environ.get('PT_MERGE_NEG', '1') != '0'                     # 0: the negatives' 400 RoIs in a pass of their own (round 4)

INF = 1e8


def _cat(ts, dim=0):
    return torch.cat(list(ts), dim) if len(ts) else None


@HEADS.register_module()
class TS_P2BFCOSHead(nn.Module):
    """Constructor keywords as fcos_head_p2b_ts.py:80-146 / anchor_free_head.py:35-70."""

    def __init__(self, num_classes, in_channels, feat_channels=256, stacked_convs=4, strides=(4, 8, 16, 32, 64),
                 dcn_on_last_conv=False, conv_bias='auto', center_sampling=False, center_sample_radius=1.5,
                 norm_on_bbox=False, centerness_on_reg=False, mil_stack_conv=1, beta=0.25, top_k=3, num_stages=2,
                 bbox_roi_extractor=None,
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox_burn1=dict(type='DIoULoss', loss_weight=1.0),
                 loss_bbox_burn2=dict(type='DN_DIoULoss', loss_weight=1.0, hyper=0.1),
                 loss_bbox_denosing=dict(type='DN_DIoULoss', loss_weight=1.0, hyper=0.3),
                 loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 conv_cfg=None, norm_cfg=dict(type='GN', num_groups=32, requires_grad=True), train_cfg=None,
                 test_cfg=None, init_cfg=None, regress_ranges=None, **kwargs):
        super().__init__()
        assert norm_cfg is None or norm_cfg['type'] == 'GN', 'head norm is None (HBB configs) or GroupNorm (OBB config)'
        self.norm_cfg = norm_cfg
        self.num_classes = self.cls_out_channels = num_classes
        self.in_channels, self.feat_channels, self.stacked_convs = in_channels, feat_channels, stacked_convs
        self.strides = list(strides)
        self.dcn_on_last_conv = dcn_on_last_conv
        assert conv_bias == 'auto' or isinstance(conv_bias, bool)
        self.conv_bias = (norm_cfg is None) if conv_bias == 'auto' else conv_bias
        self.center_sampling, self.center_sample_radius = center_sampling, center_sample_radius
        self.norm_on_bbox, self.centerness_on_reg = norm_on_bbox, centerness_on_reg
        self.mil_stack_conv, self.num_stages = mil_stack_conv, num_stages
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox_burn1)
        self.loss_centerness = build_loss(loss_centerness)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if train_cfg:
            assert 'assigner' in train_cfg, 'assigner should be provided when train_cfg is set.'
            self.assigner = build_assigner(train_cfg['assigner'])
            self.fuse_assigner = build_assigner(train_cfg['fuse_assigner'])
            self.syn_assigner = build_assigner(train_cfg['syn_assigner'])
            self.pseudo_assigner = build_assigner(train_cfg['pseudo_assigner'])
        # MIL head (fcos_head_p2b_ts.py:174-187)
        self.beta, self.topk = beta, top_k
        self.bbox_roi_extractor = build_roi_extractor(bbox_roi_extractor)
        self.loss_mil_iou = build_loss(dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=0.25))
        self.loss_mil_bbox = build_loss(dict(type='SmoothL1Loss', beta=1.0, loss_weight=0.25))
        self.mil_bbox_decoder = build_bbox_coder(dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0],
                                                      target_stds=[1.0, 1.0, 1.0, 1.0]))
        self.loss_bbox_denosing = build_loss(loss_bbox_denosing)
        self.loss_bbox_burn2 = build_loss(loss_bbox_burn2)
        self.smoothl1 = build_loss(dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0))
        self._init_layers()
        self.init_weights()
        self._points_cache = {}

    # ------------------------------------------------------------------ layers --
    def _init_layers(self):
        """anchor_free_head.py:86-135 + fcos_head_p2b_ts.py:189-263 (same parameter names)."""
        def stack(n, cin):      # `dcn_on_last_conv`: the last conv of a tower is a DCNv2 (anchor_free_head.py:101-102,121-122)
            return nn.ModuleList([ConvModule(cin if i == 0 else self.feat_channels, self.feat_channels, 3, padding=1,
                                             bias=self.conv_bias, norm_cfg=self.norm_cfg,
                                             conv_cfg=dict(type='DCNv2') if (self.dcn_on_last_conv and i == self.stacked_convs - 1)
                                             else None) for i in range(n)])
        self.cls_convs = stack(self.stacked_convs, self.in_channels)
        self.reg_convs = stack(self.stacked_convs, self.in_channels)
        for l in list(self.cls_convs) + list(self.reg_convs):
            l.plane_group = 'towers'                               # census / fall-back group of their fp16 operands (planes.CENSUS)
        self.conv_cls = nn.Conv2d(self.feat_channels, self.cls_out_channels, 3, padding=1)
        self.conv_reg = nn.Conv2d(self.feat_channels, 4, 3, padding=1)
        self.conv_centerness = nn.Conv2d(self.feat_channels, 1, 3, padding=1)
        self.scales = nn.ModuleList([Scale(1.0) for _ in self.strides])
        self.conv_mil = stack(self.mil_stack_conv, self.in_channels)
        self.num_shared_fcs, self.fc_out_channels, self.roi_feat_area = 2, 1024, 7 * 7

        def fcs():
            d = self.in_channels * self.roi_feat_area
            return nn.ModuleList([nn.Linear(d if i == 0 else self.fc_out_channels, self.fc_out_channels)
                                  for i in range(self.num_shared_fcs)])
        self.shared_fcs, self.shared_fcs_refine = fcs(), fcs()            # built but unused at train time (:219-222)
        self.cls_fcs, self.ins_fcs = nn.ModuleList(), nn.ModuleList()
        self.relu = nn.ReLU(inplace=True)
        self.fc_cls, self.fc_ins, self.fc_reg, self.fc_iou = (nn.ModuleList() for _ in range(4))
        self.shared_fcs_bag, self.shared_fcs_reg = nn.ModuleList(), nn.ModuleList()
        for _ in range(self.num_stages):
            self.shared_fcs_bag.append(fcs())
            self.shared_fcs_reg.append(fcs())
            self.fc_cls.append(nn.Linear(self.fc_out_channels, self.num_classes))
            self.fc_ins.append(nn.Linear(self.fc_out_channels, self.num_classes))
            self.fc_reg.append(nn.Linear(self.fc_out_channels, 4))
            self.fc_iou.append(nn.Linear(self.fc_out_channels, 1))

    def init_weights(self):
        """init_cfg of fcos_head_p2b_ts.py:137-145: Normal(std .01) on convs, conv_cls bias_prob .01."""
        for name, m in self.named_modules():
            if isinstance(m, nn.Conv2d) and not name.endswith('conv_offset'):   # DCN offset predictors stay zero (mmcv)
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.conv_cls.bias, float(-torch.log(torch.tensor((1 - 0.01) / 0.01))))

    # ----------------------------------------------------------------- forward --
    def get_points(self, featmap_sizes, dtype, device, flatten=False):
        """anchor_free_head.py:318-335 + fcos_head_p2b_ts.py:1007-1017 (cached per size)."""
        out = []
        for (h, w), s in zip(featmap_sizes, self.strides):
            key = (int(h), int(w), s, str(device), dtype)
            if key not in self._points_cache:
                ys, xs = torch.meshgrid(torch.arange(h, device=device).to(dtype), torch.arange(w, device=device).to(dtype),
                                        indexing='ij')
                self._points_cache[key] = torch.stack((xs.reshape(-1) * s, ys.reshape(-1) * s), dim=-1) + s // 2
            out.append(self._points_cache[key])
        return out

    def forward(self, feats, branches='both'):
        """fcos_head_p2b_ts.py:302-324 -> (cls_scores, bbox_preds, centernesses, points) lists per level.
        `branches`: 'both' | 'cls' | 'reg' - the reference always evaluates both towers, but burn-in step 1 only reads the
        regression / centerness outputs of the synthetic pass (`loss`, :470-534) and only the classification loss of the
        augmented pass (:204-207 keeps `loss_cls` alone); the unread tower is dead work (its loss is never back-propagated)
        and is skipped: the entries of the skipped outputs are None."""
        sizes = [f.size()[-2:] for f in feats]
        pts = self.get_points(sizes, feats[0].dtype if feats[0].dtype == torch.float32 else torch.float32,
                              feats[0].device)
        return multi_apply(self.forward_single, feats, self.scales, self.strides, pts, [branches] * len(feats))

    def _towers(self, x, branches):
        """(cls_feat | None, reg_feat | None) as `branches` and `centerness_on_reg` need them."""
        need_cls, need_reg = branches != 'reg', branches != 'cls'
        need_cls = need_cls or (need_reg and not self.centerness_on_reg)
        cls_feat = reg_feat = None
        convs = (list(self.cls_convs) if need_cls else []) + (list(self.reg_convs) if need_reg else [])
        if (_PLANE_FC and convs and PL.f32_ok(x) and x.shape[0] * x.shape[2] * x.shape[3] >= ConvModule.plane_min_pixels
                and all(l.with_activation and not l.with_norm and PL.plane_conv_ok(l.conv) for l in convs)):
            # plane-native towers (anchor_free_head.py:198-219): the map is split ONCE, the activations travel between the four
            # convolutions of a tower as split planes (bias + ReLU in the epilogues, masks in the input-gradient epilogues) and
            # only the last one writes the fp32 map the 8 / 4 / 1-channel output convolutions read
            if need_cls and need_reg:
                xa, xb = PL.to_planes2(x, f16=PL.use_f16('towers'))   # (the towers on fp16 x 2 operands, three MFMA products)
            else:
                xa = xb = PL.to_planes(x, f16=PL.use_f16('towers'))

            def tower(t, layers):
                for i, l in enumerate(layers):
                    t = l(t, out_planes=i + 1 < len(layers))
                return t
            if need_cls:
                cls_feat = tower(xa, self.cls_convs)
            if need_reg:
                reg_feat = tower(xb, self.reg_convs)
            return cls_feat, reg_feat
        if need_cls:
            cls_feat = x
            for l in self.cls_convs:
                cls_feat = l(cls_feat)
        if need_reg:
            reg_feat = x
            for l in self.reg_convs:
                reg_feat = l(reg_feat)
        return cls_feat, reg_feat

    def forward_single(self, x, scale, stride, points, branches='both'):
        """fcos_head_p2b_ts.py:326-353"""
        cls_feat, reg_feat = self._towers(x, branches)
        cls_score = self.conv_cls(cls_feat) if branches != 'reg' else None
        if reg_feat is None:
            return cls_score, None, None, points
        bbox_pred = self.conv_reg(reg_feat)
        centerness = self.conv_centerness(reg_feat if self.centerness_on_reg else cls_feat)
        bbox_pred = scale(bbox_pred).float()
        if self.norm_on_bbox:
            bbox_pred = bbox_pred.clamp(min=0) * stride
        else:
            bbox_pred = bbox_pred.exp()
        return cls_score, bbox_pred, centerness, points

    def forward_mil(self, feats):
        """:1080-1090"""
        outs = []
        for x in feats:
            m = x
            for l in self.conv_mil:
                m = l(m)
            outs.append(m)
        return outs

    # --------------------------------------------------------------- flattening --
    def _flat(self, cls_scores, bbox_preds, centernesses):
        """[B,C,H,W] per level -> [B, P, C] (all levels concatenated, image-major); None for a branch that was not evaluated."""
        B = (cls_scores if cls_scores[0] is not None else bbox_preds)[0].size(0)
        fc = fr = ft = None
        if cls_scores[0] is not None:
            fc = torch.cat([c.permute(0, 2, 3, 1).reshape(B, -1, self.cls_out_channels) for c in cls_scores], 1).float().contiguous()
        if bbox_preds[0] is not None:
            fr = torch.cat([r.permute(0, 2, 3, 1).reshape(B, -1, 4) for r in bbox_preds], 1).float().contiguous()
            ft = torch.cat([t.permute(0, 2, 3, 1).reshape(B, -1) for t in centernesses], 1).float().contiguous()
        return fc, fr, ft

    # -------------------------------------------------------------- pseudo boxes --
    def get_pseudo_bbox(self, cls_scores, bbox_preds, centernesses, all_level_points, gt_points, gt_labels, gt_bboxes,
                        filter_scores, img_metas, img_list, gt_bboxes_ignore=None):
        """:357-375 -> (pseudo_bboxes, pseudo_points, pseudo_labels, mean_ious_pred, valid_masks).
        The last item is a list of boolean masks (the reference returns index tensors built
        through python sets, :791; nothing downstream reads them)."""
        assert len(cls_scores) == len(bbox_preds) == len(centernesses)
        fc, fr, _ = self._flat(cls_scores, bbox_preds, centernesses)
        B, P = fc.shape[:2]
        points = torch.cat(all_level_points, 0)
        counts = [int(l.shape[0]) for l in gt_labels]
        dev = fc.device
        off, tot = F.make_offsets(counts, dev)
        a = self.fuse_assigner
        gp, gl, gb = torch.cat(gt_points), torch.cat(gt_labels), torch.cat(gt_bboxes)
        gi, cand = F.fuse_assign(points, fr.reshape(-1, 4), fc.reshape(-1, self.cls_out_channels), gp, gl, off, B,
                                 a.num_pre, a.topk, a.cls_cost.weight, a.reg_cost.weight, a.location_cost.weight)
        ps = F.pseudo_boxes(points, fr.reshape(-1, 4), fc.reshape(-1, self.cls_out_channels), gp, gl, gb, off, B, gi, cand)
        nz = ps['nassigned'] > 0
        valid = nz & (ps['scores'] >= filter_scores)
        ious, nzs = torch.split(ps['iou'], counts), torch.split(nz, counts)
        mean_iou = sum((i.sum() / z.sum()) if c else i.new_zeros(()) for i, z, c in zip(ious, nzs, counts)) / B
        return (list(torch.split(ps['bboxes'], counts)), list(torch.split(ps['points'], counts)), list(gt_labels),
                mean_iou, list(torch.split(valid, counts)))

    # --------------------------------------------------------------------- losses --
    def _reg_branch(self, points, B, fr, ft, gi_reg, boxes, box_labels, off, loss_mod, dn, cls_pos=None):
        """Shared tail of loss_pseudo (:436-463) and loss (:502-532): targets, centerness-weighted
        (DN-)DIoU and centerness BCE, dense over all B*P points with the positive mask.  `cls_pos`
        (optional device scalar): the positive count of the classification branch, reduced over the
        ranks together with this branch's two normalisers; returned as a third value then."""
        P = points.shape[0]
        labels_reg, tg, ctr_t = F.fcos_targets(points, gi_reg, boxes, box_labels, off, B, self.num_classes)
        pos = labels_reg < self.num_classes
        norms = reduce_mean_many(pos.sum(), ctr_t.sum().detach(), *([cls_pos] if cls_pos is not None else []))
        num_pos, ctr_den = norms[0].clamp(min=1.0), norms[1].clamp(min=1e-6)
        if fr is None:                                 # regression branch not evaluated (burn-in step 1, augmented pass)
            return None, None, norms[2].clamp(min=1.0)
        pts = points.repeat(B, 1)
        pred = distance2bbox(pts, fr.reshape(-1, 4))
        tgt = distance2bbox(pts, tg)
        if dn:
            loss_bbox = loss_mod.forward_masked(pred, tgt, pos, ctr_t, ctr_den)
        else:
            loss_bbox = diou_forward_masked(loss_mod, pred, tgt, pos, ctr_t, ctr_den)
        bce = TF.binary_cross_entropy_with_logits(ft.reshape(-1), ctr_t, reduction='none')
        loss_ctr = self.loss_centerness.loss_weight * torch.where(pos, bce, zero_like_scalar(bce)).sum() / num_pos
        if cls_pos is not None:
            return loss_bbox, loss_ctr, norms[2].clamp(min=1.0)
        return loss_bbox, loss_ctr

    def loss_pseudo(self, cls_scores, bbox_preds, centernesses, all_level_points, gt_points, gt_labels, pseudo_points,
                    pseudo_labels, pseudo_bboxes, gt_augument_ignore, img_metas, img_list, burn_in_step1,
                    gt_bboxes_ignore=None, gt_valid=None, pseudo_valid=None):
        """:380-465.  gt_valid / pseudo_valid: optional per-image boolean masks (see
        proposals.strong_augmentation_masked); without them the lists are used as given."""
        assert len(cls_scores) == len(bbox_preds) == len(centernesses)
        fc, fr, ft = self._flat(cls_scores, bbox_preds, centernesses)
        B, P = fc.shape[:2]
        points = torch.cat(all_level_points, 0)
        dev = fc.device
        # classification labels: assigner (1,1) on the points (:665-670)
        gcounts = [int(p.shape[0]) for p in gt_points]
        goff, _ = F.make_offsets(gcounts, dev)
        gi_cls = F.topk_assign(points, torch.cat(gt_points), goff, B, self.assigner.num_pre,
                               gt_valid=_cat(gt_valid) if gt_valid is not None else None)
        labels, _, _ = F.fcos_targets(points, gi_cls, None, torch.cat(gt_labels), goff, B, self.num_classes)
        # regression: pseudo_assigner (3,3) on the pseudo-box centres (:683-706)
        pcounts = [int(b.shape[0]) for b in pseudo_bboxes]
        poff, _ = F.make_offsets(pcounts, dev)
        pb = torch.cat(pseudo_bboxes)
        gi_reg = F.topk_assign(points, bbox_xyxy_to_cxcywh(pb), poff, B, self.pseudo_assigner.num_pre,
                               gt_valid=_cat(pseudo_valid) if pseudo_valid is not None else None)
        loss_bbox, loss_ctr, num_pos = self._reg_branch(points, B, fr, ft, gi_reg, pb, torch.cat(pseudo_labels), poff,
                                                        self.loss_bbox_burn2, dn=True,
                                                        cls_pos=(labels < self.num_classes).sum())
        loss_cls = self.loss_cls(fc.reshape(-1, self.cls_out_channels), labels, weight=None, avg_factor=num_pos)
        return loss_cls, loss_bbox, loss_ctr

    def loss(self, cls_scores, bbox_preds, centernesses, all_level_points, gt_bboxes, img_metas, gt_bboxes_ignore=None,
             gt_valid=None):
        """:470-534 (burn-in step 1: synthetic rectangles, syn_assigner (3,3), plain DIoU)."""
        _, fr, ft = self._flat(cls_scores, bbox_preds, centernesses)
        B = fr.shape[0]
        points = torch.cat(all_level_points, 0)
        counts = [int(b.shape[0]) for b in gt_bboxes]
        off, _ = F.make_offsets(counts, fr.device)
        gb = torch.cat(gt_bboxes)
        gi = F.topk_assign(points, bbox_xyxy_to_cxcywh(gb), off, B, self.syn_assigner.num_pre,
                           gt_valid=_cat(gt_valid) if gt_valid is not None else None)
        return self._reg_branch(points, B, fr, ft, gi, gb, None, off, self.loss_bbox, dn=False)

    def centerness_target(self, pos_bbox_targets):
        """:1019-1038"""
        lr, tb = pos_bbox_targets[:, [0, 2]], pos_bbox_targets[:, [1, 3]]
        if len(lr) == 0:
            return lr[..., 0]
        c = (lr.min(dim=-1)[0].clamp(min=0.01) / lr.max(dim=-1)[0]) * (tb.min(dim=-1)[0].clamp(min=0.01) / tb.max(dim=-1)[0])
        return torch.sqrt(c)

    def gfocal_loss(self, p, q, w=1.0, eps=1e-6):
        """:1074-1078 (torch form, kept for API parity; training uses the fused kernel)."""
        return -(((p - q) ** 2) * (q * (p + eps).log() + (1 - q) * (1 - p + eps).log()) * w).sum(dim=-1)

    # ----------------------------------------------------------------- MIL head --
    SPLIT_GEMM_MIN_ROWS = 2048     # below it the tile count cannot fill 256 CUs and the library GEMM wins (profiles/r03/gemm_bf16x6_vs_library.txt)

    def _fc_stack(self, fcs, x):
        """Linear + ReLU stack (:1202-1236, :1240-1256).  On the MI355X the fp32 products run on the bf16 matrix cores as six
        split-bf16 MFMA products with fp32 accumulation (`functional.split_linear`, csrc/gemm_split.hip: error against float64
        below the fp32 library kernel's, 1.4-1.6x its speed at K >= 5 000 RoIs); bias + ReLU are the GEMM's epilogue.
        PT_SPLIT_GEMM=0: library GEMMs."""
        if isinstance(x, PL.PlaneAct) or (_PLANE_FC and x.is_cuda and x.dtype == torch.float32 and x.shape[0] >= self.SPLIT_GEMM_MIN_ROWS
                                          and all(PL.linear_ok(fc, x.shape[0]) for fc in fcs)):
            # the RoI blocks arrive as split planes (RoIAlign wrote them) and stay planes between the layers: every product is a
            # 1x1 convolution over the RoIs on pt_conv_bf16x6, bias + ReLU in its epilogue; backward: ReLU masks in the epilogues,
            # weight AND bias gradients from pt_conv_wgrad_bf16x6 (no transposed split, no column-sum pass)
            for i, fc in enumerate(fcs):
                x = PL.linear(x, fc, relu=True, out_planes=i + 1 < len(fcs))
            return x
        use = x.is_cuda and x.dtype == torch.float32 and x.shape[0] >= self.SPLIT_GEMM_MIN_ROWS and _SPLIT_GEMM
        for fc in fcs:
            if use:
                x = F.split_linear(x, fc.weight, fc.bias, relu=True)
            else:
                x = TF.relu(fc(x), inplace=True)
        return x

    def _roi_feats(self, x, rois, group=1, tail=0):
        """RoI blocks [K, C * 49] for an FC stack: as split planes straight from RoIAlign when the stack runs on the plane kernels
        (K >= SPLIT_GEMM_MIN_ROWS on the MI355X), else the flattened fp32 block."""
        ext = self.bbox_roi_extractor
        if (_PLANE_FC and rois.shape[0] >= self.SPLIT_GEMM_MIN_ROWS and ext.num_inputs == 1 and type(ext).__name__ == 'SingleRoIExtractor'
                and x[0].is_cuda and not torch.is_autocast_enabled()):
            f = ext(x[:1], rois, group=group, planes=True, tail=tail)
            return f if isinstance(f, PL.PlaneAct) else f.flatten(1)
        return ext(x[:ext.num_inputs], rois, group=group).flatten(1)

    @staticmethod
    def _bag_group(U1, U2):
        """The U2 boxes shaken out of one coarse box are consecutive and overlap: the RoIAlign kernels hand runs of
        them to one workgroup."""
        return U2

    def mil_bag_extensive(self, num_gt, num_gt_pre_image, x, img_metas, proposals_list, proposals_valid_list,
                          proposals_reference_list, proposals_real_list, bbox_results, fine_proposal_cfg, stage,
                          bag_weight=None):
        """:1182-1236.  bag_weight (optional, per image [n_i*U1*U2] float) multiplies the validity
        weights - used by step 1 where padded synthetic boxes must not count."""
        U1 = max(sum(p.shape[0] for p in proposals_list) // max(num_gt, 1), 1)   # :1185 (same U1 for every image)
        bbox_results['base_shaking_num'] = U1
        points_list = [bbox_xyxy_to_cxcywh(p)[:, :2] for p in proposals_list]
        ext, ext_valid, _, ext_ref = MIL_gen_proposals_from_cfg(points_list, proposals_list, fine_proposal_cfg,
                                                               proposals_reference_list, img_metas)
        num_aug = len(fine_proposal_cfg['base_ratios']) ** 2 * (1 + 4 * len(fine_proposal_cfg['shake_ratio'] or []))
        ext_real = [r.unsqueeze(1).repeat(1, num_aug, 1).reshape(-1, 4) for r in proposals_real_list]
        bbox_results['base_bags'], bbox_results['base_bags_valid'] = proposals_list, proposals_valid_list
        bags, real, ref = torch.cat(ext), torch.cat(ext_real), torch.cat(ext_ref)
        valid = torch.cat(ext_valid).reshape(-1)
        K = bags.shape[0]
        if bag_weight is None:
            bbox_results['coarse_bags_iou'] = mean0(bbox_overlaps(bags, real, is_aligned=True))
            wgt, avg = valid.float(), float(max(K, 1))
            wsum = None
        else:
            bw = torch.cat(bag_weight)
            wsum = bw.sum().clamp(min=1.0)
            bbox_results['coarse_bags_iou'] = (bbox_overlaps(bags, real, is_aligned=True) * bw).sum() / wsum
            wgt, avg = valid.float() * bw, wsum
        bbox_results['extensive_shaking_num'] = U2 = num_aug
        rois = bbox2roi(ext)
        feats = self._roi_feats(x, rois, group=self._bag_group(U1, U2))
        bbox_pred = self.fc_reg[stage](self._fc_stack(self.shared_fcs_reg[stage], feats))
        del feats
        bbox_pred = self.mil_bbox_decoder.decode(bags, bbox_pred, max_shape=img_metas[0]['img_shape'])
        pred_d = bbox_pred.detach()
        if bag_weight is None:
            bbox_results['loss_mil_bbox'] = self.loss_bbox_denosing(bbox_pred, ref, weight=wgt, avg_factor=avg)
        else:   # padded rows must not enter the batch-mean term of DN-DIoU (iou_loss.py:412) either
            bbox_results['loss_mil_bbox'] = self.loss_bbox_denosing.forward_masked(
                bbox_pred, ref, torch.cat(bag_weight) > 0, valid.float(), avg)
        ri = bbox_overlaps(pred_d, real, is_aligned=True)
        bbox_results['refine_bags_iou'] = mean0(ri) if wsum is None else (ri * torch.cat(bag_weight)).sum() / wsum
        sizes = [e.shape[0] for e in ext]
        bbox_results['extensive_bags'] = list(torch.split(pred_d, sizes))
        bbox_results['extensive_bags_valid'] = ext_valid
        bbox_results['extensive_bags_reference'] = ext_ref
        bbox_results['extensive_bags_real'] = ext_real

    def mil_bag_classifier(self, num_gt, x, bbox_results, stage, neg_rois=None):
        """:1240-1256.  neg_rois: the negative proposals of :1266-1271 ride along - they pass through the SAME RoIAlign, FC stack
        (`shared_fcs_bag`) and `fc_cls` as the bags, row by row, so their 400 rows are appended to the K bag rows of one launch each
        (forward, input and weight gradients) instead of a second, launch-bound pass through library GEMMs."""
        rois = bbox2roi(bbox_results['extensive_bags'])
        U1, U2 = bbox_results['base_shaking_num'], bbox_results['extensive_shaking_num']
        K = rois.shape[0]
        if neg_rois is not None:
            rois = torch.cat([rois, neg_rois], 0)
        feats = self._fc_stack(self.shared_fcs_bag[stage], self._roi_feats(x, rois, group=self._bag_group(U1, U2),
                                                                            tail=rois.shape[0] - K))
        cls = self.fc_cls[stage](feats)
        bbox_results['cls_score'] = cls[:K].view(num_gt, U1, U2, self.num_classes)
        bbox_results['ins_score'] = self.fc_ins[stage](feats[:K]).view(num_gt, U1, U2, self.num_classes)
        if neg_rois is not None:
            bbox_results['neg_cls_score'] = cls[K:]

    def forward_mil_head(self, num_gt, num_gt_pre_image, x, proposals_list, proposals_valid_list,
                         proposals_reference_list, proposals_real_list, img_metas, fine_proposal_cfg, stage,
                         neg_proposal_list=None, neg_weight_list=None, bag_weight=None, need_classifier=True):
        """:1259-1277.  need_classifier=False skips mil_bag_classifier when the caller only consumes
        `loss_mil_bbox` (the synthetic branch of step 1, :1301-1305: the reference computes the bag
        scores there and throws them away)."""
        bbox_results = {}
        # the RoIAlign kernels want the [B,H,W,C] layout (1 KiB coalesced rows, full-rate atomics);
        # one 20 MB layout change here is shared by the three RoIAlign calls of this stage
        x = [f.float().contiguous(memory_format=torch.channels_last) for f in x[:self.bbox_roi_extractor.num_inputs]]
        self.mil_bag_extensive(num_gt, num_gt_pre_image, x, img_metas, proposals_list, proposals_valid_list,
                               proposals_reference_list, proposals_real_list, bbox_results, fine_proposal_cfg, stage,
                               bag_weight=bag_weight)
        merge_neg = need_classifier and neg_proposal_list is not None and _MERGE_NEG
        if need_classifier:
            self.mil_bag_classifier(num_gt, x, bbox_results, stage, neg_rois=bbox2roi(neg_proposal_list) if merge_neg else None)
        if neg_proposal_list is not None and not merge_neg:
            rois = bbox2roi(neg_proposal_list)
            feats = self._fc_stack(self.shared_fcs_bag[stage],
                                   self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois).flatten(1))
            bbox_results['neg_cls_score'] = self.fc_cls[stage](feats)
        return bbox_results

    def mil_bag_training(self, bbox_results, gt_labels, neg_weight_list):
        """:1147-1180 through the fused bag-loss kernels."""
        cls, ins = bbox_results['cls_score'], bbox_results['ins_score']
        N, U1, U2, C = cls.shape
        labels = torch.cat(gt_labels).unsqueeze(1).repeat(1, U1).reshape(-1)
        valid = torch.cat(bbox_results['extensive_bags_valid'], 0).reshape(-1)
        total, nvalid = F.mil_bag_loss_sum(cls.reshape(N * U1, U2, C), ins.reshape(N * U1, U2, C), valid, labels)
        num_sample = nvalid.clamp(min=1.0)
        loss = total / num_sample
        if neg_weight_list is not None:
            loss = loss + F.mil_neg_loss_sum(bbox_results['neg_cls_score'], torch.cat(neg_weight_list)) / num_sample
        return loss

    def mil_bag_selection(self, bbox_results, img_metas, pseudo_bboxes, pseudo_labels):
        """:1112-1145 (+ _single :1092-1110) in one launch for the whole batch."""
        cls, ins = bbox_results['cls_score'], bbox_results['ins_score']
        N, U1, U2, C = cls.shape
        h, w, _ = img_metas[0]['img_shape']
        merged = F.mil_bag_select(cls, ins, torch.cat(bbox_results['extensive_bags_valid'], 0).reshape(-1),
                                  torch.cat(pseudo_labels), torch.cat(bbox_results['extensive_bags'], 0),
                                  torch.cat(pseudo_bboxes), U1, U2, self.topk, self.beta, (h, w))
        return list(torch.split(merged, [len(b) for b in pseudo_bboxes]))

    def MIL_head_burn_in_step2(self, x, img_metas, proposals_list, proposals_valid_list, proposals_reference_list,
                               proposals_real_list, neg_proposal_list, neg_weight_list, pseudo_bboxes, pseudo_labels,
                               fine_proposal_cfg, stage):
        """:1318-1344"""
        num_gt = sum(b.shape[0] for b in pseudo_bboxes)
        npi = [b.shape[0] for b in pseudo_bboxes]
        r = self.forward_mil_head(num_gt, npi, x, proposals_list, proposals_valid_list, proposals_reference_list,
                                  proposals_real_list, img_metas, fine_proposal_cfg, stage, neg_proposal_list,
                                  neg_weight_list)
        losses = {f'stage{stage}_loss_mil_bbox': r['loss_mil_bbox'],
                  f'stage{stage}_loss_mil_bags': self.mil_bag_training(r, pseudo_labels, neg_weight_list),
                  f'stage{stage}_coarse_bags_iou': r['coarse_bags_iou'],
                  f'stage{stage}_refine_bags_iou': r['refine_bags_iou']}
        return losses, self.mil_bag_selection(r, img_metas, pseudo_bboxes, pseudo_labels)

    def MIL_head_burn_in_step1(self, x_ori, x_synethic, img_metas, proposals_list, proposals_valid_list,
                               proposals_reference_list, proposals_real_list, syn_proposals_list,
                               syn_proposals_valid_list, syn_proposals_reference_list, syn_proposals_real_list,
                               neg_proposal_list, neg_weight_list, synthetic_bboxes, pseudo_bboxes, pseudo_labels,
                               fine_proposal_cfg, stage, syn_bag_weight=None):
        """:1279-1316: the regression branch trains on the SYNTHETIC bags, the bag classifier on the real ones."""
        num_syn = sum(b.shape[0] for b in synthetic_bboxes)
        nsi = [b.shape[0] for b in synthetic_bboxes]
        num_gt = sum(b.shape[0] for b in pseudo_bboxes)
        npi = [b.shape[0] for b in pseudo_bboxes]
        losses = {}
        rs = self.forward_mil_head(num_syn, nsi, x_synethic, syn_proposals_list, syn_proposals_valid_list,
                                   syn_proposals_reference_list, syn_proposals_real_list, img_metas, fine_proposal_cfg,
                                   stage, bag_weight=syn_bag_weight, need_classifier=False)
        losses[f'stage{stage}_loss_mil_bbox'] = rs['loss_mil_bbox']
        del rs
        r = self.forward_mil_head(num_gt, npi, x_ori, proposals_list, proposals_valid_list, proposals_reference_list,
                                  proposals_real_list, img_metas, fine_proposal_cfg, stage, neg_proposal_list,
                                  neg_weight_list)
        losses[f'stage{stage}_loss_mil_bags'] = self.mil_bag_training(r, pseudo_labels, neg_weight_list)
        losses[f'stage{stage}_coarse_bags_iou'] = r['coarse_bags_iou']
        losses[f'stage{stage}_refine_bags_iou'] = r['refine_bags_iou']
        return losses, self.mil_bag_selection(r, img_metas, pseudo_bboxes, pseudo_labels)

    # ------------------------------------------------------------------ inference --
    def simple_test(self, feats, img_metas, rescale=False):
        return self.get_bboxes(*self.forward(feats), img_metas, rescale=rescale)

    def get_bboxes(self, cls_scores, bbox_preds, centernesses, all_level_points, img_metas, cfg=None, rescale=False,
                   with_nms=True):
        """:796-1005 (non-ONNX path)."""
        cfg = self.test_cfg if cfg is None else cfg
        B = cls_scores[0].shape[0]
        img_shapes = [m['img_shape'] for m in img_metas]
        nms_pre = cfg.get('nms_pre', -1)
        mb, ms, mc = [], [], []
        for cls_score, bbox_pred, ctr, points in zip(cls_scores, bbox_preds, centernesses, all_level_points):
            scores = cls_score.detach().permute(0, 2, 3, 1).reshape(B, -1, self.cls_out_channels).sigmoid()
            ctr = ctr.detach().permute(0, 2, 3, 1).reshape(B, -1).sigmoid()
            bbox_pred = bbox_pred.detach().permute(0, 2, 3, 1).reshape(B, -1, 4)
            points = points.expand(B, -1, 2)
            if 0 < nms_pre < bbox_pred.shape[1]:
                max_scores, _ = (scores * ctr[..., None]).max(-1)
                _, topk_inds = max_scores.topk(nms_pre)
                bi = torch.arange(B, device=scores.device).view(-1, 1).expand_as(topk_inds)
                points, bbox_pred = points[bi, topk_inds, :], bbox_pred[bi, topk_inds, :]
                scores, ctr = scores[bi, topk_inds, :], ctr[bi, topk_inds]
            mb.append(distance2bbox(points, bbox_pred, max_shape=img_shapes))
            ms.append(scores)
            mc.append(ctr)
        bb, sc, ct = torch.cat(mb, 1), torch.cat(ms, 1), torch.cat(mc, 1)
        if rescale:
            import numpy as np
            bb = bb / bb.new_tensor(np.stack([np.asarray(m['scale_factor'], np.float32) for m in img_metas])).unsqueeze(1)
        sc = torch.cat([sc, sc.new_zeros(B, sc.shape[1], 1)], dim=-1)
        if not with_nms:
            return [tuple(t) for t in zip(bb, sc, ct)]
        return [multiclass_nms(b, s, cfg['score_thr'], cfg['nms'], cfg['max_per_img'], score_factors=c)
                for b, s, c in zip(bb, sc, ct)]
