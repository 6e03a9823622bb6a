"""The fully supervised FCOS baseline the Point-Teacher paper compares against (first entry):
`configs/baselines/aitodv2_fcos_r50_1x.py` = detector `FCOS` + `FCOSHead` on the same ResNet-50 / FPN.

    FCOS      /root/reference/HBB_TOD/mmdet/models/detectors/fcos.py:6-28 over single_stage.py (forward_train :80-104)
    FCOSHead  /root/reference/HBB_TOD/mmdet/models/dense_heads/fcos_head.py:24-1035 over anchor_free_head.py

The network, the inference path (`get_bboxes`, NMS) and the loss kernels are the ones of the Point-Teacher student; what
the baseline adds is FCOS' own dense target assignment (regress ranges + centre sampling + smallest area), which the
reference evaluates as five [P, G] tensors per image and which is ONE launch here (`pt_fcos_dense_targets`).  The loss
stays dense over all B*P points with a positive mask - no `nonzero`, no host synchronisation."""
import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import functional as F
from .core import distance2bbox, reduce_mean_many
from .detectors import Student_FCOS
from .head import TS_P2BFCOSHead
from .losses import diou_forward_masked
from .nn_modules import ConvModule, Scale
from .registry import DETECTORS, HEADS, build_loss

INF = 1e8


@HEADS.register_module()
class FCOSHead(TS_P2BFCOSHead):
    """Parameter names of mmdet's FCOSHead (cls_convs / reg_convs / conv_cls / conv_reg / conv_centerness / scales).
    Inherits the point grid, the flattening and `get_bboxes` of the Point-Teacher head; none of its MIL layers."""

    def __init__(self, num_classes, in_channels, feat_channels=256, stacked_convs=4, strides=(4, 8, 16, 32, 64),
                 regress_ranges=((-1, 64), (64, 128), (128, 256), (256, 512), (512, INF)), center_sampling=False,
                 center_sample_radius=1.5, norm_on_bbox=False, centerness_on_reg=False, dcn_on_last_conv=False, conv_bias='auto',
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox=dict(type='IoULoss', loss_weight=1.0),
                 loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 conv_cfg=None, norm_cfg=dict(type='GN', num_groups=32, requires_grad=True), train_cfg=None, test_cfg=None,
                 init_cfg=None, **kwargs):
        nn.Module.__init__(self)
        assert norm_cfg is None or norm_cfg['type'] == 'GN'
        assert len(regress_ranges) == len(strides)
        if loss_bbox['type'] != 'DIoULoss':
            raise NotImplementedError(f"loss_bbox {loss_bbox['type']}: the AI-TOD baseline config trains with DIoULoss "
                                      '(the fused pt_diou_* kernels)')
        self.norm_cfg, self.conv_cfg = norm_cfg, conv_cfg
        self.num_classes = self.cls_out_channels = num_classes
        self.in_channels, self.feat_channels, self.stacked_convs = in_channels, feat_channels, stacked_convs
        self.strides, self.regress_ranges = list(strides), tuple(tuple(r) for r in regress_ranges)
        self.dcn_on_last_conv = dcn_on_last_conv
        assert conv_bias == 'auto' or isinstance(conv_bias, bool)
        self.conv_bias = (norm_cfg is None) if conv_bias == 'auto' else conv_bias
        self.center_sampling, self.center_sample_radius = center_sampling, center_sample_radius
        self.norm_on_bbox, self.centerness_on_reg = norm_on_bbox, centerness_on_reg
        self.loss_cls, self.loss_bbox, self.loss_centerness = build_loss(loss_cls), build_loss(loss_bbox), build_loss(loss_centerness)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self._init_layers()
        self.init_weights()
        self._points_cache, self._target_cache = {}, {}

    def _init_layers(self):
        """anchor_free_head.py:86-135 + fcos_head.py:121-170"""
        def stack(n, cin):
            return nn.ModuleList([ConvModule(cin if i == 0 else self.feat_channels, self.feat_channels, 3, padding=1,
                                             bias=self.conv_bias, norm_cfg=self.norm_cfg,
                                             conv_cfg=dict(type='DCNv2') if (self.dcn_on_last_conv and i == self.stacked_convs - 1)
                                             else None) for i in range(n)])
        self.cls_convs = stack(self.stacked_convs, self.in_channels)
        self.reg_convs = stack(self.stacked_convs, self.in_channels)
        self.conv_cls = nn.Conv2d(self.feat_channels, self.cls_out_channels, 3, padding=1)
        self.conv_reg = nn.Conv2d(self.feat_channels, 4, 3, padding=1)
        self.conv_centerness = nn.Conv2d(self.feat_channels, 1, 3, padding=1)
        self.scales = nn.ModuleList([Scale(1.0) for _ in self.strides])

    def forward_single(self, x, scale, stride, points, branches='both'):
        """fcos_head.py:195-252: with norm_on_bbox the regression is in units of the stride while training.  (`branches` of
        TS_P2BFCOSHead.forward is accepted and ignored: the supervised head always reads both towers.)"""
        cls_feat = reg_feat = x
        for l in self.cls_convs:
            cls_feat = l(cls_feat)
        cls_score = self.conv_cls(cls_feat)
        for l in self.reg_convs:
            reg_feat = l(reg_feat)
        bbox_pred = self.conv_reg(reg_feat)
        centerness = self.conv_centerness(reg_feat if self.centerness_on_reg else cls_feat)
        bbox_pred = scale(bbox_pred).float()
        if self.norm_on_bbox:
            bbox_pred = TF.relu(bbox_pred)
            if not self.training:
                bbox_pred = bbox_pred * stride
        else:
            bbox_pred = bbox_pred.exp()
        return cls_score, bbox_pred, centerness, points

    def _point_tables(self, all_level_points):
        """Per-point regress range, centre-sampling radius and target normaliser of get_targets (:806-823, :917-931, :870-873)."""
        key = tuple(int(p.shape[0]) for p in all_level_points) + (str(all_level_points[0].device),)
        if key not in self._target_cache:
            dev = all_level_points[0].device
            rr, rad, nm = [], [], []
            for pts, rng, s in zip(all_level_points, self.regress_ranges, self.strides):
                n = pts.shape[0]
                rr.append(torch.tensor(rng, dtype=torch.float32, device=dev)[None].expand(n, 2))
                rad.append(torch.full((n,), s * self.center_sample_radius, dtype=torch.float32, device=dev))
                nm.append(torch.full((n,), float(s) if self.norm_on_bbox else 1.0, dtype=torch.float32, device=dev))
            self._target_cache[key] = (torch.cat(rr).contiguous(), torch.cat(rad), torch.cat(nm))
        return self._target_cache[key]

    def get_targets(self, all_level_points, gt_bboxes, gt_labels):
        """fcos_head.py:698-875 -> image-major (labels [B*P], bbox_targets [B*P,4], centerness targets [B*P])."""
        points = torch.cat(all_level_points, 0)
        rr, rad, nm = self._point_tables(all_level_points)
        off, _ = F.make_offsets([int(b.shape[0]) for b in gt_bboxes], points.device)
        return F.fcos_dense_targets(points, rr, rad, nm, torch.cat(gt_bboxes), torch.cat(gt_labels), off, len(gt_bboxes),
                                    self.num_classes, self.center_sampling)

    def loss(self, cls_scores, bbox_preds, centernesses, all_level_points, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
        """fcos_head.py:254-473"""
        assert len(cls_scores) == len(bbox_preds) == len(centernesses)
        fc, fr, ft = self._flat(cls_scores, bbox_preds, centernesses)
        B = fc.shape[0]
        labels, tg, ctr_t = self.get_targets(all_level_points, gt_bboxes, gt_labels)
        pos = labels < self.num_classes
        norms = reduce_mean_many(pos.sum(), ctr_t.sum().detach())
        num_pos, ctr_den = norms[0].clamp(min=1.0), norms[1].clamp(min=1e-6)
        loss_cls = self.loss_cls(fc.reshape(-1, self.cls_out_channels), labels, avg_factor=num_pos)
        pts = torch.cat(all_level_points, 0).repeat(B, 1)
        pred = distance2bbox(pts, fr.reshape(-1, 4))
        tgt = distance2bbox(pts, tg)
        loss_bbox = diou_forward_masked(self.loss_bbox, pred, tgt, pos, ctr_t, ctr_den)
        bce = TF.binary_cross_entropy_with_logits(ft.reshape(-1), ctr_t, reduction='none')
        loss_ctr = self.loss_centerness.loss_weight * torch.where(pos, bce, torch.zeros_like(bce)).sum() / num_pos
        return dict(loss_cls=loss_cls, loss_bbox=loss_bbox, loss_centerness=loss_ctr)

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, **kwargs):
        """base_dense_head.py forward_train: loss(*outs, gt_bboxes, gt_labels, img_metas)."""
        return self.loss(*self(x), gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=gt_bboxes_ignore)


@DETECTORS.register_module()
class FCOS(Student_FCOS):
    """detectors/fcos.py over single_stage.py:80-104 (`forward_train` = head losses on the extracted features)."""

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
        x = self.extract_feat(img.to(torch.float))
        return self.bbox_head.forward_train(x, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore)
