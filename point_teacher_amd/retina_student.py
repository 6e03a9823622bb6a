"""A second student architecture INSIDE the teacher-student detector ("other student architectures under the
same teacher"; README.md:11 "can be integrated into various detectors"): `TS_P2BRetinaHead`, a RetinaNet-style dense head -
anchor-based box parametrisation (`AnchorGenerator` + `DeltaXYWHBBoxCoder`, A anchors per location), no centerness branch,
L1 loss on the encoded deltas (retina_head.py:9-114, anchor_head.py of the reference) - that plugs into `TS_P2B_FCOS`
unchanged: EMA teacher, point -> pseudo-box generation with the FUSE assigner, the dynamic-MIL refinement head, strong
augmentation and the burn-in schedule all come from `TS_P2BFCOSHead` / `TS_P2B_FCOS`.

What is new relative to the FCOS student is only the dense part:
  * every (location, base anchor) pair is a "point" of the Point-Teacher assigners (P = H * W * A, anchor centres);
  * the regression branch emits (dx, dy, dw, dh) relative to its anchor; they are decoded to boxes and handed to the shared
    kernels as (l, t, r, b) distances from the anchor centre, so pseudo-box fusion, target build and NMS are the same launches;
  * positives are regressed with L1 on the deltas of their anchor (RetinaNet's loss), normalised by their number;
  * `loss_centerness` is reported as 0 (the key set of the loss dict stays the one the runner / logger know).
The reference ships no such head (its `configs/baselines/aitodv2_retinanet_r50_1x.py` is the SUPERVISED RetinaNet, built in
retina_baseline.py), so there is nothing to pin it against: parity unpinned, covered by property tests
(tests/test_retina_student.py)."""
import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import functional as F
from .core import distance2bbox, reduce_mean_many
from .head import TS_P2BFCOSHead
from .registry import HEADS
from .retina_baseline import AnchorGenerator, bbox2delta


@HEADS.register_module()
class TS_P2BRetinaHead(TS_P2BFCOSHead):
    def __init__(self, num_classes, in_channels,
                 anchor_generator=dict(type='AnchorGenerator', octave_base_scale=2, scales_per_octave=1, ratios=[0.5, 1.0, 2.0],
                                       strides=[8]),
                 loss_bbox_l1_weight=1.0, wh_ratio_clip=16 / 1000, **kwargs):
        ag = dict(anchor_generator)
        assert ag.pop('type', 'AnchorGenerator') == 'AnchorGenerator'
        self.anchor_generator = AnchorGenerator(**ag)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self.loss_bbox_l1_weight, self.wh_ratio_clip = loss_bbox_l1_weight, wh_ratio_clip
        kwargs.setdefault('strides', [s[0] for s in self.anchor_generator.strides])
        super().__init__(num_classes, in_channels, **kwargs)
        self._anchor_cache = {}

    def _init_layers(self):
        """retina_head.py:60-88: the towers of the parent, `retina_cls` (A*C) and `retina_reg` (A*4) predictors."""
        super()._init_layers()
        del self.conv_cls, self.conv_reg, self.conv_centerness, self.scales
        self.retina_cls = nn.Conv2d(self.feat_channels, self.num_anchors * self.cls_out_channels, 3, padding=1)
        self.retina_reg = nn.Conv2d(self.feat_channels, self.num_anchors * 4, 3, padding=1)
        self.scales = nn.ModuleList()

    def init_weights(self):
        for name, m in self.named_modules():
            if isinstance(m, nn.Conv2d) and not name.endswith('conv_offset'):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.retina_cls.bias, float(-torch.log(torch.tensor((1 - 0.01) / 0.01))))

    # ------------------------------------------------------------------ anchors as points --
    def _anchors(self, h, w, device):
        """([H*W*A, 4] anchors in (y, x, base anchor) order, their centres [H*W*A, 2]) for a stride-s map, cached."""
        key = (int(h), int(w), str(device))
        if key not in self._anchor_cache:
            a = self.anchor_generator.grid_anchors([(h, w)], device=device)[0].float()
            # mmdet centres its anchors on the top-left corner of a cell (center_offset 0); FCOS points sit at the cell
            # centre - keep the anchor grid as configured and use its own centres as the points
            self._anchor_cache[key] = (a, torch.stack(((a[:, 0] + a[:, 2]) * 0.5, (a[:, 1] + a[:, 3]) * 0.5), -1))
        return self._anchor_cache[key]

    def get_points(self, featmap_sizes, dtype, device, flatten=False):
        return [self._anchors(h, w, device)[1] for (h, w) in featmap_sizes]

    def forward(self, feats, branches='both'):
        """`branches` (TS_P2BFCOSHead.forward) is accepted and ignored: this student always evaluates both towers."""
        outs = [self.forward_single(x) for x in feats]
        return tuple(map(list, zip(*outs)))

    def forward_single(self, x):
        """-> (cls [B,C,H,W*A], distances [B,4,H,W*A], constant centerness logits [B,1,H,W*A], points [H*W*A,2]): the
        layout `TS_P2BFCOSHead._flat` flattens to point-major rows in (y, x, anchor) order."""
        B, _, H, W = x.shape
        A, C = self.num_anchors, self.cls_out_channels
        cls_feat = reg_feat = x
        for l in self.cls_convs:
            cls_feat = l(cls_feat)
        for l in self.reg_convs:
            reg_feat = l(reg_feat)
        cls = self.retina_cls(cls_feat).float().view(B, A, C, H, W).permute(0, 2, 3, 4, 1).reshape(B, C, H, W * A)
        deltas = self.retina_reg(reg_feat).float().view(B, A, 4, H, W).permute(0, 3, 4, 1, 2).reshape(B, H * W * A, 4)
        anchors, centres = self._anchors(H, W, x.device)
        boxes = self._decode(anchors, deltas)                                    # [B, P, 4]
        dist = torch.stack((centres[:, 0] - boxes[..., 0], centres[:, 1] - boxes[..., 1],
                            boxes[..., 2] - centres[:, 0], boxes[..., 3] - centres[:, 1]), -1)
        dist = dist.view(B, H, W * A, 4).permute(0, 3, 1, 2)
        ctr = x.new_full((B, 1, H, W * A), 20.0, dtype=torch.float32)            # sigmoid = 1: no centerness in this head
        return cls, dist, ctr, centres

    def _decode(self, anchors, deltas):
        """delta_xywh_bbox_coder.py:144-260 (means 0, stds 1, no border clipping at training time)."""
        import numpy as np
        px, py = (anchors[:, 0] + anchors[:, 2]) * 0.5, (anchors[:, 1] + anchors[:, 3]) * 0.5
        pw, ph = anchors[:, 2] - anchors[:, 0], anchors[:, 3] - anchors[:, 1]
        mr = abs(float(np.log(self.wh_ratio_clip)))
        dw, dh = deltas[..., 2].clamp(-mr, mr), deltas[..., 3].clamp(-mr, mr)
        gx, gy = px + pw * deltas[..., 0], py + ph * deltas[..., 1]
        gw, gh = pw * dw.exp(), ph * dh.exp()
        return torch.stack((gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5), -1)

    # --------------------------------------------------------------------------- losses --
    def _reg_branch(self, points, B, fr, ft, gi_reg, boxes, box_labels, off, loss_mod, dn, cls_pos=None):
        """The regression tail of `loss_pseudo` / `loss` for the anchor-based student: same assignment and targets as the
        parent (pt_fcos_targets), L1 between the encoded deltas of the prediction and of the target box w.r.t. the anchor
        of every positive (anchor_head.py loss_single with `reg_decoded_bbox=False`), averaged over the positives."""
        P = points.shape[0]
        labels_reg, tg, _ = F.fcos_targets(points, gi_reg, boxes, box_labels, off, B, self.num_classes)
        pos = labels_reg < self.num_classes
        norms = reduce_mean_many(pos.sum(), *([cls_pos] if cls_pos is not None else []))
        num_pos = norms[0].clamp(min=1.0)
        pts = points.repeat(B, 1)
        pred = distance2bbox(pts, fr.reshape(-1, 4))
        tgt = distance2bbox(pts, tg)
        anchors = None
        for (a, c) in self._anchor_cache.values():
            if c.shape[0] == P and c.device == points.device:
                anchors = a
        assert anchors is not None
        anc = anchors.repeat(B, 1)
        safe = torch.where(pos[:, None], tgt, anc)                              # rows without a target encode to zero deltas
        d_pred, d_tgt = bbox2delta(anc, pred), bbox2delta(anc, safe)
        l1 = torch.abs(d_pred - d_tgt).sum(-1)
        loss_bbox = self.loss_bbox_l1_weight * torch.where(pos, l1, torch.zeros_like(l1)).sum() / num_pos
        loss_ctr = ft.sum() * 0.0
        if cls_pos is not None:
            return loss_bbox, loss_ctr, norms[1].clamp(min=1.0)
        return loss_bbox, loss_ctr
