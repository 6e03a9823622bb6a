"""CPU ORACLE of the RetinaNet baseline's anchors, assignment, targets and loss - TEST INFRASTRUCTURE ONLY.

Restates, in plain torch fp32 on the CPU: core/anchor/anchor_generator.py (gen_single_level_base_anchors :160-196,
single_level_grid_anchors :398-441, valid_flags :443-497), core/bbox/assigners/max_iou_assigner.py (assign_wrt_overlaps
:130-212 over BboxOverlaps2D), core/bbox/coder/delta_xywh_bbox_coder.py bbox2delta :98-140 and
models/dense_heads/anchor_head.py (_get_targets_single :181-278, get_targets :280-380, loss_single / loss :382-500) of
/root/reference/HBB_TOD/mmdet.  PINNED by tests/golden/retina_baseline.npz (outputs of the reference's own classes,
oracle/gen_golden_retina.py); only tests import this file."""
import numpy as np
import torch
import torch.nn.functional as F


def base_anchors(base_size, scales, ratios):
    scales, ratios = torch.Tensor(scales), torch.Tensor(ratios)
    h_ratios = torch.sqrt(ratios)
    w_ratios = 1 / h_ratios
    ws = (base_size * w_ratios[:, None] * scales[None, :]).view(-1)
    hs = (base_size * h_ratios[:, None] * scales[None, :]).view(-1)
    return torch.stack([0 - 0.5 * ws, 0 - 0.5 * hs, 0 + 0.5 * ws, 0 + 0.5 * hs], dim=-1)


def octave_scales(octave_base_scale=4, scales_per_octave=3):
    return np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)]) * octave_base_scale


def grid_anchors(featmap_sizes, strides, scales, ratios):
    out = []
    for (fh, fw), s in zip(featmap_sizes, strides):
        base = base_anchors(s, scales, ratios)
        sx, sy = torch.arange(0, fw) * s, torch.arange(0, fh) * s
        xx, yy = sx.repeat(fh), sy.view(-1, 1).repeat(1, fw).view(-1)
        shifts = torch.stack([xx, yy, xx, yy], dim=-1).type_as(base)
        out.append((base[None] + shifts[:, None]).view(-1, 4))
    return out


def valid_flags(featmap_sizes, strides, pad_hw, num_base):
    out = []
    for (fh, fw), s in zip(featmap_sizes, strides):
        vh, vw = min(int(np.ceil(pad_hw[0] / s)), fh), min(int(np.ceil(pad_hw[1] / s)), fw)
        vx, vy = torch.zeros(fw, dtype=torch.bool), torch.zeros(fh, dtype=torch.bool)
        vx[:vw], vy[:vh] = 1, 1
        v = vx.repeat(fh) & vy.view(-1, 1).repeat(1, fw).view(-1)
        out.append(v[:, None].expand(v.size(0), num_base).contiguous().view(-1))
    return out


def bbox_overlaps(b1, b2, eps=1e-6):
    """iou2d_calculator.py bbox_overlaps(mode='iou'), rows b1, columns b2."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt, rb = torch.max(b1[:, None, :2], b2[None, :, :2]), torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    ov = wh[..., 0] * wh[..., 1]
    union = torch.max(a1[:, None] + a2[None, :] - ov, ov.new_tensor([eps]))
    return ov / union


def max_iou_assign(anchors, gts, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, gt_max_assign_all=True, match_low_quality=True):
    A, G = anchors.shape[0], gts.shape[0]
    assigned = torch.full((A,), -1, dtype=torch.long)
    if G == 0:
        assigned[:] = 0
        return assigned, torch.zeros(A)
    ov = bbox_overlaps(gts, anchors)
    mo, am = ov.max(dim=0)
    gmo, gam = ov.max(dim=1)
    if isinstance(neg_iou_thr, float):
        assigned[(mo >= 0) & (mo < neg_iou_thr)] = 0
    else:
        assigned[(mo >= neg_iou_thr[0]) & (mo < neg_iou_thr[1])] = 0
    pos = mo >= pos_iou_thr
    assigned[pos] = am[pos] + 1
    if match_low_quality:
        for i in range(G):
            if gmo[i] >= min_pos_iou:
                if gt_max_assign_all:
                    assigned[ov[i] == gmo[i]] = i + 1
                else:
                    assigned[gam[i]] = i + 1
    return assigned, mo


def bbox2delta(p, g):
    px, py, pw, ph = (p[:, 0] + p[:, 2]) * 0.5, (p[:, 1] + p[:, 3]) * 0.5, p[:, 2] - p[:, 0], p[:, 3] - p[:, 1]
    gx, gy, gw, gh = (g[:, 0] + g[:, 2]) * 0.5, (g[:, 1] + g[:, 3]) * 0.5, g[:, 2] - g[:, 0], g[:, 3] - g[:, 1]
    return torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)


def targets(anchors, flags, gt_bboxes, gt_labels, num_classes, assigner_kw):
    """Per image: (labels [A], label_weights [A], bbox_targets [A,4], bbox_weights [A,4], num_pos, num_neg); `flags` = valid."""
    A = anchors.shape[0]
    inside = flags
    an = anchors[inside]
    asg, _ = max_iou_assign(an, gt_bboxes, **assigner_kw)
    n = an.shape[0]
    bt, bw = torch.zeros_like(an), torch.zeros_like(an)
    lab = torch.full((n,), num_classes, dtype=torch.long)
    lw = torch.zeros(n)
    pos, neg = (asg > 0).nonzero().reshape(-1), (asg == 0).nonzero().reshape(-1)
    if len(pos):
        bt[pos] = bbox2delta(an[pos], gt_bboxes[asg[pos] - 1])
        bw[pos] = 1.0
        lab[pos] = gt_labels[asg[pos] - 1]
        lw[pos] = 1.0
    lw[neg] = 1.0

    def unmap(d, fill=0):
        r = d.new_full((A,) + d.shape[1:], fill)
        r[inside] = d
        return r
    return unmap(lab, num_classes), unmap(lw), unmap(bt), unmap(bw), len(pos), len(neg)


def focal_sum(pred, target, weight, num_classes, gamma=2.0, alpha=0.25):
    onehot = F.one_hot(target, num_classes + 1)[:, :num_classes].type_as(pred)
    p = pred.sigmoid()
    pt = (1 - p) * onehot + p * (1 - onehot)
    w = (alpha * onehot + (1 - alpha) * (1 - onehot)) * pt.pow(gamma)
    return (F.binary_cross_entropy_with_logits(pred, onehot, reduction='none') * w * weight[:, None]).sum()


def loss(cls_scores, bbox_preds, gt_bboxes, gt_labels, strides, pad_hw, num_classes, assigner_kw, scales=None, ratios=(0.5, 1.0, 2.0)):
    scales = octave_scales() if scales is None else scales
    sizes = [c.shape[-2:] for c in cls_scores]
    lvl = grid_anchors(sizes, strides, scales, ratios)
    nb = len(scales) * len(ratios)
    flat, flags = torch.cat(lvl), torch.cat(valid_flags(sizes, strides, pad_hw, nb))
    per = [targets(flat, flags, b, l, num_classes, assigner_kw) for b, l in zip(gt_bboxes, gt_labels)]
    num_total_pos = sum(max(p[4], 1) for p in per)
    n = [a.shape[0] for a in lvl]
    out_c, out_b, lv_t = [], [], []
    o = 0
    for i, (cs, bp) in enumerate(zip(cls_scores, bbox_preds)):
        lab = torch.stack([p[0][o:o + n[i]] for p in per]).reshape(-1)
        lw = torch.stack([p[1][o:o + n[i]] for p in per]).reshape(-1)
        bt = torch.stack([p[2][o:o + n[i]] for p in per]).reshape(-1, 4)
        bw = torch.stack([p[3][o:o + n[i]] for p in per]).reshape(-1, 4)
        cs = cs.permute(0, 2, 3, 1).reshape(-1, num_classes)
        bp = bp.permute(0, 2, 3, 1).reshape(-1, 4)
        out_c.append(focal_sum(cs, lab, lw, num_classes) / num_total_pos)
        out_b.append((torch.abs(bp - bt) * bw).sum() / num_total_pos)
        lv_t.append((lab, lw, bt, bw))
        o += n[i]
    return dict(loss_cls=out_c, loss_bbox=out_b), lv_t, num_total_pos
