"""CPU ORACLE of the supervised FCOS baseline's target assignment and loss - TEST INFRASTRUCTURE ONLY.

Restates /root/reference/HBB_TOD/mmdet/models/dense_heads/fcos_head.py: get_targets :698-875, _get_target_single :877-1007,
centerness_target :1009-1031, loss :254-473, in plain torch fp32 on the CPU.  PINNED: tests/golden/fcos_baseline.npz holds
what the reference's own methods return on seeded inputs (oracle/gen_golden_fcos.py); tests/test_fcos_baseline.py checks
this file against it.  Only tests import this file."""
import torch
import torch.nn.functional as F

INF = 1e8


def get_points(featmap_sizes, strides):
    """anchor_free_head.py:318-335 + fcos_head.py:686-696: (x*s + s//2, y*s + s//2), row-major."""
    out = []
    for (h, w), s in zip(featmap_sizes, strides):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
        out.append(torch.stack((xs.reshape(-1) * s, ys.reshape(-1) * s), dim=-1) + s // 2)
    return out


def target_single(gt_bboxes, gt_labels, points, ranges, radius, num_classes, center_sampling):
    """One image.  points [P,2], ranges [P,2], radius [P] (= stride * center_sample_radius)."""
    P, G = points.shape[0], gt_labels.shape[0]
    if G == 0:
        return gt_labels.new_full((P,), num_classes), gt_bboxes.new_zeros((P, 4))
    areas = ((gt_bboxes[:, 2] - gt_bboxes[:, 0]) * (gt_bboxes[:, 3] - gt_bboxes[:, 1]))[None].repeat(P, 1)
    xs, ys = points[:, 0:1].expand(P, G), points[:, 1:2].expand(P, G)
    g = gt_bboxes[None].expand(P, G, 4)
    tg = torch.stack((xs - g[..., 0], ys - g[..., 1], g[..., 2] - xs, g[..., 3] - ys), -1)
    if center_sampling:
        cx, cy = (g[..., 0] + g[..., 2]) / 2, (g[..., 1] + g[..., 3]) / 2
        st = radius[:, None].expand(P, G)
        x0 = torch.where(cx - st > g[..., 0], cx - st, g[..., 0])
        y0 = torch.where(cy - st > g[..., 1], cy - st, g[..., 1])
        x1 = torch.where(cx + st > g[..., 2], g[..., 2], cx + st)
        y1 = torch.where(cy + st > g[..., 3], g[..., 3], cy + st)
        inside = torch.stack((xs - x0, ys - y0, x1 - xs, y1 - ys), -1).min(-1)[0] > 0
    else:
        inside = tg.min(-1)[0] > 0
    maxd = tg.max(-1)[0]
    in_range = (maxd >= ranges[:, 0:1]) & (maxd <= ranges[:, 1:2])
    areas = areas.clone()
    areas[~inside] = INF
    areas[~in_range] = INF
    min_area, idx = areas.min(dim=1)
    labels = gt_labels[idx].clone()
    labels[min_area == INF] = num_classes
    return labels, tg[torch.arange(P), idx]


def get_targets(points_per_level, gt_bboxes, gt_labels, regress_ranges, strides, num_classes, center_sampling=True,
                center_sample_radius=1.5, norm_on_bbox=True):
    """-> per-level lists (labels [B*n_l], bbox_targets [B*n_l, 4]) like the reference (images concatenated inside a level)."""
    pts = torch.cat(points_per_level)
    n = [p.shape[0] for p in points_per_level]
    rr = torch.cat([pts.new_tensor(r)[None].expand(k, 2) for r, k in zip(regress_ranges, n)])
    rad = torch.cat([pts.new_full((k,), s * center_sample_radius) for s, k in zip(strides, n)])
    per_img = [target_single(b, l, pts, rr, rad, num_classes, center_sampling) for b, l in zip(gt_bboxes, gt_labels)]
    labels = [torch.cat([li.split(n, 0)[i] for li, _ in per_img]) for i in range(len(n))]
    targets = []
    for i in range(len(n)):
        t = torch.cat([ti.split(n, 0)[i] for _, ti in per_img])
        targets.append(t / strides[i] if norm_on_bbox else t)
    return labels, targets


def centerness_target(t):
    lr, tb = t[:, [0, 2]], t[:, [1, 3]]
    if len(lr) == 0:
        return lr[..., 0]
    return torch.sqrt((lr.min(dim=-1)[0] / lr.max(dim=-1)[0]) * (tb.min(dim=-1)[0] / tb.max(dim=-1)[0]))


def sigmoid_focal_loss_sum(pred, target, num_classes, gamma=2.0, alpha=0.25):
    """focal_loss.py:11-56 py_sigmoid_focal_loss, summed."""
    onehot = F.one_hot(target, num_classes + 1)[:, :num_classes].type_as(pred)
    p = pred.sigmoid()
    pt = (1 - p) * onehot + p * (1 - onehot)
    w = (alpha * onehot + (1 - alpha) * (1 - onehot)) * pt.pow(gamma)
    return (F.binary_cross_entropy_with_logits(pred, onehot, reduction='none') * w).sum()


def diou_loss(pred, target, eps=1e-7):
    """iou_loss.py:139-189"""
    lt, rb = torch.max(pred[:, :2], target[:, :2]), torch.min(pred[:, 2:], target[:, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[:, 0] * wh[:, 1]
    ap = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    ag = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    ious = overlap / (ap + ag - overlap + eps)
    e1, e2 = torch.min(pred[:, :2], target[:, :2]), torch.max(pred[:, 2:], target[:, 2:])
    ewh = (e2 - e1).clamp(min=0)
    c2 = ewh[:, 0] ** 2 + ewh[:, 1] ** 2 + eps
    left = ((target[:, 0] + target[:, 2]) - (pred[:, 0] + pred[:, 2])) ** 2 / 4
    right = ((target[:, 1] + target[:, 3]) - (pred[:, 1] + pred[:, 3])) ** 2 / 4
    return 1 - (ious - (left + right) / c2)


def loss(cls_scores, bbox_preds, centernesses, gt_bboxes, gt_labels, strides, regress_ranges, num_classes,
         center_sampling=True, center_sample_radius=1.5, norm_on_bbox=True):
    """fcos_head.py:254-473 (single process: reduce_mean is the identity)."""
    sizes = [c.shape[-2:] for c in cls_scores]
    pts = get_points(sizes, strides)
    labels, targets = get_targets(pts, gt_bboxes, gt_labels, regress_ranges, strides, num_classes, center_sampling,
                                  center_sample_radius, norm_on_bbox)
    B = cls_scores[0].shape[0]
    fc = torch.cat([c.permute(0, 2, 3, 1).reshape(-1, num_classes) for c in cls_scores])
    fr = torch.cat([b.permute(0, 2, 3, 1).reshape(-1, 4) for b in bbox_preds])
    ft = torch.cat([c.permute(0, 2, 3, 1).reshape(-1) for c in centernesses])
    fl, fb = torch.cat(labels), torch.cat(targets)
    fp = torch.cat([p.repeat(B, 1) for p in pts])
    pos = ((fl >= 0) & (fl < num_classes)).nonzero().reshape(-1)
    num_pos = max(float(len(pos)), 1.0)
    loss_cls = sigmoid_focal_loss_sum(fc, fl, num_classes) / num_pos
    pt_, pc, ptg = fr[pos], ft[pos], fb[pos]
    ctr_t = centerness_target(ptg)
    denorm = max(float(ctr_t.sum()), 1e-6)
    if len(pos) > 0:
        def d2b(p, d):
            return torch.stack([p[:, 0] - d[:, 0], p[:, 1] - d[:, 1], p[:, 0] + d[:, 2], p[:, 1] + d[:, 3]], -1)
        loss_bbox = (diou_loss(d2b(fp[pos], pt_), d2b(fp[pos], ptg)) * ctr_t).sum() / denorm
        loss_ctr = F.binary_cross_entropy_with_logits(pc, ctr_t, reduction='sum') / num_pos
    else:
        loss_bbox, loss_ctr = pt_.sum(), pc.sum()
    return dict(loss_cls=loss_cls, loss_bbox=loss_bbox, loss_centerness=loss_ctr), labels, targets
