"""CPU ORACLE for the Point-Teacher hot path - TEST INFRASTRUCTURE ONLY.

A plain numpy / torch-CPU fp32 restatement of the reference's algorithm for every
function in SURVEY.md section 8(a).  It exists to CHECK the HIP path:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker - never as the thing measured or
shipped.  The product (``point_teacher_amd``) never imports this package and has
no CPU fallback.

Pinning: every function below is checked against golden vectors captured from the
reference itself (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``), see
``tests/test_oracle_vs_golden.py``.  Exceptions - "parity unpinned" because the
arithmetic lives in mmcv-full (un-vendored, version range 1.3.2-1.7.2, absent
here): ``roi_align`` (mmcv.ops.RoIAlign), ``nms`` (mmcv.ops.nms),
``box_iou_rotated`` / ``nms_rotated`` and the ``cv2.fillPoly`` rasteriser.  Those
follow the published algorithms and are validated by invariants in the tests.

Citations are ``file:line`` under /root/reference/HBB_TOD/mmdet/.
Integer / index results are bit-exact restatements; where the reference's
``torch.topk`` leaves tie order unspecified the oracle fixes it to the LOWEST
point index (stable sort) - the same rule the HIP kernels implement.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# box codecs                                   core/bbox/transforms.py
# ----------------------------------------------------------------------------


def bbox_xyxy_to_cxcywh(b):
    """transforms.py:250-262"""
    x1, y1, x2, y2 = b.unbind(-1)
    return torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], -1)


def bbox_cxcywh_to_xyxy(b):
    """transforms.py:236-247"""
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)


def distance2bbox(points, distance, max_shape=None):
    """transforms.py:134-177.  max_shape=(H, W[, C]) clips x to [0,W], y to [0,H]."""
    x1 = points[..., 0] - distance[..., 0]
    y1 = points[..., 1] - distance[..., 1]
    x2 = points[..., 0] + distance[..., 2]
    y2 = points[..., 1] + distance[..., 3]
    b = torch.stack([x1, y1, x2, y2], -1)
    if max_shape is not None:
        h, w = float(max_shape[0]), float(max_shape[1])
        mx = b.new_tensor([w, h, w, h])
        b = torch.where(b < 0, b.new_zeros(()), b)
        b = torch.where(b > mx, mx.expand_as(b), b)
    return b


def delta2bbox(rois, deltas, max_shape=None, wh_ratio_clip=16 / 1000):
    """coder/delta_xywh_bbox_coder.py:144-270 with means 0 / stds 1 (the MIL decoder,
    fcos_head_p2b_ts.py:182-184) and clip_border=True."""
    dx, dy, dw, dh = deltas.unbind(-1)
    x1, y1, x2, y2 = rois.unbind(-1)
    px = (x1 + x2) * 0.5
    py = (y1 + y2) * 0.5
    pw = x2 - x1
    ph = y2 - y1
    max_ratio = float(np.abs(np.log(wh_ratio_clip)))
    dw = dw.clamp(min=-max_ratio, max=max_ratio)
    dh = dh.clamp(min=-max_ratio, max=max_ratio)
    gw = pw * dw.exp()
    gh = ph * dh.exp()
    gx = px + pw * dx
    gy = py + ph * dy
    b = torch.stack([gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5], -1)
    if max_shape is not None:
        h, w = float(max_shape[0]), float(max_shape[1])
        mx = b.new_tensor([w, h, w, h])
        b = torch.where(b < 0, b.new_zeros(()), b)
        b = torch.where(b > mx, mx.expand_as(b), b)
    return b


def bbox2roi(bbox_list):
    """transforms.py:58-78"""
    out = []
    for i, b in enumerate(bbox_list):
        if b.shape[0] > 0:
            out.append(torch.cat([b.new_full((b.shape[0], 1), i), b[:, :4]], -1))
        else:
            out.append(b.new_zeros((0, 5)))
    return torch.cat(out, 0)


def bbox_overlaps(b1, b2, mode='iou', is_aligned=False, eps=1e-6):
    """core/bbox/iou_calculators/iou2d_calculator.py:74-260"""
    rows, cols = b1.shape[0], b2.shape[0]
    if rows * cols == 0:
        return b1.new_zeros((rows,)) if is_aligned else b1.new_zeros((rows, cols))
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    if is_aligned:
        lt = torch.max(b1[:, :2], b2[:, :2])
        rb = torch.min(b1[:, 2:], b2[:, 2:])
        wh = (rb - lt).clamp(min=0)
        ov = wh[:, 0] * wh[:, 1]
        union = a1 + a2 - ov if mode in ('iou', 'giou') else a1
        if mode == 'giou':
            elt = torch.min(b1[:, :2], b2[:, :2])
            erb = torch.max(b1[:, 2:], b2[:, 2:])
    else:
        lt = torch.max(b1[:, None, :2], b2[None, :, :2])
        rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        ov = wh[..., 0] * wh[..., 1]
        union = a1[:, None] + a2[None, :] - ov if mode in ('iou', 'giou') else a1[:, None].expand_as(ov)
        if mode == 'giou':
            elt = torch.min(b1[:, None, :2], b2[None, :, :2])
            erb = torch.max(b1[:, None, 2:], b2[None, :, 2:])
    e = union.new_tensor(eps)
    union = torch.max(union, e)
    ious = ov / union
    if mode in ('iou', 'iof'):
        return ious
    ewh = (erb - elt).clamp(min=0)
    ea = torch.max(ewh[..., 0] * ewh[..., 1], e)
    return ious - (ea - union) / ea


def obb2poly_le90(r):
    """detectors/data_augument_bank.py:516-541: corners in the order
    (tl, tr, br, bl) of the un-rotated rectangle, rotated by +angle."""
    if r.shape[0] == 0:
        return r.new_zeros((0, 8))
    cx, cy, w, h, a = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4]
    xs = torch.stack([-w * .5, w * .5, w * .5, -w * .5], 1)
    ys = torch.stack([-h * .5, -h * .5, h * .5, h * .5], 1)
    s, c = torch.sin(a)[:, None], torch.cos(a)[:, None]
    px = c * xs - s * ys + cx[:, None]
    py = s * xs + c * ys + cy[:, None]
    return torch.stack([px, py], -1).reshape(-1, 8)


def obb2xyxy(o):
    """detectors/syn_images_generator_v2.py:382-396"""
    w, h, a = o[:, 2], o[:, 3], o[:, 4]
    ca, sa = torch.cos(a).abs(), torch.sin(a).abs()
    dw, dh = ca * w + sa * h, sa * w + ca * h
    return torch.stack([o[:, 0] - dw / 2, o[:, 1] - dh / 2, o[:, 0] + dw / 2, o[:, 1] + dh / 2], -1)


def random_point_in_box(boxes, position, ux, uy):
    """data_augument_bank.py:1471-1493 with the two torch.rand draws injected."""
    w = boxes[:, 2] - boxes[:, 0]
    h = boxes[:, 3] - boxes[:, 1]
    x = boxes[:, 0] + w * (1 - position) / 2 + ux * (w * position)
    y = boxes[:, 1] + h * (1 - position) / 2 + uy * (h * position)
    return torch.stack([x, y], 1)


# ----------------------------------------------------------------------------
# match costs                         core/bbox/match_costs/match_cost.py
# ----------------------------------------------------------------------------


def point_cost(a, b, mode='L1', weight=1.0):
    """match_cost.py:188-214"""
    d = a[:, None, :2] - b[None, :, :2]
    if mode == 'L1':
        return d.abs().sum(-1) * weight
    return torch.sqrt((d ** 2).sum(-1)) * weight


def focal_loss_cost(cls_pred, gt_labels, weight=1.0, alpha=0.25, gamma=2, eps=1e-12):
    """match_cost.py:54-99"""
    p = cls_pred.sigmoid()
    neg = -(1 - p + eps).log() * (1 - alpha) * p.pow(gamma)
    pos = -(p + eps).log() * alpha * (1 - p).pow(gamma)
    return (pos[:, gt_labels] - neg[:, gt_labels]) * weight


def insider_cost(boxes_cxcywh, gt, weight=1.0):
    """match_cost.py:217-252: 0 where the gt POINT lies inside the predicted box."""
    x1 = boxes_cxcywh[:, 0] - boxes_cxcywh[:, 2] / 2
    y1 = boxes_cxcywh[:, 1] - boxes_cxcywh[:, 3] / 2
    x2 = boxes_cxcywh[:, 0] + boxes_cxcywh[:, 2] / 2
    y2 = boxes_cxcywh[:, 1] + boxes_cxcywh[:, 3] / 2
    gx, gy = gt[None, :, 0], gt[None, :, 1]
    inside = (gx >= x1[:, None]) & (gx <= x2[:, None]) & (gy >= y1[:, None]) & (gy <= y2[:, None])
    return (~inside).to(boxes_cxcywh.dtype) * weight


# ----------------------------------------------------------------------------
# assigners            core/bbox/assigners/{topk,fuse_topk}_assigner.py
# ----------------------------------------------------------------------------


def _stable_smallest(cost, k):
    """Indices of the k smallest entries along axis 0, ties -> lowest index."""
    return np.argsort(cost, axis=0, kind='stable')[:k]


def topk_assign(points, gt, gt_labels, num_pre, topk=None, cls_cost=None):
    """topk_assigner.py:54-147.  ``points`` [P,>=2], ``gt`` [G,>=2] (only the first
    two columns are read by PointCost).  Returns (gt_inds[P] int64 0=bg / i+1,
    labels[P] int64 -1=unassigned).  With num_pre <= topk (every shipped config)
    the class cost is never consulted (:137-140)."""
    P, G = points.shape[0], gt.shape[0]
    gt_inds = np.zeros(P, np.int64)
    labels = np.full(P, -1, np.int64)
    if G == 0 or P == 0:
        return torch.from_numpy(gt_inds), torch.from_numpy(labels)
    d = point_cost(points.float(), gt.float()).numpy()
    cand = _stable_smallest(d, num_pre)                    # [num_pre, G]
    gl = gt_labels.numpy()
    topk = num_pre if topk is None else topk
    for i in range(G):                                     # sequential: later gt wins
        rows = cand[:, i]
        if num_pre > topk:
            ci = cls_cost.numpy()[rows, :]                 # [num_pre, G]
            sel = np.unique(_stable_smallest(ci, topk).reshape(-1))
            rows = rows[sel]
        gt_inds[rows] = i + 1
        labels[rows] = gl[i]
    return torch.from_numpy(gt_inds), torch.from_numpy(labels)


def fuse_topk_assign(bbox_cxcywh, points, cls_pred, gt_points, gt_labels, num_pre=5, topk=3,
                     cls_weight=1.0, reg_weight=1.0, loc_weight=1.0):
    """fuse_topk_assigner.py:56-121.  Stage 1: per gt the num_pre L1-nearest grid
    points.  Stage 2 (:104-118): for gt i take the [num_pre, G] slab of
    (focal + insider) cost, ``topk`` smallest rows of EVERY column, and assign the
    union.  Sequential overwrite, later gt wins."""
    P, G = points.shape[0], gt_points.shape[0]
    gt_inds = np.zeros(P, np.int64)
    labels = np.full(P, -1, np.int64)
    if G == 0 or P == 0:
        return torch.from_numpy(gt_inds), torch.from_numpy(labels)
    d = point_cost(points.float(), gt_points.float(), weight=reg_weight).numpy()
    cand = _stable_smallest(d, num_pre)
    cost = (focal_loss_cost(cls_pred, gt_labels, cls_weight)
            + insider_cost(bbox_cxcywh, gt_points, loc_weight)).numpy()
    gl = gt_labels.numpy()
    for i in range(G):
        rows = cand[:, i]
        if num_pre <= topk:
            sel = np.arange(num_pre)
        else:
            sel = np.unique(_stable_smallest(cost[rows, :], topk).reshape(-1))
        gt_inds[rows[sel]] = i + 1
        labels[rows[sel]] = gl[i]
    return torch.from_numpy(gt_inds), torch.from_numpy(labels)


# ----------------------------------------------------------------------------
# FCOS targets / pseudo boxes             dense_heads/fcos_head_p2b_ts.py
# ----------------------------------------------------------------------------


def fcos_ltrb_targets(points, gt_inds, boxes):
    """fcos_head_p2b_ts.py:589-603 / :692-706: (l,t,r,b) of the assigned box;
    unassigned points use box 0 (``inds * 0``)."""
    idx = torch.where(gt_inds > 0, gt_inds - 1, torch.zeros_like(gt_inds))
    b = boxes[idx]
    xs, ys = points[:, 0], points[:, 1]
    return torch.stack([xs - b[:, 0], ys - b[:, 1], b[:, 2] - xs, b[:, 3] - ys], -1)


def centerness_target(t):
    """fcos_head_p2b_ts.py:1019-1038 (only the MIN side is clamped at 0.01)."""
    lr = t[:, [0, 2]]
    tb = t[:, [1, 3]]
    if t.shape[0] == 0:
        return lr[..., 0]
    c = (lr.min(-1)[0].clamp(min=0.01) / lr.max(-1)[0]) * (tb.min(-1)[0].clamp(min=0.01) / tb.max(-1)[0])
    return torch.sqrt(c)


def get_target_pseudo_single(points, cls, gt_points, gt_labels, pseudo_bboxes, pseudo_labels, num_classes=8):
    """fcos_head_p2b_ts.py:658-708 with assigner (1,1) and pseudo_assigner (3,3)."""
    P = points.shape[0]
    gi, gl = topk_assign(points, gt_points, gt_labels, 1)
    labels = torch.full((P,), num_classes, dtype=torch.long)
    labels[gi != 0] = gl[gi != 0]
    weights = torch.ones(P)
    if pseudo_bboxes.shape[0] == 0:
        return torch.full((P,), num_classes, dtype=torch.long), points.new_zeros((P, 4)), labels, weights
    gi, gl = topk_assign(points, bbox_xyxy_to_cxcywh(pseudo_bboxes), pseudo_labels, 3)
    labels_reg = torch.full((P,), num_classes, dtype=torch.long)
    labels_reg[gi != 0] = gl[gi != 0]
    return labels_reg, fcos_ltrb_targets(points, gi, pseudo_bboxes), labels, weights


def get_target_single(points, gt_bboxes, num_classes=8):
    """fcos_head_p2b_ts.py:570-605 with syn_assigner (3,3); all labels are 0."""
    P = points.shape[0]
    if gt_bboxes.shape[0] == 0:
        return torch.full((P,), num_classes, dtype=torch.long), points.new_zeros((P, 4))
    gl0 = torch.zeros(gt_bboxes.shape[0], dtype=torch.long)
    gi, gl = topk_assign(points, bbox_xyxy_to_cxcywh(gt_bboxes), gl0, 3)
    labels = torch.full((P,), num_classes, dtype=torch.long)
    labels[gi != 0] = gl[gi != 0]
    return labels, fcos_ltrb_targets(points, gi, gt_bboxes)


def generate_pseudo_single(points, cls, reg, gt_points, gt_labels, gt_bboxes, filter_score=0.0):
    """fcos_head_p2b_ts.py:736-794.  Returns pseudo_bboxes[G,4], pseudo_points[G,2],
    pseudo_scores[G], mean_iou (scalar), valid_inds (sorted int64)."""
    G = gt_labels.shape[0]
    if G == 0:
        return (torch.empty(0, 4), torch.empty(0, 2), torch.empty(0), torch.tensor(0.0),
                torch.empty(0, dtype=torch.long))
    act = cls.sigmoid()
    dec = distance2bbox(points, reg)
    gi, gl = fuse_topk_assign(bbox_xyxy_to_cxcywh(dec), points, cls, gt_points, gt_labels)
    pos = (gi != 0).nonzero().reshape(-1)
    lab = torch.zeros(points.shape[0], dtype=torch.long)
    lab[pos] = gl[pos]
    score = act[torch.arange(points.shape[0]), lab]
    A, B, C = dec[pos], gi[pos] - 1, score[pos]
    nums = torch.bincount(B, minlength=G)
    pb = torch.cat([gt_points, gt_points.new_full((G, 2), 8.0)], 1)
    pb = bbox_cxcywh_to_xyxy(pb)
    pp = gt_points.clone()
    ps = torch.zeros(G)
    bsum = torch.zeros(G, 4).index_add_(0, B, A * C[:, None])
    ssum = torch.zeros(G).index_add_(0, B, C)
    nz = nums != 0
    pb[nz] = bsum[nz] / ssum[nz, None]
    ps[nz] = ssum[nz] / nums[nz]
    pp[nz] = bbox_xyxy_to_cxcywh(pb[nz])[:, :2]
    miou = bbox_overlaps(pb[nz], gt_bboxes[nz], is_aligned=True).mean()
    valid = (nz & (ps >= filter_score)).nonzero().reshape(-1)
    return pb, pp, ps, miou, valid


# ----------------------------------------------------------------------------
# losses (differentiable torch restatements)           models/losses/*.py
# ----------------------------------------------------------------------------


def sigmoid_focal_loss_elem(pred, labels, gamma=2.0, alpha=0.25):
    """focal_loss.py:11-56 (py_sigmoid_focal_loss) element-wise, integer labels with
    background = C (FocalLoss.forward :165-167 builds the one-hot)."""
    C = pred.shape[1]
    t = F.one_hot(labels, C + 1)[:, :C].type_as(pred)
    p = pred.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    fw = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    return F.binary_cross_entropy_with_logits(pred, t, reduction='none') * fw


def sigmoid_focal_loss(pred, labels, weight=None, gamma=2.0, alpha=0.25, avg_factor=None):
    """FocalLoss.forward (focal_loss.py:134-181) with reduction 'mean'."""
    l = sigmoid_focal_loss_elem(pred, labels, gamma, alpha)
    if weight is not None:
        l = l * weight.view(-1, 1)
    return l.mean() if avg_factor is None else l.sum() / avg_factor


def diou_loss_elem(pred, target, eps=1e-6):
    """iou_loss.py:139-189"""
    lt = torch.max(pred[:, :2], target[:, :2])
    rb = torch.min(pred[:, 2:], target[:, 2:])
    wh = (rb - lt).clamp(min=0)
    ov = wh[:, 0] * wh[:, 1]
    ap = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    ag = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    ious = ov / (ap + ag - ov + eps)
    e1 = torch.min(pred[:, :2], target[:, :2])
    e2 = torch.max(pred[:, 2:], target[:, 2:])
    ewh = (e2 - e1).clamp(min=0)
    c2 = ewh[:, 0] ** 2 + ewh[:, 1] ** 2 + eps
    rho2 = ((target[:, 0] + target[:, 2]) - (pred[:, 0] + pred[:, 2])) ** 2 / 4 + \
           ((target[:, 1] + target[:, 3]) - (pred[:, 1] + pred[:, 3])) ** 2 / 4
    return 1 - (ious - rho2 / c2)


def dn_diou_loss_elem(pred, targets, hyper=0.2, eps=1e-6):
    """iou_loss.py:398-465, including the quirk at :412: ``base_loss`` goes through
    @weighted_loss with the default reduction, i.e. it is the SCALAR MEAN of the
    plain DIoU over the batch, added to every element's min-over-9."""
    base = diou_loss_elem(pred, targets, eps).mean()
    anx = hyper / 2
    w = targets[:, 2] - targets[:, 0]
    h = targets[:, 3] - targets[:, 1]
    bank = []
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            t = torch.stack([targets[:, 0] - anx * w * i, targets[:, 1] - anx * h * i,
                             targets[:, 2] + anx * w * j, targets[:, 3] + anx * h * j], 1)
            bank.append(diou_loss_elem(pred, t, eps))
    return (base + torch.stack(bank, 1).min(1)[0]) / 2


def iou_loss_module(pred, target, weight=None, avg_factor=None, kind='diou', hyper=0.2, eps=1e-6,
                    loss_weight=1.0):
    """DIoULoss.forward / DN_DIoULoss.forward (iou_loss.py:803-880) incl. the
    zero-weight shortcut (:858-861)."""
    if weight is not None and not torch.any(weight > 0):
        return (pred * weight.unsqueeze(1)).sum()
    l = diou_loss_elem(pred, target, eps) if kind == 'diou' else dn_diou_loss_elem(pred, target, hyper, eps)
    if weight is not None:
        l = l * weight
    return loss_weight * (l.mean() if avg_factor is None else l.sum() / avg_factor)


def bce_logits_loss(x, t, avg_factor=None):
    """cross_entropy_loss.py:62-100 on 1-D logits with float targets."""
    l = F.binary_cross_entropy_with_logits(x, t.float(), reduction='none')
    return l.mean() if avg_factor is None else l.sum() / avg_factor


def gfocal(p, q, w, eps=1e-6):
    """fcos_head_p2b_ts.py:1074-1078"""
    return -(((p - q) ** 2) * (q * (p + eps).log() + (1 - q) * (1 - p + eps).log()) * w).sum(-1)


def mil_bag_loss(cls, ins, valid, labels, neg_cls=None, neg_w=None):
    """fcos_head_p2b_ts.py:1147-1180.  cls/ins [N,U1,U2,C]; valid [N*U1*U2,1] bool;
    labels [N]; neg_cls [M,C]; neg_w [M] bool."""
    N, U1, U2, C = cls.shape
    v = valid.reshape(N, U1, U2, 1).to(cls.dtype)
    lab = labels[:, None].repeat(1, U1).reshape(-1)
    c = cls.sigmoid()
    i = F.normalize(ins.softmax(2) * v, dim=2, p=1)
    bag = (c * i).sum(2).reshape(-1, C)
    lw = (v.reshape(N * U1, U2, 1).sum(1) > 0).float()
    ns = max(float((lw.sum(-1) > 0).sum()), 1.0)
    q = F.one_hot(lab, C).float()
    loss = gfocal(bag, q, lw).sum() / ns
    if neg_cls is not None:
        loss = loss + gfocal(neg_cls.sigmoid(), torch.zeros_like(neg_cls),
                             neg_w.reshape(-1, 1).float()).sum() / ns
    return loss


def mil_bag_select(cls, ins, valid, labels, bags, pseudo, img_hw, topk=1, beta=0.25):
    """fcos_head_p2b_ts.py:1112-1145 + :1092-1110 (all images share img_hw)."""
    N, U1, U2, C = cls.shape
    v = valid.reshape(N, U1, U2, 1).to(cls.dtype)
    c = cls.detach().reshape(N, U1 * U2, C).sigmoid()
    i = F.normalize(ins.detach().softmax(2) * v, dim=2, p=1).reshape(N, U1 * U2, C)
    ar = torch.arange(N)
    s = c[ar, :, labels] * i[ar, :, labels]
    sc, idx = s.topk(topk, dim=1)
    w = sc / (sc.sum(1, keepdim=True) + 1e-8)
    fb = bags.reshape(N, U1 * U2, 4)[ar[:, None], idx]
    b = (fb * w[..., None]).sum(1)
    b[:, 0::2] = b[:, 0::2].clamp(0, img_hw[1])
    b[:, 1::2] = b[:, 1::2].clamp(0, img_hw[0])
    return (1 - beta) * b + beta * pseudo


# ----------------------------------------------------------------------------
# proposal bags                       detectors/syn_images_generator_v2.py
# ----------------------------------------------------------------------------


def fine_proposals(boxes, base_ratios, shake_ratio, min_scale, img_hw):
    """syn_images_generator_v2.py:262-324 for one image.  Returns
    (proposals [G*U,4], valid [G*U,1] bool) with U = R*R*(1 + 4*len(shake))."""
    c = bbox_xyxy_to_cxcywh(boxes)
    pps = []
    for rw in base_ratios:
        for rh in base_ratios:
            w = c[:, 2].clamp(min_scale, 1000) * rw
            h = c[:, 3].clamp(min_scale, 1000) * rh
            pps.append(bbox_cxcywh_to_xyxy(torch.stack([c[:, 0], c[:, 1], w, h], 1))[:, None])
    old = torch.cat(pps, 1)                                      # [G, R*R, 4]
    if shake_ratio is not None:
        new = [old[:, :, None, :]]
        for r in shake_ratio:
            p = bbox_xyxy_to_cxcywh(old)
            ctr, wh = p[..., :2], p[..., 2:4]
            xl, xr = ctr[..., 0] - r * wh[..., 0], ctr[..., 0] + r * wh[..., 0]
            yt, yd = ctr[..., 1] - r * wh[..., 1], ctr[..., 1] + r * wh[..., 1]
            cs = torch.stack([torch.stack([xl, ctr[..., 1]], -1), torch.stack([xr, ctr[..., 1]], -1),
                              torch.stack([ctr[..., 0], yt], -1), torch.stack([ctr[..., 0], yd], -1)], 2)
            q = torch.cat([cs, wh[:, :, None, :].expand_as(cs)], -1)
            new.append(bbox_cxcywh_to_xyxy(q))
        new = torch.cat(new, 2)
    else:
        new = old
    props = new.reshape(-1, 4)
    img = props.new_tensor([[0, 0, img_hw[1], img_hw[0]]])
    valid = bbox_overlaps(props, img, mode='iof') > 0.7
    return props, valid


def negative_proposals(u, pos_props, img_hw, iou_thr=0.3):
    """syn_images_generator_v2.py:234-259 for one image; ``u`` [4,n] are the four
    torch.rand draws in call order."""
    h, w = img_hw
    x1 = u[0] * w * 0.8
    y1 = u[1] * h * 0.8
    x2 = x1 + u[2] * 100
    y2 = y1 + u[3] * 100
    neg = torch.stack([x1, y1, x2, y2], 1)
    iou = bbox_overlaps(neg, pos_props)
    return neg, (iou < iou_thr).sum(1) == iou.shape[1]


def strong_augmentation_single(img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes,
                               flip, scale):
    """syn_images_generator_v2.py:24-132 for one image with the draws injected:
    ``flip`` in {'horizontal','vertical','diagonal','None'}; ``scale`` is the
    already-rounded factor (np.around(U(0.8,1.2),1))."""
    C, H, W = img.shape
    gp, gl = gt_points.clone(), gt_labels.clone()
    pb, pl, pp = pseudo_bboxes.clone(), pseudo_labels.clone(), pseudo_points.clone()
    if flip in ('horizontal', 'diagonal'):
        img = torch.flip(img, dims=[2])
        pb[:, 0::2] = W - pb[:, 0::2]; pp[:, 0] = W - pp[:, 0]; gp[:, 0] = W - gp[:, 0]
    if flip in ('vertical', 'diagonal'):
        img = torch.flip(img, dims=[1])
        pb[:, 1::2] = H - pb[:, 1::2]; pp[:, 1] = H - pp[:, 1]; gp[:, 1] = H - gp[:, 1]
    sH, sW = int(H * scale), int(W * scale)
    if scale < 1.0:
        bh, bw = int((H - sH) / 2), int((W - sW) / 2)
    else:
        bh, bw = int((sH - H) / 2), int((sW - W) / 2)
    pb = pb * scale; pp = pp * scale; gp = gp * scale
    if scale >= 1.0:
        k = ((gp[:, 0] >= bw) & (gp[:, 0] < W + bw) & (gp[:, 1] >= bh) & (gp[:, 1] < H + bh)).nonzero().reshape(-1)
        gp, gl = gp[k], gl[k]
        gp = gp - gp.new_tensor([bw, bh])
        k = ((pp[:, 0] >= bw) & (pp[:, 0] < W + bw) & (pp[:, 1] >= bh) & (pp[:, 1] < H + bh)).nonzero().reshape(-1)
        pb, pp, pl = pb[k], pp[k], pl[k]
        pp = pp - pp.new_tensor([bw, bh])
        pb = pb - pb.new_tensor([bw, bh, bw, bh])
    else:
        gp = gp + gp.new_tensor([bw, bh])
        pp = pp + pp.new_tensor([bw, bh])
        pb = pb + pb.new_tensor([bw, bh, bw, bh])
    r = F.interpolate(img[None], size=(sH, sW), mode='bilinear', align_corners=False)[0]
    if scale < 1.0:
        out = torch.zeros_like(img)
        sy, sx = (H - sH) // 2, (W - sW) // 2
        out[:, sy:sy + sH, sx:sx + sW] = r
    else:
        sy, sx = (sH - H) // 2, (sW - W) // 2
        out = r[:, sy:sy + H, sx:sx + W]
    out = torch.round(out)
    if pb.shape[0] != 0:
        w = (pb[:, 0] - pb[:, 2]).abs()
        h = (pb[:, 1] - pb[:, 3]).abs()
        x = torch.min(pb[:, 0], pb[:, 2])
        y = torch.min(pb[:, 1], pb[:, 3])
        pb = bbox_cxcywh_to_xyxy(torch.stack([x + w / 2, y + h / 2, w, h], 1))
    return out, gp, gl, pp, pl, pb


# ----------------------------------------------------------------------------
# head-level losses                      dense_heads/fcos_head_p2b_ts.py
# ----------------------------------------------------------------------------


def _flatten_head(cls, reg, ctr):
    B, C = cls.shape[:2]
    return (cls.permute(0, 2, 3, 1).reshape(B, -1, C), reg.permute(0, 2, 3, 1).reshape(B, -1, 4),
            ctr.permute(0, 2, 3, 1).reshape(B, -1))


def loss_pseudo(cls, reg, ctr, points, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes,
                num_classes=8, hyper=0.1):
    """fcos_head_p2b_ts.py:380-465 (single level, single process)."""
    fc, fr, ft = _flatten_head(cls, reg, ctr)
    B, P = fc.shape[:2]
    lr, bt, lb = [], [], []
    for i in range(B):
        a, b, c, _ = get_target_pseudo_single(points, fc[i].detach(), gt_points[i], gt_labels[i],
                                              pseudo_bboxes[i], pseudo_labels[i], num_classes)
        lr.append(a); bt.append(b); lb.append(c)
    lr, bt, lb = torch.cat(lr), torch.cat(bt), torch.cat(lb)
    fc, fr, ft = fc.reshape(-1, num_classes), fr.reshape(-1, 4), ft.reshape(-1)
    pts = points.repeat(B, 1)
    npos = max(float(((lb >= 0) & (lb < num_classes)).sum()), 1.0)
    loss_cls = sigmoid_focal_loss(fc, lb, torch.ones(B * P), avg_factor=npos)
    pos = ((lr >= 0) & (lr < num_classes)).nonzero().reshape(-1)
    npos = max(float(pos.numel()), 1.0)
    ct = centerness_target(bt[pos])
    den = max(float(ct.sum()), 1e-6)
    if pos.numel() > 0:
        loss_bbox = iou_loss_module(distance2bbox(pts[pos], fr[pos]), distance2bbox(pts[pos], bt[pos]),
                                    weight=ct, avg_factor=den, kind='dn', hyper=hyper)
        loss_ctr = bce_logits_loss(ft[pos], ct, avg_factor=npos)
    else:
        loss_bbox, loss_ctr = fr[pos].sum(), ft[pos].sum()
    return loss_cls, loss_bbox, loss_ctr


def loss_synthetic(reg, ctr, points, gt_bboxes, num_classes=8):
    """fcos_head_p2b_ts.py:470-534 (plain DIoU on the synthetic rectangles)."""
    B = reg.shape[0]
    fr = reg.permute(0, 2, 3, 1).reshape(-1, 4)
    ft = ctr.permute(0, 2, 3, 1).reshape(-1)
    lb, bt = [], []
    for i in range(B):
        a, b = get_target_single(points, gt_bboxes[i], num_classes)
        lb.append(a); bt.append(b)
    lb, bt = torch.cat(lb), torch.cat(bt)
    pts = points.repeat(B, 1)
    pos = ((lb >= 0) & (lb < num_classes)).nonzero().reshape(-1)
    npos = max(float(pos.numel()), 1.0)
    ct = centerness_target(bt[pos])
    den = max(float(ct.sum()), 1e-6)
    if pos.numel() > 0:
        loss_bbox = iou_loss_module(distance2bbox(pts[pos], fr[pos]), distance2bbox(pts[pos], bt[pos]),
                                    weight=ct, avg_factor=den, kind='diou')
        loss_ctr = bce_logits_loss(ft[pos], ct, avg_factor=npos)
    else:
        loss_bbox, loss_ctr = fr[pos].sum(), ft[pos].sum()
    return loss_bbox, loss_ctr


# ----------------------------------------------------------------------------
# RoIAlign  (mmcv.ops.RoIAlign, aligned=True, sampling_ratio=0, avg) - PARITY UNPINNED:
# restated from the published algorithm (Detectron2/torchvision ``aligned`` semantics)
# ----------------------------------------------------------------------------


def roi_align(feat, rois, out_size=7, spatial_scale=0.125, sampling_ratio=0, aligned=True):
    """feat [B,C,H,W], rois [K,5]=(batch,x1,y1,x2,y2) -> [K,C,out,out].  Pure torch
    gathers, so autograd gives the reference backward (scatter-add of the bilinear
    weights / count).  Call sites: fcos_head_p2b_ts.py:1202,1243,1268."""
    K = rois.shape[0]
    B, C, H, W = feat.shape
    out = feat.new_zeros((K, C, out_size, out_size))
    off = 0.5 if aligned else 0.0
    flat = feat.reshape(B, C, H * W)
    for k in range(K):
        b = int(rois[k, 0])
        sw = float(rois[k, 1]) * spatial_scale - off
        sh = float(rois[k, 2]) * spatial_scale - off
        ew = float(rois[k, 3]) * spatial_scale - off
        eh = float(rois[k, 4]) * spatial_scale - off
        rw, rh = ew - sw, eh - sh
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / out_size, rw / out_size
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / out_size))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / out_size))
        cnt = max(gh * gw, 1)
        if gh <= 0 or gw <= 0:
            continue
        ph = torch.arange(out_size, dtype=torch.float32)
        iy = torch.arange(gh, dtype=torch.float32)
        ix = torch.arange(gw, dtype=torch.float32)
        ys = (np.float32(sh) + ph[:, None] * np.float32(bh) + (iy[None, :] + 0.5) * np.float32(bh) / gh).reshape(-1)
        xs = (np.float32(sw) + ph[:, None] * np.float32(bw) + (ix[None, :] + 0.5) * np.float32(bw) / gw).reshape(-1)

        def prep(v, L):
            oob = (v < -1.0) | (v > L)
            v = v.clamp(min=0)
            lo = v.floor().long()
            hi_clip = lo >= L - 1
            lo = torch.where(hi_clip, torch.full_like(lo, L - 1), lo)
            hi = torch.where(hi_clip, lo, lo + 1)
            v = torch.where(hi_clip, lo.float(), v)
            l = v - lo.float()
            return lo, hi, l, 1 - l, oob
        ylo, yhi, ly, hy, yo = prep(ys, H)
        xlo, xhi, lx, hx, xo = prep(xs, W)
        m = (~yo)[:, None] & (~xo)[None, :]
        f = flat[b]                                             # [C, H*W]

        def g(yy, xx):
            return f[:, (yy[:, None] * W + xx[None, :]).reshape(-1)].reshape(C, ys.numel(), xs.numel())
        val = (g(ylo, xlo) * (hy[:, None] * hx[None, :]) + g(ylo, xhi) * (hy[:, None] * lx[None, :]) +
               g(yhi, xlo) * (ly[:, None] * hx[None, :]) + g(yhi, xhi) * (ly[:, None] * lx[None, :]))
        val = val * m.to(val.dtype)
        out[k] = val.reshape(C, out_size, gh, out_size, gw).sum((2, 4)) / cnt
    return out


# ----------------------------------------------------------------------------
# NMS (mmcv.ops.nms, offset=0) and rotated IoU / NMS (mmcv box_iou_rotated, nms_rotated)
# PARITY UNPINNED: published greedy algorithm / Sutherland-Hodgman style polygon clip.
# ----------------------------------------------------------------------------


def nms(boxes, scores, iou_thr):
    """Greedy NMS, descending score (stable), suppress IoU > thr; returns keep indices."""
    order = np.argsort(-scores.numpy(), kind='stable')
    b = boxes.numpy().astype(np.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sup = np.zeros(len(order), bool)
    keep = []
    for _i, i in enumerate(order):
        if sup[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(b[i, 0], b[rest, 0]); yy1 = np.maximum(b[i, 1], b[rest, 1])
        xx2 = np.minimum(b[i, 2], b[rest, 2]); yy2 = np.minimum(b[i, 3], b[rest, 3])
        inter = np.maximum(xx2 - xx1, np.float32(0)) * np.maximum(yy2 - yy1, np.float32(0))
        iou = inter / (area[i] + area[rest] - inter)
        sup[rest[iou > iou_thr]] = True
    return torch.tensor(keep, dtype=torch.long)


def batched_nms(boxes, scores, idxs, iou_thr):
    """mmcv.ops.batched_nms (class-aware): offset boxes by idx * (max coord + 1)."""
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.long)
    off = idxs.to(boxes) * (boxes.max() + 1)
    keep = nms(boxes + off[:, None], scores, iou_thr)
    return keep


def _rbox_corners(b):
    cx, cy, w, h, a = [float(v) for v in b]
    c, s = math.cos(a), math.sin(a)
    pts = []
    for dx, dy in ((-w / 2, -h / 2), (w / 2, -h / 2), (w / 2, h / 2), (-w / 2, h / 2)):
        pts.append((cx + dx * c - dy * s, cy + dx * s + dy * c))
    return pts


def _clip_poly(subject, a, b):
    """Keep the part of ``subject`` on the left of the directed edge a->b."""
    out = []
    n = len(subject)

    def side(p):
        return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
    for i in range(n):
        p, q = subject[i], subject[(i + 1) % n]
        sp, sq = side(p), side(q)
        if sp >= 0:
            out.append(p)
        if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
            t = sp / (sp - sq)
            out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def rotated_iou_single(b1, b2):
    """IoU of two (cx,cy,w,h,angle) boxes by convex polygon clipping (fp64)."""
    a1, a2 = float(b1[2]) * float(b1[3]), float(b2[2]) * float(b2[3])
    if a1 < 1e-14 or a2 < 1e-14:
        return 0.0
    p1, p2 = _rbox_corners(b1), _rbox_corners(b2)
    poly = p1
    for i in range(4):
        if not poly:
            break
        poly = _clip_poly(poly, p2[i], p2[(i + 1) % 4])
    inter = 0.0
    for i in range(len(poly)):
        x1, y1 = poly[i]
        x2, y2 = poly[(i + 1) % len(poly)]
        inter += x1 * y2 - x2 * y1
    inter = abs(inter) / 2
    return inter / (a1 + a2 - inter)


def box_iou_rotated(b1, b2, aligned=False):
    if aligned:
        return torch.tensor([rotated_iou_single(b1[i], b2[i]) for i in range(b1.shape[0])], dtype=torch.float32)
    return torch.tensor([[rotated_iou_single(x, y) for y in b2] for x in b1], dtype=torch.float32).reshape(
        b1.shape[0], b2.shape[0])


def nms_rotated(dets, scores, iou_thr):
    order = np.argsort(-scores.numpy(), kind='stable')
    sup = np.zeros(len(order), bool)
    keep = []
    for _i, i in enumerate(order):
        if sup[i]:
            continue
        keep.append(i)
        for j in order[_i + 1:]:
            if not sup[j] and rotated_iou_single(dets[i], dets[j]) > iou_thr:
                sup[j] = True
    return torch.tensor(keep, dtype=torch.long)


def fill_convex_quads(polys_int, H, W):
    """Contract for the step-1 rasteriser (syn_images_generator_v2.py:678-683): a pixel
    (x, y) is painted iff it lies inside or on the boundary of the quadrilateral whose
    vertices were truncated to int32.  cv2.fillPoly itself is absent here (PARITY
    UNPINNED at pixel level)."""
    mask = np.zeros((H, W), np.uint8)
    ys, xs = np.mgrid[0:H, 0:W]
    for q in polys_int.reshape(-1, 4, 2):
        q = q.astype(np.int64)
        area2 = 0
        for i in range(4):
            x1, y1 = q[i]; x2, y2 = q[(i + 1) % 4]
            area2 += x1 * y2 - x2 * y1
        sgn = 1 if area2 >= 0 else -1
        inside = np.ones((H, W), bool)
        for i in range(4):
            x1, y1 = q[i]; x2, y2 = q[(i + 1) % 4]
            cr = (x2 - x1) * (ys - y1) - (y2 - y1) * (xs - x1)
            inside &= (cr * sgn >= 0)
        if area2 == 0:
            continue
        mask[inside] = 1
    return mask


def generate_black_paper(img, gt_bboxes, prior_size, draws, imgsize=None, oriented=False):
    """Burn-in step 1 "white rectangle" generator for ONE image: detectors/fcos_p2b_teacher_student.py:469-502
    (`genrate_syn.synthesis_single`) around detectors/syn_images_generator_v2.py:591-690 (`generate_black_paper`),
    with every random draw supplied in `draws` (per-object arrays: `cls` = the torch.randint prior index :473,
    `scale` :597, `x`,`y` :613, `wn` :616, `rn` :619, `a` :625, `boost` = np.random.random() :640, `itv` / `dev` =
    the two torch.rand of :643-645 / :654-656; only read for objects whose boost fires).

    Returns (img_syn [C,H,W], hull boxes xyxy [M,4] in the reference's `bb[keep]` order, keep indices into
    [real objects | candidate rectangles], mask [H,W] uint8).  Restated one object at a time, python floats
    rounded through float32 tensors exactly where the reference holds tensors.

    oriented=True: the OBB twin (OBB_TOD/mmrotate/models/detectors/rotated_fcos_teacher_student.py:391-414 around
    OBB_TOD/.../syn_images_generator_v2.py generate_black_paper, identical up to its last statement): `gt_bboxes` are
    (cx,cy,w,h,a) rows of which only the centre is read, the rectangles are painted with the IMAGE MAXIMUM and returned
    as (cx,cy,w,h,a) rows instead of their hulls."""
    C, H, W = img.shape
    imgsize = min(H, W) if imgsize is None else imgsize
    prior = prior_size.float()
    n_dense = int(prior.shape[0] / 2)                                   # dense_cls = range(len(pattern)/2), :483
    G = gt_bboxes.shape[0]
    f32 = torch.float32
    cxcy = gt_bboxes.float() if oriented else bbox_xyxy_to_cxcywh(gt_bboxes.float())
    rows = []                                                           # candidate rectangles [x,y,w,h,a,score]
    adj = 2
    lo, hi = 50.0, float(imgsize - 50)
    for j in range(G):
        c = int(draws['cls'][j])
        base = draws['scale'][j].to(f32) * 2.0 + 0.5                    # :597
        x = draws['x'][j].to(f32) * (hi - lo) + lo                      # :613-614
        y = draws['y'][j].to(f32) * (hi - lo) + lo
        w = base * torch.exp((draws['wn'][j].to(f32) * 0.4).clamp(-1, 1) * prior[c, 2])     # :615-617
        h = w * torch.exp((draws['rn'][j].to(f32) * 0.4).clamp(-1, 1) * prior[c, 3])        # :618-621
        w = w * prior[c, 0]
        h = h * prior[c, 1]
        a = draws['a'][j].to(f32) * math.pi - math.pi / 2                # :625 (torch.pi = math.pi, :582)
        # Tensor.clip(min, max) = min(max(x, lo), hi): the upper bound wins when the bounds cross (:627-628)
        x = torch.minimum(torch.maximum(x, 0.71 * w), imgsize - 1 - 0.71 * w)
        y = torch.minimum(torch.maximum(y, 0.71 * h), imgsize - 1 - 0.71 * h)
        score = (w * h) / imgsize / imgsize + 0.1
        rows.append([x, y, w, h, a, score])
        if float(draws['boost'][j]) < 0.2 and adj > 0:                   # :640-663
            adj -= 1
            if c < n_dense:
                itv, dev, n = draws['itv'][j].to(f32) * 4 + 2, draws['dev'][j].to(f32) * 8 - 4, 5
            else:
                itv, dev, n = draws['itv'][j].to(f32) * 40 + 10, draws['dev'][j].to(f32) * 0, 3
            ofx = (h + itv) * torch.sin(-a) + dev * torch.cos(a)
            ofy = (h + itv) * torch.cos(a) + dev * torch.sin(a)
            for k in range(1, n + 1):
                rows.append([x + k * ofx, y + k * ofy, w, h, a, (w * h) / imgsize / imgsize + 0.1 - 0.001 * k])
    cand = torch.tensor([[float(v) for v in r] for r in rows], dtype=f32).reshape(-1, 6)      # torch.tensor(bb), :665
    cls = draws['cls'].long()
    occ = torch.stack([cxcy[:, 0], cxcy[:, 1], prior[cls, 0] * 0.7, prior[cls, 0] * 0.7, torch.zeros(G), torch.ones(G)], 1)
    allb = torch.cat([occ, cand], 0)                                     # :666
    keep = nms_rotated(allb[:, :5], allb[:, 5], 0.05)                    # :667 (mmcv op: parity unpinned)
    bb = allb[keep]
    bb = bb[bb[:, 5] < 1]                                                # the real objects leave, :669
    xyxy = obb2xyxy(bb)
    inside = (xyxy.min(-1)[0] >= 0) & (xyxy.max(-1)[0] <= imgsize - 1)   # :671-675
    bb = bb[inside]
    polys = obb2poly_le90(bb[:, :5])
    mask = fill_convex_quads(polys.numpy().astype(np.int32), H, W)       # cv2.fillPoly on int32-truncated corners, :678-683
    img_syn = img.clone()
    img_syn[:, torch.from_numpy(mask) == 1] = img.max() if oriented else 255          # :686-688
    if oriented:
        return img_syn, bb[:, :5], keep, mask
    px, py = polys[:, 0::2], polys[:, 1::2]
    hull = torch.stack([px.min(1)[0], py.min(1)[0], px.max(1)[0], py.max(1)[0]], 1) if bb.shape[0] else bb.new_zeros((0, 4))
    return img_syn, hull, keep, mask


# ----------------------------------------------------------------------------
# EMA / optimizer      detectors/fcos_p2b_teacher_student.py:254-257; mmcv OptimizerHook
# ----------------------------------------------------------------------------


def ema_update(teacher, student, alpha=0.999):
    return teacher * alpha + (1 - alpha) * student


def sgd_momentum_step(p, g, buf, lr, momentum=0.9, weight_decay=1e-4, clip_coef=1.0):
    """torch.optim.SGD semantics (dampening 0, no nesterov) after grad-clip scaling."""
    g = g * clip_coef + weight_decay * p
    buf = momentum * buf + g
    return p - lr * buf, buf


# ----------------------------------------------------------------------------
# Oriented-box ops of the OBB variant (mmcv.ops.diff_iou_rotated_2d / RoIAlignRotated) -
# PARITY UNPINNED (mmcv-full absent): published algorithms, checked by invariants.
# ----------------------------------------------------------------------------


def diff_iou_rotated_grad(b1, b2, h=1e-4):
    """(iou [N], d iou / d b1 [N,5]) by central differences of the fp64 polygon-clip IoU."""
    N = b1.shape[0]
    iou = np.zeros(N)
    grad = np.zeros((N, 5))
    a, b = b1.double().numpy(), b2.double().numpy()
    for n in range(N):
        iou[n] = rotated_iou_single(a[n], b[n])
        for k in range(5):
            p, m = a[n].copy(), a[n].copy()
            p[k] += h; m[k] -= h
            grad[n, k] = (rotated_iou_single(p, b[n]) - rotated_iou_single(m, b[n])) / (2 * h)
    return torch.from_numpy(iou).float(), torch.from_numpy(grad).float()


def roi_align_rotated(feat, rois, out_size=7, spatial_scale=0.125, sample_num=2, aligned=True, clockwise=True):
    """Published mmcv algorithm (roi_align_rotated_cuda_kernel.cuh): the sampling grid of each bin is
    rotated by theta about the RoI centre.  torch gathers -> autograd supplies the backward."""
    K = rois.shape[0]
    B, C, H, W = feat.shape
    out = feat.new_zeros((K, C, out_size, out_size))
    flat = feat.reshape(B, C, H * W)
    off = 0.5 if aligned else 0.0
    for k in range(K):
        b = int(rois[k, 0])
        cw, ch = float(rois[k, 1]) * spatial_scale - off, float(rois[k, 2]) * spatial_scale - off
        rw, rh = float(rois[k, 3]) * spatial_scale, float(rois[k, 4]) * spatial_scale
        th = -float(rois[k, 5]) if clockwise else float(rois[k, 5])
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        gh = sample_num if sample_num > 0 else int(math.ceil(rh / out_size))
        gw = sample_num if sample_num > 0 else int(math.ceil(rw / out_size))
        if gh <= 0 or gw <= 0:
            continue
        bh, bw = np.float32(rh / out_size), np.float32(rw / out_size)
        ph = torch.arange(out_size, dtype=torch.float32)
        yy = (np.float32(-rh / 2) + ph[:, None] * bh + (torch.arange(gh, dtype=torch.float32)[None] + 0.5) * bh / gh).reshape(-1)
        xx = (np.float32(-rw / 2) + ph[:, None] * bw + (torch.arange(gw, dtype=torch.float32)[None] + 0.5) * bw / gw).reshape(-1)
        c, s = np.float32(math.cos(th)), np.float32(math.sin(th))
        Y = yy[:, None] * c - xx[None, :] * s + np.float32(ch)
        X = yy[:, None] * s + xx[None, :] * c + np.float32(cw)
        oob = (Y < -1.0) | (Y > H) | (X < -1.0) | (X > W)

        def prep(v, L):
            v = v.clamp(min=0)
            lo = v.floor().long()
            hc = lo >= L - 1
            lo = torch.where(hc, torch.full_like(lo, L - 1), lo)
            hi = torch.where(hc, lo, lo + 1)
            v = torch.where(hc, lo.float(), v)
            l = v - lo.float()
            return lo, hi, l, 1 - l
        ylo, yhi, ly, hy = prep(Y, H)
        xlo, xhi, lx, hx = prep(X, W)
        f = flat[b]

        def g(yi, xi):
            return f[:, (yi * W + xi).reshape(-1)].reshape(C, *Y.shape)
        val = g(ylo, xlo) * (hy * hx) + g(ylo, xhi) * (hy * lx) + g(yhi, xlo) * (ly * hx) + g(yhi, xhi) * (ly * lx)
        val = val * (~oob).to(val.dtype)
        out[k] = val.reshape(C, out_size, gh, out_size, gw).sum((2, 4)) / max(gh * gw, 1)
    return out


def distance2obb(points, distance, angle_version='le90'):
    """OBB_TOD/mmrotate/core/bbox/coder/distance_angle_point_coder.py:93-111"""
    d, angle = distance.split([4, 1], dim=1)
    c, s = torch.cos(angle), torch.sin(angle)
    rot = torch.cat([c, -s, s, c], dim=1).reshape(-1, 2, 2)
    wh = d[:, :2] + d[:, 2:]
    off = torch.bmm(rot, ((d[:, 2:] - d[:, :2]) / 2).unsqueeze(2)).squeeze(2)
    if angle_version == 'le90':
        angle = (angle + np.pi / 2) % np.pi - np.pi / 2
    elif angle_version == 'le135':
        angle = (angle + np.pi / 4) % np.pi - np.pi / 4
    return torch.cat([points + off, wh, angle], dim=-1)


def rotated_targets_single(points, gt_inds, boxes5):
    """OBB_TOD/mmrotate/models/dense_heads/rotated_fcos_head_p2rb_ts.py:671-716 through the dense
    [P,G] construction the reference uses."""
    P, G = points.shape[0], boxes5.shape[0]
    pts = points[:, None, :].expand(P, G, 2)
    gb = boxes5[None].expand(P, G, 5)
    ctr, wh, ang = torch.split(gb, [2, 2, 1], dim=2)
    c, s = torch.cos(ang), torch.sin(ang)
    rot = torch.cat([c, s, -s, c], dim=-1).reshape(P, G, 2, 2)
    off = torch.matmul(rot, (pts - ctr)[..., None]).squeeze(-1)
    w, h = wh[..., 0], wh[..., 1]
    t = torch.stack((w / 2 + off[..., 0], h / 2 + off[..., 1], w / 2 - off[..., 0], h / 2 - off[..., 1]), -1)
    idx = torch.where(gt_inds > 0, gt_inds - 1, torch.zeros_like(gt_inds))
    ar = torch.arange(P)
    return t[ar, idx], ang[ar, idx]


def dn_iou_loss_values(pred, target, hyper=0.2, eps=1e-6, mode='log'):
    """OBB_TOD/mmrotate/models/losses/rotated_iou_loss.py:105-147 (values only; fp64 polygon clip)."""
    def loss_of(t):
        i = box_iou_rotated(pred, t, aligned=True).clamp(min=eps)
        return -i.log() if mode == 'log' else (1 - i if mode == 'linear' else 1 - i ** 2)
    base = loss_of(target)
    anx = hyper / 2
    w, h = target[:, 2], target[:, 3]
    bank = []
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            t = target.clone()
            t[:, 2] = t[:, 2] - anx * w * i
            t[:, 3] = t[:, 3] - anx * h * j
            bank.append(loss_of(t))
    return (base + torch.stack(bank, 1).min(1)[0]) / 2


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, deform_groups=1):
    """mmcv.ops.(modulated_)deform_conv2d restated from the published algorithm
    (modulated_deform_conv_cuda_kernel.cuh: dmcn_im2col_bilinear + GEMM), PARITY UNPINNED (mmcv-full absent).
    Pure torch gathers, so autograd supplies the reference gradients for x, offset, mask, weight and bias.
    offset[B, 2*dg*K, Ho, Wo] = (dy, dx) per tap; mask[B, dg*K, Ho, Wo] or None; groups = 1."""
    B, C, H, W = x.shape
    O, _, kh, kw = weight.shape
    s, p, d = (stride, stride) if isinstance(stride, int) else stride, (padding, padding) if isinstance(padding, int) else padding, \
        (dilation, dilation) if isinstance(dilation, int) else dilation
    Ho = (H + 2 * p[0] - (d[0] * (kh - 1) + 1)) // s[0] + 1
    Wo = (W + 2 * p[1] - (d[1] * (kw - 1) + 1)) // s[1] + 1
    K, cpg = kh * kw, C // deform_groups
    ho = torch.arange(Ho, dtype=x.dtype)[:, None] * s[0] - p[0]
    wo = torch.arange(Wo, dtype=x.dtype)[None, :] * s[1] - p[1]
    cols = []
    flat = x.reshape(B, C, H * W)
    for t in range(K):
        ki, kj = t // kw, t % kw
        per_group = []
        for g in range(deform_groups):
            h = ho + ki * d[0] + offset[:, (g * K + t) * 2]                  # [B, Ho, Wo]
            w = wo + kj * d[1] + offset[:, (g * K + t) * 2 + 1]
            inside = (h > -1) & (w > -1) & (h < H) & (w < W)
            hl, wl = torch.floor(h), torch.floor(w)
            lh, lw = h - hl, w - wl
            val = 0
            for dy, dx, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
                yy, xx = hl.long() + dy, wl.long() + dx
                ok = inside & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
                idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(B, 1, -1).expand(B, cpg, -1)
                v = torch.gather(flat[:, g * cpg:(g + 1) * cpg], 2, idx).reshape(B, cpg, Ho, Wo)
                val = val + v * (wt * ok.to(x.dtype))[:, None]
            if mask is not None:
                val = val * mask[:, g * K + t][:, None]
            per_group.append(val)
        cols.append(torch.cat(per_group, 1))                                   # [B, C, Ho, Wo]
    col = torch.stack(cols, 2).reshape(B, C * K, Ho * Wo)                      # channel-major, tap-minor
    out = torch.matmul(weight.reshape(O, -1), col).reshape(B, O, Ho, Wo)
    return out if bias is None else out + bias.view(1, -1, 1, 1)
