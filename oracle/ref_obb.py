"""CPU ORACLE of the ORIENTED-BOX Point-Teacher iteration (config 5) - TEST INFRASTRUCTURE ONLY.

A functional torch-CPU fp32 restatement of `RotatedFCOS_TS.forward_train` (reference:
/root/reference/OBB_TOD/mmrotate/models/detectors/rotated_fcos_teacher_student.py:128-252,
dense_heads/rotated_fcos_head_p2rb_ts.py and detectors/syn_images_generator_v2.py), built from
`torch.nn.functional` and the op-level oracle `oracle/ref_ops.py`.  It shares no code with the
product package.  Only tests/ may import it; the product never does.

Pinning.  Pure-torch reference functions are pinned by goldens captured from the reference
itself (tests/golden/obb_*.npz, oracle/gen_golden.py): poly2obb_le90 / obb2poly_le90,
DistanceAnglePointCoder, the geometric half of strong_augmentation.  The assigners and the MIL
bag construction are the HBB ones (the OBB tree calls the HBB fork's TopkAssigner /
FUSETopkAssigner / InsiderCost and fine_proposals_from_cfg) and are pinned there.
PARITY UNPINNED (mmcv-full / torchvision absent here, published algorithms restated): rotated IoU
and its gradient (fp64 polygon clipping + central differences), RoIAlignRotated, rotated NMS,
torchvision.transforms.functional.rotate (nearest-neighbour affine_grid / grid_sample).

Citations are file:line under /root/reference/OBB_TOD/mmrotate/ unless noted.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ref_model as M
from . import ref_ops as R

NUM_CLASSES = 9
STRIDE = 8


# --------------------------------------------------------------------- transforms --
def norm_angle_le90(a):
    """core/bbox/transforms.py:850-867"""
    return (a + np.pi / 2) % np.pi - np.pi / 2


def poly2obb_le90(polys):
    """core/bbox/transforms.py:301-331"""
    polys = polys.reshape(-1, 8)
    p1, p2, p3, p4 = polys[:, 0:2], polys[:, 2:4], polys[:, 4:6], polys[:, 6:8]
    e1 = torch.sqrt((p1[:, 0] - p2[:, 0]) ** 2 + (p1[:, 1] - p2[:, 1]) ** 2)
    e2 = torch.sqrt((p2[:, 0] - p3[:, 0]) ** 2 + (p2[:, 1] - p3[:, 1]) ** 2)
    a1 = torch.atan2(p2[:, 1] - p1[:, 1], p2[:, 0] - p1[:, 0])
    a2 = torch.atan2(p4[:, 1] - p1[:, 1], p4[:, 0] - p1[:, 0])
    ang = polys.new_zeros(polys.shape[0])
    ang[e1 > e2] = a1[e1 > e2]
    ang[e1 <= e2] = a2[e1 <= e2]
    ang = norm_angle_le90(ang)
    return torch.stack([(p1[:, 0] + p3[:, 0]) / 2.0, (p1[:, 1] + p3[:, 1]) / 2.0, torch.max(e1, e2), torch.min(e1, e2), ang], 1)


def obb2distance(points, boxes5, max_dis=None, eps=None):
    """core/bbox/coder/distance_angle_point_coder.py:73-91 (encode)."""
    ctr, wh, ang = torch.split(boxes5, [2, 2, 1], dim=1)
    c, s = torch.cos(ang), torch.sin(ang)
    rot = torch.cat([c, s, -s, c], dim=1).reshape(-1, 2, 2)
    off = torch.bmm(rot, (points - ctr)[..., None]).squeeze(-1)
    w, h = wh[..., 0], wh[..., 1]
    l, r, t, b = w / 2 + off[..., 0], w / 2 - off[..., 0], h / 2 + off[..., 1], h / 2 - off[..., 1]
    if max_dis is not None:
        l, t, r, b = [v.clamp(min=0, max=max_dis - eps) for v in (l, t, r, b)]
    return torch.stack((l, t, r, b, ang[:, 0]), -1)


# ----------------------------------------------------------- differentiable rotated IoU --
class _RotIoU(torch.autograd.Function):
    """mmcv.ops.diff_iou_rotated_2d for aligned [N,5] boxes: value = fp64 polygon-clip IoU,
    gradient w.r.t. the first operand = central differences of the same function."""

    @staticmethod
    def forward(ctx, pred, target):
        iou, grad = R.diff_iou_rotated_grad(pred.detach(), target.detach(), h=1e-4)
        ctx.save_for_backward(grad)
        return iou

    @staticmethod
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        return grad * g[:, None], None


def rotated_iou_loss(pred, target, weight, avg_factor, eps=1e-6):
    """models/losses/rotated_iou_loss.py:17-60 + :149-226 (mode 'log', loss_weight 1)."""
    if not torch.any(weight > 0):
        return (pred * weight.unsqueeze(-1)).sum()
    iou = _RotIoU.apply(pred, target).clamp(min=eps)
    return ((-iou.log()) * weight).sum() / avg_factor


# ---------------------------------------------------------------------- head pieces --
def generate_pseudo_single_obb(points, cls, reg5, gt_points, gt_labels, gt_bboxes, filter_score=0.0):
    """dense_heads/rotated_fcos_head_p2rb_ts.py:871-929 -> (pseudo_bboxes[G,5], pseudo_points[G,2],
    scores[G], mean_iou, valid_inds)."""
    G = gt_labels.shape[0]
    if G == 0:
        return torch.empty(0, 5), torch.empty(0, 2), torch.empty(0), torch.tensor(0.0), torch.empty(0, dtype=torch.long)
    act = cls.sigmoid()
    dec = R.distance2obb(points, reg5, 'le90')
    gi, gl = R.fuse_topk_assign(dec[:, :4], points, cls, gt_points, gt_labels)
    pos = (gi != 0).nonzero().reshape(-1)
    lab = torch.zeros(points.shape[0], dtype=torch.long)
    lab[pos] = gl[pos]
    score = act[torch.arange(points.shape[0]), lab]
    A, B, C = dec[pos], gi[pos] - 1, score[pos]
    nums = torch.bincount(B, minlength=G)
    pb = 8 * torch.ones(G, 5)
    pb[:, :2] = gt_points
    pb[:, -1] = 0
    pp = gt_points.clone()
    ps = torch.zeros(G)
    bsum = torch.zeros(G, 5).index_add_(0, B, A * C[:, None])
    ssum = torch.zeros(G).index_add_(0, B, C)
    nz = nums != 0
    pb[nz] = bsum[nz] / ssum[nz, None]
    ps[nz] = ssum[nz] / nums[nz]
    pp[nz] = pb[nz, :2]
    miou = R.box_iou_rotated(pb[nz], gt_bboxes[nz], aligned=True).mean()
    valid = (nz & (ps >= filter_score)).nonzero().reshape(-1)
    return pb, pp, ps, miou, valid


def _flatten_head5(cls, reg, ang, ctr):
    """concat_per_img :1140-1161"""
    B, C = cls.shape[:2]
    return (cls.permute(0, 2, 3, 1).reshape(B, -1, C),
            torch.cat([reg.permute(0, 2, 3, 1).reshape(B, -1, 4), ang.permute(0, 2, 3, 1).reshape(B, -1, 1)], -1),
            ctr.permute(0, 2, 3, 1).reshape(B, -1))


def _reg_losses(fr5, ft, pts, labels_reg, bt, at, num_classes):
    pos = ((labels_reg >= 0) & (labels_reg < num_classes)).nonzero().reshape(-1)
    npos = max(float(pos.numel()), 1.0)
    ct = R.centerness_target(bt[pos])
    den = max(float(ct.sum()), 1e-6)
    if pos.numel() > 0:
        pred = R.distance2obb(pts[pos], fr5[pos], 'le90')
        tgt = R.distance2obb(pts[pos], torch.cat([bt[pos], at[pos]], -1), 'le90')
        return rotated_iou_loss(pred, tgt, ct, den), R.bce_logits_loss(ft[pos], ct, avg_factor=npos)
    return fr5[pos].sum(), ft[pos].sum()


def loss_pseudo_obb(cls, reg, ang, ctr, points, gt_points, gt_labels, pseudo_labels, pseudo_bboxes,
                    num_classes=NUM_CLASSES):
    """:417-529 with assigner (1,1) and pseudo_assigner (3,3) (:781-843)."""
    fc, fr, ft = _flatten_head5(cls, reg, ang, ctr)
    B, P = fc.shape[:2]
    lr, bt, at, lb = [], [], [], []
    for i in range(B):
        gi, gl = R.topk_assign(points, gt_points[i], gt_labels[i], 1)
        labels = torch.full((P,), num_classes, dtype=torch.long)
        labels[gi != 0] = gl[gi != 0]
        lb.append(labels)
        if pseudo_labels[i].shape[0] == 0:
            lr.append(torch.full((P,), num_classes, dtype=torch.long)); bt.append(points.new_zeros((P, 4)))
            at.append(points.new_zeros((P, 1)))
            continue
        gi, gl = R.topk_assign(points, pseudo_bboxes[i], pseudo_labels[i], 3)
        labels_reg = torch.full((P,), num_classes, dtype=torch.long)
        labels_reg[gi != 0] = gl[gi != 0]
        t, a = R.rotated_targets_single(points, gi, pseudo_bboxes[i])
        lr.append(labels_reg); bt.append(t); at.append(a)
    lr, bt, at, lb = torch.cat(lr), torch.cat(bt), torch.cat(at), torch.cat(lb)
    fc, fr, ft = fc.reshape(-1, num_classes), fr.reshape(-1, 5), ft.reshape(-1)
    pts = points.repeat(B, 1)
    npos = max(float(((lb >= 0) & (lb < num_classes)).sum()), 1.0)
    loss_cls = R.sigmoid_focal_loss(fc, lb, torch.ones(B * P), avg_factor=npos)
    loss_bbox, loss_ctr = _reg_losses(fr, ft, pts, lr, bt, at, num_classes)
    return loss_cls, loss_bbox, loss_ctr


def loss_synthetic_obb(reg, ang, ctr, points, gt_bboxes, num_classes=NUM_CLASSES):
    """:535-626 with syn_assigner (3,3) (:671-716); every synthetic box has label 0."""
    B = reg.shape[0]
    fr = torch.cat([reg.permute(0, 2, 3, 1).reshape(B, -1, 4), ang.permute(0, 2, 3, 1).reshape(B, -1, 1)], -1).reshape(-1, 5)
    ft = ctr.permute(0, 2, 3, 1).reshape(-1)
    P = points.shape[0]
    lb, bt, at = [], [], []
    for i in range(B):
        if gt_bboxes[i].shape[0] == 0:
            lb.append(torch.full((P,), num_classes, dtype=torch.long)); bt.append(points.new_zeros((P, 4)))
            at.append(points.new_zeros((P, 1)))
            continue
        gi, gl = R.topk_assign(points, gt_bboxes[i], torch.zeros(gt_bboxes[i].shape[0], dtype=torch.long), 3)
        labels = torch.full((P,), num_classes, dtype=torch.long)
        labels[gi != 0] = gl[gi != 0]
        t, a = R.rotated_targets_single(points, gi, gt_bboxes[i])
        lb.append(labels); bt.append(t); at.append(a)
    return _reg_losses(fr, ft, points.repeat(B, 1), torch.cat(lb), torch.cat(bt), torch.cat(at), num_classes)


# ------------------------------------------------------------------------ MIL pieces --
def fine_proposals_obb(boxes5, cfg, img_hw):
    """detectors/syn_images_generator_v2.py:24-39 for one image: bags around the (cx,cy,w,h) part, angle re-attached."""
    props, valid = R.fine_proposals(R.bbox_cxcywh_to_xyxy(boxes5[:, :4]), cfg['base_ratios'], cfg['shake_ratio'],
                                    cfg['min_scale'], img_hw)
    U = props.shape[0] // max(boxes5.shape[0], 1)
    ang = boxes5[:, 4:5][:, None].repeat(1, U, 1).reshape(-1, 1)
    return torch.cat([R.bbox_xyxy_to_cxcywh(props), ang], 1), valid


def negative_proposals_obb(u, pos5, img_hw, iou_thr=0.3):
    """detectors/syn_images_generator_v2.py:129-156 for one image; `u` [5,n] = the five torch.rand draws."""
    h, w = img_hw
    x1 = u[0] * w * 0.8
    y1 = u[1] * h * 0.8
    neg = torch.stack([x1, y1, x1 + u[2] * 200, y1 + u[3] * 200, u[4] * math.pi - math.pi / 2], 1)
    iou = R.box_iou_rotated(neg, pos5)
    return neg, (iou < iou_thr).sum(1) == iou.shape[1]


def mil_bag_select_obb(cls, ins, valid, labels, bags5, pseudo5, img_hw, topk=3, beta=0.25):
    """:1218-1250 + :1198-1216 (columns 0,1 clamped to [0,w] THEN to [0,h])."""
    N, U1, U2, C = cls.shape
    v = valid.reshape(N, U1, U2, 1).to(cls.dtype)
    c = cls.detach().reshape(N, U1 * U2, C).sigmoid()
    i = F.normalize(ins.detach().softmax(2) * v, dim=2, p=1).reshape(N, U1 * U2, C)
    ar = torch.arange(N)
    s = c[ar, :, labels] * i[ar, :, labels]
    sc, idx = s.topk(topk, dim=1)
    w = sc.unsqueeze(2).repeat(1, 1, 5)
    w = w / (w.sum(dim=1, keepdim=True) + 1e-8)
    fb = bags5.reshape(N, U1 * U2, 5)[ar[:, None], idx]
    b = (fb * w).sum(1)
    b[:, [0, 1]] = b[:, [0, 1]].clamp(0, img_hw[1])
    b[:, [0, 1]] = b[:, [0, 1]].clamp(0, img_hw[0])
    return (1 - beta) * b + beta * pseudo5


def mil_select_margin(cls, ins, valid, labels, bags5, topk=3):
    """How far every gt's selection is from being decided by rounding.  Returns (gap [N], shift [N]): `gap` = relative
    distance between the last selected and the first rejected bag score; `shift` = how far (px / rad, max over the five
    columns) the weighted-mean box would move if those two swapped.  A row is ROUNDING-SENSITIVE when gap is below the
    fp32 noise of the 12544-long FC dot products (a few 1e-4 relative between two GEMM implementations) AND shift is
    visible.  Rows whose rejected candidate has the same score because it is the same box, or a zero score because it
    is an invalid proposal, have shift = 0 and are not sensitive.  Whole-iteration tests assert that their fixtures
    contain NO sensitive row, so that every selection is independent evidence."""
    N, U1, U2, C = cls.shape
    v = valid.reshape(N, U1, U2, 1).to(cls.dtype)
    c = cls.detach().reshape(N, U1 * U2, C).sigmoid()
    i = F.normalize(ins.detach().softmax(2) * v, dim=2, p=1).reshape(N, U1 * U2, C)
    ar = torch.arange(N)
    t, idx = (c[ar, :, labels] * i[ar, :, labels]).topk(topk + 1, dim=1)
    gap = (t[:, topk - 1] - t[:, topk]) / t[:, topk - 1].clamp(min=1e-20)
    fb = bags5.reshape(N, U1 * U2, 5)

    def mean_box(sel_t, sel_i):
        w = sel_t / (sel_t.sum(1, keepdim=True) + 1e-8)
        return (fb[ar[:, None], sel_i] * w[..., None]).sum(1)
    keep = list(range(topk - 1)) + [topk]
    shift = (mean_box(t[:, :topk], idx[:, :topk]) - mean_box(t[:, keep], idx[:, keep])).abs().max(1)[0]
    return gap, shift


def mil_bag_loss_obb(cls, ins, valid, labels, neg_cls, neg_w):
    """:1252-1283 - the HBB bag loss with weights 0.25 (positive bags) / 0.75 (negatives)."""
    N, U1, U2, C = cls.shape
    v = valid.reshape(N, U1, U2, 1).to(cls.dtype)
    lab = labels[:, None].repeat(1, U1).reshape(-1)
    bag = (cls.sigmoid() * F.normalize(ins.softmax(2) * v, dim=2, p=1)).sum(2).reshape(-1, C)
    lw = (v.reshape(N * U1, U2, 1).sum(1) > 0).float()
    ns = max(float((lw.sum(-1) > 0).sum()), 1.0)
    loss = 0.25 * R.gfocal(bag, F.one_hot(lab, C).float(), lw).sum() / ns
    if neg_cls is not None:
        loss = loss + 0.75 * R.gfocal(neg_cls.sigmoid(), torch.zeros_like(neg_cls), neg_w.reshape(-1, 1).float()).sum() / ns
    return loss


# --------------------------------------------------------------- strong augmentation --
def rotate_image_nearest(img, angle):
    """torchvision.transforms.functional.rotate(img [C,H,W] float, angle, fill=0): NEAREST, no
    expand, centre of the image.  Published algorithm (torchvision/transforms/functional.py rotate ->
    _get_inverse_affine_matrix(center 0, -angle) -> _functional_tensor.rotate: _gen_affine_grid +
    grid_sample(mode nearest, zeros padding, align_corners False))."""
    C, H, W = img.shape
    rot = math.radians(angle)
    theta = torch.tensor([[math.cos(rot), -math.sin(rot), 0.0], [math.sin(rot), math.cos(rot), 0.0]], dtype=img.dtype)
    xs = torch.linspace(-W * 0.5 + 0.5, W * 0.5 + 0.5 - 1, steps=W)
    ys = torch.linspace(-H * 0.5 + 0.5, H * 0.5 + 0.5 - 1, steps=H)
    base = torch.stack([xs[None, :].expand(H, W), ys[:, None].expand(H, W), torch.ones(H, W)], -1).reshape(1, H * W, 3)
    grid = base.bmm((theta / torch.tensor([[0.5 * W], [0.5 * H]])).transpose(0, 1)[None]).reshape(1, H, W, 2)
    return F.grid_sample(img[None], grid, mode='nearest', padding_mode='zeros', align_corners=False)[0]


def strong_augmentation_geometry_obb(gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes, flip, angle,
                                     scale, H, W):
    """detectors/syn_images_generator_v2.py:217-343 - everything except the pixels, draws injected."""
    gp, gl = gt_points.clone(), gt_labels.clone()
    pb, pl, pp = R.obb2poly_le90(pseudo_bboxes.clone()), pseudo_labels.clone(), pseudo_points.clone()
    if flip in ('horizontal', 'diagonal'):
        pb[:, 0::2] = W - pb[:, 0::2]; pp[:, 0] = W - pp[:, 0]; gp[:, 0] = W - gp[:, 0]
    if flip in ('vertical', 'diagonal'):
        pb[:, 1::2] = H - pb[:, 1::2]; pp[:, 1] = H - pp[:, 1]; gp[:, 1] = H - gp[:, 1]
    cx, cy = W / 2, H / 2
    rad = np.deg2rad(-angle)
    ca, sa = np.cos(rad), np.sin(rad)
    tb, tp, tg = pb.clone(), pp.clone(), gp.clone()
    pb[:, 0::2] = ca * (tb[:, 0::2] - cx) - sa * (tb[:, 1::2] - cy) + cx
    pb[:, 1::2] = sa * (tb[:, 0::2] - cx) + ca * (tb[:, 1::2] - cy) + cy
    pp[:, 0] = ca * (tp[:, 0] - cx) - sa * (tp[:, 1] - cy) + cx
    pp[:, 1] = sa * (tp[:, 0] - cx) + ca * (tp[:, 1] - cy) + cy
    gp[:, 0] = ca * (tg[:, 0] - cx) - sa * (tg[:, 1] - cy) + cx
    gp[:, 1] = sa * (tg[:, 0] - cx) + ca * (tg[:, 1] - cy) + cy
    k = ((0 <= gp[:, 0]) & (gp[:, 0] < W) & (0 <= gp[:, 1]) & (gp[:, 1] < H)).nonzero().reshape(-1)
    gp, gl = gp[k], gl[k]
    k = ((0 <= pp[:, 0]) & (pp[:, 0] < W) & (0 <= pp[:, 1]) & (pp[:, 1] < H)).nonzero().reshape(-1)
    pp, pl, pb = pp[k], pl[k], pb[k]
    sH, sW = int(H * scale), int(W * scale)
    if scale < 1.0:
        bh, bw = int((H - sH) / 2), int((W - sW) / 2)
    else:
        bh, bw = int((sH - H) / 2), int((sW - W) / 2)
    pb, pp, gp = pb * scale, pp * scale, gp * scale
    if scale >= 1.0:
        k = ((gp[:, 0] >= bw) & (gp[:, 0] < W + bw) & (gp[:, 1] >= bh) & (gp[:, 1] < H + bh)).nonzero().reshape(-1)
        gp, gl = gp[k], gl[k]
        gp[:, 0] -= bw; gp[:, 1] -= bh
        k = ((pp[:, 0] >= bw) & (pp[:, 0] < W + bw) & (pp[:, 1] >= bh) & (pp[:, 1] < H + bh)).nonzero().reshape(-1)
        pb, pp, pl = pb[k], pp[k], pl[k]
        pp[:, 0] -= bw; pp[:, 1] -= bh
        pb[:, 0::2] -= bw; pb[:, 1::2] -= bh
    else:
        gp[:, 0] += bw; gp[:, 1] += bh
        pp[:, 0] += bw; pp[:, 1] += bh
        pb[:, 0::2] += bw; pb[:, 1::2] += bh
    pb = poly2obb_le90(pb) if pb.shape[0] else torch.empty(0, 5)
    return gp, gl, pp, pl, pb


def strong_augmentation_single_obb(img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes, flip, angle,
                                   scale):
    C, H, W = img.shape
    geo = strong_augmentation_geometry_obb(gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes, flip,
                                           angle, scale, H, W)
    if flip in ('horizontal', 'diagonal'):
        img = torch.flip(img, dims=[2])
    if flip in ('vertical', 'diagonal'):
        img = torch.flip(img, dims=[1])
    img = rotate_image_nearest(img, angle)
    sH, sW = int(H * scale), int(W * scale)
    r = F.interpolate(img[None], size=(sH, sW), mode='bilinear', align_corners=False)[0]
    if scale < 1.0:
        out = torch.zeros_like(img)
        sy, sx = (H - sH) // 2, (W - sW) // 2
        out[:, sy:sy + sH, sx:sx + sW] = r
    else:
        sy, sx = (sH - H) // 2, (sW - W) // 2
        out = r[:, sy:sy + H, sx:sx + W]
    return (torch.round(out),) + geo


# ------------------------------------------------------------------------- the network --
def init_detector_state_obb(seed=0, num_stages=1):
    """Random weights with the shapes/names of `RotatedFCOS_Student` (R50 pytorch style with trainable
    BN affine + FPN + PSAGG + GN head with an angle branch)."""
    sd = M.init_detector_state(seed, num_stages)
    sd = {k: v for k, v in sd.items() if not k.startswith('bbox_head.')}
    sd['backbone.conv1.weight'] = sd['backbone.conv1.weight'] * 64.0           # inputs are mean/std-normalised here
    g = torch.Generator().manual_seed(seed + 1000)
    for br in ('cls_convs', 'reg_convs'):
        for i in range(4):
            sd[f'bbox_head.{br}.{i}.conv.weight'] = torch.randn(256, 256, 3, 3, generator=g) * 0.01
            sd[f'bbox_head.{br}.{i}.gn.weight'] = torch.ones(256)
            sd[f'bbox_head.{br}.{i}.gn.bias'] = torch.zeros(256)
    for name, co in (('conv_cls', NUM_CLASSES), ('conv_reg', 4), ('conv_centerness', 1), ('conv_angle', 1)):
        sd[f'bbox_head.{name}.weight'] = torch.randn(co, 256, 3, 3, generator=g) * 0.01
        sd[f'bbox_head.{name}.bias'] = torch.zeros(co)
    sd['bbox_head.conv_cls.bias'] = torch.full((NUM_CLASSES,), -math.log((1 - 0.01) / 0.01))
    sd['bbox_head.scales.0.scale'] = torch.tensor(1.0)
    sd['bbox_head.scale_angle.scale'] = torch.tensor(1.0)

    def lin(name, cout, cin, scale=1.0):
        b = 1.0 / math.sqrt(cin)
        sd[name + '.weight'] = (torch.rand(cout, cin, generator=g) * 2 - 1) * b * scale
        sd[name + '.bias'] = (torch.rand(cout, generator=g) * 2 - 1) * b * (0.0 if scale != 1.0 else 1.0)
    for fam in ('shared_fcs', 'shared_fcs_refine'):
        lin(f'bbox_head.{fam}.0', 1024, 12544); lin(f'bbox_head.{fam}.1', 1024, 1024)
    for s in range(num_stages):
        for fam in ('shared_fcs_bag', 'shared_fcs_reg'):
            lin(f'bbox_head.{fam}.{s}.0', 1024, 12544); lin(f'bbox_head.{fam}.{s}.1', 1024, 1024)
        lin(f'bbox_head.fc_cls.{s}', NUM_CLASSES, 1024); lin(f'bbox_head.fc_ins.{s}', NUM_CLASSES, 1024)
        lin(f'bbox_head.fc_reg.{s}', 4, 1024, scale=0.01)
        # (no fc_iou: the oriented head builds none, rotated_fcos_head_p2rb_ts.py:243-283)
    return sd


def trainable_obb(name):
    """norm_cfg requires_grad=True + frozen_stages=1: everything trains except the stem and layer1
    (resnet.py _freeze_stages); BatchNorm runs on its running statistics (norm_eval=True)."""
    if not M.is_param(name):
        return False
    if name.startswith('backbone.conv1') or name.startswith('backbone.bn1') or name.startswith('backbone.layer1.'):
        return False
    return True


def resnet50_pytorch(sd, x, pre='backbone'):
    """mmdet backbones/resnet.py with style='pytorch' (stride on the 3x3), BN in eval mode."""
    x = F.max_pool2d(F.relu(M._bn_eval(sd, pre + '.bn1', F.conv2d(x, sd[pre + '.conv1.weight'], None, 2, 3))), 3, 2, 1)
    outs = []
    for li, nb in enumerate((3, 4, 6, 3)):
        for j in range(nb):
            p = f'{pre}.layer{li + 1}.{j}'
            s = 2 if (j == 0 and li > 0) else 1
            idt = x
            o = F.relu(M._bn_eval(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'])))
            o = F.relu(M._bn_eval(sd, p + '.bn2', F.conv2d(o, sd[p + '.conv2.weight'], None, s, 1)))
            o = M._bn_eval(sd, p + '.bn3', F.conv2d(o, sd[p + '.conv3.weight']))
            if j == 0:
                idt = M._bn_eval(sd, p + '.downsample.1', F.conv2d(x, sd[p + '.downsample.0.weight'], None, s))
            x = F.relu(o + idt)
        outs.append(x)
    return outs


def extract_feat(sd, img):
    return M.fpn_psagg(sd, resnet50_pytorch(sd, img))


def head_forward_obb(sd, feat):
    """:356-391 over rotated_anchor_free_head.py forward_single: conv(no bias)+GN(32)+ReLU towers, norm_on_bbox,
    centerness from the CLASSIFICATION tower (centerness_on_reg defaults to False), scaled angle."""
    c = r = feat
    for i in range(4):
        for br in ('cls_convs', 'reg_convs'):
            x = c if br == 'cls_convs' else r
            x = F.conv2d(x, sd[f'bbox_head.{br}.{i}.conv.weight'], None, 1, 1)
            x = F.relu(F.group_norm(x, 32, sd[f'bbox_head.{br}.{i}.gn.weight'], sd[f'bbox_head.{br}.{i}.gn.bias'], 1e-5))
            if br == 'cls_convs':
                c = x
            else:
                r = x
    cls = F.conv2d(c, sd['bbox_head.conv_cls.weight'], sd['bbox_head.conv_cls.bias'], 1, 1)
    reg = F.conv2d(r, sd['bbox_head.conv_reg.weight'], sd['bbox_head.conv_reg.bias'], 1, 1)
    ctr = F.conv2d(c, sd['bbox_head.conv_centerness.weight'], sd['bbox_head.conv_centerness.bias'], 1, 1)
    ang = F.conv2d(r, sd['bbox_head.conv_angle.weight'], sd['bbox_head.conv_angle.bias'], 1, 1)
    reg = (reg * sd['bbox_head.scales.0.scale']).float().clamp(min=0) * STRIDE
    ang = (ang * sd['bbox_head.scale_angle.scale']).float()
    return cls, reg, ang, ctr


# ------------------------------------------------------------------------ train step --
def _rroi(boxes_list):
    return torch.cat([torch.cat([torch.full((b.shape[0], 1), float(i)), b[:, :5]], 1) for i, b in enumerate(boxes_list)])


def _rra(feat, rois):
    return R.roi_align_rotated(feat, rois, 7, 1.0 / STRIDE, 2, True, True)


def mil_stage_obb(sd, feat, stage, cfg, ext_cfg, pb_t, gb_t, pl_t, img_hw, neg_u, alpha, topk, beta, syn=None, stats=None):
    """One MIL stage (detectors/rotated_fcos_teacher_student.py:473-506 + dense_heads :1285-1448).
    `stats` (dict, optional) receives the oracle's own merged boxes and the rounding margins of the selection
    (mil_select_margin)."""
    losses = {}
    B = len(pb_t)
    U1 = len(cfg['base_ratios']) ** 2 * (1 + 4 * len(cfg['shake_ratio'] or []))
    U2 = len(ext_cfg['base_ratios']) ** 2 * (1 + 4 * len(ext_cfg['shake_ratio'] or []))

    def reg_branch(x, boxes_list, real_list):
        props = [fine_proposals_obb(b, cfg, img_hw)[0] for b in boxes_list]
        ext = [fine_proposals_obb(p, ext_cfg, img_hw) for p in props]
        bags = torch.cat([e[0] for e in ext])
        valid = torch.cat([e[1] for e in ext]).reshape(-1)
        ref = torch.cat([b[:, None].repeat(1, U1 * U2, 1).reshape(-1, 5) for b in boxes_list])
        real = torch.cat([b[:, None].repeat(1, U1 * U2, 1).reshape(-1, 5) for b in real_list])
        f = _rra(x, _rroi([e[0] for e in ext])).flatten(1)
        pred = R.delta2bbox(R.bbox_cxcywh_to_xyxy(bags[:, :4]),
                            M._fc(sd, f'bbox_head.fc_reg.{stage}', M._fc_stack(sd, 'shared_fcs_reg', stage, f)),
                            (img_hw[0], img_hw[1], 3))
        loss = R.iou_loss_module(pred, R.bbox_cxcywh_to_xyxy(ref[:, :4]), weight=valid.float(),
                                 avg_factor=float(pred.shape[0]), kind='dn', hyper=0.2)
        refined = torch.cat([R.bbox_xyxy_to_cxcywh(pred.detach()), bags[:, 4:5]], 1)
        return loss, refined, valid, bags, real, [e[0].shape[0] for e in ext], props
    if syn is None:
        loss_bbox, refined, valid, bags, real, sizes, props = reg_branch(feat, pb_t, gb_t)
    else:
        loss_bbox = reg_branch(syn[0], syn[1], syn[1])[0]
        _, refined, valid, bags, real, sizes, props = reg_branch(feat, pb_t, gb_t)
    losses[f'stage{stage}_loss_mil_bbox'] = loss_bbox * alpha[0]
    losses[f'stage{stage}_coarse_bags_iou'] = R.box_iou_rotated(bags, real, aligned=True).mean()
    losses[f'stage{stage}_refine_bags_iou'] = R.box_iou_rotated(refined, real, aligned=True).mean()
    f = M._fc_stack(sd, 'shared_fcs_bag', stage, _rra(feat, _rroi(list(torch.split(refined, sizes)))).flatten(1))
    N = sum(b.shape[0] for b in pb_t)
    cls = M._fc(sd, f'bbox_head.fc_cls.{stage}', f).view(N, U1, U2, -1)
    ins = M._fc(sd, f'bbox_head.fc_ins.{stage}', f).view(N, U1, U2, -1)
    negs = [negative_proposals_obb(neg_u[i], props[i], img_hw) for i in range(B)]
    ncls = M._fc(sd, f'bbox_head.fc_cls.{stage}',
                 M._fc_stack(sd, 'shared_fcs_bag', stage, _rra(feat, _rroi([n[0] for n in negs])).flatten(1)))
    labels = torch.cat(pl_t)
    losses[f'stage{stage}_loss_mil_bags'] = mil_bag_loss_obb(cls, ins, valid[:, None], labels, ncls,
                                                             torch.cat([n[1] for n in negs])) * alpha[1]
    merged = mil_bag_select_obb(cls, ins, valid[:, None], labels, refined, torch.cat(pb_t), img_hw, topk, beta)
    if stats is not None:
        stats[f'merged{stage}'] = merged.clone()
        stats[f'gap{stage}'], stats[f'shift{stage}'] = mil_select_margin(cls, ins, valid[:, None], labels, refined, topk)
    return losses, list(torch.split(merged, [b.shape[0] for b in pb_t]))


def _teacher_pseudo(sd_t, img, pts, gt_points, gt_labels, gt_bboxes):
    with torch.no_grad():
        cls, reg, ang, ctr = head_forward_obb(sd_t, extract_feat(sd_t, img))
        fc, fr, _ = _flatten_head5(cls, reg, ang, ctr)
        return [generate_pseudo_single_obb(pts, fc[i], fr[i], gt_points[i], gt_labels[i], gt_bboxes[i], 0.0)
                for i in range(img.shape[0])]


def _aug_loss(sd_s, img, gt_points, gt_labels, pp, pl, pb, aug, pts):
    flips, angles, scales = aug
    outs = [strong_augmentation_single_obb(img[i], gt_points[i], gt_labels[i], pp[i], pl[i], pb[i], flips[i], angles[i],
                                           scales[i]) for i in range(img.shape[0])]
    cls, reg, ang, ctr = head_forward_obb(sd_s, extract_feat(sd_s, torch.stack([o[0] for o in outs])))
    return loss_pseudo_obb(cls, reg, ang, ctr, pts, [o[1] for o in outs], [o[2] for o in outs], [o[4] for o in outs],
                           [o[5] for o in outs])


def _points_distance(new_points, gt_bboxes):
    real = torch.cat(gt_bboxes)
    return (torch.sqrt((torch.cat(new_points) - real[:, :2]) ** 2) / torch.sqrt((real[:, 2:4] / 2) ** 2)).mean()


def forward_train_step2(sd_s, sd_t, img, gt_bboxes, gt_labels, gt_points, cfg, inject):
    """detectors/rotated_fcos_teacher_student.py:207-252.  inject: neg0 [B,5,n], aug=(flips, angles, scales);
    optional `stats`, see mil_stage_obb."""
    B, _, H, W = img.shape
    pts = M.grid_points(H // STRIDE, W // STRIDE)
    losses = {}
    pseudo = _teacher_pseudo(sd_t, img, pts, gt_points, gt_labels, gt_bboxes)
    pb_c, pp_c = [p[0] for p in pseudo], [p[1] for p in pseudo]
    feat = extract_feat(sd_s, img)
    n = cfg['num_training_burninstep2']
    pb_t, gb_t, pl_t = [b[:n].clone() for b in pb_c], [b[:n] for b in gt_bboxes], [l[:n] for l in gt_labels]
    losses['coarse_bboxes_iou'] = R.box_iou_rotated(torch.cat(pb_t), torch.cat(gb_t), aligned=True).mean()
    for stage in range(cfg['num_stages']):
        ml, pb_t = mil_stage_obb(sd_s, feat, stage, cfg['fine_proposal_cfg'][stage], cfg['fine_proposal_extensive_cfg'][stage],
                                 pb_t, gb_t, pl_t, (H, W), inject[f'neg{stage}'], cfg['alpha'], cfg['top_k'], cfg['beta'],
                                 stats=inject.get('stats'))
        losses[f'stage{stage}_refine_bboxes_iou'] = R.box_iou_rotated(torch.cat(pb_t), torch.cat(gb_t), aligned=True).mean()
        losses.update(ml)
    pb_r, pp_r = [b.clone() for b in pb_c], [p.clone() for p in pp_c]
    for i in range(B):
        k = pb_t[i].shape[0]
        pb_r[i][:k] = pb_t[i]
        pp_r[i][:k] = pb_t[i][:, :2]
    new_points = [(1 - cfg['lamda']) * pb_r[i][:, :2] + cfg['lamda'] * gt_points[i] for i in range(B)]
    losses['refined_points_distance'] = _points_distance(new_points, gt_bboxes)
    lc, lb, lt = _aug_loss(sd_s, img, new_points, gt_labels, pp_r, gt_labels, pb_r, inject['aug'], pts)
    losses['loss_cls'], losses['loss_bbox'], losses['loss_centerness'] = lc, lb, lt
    return losses, new_points


def forward_train_step1(sd_s, sd_t, img, gt_bboxes, gt_labels, gt_points, cfg, inject):
    """detectors/rotated_fcos_teacher_student.py:149-205; the synthetic image / boxes are generated from
    inject['syn'] (per-image draw dictionaries, ref_ops.generate_black_paper(oriented=True)) or given as inputs
    (inject['img_syn'], inject['syn_boxes'] [n_i,5])."""
    B, _, H, W = img.shape
    pts = M.grid_points(H // STRIDE, W // STRIDE)
    losses = {}
    if 'syn' in inject:
        prior = torch.tensor(cfg['shape_list'], dtype=torch.float32)
        gen = [R.generate_black_paper(img[i], gt_bboxes[i], prior, inject['syn'][i], oriented=True) for i in range(B)]
        img_syn, syn_boxes = torch.stack([g[0] for g in gen]), [g[1] for g in gen]
    else:
        img_syn, syn_boxes = inject['img_syn'], inject['syn_boxes']
    feat_all = extract_feat(sd_s, torch.cat([img_syn, img]))
    feat_syn, feat_ori = feat_all[:B], feat_all[B:]
    _, reg_s, ang_s, ctr_s = head_forward_obb(sd_s, feat_syn)
    loss_bbox, loss_ctr = loss_synthetic_obb(reg_s, ang_s, ctr_s, pts, syn_boxes)
    pseudo = _teacher_pseudo(sd_t, img, pts, gt_points, gt_labels, gt_bboxes)
    pb_c, pp_c = [p[0] for p in pseudo], [p[1] for p in pseudo]
    n = cfg['num_training_burninstep1']
    syn_t = [s[:n] for s in syn_boxes]
    pb_t, gb_t, pl_t = [b[:n] for b in pb_c], [b[:n] for b in gt_bboxes], [l[:n] for l in gt_labels]
    losses['coarse_bboxes_iou'] = R.box_iou_rotated(torch.cat(pb_t), torch.cat(gb_t), aligned=True).mean()
    for stage in range(cfg['num_stages']):
        ml, pb_t = mil_stage_obb(sd_s, feat_ori, stage, cfg['fine_proposal_cfg'][stage],
                                 cfg['fine_proposal_extensive_cfg'][stage], pb_t, gb_t, pl_t, (H, W), inject[f'neg{stage}'],
                                 cfg['alpha'], cfg['top_k'], cfg['beta'], syn=(feat_syn, syn_t),
                                 stats=inject.get('stats'))
        losses[f'stage{stage}_refine_bboxes_iou'] = R.box_iou_rotated(torch.cat(pb_t), torch.cat(gb_t), aligned=True).mean()
        losses.update(ml)
    new_points = [(1 - cfg['lamda']) * pb_c[i][:, :2] + cfg['lamda'] * gt_points[i] for i in range(B)]
    losses['refined_points_distance'] = _points_distance(new_points, gt_bboxes)
    lc, _, _ = _aug_loss(sd_s, img, new_points, gt_labels, pp_c, gt_labels, pb_c, inject['aug'], pts)
    losses['loss_cls'], losses['loss_bbox'], losses['loss_centerness'] = lc, loss_bbox, loss_ctr
    return losses, new_points


MODEL_CFG = dict(num_stages=1, num_training_burninstep1=100, num_training_burninstep2=100, lamda=1.0, alpha=[0.01, 0.25],
                 top_k=3, beta=0.25,
                 shape_list=[[20, 20, 0.5, 0.5], [10, 20, 0.5, 0.5], [10, 30, 0.5, 0.5], [40, 20, 0.5, 0.5], [30, 10, 0.5, 0.5],
                             [20, 50, 0.5, 0.5], [30, 20, 0.5, 0.5], [35, 40, 0.6, 0.5]],   # sodaa_fcos_pointteacher_1x.py:22-24
                 fine_proposal_cfg=[dict(base_ratios=[1.0], shake_ratio=None, min_scale=0)],
                 fine_proposal_extensive_cfg=[dict(base_ratios=[1.0, 1.2, 1.3, 0.8, 0.6], shake_ratio=None, min_scale=4)])


def cpu_baseline(workload='step2', batch=2, size=1200, objects=300, sample_div=4):
    """Time ONE iteration of this oracle (forward + backward + clip + SGD + EMA) on the host cores for bench.py's `cpu_baseline` of
    the oriented variant.  The oracle evaluates RoIAlignRotated, rotated IoU and their gradients RoI by RoI in python, so the
    full-size iteration (1200 x 1200, ~300 objects) takes minutes; the timed SAMPLE keeps the object density and divides the
    pixels AND the objects by `sample_div` - every part of the iteration (convolutions ~ pixels; bags, RoIs, rotated IoU ~ objects)
    shrinks by that factor - and the reported value is the sample's rate / sample_div."""
    import time
    cores = M.effective_cpus()
    torch.set_num_threads(cores)
    s_size = int(round(size / math.sqrt(sample_div) / 32)) * 32
    s_obj = max(1, int(round(objects / sample_div)))
    g = torch.Generator().manual_seed(0)
    sd_s, sd_t = init_detector_state_obb(1, 1), init_detector_state_obb(2, 1)
    if workload == 'step2':
        for sd in (sd_s, sd_t):
            sd['bbox_head.conv_reg.bias'] = torch.ones(4)

    def one(size_, objects_):
        nonlocal sd_t
        img = (torch.rand(batch, 3, size_, size_, generator=g) * 2 - 1)                      # mean / std normalised pixels
        boxes, labels = [], []
        for _ in range(batch):
            c = torch.rand(objects_, 2, generator=g) * (size_ - 64) + 32
            wh = torch.exp(torch.randn(objects_, 2, generator=g) * 0.4 + math.log(14.0)).clamp(5, 36)
            a = (torch.rand(objects_, 1, generator=g) - 0.5) * math.pi * 0.98
            boxes.append(torch.cat([c, wh, a], 1))
            labels.append(torch.randint(0, NUM_CLASSES, (objects_,), generator=g))
        points = [b[:, :2] for b in boxes]
        inject = dict(neg0=torch.rand(batch, 5, 200, generator=g), aug=((['horizontal', 'None'] * batch)[:batch], ([0, 90] * batch)[:batch],
                                                                      ([0.9, 1.1] * batch)[:batch]))
        params = {k: (v.clone().requires_grad_(True) if trainable_obb(k) else v) for k, v in sd_s.items()}
        sd_t = M.ema(sd_t, sd_s)
        if workload == 'step1':
            inject.update(img_syn=img.clone(), syn_boxes=[b[: max(1, len(b) // 2)].clone() for b in boxes])
            losses, _ = forward_train_step1(params, sd_t, img, boxes, labels, points, dict(MODEL_CFG), inject)
        else:
            losses, _ = forward_train_step2(params, sd_t, img, boxes, labels, points, dict(MODEL_CFG), inject)
        loss = M.total_loss(losses)
        names = [k for k in params if trainable_obb(k)]
        gr = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
        M.sgd_clip_step(sd_s, dict(zip(names, gr)), {}, 0.005 / 3, True)
        return float(loss)
    one(256, 8)                                   # warm-up (allocator, thread pools)
    t0 = time.perf_counter()
    one(s_size, s_obj)
    dt = time.perf_counter() - t0
    return dict(value=round(1.0 / (dt * sample_div), 6), unit='iters/s', cores=cores, kind='port',
                sample=f'ONE {workload} iteration of oracle/ref_obb.py (torch-CPU fp32, {cores} threads; RoIAlignRotated / rotated IoU RoI by RoI in '
                       f'python) on a 1/{sample_div} sample: bs {batch}, {s_size}x{s_size}, {s_obj} objects/img (same object density as '
                       f'{size}x{size} with {objects}) took {dt:.1f} s; value = sample rate / {sample_div} (pixels and objects both scale the work)')
