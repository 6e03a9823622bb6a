"""OBB twins of proposals.py, mirroring
/root/reference/OBB_TOD/mmrotate/models/detectors/syn_images_generator_v2.py (+ the pure-torch
helpers of mmrotate/core/bbox/transforms.py): MIL bags around oriented pseudo boxes, rotated
negatives, the flip + rotate + rescale strong augmentation and the white-rectangle generator
with the image-maximum fill.  Boxes are (cx, cy, w, h, angle[rad]) in the le90 convention.

As in proposals.py, sizes the reference discovers with `.nonzero()` are carried as validity
masks (no host synchronisation); the list-returning forms filter for API parity.
"""
import math
import random

import numpy as np
import torch
import torch.nn.functional as TF

from . import functional as F
from .core import bbox_cxcywh_to_xyxy, bbox_xyxy_to_cxcywh
from .obb import norm_angle, rbbox_overlaps
from .proposals import fine_proposals_from_cfg, generate_black_paper_batch, obb2poly_le90


# ------------------------------------------------------------------ transforms --
def poly2obb_le90(polys):
    """core/bbox/transforms.py:301-331"""
    polys = torch.reshape(polys, [-1, 8])
    pt1, pt2, pt3, pt4 = polys[..., :8].chunk(4, 1)
    edge1 = torch.sqrt(torch.pow(pt1[..., 0] - pt2[..., 0], 2) + torch.pow(pt1[..., 1] - pt2[..., 1], 2))
    edge2 = torch.sqrt(torch.pow(pt2[..., 0] - pt3[..., 0], 2) + torch.pow(pt2[..., 1] - pt3[..., 1], 2))
    angles1 = torch.atan2(pt2[..., 1] - pt1[..., 1], pt2[..., 0] - pt1[..., 0])
    angles2 = torch.atan2(pt4[..., 1] - pt1[..., 1], pt4[..., 0] - pt1[..., 0])
    angles = norm_angle(torch.where(edge1 > edge2, angles1, angles2), 'le90')
    x_ctr = (pt1[..., 0] + pt3[..., 0]) / 2.0
    y_ctr = (pt1[..., 1] + pt3[..., 1]) / 2.0
    return torch.stack([x_ctr, y_ctr, torch.max(edge1, edge2), torch.min(edge1, edge2), angles], 1)


def poly2obb(polys, version='le90'):
    """core/bbox/transforms.py:93-113"""
    assert version == 'le90', "the Point-Teacher OBB config uses angle_version='le90'"
    return poly2obb_le90(polys)


def obb2poly(rboxes, version='le90'):
    """core/bbox/transforms.py:139-158"""
    assert version == 'le90', "the Point-Teacher OBB config uses angle_version='le90'"
    return obb2poly_le90(rboxes)


# ------------------------------------------------------------------- MIL bags --
def MIL_gen_proposals_from_cfg(pseudo_points, pseudo_boxes_obb, fine_proposal_cfg, gt_boxes_obb, img_meta):
    """syn_images_generator_v2.py:24-39: the bags are built around the axis-aligned (cx,cy,w,h) part
    of each oriented box and get its angle re-attached."""
    assert fine_proposal_cfg['gen_mode'] == 'refine', "only gen_mode='refine' is used by the shipped configs"
    hbb = [bbox_cxcywh_to_xyxy(b[:, :4]) for b in pseudo_boxes_obb]
    proposals_list, proposals_valid_list = fine_proposals_from_cfg(hbb, fine_proposal_cfg, img_meta)
    ratios, shake = fine_proposal_cfg['base_ratios'], fine_proposal_cfg['shake_ratio']
    num_aug = len(ratios) ** 2 * (1 + 4 * len(shake or []))
    ref, real = [], []
    for i in range(len(pseudo_boxes_obb)):
        ref.append(pseudo_boxes_obb[i].unsqueeze(1).repeat(1, num_aug, 1).reshape(-1, 5))
        real.append(gt_boxes_obb[i].unsqueeze(1).repeat(1, num_aug, 1).reshape(-1, 5))
        angle = pseudo_boxes_obb[i][:, -1].reshape(-1, 1).unsqueeze(1).repeat(1, num_aug, 1).reshape(-1, 1)
        proposals_list[i] = torch.cat([bbox_xyxy_to_cxcywh(proposals_list[i]), angle], dim=1)
    return proposals_list, proposals_valid_list, ref, real


def gen_negative_proposals(gt_points, proposal_cfg, aug_generate_proposals, img_meta, uniforms=None):
    """syn_images_generator_v2.py:129-156.  The five uniform draws become the oriented box
    (x1, y1, x1 + 200 u, y1 + 200 u, theta) - the reference stacks corner-style numbers into the
    (cx, cy, w, h, a) slots, which is kept - and a negative counts when its rotated IoU with every
    positive bag of the image is below 0.3.  `uniforms` [B,5,n] injects the draws (tests)."""
    n = proposal_cfg['gen_num_neg']
    if n == 0:
        return None, None
    B = len(gt_points)
    dev = aug_generate_proposals[0].device
    u = uniforms.to(dev) if uniforms is not None else torch.rand(B, 5, n, device=dev)
    negs, oks = [], []
    for i in range(B):
        h, w, _ = img_meta[i]['img_shape']
        x1 = u[i, 0] * w * 0.8
        y1 = u[i, 1] * h * 0.8
        neg = torch.stack([x1, y1, x1 + u[i, 2] * 200, y1 + u[i, 3] * 200, u[i, 4] * math.pi - math.pi / 2], dim=1)
        pos = aug_generate_proposals[i]
        if pos.shape[0]:
            ok = (rbbox_overlaps(neg, pos) < 0.3).all(dim=1)
        else:
            ok = torch.ones(n, dtype=torch.bool, device=dev)
        negs.append(neg)
        oks.append(ok)
    return negs, oks


# ------------------------------------------------------------ strong augmentation --
def draw_strong_aug_params(B):
    """The host-side draws of strong_augmentation (:239, :262, :291), one (flip, angle, scale) per image."""
    flips, angles, scales = [], [], []
    for _ in range(B):
        flips.append(random.choice(['horizontal', 'vertical', 'diagonal', 'None']))
        angles.append(int(np.random.randint(1, 20)))
        scales.append(float(np.around(np.random.uniform(0.8, 1.2), 1)))
    return flips, angles, scales


def rotate_image_nearest(img, angle):
    """torchvision.transforms.functional.rotate(img, angle, fill=0) for a float [C,H,W] tensor
    (interpolation NEAREST, expand False, centre = image centre): the inverse affine matrix of a
    counter-clockwise rotation by `angle` degrees, torchvision's half-pixel-centred base grid and
    grid_sample(nearest, zeros, align_corners=False).  torchvision is not in this image - the
    published algorithm is restated (torchvision/transforms/_functional_tensor.py rotate,
    _gen_affine_grid, _apply_grid_transform)."""
    C, H, W = img.shape
    rot = math.radians(angle)
    # _get_inverse_affine_matrix(center=0, angle=-angle, translate=0, scale=1, shear=0) = [[cos, -sin, 0], [sin, cos, 0]];
    # theta is folded into scalar multipliers (a device tensor built from host numbers would be a blocking upload):
    # grid = base @ (theta^T / [W/2, H/2]) with base = (x, y, 1) on torchvision's half-pixel-centred lattice
    xs = torch.linspace(-W * 0.5 + 0.5, W * 0.5 + 0.5 - 1, steps=W, dtype=img.dtype, device=img.device)[None, :]
    ys = torch.linspace(-H * 0.5 + 0.5, H * 0.5 + 0.5 - 1, steps=H, dtype=img.dtype, device=img.device)[:, None]
    c, s_ = math.cos(rot), math.sin(rot)
    gx = xs * (c / (0.5 * W)) + ys * (-s_ / (0.5 * W))
    gy = xs * (s_ / (0.5 * H)) + ys * (c / (0.5 * H))
    grid = torch.stack([gx, gy], dim=-1).unsqueeze(0)
    return TF.grid_sample(img.unsqueeze(0), grid, mode='nearest', padding_mode='zeros', align_corners=False).squeeze(0)


def _aug_image(img, flip, angle, scale):
    C, H, W = img.shape
    if flip == 'horizontal':
        img = torch.flip(img, dims=[2])
    elif flip == 'vertical':
        img = torch.flip(img, dims=[1])
    elif flip == 'diagonal':
        img = torch.flip(img, dims=[1, 2])
    img = rotate_image_nearest(img, angle)
    sH, sW = int(H * scale), int(W * scale)
    r = TF.interpolate(img.unsqueeze(0), size=(sH, sW), mode='bilinear', align_corners=False).squeeze(0)
    if scale < 1.0:
        out = torch.zeros_like(img)
        sy, sx = (H - sH) // 2, (W - sW) // 2
        out[:, sy:sy + sH, sx:sx + sW] = r
    else:
        sy, sx = (sH - H) // 2, (sW - W) // 2
        out = r[:, sy:sy + H, sx:sx + W]
    return torch.round(out)


def _aug_xy(xy, flip, angle, scale, H, W):
    """[n, 2k] interleaved x,y -> (after flip+rotation, after scaling, final) following :241-316."""
    xy = xy.clone()
    if flip in ('horizontal', 'diagonal'):
        xy[:, 0::2] = W - xy[:, 0::2]
    if flip in ('vertical', 'diagonal'):
        xy[:, 1::2] = H - xy[:, 1::2]
    rad = np.deg2rad(-angle)
    cos_a, sin_a = float(np.cos(rad)), float(np.sin(rad))
    cx, cy = W / 2, H / 2
    rot = xy.clone()
    rot[:, 0::2] = cos_a * (xy[:, 0::2] - cx) - sin_a * (xy[:, 1::2] - cy) + cx
    rot[:, 1::2] = sin_a * (xy[:, 0::2] - cx) + cos_a * (xy[:, 1::2] - cy) + cy
    sH, sW = int(H * scale), int(W * scale)
    if scale < 1.0:
        bh, bw = int((H - sH) / 2), int((W - sW) / 2)
    else:
        bh, bw = int((sH - H) / 2), int((sW - W) / 2)
    scaled = rot * scale
    out = scaled.clone()
    if scale >= 1.0:
        out[:, 0::2] -= bw
        out[:, 1::2] -= bh
    else:
        out[:, 0::2] += bw
        out[:, 1::2] += bh
    return rot, scaled, out, (bh, bw)


def _inside_masks(rot, scaled, scale, H, W, bh, bw):
    m = (0 <= rot[:, 0]) & (rot[:, 0] < W) & (0 <= rot[:, 1]) & (rot[:, 1] < H)              # :277, :281
    if scale >= 1.0:                                                                          # :303, :309
        m = m & (scaled[:, 0] >= bw) & (scaled[:, 0] < W + bw) & (scaled[:, 1] >= bh) & (scaled[:, 1] < H + bh)
    return m


def strong_augmentation_images(img, params=None):
    """The pixel half of strong_augmentation (depends only on the image and the draws), see
    proposals.strong_augmentation_images.  Returns (params, list of [C,H,W])."""
    B = img.shape[0]
    params = params if params is not None else draw_strong_aug_params(B)
    return params, [_aug_image(img[i], params[0][i], params[1][i], params[2][i]) for i in range(B)]


def strong_augmentation_masked(img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes, angle_version,
                               params=None, imgs=None):
    """Sync-free strong_augmentation: nothing is filtered; `gt_valid` / `pseudo_valid` say which
    entries the reference would have kept (order preserved).  `imgs`: precomputed augmented images."""
    B, C, H, W = img.shape
    flips, angles, scales = params if params is not None else draw_strong_aug_params(B)
    pre = imgs
    imgs, gp_l, pp_l, pb_l, gv_l, pv_l = [], [], [], [], [], []
    for i in range(B):
        f, a, s = flips[i], angles[i], scales[i]
        imgs.append(pre[i] if pre is not None else _aug_image(img[i], f, a, s))
        rot, scaled, out, (bh, bw) = _aug_xy(gt_points[i], f, a, s, H, W)
        gp_l.append(out)
        gv_l.append(_inside_masks(rot, scaled, s, H, W, bh, bw))
        rot, scaled, out, _ = _aug_xy(pseudo_points[i], f, a, s, H, W)
        pp_l.append(out)
        pv_l.append(_inside_masks(rot, scaled, s, H, W, bh, bw))
        poly = _aug_xy(obb2poly(pseudo_bboxes[i], angle_version), f, a, s, H, W)[2]
        pb_l.append(poly2obb(poly, angle_version) if poly.shape[0] else pseudo_bboxes[i].new_zeros((0, 5)))
    return torch.stack(imgs, 0), imgs, gp_l, gt_labels, pp_l, pseudo_labels, pb_l, gv_l, pv_l


def strong_augmentation(img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes, angle_version,
                        params=None):
    """syn_images_generator_v2.py:217-343 with the reference's return convention (filtered lists)."""
    out = strong_augmentation_masked(img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes,
                                     angle_version, params)
    imgs, iml, gp, gl, pp, pl, pb, gv, pv = out
    gp = [p[m] for p, m in zip(gp, gv)]
    gl = [l[m] for l, m in zip(gl, gv)]
    pp = [p[m] for p, m in zip(pp, pv)]
    pl = [l[m] for l, m in zip(pl, pv)]
    pb = [b[m] for b, m in zip(pb, pv)]
    return imgs, iml, gp, gl, pp, pl, pb


# ------------------------------------------------- burn-in step 1: white rectangles --
def generate_black_paper_obb_batch(imgs, gt_bboxes, prior_size, dense_cls, imgsize, draws=None):
    """syn_images_generator_v2.py:625-724 for a batch of normalised images [C,H,W] whose real objects are the
    oriented boxes gt_bboxes[i] [G,5]: the same generator as the HBB one (the real objects only
    contribute their centres) except that the rectangles are painted with the image maximum
    (:722) and come back as oriented boxes `bb[:, :5]`.
    Returns per image (img_syn, syn_obb [M,5], alive [M])."""
    xyxy = [torch.cat([g[:, :2], g[:, :2]], dim=1) for g in gt_bboxes]          # only the centres are read (:633-636)
    return generate_black_paper_batch(imgs, xyxy, prior_size, dense_cls, imgsize, draws=draws, fill='max', return_obb=True)
