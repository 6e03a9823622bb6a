"""Mirror of /root/reference/HBB_TOD/mmdet/models/detectors/syn_images_generator_v2.py
(+ data_augument_bank.py helpers) on the HIP device: MIL proposal bags, negative
proposals, strong augmentation and the burn-in step-1 "white rectangle" generator.

List-of-tensors signatures are kept; sizes that the reference discovers with `.nonzero()`
(host sync) are carried as validity masks instead, see the *_masked variants."""
import math
import random

import numpy as np
import torch
import torch.nn.functional as TF

from . import functional as F
from .core import bbox_cxcywh_to_xyxy, bbox_xyxy_to_cxcywh

EPS = 1e-2


def _split(t, counts):
    return list(torch.split(t, counts)) if len(counts) else []


def fine_proposals_from_cfg(pseudo_boxes, fine_proposal_cfg, img_meta):
    """syn_images_generator_v2.py:262-324.  Returns (proposal_list, proposals_valid_list)."""
    assert fine_proposal_cfg['gen_proposal_mode'] == 'fix_gen'
    counts = [b.shape[0] for b in pseudo_boxes]
    h, w, _ = img_meta[0]['img_shape']
    assert all(tuple(m['img_shape'][:2]) == (h, w) for m in img_meta), 'one image size per batch'
    ratios, shake = fine_proposal_cfg['base_ratios'], fine_proposal_cfg['shake_ratio']
    props, valid = F.fine_proposals(torch.cat(pseudo_boxes), ratios, shake, fine_proposal_cfg['min_scale'], (h, w))
    U = len(ratios) ** 2 * (1 + 4 * len(shake or []))
    sizes = [c * U for c in counts]
    return _split(props, sizes), [v.bool().reshape(-1, 1) for v in _split(valid, sizes)]


def MIL_gen_proposals_from_cfg(pseudo_points, pseudo_boxes, fine_proposal_cfg, gt_boxes, img_meta):
    """syn_images_generator_v2.py:134-145"""
    assert fine_proposal_cfg['gen_mode'] == 'refine', "only gen_mode='refine' is used by the shipped configs"
    proposals_list, proposals_valid_list = fine_proposals_from_cfg(pseudo_boxes, fine_proposal_cfg, img_meta)
    ratios, shake = fine_proposal_cfg['base_ratios'], fine_proposal_cfg['shake_ratio']
    num_aug = len(ratios) ** 2 * (1 + 4 * len(shake or []))
    ref, real = [], []
    for i in range(len(pseudo_boxes)):
        ref.append(pseudo_boxes[i].unsqueeze(1).repeat(1, num_aug, 1).reshape(-1, 4))
        real.append(gt_boxes[i].unsqueeze(1).repeat(1, num_aug, 1).reshape(-1, 4))
    return proposals_list, proposals_valid_list, ref, real


def gen_negative_proposals(gt_points, proposal_cfg, aug_generate_proposals, img_meta, uniforms=None):
    """syn_images_generator_v2.py:234-259.  `uniforms` [B,4,n] injects the four torch.rand draws
    (tests); by default they are drawn on the device (the reference draws on the CPU and copies)."""
    n = proposal_cfg['gen_num_neg']
    if n == 0:
        return None, None
    B = len(gt_points)
    dev = aug_generate_proposals[0].device
    h, w, _ = img_meta[0]['img_shape']
    u = uniforms if uniforms is not None else torch.rand(B, 4, n, device=dev)
    off, _ = F.make_offsets([p.shape[0] for p in aug_generate_proposals], dev)
    neg, ok = F.negative_proposals(u, torch.cat(aug_generate_proposals), off, (h, w), 0.3)
    return _split(neg, [n] * B), _split(ok.bool(), [n] * B)


def random_point_in_quadrilateral(quads, position, ux=None, uy=None):
    """data_augument_bank.py:1471-1493"""
    M = quads.shape[0]
    w = quads[:, 2] - quads[:, 0]
    h = quads[:, 3] - quads[:, 1]
    ux = torch.rand(M, device=quads.device) if ux is None else ux
    uy = torch.rand(M, device=quads.device) if uy is None else uy
    x = quads[:, 0] + w * (1 - position) / 2 + ux.to(quads.dtype) * (w * position)
    y = quads[:, 1] + h * (1 - position) / 2 + uy.to(quads.dtype) * (h * position)
    return torch.stack((x, y), dim=1)


def obb2poly_le90(rboxes):
    """data_augument_bank.py:516-541"""
    if rboxes.shape[0] == 0:
        return rboxes.new_zeros((0, 8))
    cx, cy, w, h, a = rboxes[:, 0], rboxes[:, 1], rboxes[:, 2], rboxes[:, 3], rboxes[:, 4]
    xs = torch.stack([-w * .5, w * .5, w * .5, -w * .5], 1)
    ys = torch.stack([-h * .5, -h * .5, h * .5, h * .5], 1)
    s, c = torch.sin(a)[:, None], torch.cos(a)[:, None]
    return torch.stack([c * xs - s * ys + cx[:, None], s * xs + c * ys + cy[:, None]], -1).reshape(-1, 8)


def obb2xyxy(obb):
    """syn_images_generator_v2.py:382-396"""
    w, h, a = obb[:, 2], obb[:, 3], obb[:, 4]
    ca, sa = torch.cos(a).abs(), torch.sin(a).abs()
    dw, dh = ca * w + sa * h, sa * w + ca * h
    return torch.stack((obb[..., 0] - dw / 2, obb[..., 1] - dh / 2, obb[..., 0] + dw / 2, obb[..., 1] + dh / 2), -1)


def load_basic_shape(shape_list):
    """syn_images_generator_v2.py:581-589"""
    prior_size = torch.Tensor(shape_list).float()
    pattern = [[torch.zeros([int(v) for v in s[:2]]).float()] for s in shape_list]
    return pattern, prior_size


# ------------------------------------------------------------ strong augmentation --
def draw_strong_aug_params(B):
    """The host-side draws of strong_augmentation (:41, :64), one (flip, scale) per image."""
    flips = [random.choice(['horizontal', 'vertical', 'diagonal', 'None']) for _ in range(B)]
    scales = [float(np.around(np.random.uniform(0.8, 1.2), 1)) for _ in range(B)]
    return flips, scales


def _aug_image(img, flip, scale):
    C, H, W = img.shape
    if flip == 'horizontal':
        img = torch.flip(img, dims=[2])
    elif flip == 'vertical':
        img = torch.flip(img, dims=[1])
    elif flip == 'diagonal':
        img = torch.flip(img, dims=[1, 2])
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


def _aug_geometry(pts, boxes, flip, scale, H, W):
    """Transform points [n,2] and boxes [m,4] (either may be None); returns transformed
    tensors and the 'still inside the crop' masks (all True when scale < 1)."""
    sH, sW = int(H * scale), int(W * scale)
    if scale < 1.0:
        bh, bw = int((H - sH) / 2), int((W - sW) / 2)
    else:
        bh, bw = int((sH - H) / 2), int((sW - W) / 2)

    def tf(xy):          # xy [n, 2k] interleaved x,y -> (scaled, scaled+shifted); python-scalar ops only
        xy = xy.clone()
        if flip in ('horizontal', 'diagonal'):
            xy[:, 0::2] = W - xy[:, 0::2]
        if flip in ('vertical', 'diagonal'):
            xy[:, 1::2] = H - xy[:, 1::2]
        scaled = xy * scale
        out = scaled.clone()
        if scale >= 1.0:
            out[:, 0::2] -= bw
            out[:, 1::2] -= bh
        else:
            out[:, 0::2] += bw
            out[:, 1::2] += bh
        return scaled, out
    out_p = out_b = mask = None
    if boxes is not None:
        out_b = tf(boxes)[1]
    if pts is not None:
        q, out_p = tf(pts)
        if scale >= 1.0:   # tested on the scaled, un-shifted point (:78-79, :84-85)
            mask = (q[:, 0] >= bw) & (q[:, 0] < W + bw) & (q[:, 1] >= bh) & (q[:, 1] < H + bh)
        else:
            mask = torch.ones(pts.shape[0], dtype=torch.bool, device=pts.device)
    return out_p, out_b, mask


def _refine_boxes(b):
    """:114-120 (re-order corners after flips)"""
    if b.shape[0] == 0:
        return b
    w = (b[:, 0] - b[:, 2]).abs().reshape(-1, 1)
    h = (b[:, 1] - b[:, 3]).abs().reshape(-1, 1)
    x = torch.minimum(b[:, 0], b[:, 2]).reshape(-1, 1)     # (a python-list index would be uploaded with a blocking copy)
    y = torch.minimum(b[:, 1], b[:, 3]).reshape(-1, 1)
    return bbox_cxcywh_to_xyxy(torch.cat([x + w / 2, y + h / 2, w, h], dim=1))


def strong_augmentation_images(img, params=None):
    """The pixel half of strong_augmentation (:41-63, :93-111): it depends only on the input image and the
    (flip, scale) draws, so the detector can produce it BEFORE the MIL stage and push clean and augmented
    images through the student in one batched pass.  Returns (params, list of [C,H,W])."""
    B = img.shape[0]
    params = params if params is not None else draw_strong_aug_params(B)
    return params, [_aug_image(img[i], params[0][i], params[1][i]) for i in range(B)]


def strong_augmentation_masked(img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes, params=None,
                               imgs=None):
    """Sync-free strong_augmentation: nothing is filtered; instead `gt_valid` / `pseudo_valid`
    masks say which entries the reference would have kept (order is preserved, so assigning
    with the masks equals assigning the filtered lists).  `imgs`: the augmented images when they
    were already produced by strong_augmentation_images with the same `params`."""
    B, C, H, W = img.shape
    flips, scales = params if params is not None else draw_strong_aug_params(B)
    pre = imgs
    imgs, gp_l, pp_l, pb_l, gv_l, pv_l = [], [], [], [], [], []
    for i in range(B):
        imgs.append(pre[i] if pre is not None else _aug_image(img[i], flips[i], scales[i]))
        gp, _, gv = _aug_geometry(gt_points[i], None, flips[i], scales[i], H, W)
        pp, pb, pv = _aug_geometry(pseudo_points[i], pseudo_bboxes[i], flips[i], scales[i], H, W)
        gp_l.append(gp); gv_l.append(gv)
        pp_l.append(pp); pb_l.append(_refine_boxes(pb)); pv_l.append(pv)
    return torch.stack(imgs, 0), imgs, gp_l, gt_labels, pp_l, pseudo_labels, pb_l, gv_l, pv_l


def strong_augmentation(img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes, params=None):
    """syn_images_generator_v2.py:24-132 with the reference's return convention (filtered
    lists).  The boolean gathers synchronise the host; the training loop uses the masked form."""
    out = strong_augmentation_masked(img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes, params)
    imgs, iml, gp, gl, pp, pl, pb, gv, pv = out
    gp = [p[m] for p, m in zip(gp, gv)]
    gl = [l[m] for l, m in zip(gl, gv)]
    pp = [p[m] for p, m in zip(pp, pv)]
    pl = [l[m] for l, m in zip(pl, pv)]
    pb = [b[m] for b, m in zip(pb, pv)]
    return imgs, iml, gp, gl, pp, pl, pb


# ------------------------------------------------- burn-in step 1: white rectangles --
def generate_black_paper_masked(img, gt_bboxes, prior_size, dense_cls, imgsize, max_extra=10, draws=None, fill=255.0,
                                return_obb=False):
    """GPU, sync-free form of generate_black_paper (syn_images_generator_v2.py:591-690) for ONE
    image [C,H,W] whose real objects are gt_bboxes [G,4] (xyxy).

    For every real object one rotated rectangle is drawn from the `prior_size` shape priors
    (:610-629), the first two objects whose Bernoulli(0.2) draw fires also spawn a row of 3 or
    5 copies (:640-663), everything is rotated-NMS'd at IoU 0.05 against the real objects
    (score 1, so they always win) and each other (:665-669), rectangles leaving the image
    are dropped (:671-675) and the survivors are rasterised with value 255 (:678-688).

    Returns (img_syn, syn_xyxy [M,4], alive bool [M]) with M = 2G + max_extra fixed; rows
    are in the order the reference's `bb[keep]` would have (descending score), `alive`
    marks the rows that exist in the reference's output.  The rasteriser contract is
    "pixels inside or on the int32-truncated quadrilateral" (cv2.fillPoly is not
    available: parity unpinned at pixel level).

    fill='max' paints with the image maximum instead of 255 (the OBB generator,
    OBB_TOD/.../syn_images_generator_v2.py:722, whose inputs are mean/std-normalised);
    return_obb=True returns the rectangles as (cx,cy,w,h,a) rows instead of their hulls."""
    C, H, W = img.shape
    dev = img.device
    G = gt_bboxes.shape[0]
    L = prior_size.shape[0]
    prior = prior_size if prior_size.device == dev else prior_size.to(dev)   # callers on the training path pass a device copy
    d = draws or {}

    def draw(name, shape, kind='rand'):
        if name in d:
            return d[name].to(dev)
        return torch.rand(shape, device=dev) if kind == 'rand' else torch.randn(shape, device=dev)
    cls_idx = d['cls'].to(dev) if 'cls' in d else torch.randint(0, L, (G,), device=dev)
    scale_vary = draw('scale', (G,)) * 2.0 + 0.5
    cen_lo, cen_hi = 50.0, imgsize - 50.0
    x = draw('x', (G,)) * (cen_hi - cen_lo) + cen_lo
    y = draw('y', (G,)) * (cen_hi - cen_lo) + cen_lo
    pr = prior[cls_idx]
    w = scale_vary * torch.exp((draw('wn', (G,), 'randn') * 0.4).clamp(-1, 1) * pr[:, 2])
    h = w * torch.exp((draw('rn', (G,), 'randn') * 0.4).clamp(-1, 1) * pr[:, 3])
    w = w * pr[:, 0]
    h = h * pr[:, 1]
    a = draw('a', (G,)) * math.pi - math.pi / 2
    # Tensor.clip(lo, hi) = min(max(x, lo), hi): the UPPER bound wins when a rectangle is so large that the bounds cross
    x = torch.minimum(torch.maximum(x, 0.71 * w), imgsize - 1 - 0.71 * w)
    y = torch.minimum(torch.maximum(y, 0.71 * h), imgsize - 1 - 0.71 * h)
    score = (w * h) / imgsize / imgsize + 0.1
    syn = torch.stack([x, y, w, h, a, score], 1)                                # [G,6]

    # adjacency boost: the first two objects (in order) whose np.random.random() < 0.2
    fire = (d['boost'].to(dev) if 'boost' in d else torch.rand(G, device=dev)) < 0.2
    rank = torch.cumsum(fire.int(), 0)
    dense = cls_idx < len(dense_cls)
    itv = torch.where(dense, draw('itv', (G,)) * 4 + 2, draw('itv2', (G,)) * 40 + 10)
    devi = torch.where(dense, draw('dev', (G,)) * 8 - 4, torch.zeros(G, device=dev))
    ofx = (h + itv) * torch.sin(-a) + devi * torch.cos(a)
    ofy = (h + itv) * torch.cos(a) + devi * torch.sin(a)
    ncopy = torch.where(dense, 5, 3)
    kk = torch.arange(1, 6, device=dev, dtype=torch.float32)
    ex = torch.stack([x[:, None] + kk * ofx[:, None], y[:, None] + kk * ofy[:, None], w[:, None].expand(G, 5),
                      h[:, None].expand(G, 5), a[:, None].expand(G, 5), score[:, None] - 0.001 * kk], 2)   # [G,5,6]
    eok = kk[None, :] <= ncopy[:, None]                                          # [G,5]
    extras, extras_ok = [], []
    for r in (1, 2):                       # adjboost = 2: only the first two firing objects spawn a row
        hit = fire & (rank == r)
        idx = torch.argmax(hit.int()) if G else None          # first hit (0 when there is none)
        has = hit.any() if G else None
        if G:                                  # index_select: `ex[idx]` with a 0-dim device index reads it back to the host
            extras.append(ex.index_select(0, idx.reshape(1))[0])
            extras_ok.append(eok.index_select(0, idx.reshape(1))[0] & has)
        else:
            extras.append(torch.zeros(5, 6, device=dev))
            extras_ok.append(torch.zeros(5, dtype=torch.bool, device=dev))
    extra = torch.cat(extras, 0)
    extra_ok = torch.cat(extras_ok, 0)

    # real objects: fixed 0.7*prior[0] squares, angle 0, score 1 (:599-602, bb_occupied[:,5]=1)
    gc = bbox_xyxy_to_cxcywh(gt_bboxes)
    occ_wh = prior[cls_idx][:, 0] * 0.7
    occ = torch.stack([gc[:, 0], gc[:, 1], occ_wh, occ_wh, torch.zeros(G, device=dev), torch.ones(G, device=dev)], 1)
    allb = torch.cat([occ, syn, extra], 0)                                       # [2G+10, 6]
    exist = torch.cat([torch.ones(2 * G, dtype=torch.bool, device=dev), extra_ok])
    sc = torch.where(exist, allb[:, 5], torch.full_like(allb[:, 5], -1.0))
    order = torch.sort(sc, descending=True, stable=True)[1]
    sb = allb[order]
    far = sb.new_zeros(6)                      # a box nowhere near the image for the rows that do not exist
    far[:2] = -1e4
    far[2:4] = 1e-3
    keep = F.nms_rotated_mask(torch.where(exist[order][:, None], sb, far), 0.05).bool()
    xyxy = obb2xyxy(sb)
    inside = (xyxy.min(-1)[0] >= 0) & (xyxy.max(-1)[0] <= imgsize - 1)
    alive = keep & exist[order] & (sb[:, 5] < 1) & inside
    polys = obb2poly_le90(sb[:, :5])
    if fill == 'max':      # device-side value: rasterise a 0/1 mask, then select (no host read of img.max())
        mask = F.fill_quads_(torch.zeros((1, H, W), dtype=img.dtype, device=dev), polys, alive, 1.0)
        img_syn = torch.where(mask == 1, img.max(), img)
    else:
        img_syn = F.fill_quads_(img.clone().contiguous(), polys, alive, float(fill))
    if return_obb:
        return img_syn, sb[:, :5].contiguous(), alive
    hull = torch.stack([polys[:, 0::2].min(1)[0], polys[:, 1::2].min(1)[0], polys[:, 0::2].max(1)[0],
                        polys[:, 1::2].max(1)[0]], 1)
    return img_syn, hull, alive
