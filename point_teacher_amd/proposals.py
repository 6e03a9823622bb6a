"""Mirror of /root/reference/HBB_TOD/mmdet/models/detectors/syn_images_generator_v2.py
(+ data_augument_bank.py helpers) on the HIP device: MIL proposal bags, negative
proposals, strong augmentation and the burn-in step-1 "white rectangle" generator.

List-of-tensors signatures are kept; sizes that the reference discovers with `.nonzero()`
(host sync) are carried as validity masks instead, see the *_masked variants."""
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
    return _split(props, sizes), [v.reshape(-1, 1) for v in _split(valid.bool(), sizes)]      # (one conversion for the batch)


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


def strong_augmentation_images(img, params=None):
    """The pixel half of strong_augmentation (:41-63, :93-111): it depends only on the input image and the
    (flip, scale) draws, so the detector can produce it BEFORE the MIL stage and push clean and augmented
    images through the student in one batched pass.  Returns (params, list of [C,H,W])."""
    B = img.shape[0]
    params = params if params is not None else draw_strong_aug_params(B)
    return params, [_aug_image(img[i], params[0][i], params[1][i]) for i in range(B)]


def _aug_params(flips, scales, H, W):
    """Per-image row of pt_aug_geometry: (flip_x, flip_y, scale, margin_w, margin_h, scale >= 1); the integer margins are
    computed on the host exactly as :66-71 does."""
    rows = []
    for flip, scale in zip(flips, scales):
        sH, sW = int(H * scale), int(W * scale)
        if scale < 1.0:
            bh, bw = int((H - sH) / 2), int((W - sW) / 2)
        else:
            bh, bw = int((sH - H) / 2), int((sW - W) / 2)
        rows.append([float(flip in ('horizontal', 'diagonal')), float(flip in ('vertical', 'diagonal')), float(scale),
                     float(bw), float(bh), float(scale >= 1.0)])
    return rows


def strong_augmentation_masked(img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes, params=None,
                               imgs=None):
    """Sync-free strong_augmentation: nothing is filtered; instead `gt_valid` / `pseudo_valid`
    masks say which entries the reference would have kept (order is preserved, so assigning
    with the masks equals assigning the filtered lists).  `imgs`: the augmented images when they
    were already produced by strong_augmentation_images with the same `params`.  The geometry of all
    points and boxes of the batch is three launches of pt_aug_geometry."""
    B, C, H, W = img.shape
    flips, scales = params if params is not None else draw_strong_aug_params(B)
    pre = imgs
    imgs = [pre[i] if pre is not None else _aug_image(img[i], flips[i], scales[i]) for i in range(B)]
    dev = img.device
    prm = F.upload_f32(_aug_params(flips, scales, H, W), dev)
    gc = [p.shape[0] for p in gt_points]
    pc = [p.shape[0] for p in pseudo_points]
    goff, _ = F.make_offsets(gc, dev)
    poff, _ = F.make_offsets(pc, dev)
    gp, gv = F.aug_geometry(torch.cat(gt_points), goff, B, prm, H, W, want_valid=True)
    pp, pv = F.aug_geometry(torch.cat(pseudo_points), poff, B, prm, H, W, want_valid=True)
    pb = F.aug_geometry(torch.cat(pseudo_bboxes), poff, B, prm, H, W)
    return (torch.stack(imgs, 0), imgs, _split(gp, gc), gt_labels, _split(pp, pc), pseudo_labels, _split(pb, pc),
            _split(gv.bool(), gc), _split(pv.bool(), pc))


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
DRAW_ROWS = ('scale', 'x', 'y', 'wn', 'rn', 'a', 'boost', 'itv', 'itv2', 'dev')      # row order of pt_black_paper_rects


SEGMENT_MAX_IMAGES, SEGMENT_MAX_CANDIDATES = 16, 8192      # limits of pt_nms_rotated_sorted_segments (PT_ELIMIT beyond)


def plan_black_paper_groups(counts, shapes):
    """Host logic of the fallback: consecutive images are grouped while a group stays within one segmented launch (<= 16 images,
    <= 8192 candidates = 2 G + 10 per image, one image shape / dtype); an image beyond the candidate limit forms a group of one
    (per-image NMS).  -> list of (start, stop)."""
    groups, start = [], 0
    for i in range(len(counts) + 1):
        close = i == len(counts)
        if not close and i > start:
            big = 2 * counts[i] + 10 > SEGMENT_MAX_CANDIDATES or 2 * counts[i - 1] + 10 > SEGMENT_MAX_CANDIDATES
            close = i - start >= SEGMENT_MAX_IMAGES or shapes[i] != shapes[start] or big
        if close and i > start:
            groups.append((start, i))
            start = i
    return groups


def _black_paper_in_groups(imgs, gt_bboxes, prior_size, dense_cls, imgsize, draws, fill, return_obb, counts):
    out = []
    for a, b in plan_black_paper_groups(counts, [(tuple(im.shape), im.dtype) for im in imgs]):
        out += generate_black_paper_batch(imgs[a:b], gt_bboxes[a:b], prior_size, dense_cls, imgsize, None if draws is None else draws[a:b],
                                          fill=fill, return_obb=return_obb)
    return out


def generate_black_paper_batch(imgs, gt_bboxes, prior_size, dense_cls, imgsize, draws=None, fill=255.0, return_obb=False):
    """GPU, sync-free form of generate_black_paper (syn_images_generator_v2.py:591-690) for a BATCH of images
    [C,H,W] whose real objects are gt_bboxes[i] [G_i,4] (xyxy).

    For every real object one rotated rectangle is drawn from the `prior_size` shape priors
    (:610-629), the first two objects whose Bernoulli(0.2) draw fires also spawn a row of 3 or
    5 copies (:640-663), everything is rotated-NMS'd at IoU 0.05 against the real objects
    (score 1, so they always win) and each other (:665-669), rectangles leaving the image
    are dropped (:671-675) and the survivors are rasterised with value 255 (:678-688).

    Returns per image (img_syn, syn_xyxy [M,4], alive bool [M]) with M = 2G + 10 fixed; rows
    are in the order the reference's `bb[keep]` would have (descending score), `alive`
    marks the rows that exist in the reference's output.  The rasteriser contract is
    "pixels inside or on the int32-truncated quadrilateral" (cv2.fillPoly is not
    available: parity unpinned at pixel level).

    The candidate table of the whole batch is ONE launch (pt_black_paper_rects), every image's score order ONE stable
    int64 sort, rows / polygons / hulls / filters one more launch (pt_black_paper_sorted); per image remain the rotated
    NMS and the rasteriser.  `draws` (tests): per image a dict of per-object draws named as DRAW_ROWS + 'cls'.

    fill='max' paints with the image maximum instead of 255 (the OBB generator,
    OBB_TOD/.../syn_images_generator_v2.py:722, whose inputs are mean/std-normalised);
    return_obb=True returns the rectangles as (cx,cy,w,h,a) rows instead of their hulls."""
    B = len(imgs)
    counts = [int(b.shape[0]) for b in gt_bboxes]
    if (B > SEGMENT_MAX_IMAGES or (B > 1 and max(2 * c + 10 for c in counts) > SEGMENT_MAX_CANDIDATES)
            or len({(tuple(im.shape), im.dtype) for im in imgs}) > 1):
        # beyond what one segmented launch takes (pt_nms_rotated_sorted_segments: 16 images x 8192 candidates; the batched rasteriser:
        # one shape): groups of images it does take; an image with more candidates than a segment holds goes through the
        # per-image NMS inside a group of one (round-3 advice: samples_per_gpu > 16 used to raise in the middle of training)
        return _black_paper_in_groups(imgs, gt_bboxes, prior_size, dense_cls, imgsize, draws, fill, return_obb, counts)
    C, H, W = imgs[0].shape
    dev = imgs[0].device
    sumG = sum(counts)
    L = prior_size.shape[0]
    prior = prior_size if prior_size.device == dev else prior_size.to(dev)   # callers on the training path pass a device copy
    if draws is None:
        D = torch.rand(len(DRAW_ROWS), max(sumG, 1), device=dev)
        D[3:5] = torch.randn(2, max(sumG, 1), device=dev)
        cls = torch.randint(0, L, (max(sumG, 1),), device=dev, dtype=torch.int32)
    else:
        D = torch.stack([torch.cat([d[n].to(dev).float() for d in draws]) for n in DRAW_ROWS]) if sumG else torch.zeros(10, 1, device=dev)
        cls = torch.cat([d['cls'].to(dev) for d in draws]).to(torch.int32) if sumG else torch.zeros(1, dtype=torch.int32, device=dev)
    goff, _ = F.make_offsets(counts, dev)
    gt = torch.cat(gt_bboxes) if sumG else torch.zeros((0, 4), device=dev)
    table, key, exist = F.black_paper_rects(gt, goff, B, prior, len(dense_cls), D, cls, imgsize)
    order = torch.sort(key, stable=True)[1]
    sb, nms_in, polys, hull, pre = F.black_paper_sorted(table, order, exist, imgsize)
    keep = torch.ones_like(pre)
    Ms = [2 * c + 10 for c in counts]
    if B == 1 and Ms[0] > SEGMENT_MAX_CANDIDATES:
        F.nms_rotated_mask(nms_in, 0.05, keep)                    # the per-image kernels take up to 32 768 candidates
    else:
        # one rotated NMS and one rasteriser pass for the whole batch (each image's serial scan is its own workgroup)
        F.nms_rotated_mask_segments(nms_in, Ms, 0.05, keep)
    alive_all = (keep & pre).bool()
    img_of = F.upload_i32(np.repeat(np.arange(B, dtype=np.int32), Ms), dev)
    if fill == 'max':          # device-side value: rasterise a 0/1 mask, then select (no host read of img.max())
        mask = F.fill_quads_batch_(torch.zeros((B, 1, H, W), dtype=imgs[0].dtype, device=dev), polys, alive_all, img_of, 1.0)
        syn = [torch.where(mask[i] == 1, imgs[i].max(), imgs[i]) for i in range(B)]
    else:
        syn = F.fill_quads_batch_(torch.stack([im.float() for im in imgs]).contiguous(), polys, alive_all, img_of, float(fill))
    out, t0 = [], 0
    for i in range(B):
        M = Ms[i]
        out.append((syn[i], sb[t0:t0 + M, :5].contiguous() if return_obb else hull[t0:t0 + M], alive_all[t0:t0 + M]))
        t0 += M
    return out
