"""CPU ORACLE of the data pipeline - TEST INFRASTRUCTURE ONLY.

Only tests import this file.  It restates, in plain numpy, what the reference's train/test pipelines do to one
image and its boxes (configs `train_pipeline` / `test_pipeline`, e.g.
/root/reference/HBB_TOD/configs/point_teacher/aitodv2_point_teacher_0%.py:180-203 and
/root/reference/OBB_TOD/configs/_base_/datasets/sodaa.py:6-29):

    LoadImageFromFile -> LoadAnnotations -> Resize|RResize -> RandomFlip|RRandomFlip -> Normalize -> Pad
    -> DefaultFormatBundle -> Collect      (+ mmcv's collate: zero-pad the batch to its largest image)

Two kinds of statements live here:

* HOST LOGIC of the reference's own files - box scaling / clipping / flipping, scale sampling, meta keys,
  annotation parsing, the group samplers.  PINNED: tests/golden/pipeline_flow.npz holds outputs of the
  reference's own methods (oracle/gen_golden_pipeline.py calls them by path), and
  tests/test_data_pipeline.py checks this file against them.
* PIXEL WORK, which the reference delegates to mmcv 1.x (`mmcv.imrescale`, `imflip`, `imnormalize`,
  `impad_to_multiple`) over OpenCV.  Neither mmcv nor cv2 is vendored in /root/reference or installed here:
  PARITY UNPINNED for these functions.  They restate the published algorithms:
    - mmcv/image/geometric.py `rescale_size` (int(x * scale + 0.5)), `imrescale`, `imresize` -> cv2.resize;
    - OpenCV modules/imgproc/src/resize.cpp, INTER_LINEAR on CV_8U: 11-bit fixed-point taps
      (INTER_RESIZE_COEF_BITS = 11, weights = cvRound(float weight * 2048)), HResizeLinear in int32,
      VResizeLinear<uchar>  (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2; an exact 2x shrink is routed to
      INTER_AREA (rounded 2x2 mean);
    - mmcv/image/photometric.py `imnormalize_`: float32 copy, cv2.cvtColor(BGR2RGB), cv2.subtract with the
      float32 mean (computed in float32), cv2.multiply with 1/float64(std) (scalar multiply is evaluated in
      double and rounded once to float32);
    - mmcv `impad` / `impad_to_multiple`: constant border on the right / bottom.
"""
import math

import numpy as np


# ----------------------------------------------------------------------------- pixel work (unpinned)
def rescale_size(old_size, scale):
    """mmcv/image/geometric.py rescale_size: (w, h), scale = float factor or (long, short) bound."""
    w, h = old_size
    if isinstance(scale, (float, int)):
        if scale <= 0:
            raise ValueError(f'Invalid scale {scale}, must be positive.')
        factor = scale
    else:
        max_long, max_short = max(scale), min(scale)
        factor = min(max_long / max(h, w), max_short / min(h, w))
    return int(w * float(factor) + 0.5), int(h * float(factor) + 0.5)


def _linear_taps(dst_n, src_n, clamp_frac):
    scale = 1.0 / (float(dst_n) / float(src_n))
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_frac:
        lo = s < 0
        f[lo], s[lo] = 0.0, 0
        hi = s >= src_n - 1
        f[hi], s[hi] = 0.0, src_n - 1
    w0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)      # cvRound = round half to even
    w1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, w0, w1


def imresize_linear_u8(img, size):
    """cv2.resize(img, size=(w, h), interpolation=INTER_LINEAR) for uint8 HxWxC."""
    assert img.dtype == np.uint8 and img.ndim == 3
    dst_w, dst_h = size
    src_h, src_w = img.shape[:2]
    if (dst_h, dst_w) == (src_h, src_w):
        return img.copy()
    if src_h == 2 * dst_h and src_w == 2 * dst_w:
        a = img.astype(np.int64)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, a0, a1 = _linear_taps(dst_w, src_w, True)
    sy, b0, b1 = _linear_taps(dst_h, src_h, False)
    sx1 = np.minimum(sx + 1, src_w - 1)
    S = img.astype(np.int64)
    H = S[:, sx, :] * a0[None, :, None] + S[:, sx1, :] * a1[None, :, None]          # [src_h, dst_w, C]
    y0 = np.clip(sy, 0, src_h - 1)
    y1 = np.clip(sy + 1, 0, src_h - 1)
    out = (((b0[:, None, None] * (H[y0] >> 4)) >> 16) + ((b1[:, None, None] * (H[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def imrescale(img, scale):
    """mmcv.imrescale(img, scale, return_scale=True) -> (img, scale_factor)."""
    h, w = img.shape[:2]
    new_size = rescale_size((w, h), scale)
    if isinstance(scale, (float, int)):
        factor = scale
    else:
        factor = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return imresize_linear_u8(img, new_size), factor


def imflip(img, direction='horizontal'):
    if direction == 'horizontal':
        return np.flip(img, axis=1)
    if direction == 'vertical':
        return np.flip(img, axis=0)
    assert direction == 'diagonal'
    return np.flip(img, axis=(0, 1))


def imnormalize(img, mean, std, to_rgb=True):
    img = img.copy().astype(np.float32)
    mean = np.float64(np.asarray(mean, np.float32).reshape(1, -1))
    stdinv = 1 / np.float64(np.asarray(std, np.float32).reshape(1, -1))
    if to_rgb:
        img = img[..., ::-1]
    img = (img - mean.astype(np.float32)).astype(np.float32)
    return (img.astype(np.float64) * stdinv).astype(np.float32)


def impad(img, shape, pad_val=0):
    out = np.full((shape[0], shape[1]) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:img.shape[0], :img.shape[1]] = img
    return out


def impad_to_multiple(img, divisor, pad_val=0):
    pad_h = int(np.ceil(img.shape[0] / divisor)) * divisor
    pad_w = int(np.ceil(img.shape[1] / divisor)) * divisor
    return impad(img, (pad_h, pad_w), pad_val)


def prepare_image(img, scale=None, keep_ratio=True, flip=None, norm=None, size_divisor=None, pad_val=0):
    """The image half of the pipeline on one decoded uint8 BGR image -> (float32 CHW, meta)."""
    meta = dict(ori_shape=img.shape)
    h, w = img.shape[:2]
    if scale is not None:
        if keep_ratio:
            out, _ = imrescale(img, scale)
        else:
            out = imresize_linear_u8(img, scale)
        nh, nw = out.shape[:2]
        meta['scale_factor'] = np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32)   # transforms.py:220-232
        img = out
    meta['img_shape'] = img.shape
    if flip:
        img = imflip(img, flip)
    if norm is not None:
        img = imnormalize(img, norm['mean'], norm['std'], norm['to_rgb'])
    if size_divisor is not None:
        img = impad_to_multiple(img, size_divisor, pad_val)
    meta['pad_shape'] = img.shape
    return np.ascontiguousarray(img.transpose(2, 0, 1)), meta


def collate_images(chw_list):
    """mmcv.parallel.collate for stacked images: zero-pad right/bottom to the largest H, W of the batch."""
    H = max(a.shape[1] for a in chw_list)
    W = max(a.shape[2] for a in chw_list)
    out = np.zeros((len(chw_list), chw_list[0].shape[0], H, W), dtype=chw_list[0].dtype)
    for i, a in enumerate(chw_list):
        out[i, :, :a.shape[1], :a.shape[2]] = a
    return out


# ----------------------------------------------------------------------------- host logic (pinned)
def resize_bboxes(bboxes, scale_factor, img_shape, clip=True):
    """Resize._resize_bboxes, transforms.py:239-247."""
    b = bboxes * scale_factor
    if clip:
        b[:, 0::2] = np.clip(b[:, 0::2], 0, img_shape[1])
        b[:, 1::2] = np.clip(b[:, 1::2], 0, img_shape[0])
    return b


def rresize_bboxes(bboxes, scale_factor):
    """RResize._resize_bboxes, OBB_TOD/mmrotate/datasets/pipelines/transforms.py:37-46 (in place on a copy)."""
    b = bboxes.copy().reshape(-1, 5)
    w_scale, h_scale, _, _ = scale_factor
    b[:, 0] *= w_scale
    b[:, 1] *= h_scale
    b[:, 2:4] *= np.sqrt(w_scale * h_scale)
    return b.reshape(bboxes.shape)


def bbox_flip(bboxes, img_shape, direction):
    """RandomFlip.bbox_flip, transforms.py:374-401."""
    f = bboxes.copy()
    h, w = img_shape[:2]
    if direction in ('horizontal', 'diagonal'):
        f[..., 0::4] = w - bboxes[..., 2::4]
        f[..., 2::4] = w - bboxes[..., 0::4]
    if direction in ('vertical', 'diagonal'):
        f[..., 1::4] = h - bboxes[..., 3::4]
        f[..., 3::4] = h - bboxes[..., 1::4]
    return f


def norm_angle(angle, version):
    """mmrotate/core/bbox/transforms.py norm_angle."""
    if version == 'oc':
        return angle
    if version == 'le90':
        return (angle + np.pi / 2) % np.pi - np.pi / 2
    assert version == 'le135'
    return (angle + np.pi / 4) % np.pi - np.pi / 4


def rbbox_flip(bboxes, img_shape, direction, version='oc'):
    """RRandomFlip.bbox_flip, OBB_TOD/mmrotate/datasets/pipelines/transforms.py:64-95.  The SODA-A config builds
    it WITHOUT `version` (sodaa.py:10), so le90 boxes are flipped with the 'oc' rule - kept."""
    b = bboxes.reshape(-1, 5)
    f = b.copy()
    if direction == 'horizontal':
        f[:, 0] = img_shape[1] - b[:, 0] - 1
    elif direction == 'vertical':
        f[:, 1] = img_shape[0] - b[:, 1] - 1
    else:
        assert direction == 'diagonal'
        f[:, 0] = img_shape[1] - b[:, 0] - 1
        f[:, 1] = img_shape[0] - b[:, 1] - 1
        return f.reshape(bboxes.shape)
    if version == 'oc':
        rot = b[:, 4] != np.pi / 2
        f[rot, 4] = np.pi / 2 - b[rot, 4]
        f[rot, 2] = b[rot, 3]
        f[rot, 3] = b[rot, 2]
    else:
        f[:, 4] = norm_angle(np.pi - b[:, 4], version)
    return f.reshape(bboxes.shape)


def parse_coco_ann(img_info, ann_info, cat_ids):
    """CocoDataset._parse_ann_info, HBB_TOD/mmdet/datasets/coco.py:124-180 (boxes / labels / ignore)."""
    cat2label = {c: i for i, c in enumerate(cat_ids)}
    boxes, labels, ignore = [], [], []
    for ann in ann_info:
        if ann.get('ignore', False):
            continue
        x1, y1, w, h = ann['bbox']
        inter_w = max(0, min(x1 + w, img_info['width']) - max(x1, 0))
        inter_h = max(0, min(y1 + h, img_info['height']) - max(y1, 0))
        if inter_w * inter_h == 0:
            continue
        if ann['area'] <= 0 or w < 1 or h < 1:
            continue
        if ann['category_id'] not in cat_ids:
            continue
        bbox = [x1, y1, x1 + w, y1 + h]
        if ann.get('iscrowd', False):
            ignore.append(bbox)
        else:
            boxes.append(bbox)
            labels.append(cat2label[ann['category_id']])
    return (np.array(boxes, dtype=np.float32).reshape(-1, 4), np.array(labels, dtype=np.int64),
            np.array(ignore, dtype=np.float32).reshape(-1, 4))


def group_sampler_indices(flag, samples_per_gpu, rng):
    """GroupSampler.__iter__, samplers/group_sampler.py:24-47, drawing from `rng` (a numpy RandomState) in the
    same order the reference draws from the global numpy state."""
    flag = np.asarray(flag).astype(np.int64)
    sizes = np.bincount(flag)
    indices = []
    for i, size in enumerate(sizes):
        if size == 0:
            continue
        indice = np.where(flag == i)[0]
        rng.shuffle(indice)
        num_extra = int(np.ceil(size / samples_per_gpu)) * samples_per_gpu - len(indice)
        indice = np.concatenate([indice, rng.choice(indice, num_extra)])
        indices.append(indice)
    indices = np.concatenate(indices)
    indices = [indices[i * samples_per_gpu:(i + 1) * samples_per_gpu]
               for i in rng.permutation(range(len(indices) // samples_per_gpu))]
    return np.concatenate(indices).astype(np.int64).tolist()


def distributed_group_sampler_indices(flag, samples_per_gpu, num_replicas, rank, epoch, seed=0):
    """DistributedGroupSampler.__iter__, samplers/group_sampler.py:106-140."""
    import torch
    flag = np.asarray(flag)
    sizes = np.bincount(flag)
    num_samples = sum(int(math.ceil(s * 1.0 / samples_per_gpu / num_replicas)) * samples_per_gpu for s in sizes)
    g = torch.Generator()
    g.manual_seed(epoch + seed)
    indices = []
    for i, size in enumerate(sizes):
        if size > 0:
            indice = np.where(flag == i)[0]
            indice = indice[list(torch.randperm(int(size), generator=g).numpy())].tolist()
            extra = int(math.ceil(size * 1.0 / samples_per_gpu / num_replicas)) * samples_per_gpu * num_replicas - len(indice)
            tmp = indice.copy()
            for _ in range(extra // size):
                indice.extend(tmp)
            indice.extend(tmp[:extra % size])
            indices.extend(indice)
    indices = [indices[j] for i in list(torch.randperm(len(indices) // samples_per_gpu, generator=g))
               for j in range(i * samples_per_gpu, (i + 1) * samples_per_gpu)]
    offset = num_samples * rank
    return indices[offset:offset + num_samples]


def min_area_rect(pts):
    """cv2.minAreaRect restated for a small convex point set (rotating calipers over the hull edges):
    -> ((cx, cy), (w, h), angle_deg) up to the rectangle's own symmetries (PARITY UNPINNED - cv2 absent)."""
    p = np.asarray(pts, np.float64).reshape(-1, 2)
    hull = _convex_hull(p)
    best = None
    for i in range(len(hull)):
        e = hull[(i + 1) % len(hull)] - hull[i]
        n = np.hypot(*e)
        if n == 0:
            continue
        u = e / n
        v = np.array([-u[1], u[0]])
        pu, pv = hull @ u, hull @ v
        w, h = pu.max() - pu.min(), pv.max() - pv.min()
        if best is None or w * h < best[0]:
            c = u * (pu.max() + pu.min()) / 2 + v * (pv.max() + pv.min()) / 2
            best = (w * h, c, w, h, math.degrees(math.atan2(u[1], u[0])))
    _, c, w, h, a = best
    return (c[0], c[1]), (w, h), a


def _cross(a, b):
    return a[0] * b[1] - a[1] * b[0]


def _convex_hull(p):
    p = np.unique(p, axis=0)
    if len(p) <= 2:
        return p
    p = p[np.lexsort((p[:, 1], p[:, 0]))]

    def half(points):
        out = []
        for q in points:
            while len(out) >= 2 and _cross(out[-1] - out[-2], q - out[-2]) <= 0:
                out.pop()
            out.append(q)
        return out
    lower, upper = half(p), half(p[::-1])
    return np.array(lower[:-1] + upper[:-1])


def poly2obb_np_le90(poly):
    """mmrotate/core/bbox/transforms.py:393-418 on top of min_area_rect."""
    (x, y), (w, h), a = min_area_rect(np.array(poly).reshape((4, 2)))
    if w < 2 or h < 2:
        return None
    a = a / 180 * np.pi
    if w < h:
        w, h = h, w
        a += np.pi / 2
    while not np.pi / 2 > a >= -np.pi / 2:
        if a >= np.pi / 2:
            a -= np.pi
        else:
            a += np.pi
    return x, y, w, h, a
