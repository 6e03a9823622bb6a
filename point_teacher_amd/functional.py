"""Tensor-level wrappers (and autograd Functions) over the C ABI in include/pt_hip.h.

Everything here runs on the HIP device through libpt_hip.so; there is no torch/CPU
fallback.  Batched inputs follow the C ABI convention: per-image ground truths are
concatenated and described by an int32 offsets tensor `off[B+1]` on the device.
"""
import numpy as np
import os

import torch

from . import hip

f32, i32, u8 = torch.float32, torch.int32, torch.uint8


def _f(x):
    return x.contiguous().to(f32)


class _PinnedRing:
    """Persistent pinned staging ring for small host->device index uploads (offsets, batch ids).  A fresh pageable upload
    would serialise the host behind the stream every iteration; copies from this ring are truly asynchronous.
    A staged slot may only be rewritten after the copy that read it has executed, and the host runs ahead of the GPU by whole
    iterations: the ring is cut into SEG segments, an event is recorded on the stream when the writer leaves a segment and
    awaited before the writer enters that segment again (normally long complete: no stall).  Uploads that would take a large
    share of a segment (the batch ids of 60 750 RoIs at the 100 % configuration: an unguarded 256 K-word ring wrapped inside
    ONE iteration there and handed kernels overwritten offsets - a GPU memory fault), uploads from another stream than the
    default one, and host targets go through torch's own pinned allocator / a plain copy instead."""
    SEG = 8

    def __init__(self, words=1 << 18):
        self.words = words
        self.seg_words = words // self.SEG
        self.buf = None
        self.pos = 0                     # invariant: 0 <= pos < words and pos is never a segment END (see `_turn`)
        self.events = [None] * self.SEG

    # the three touch points with the device, separate so that the cursor logic can be driven without one (tests/test_host_logic.py)
    def _alloc(self):
        return torch.empty(self.words, dtype=torch.int32).pin_memory()

    def _record(self, seg, stream):
        ev = self.events[seg] or torch.cuda.Event()
        ev.record(stream)                                  # covers every copy staged in the segment being left
        self.events[seg] = ev

    def _await(self, seg):
        if self.events[seg] is not None:
            self.events[seg].synchronize()                 # the previous lap's copies out of this segment have executed

    def _turn(self, seg, stream):
        """Leave segment `seg` (its copies are all enqueued: record the event that covers them) and enter the next one
        (await the event of the previous lap).  -> the segment entered."""
        self._record(seg, stream)
        seg = (seg + 1) % self.SEG
        self.pos = seg * self.seg_words
        self._await(seg)
        return seg

    def _stage(self, arr, n, dtype, stream):
        """Reserve n words, write `arr` there -> the staged view.  An upload that does not fit the rest of the current segment
        starts the next one; an upload that ends EXACTLY on a segment end also turns the segment (after its copy is enqueued,
        `_finish`): a cursor parked on a segment end - or on the end of the ring, where the slice is empty and the write
        raises - would skip both the event of the segment left and the wait of the segment entered."""
        seg = self.pos // self.seg_words
        if self.pos - seg * self.seg_words + n > self.seg_words:
            seg = self._turn(seg, stream)
        view = self.buf[self.pos:self.pos + n]
        self.pos += n
        if dtype != torch.int32:
            view = view.view(torch.float32)
        view.numpy()[:] = arr.reshape(-1)
        return view, seg

    def _finish(self, seg, stream):
        if self.pos == (seg + 1) * self.seg_words:          # exact fill: the copy of this upload is enqueued, turn now
            self._turn(seg, stream)

    def upload(self, arr, device, dtype):
        np_dtype = np.int32 if dtype == torch.int32 else np.float32
        arr = np.ascontiguousarray(arr, dtype=np_dtype)
        n = arr.size
        device = torch.device(device)
        if device.type != 'cuda':
            return torch.from_numpy(arr.copy())
        stream = torch.cuda.current_stream(device)
        if n > self.seg_words // 4 or stream != torch.cuda.default_stream(device):
            # torch's caching host allocator keeps the staging block alive until the stream has executed the copy
            return torch.from_numpy(arr).pin_memory().to(device, non_blocking=True)
        if self.buf is None:
            self.buf = self._alloc()
        view, seg = self._stage(arr, n, dtype, stream)
        out = view.to(device, non_blocking=True).reshape(arr.shape)
        self._finish(seg, stream)
        return out


_ring = _PinnedRing()


def upload_i32(values, device):
    return _ring.upload(np.asarray(values, dtype=np.int32), device, torch.int32)


def upload_f32(values, device):
    return _ring.upload(np.asarray(values, dtype=np.float32), device, torch.float32)


_OFFSETS = {}


def make_offsets(counts, device):
    """Host list of per-image counts -> (off int32 [B+1] on device, total).  A pure function of the counts: the device tensor is kept
    per (counts, device, stream) - an iteration asks for the same offsets several times (the assigners, the pseudo boxes, the MIL
    stages), and a data set cycles through the same count tuples; per STREAM, because an upload is ordered only on the stream that
    issued it (the teacher pass runs beside the student's)."""
    device = torch.device(device)
    key = (tuple(int(c) for c in counts), device, torch.cuda.current_stream(device).stream_id if device.type == 'cuda' else 0)
    hit = _OFFSETS.get(key)
    if hit is not None:
        return hit
    off = np.zeros(len(counts) + 1, np.int32)
    off[1:] = np.cumsum(counts)
    if len(_OFFSETS) >= 512:
        _OFFSETS.clear()
    hit = _OFFSETS[key] = (upload_i32(off, device), int(off[-1]))
    return hit


# ------------------------------------------------------------------ assigners --

def topk_assign(points, gt_xy, off, B, num_pre, want_cand=False, gt_valid=None):
    """pt_topk_assign.  points [P,2]; gt_xy [sumG,>=2] (first two columns used);
    gt_valid optional uint8/bool [sumG].  Returns gt_inds int32 [B,P] (and cand [sumG,num_pre])."""
    P = points.shape[0]
    sumG = gt_xy.shape[0]
    gt_inds = torch.empty((B, P), dtype=i32, device=points.device)
    cand = torch.empty((sumG, num_pre), dtype=i32, device=points.device) if want_cand else None
    xy = _f(gt_xy[:, :2]) if sumG else None
    gv = gt_valid.to(u8).contiguous() if (gt_valid is not None and sumG) else None
    hip.call('pt_topk_assign', _f(points), P, xy, gv, off, B, sumG, num_pre, gt_inds, cand)
    return (gt_inds, cand) if want_cand else gt_inds


def fuse_assign(points, reg, cls, gt_xy, gt_labels, off, B, num_pre=5, topk=3, cls_w=1.0, reg_w=1.0, loc_w=1.0):
    """pt_fuse_assign.  reg [B*P,4] distances, cls [B*P,C] logits."""
    P = points.shape[0]
    sumG = gt_xy.shape[0]
    C = cls.shape[-1]
    gt_inds = torch.empty((B, P), dtype=i32, device=points.device)
    cand = torch.empty((sumG, num_pre), dtype=i32, device=points.device)
    hip.call('pt_fuse_assign', _f(points), P, _f(reg), _f(cls), C, _f(gt_xy[:, :2]) if sumG else None,
             gt_labels.to(i32).contiguous() if sumG else None, off, B, sumG, num_pre, topk,
             float(cls_w), float(reg_w), float(loc_w), gt_inds, cand)
    return gt_inds, cand


def pseudo_boxes(points, reg, cls, gt_xy, gt_labels, gt_bboxes, off, B, gt_inds, cand):
    """pt_pseudo_boxes -> dict(bboxes, points, scores, nassigned, iou)."""
    P, sumG, C = points.shape[0], gt_xy.shape[0], cls.shape[-1]
    dev = points.device
    out = dict(bboxes=torch.empty((sumG, 4), dtype=f32, device=dev), points=torch.empty((sumG, 2), dtype=f32, device=dev),
               scores=torch.empty((sumG,), dtype=f32, device=dev), nassigned=torch.empty((sumG,), dtype=i32, device=dev),
               iou=torch.empty((sumG,), dtype=f32, device=dev))
    if sumG:
        hip.call('pt_pseudo_boxes', _f(points), P, _f(reg), _f(cls), C, _f(gt_xy[:, :2]), gt_labels.to(i32).contiguous(),
                 _f(gt_bboxes) if gt_bboxes is not None else None, off, B, sumG, cand.shape[1], gt_inds, cand,
                 out['bboxes'], out['points'], out['scores'], out['nassigned'], out['iou'] if gt_bboxes is not None else None)
    return out


def fcos_targets(points, gt_inds, boxes, box_labels, off, B, num_classes, want_ctr=True):
    """pt_fcos_targets -> labels int32 [B*P], bbox_targets [B*P,4], ctr_target [B*P]."""
    P = points.shape[0]
    dev = points.device
    labels = torch.empty((B * P,), dtype=i32, device=dev)
    nb = boxes.shape[0] if boxes is not None else 0
    if boxes is None:                      # labels only (classification branch)
        tg, ctr = None, None
    else:
        tg = torch.zeros((B * P, 4), dtype=f32, device=dev)
        ctr = torch.zeros((B * P,), dtype=f32, device=dev) if want_ctr else None
    hip.call('pt_fcos_targets', _f(points), P, gt_inds.contiguous(), _f(boxes) if nb else None,
             box_labels.to(i32).contiguous() if (box_labels is not None and box_labels.numel()) else None,
             off, B, num_classes, labels, tg if nb else None, ctr if nb else None)
    return labels, tg, ctr


def fcos_dense_targets(points, regress_ranges, sample_radius, target_norm, boxes, box_labels, off, B, num_classes,
                       center_sampling=True):
    """pt_fcos_dense_targets (FCOSHead.get_targets of the supervised baseline) -> labels int32 [B*P],
    bbox_targets [B*P,4] (already / target_norm), ctr_target [B*P]."""
    P = points.shape[0]
    dev = points.device
    labels = torch.empty((B * P,), dtype=i32, device=dev)
    tg = torch.zeros((B * P, 4), dtype=f32, device=dev)
    ctr = torch.zeros((B * P,), dtype=f32, device=dev)
    nb = boxes.shape[0]
    hip.call('pt_fcos_dense_targets', _f(points), _f(regress_ranges), _f(sample_radius), _f(target_norm), P,
             _f(boxes) if nb else None, box_labels.to(i32).contiguous() if nb else None, off, B, num_classes,
             int(bool(center_sampling)), labels, tg, ctr)
    return labels, tg, ctr


def max_iou_assign(anchors, gt_boxes, off, B, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, match_low_quality=True,
                   gt_max_assign_all=True):
    """pt_max_iou_assign -> (assigned_gt_inds int32 [B, A] (-1 ignore, 0 background, i+1 = box i of the image), max_overlaps [B, A])."""
    A = anchors.shape[0]
    dev = anchors.device
    if A == 0:                                             # a network that predicts no box (max_iou_assigner.py:137-151)
        return torch.empty((B, 0), dtype=i32, device=dev), torch.empty((B, 0), dtype=f32, device=dev)
    lo, hi = (0.0, float(neg_iou_thr)) if isinstance(neg_iou_thr, float) else (float(neg_iou_thr[0]), float(neg_iou_thr[1]))
    mo = torch.empty((B, A), dtype=f32, device=dev)
    am = torch.empty((B, A), dtype=i32, device=dev)
    out = torch.empty((B, A), dtype=i32, device=dev)
    nb = gt_boxes.shape[0]
    best = torch.zeros((max(nb, 1),), dtype=torch.int64, device=dev)
    hip.call('pt_max_iou_assign', _f(anchors), A, _f(gt_boxes) if nb else None, off, B, float(pos_iou_thr), lo, hi,
             float(min_pos_iou), int(bool(match_low_quality)), int(bool(gt_max_assign_all)), mo, am, best, out)
    return out, mo


def fuse_assign_obb(points, dec, cls, gt_xy, gt_labels, off, B, num_pre=5, topk=3, cls_w=1.0, reg_w=1.0, loc_w=1.0):
    """pt_fuse_assign_obb.  dec [B*P,5] decoded (cx,cy,w,h,a), cls [B*P,C] logits."""
    P = points.shape[0]
    sumG = gt_xy.shape[0]
    C = cls.shape[-1]
    gt_inds = torch.empty((B, P), dtype=i32, device=points.device)
    cand = torch.empty((sumG, num_pre), dtype=i32, device=points.device)
    hip.call('pt_fuse_assign_obb', _f(points), P, _f(dec), _f(cls), C, _f(gt_xy[:, :2]) if sumG else None,
             gt_labels.to(i32).contiguous() if sumG else None, off, B, sumG, num_pre, topk,
             float(cls_w), float(reg_w), float(loc_w), gt_inds, cand)
    return gt_inds, cand


def pseudo_boxes_obb(P, dec, cls, gt_xy, gt_labels, off, B, gt_inds, cand):
    """pt_pseudo_boxes_obb -> dict(bboxes [sumG,5], points, scores, nassigned)."""
    sumG, C = gt_xy.shape[0], cls.shape[-1]
    dev = dec.device
    out = dict(bboxes=torch.empty((sumG, 5), dtype=f32, device=dev), points=torch.empty((sumG, 2), dtype=f32, device=dev),
               scores=torch.empty((sumG,), dtype=f32, device=dev), nassigned=torch.empty((sumG,), dtype=i32, device=dev))
    if sumG:
        hip.call('pt_pseudo_boxes_obb', _f(dec), P, _f(cls), C, _f(gt_xy[:, :2]), gt_labels.to(i32).contiguous(), off, B,
                 sumG, cand.shape[1], gt_inds, cand, out['bboxes'], out['points'], out['scores'], out['nassigned'])
    return out


def fcos_targets_obb(points, gt_inds, boxes5, box_labels, off, B, num_classes):
    """pt_fcos_targets_obb -> labels int32 [B*P], bbox_targets [B*P,4], angle_targets [B*P,1], ctr_target [B*P]."""
    P = points.shape[0]
    dev = points.device
    labels = torch.empty((B * P,), dtype=i32, device=dev)
    tg = torch.empty((B * P, 4), dtype=f32, device=dev)
    ang = torch.empty((B * P, 1), dtype=f32, device=dev)
    ctr = torch.empty((B * P,), dtype=f32, device=dev)
    if boxes5.shape[0] == 0:               # no box anywhere: every image takes the G == 0 branch
        boxes5 = boxes5.new_zeros((1, 5))
    hip.call('pt_fcos_targets_obb', _f(points), P, gt_inds.contiguous(), _f(boxes5[:, :5]),
             box_labels.to(i32).contiguous() if (box_labels is not None and box_labels.numel()) else None,
             off, B, num_classes, labels, tg, ang, ctr)
    return labels, tg, ang, ctr


# ---------------------------------------------------------------------- losses --

class _FocalSum(torch.autograd.Function):
    """sum_n,c weight[n] * focal(logit[n,c]) (un-normalised)."""

    @staticmethod
    def forward(ctx, logits, labels, weight, gamma, alpha):
        N, C = logits.shape
        logits = _f(logits)
        nb = hip.call('pt_focal_nblocks', N, C)
        partial = torch.empty((nb,), dtype=f32, device=logits.device)
        hip.call('pt_sigmoid_focal_loss_fwd', logits, labels, weight, N, C, float(gamma), float(alpha), None, partial)
        ctx.save_for_backward(logits, labels, weight)
        ctx.ga = (float(gamma), float(alpha))
        return partial.sum()

    @staticmethod
    def backward(ctx, g):
        logits, labels, weight = ctx.saved_tensors
        N, C = logits.shape
        grad = torch.empty_like(logits)
        hip.call('pt_sigmoid_focal_loss_bwd', logits, labels, weight, g.reshape(1).contiguous().to(f32), N, C,
                 ctx.ga[0], ctx.ga[1], grad)
        return grad, None, None, None, None


def sigmoid_focal_loss_sum(logits, labels, weight=None, gamma=2.0, alpha=0.25):
    labels = labels.to(i32).contiguous()
    weight = _f(weight) if weight is not None else None
    return _FocalSum.apply(logits, labels, weight, gamma, alpha)


def sigmoid_focal_loss_elem(logits, labels, gamma=2.0, alpha=0.25):
    """Element-wise loss [N,C] (no grad), as mmcv.ops.sigmoid_focal_loss(..., 'none')."""
    N, C = logits.shape
    out = torch.empty((N, C), dtype=f32, device=logits.device)
    nb = hip.call('pt_focal_nblocks', N, C)
    partial = torch.empty((nb,), dtype=f32, device=logits.device)
    hip.call('pt_sigmoid_focal_loss_fwd', _f(logits.detach()), labels.to(i32).contiguous(), None, N, C, float(gamma),
             float(alpha), out, partial)
    return out


class _DiouPair(torch.autograd.Function):
    """(diou[N], dnmin[N]) of pred/target [N,4]."""

    @staticmethod
    def forward(ctx, pred, target, eps, hyper, want_dn):
        pred, target = _f(pred), _f(target)
        N = pred.shape[0]
        diou = torch.empty((N,), dtype=f32, device=pred.device)
        dn = torch.empty((N,), dtype=f32, device=pred.device) if want_dn else None
        hip.call('pt_diou_fwd', pred, target, N, float(eps), float(hyper), diou, dn)
        ctx.save_for_backward(pred, target)
        ctx.cfg = (float(eps), float(hyper), want_dn)
        if want_dn:
            return diou, dn
        dummy = diou.new_zeros(())
        ctx.mark_non_differentiable(dummy)
        return diou, dummy

    @staticmethod
    def backward(ctx, g1, g2):
        pred, target = ctx.saved_tensors
        eps, hyper, want_dn = ctx.cfg
        N = pred.shape[0]
        grad = torch.empty_like(pred)
        hip.call('pt_diou_bwd', pred, target, _f(g1) if g1 is not None else None,
                 _f(g2) if (want_dn and g2 is not None) else None, N, eps, hyper, grad)
        return grad, None, None, None, None


def diou_loss_elem(pred, target, eps=1e-6):
    return _DiouPair.apply(pred, target, eps, 0.0, False)[0]


def dn_diou_pair(pred, target, hyper, eps=1e-6):
    return _DiouPair.apply(pred, target, eps, hyper, True)


_MODE = {'iou': 0, 'iof': 1, 'giou': 2}


def bbox_overlaps(b1, b2, mode='iou', is_aligned=False, eps=1e-6):
    """Same contract as mmdet's bbox_overlaps (iou2d_calculator.py:74)."""
    M, N = b1.shape[0], b2.shape[0]
    if is_aligned:
        assert M == N
        out = torch.empty((M,), dtype=f32, device=b1.device)
        hip.call('pt_bbox_overlaps_aligned', _f(b1), _f(b2), M, _MODE[mode], float(eps), out)
    else:
        out = torch.empty((M, N), dtype=f32, device=b1.device)
        hip.call('pt_bbox_overlaps_pairwise', _f(b1), _f(b2), M, N, _MODE[mode], float(eps), out)
    return out


class _Delta2BBox(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, deltas, max_h, max_w, clip):
        rois, deltas = _f(rois), _f(deltas)
        N = rois.shape[0]
        out = torch.empty((N, 4), dtype=f32, device=rois.device)
        hip.call('pt_delta2bbox_fwd', rois, deltas, N, float(max_h), float(max_w), float(clip), out)
        ctx.save_for_backward(rois, deltas)
        ctx.cfg = (float(max_h), float(max_w), float(clip))
        return out

    @staticmethod
    def backward(ctx, g):
        rois, deltas = ctx.saved_tensors
        N = rois.shape[0]
        gd = torch.empty_like(deltas)
        hip.call('pt_delta2bbox_bwd', rois, deltas, _f(g), N, *ctx.cfg, gd)
        return None, gd, None, None, None


def delta2bbox(rois, deltas, max_shape=None, wh_ratio_clip=16 / 1000):
    mh, mw = (float(max_shape[0]), float(max_shape[1])) if max_shape is not None else (0.0, 0.0)
    return _Delta2BBox.apply(rois, deltas, mh, mw, wh_ratio_clip)


# -------------------------------------------------------------------- RoIAlign --

class _RoIAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, out_size, scale, sampling_ratio, aligned, group):
        B, C, H, W = feat.shape
        in_dtype = feat.dtype
        feat = feat.float()                       # bf16/fp16 maps (autocast) are widened: kernels are fp32
        cl = feat.is_contiguous(memory_format=torch.channels_last) and not feat.is_contiguous()
        if cl:
            fbuf = feat.permute(0, 2, 3, 1)            # a contiguous [B,H,W,C] view
        else:
            fbuf = feat.contiguous()
        rois = _f(rois)
        K = rois.shape[0]
        out = torch.empty((K, C, out_size, out_size), dtype=f32, device=feat.device)
        hip.call('pt_roi_align_fwd', fbuf, rois, B, C, H, W, K, out_size, float(scale), sampling_ratio,
                 int(aligned), int(cl), int(group), out)
        ctx.save_for_backward(rois)
        ctx.cfg = (B, C, H, W, out_size, float(scale), sampling_ratio, int(aligned), cl, int(group), in_dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        rois, = ctx.saved_tensors
        B, C, H, W, out_size, scale, sr, aligned, cl, group, in_dtype = ctx.cfg
        K = rois.shape[0]
        if cl:
            gbuf = torch.zeros((B, H, W, C), dtype=f32, device=g.device)
        else:
            gbuf = torch.zeros((B, C, H, W), dtype=f32, device=g.device)
        hip.call('pt_roi_align_bwd', _f(g), rois, B, C, H, W, K, out_size, scale, sr, aligned, int(cl), group, gbuf)
        gfeat = gbuf.permute(0, 3, 1, 2) if cl else gbuf
        return gfeat.to(in_dtype), None, None, None, None, None, None


class _RoIAlignPlanes(torch.autograd.Function):
    """The out-7 channels_last RoIAlign whose result leaves as the split planes of the [K, C * 49] operand of the FC stack
    (pt_roi_align_fwd_planes) - no fp32 block, no split pass.  Its gradient arrives in a CARRIER: a tensor of the planes' shape
    whose first K * C * 49 * 4 bytes hold the fp32 gradient block [K, C, 7, 7] (planes._PlaneConv with `x_gcarrier`); autograd only
    checks shape and dtype."""

    @staticmethod
    def forward(ctx, feat, rois, scale, sampling_ratio, aligned, group, f16=False, tail=0):
        """`tail`: the last `tail` RoIs are not members of `group`-sized bags (the negative proposals appended to the bag rows):
        they run as a launch of their own with runs of one - inside a bag launch 25 scattered negatives would be taken for one run
        whose union spans the image, the slowest path of the backward (364 us for 5 000 + 400 rows against 104 + 45)."""
        B, C, H, W = feat.shape
        assert feat.dtype == f32 and feat.is_contiguous(memory_format=torch.channels_last)
        fbuf = feat.permute(0, 2, 3, 1)
        rois = _f(rois)
        K = rois.shape[0]
        tail = int(tail) if 0 < int(tail) < K and int(group) > 1 else 0
        n = (K + 1) * C * 49
        parts = [(0, K - tail, int(group))] + ([(K - tail, tail, 1)] if tail else [])
        if f16:     # H2 planes (value = h0 + h1, tail 1): the operand of the first FC layer's three-product kernels (F16_FC)
            t = torch.empty((2, n + 8), dtype=torch.float16, device=feat.device)
            for k0, kn, g in parts:       # (each launch ends its rows with the zero row and the scale word: the last one's stay)
                hip.call('pt_roi_align_fwd_planes_f16', fbuf, rois[k0:k0 + kn], B, C, H, W, kn, float(scale), sampling_ratio, int(aligned), g,
                         t.data_ptr() + k0 * C * 49 * 2, n + 8)
        else:
            t = torch.empty((3, n), dtype=torch.bfloat16, device=feat.device)
            for k0, kn, g in parts:
                hip.call('pt_roi_align_fwd_planes', fbuf, rois[k0:k0 + kn], B, C, H, W, kn, float(scale), sampling_ratio, int(aligned), g,
                         t.data_ptr() + k0 * C * 49 * 2, n)
        ctx.save_for_backward(rois)
        ctx.cfg = (B, C, H, W, float(scale), sampling_ratio, int(aligned), parts)
        return t

    @staticmethod
    def backward(ctx, carrier):
        rois, = ctx.saved_tensors
        B, C, H, W, scale, sr, aligned, parts = ctx.cfg
        K = rois.shape[0]
        g = carrier.contiguous().view(-1).view(f32)[:K * C * 49]
        gbuf = torch.zeros((B, H, W, C), dtype=f32, device=g.device)
        for k0, kn, grp in parts:
            hip.call('pt_roi_align_bwd', g[k0 * C * 49:(k0 + kn) * C * 49], rois[k0:k0 + kn], B, C, H, W, kn, 7, scale, sr, aligned, 1, grp, gbuf)
        return gbuf.permute(0, 3, 1, 2), None, None, None, None, None, None, None


def roi_align_planes_ok(feat, rois, output_size):
    return (output_size == 7 and feat.is_cuda and feat.dtype == f32 and feat.dim() == 4 and (feat.shape[1] * 49) % 128 == 0
            and feat.is_contiguous(memory_format=torch.channels_last) and not torch.is_autocast_enabled())


# fp16 x 2 operands ("H2" planes, planes.py): three MFMA products per fp32 product instead of six, 4 bytes per element instead of 6 -
# since round 5 THE activation / gradient format of the trainable trunk, necks, towers and FC stacks.  22
# significant bits per stored value, fp32 accumulation; fp16's 5 exponent bits are watched by planes.CENSUS, which demotes a group to
# bf16 x 3 planes (six products, fp32's range) when it sees saturation.  PT_F16_FC=0: bf16 x 3 everywhere - the strict 24-bit path
# (`bench.py` times it as `strict_fp32`); the other switches turn single groups off.
F16_FC = os.environ.get('PT_F16_FC', '1') != '0'                         # master switch + the MIL head's FC stacks
F16_TOWERS = F16_FC and os.environ.get('PT_F16_TOWERS', '1') != '0'      # the dense head's towers
F16_NECK = F16_FC and os.environ.get('PT_F16_NECK3', '1') != '0'         # FPN's laterals / output convolutions, PSAGG
F16_NECK3 = F16_NECK
F16_TRUNK = F16_FC and os.environ.get('PT_F16_TRUNK', '1') != '0'        # the Bottlenecks: H2 planes between ALL their convolutions
F16_BLOCK3 = F16_FC and os.environ.get('PT_F16_BLOCK3', '1') != '0'      # (PT_F16_TRUNK=0 only) the Bottlenecks' 3x3 alone on H2 operands
F16_CHAIN = os.environ.get('PT_F16_CHAIN', '1') != '0'                   # one gradient scale per backward chain (0: every link re-scales)
F16_WEIGHT_SCALE = hip.header_constant('PT_F16_WEIGHT_SCALE')        # weight planes with np = 2 hold 16 w (include/pt_hip.h)


def roi_align_planes(feat, rois, spatial_scale=1.0, sampling_ratio=0, aligned=True, group=1, f16=False, tail=0):
    """-> bf16 [3, (K + 1) * C * 49]: row-major split planes of roi_align(...).flatten(1) (+ a zero row); f16: fp16 [2, ...].
    `tail`: trailing RoIs outside the `group`-sized bags (see _RoIAlignPlanes)."""
    return _RoIAlignPlanes.apply(feat, rois, spatial_scale, int(sampling_ratio), bool(aligned), int(group), bool(f16), int(tail))


def roi_align(feat, rois, output_size=7, spatial_scale=1.0, sampling_ratio=0, aligned=True, group=1):
    """mmcv.ops.roi_align(input, rois, output_size, spatial_scale, sampling_ratio, 'avg', aligned).
    `group`: how many consecutive RoIs belong together (the U2 boxes of one MIL bag overlap); a locality hint, any value
    gives the same result."""
    return _RoIAlign.apply(feat, rois, int(output_size), spatial_scale, int(sampling_ratio), bool(aligned), int(group))


# ------------------------------------------------------------------ MIL bags --

def fine_proposals(boxes, base_ratios, shake_ratio, min_scale, img_hw):
    """boxes [sumG,4] -> (props [sumG*U,4], valid uint8 [sumG*U])."""
    sumG = boxes.shape[0]
    shake = list(shake_ratio) if shake_ratio is not None else []
    U = len(base_ratios) ** 2 * (1 + 4 * len(shake))
    props = torch.empty((sumG * U, 4), dtype=f32, device=boxes.device)
    valid = torch.empty((sumG * U,), dtype=u8, device=boxes.device)
    hip.call('pt_fine_proposals', _f(boxes), sumG, hip.host_floats(base_ratios), len(base_ratios),
             hip.host_floats(shake), len(shake), float(min_scale), float(img_hw[0]), float(img_hw[1]), props, valid)
    return props, valid


def negative_proposals(u, pos, pos_off, img_hw, iou_thr=0.3):
    """u [B,4,n] uniforms -> (neg [B*n,4], ok uint8 [B*n])."""
    B, _, n = u.shape
    neg = torch.empty((B * n, 4), dtype=f32, device=u.device)
    ok = torch.empty((B * n,), dtype=u8, device=u.device)
    hip.call('pt_negative_proposals', _f(u), B, n, _f(pos), pos_off, float(img_hw[0]), float(img_hw[1]),
             float(iou_thr), neg, ok)
    return neg, ok


class _MilBagLoss(torch.autograd.Function):
    """sum over bags of the weighted gfocal bag loss; also returns #valid bags."""

    @staticmethod
    def forward(ctx, cls, ins, valid, labels):
        NB, U2, C = cls.shape
        cls, ins = _f(cls), _f(ins)
        bl = torch.empty((NB,), dtype=f32, device=cls.device)
        bv = torch.empty((NB,), dtype=u8, device=cls.device)
        hip.call('pt_mil_bag_loss_fwd', cls, ins, valid, labels, NB, U2, C, bl, bv)
        ctx.save_for_backward(cls, ins, valid, labels)
        nvalid = bv.sum().to(f32)
        ctx.mark_non_differentiable(nvalid)
        return bl.sum(), nvalid

    @staticmethod
    def backward(ctx, g, _):
        cls, ins, valid, labels = ctx.saved_tensors
        NB, U2, C = cls.shape
        gc, gi = torch.empty_like(cls), torch.empty_like(ins)
        hip.call('pt_mil_bag_loss_bwd', cls, ins, valid, labels, g.reshape(1).contiguous().to(f32), NB, U2, C, gc, gi)
        return gc, gi, None, None


def mil_bag_loss_sum(cls, ins, valid, labels):
    """cls/ins [NB,U2,C]; valid uint8 [NB*U2]; labels [NB] -> (sum loss, n valid bags)."""
    return _MilBagLoss.apply(cls, ins, valid.to(u8).contiguous(), labels.to(i32).contiguous())


class _MilNegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        M, C = x.shape
        x = _f(x)
        out = torch.empty((M,), dtype=f32, device=x.device)
        hip.call('pt_mil_neg_loss_fwd', x, w, M, C, out)
        ctx.save_for_backward(x, w)
        return out.sum()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        M, C = x.shape
        gx = torch.empty_like(x)
        hip.call('pt_mil_neg_loss_bwd', x, w, g.reshape(1).contiguous().to(f32), M, C, gx)
        return gx, None


def mil_neg_loss_sum(neg_cls, neg_w):
    return _MilNegLoss.apply(neg_cls, neg_w.to(u8).contiguous())


def mil_bag_select(cls, ins, valid, labels, bags, pseudo, U1, U2, topk, beta, img_hw):
    """cls/ins [NG,U1,U2,C] (any view of that size) -> merged boxes [NG,D]; D = 4 (xyxy) or 5 (cx,cy,w,h,a)."""
    NG, D = pseudo.shape
    C = cls.shape[-1]
    assert D in (4, 5) and bags.shape[-1] == D
    merged = torch.empty((NG, D), dtype=f32, device=cls.device)
    hip.call('pt_mil_bag_select' if D == 4 else 'pt_mil_bag_select_obb', _f(cls.detach()), _f(ins.detach()),
             valid.to(u8).contiguous(),
             labels.to(i32).contiguous(), _f(bags), _f(pseudo), NG, U1, U2, C, int(topk), float(beta),
             float(img_hw[0]), float(img_hw[1]), merged)
    return merged


# ------------------------------------------------------- EMA / optimizer step --

# Bumped by every kernel that rewrites parameters through raw pointers (autograd version counters do not
# see those writes); caches derived from parameter values key on it (nn_modules._bn_affine).
PARAM_EPOCH = [0]


def ema_update_(teacher_flat, student_flat, alpha):
    PARAM_EPOCH[0] += 1
    hip.call('pt_ema_update', teacher_flat, student_flat, teacher_flat.numel(), float(alpha), float(1 - alpha))


def grad_sqnorm(flat_grad):
    nb = hip.call('pt_sqnorm_nblocks', flat_grad.numel())
    partial = torch.empty((nb,), dtype=f32, device=flat_grad.device)
    hip.call('pt_sqnorm_partial', flat_grad, flat_grad.numel(), partial)
    return partial.sum().reshape(1)


def sgd_step_(param, grad, mom, split, lr_t, momentum, weight_decay, bias_lr_mult, bias_decay_mult, sqnorm,
              max_norm, first_step):
    PARAM_EPOCH[0] += 1
    hip.call('pt_sgd_step', param, grad, mom, param.numel(), int(split), lr_t, float(momentum), float(weight_decay),
             float(bias_lr_mult), float(bias_decay_mult), sqnorm, float(max_norm if max_norm else 0.0),
             int(bool(first_step)))


def sgd_step_groups_(param, grad, mom, tables, lr_t, momentum, weight_decay, sqnorm, max_norm, first_step):
    """pt_sgd_step_groups: `tables` = FlatParams.group_tables() ([host] ends, lr multipliers, decay multipliers, count)."""
    PARAM_EPOCH[0] += 1
    ends, lr_mult, wd_mult, n_groups = tables
    hip.call('pt_sgd_step_groups', param, grad, mom, param.numel(), ends, lr_mult, wd_mult, int(n_groups), lr_t,
             float(momentum), float(weight_decay), sqnorm, float(max_norm if max_norm else 0.0), int(bool(first_step)))


# ------------------------------------------------------------- fp32 Linear on the bf16 matrix cores (bf16x6 split) --

class SplitPlanes:
    """Three bf16 planes of a GEMM operand [rows, k] in the blocked layout of pt_split_bf16x3."""
    __slots__ = ('planes', 'rows', 'k')

    def __init__(self, planes, rows, k):
        self.planes, self.rows, self.k = planes, rows, k

    def dense(self):
        """-> fp32 [3, rows, k]: the three terms, un-blocked (tests)."""
        RB, KB = (self.rows + 15) // 16, (self.k + 31) // 32
        t = self.planes.view(3, RB, KB, 16, 4, 8).float()
        r16 = torch.arange(16, device=t.device)
        q = torch.arange(4, device=t.device)
        g = (r16 >> 2) & 3
        f = (((g ^ (g >> 1)) & 1) << 1) | (g >> 1)                           # slot_swz of csrc/gemm_split.hip: f = (0, 2, 3, 1)
        phys = q[None, :] ^ f[:, None]                                       # physical slot of logical slot q in row r16
        t = torch.gather(t, 4, phys[None, None, None, :, :, None].expand(3, RB, KB, 16, 4, 8))
        return t.permute(0, 1, 3, 2, 4, 5).reshape(3, RB * 16, KB * 32)


def split_bf16x3(x, transpose=False):
    """pt_split_bf16x3: fp32 [R, C] -> SplitPlanes of the operand [rows = R, k = C] (or [rows = C, k = R] when transposed)."""
    assert x.dim() == 2 and x.dtype == f32 and x.stride(1) == 1
    R, C = x.shape
    rows, k = (C, R) if transpose else (R, C)
    n = ((rows + 15) // 16) * ((k + 31) // 32) * 512
    planes = torch.empty((3, n), dtype=torch.bfloat16, device=x.device)
    hip.call('pt_split_bf16x3', x if x.is_contiguous() else x.data_ptr(), x.stride(0), R, C, int(transpose), planes, n)   # (a row-strided view travels as pointer + ld)
    return SplitPlanes(planes, rows, k)


def gemm_bf16x6_nt(a, b, bias=None, relu=False, out=None, tile_rows=0):
    """pt_gemm_bf16x6_nt: [M, N] = A[M, K] B[N, K]^T (+ bias) (ReLU) from SplitPlanes a (rows M) and b (rows N)."""
    assert a.k == b.k, (a.k, b.k)
    M, N = a.rows, b.rows
    c = out if out is not None else torch.empty((M, N), dtype=f32, device=a.planes.device)
    hip.call('pt_gemm_bf16x6_nt', a.planes, a.planes.shape[1], b.planes, b.planes.shape[1], c, c.stride(0), bias, M, N, a.k,
             int(bool(relu)), int(tile_rows))
    return c


_SPLIT_WGRAD = os.environ.get('PT_SPLIT_WGRAD', '1') != '0'      # 0: the 3x3 convolutions' weight gradient stays with the library
_SPLIT_W_CACHE = {}      # id(weight) -> [PARAM_EPOCH, data_ptr, planes of W, planes of W^T or None, weight version]


def _split_weight(w, transposed):
    """The split planes of a weight change only when the parameters do (optimizer step / EMA / re-layout bump PARAM_EPOCH):
    the stack is evaluated 2-3 times per iteration (boxes, negatives, both MIL branches' backward) on the same weights."""
    key = id(w)
    ent = _SPLIT_W_CACHE.get(key)
    if ent is None or ent[0] != PARAM_EPOCH[0] or ent[1] != w.data_ptr() or ent[4] != w._version:
        ent = [PARAM_EPOCH[0], w.data_ptr(), None, None, w._version]      # (_version: load_state_dict / in-place torch edits)
        _SPLIT_W_CACHE[key] = ent
    i = 3 if transposed else 2
    if ent[i] is None:
        ent[i] = split_bf16x3(w.detach(), transpose=transposed)
    return ent[i]


class _SplitLinear(torch.autograd.Function):
    """y = [relu](x W^T + b) with fp32 operands and results, evaluated as six bf16 MFMA products per term with fp32
    accumulation (csrc/gemm_split.hip).  Backward: dx = dy W, dW = dy^T x through the same NT kernel (the split kernel
    transposes), db = column sums."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        xp = split_bf16x3(x)
        y = gemm_bf16x6_nt(xp, _split_weight(w, False), b, relu)
        ctx.relu = relu
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        gy = gy.contiguous()
        if ctx.relu:
            gy = gy * (y > 0)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm_bf16x6_nt(split_bf16x3(gy), _split_weight(w, True))                       # [M, K] = gy[M, N] (W^T)[K, N]^T
        if ctx.needs_input_grad[1]:
            gw = gemm_bf16x6_nt(split_bf16x3(gy, transpose=True), split_bf16x3(x, transpose=True))   # [N, K] = gy^T[N, M] (x^T)[K, M]^T
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(0)
        return gx, gw, gb, None


def split_linear(x, weight, bias=None, relu=False):
    return _SplitLinear.apply(x.contiguous(), weight, bias, bool(relu))


# --------------------------------------------------- fp32 3x3 convolution on the bf16 matrix cores (implicit GEMM, bf16x6) --

class _ConvWeightPlanes:
    """All convolution weights' split planes (1x1 and 3x3), refreshed by ONE launch per parameter epoch (pt_conv_weight_planes_batch).
    A (weight, form, folded scale) triple registers itself at its first use; from then on the first request after the parameters
    changed (optimizer step / EMA bump PARAM_EPOCH; load_state_dict and in-place torch edits move the weight's version counter)
    re-splits every registered triple at once - the student's and the teacher's weights, forward and input-gradient forms - instead
    of one split (and, for the second form, a flip and a copy) per weight and form."""

    def __init__(self):
        self.ent = {}            # (id(w), mode, scale ptr) -> [weakref(w), data_ptr, SplitPlanes, scale tensor or None, version]
        self.table = None        # (device uint8 tensor of pt_conv_weight_item, n_items, total_blocks)
        self.epoch = -1
        self._retired = []       # device tables replaced since the last full refresh: a launch on ANOTHER stream may still read them
        self.generation = 0      # bumped by every (re-)registration: captured HIP graphs hold the planes' addresses (graphs.py)

    @staticmethod
    def ok(w):
        O, I = w.shape[:2]
        if w.dim() == 2:                                         # a Linear weight = the weight of a 1x1 convolution over rows
            return w.is_cuda and w.dtype == f32 and I % 32 == 0 and O % 32 == 0 and w.is_contiguous()
        return (w.is_cuda and w.dtype == f32 and w.dim() == 4 and tuple(w.shape[2:]) in ((3, 3), (1, 1)) and I % 32 == 0 and O % 32 == 0
                and w.permute(0, 2, 3, 1).is_contiguous())

    @staticmethod
    def taps(w):
        return w.shape[2] * w.shape[3] if w.dim() == 4 else 1

    def _records(self, items, weights):
        """Device table (pt_conv_weight_item records) of `items` = [(key, entry)], `weights` their tensors -> (table, n, blocks)."""
        rec = np.zeros(len(items), dtype=np.dtype([('w', '<u8'), ('dst', '<u8'), ('plane', '<i8'), ('O', '<i4'), ('I', '<i4'),
                                                    ('mode', '<i4'), ('first', '<i4'), ('taps', '<i4'), ('np', '<i4'),
                                                    ('scale', '<u8')]))
        first = 0
        dev = None
        for i, (((_, mode, _sp, npl), (_ref, ptr, sp, scale, _ver)), w) in enumerate(zip(items, weights)):
            O, I = w.shape[:2]
            taps = self.taps(w)
            rows, k = self._dims(mode, O, I, taps)
            rec[i] = (ptr, sp.planes.data_ptr(), sp.planes.shape[1], O, I, mode, first, taps, npl, scale.data_ptr() if scale is not None else 0)
            first += ((rows + 15) // 16) * (k // 32)
            dev = w.device
        return torch.from_numpy(rec.view(np.uint8)).to(dev), len(items), first

    def _build_table(self, weights):
        """`weights`: the registered weights, in the order of self.ent (strong references held by the caller)."""
        self.table = self._records(list(self.ent.items()), weights)

    @staticmethod
    def _alive(v):
        w = v[0]()
        return w is not None and w.data_ptr() == v[1]

    @staticmethod
    def _dims(mode, O, I, taps):
        """(rows, k) of the operand matrix: mode 0 [O][taps I] (forward), 1 [I][taps O] (input gradient of the convolution),
        2 [taps I][O] (the forward matrix transposed: d col = g W of a convolution over gathered columns)."""
        return (taps * I, O) if mode == 2 else ((I, taps * O) if mode else (O, taps * I))

    def get(self, w, dgrad, scale=None, np_=3):
        import weakref
        mode = 2 if dgrad == 2 else int(bool(dgrad))
        key = (id(w), mode, scale.data_ptr() if scale is not None else 0, np_)
        e = self.ent.get(key)
        if e is not None and (e[0]() is not w or e[1] != w.data_ptr()):
            e = None                                            # the id was reused, or the storage moved (re-layout of the flat buffer)
        if e is None:
            # drop entries whose weight is gone or moved, then register this triple with planes of its own
            self.ent = {k: v for k, v in self.ent.items() if self._alive(v) and k != key}
            O, I = w.shape[:2]
            taps = self.taps(w)
            rows, k = self._dims(mode, O, I, taps)
            n = ((rows + 15) // 16) * ((k + 31) // 32) * 512
            sp = SplitPlanes(torch.empty((np_, n), dtype=torch.bfloat16, device=w.device), rows, k)
            e = self.ent[key] = [weakref.ref(w), w.data_ptr(), sp, scale, -1]
            self.generation += 1
            if self.table is not None:
                self._retired.append(self.table)                # (a refresh on the other stream may still be reading it)
            self.table = None
            if self.epoch == PARAM_EPOCH[0]:
                # the other entries are current (a weight's first use in the middle of an iteration - the first iteration, the switch
                # of the burn-in phase - possibly while the OTHER stream reads their planes, round-4 advice): split this entry alone
                # instead of re-splitting every registered plane under the reader
                one = self._records([(key, e)], [w])
                self._retired.append(one)
                hip.call('pt_conv_weight_planes_batch', one[0], one[1], one[2])
                e[4] = w._version
                # the full table for the NEXT epoch's refresh is uploaded now (a blocking upload: registrations happen in the first
                # iteration of a phase, a steady-state iteration never synchronises)
                strong = [v[0]() for v in self.ent.values()]
                if all(x is not None for x in strong):
                    self._build_table(strong)
                return sp
            self.epoch = -1
        if e[4] != w._version:                                  # torch wrote the weight (load_state_dict, init, copy_): re-split
            self.epoch = -1
        if self.epoch != PARAM_EPOCH[0]:                        # (a table dropped by a mid-epoch registration is rebuilt by the next refresh)
            self.refresh()
        return self.ent[key][2]

    def refresh(self):
        """Re-split every registered (weight, form, scale, plane count) now, on the current stream: one launch."""
        # strong references for the duration of the call: weights of a discarded model sit in reference cycles, and the cyclic
        # collector may run at any allocation below - a weak reference that was alive a line ago can be dead the next
        live, strong = {}, []
        for k, v in self.ent.items():
            w = v[0]()
            if w is not None and w.data_ptr() == v[1]:
                live[k] = v
                strong.append(w)
        if len(live) != len(self.ent):
            self.ent = live
            self.table = None
        if not self.ent:
            return
        if self.table is None:
            self._build_table(strong)
        self._retired.clear()                                   # (a full refresh runs between iterations: both streams have joined)
        tab, n_items, blocks = self.table
        hip.call('pt_conv_weight_planes_batch', tab, n_items, blocks)
        self.epoch = PARAM_EPOCH[0]
        for v, w in zip(self.ent.values(), strong):
            v[4] = w._version


_CONV_W = _ConvWeightPlanes()


def refresh_conv_weight_planes():
    """Bring the cached weight planes up to the current parameter epoch NOW, on the current stream.  A caller that is about to use
    them from two streams (the teacher pass on its side stream next to the student's, detectors._teacher_fork) calls this first, so
    that neither stream's first convolution launches the refresh while the other reads the planes."""
    c = _CONV_W
    if c.ent and c.epoch != PARAM_EPOCH[0]:
        c.refresh()


def _conv_weight_planes(w, dgrad, scale=None, np_=3):
    """Split planes of a [O, I, KH, KW] weight as the [O, taps I] matrix with k = (ky, kx, i) (forward; a channels_last weight IS
    that matrix) or as w'[i, (KH - 1 - ky, KW - 1 - kx), o] (input gradient), optionally with a per-output-channel scale folded in;
    cached until the parameters change (PARAM_EPOCH / the weight's version counter).  Channels_last weights go through the batched
    refresh (_ConvWeightPlanes); any other layout is split on its own."""
    c = _CONV_W
    e = c.ent.get((id(w), 2 if dgrad == 2 else (1 if dgrad else 0), scale.data_ptr() if scale is not None else 0, np_))
    if (e is not None and c.epoch == PARAM_EPOCH[0] and e[0]() is w and e[1] == w.data_ptr()
            and e[4] == w._version):
        return e[2]                                              # (the steady state: a registered weight, planes of this parameter epoch)
    if _ConvWeightPlanes.ok(w):
        return c.get(w, dgrad, scale, np_)
    assert np_ == 3, 'one-plane (bf16) operands need a channels_last weight of the batched plane cache'
    key = (id(w), 'conv', scale.data_ptr() if scale is not None else 0)
    ent = _SPLIT_W_CACHE.get(key)
    if ent is None or ent[0] != PARAM_EPOCH[0] or ent[1] != w.data_ptr() or ent[4] != w._version:
        ent = [PARAM_EPOCH[0], w.data_ptr(), None, None, w._version]
        _SPLIT_W_CACHE[key] = ent
    i = 3 if dgrad else 2
    if ent[i] is None:
        wd = w.detach()
        if scale is not None:
            wd = wd * scale.view(-1, 1, 1, 1)
        O, I = wd.shape[:2]
        taps = wd.shape[2] * wd.shape[3]
        m = (wd.flip(2, 3).permute(1, 2, 3, 0).reshape(I, taps * O) if dgrad else wd.permute(0, 2, 3, 1).reshape(O, taps * I))
        ent[i] = split_bf16x3(m.contiguous())
    return ent[i]


def _split_rows(x_nhwc_rows, relu_of=None, col_scale=None, masked_out=None):
    """[P, C] fp32 contiguous -> ROW-MAJOR bf16 planes [3, (P + 1) * C] (row P = zeros), pt_split_bf16x3_rows.
    relu_of / col_scale / masked_out: its backward preparation (ReLU mask, BatchNorm scale, fp32 copy of the effective gradient)."""
    P, C = x_nhwc_rows.shape
    n = (P + 1) * C
    xp = torch.empty((3, n), dtype=torch.bfloat16, device=x_nhwc_rows.device)
    hip.call('pt_split_bf16x3_rows', x_nhwc_rows, C, P, C, relu_of, col_scale, masked_out, xp, n)
    return xp


def _conv3x3_from_planes(xp, P, C, B, H, W, wp, bias, relu, scale=None):
    """planes of [B*H*W, Cin] -> [B*H*W, Cout] fp32 (pt_conv3x3_bf16x6_nhwc)."""
    out = torch.empty((P, wp.rows), dtype=f32, device=xp.device)
    hip.call('pt_conv3x3_bf16x6_nhwc', xp, xp.shape[1], wp.planes, wp.planes.shape[1], out, wp.rows, bias, scale, B, H, W, C, wp.rows,
             int(bool(relu)), 0)
    return out


def conv3x3_wgrad_ok(Cin, Cout):
    return Cin % 128 == 0 and Cout % 128 == 0


def _conv3x3_wgrad(gp, xp, B, H, W, Cin, Cout):
    """planes of the output gradient [P, Cout] and of the activations [P, Cin] -> the weight gradient as a channels_last
    [Cout, Cin, 3, 3] tensor (pt_conv3x3_wgrad_bf16x6_nhwc: split over pixel chunks, summed in a fixed order)."""
    S = hip.call('pt_conv3x3_wgrad_bf16x6_splits', B, H, W, Cin, Cout)
    n = Cout * 9 * Cin
    ws = torch.empty((S * n,), dtype=f32, device=gp.device)
    dw = torch.empty((Cout, 3, 3, Cin), dtype=f32, device=gp.device)
    hip.call('pt_conv3x3_wgrad_bf16x6_nhwc', gp, gp.shape[1], xp, xp.shape[1], dw, ws, S * n, B, H, W, Cin, Cout, S)
    return dw.permute(0, 3, 1, 2)


class _SplitConv3x3(torch.autograd.Function):
    """y = [relu](conv2d(x, w, b, stride 1, padding 1)) for channels_last fp32 tensors: forward, input gradient and weight
    gradient as implicit GEMMs of six bf16 MFMA products per fp32 product with fp32 accumulation (csrc/gemm_split.hip; no im2col
    buffer).  The forward's activation planes are kept for the weight gradient (its reduce dimension is their row index); shapes
    the weight-gradient kernel does not take (channels not multiples of 128) fall back to the library for that one product."""

    @staticmethod
    def forward(ctx, x, w, b, relu, scale):
        B, C, H, W = x.shape
        O = w.shape[0]
        rows = x.permute(0, 2, 3, 1).reshape(B * H * W, C)                      # a view of a channels_last tensor
        xp = _split_rows(rows)
        y = _conv3x3_from_planes(xp, B * H * W, C, B, H, W, _conv_weight_planes(w, False), b, relu, scale).view(B, H, W, O).permute(0, 3, 1, 2)
        ctx.relu, ctx.has_bias = relu, b is not None
        ctx.own_wgrad = bool(_SPLIT_WGRAD and conv3x3_wgrad_ok(C, O))
        keep = w.requires_grad and ctx.own_wgrad
        ctx.save_for_backward(None if keep else x, w, y if relu else None, scale, xp if keep else None)
        ctx.xshape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y, scale, xp = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last)
        B, O, H, W = gy.shape
        C = ctx.xshape[1]
        P = B * H * W
        rows = gy.permute(0, 2, 3, 1).reshape(P, O)
        need_w = ctx.needs_input_grad[1]
        own = need_w and xp is not None
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        masked = ctx.relu or scale is not None
        # the fp32 effective gradient (ReLU mask, BatchNorm scale) is only written for a consumer outside the planes
        need_eff = masked and (need_b or (need_w and not own))
        eff = torch.empty((P, O), dtype=f32, device=gy.device) if need_eff else (None if masked else rows)
        yrows = y.permute(0, 2, 3, 1).reshape(P, O) if ctx.relu else None
        gx = gw = gb = None
        if ctx.needs_input_grad[0] or own:
            # ONE pass: ReLU mask (from y), BatchNorm scale, [the fp32 effective gradient] and the split planes
            gp = _split_rows(rows, relu_of=yrows, col_scale=scale, masked_out=eff if need_eff else None)
            if ctx.needs_input_grad[0]:
                gx = _conv3x3_from_planes(gp, P, O, B, H, W, _conv_weight_planes(w, True), None, False).view(B, H, W, C).permute(0, 3, 1, 2)
            if own:
                gw = _conv3x3_wgrad(gp, xp, B, H, W, C, O)
        elif need_eff:                         # an input that takes no gradient (not on the training path): plain element-wise ops
            e = rows * (yrows > 0) if ctx.relu else rows
            eff = e * scale if scale is not None else e
        if need_w and not own:
            geff = eff.view(B, H, W, O).permute(0, 3, 1, 2)
            gw = torch.ops.aten.convolution_backward(geff, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        if need_b:
            gb = eff.sum(0)                   # (only without a scale: a frozen BatchNorm's shift takes no gradient)
        return gx, gw, gb, None, None


def split_conv3x3_ok(x, conv):
    """The tower shape: 3x3 / stride 1 / pad 1 / one group, Cin % 32 == 0, Cout % 128 == 0, channels_last fp32 on the device, enough
    pixels to fill 256 CUs (below ~8 000 pixels the library's kernels win)."""
    return (x.is_cuda and x.dtype == f32 and x.dim() == 4 and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels % 32 == 0
            and conv.out_channels % 128 == 0 and x.shape[0] * x.shape[2] * x.shape[3] >= 8192
            and x.is_contiguous(memory_format=torch.channels_last) and not torch.is_autocast_enabled())


def split_conv3x3(x, weight, bias=None, relu=False, scale=None):
    """[relu](conv2d(x, weight, padding 1) * scale[o] + bias[o]); scale / bias may be the (scale, shift) of a frozen BatchNorm."""
    return _SplitConv3x3.apply(x, weight, bias, bool(relu), scale)


class _GroupNormCL(torch.autograd.Function):
    """y = [relu](group_norm(x, G, gamma, beta, eps)) for a channels_last fp32 [N, C, H, W] tensor, staying channels_last
    (csrc/group_norm.hip: two launches forward, three backward, no layout copies; fixed-order reductions)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, relu):
        N, C, H, W = x.shape
        xr = x.permute(0, 2, 3, 1)                                             # [N, H, W, C] view of the channels_last memory
        y = torch.empty_like(xr)
        mean = torch.empty((N, G), dtype=f32, device=x.device)
        rstd = torch.empty((N, G), dtype=f32, device=x.device)
        ws = torch.empty((hip.call('pt_group_norm_cl_workspace_bytes', N, H * W, C, G) + 15) // 16 * 2, dtype=torch.float64, device=x.device)
        hip.call('pt_group_norm_cl_fwd', xr, gamma, beta, N, H * W, C, G, float(eps), int(relu), y, mean, rstd, ws)
        ctx.save_for_backward(x, gamma, mean, rstd, y if relu else None)
        ctx.G = G
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        x, gamma, mean, rstd, y = ctx.saved_tensors
        N, C, H, W = x.shape
        G = ctx.G
        gy = gy.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        gx = torch.empty_like(gy)
        gg, gb = torch.empty_like(gamma), torch.empty_like(gamma)
        ws = torch.empty((hip.call('pt_group_norm_cl_workspace_bytes', N, H * W, C, G) + 15) // 16 * 2, dtype=torch.float64, device=x.device)
        hip.call('pt_group_norm_cl_bwd', gy, x.permute(0, 2, 3, 1), y, gamma, mean, rstd, N, H * W, C, G, gx, gg, gb, ws)
        return gx.permute(0, 3, 1, 2), gg, gb, None, None, None


def group_norm_cl_ok(x, gn):
    """channels_last fp32 on the device, a shape the kernels take (C / 4 divides 256, groups of a multiple of 4 channels that divide
    256, at most 64 groups), both affine parameters present and training together."""
    C, G = gn.num_channels, gn.num_groups
    return (x.is_cuda and x.dtype == f32 and x.dim() == 4 and gn.affine and C % 4 == 0 and C <= 1024 and 256 % (C // 4) == 0
            and C % G == 0 and (C // G) % 4 == 0 and 256 % (C // G) == 0 and G <= 64 and x.shape[0] * x.shape[2] * x.shape[3] > 0
            and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous() and not torch.is_autocast_enabled())


def group_norm_cl(x, gn, relu=False):
    """torch.nn.GroupNorm `gn` (+ ReLU) on a channels_last activation; the result stays channels_last."""
    return _GroupNormCL.apply(x, gn.weight, gn.bias, int(gn.num_groups), float(gn.eps), bool(relu))


class _AffineReLU(torch.autograd.Function):
    """y = [relu](x*scale[c] + shift[c] [+ residual]) in place on x (x is a fresh conv output)."""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu):
        cl = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        C = x.shape[1]
        inner = 1 if cl else x.shape[2] * x.shape[3]
        xb = x.permute(0, 2, 3, 1) if cl else x
        rb = None
        if residual is not None:
            rb = (residual.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1) if cl
                  else residual.contiguous())
        hip.call('pt_affine_relu_fwd', xb, scale, shift, rb, x.numel(), C, inner, int(relu), xb)
        ctx.mark_dirty(x)
        ctx.save_for_backward(x if relu else None, scale)
        ctx.cfg = (cl, C, inner, bool(relu), residual is not None)
        return x

    @staticmethod
    def backward(ctx, g):
        y, scale = ctx.saved_tensors
        cl, C, inner, relu, has_res = ctx.cfg
        if cl:
            g = g.contiguous(memory_format=torch.channels_last)
            gb = g.permute(0, 2, 3, 1)
            yb = y.permute(0, 2, 3, 1) if y is not None else None
        else:
            g = g.contiguous()
            gb, yb = g, y
        need_x = ctx.needs_input_grad[0]
        gx = torch.empty_like(g) if need_x else None
        gres = torch.empty_like(g) if (has_res and ctx.needs_input_grad[3]) else None
        if gx is None and gres is None:
            return None, None, None, None, None
        hip.call('pt_affine_relu_bwd', gb, yb, scale, g.numel(), C, inner, int(relu),
                 (gx.permute(0, 2, 3, 1) if cl else gx) if gx is not None else None,
                 (gres.permute(0, 2, 3, 1) if cl else gres) if gres is not None else None)
        return gx, None, None, gres, None


class _AffineReLUTrain(torch.autograd.Function):
    """y = [relu](BN_eval(x) [+ residual]) for a BatchNorm whose weight/bias train while it normalises with its
    running statistics (config 5).  x is kept (the weight gradient needs it), y is a new tensor."""

    @staticmethod
    def forward(ctx, x, weight, bias, mean, var, eps, residual, relu):
        C = x.shape[1]
        with torch.no_grad():
            rstd = torch.rsqrt(var.float() + eps)
            scale = (weight.float() * rstd).contiguous()
            shift = (bias.float() - mean.float() * scale).contiguous()
        xb = x.permute(0, 2, 3, 1)
        y = torch.empty_like(x)
        rb = residual.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1) if residual is not None else None
        hip.call('pt_affine_relu_fwd', xb, scale, shift, rb, x.numel(), C, 1, int(relu), y.permute(0, 2, 3, 1))
        ctx.save_for_backward(x, y if relu else None, scale, rstd, mean)
        ctx.cfg = (C, bool(relu), residual is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y, scale, rstd, mean = ctx.saved_tensors
        C, relu, has_res = ctx.cfg
        g = g.contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(g) if ctx.needs_input_grad[0] else None
        gres = torch.empty_like(g) if (has_res and ctx.needs_input_grad[6]) else None
        sums = torch.empty((2, C), dtype=f32, device=g.device)
        ws = torch.empty((hip.call('pt_affine_train_rows', g.numel(), C), 2 * C), dtype=f32, device=g.device)
        hip.call('pt_affine_relu_bwd_train', g.permute(0, 2, 3, 1), y.permute(0, 2, 3, 1) if y is not None else None,
                 x.permute(0, 2, 3, 1), scale, g.numel(), C, int(relu),
                 gx.permute(0, 2, 3, 1) if gx is not None else None, gres.permute(0, 2, 3, 1) if gres is not None else None,
                 sums, ws)
        gw = (sums[1] - mean.float() * sums[0]) * rstd          # d/dw of w*rstd*(x - mean) + b
        return gx, gw, sums[0], None, None, None, gres, None


_ONES = {}


class _BiasReLU(torch.autograd.Function):
    """y = [relu](x + bias[c]) in place on a fresh fp32 channels_last convolution output; the backward produces grad_x and the
    bias gradient in ONE pass (mmcv ConvModule does conv-with-bias then ReLU: add, clamp / threshold, channel sum)."""

    @staticmethod
    def forward(ctx, x, bias, relu):
        C = x.shape[1]
        key = (C, x.device)
        if key not in _ONES:
            _ONES[key] = torch.ones(C, dtype=f32, device=x.device)
        xb = x.permute(0, 2, 3, 1)
        hip.call('pt_affine_relu_fwd', xb, _ONES[key], bias, None, x.numel(), C, 1, int(relu), xb)
        ctx.mark_dirty(x)
        ctx.save_for_backward(x if relu else None)
        ctx.cfg = (C, bool(relu), key)
        return x

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        C, relu, key = ctx.cfg
        g = g.contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(g) if ctx.needs_input_grad[0] else None
        sums = torch.empty((2, C), dtype=f32, device=g.device)
        ws = torch.empty((hip.call('pt_affine_train_rows', g.numel(), C), 2 * C), dtype=f32, device=g.device)
        hip.call('pt_affine_relu_bwd_train', g.permute(0, 2, 3, 1), y.permute(0, 2, 3, 1) if y is not None else None, None,
                 _ONES[key], g.numel(), C, int(relu), gx.permute(0, 2, 3, 1) if gx is not None else None, None, sums, ws)
        return gx, sums[0], None


def bias_relu_(x, bias, relu=True):
    """Convolution-bias (+ ReLU) epilogue on a fresh conv output: x fp32 [N,C,H,W] channels_last with C/4 dividing 256 (or a
    multiple of it); anything else must take the caller's unfused path (`bias_relu_ok`)."""
    return _BiasReLU.apply(x, bias, relu)


def bias_relu_ok(x, bias):
    C = x.shape[1]
    G = C // 4
    return (x.is_cuda and x.dtype == f32 and bias is not None and bias.dtype == f32 and x.dim() == 4 and C % 4 == 0 and x.numel() > 0
            and ((G <= 256 and 256 % G == 0) or G % 256 == 0)
            and x.is_contiguous(memory_format=torch.channels_last) and (C == 1 or not x.is_contiguous()))


class _UpsampleAdd(torch.autograd.Function):
    """out = a + nearest_upsample(b -> a's size) on channels_last maps (pt_upsample_add_*): one pass instead of torch's
    upsample kernel + add; backward: grad_a is grad_out itself, grad_b one gather pass."""

    @staticmethod
    def forward(ctx, a, b):
        N, C, Ha, Wa = a.shape
        Hb, Wb = b.shape[2:]
        out = torch.empty_like(a)
        bf = a.dtype == torch.bfloat16
        hip.call('pt_upsample_add_fwd', a.permute(0, 2, 3, 1), b.permute(0, 2, 3, 1), N, Ha, Wa, Hb, Wb, C, int(bf),
                 out.permute(0, 2, 3, 1))
        ctx.cfg = (N, C, Ha, Wa, Hb, Wb, bf)
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, Ha, Wa, Hb, Wb, bf = ctx.cfg
        gb = None
        if ctx.needs_input_grad[1]:
            g = g.contiguous(memory_format=torch.channels_last)
            gb = torch.empty((N, C, Hb, Wb), dtype=g.dtype, device=g.device).contiguous(memory_format=torch.channels_last)
            hip.call('pt_upsample_add_bwd', g.permute(0, 2, 3, 1), N, Ha, Wa, Hb, Wb, C, int(bf), gb.permute(0, 2, 3, 1))
        return (g if ctx.needs_input_grad[0] else None), gb


def _cl_dense(t):
    """A 4-D tensor whose memory is dense NHWC (a [N,1,H,W] or [N,C,1,1] tensor is both layouts at once)."""
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def upsample_add(a, b):
    """a + F.interpolate(b, size=a.shape[2:], mode='nearest') (necks/fpn.py:165-173, ps_fpn.py:64-72).  The fused kernel takes
    CUDA channels_last fp32 / bf16 maps of one dtype; any other layout takes torch's two kernels."""
    C = a.shape[1]
    if (a.is_cuda and a.dtype == b.dtype and a.dtype in (f32, torch.bfloat16) and _cl_dense(a) and _cl_dense(b)
            and a.shape[:2] == b.shape[:2] and C % (8 if a.dtype == torch.bfloat16 else 4) == 0
            and a.shape[0] * max(a.shape[2], b.shape[2]) <= 65535 and a.numel() > 0):
        return _UpsampleAdd.apply(a, b)
    return a + torch.nn.functional.interpolate(b, size=a.shape[2:], mode='nearest')


def bn_eval_relu(x, bn, residual=None, relu=True):
    """Fused eval-mode BatchNorm with a TRAINABLE affine (+ residual) (+ ReLU); x: fp32, channels_last."""
    return _AffineReLUTrain.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, residual, relu)


class _AffineReLUBf16(torch.autograd.Function):
    """The frozen-BN epilogue on a channels_last bf16 activation (bf16 autocast backbone), in place on x."""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu):
        C = x.shape[1]
        xb = x.permute(0, 2, 3, 1)
        rb = residual.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1) if residual is not None else None
        hip.call('pt_affine_relu_fwd_bf16', xb, scale, shift, rb, x.numel(), C, int(relu), xb)
        ctx.mark_dirty(x)
        ctx.save_for_backward(x if relu else None, scale)
        ctx.cfg = (C, bool(relu), residual is not None)
        return x

    @staticmethod
    def backward(ctx, g):
        y, scale = ctx.saved_tensors
        C, relu, has_res = ctx.cfg
        g = g.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(g) if ctx.needs_input_grad[0] else None
        gres = torch.empty_like(g) if (has_res and ctx.needs_input_grad[3]) else None
        if gx is None and gres is None:
            return None, None, None, None, None
        hip.call('pt_affine_relu_bwd_bf16', g.permute(0, 2, 3, 1), y.permute(0, 2, 3, 1) if y is not None else None, scale,
                 g.numel(), C, int(relu), gx.permute(0, 2, 3, 1) if gx is not None else None,
                 gres.permute(0, 2, 3, 1) if gres is not None else None)
        return gx, None, None, gres, None


def affine_relu_bf16_(x, scale, shift, residual=None, relu=True):
    return _AffineReLUBf16.apply(x, scale, shift, residual, relu)


def affine_relu_(x, scale, shift, residual=None, relu=True):
    """Frozen-BN epilogue (pt_affine_relu_*): x must be fp32, 4-D, dense NCHW or channels_last."""
    return _AffineReLU.apply(x, scale, shift, residual, relu)


def affine_relu_maxpool(x, scale, shift):
    """max_pool2d(relu(x * scale + shift), 3, 2, 1) of a channels_last fp32 map that takes no gradient, in one pass
    (pt_affine_relu_maxpool_fwd: the frozen stem's norm1 -> relu -> maxpool, backbones/resnet.py:633-640)."""
    B, C, H, W = x.shape
    assert x.dtype == f32 and x.is_cuda and C % 4 == 0 and x.permute(0, 2, 3, 1).is_contiguous() and not x.requires_grad
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, Ho, Wo, C), dtype=f32, device=x.device)
    hip.call('pt_affine_relu_maxpool_fwd', x.permute(0, 2, 3, 1), scale, shift, B, H, W, C, y)
    return y.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------ NMS --

def nms(boxes, scores, iou_threshold, class_ids=None):
    """mmcv.ops.nms contract: returns (dets [M,5], keep [M]) in descending score order."""
    N = boxes.shape[0]
    if N == 0:
        return torch.cat([boxes, scores[:, None]], 1), torch.empty((0,), dtype=torch.long, device=boxes.device)
    order = torch.sort(scores, descending=True, stable=True)[1]
    sb = _f(boxes[order])
    cid = class_ids[order].to(i32).contiguous() if class_ids is not None else None
    ws = torch.empty((N * ((N + 63) // 64),), dtype=torch.int64, device=boxes.device)
    keep_m = torch.empty((N,), dtype=u8, device=boxes.device)
    hip.call('pt_nms_sorted', sb, cid, N, float(iou_threshold), ws, keep_m)
    keep = order[keep_m.bool()]
    return torch.cat([boxes[keep], scores[keep, None]], 1), keep


NMS_MAXN = 32768       # csrc/nms.hip: candidates whose suppression bitmasks one launch holds


def batched_nms(boxes, scores, idxs, nms_cfg):
    """mmcv.ops.batched_nms (class-aware) for nms_cfg = dict(type='nms', iou_threshold=...).  Above NMS_MAXN candidates
    (a 5-level head at nms_pre 3000 x 8 classes can reach ~88 000 at a low score threshold) the classes are processed
    one after the other, as mmcv does above its `split_thr`; a single class above the limit keeps its best NMS_MAXN."""
    thr = nms_cfg.get('iou_threshold', 0.5)
    if boxes.shape[0] <= NMS_MAXN:
        return nms(boxes, scores, thr, class_ids=idxs)
    keeps = []
    for c in torch.unique(idxs).tolist():
        sel = (idxs == c).nonzero(as_tuple=False).squeeze(1)
        if sel.numel() > NMS_MAXN:
            sel = sel[torch.topk(scores[sel], NMS_MAXN, sorted=False)[1]]
        keeps.append(sel[nms(boxes[sel], scores[sel], thr)[1]])
    keep = torch.cat(keeps)
    keep = keep[torch.sort(scores[keep], descending=True, stable=True)[1]]
    return torch.cat([boxes[keep], scores[keep, None]], 1), keep


def box_iou_rotated(b1, b2, aligned=False):
    M, N = b1.shape[0], b2.shape[0]
    out = torch.empty((M,) if aligned else (M, N), dtype=f32, device=b1.device)
    hip.call('pt_box_iou_rotated', _f(b1), _f(b2), M, N, int(aligned), out)
    return out


def nms_rotated(dets, scores, iou_threshold):
    """mmcv.ops.nms_rotated contract: (dets [M,6], keep [M])."""
    N = dets.shape[0]
    if N == 0:
        return torch.cat([dets, scores[:, None]], 1), torch.empty((0,), dtype=torch.long, device=dets.device)
    order = torch.sort(scores, descending=True, stable=True)[1]
    sd = _f(dets[order][:, :5])
    ws = torch.empty((N * ((N + 63) // 64),), dtype=torch.int64, device=dets.device)
    keep_m = torch.empty((N,), dtype=u8, device=dets.device)
    hip.call('pt_nms_rotated_sorted', sd, N, float(iou_threshold), ws, keep_m)
    keep = order[keep_m.bool()]
    return torch.cat([dets[keep], scores[keep, None]], 1), keep


def nms_rotated_mask(dets_sorted, iou_threshold, keep_out=None):
    """Sync-free form: dets already sorted by descending score -> keep mask (uint8 [N]).  keep_out: a contiguous uint8 [N]
    buffer already filled with ones (a slice of a batch buffer) to receive the mask."""
    N = dets_sorted.shape[0]
    ws = torch.empty((max(N, 1) * ((N + 63) // 64 + 1),), dtype=torch.int64, device=dets_sorted.device)
    keep_m = keep_out if keep_out is not None else torch.ones((N,), dtype=u8, device=dets_sorted.device)
    if N:
        hip.call('pt_nms_rotated_sorted', _f(dets_sorted[:, :5]), N, float(iou_threshold), ws, keep_m)
    return keep_m


def nms_rotated_mask_segments(dets_sorted, counts, iou_threshold, keep_out):
    """pt_nms_rotated_sorted_segments: the candidates of every image of a batch (rows concatenated, each image's rows sorted by
    descending score; `counts` = rows per image, host ints) in one pair of launches -> keep_out (uint8, pre-filled with ones)."""
    import ctypes
    off = [0]
    for c in counts:
        off.append(off[-1] + int(c))
    words = sum(int(c) * ((int(c) + 63) // 64) for c in counts)
    ws = torch.empty((max(words, 1),), dtype=torch.int64, device=dets_sorted.device)
    hip.call('pt_nms_rotated_sorted_segments', _f(dets_sorted[:, :5]), (ctypes.c_int32 * len(off))(*off), len(counts),
             float(iou_threshold), ws, keep_out)
    return keep_out


def fill_quads_batch_(imgs, quads, alive, img_of, value=255.0):
    """imgs [B,C,H,W] float contiguous (in place); quads [Q,8]; alive uint8 [Q]; img_of int32 [Q]."""
    B, C, H, W = imgs.shape
    hip.call('pt_fill_quads_batch', imgs, B, C, H, W, _f(quads), alive.to(u8).contiguous(), img_of, quads.shape[0], float(value))
    return imgs


def fill_quads_(img, quads, alive, value=255.0):
    """img [C,H,W] float (in place); quads [Q,8]; alive uint8 [Q] or None."""
    C, H, W = img.shape
    Q = quads.shape[0]
    hip.call('pt_fill_quads', img, C, H, W, _f(quads), alive.to(u8).contiguous() if alive is not None else None, Q,
             float(value))
    return img


# ---------------------------------------------------------------- glue (csrc/glue.hip) --
def box_convert(boxes, mode):
    """pt_box_convert: mode 0 xyxy -> cxcywh, mode 1 cxcywh -> xyxy; boxes [n,4] (no gradient)."""
    b = _f(boxes)
    out = torch.empty_like(b)
    hip.call('pt_box_convert', b, out, b.shape[0], int(mode))
    return out


def aug_geometry(rows, off, B, params, H, W, want_valid=False):
    """pt_aug_geometry: rows [N,2] points or [N,4] boxes of the whole batch, off int32 [B+1], params f32 [B,6]."""
    r = _f(rows)
    N, nc = r.shape
    out = torch.empty_like(r)
    valid = torch.empty((N,), dtype=u8, device=r.device) if want_valid else None
    hip.call('pt_aug_geometry', r, out, valid, off, B, N, nc, params, float(H), float(W))
    return (out, valid) if want_valid else out


def black_paper_rects(gt, goff, B, prior, dense_n, draws, cls, imgsize):
    """pt_black_paper_rects -> (table [T,6], key int64 [T], exist uint8 [T]) with T = 2 sumG + 10 B."""
    sumG = gt.shape[0]
    T = 2 * sumG + 10 * B
    dev = prior.device
    table = torch.empty((T, 6), dtype=f32, device=dev)
    key = torch.empty((T,), dtype=torch.int64, device=dev)
    exist = torch.empty((T,), dtype=u8, device=dev)
    hip.call('pt_black_paper_rects', _f(gt) if sumG else None, gt.shape[1] if sumG else 4, goff, B, _f(prior), prior.shape[0],
             int(dense_n), _f(draws) if sumG else None, cls.to(i32).contiguous() if sumG else None, sumG, float(imgsize), table,
             key, exist)
    return table, key, exist


def black_paper_sorted(table, order, exist, imgsize):
    """pt_black_paper_sorted -> (sorted [T,6], nms_in [T,5], polys [T,8], hull [T,4], pre uint8 [T])."""
    T = table.shape[0]
    dev = table.device
    srt = torch.empty((T, 6), dtype=f32, device=dev)
    nms_in = torch.empty((T, 5), dtype=f32, device=dev)
    polys = torch.empty((T, 8), dtype=f32, device=dev)
    hull = torch.empty((T, 4), dtype=f32, device=dev)
    pre = torch.empty((T,), dtype=u8, device=dev)
    hip.call('pt_black_paper_sorted', table, order.contiguous(), exist, T, float(imgsize), srt, nms_in, polys, hull, pre)
    return srt, nms_in, polys, hull, pre


# ------------------------------------------------- oriented boxes (OBB variant) --
class _DiffIoURotated(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b1, b2):
        b1, b2 = _f(b1), _f(b2)
        N = b1.shape[0]
        iou = torch.empty((N,), dtype=f32, device=b1.device)
        hip.call('pt_diff_iou_rotated_fwd', b1, b2, N, iou)
        ctx.save_for_backward(b1, b2)
        return iou

    @staticmethod
    def backward(ctx, g):
        b1, b2 = ctx.saved_tensors
        gb = torch.empty_like(b1)
        hip.call('pt_diff_iou_rotated_bwd', b1, b2, _f(g), b1.shape[0], gb)
        return gb, None


def diff_iou_rotated_2d(box1, box2):
    """mmcv.ops.diff_iou_rotated_2d contract: box1/box2 [B,N,5] -> IoU [B,N]; gradients flow to
    box1 only (the reference's losses detach the target)."""
    B, N = box1.shape[:2]
    return _DiffIoURotated.apply(box1.reshape(-1, 5), box2.detach().reshape(-1, 5)).reshape(B, N)


class _RoIAlignRotated(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, out_size, scale, sample_num, aligned, clockwise):
        B, C, H, W = feat.shape
        in_dtype = feat.dtype
        feat = feat.float()
        cl = feat.is_contiguous(memory_format=torch.channels_last) and not feat.is_contiguous()
        fbuf = feat.permute(0, 2, 3, 1) if cl else feat.contiguous()
        rois = _f(rois)
        K = rois.shape[0]
        out = torch.empty((K, C, out_size, out_size), dtype=f32, device=feat.device)
        hip.call('pt_roi_align_rotated_fwd', fbuf, rois, B, C, H, W, K, out_size, float(scale), int(sample_num),
                 int(aligned), int(clockwise), int(cl), out)
        ctx.save_for_backward(rois)
        ctx.cfg = (B, C, H, W, out_size, float(scale), int(sample_num), int(aligned), int(clockwise), cl, in_dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        rois, = ctx.saved_tensors
        B, C, H, W, out_size, scale, sn, aligned, cw, cl, in_dtype = ctx.cfg
        gbuf = torch.zeros((B, H, W, C) if cl else (B, C, H, W), dtype=f32, device=g.device)
        hip.call('pt_roi_align_rotated_bwd', _f(g), rois, B, C, H, W, rois.shape[0], out_size, scale, sn, aligned, cw,
                 int(cl), gbuf)
        return (gbuf.permute(0, 3, 1, 2) if cl else gbuf).to(in_dtype), None, None, None, None, None, None


def roi_align_rotated(feat, rois, out_size, spatial_scale, sample_num=0, aligned=True, clockwise=False):
    """mmcv.ops.roi_align_rotated: rois [K,6] = (batch, cx, cy, w, h, theta)."""
    return _RoIAlignRotated.apply(feat, rois, int(out_size), spatial_scale, int(sample_num), bool(aligned), bool(clockwise))


# ------------------------------------------------------ (modulated) deformable conv --
_DCN_PLANES = os.environ.get('PT_DCN_PLANES', '1') != '0'      # 0: the deformable convolution's GEMMs stay with hipBLASLt (round 4)


class _DeformConv(torch.autograd.Function):
    """mmcv.ops.(modulated_)deform_conv2d, groups = 1: gather (HIP) -> contraction and the mirrored backward.  channels_last with
    channel counts the matrix kernels tile: the contraction, its weight / bias gradient and d col run on pt_conv_bf16x6 /
    pt_conv_wgrad_bf16x6 (round 5); otherwise a library GEMM.
    A channels_last input (the training layout) stays NHWC end to end (pt_deform_*_cl: wave-wide contiguous gathers, col
    [B*L, K*C], output written as NHWC by the GEMM); an NCHW input takes the NCHW kernels."""

    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, stride, padding, dilation, deform_groups):
        B, C, H, W = x.shape
        O, _, kh, kw = weight.shape
        Ho = (H + 2 * padding[0] - (dilation[0] * (kh - 1) + 1)) // stride[0] + 1
        Wo = (W + 2 * padding[1] - (dilation[1] * (kw - 1) + 1)) // stride[1] + 1
        assert offset.shape == (B, 2 * deform_groups * kh * kw, Ho, Wo), (offset.shape, (B, 2 * deform_groups * kh * kw, Ho, Wo))
        geo = (B, C, H, W, kh, kw, padding[0], padding[1], stride[0], stride[1], dilation[0], dilation[1], deform_groups)
        cl = (x.is_cuda and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
              and (C // deform_groups) % 4 == 0)
        ctx.geo, ctx.has_bias, ctx.cl = geo, bias is not None, cl
        if cl:
            x = x.float()
            offset = offset.float().contiguous(memory_format=torch.channels_last)
            mask = mask.float().contiguous(memory_format=torch.channels_last) if mask is not None else None
            col = torch.empty((B * Ho * Wo, kh * kw * C), dtype=f32, device=x.device)
            hip.call('pt_deform_im2col_cl', x.permute(0, 2, 3, 1), offset.permute(0, 2, 3, 1),
                     mask.permute(0, 2, 3, 1) if mask is not None else None, *geo, col)
            M, KC = B * Ho * Wo, kh * kw * C
            ctx.planes = bool(_DCN_PLANES and KC % 128 == 0 and O % 128 == 0 and C % 32 == 0 and weight.dtype == f32
                              and _ConvWeightPlanes.ok(weight) and not torch.is_autocast_enabled() and M > 0)
            if ctx.planes:
                # the contraction on the path's own matrix kernels (round 5; until round 4 a hipBLASLt GEMM): the gathered columns
                # become split planes (H2, or bf16 x 3 when the group fell back), the 3x3 weight's forward planes ARE the
                # [O][(ky, kx, c)] matrix of the column order, bias in the epilogue
                from . import planes as PL
                h2 = PL.use_f16('towers')
                colp, _, _ = PL.split_nhwc(col.view(1, M, 1, KC).permute(0, 3, 1, 2), f16=h2,
                                           census=PL.CENSUS.ptr('towers', f'dcn{C}>{O}:col', x.device) if h2 else 0)
                del col
                _, out = PL.launch_conv(colp, 1, M, 1, KC, _conv_weight_planes(weight, False, None, 2 if h2 else 3), O, 1, 1, 0,
                                        shift=bias.float() if bias is not None else None, want_f32=True)
                ctx.save_for_backward(x, offset, mask, weight, colp)
                return out.view(B, Ho, Wo, O).permute(0, 3, 1, 2)
            wm = weight.float().permute(0, 2, 3, 1).reshape(O, -1)                     # [O, kh*kw*C]: a view of a channels_last weight
            out = torch.nn.functional.linear(col, wm, bias.float() if bias is not None else None)
            ctx.save_for_backward(x, offset, mask, weight, col)
            return out.view(B, Ho, Wo, O).permute(0, 3, 1, 2)
        x, offset = _f(x), _f(offset)
        mask = _f(mask) if mask is not None else None
        col = torch.empty((B, C * kh * kw, Ho * Wo), dtype=f32, device=x.device)
        hip.call('pt_deform_im2col', x, offset, mask, *geo, col)
        out = torch.matmul(weight.float().reshape(O, -1), col).reshape(B, O, Ho, Wo)
        if bias is not None:
            out = out + bias.float().view(1, -1, 1, 1)
        ctx.save_for_backward(x, offset, mask, weight, col)
        return out

    @staticmethod
    def backward(ctx, g):
        x, offset, mask, weight, col = ctx.saved_tensors
        geo = ctx.geo
        B, C, H, W, kh, kw = geo[:6]
        O = weight.shape[0]
        gx = goff = gmask = gw = gb = None
        need_in = ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or (mask is not None and ctx.needs_input_grad[2])
        if ctx.cl:
            g2 = g.float().contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(-1, O)     # [B*L, O]
            gcol = None
            if ctx.planes:
                # weight / bias gradient and d col on the plane kernels: the output gradient enters as planes (H2: with the
                # power-of-two scale of its largest magnitude), d W = g^T col from the two plane sets, d col = g W through the
                # transposed weight planes (mode 2)
                from . import planes as PL
                M, KC = g2.shape[0], kh * kw * C
                h2 = PL.is_h2(col)
                E = PL.grad_to_planes(g2, h2, census=PL.CENSUS.ptr('towers_grad', f'dcn{C}>{O}:gy', g.device) if h2 else 0)
                if ctx.needs_input_grad[3]:
                    dw, db = PL.launch_wgrad(E, col, 1, M, 1, KC, O, 1, 1, 0, want_bias=bool(ctx.has_bias and ctx.needs_input_grad[4]))
                    gw = dw.reshape(O, kh, kw, C).permute(0, 3, 1, 2)
                    gb = db
                elif ctx.has_bias and ctx.needs_input_grad[4]:
                    gb = g2.sum(0)
                if need_in:
                    _, gcol = PL.launch_conv(E, 1, M, 1, O, _conv_weight_planes(weight, 2, None, 2 if h2 else 3), KC, 1, 1, 0,
                                             want_f32=True, x_inv=PL.tail_ptr(E))
            else:
                if ctx.needs_input_grad[3]:
                    gw = torch.matmul(g2.t(), col).view(O, kh, kw, C).permute(0, 3, 1, 2)
                if ctx.has_bias and ctx.needs_input_grad[4]:
                    gb = g2.sum(0)
            if need_in:
                if gcol is None:
                    gcol = torch.matmul(g2, weight.float().permute(0, 2, 3, 1).reshape(O, -1))                  # [B*L, K*C]
                gx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None                                    # channels_last
                goff = torch.empty_like(offset)
                gmask = torch.empty_like(mask) if mask is not None else None
                hip.call('pt_deform_col2im_cl', gcol, x.permute(0, 2, 3, 1), offset.permute(0, 2, 3, 1),
                         mask.permute(0, 2, 3, 1) if mask is not None else None, *geo,
                         gx.permute(0, 2, 3, 1) if gx is not None else None, goff.permute(0, 2, 3, 1),
                         gmask.permute(0, 2, 3, 1) if gmask is not None else None)
            return gx, goff, gmask, gw, gb, None, None, None, None
        g2 = _f(g).reshape(B, O, -1)
        if ctx.needs_input_grad[3]:
            gw = torch.matmul(g2, col.transpose(1, 2)).sum(0).reshape(weight.shape)
        if ctx.has_bias and ctx.needs_input_grad[4]:
            gb = g2.sum((0, 2))
        if need_in:
            gcol = torch.matmul(weight.float().reshape(O, -1).t(), g2).contiguous()          # [B, C*K, L]
            if ctx.needs_input_grad[0]:
                gx = torch.zeros_like(x)
                hip.call('pt_deform_col2im', gcol, offset, mask, *geo, gx)
            goff = torch.empty_like(offset)
            gmask = torch.empty_like(mask) if mask is not None else None
            hip.call('pt_deform_col2im_coord', gcol, x, offset, mask, *geo, goff, gmask)
        return gx, goff, gmask, gw, gb, None, None, None, None


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def deform_conv2d(x, offset, weight, stride=1, padding=0, dilation=1, groups=1, deform_groups=1):
    """mmcv.ops.deform_conv2d (DCNv1, no bias)."""
    assert groups == 1, 'groups=1 is what the path can reach'
    return _DeformConv.apply(x, offset, None, weight, None, _pair(stride), _pair(padding), _pair(dilation), deform_groups)


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deform_groups=1):
    """mmcv.ops.modulated_deform_conv2d (DCNv2)."""
    assert groups == 1, 'groups=1 is what the path can reach'
    return _DeformConv.apply(x, offset, mask, weight, bias, _pair(stride), _pair(padding), _pair(dilation), deform_groups)
