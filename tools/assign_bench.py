#!/usr/bin/env python3
"""HIP-event timing of the two baseline assignment kernels at BASELINE size (800x800, 600 boxes / image, batch 2):
pt_fcos_dense_targets (13 343 points) and pt_max_iou_assign (120 087 anchors)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from point_teacher_amd import functional as F  # noqa: E402
from point_teacher_amd.fcos_baseline import FCOSHead  # noqa: E402
from point_teacher_amd.retina_baseline import AnchorGenerator  # noqa: E402
from point_teacher_amd.synthetic import make_tile  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device('cuda:0')
    sizes = [(100, 100), (50, 50), (25, 25), (13, 13), (7, 7)]
    tiles = [make_tile(i, 800, 8, 600, seed=4, device=dev, max_objects=600) for i in range(2)]
    boxes, labels = [t[1] for t in tiles], [t[2] for t in tiles]
    G = sum(b.shape[0] for b in boxes)
    h = FCOSHead(8, 4, feat_channels=4, stacked_convs=1, strides=[8, 16, 32, 64, 128], center_sampling=True, norm_on_bbox=True,
                 centerness_on_reg=True, norm_cfg=None, conv_bias=True, loss_bbox=dict(type='DIoULoss', loss_weight=1.0)).to(dev)
    pts = h.get_points(sizes, torch.float32, dev)
    t_dense = timed(lambda: h.get_targets(pts, boxes, labels))
    gen = AnchorGenerator(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128])
    anchors = torch.cat(gen.grid_anchors(sizes, dev))
    off, _ = F.make_offsets([int(b.shape[0]) for b in boxes], dev)
    gb = torch.cat(boxes)
    t_iou = timed(lambda: F.max_iou_assign(anchors, gb, off, 2, 0.5, 0.4, 0.0, True, True))
    A, P = anchors.shape[0], sum(p.shape[0] for p in pts)
    print(json.dumps(dict(boxes=G, points=P, anchors=A,
                          fcos_dense_targets_us=round(t_dense, 1), pairs_per_us_dense=round(P * G / t_dense, 0),
                          max_iou_assign_us=round(t_iou, 1), iou_pairs_per_us=round(2 * A * G / t_iou, 0),
                          matrix_bytes_avoided=int(A * G * 4))))


if __name__ == '__main__':
    main()
