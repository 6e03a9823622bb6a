"""point_teacher_amd.ops: the mmcv 1.x signatures a maintainer of the reference imports (SURVEY 8b op surface),
checked on the GPU against the oracle."""
import math

import numpy as np
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def close(a, b, rtol=1e-4, atol=1e-6):
    torch.testing.assert_close(torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double(), rtol=rtol, atol=atol)


def test_sigmoid_focal_loss_signature():
    from point_teacher_amd import ops
    g = torch.Generator().manual_seed(0)
    N, C = 300, 8
    x = torch.randn(N, C, generator=g) * 2
    t = torch.randint(0, C + 1, (N,), generator=g)
    w = torch.rand(C, generator=g)
    ref_elem = R.sigmoid_focal_loss_elem(x, t, 2.0, 0.25)
    xg = x.to(DEV).requires_grad_(True)
    close(ops.sigmoid_focal_loss(xg, t.to(DEV), 2.0, 0.25, None, 'none'), ref_elem)
    close(ops.sigmoid_focal_loss(xg, t.to(DEV), 2.0, 0.25, None, 'sum'), ref_elem.sum())
    close(ops.sigmoid_focal_loss(xg, t.to(DEV)), ref_elem.sum() / N)                       # mmcv: mean over N
    xr = x.clone().requires_grad_(True)
    lr = (R.sigmoid_focal_loss_elem(xr, t, 1.5, 0.3) * w[None]).sum() / N
    lr.backward()
    l = ops.sigmoid_focal_loss(xg, t.to(DEV), 1.5, 0.3, w.to(DEV), 'mean')
    l.backward()
    close(l, lr)
    close(xg.grad, xr.grad, rtol=1e-3, atol=1e-7)
    assert float(ops.SigmoidFocalLoss(2.0, 0.25)(xg.detach(), t.to(DEV))) == pytest.approx(float(ref_elem.sum() / N), rel=1e-4)


def test_nms_signatures():
    from point_teacher_amd import ops
    g = torch.Generator().manual_seed(1)
    n = 400
    c = torch.rand(n, 2, generator=g) * 100
    wh = torch.rand(n, 2, generator=g) * 20 + 2
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1)
    scores = torch.rand(n, generator=g)
    dets, keep = ops.nms(boxes.to(DEV), scores.to(DEV), 0.5)
    assert torch.equal(keep.cpu(), R.nms(boxes, scores, 0.5)) and dets.shape == (keep.numel(), 5)
    plus1 = boxes.clone(); plus1[:, 2:] += 1
    assert torch.equal(ops.nms(boxes.to(DEV), scores.to(DEV), 0.5, offset=1)[1].cpu(), R.nms(plus1, scores, 0.5))
    k2 = ops.nms(boxes.to(DEV), scores.to(DEV), 0.5, score_threshold=0.3, max_num=20)[1].cpu()
    sel = (scores > 0.3).nonzero().squeeze(1)
    assert torch.equal(k2, sel[R.nms(boxes[sel], scores[sel], 0.5)][:20])
    ids = torch.randint(0, 5, (n,), generator=g)
    d, k = ops.batched_nms(boxes.to(DEV), scores.to(DEV), ids.to(DEV), dict(type='nms', iou_threshold=0.4))
    assert sorted(k.cpu().tolist()) == sorted(R.batched_nms(boxes, scores, ids, 0.4).tolist())
    assert torch.all(d[:-1, 4] >= d[1:, 4])
    # above the single-launch limit the class-aware pass is split per class: same keep set as class-wise NMS
    big = 9000
    cb = torch.rand(5 * big, 2, generator=g) * 4000
    whb = torch.rand(5 * big, 2, generator=g) * 30 + 4
    bb = torch.cat([cb - whb / 2, cb + whb / 2], 1).to(DEV)
    sb = torch.rand(5 * big, generator=g).to(DEV)
    ib = torch.arange(5).repeat_interleave(big).to(DEV)
    assert bb.shape[0] > ops.NMS_MAX_CANDIDATES
    _, kb = ops.batched_nms(bb, sb, ib, dict(type='nms', iou_threshold=0.5))
    want = torch.cat([(ib == cc).nonzero().squeeze(1)[ops.nms(bb[ib == cc], sb[ib == cc], 0.5)[1]] for cc in range(5)])
    assert sorted(kb.tolist()) == sorted(want.tolist())


def test_rotated_signatures():
    from point_teacher_amd import ops
    g = torch.Generator().manual_seed(2)
    n = 60
    d = torch.cat([torch.rand(n, 2, generator=g) * 60, torch.rand(n, 2, generator=g) * 20 + 3,
                   (torch.rand(n, 1, generator=g) - 0.5) * math.pi], 1)
    s = torch.rand(n, generator=g)
    lab = torch.randint(0, 3, (n,), generator=g)
    out, keep = ops.nms_rotated(d.to(DEV), s.to(DEV), 0.1)
    assert torch.equal(keep.cpu(), R.nms_rotated(d, s, 0.1)) and out.shape == (keep.numel(), 6)
    _, kl = ops.nms_rotated(d.to(DEV), s.to(DEV), 0.1, lab.to(DEV))
    want = torch.cat([(lab == c).nonzero().squeeze(1)[R.nms_rotated(d[lab == c], s[lab == c], 0.1)] for c in range(3)])
    assert sorted(kl.cpu().tolist()) == sorted(want.tolist())
    iou = ops.box_iou_rotated(d[:20].to(DEV), d[20:45].to(DEV))
    close(iou, R.box_iou_rotated(d[:20], d[20:45]), rtol=1e-3, atol=1e-5)
    iof = ops.box_iou_rotated(d[:20].to(DEV), d[20:40].to(DEV), mode='iof', aligned=True)
    ia = R.box_iou_rotated(d[:20], d[20:40], aligned=True)
    a1, a2 = d[:20, 2] * d[:20, 3], d[20:40, 2] * d[20:40, 3]
    close(iof, ia * (a1 + a2) / (1 + ia) / a1, rtol=1e-3, atol=1e-5)
    b1 = d[:16].reshape(2, 8, 5).to(DEV).requires_grad_(True)
    v = ops.diff_iou_rotated_2d(b1, d[16:32].reshape(2, 8, 5).to(DEV))
    close(v.reshape(-1), R.box_iou_rotated(d[:16], d[16:32], aligned=True), rtol=1e-3, atol=1e-5)
    v.sum().backward()
    assert torch.isfinite(b1.grad).all()


def test_roi_modules():
    from point_teacher_amd import ops
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(2, 16, 20, 20, generator=g)
    rois = torch.tensor([[0, 8., 8., 40., 56.], [1, 20.3, 11.1, 23.9, 14.2], [1, 1., 1., 150., 150.]])
    layer = ops.RoIAlign(output_size=7, spatial_scale=1 / 8)
    close(layer(feat.to(DEV), rois.to(DEV)), R.roi_align(feat, rois, 7, 0.125), rtol=1e-4, atol=1e-5)
    close(ops.roi_align(feat.to(DEV), rois.to(DEV), (7, 7), 0.125), R.roi_align(feat, rois, 7, 0.125), rtol=1e-4, atol=1e-5)
    rr = torch.tensor([[0, 40., 50., 30., 12., 0.4], [1, 80., 80., 60., 20., -1.0]])
    lay = ops.RoIAlignRotated(out_size=7, sample_num=2, spatial_scale=1 / 8, clockwise=True)      # the OBB config's keywords
    close(lay(feat.to(DEV), rr.to(DEV)), R.roi_align_rotated(feat, rr, 7, 0.125, 2, True, True), rtol=1e-4, atol=1e-5)
    x = torch.randn(1, 4, 9, 9, generator=g)
    w = torch.randn(6, 4, 3, 3, generator=g)
    off = torch.zeros(1, 18, 9, 9)
    y = ops.deform_conv2d(x.to(DEV), off.to(DEV), w.to(DEV), 1, 1)
    close(y, torch.nn.functional.conv2d(x, w, None, 1, 1), rtol=1e-4, atol=1e-5)                   # zero offsets = a convolution
    m = ops.ModulatedDeformConv2d(4, 6, 3, padding=1).to(DEV)
    ym = m(x.to(DEV), off.to(DEV), torch.ones(1, 9, 9, 9, device=DEV))
    close(ym, torch.nn.functional.conv2d(x, m.weight.detach().cpu(), m.bias.detach().cpu(), 1, 1), rtol=1e-4, atol=1e-5)
