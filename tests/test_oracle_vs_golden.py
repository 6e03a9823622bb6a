"""Pin the CPU oracle (oracle/ref_ops.py) to golden vectors captured from the reference
(oracle/gen_golden.py).  Index results bit-exact; float results to fp32 round-off."""
import numpy as np
import pytest
import torch

from oracle import ref_ops as R
from conftest import load_golden

TOL = dict(rtol=1e-5, atol=1e-6)


def close(a, b, **kw):
    kw = {**TOL, **kw}
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    torch.testing.assert_close(a, b, equal_nan=True, **kw)


def test_costs():
    g = load_golden('costs')
    bx, gt, cls, lab = g.t('in_boxes_cxcywh'), g.t('in_gt_cxcywh'), g.t('in_cls'), g.t('in_labels')
    assert torch.equal(R.point_cost(bx, gt, 'L1', 1.0), g.t('out_point_l1'))
    close(R.point_cost(bx, gt, 'L2', 2.0), g['out_point_l2'])
    close(R.focal_loss_cost(cls, lab), g['out_focal'])
    assert torch.equal(R.insider_cost(bx, gt), g.t('out_insider'))


@pytest.mark.parametrize('name', ['g0', 'g1', 'g7', 'g40', 'clustered', 'full'])
def test_assigners(name):
    g = load_golden('assign_' + name)
    pts, cls, reg = g.t('in_points'), g.t('in_cls'), g.t('in_reg')
    gtb, lab = g.t('in_gt_bboxes'), g.t('in_gt_labels')
    gcx = R.bbox_xyxy_to_cxcywh(gtb) if gtb.shape[0] else gtb
    gi, gl = R.topk_assign(pts, gcx, lab, 1)
    assert torch.equal(gi, g.t('out_a11_gt_inds'))
    assert torch.equal(gi, g.t('out_a11pts_gt_inds'))
    ref_l = g.t('out_a11_labels')
    assert torch.equal(gl, ref_l)
    gi, gl = R.topk_assign(pts, gcx, lab, 3)
    assert torch.equal(gi, g.t('out_a33_gt_inds')) and torch.equal(gl, g.t('out_a33_labels'))
    dcx = R.bbox_xyxy_to_cxcywh(R.distance2bbox(pts, reg))
    gi, gl = R.fuse_topk_assign(dcx, pts, cls, gcx[:, :2] if gtb.shape[0] else gcx, lab, 5, 3)
    assert torch.equal(gi, g.t('out_fuse_gt_inds')) and torch.equal(gl, g.t('out_fuse_labels'))


@pytest.mark.parametrize('name', ['small', 'mid'])
def test_pseudo_and_targets(name):
    g = load_golden('pseudo_' + name)
    pts, cls, reg = g.t('in_points'), g.t('in_cls'), g.t('in_reg')
    gp, gl, gb = g.t('in_gt_points'), g.t('in_gt_labels'), g.t('in_gt_bboxes')
    pb, pp, ps, miou, valid = R.generate_pseudo_single(pts, cls, reg, gp, gl, gb, 0.0)
    close(pb, g['out_pseudo_bboxes'], rtol=1e-5, atol=1e-4)
    close(pp, g['out_pseudo_points'], rtol=1e-5, atol=1e-4)
    close(miou, g['out_mean_iou'], rtol=1e-5)
    assert np.array_equal(valid.numpy(), g['out_valid_inds'])
    pbg, ppg = g.t('out_pseudo_bboxes'), g.t('out_pseudo_points')
    lr, bt, lab, w = R.get_target_pseudo_single(pts, cls, gp, gl, pbg, gl)
    assert torch.equal(lr, g.t('out_labels_reg')) and torch.equal(lab, g.t('out_labels'))
    close(bt, g['out_bbox_targets'])
    close(w, g['out_weights'])
    sl, sbt = R.get_target_single(pts, gb)
    assert torch.equal(sl, g.t('out_syn_labels'))
    close(sbt, g['out_syn_bbox_targets'])
    close(R.centerness_target(bt[lr < 8]), g['out_ctr_target'])


def test_diou_losses():
    g = load_golden('loss_diou')
    tgt, w, af = g.t('in_target'), g.t('in_weight'), float(g['avg_factor'])
    pred = g.t('in_pred').clone().requires_grad_(True)
    close(R.diou_loss_elem(pred, tgt), g['out_diou_none'])
    l = R.iou_loss_module(pred, tgt, w, af, 'diou')
    close(l, g['out_diou_loss'])
    close(torch.autograd.grad(l, pred)[0], g['out_diou_grad'], rtol=1e-4, atol=1e-6)
    for hyper, tag in ((0.1, '01'), (0.2, '02')):
        close(R.dn_diou_loss_elem(pred, tgt, hyper), g[f'out_dn{tag}_none'])
        l = R.iou_loss_module(pred, tgt, w, af, 'dn', hyper)
        close(l, g[f'out_dn{tag}_loss'])
        close(torch.autograd.grad(l, pred)[0], g[f'out_dn{tag}_grad'], rtol=1e-4, atol=1e-6)
    close(R.iou_loss_module(pred, tgt, torch.zeros_like(w), 1.0, 'dn', 0.1), g['out_dn_zero_w'])


def test_focal_and_bce():
    g = load_golden('loss_focal')
    x = g.t('in_logits').clone().requires_grad_(True)
    l = R.sigmoid_focal_loss(x, g.t('in_labels'), g.t('in_weight'), avg_factor=float(g['avg_factor']))
    close(l, g['out_loss'])
    close(torch.autograd.grad(l, x)[0], g['out_grad'], rtol=1e-4, atol=1e-7)
    close(R.sigmoid_focal_loss_elem(x, g.t('in_labels')), g['out_elem'])
    g = load_golden('loss_bce')
    x = g.t('in_logits').clone().requires_grad_(True)
    l = R.bce_logits_loss(x, g.t('in_target'), float(g['avg_factor']))
    close(l, g['out_loss'])
    close(torch.autograd.grad(l, x)[0], g['out_grad'])


def test_bbox_math():
    g = load_golden('bbox_math')
    a, b, b2 = g.t('in_a'), g.t('in_b'), g.t('in_b2')
    for mode in ('iou', 'iof', 'giou'):
        close(R.bbox_overlaps(a, b, mode), g[f'out_pair_{mode}'])
        close(R.bbox_overlaps(a, b2, mode, True), g[f'out_align_{mode}'])
    # known answers from the reference's own tests (test_box_overlap.py:87-105,
    # test_misc.py:51-76, test_coder.py:26-75)
    k1 = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]])
    k2 = torch.FloatTensor([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]])
    gi = R.bbox_overlaps(k1, k2, 'giou', True)
    close(gi, g['known_giou_aligned'])
    close(gi, torch.tensor([0.5000, -0.0500, -0.8214]), rtol=1e-3, atol=1e-4)
    point = torch.Tensor([[74., 61.], [-29., 106.], [138., 61.], [29., 170.]])
    dist = torch.Tensor([[0., 0, 1., 1.], [1., 2., 10., 6.], [22., -29., 138., 61.], [54., -29., 170., 0.]])
    exp = torch.Tensor([[74., 61., 75., 62.], [0., 104., 0., 112.], [100., 90., 100., 120.], [0., 120., 100., 120.]])
    close(R.distance2bbox(point, dist, (110, 200)), g['known_d2b'])
    close(R.distance2bbox(point, dist, (120, 100)), exp)
    rois = torch.Tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    deltas = torch.Tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
    exp = torch.Tensor([[0.0000, 0.0000, 1.0000, 1.0000], [0.1409, 0.1409, 2.8591, 2.8591],
                        [0.0000, 0.3161, 4.1945, 0.6839], [5.0000, 5.0000, 5.0000, 5.0000]])
    close(R.delta2bbox(rois, deltas, (32, 32)), g['known_delta2bbox'])
    close(R.delta2bbox(rois, deltas, (32, 32)), exp, rtol=1e-4, atol=1e-4)
    close(R.distance2bbox(g.t('in_pts'), g.t('in_d4')), g['out_d2b'])
    close(R.distance2bbox(g.t('in_pts'), g.t('in_d4'), (200, 160, 3)), g['out_d2b_clip'])
    close(R.delta2bbox(a, g.t('in_deltas'), (300, 300, 3)), g['out_delta_decode'], rtol=1e-5, atol=1e-4)
    close(R.bbox_xyxy_to_cxcywh(a), g['out_xyxy2cxcywh'])
    close(R.bbox_cxcywh_to_xyxy(R.bbox_xyxy_to_cxcywh(a)), g['out_cxcywh2xyxy'])
    close(R.bbox2roi([a[:5], a[:0], b[:3]]), g['out_bbox2roi'])
    close(R.obb2poly_le90(g.t('in_obb')), g['out_obb2poly'], rtol=1e-5, atol=1e-4)
    close(R.obb2xyxy(g.t('in_obb')), g['out_obb2xyxy'], rtol=1e-5, atol=1e-4)
    u = g.t('in_u')
    for pos in (0.0, 0.3, 1.0):
        close(R.random_point_in_box(a, pos, u[0], u[1]), g[f'out_rpq_{int(pos * 100)}'], rtol=1e-5, atol=1e-4)


CFGS = [([1.0], None, 0), ([1.0, 1.2, 1.3, 0.8, 0.7], None, 4), ([1.0, 1.2, 1.3, 0.8, 0.7], [0.1], 16),
        ([1.0, 1.3, 0.8], None, 0), ([1.0, 1.3, 0.7], [0.1], 4)]


def test_proposals():
    g = load_golden('proposals')
    hw = tuple(int(v) for v in g['img_hw'])
    boxes = [g.t('in_boxes0'), g.t('in_boxes1')]
    for ci, (ratios, shake, ms) in enumerate(CFGS):
        for i in range(2):
            p, v = R.fine_proposals(boxes[i], ratios, shake, ms, hw)
            close(p, g[f'out_c{ci}_prop{i}'], rtol=1e-6, atol=1e-4)
            assert np.array_equal(v.numpy(), g[f'out_c{ci}_valid{i}'])
            U = p.shape[0] // boxes[i].shape[0]
            close(boxes[i][:, None].repeat(1, U, 1).reshape(-1, 4), g[f'out_c{ci}_ref{i}'])
    u = g.t('in_u')
    for i in range(2):
        pos, _ = R.fine_proposals(boxes[i], *CFGS[0][:3], hw)
        n, w = R.negative_proposals(u[i], pos, hw)
        close(n, g[f'out_neg{i}'])
        assert np.array_equal(w.numpy(), g[f'out_negw{i}'])


@pytest.mark.parametrize('name', ['u1x25_k1', 'u9x45_k3', 'full_k1'])
def test_mil_bags(name):
    g = load_golden('mil_' + name)
    cls = g.t('in_cls').clone().requires_grad_(True)
    ins = g.t('in_ins').clone().requires_grad_(True)
    neg = g.t('in_neg_cls').clone().requires_grad_(True)
    valid, lab = g.t('in_valid'), g.t('in_labels')
    l = R.mil_bag_loss(cls, ins, valid, lab, neg, g.t('in_neg_w'))
    close(l, g['out_loss'], rtol=2e-5)
    gc, gi, gn = torch.autograd.grad(l, [cls, ins, neg])
    close(gc, g['out_grad_cls'], rtol=1e-4, atol=1e-7)
    close(gi, g['out_grad_ins'], rtol=1e-4, atol=1e-7)
    close(gn, g['out_grad_neg'], rtol=1e-4, atol=1e-7)
    m = R.mil_bag_select(cls, ins, valid, lab, g.t('in_bags'), g.t('in_pseudo'), (800, 800),
                         int(g['topk']), float(g['beta']))
    close(m, g['out_merged'], rtol=1e-5, atol=1e-3)


def test_strong_aug():
    g = load_golden('strong_aug')
    img = g.t('in_img')
    for i in range(img.shape[0]):
        pb = g.t(f'in_pseudo_bboxes{i}')
        pp = (pb[:, :2] + pb[:, 2:]) / 2
        out = R.strong_augmentation_single(img[i], g.t(f'in_gt_points{i}'), g.t(f'in_gt_labels{i}'), pp,
                                           g.t(f'in_gt_labels{i}'), pb, str(g['flips'][i]), float(g['scales'][i]))
        close(out[0], g['out_img'][i], atol=0)
        close(out[1], g[f'out_gt_points{i}'], rtol=1e-6, atol=1e-5)
        assert torch.equal(out[2], g.t(f'out_gt_labels{i}'))
        close(out[3], g[f'out_pseudo_points{i}'], rtol=1e-6, atol=1e-5)
        assert torch.equal(out[4], g.t(f'out_pseudo_labels{i}'))
        close(out[5], g[f'out_pseudo_bboxes{i}'], rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize('name', ['small', 'mid'])
def test_head_losses(name):
    g = load_golden('head_' + name)
    cls = g.t('in_cls').clone().requires_grad_(True)
    reg = g.t('in_reg').clone().requires_grad_(True)
    ctr = g.t('in_ctr').clone().requires_grad_(True)
    pts = g.t('in_points')
    B = cls.shape[0]
    gp = [g.t(f'in_gt_points{i}') for i in range(B)]
    gl = [g.t(f'in_gt_labels{i}') for i in range(B)]
    gb = [g.t(f'in_gt_bboxes{i}') for i in range(B)]
    fc, fr, _ = R._flatten_head(cls.detach(), reg.detach(), ctr.detach())
    mious = []
    for i in range(B):
        pb, pp, ps, miou, valid = R.generate_pseudo_single(pts, fc[i], fr[i], gp[i], gl[i], gb[i])
        close(pb, g[f'out_pseudo_bboxes{i}'], rtol=1e-5, atol=1e-4)
        close(pp, g[f'out_pseudo_points{i}'], rtol=1e-5, atol=1e-4)
        mious.append(miou)
    close(sum(mious) / B, g['out_mean_iou'], rtol=1e-5)
    pb2 = [g.t(f'in_pseudo_bboxes_jit{i}') for i in range(B)]
    pp2 = [R.bbox_xyxy_to_cxcywh(b)[:, :2] for b in pb2]
    lc, lb, lt = R.loss_pseudo(cls, reg, ctr, pts, gp, gl, pp2, gl, pb2)
    close(lc, g['out_loss_cls'], rtol=2e-5)
    close(lb, g['out_loss_bbox'], rtol=2e-5)
    close(lt, g['out_loss_ctr'], rtol=2e-5)
    gr = torch.autograd.grad(lc + lb + lt, [cls, reg, ctr])
    close(gr[0], g['out_grad_cls'], rtol=1e-4, atol=1e-7)
    close(gr[1], g['out_grad_reg'], rtol=1e-4, atol=1e-7)
    close(gr[2], g['out_grad_ctr'], rtol=1e-4, atol=1e-7)
    sb, sc = R.loss_synthetic(reg, ctr, pts, gb)
    close(sb, g['out_syn_loss_bbox'], rtol=2e-5)
    close(sc, g['out_syn_loss_ctr'], rtol=2e-5)
    gs = torch.autograd.grad(sb + sc, [reg, ctr])
    close(gs[0], g['out_syn_grad_reg'], rtol=1e-4, atol=1e-7)
    close(gs[1], g['out_syn_grad_ctr'], rtol=1e-4, atol=1e-7)


def test_ema():
    g = load_golden('ema')
    close(R.ema_update(g.t('in_teacher'), g.t('in_student'), float(g['alpha'])), g['out_teacher'])


# ---- invariants for the "parity unpinned" mmcv ops ---------------------------------

def test_roi_align_invariants():
    torch.manual_seed(0)
    H = W = 12
    const = torch.full((1, 3, H, W), 2.5)
    rois = torch.tensor([[0, 8., 8., 40., 56.], [0, 20.3, 11.1, 23.9, 14.2], [0, 1., 1., 90., 90.]])
    out = R.roi_align(const, rois)
    inside = out[:2]
    close(inside, torch.full_like(inside, 2.5))
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    ramp = (2 * xs + 3 * ys)[None, None]
    r = torch.tensor([[0, 16., 24., 48., 72.]])
    o = R.roi_align(ramp, r)[0, 0]
    # bin centres in feature coordinates (aligned: -0.5 offset), value = ramp at centre
    bw, bh = (48 - 16) / 8 / 7, (72 - 24) / 8 / 7
    cx = 16 / 8 - 0.5 + (torch.arange(7) + 0.5) * bw
    cy = 24 / 8 - 0.5 + (torch.arange(7) + 0.5) * bh
    close(o, 2 * cx[None, :] + 3 * cy[:, None], rtol=1e-5, atol=1e-4)
    feat = torch.randn(2, 4, H, W, dtype=torch.float32, requires_grad=True)
    rr = torch.tensor([[1, 10., 12., 30., 44.], [0, 50., 50., 53., 52.]])
    (R.roi_align(feat, rr) ** 2).sum().backward()
    assert feat.grad.abs().sum() > 0 and feat.grad[0, :, :3].abs().sum() == 0


def test_rotated_iou_matches_axis_aligned():
    g = torch.Generator().manual_seed(3)
    c = torch.rand(40, 2, generator=g) * 50
    wh = torch.rand(40, 2, generator=g) * 20 + 1
    b = torch.cat([c, wh, torch.zeros(40, 1)], 1)
    hb = R.bbox_cxcywh_to_xyxy(b[:, :4])
    close(R.box_iou_rotated(b[:20], b[20:]), R.bbox_overlaps(hb[:20], hb[20:]), rtol=1e-4, atol=1e-5)
    # a square rotated by 45 degrees inside itself: octagon overlap
    s = torch.tensor([[0., 0., 2., 2., 0.]])
    r = torch.tensor([[0., 0., 2., 2., np.pi / 4]])
    inter = 8 * (np.sqrt(2) - 1)
    close(R.box_iou_rotated(s, r), torch.tensor([[inter / (8 - inter)]]), rtol=1e-5)
    keep = R.nms_rotated(torch.cat([s, r, torch.tensor([[10., 10., 2., 2., 0.3]])]), torch.tensor([0.9, 0.8, 0.7]), 0.05)
    assert keep.tolist() == [0, 2]


def test_nms_basic():
    boxes = torch.tensor([[0., 0., 10., 10.], [1., 1., 11., 11.], [20., 20., 30., 30.], [0., 0., 10., 10.]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95])
    assert R.nms(boxes, scores, 0.5).tolist() == [3, 2]
    assert sorted(R.batched_nms(boxes, scores, torch.tensor([0, 1, 0, 2]), 0.5).tolist()) == [0, 1, 2, 3]
