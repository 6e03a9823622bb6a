"""GPU parity tests: every kernel behind include/pt_hip.h against the CPU oracle
(oracle/ref_ops.py) on seeded inputs and against the golden fixtures captured from the
reference.  Index outputs bit-exact; fp32 outputs within 1e-3 relative (north_star's
tolerance; most are ~1e-6).  All calls go through the C ABI (point_teacher_amd.hip)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = 'cuda'
RT = 1e-3     # north_star: within 1e-3 relative on fp32


def F():
    from point_teacher_amd import functional
    return functional


def close(a, b, rtol=RT, atol=1e-5):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(np.asarray(b) if not isinstance(b, torch.Tensor) else b.detach().cpu()).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol, equal_nan=True)


def cu(x):
    return x.to(DEV)


def off1(n):
    return torch.tensor([0, n], dtype=torch.int32, device=DEV)


@pytest.mark.parametrize('name', ['g0', 'g1', 'g7', 'g40', 'clustered', 'full'])
def test_assigners_vs_golden(name):
    g = load_golden('assign_' + name)
    f = F()
    pts, cls, reg = g.t('in_points'), g.t('in_cls'), g.t('in_reg')
    gtb, lab = g.t('in_gt_bboxes'), g.t('in_gt_labels')
    G = gtb.shape[0]
    gcx = R.bbox_xyxy_to_cxcywh(gtb) if G else torch.zeros(0, 4)
    off = off1(G)
    for k, key in ((1, 'out_a11_gt_inds'), (3, 'out_a33_gt_inds')):
        gi = f.topk_assign(cu(pts), cu(gcx), off, 1, k)
        assert torch.equal(gi.cpu().long().reshape(-1), g.t(key)), key
    gi, cand = f.fuse_assign(cu(pts), cu(reg), cu(cls), cu(gcx[:, :2].contiguous()), cu(lab), off, 1, 5, 3)
    assert torch.equal(gi.cpu().long().reshape(-1), g.t('out_fuse_gt_inds'))
    # labels follow from gt_inds
    if G:
        lab_out = torch.where(gi.cpu().reshape(-1) > 0, lab[(gi.cpu().reshape(-1).long() - 1).clamp(min=0)],
                              torch.full((pts.shape[0],), -1, dtype=torch.long))
        assert torch.equal(lab_out, g.t('out_fuse_labels'))


def test_assign_batched_and_tie_rule():
    """Two images in one launch + exact distance ties (gt on a grid point / cell centre):
    the documented rule is lowest point index, identical to the oracle."""
    f = F()
    H = W = 40
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    pts = torch.stack([xs.reshape(-1) * 8 + 4, ys.reshape(-1) * 8 + 4], 1)
    g = torch.Generator().manual_seed(5)
    gts = [torch.round(torch.rand(57, 2, generator=g) * 300 * 2) / 2, torch.round(torch.rand(31, 2, generator=g) * 300)]
    labs = [torch.randint(0, 8, (57,), generator=g), torch.randint(0, 8, (31,), generator=g)]
    off, tot = f.make_offsets([57, 31], DEV)
    allg = torch.cat(gts)
    for k in (1, 3, 5):
        gi = f.topk_assign(cu(pts), cu(allg), off, 2, k).cpu().long()
        for b in range(2):
            exp, _ = R.topk_assign(pts, gts[b], labs[b], k)
            assert torch.equal(gi[b], exp), (k, b)
    cls = torch.randn(2, H * W, 8, generator=g)
    reg = torch.rand(2, H * W, 4, generator=g) * 20
    gi, _ = f.fuse_assign(cu(pts), cu(reg.reshape(-1, 4)), cu(cls.reshape(-1, 8)), cu(allg), cu(torch.cat(labs)), off, 2)
    for b in range(2):
        dcx = R.bbox_xyxy_to_cxcywh(R.distance2bbox(pts, reg[b]))
        exp, _ = R.fuse_topk_assign(dcx, pts, cls[b], gts[b], labs[b])
        assert torch.equal(gi.cpu().long()[b], exp), b


@pytest.mark.parametrize('name', ['small', 'mid'])
def test_pseudo_and_targets_vs_golden(name):
    g = load_golden('pseudo_' + name)
    f = F()
    pts, cls, reg = cu(g.t('in_points')), cu(g.t('in_cls')), cu(g.t('in_reg'))
    gp, gl, gb = cu(g.t('in_gt_points')), cu(g.t('in_gt_labels')), cu(g.t('in_gt_bboxes'))
    G = gp.shape[0]
    off = off1(G)
    gi, cand = f.fuse_assign(pts, reg, cls, gp, gl, off, 1)
    ps = f.pseudo_boxes(pts, reg, cls, gp, gl, gb, off, 1, gi, cand)
    close(ps['bboxes'], g['out_pseudo_bboxes'], atol=1e-3)
    close(ps['points'], g['out_pseudo_points'], atol=1e-3)
    nz = ps['nassigned'] > 0
    close(ps['iou'][nz].mean(), g['out_mean_iou'])
    assert np.array_equal(torch.nonzero(nz).reshape(-1).cpu().numpy(), g['out_valid_inds'])
    pb = cu(g.t('out_pseudo_bboxes'))
    gi1 = f.topk_assign(pts, gp, off, 1, 1)
    labels, _, _ = f.fcos_targets(pts, gi1, None, gl, off, 1, 8, want_ctr=False)
    assert torch.equal(labels.cpu().long(), g.t('out_labels'))
    gi3 = f.topk_assign(pts, R.bbox_xyxy_to_cxcywh(pb.cpu()).to(DEV), off, 1, 3)
    lr, bt, ctr = f.fcos_targets(pts, gi3, pb, gl, off, 1, 8)
    assert torch.equal(lr.cpu().long(), g.t('out_labels_reg'))
    close(bt, g['out_bbox_targets'], atol=1e-4)
    close(ctr[lr < 8], g['out_ctr_target'], atol=1e-5)
    gis = f.topk_assign(pts, R.bbox_xyxy_to_cxcywh(gb.cpu()).to(DEV), off, 1, 3)
    sl, sbt, _ = f.fcos_targets(pts, gis, gb, None, off, 1, 8)
    assert torch.equal(sl.cpu().long(), g.t('out_syn_labels'))
    close(sbt, g['out_syn_bbox_targets'], atol=1e-4)


def test_focal_loss():
    g = load_golden('loss_focal')
    f = F()
    x = cu(g.t('in_logits')).requires_grad_(True)
    l = f.sigmoid_focal_loss_sum(x, cu(g.t('in_labels')), cu(g.t('in_weight'))) / float(g['avg_factor'])
    close(l, g['out_loss'])
    l.backward()
    close(x.grad, g['out_grad'], atol=1e-7)
    close(f.sigmoid_focal_loss_elem(x, cu(g.t('in_labels'))), g['out_elem'], atol=1e-7)
    # full size (B*P = 20000 points, 8 classes) against the oracle
    gen = torch.Generator().manual_seed(1)
    X = torch.randn(20000, 8, generator=gen) * 3
    L = torch.randint(0, 9, (20000,), generator=gen)
    xr = X.clone().requires_grad_(True)
    lo = R.sigmoid_focal_loss(xr, L, None, avg_factor=311.0)
    lo.backward()
    xg = cu(X).requires_grad_(True)
    lg = f.sigmoid_focal_loss_sum(xg, cu(L)) / 311.0
    lg.backward()
    close(lg, lo)
    close(xg.grad, xr.grad, atol=1e-8)


def test_diou_and_dn_diou():
    g = load_golden('loss_diou')
    f = F()
    tgt, w, af = cu(g.t('in_target')), cu(g.t('in_weight')), float(g['avg_factor'])
    pred = cu(g.t('in_pred')).requires_grad_(True)
    d = f.diou_loss_elem(pred, tgt)
    close(d, g['out_diou_none'], atol=1e-6)
    l = (d * w).sum() / af
    close(l, g['out_diou_loss'])
    close(torch.autograd.grad(l, pred)[0], g['out_diou_grad'], atol=1e-6)
    for hyper, tag in ((0.1, '01'), (0.2, '02')):
        d, dn = f.dn_diou_pair(pred, tgt, hyper)
        le = (d.mean() + dn) / 2                     # the DN quirk, iou_loss.py:412,463-464
        close(le, g[f'out_dn{tag}_none'], atol=1e-6)
        l = (le * w).sum() / af
        close(l, g[f'out_dn{tag}_loss'])
        close(torch.autograd.grad(l, pred)[0], g[f'out_dn{tag}_grad'], atol=1e-6)


def test_bbox_math():
    g = load_golden('bbox_math')
    f = F()
    a, b, b2 = cu(g.t('in_a')), cu(g.t('in_b')), cu(g.t('in_b2'))
    for mode in ('iou', 'iof', 'giou'):
        close(f.bbox_overlaps(a, b, mode), g[f'out_pair_{mode}'], atol=1e-6)
        close(f.bbox_overlaps(a, b2, mode, True), g[f'out_align_{mode}'], atol=1e-6)
    k1 = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]])
    k2 = torch.FloatTensor([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]])
    close(f.bbox_overlaps(cu(k1), cu(k2), 'giou', True), torch.tensor([0.5000, -0.0500, -0.8214]), atol=1e-4)
    assert tuple(f.bbox_overlaps(cu(torch.zeros(0, 4)), cu(k1)).shape) == (0, 3)
    close(f.delta2bbox(a, cu(g.t('in_deltas')), (300, 300, 3)), g['out_delta_decode'], atol=1e-3)
    rois = torch.Tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    deltas = torch.Tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
    close(f.delta2bbox(cu(rois), cu(deltas), (32, 32)), g['known_delta2bbox'], atol=1e-4)
    # gradient of the decode against autograd through the oracle
    dl = g.t('in_deltas').clone().requires_grad_(True)
    wgt = torch.randn(64, 4, generator=torch.Generator().manual_seed(0))
    (R.delta2bbox(g.t('in_a'), dl, (300, 300, 3)) * wgt).sum().backward()
    dg = cu(g.t('in_deltas')).requires_grad_(True)
    (f.delta2bbox(a, dg, (300, 300, 3)) * cu(wgt)).sum().backward()
    close(dg.grad, dl.grad, atol=1e-4)


CFGS = [([1.0], None, 0), ([1.0, 1.2, 1.3, 0.8, 0.7], None, 4), ([1.0, 1.2, 1.3, 0.8, 0.7], [0.1], 16),
        ([1.0, 1.3, 0.8], None, 0), ([1.0, 1.3, 0.7], [0.1], 4)]


def test_proposals():
    g = load_golden('proposals')
    f = F()
    hw = tuple(int(v) for v in g['img_hw'])
    boxes = torch.cat([g.t('in_boxes0'), g.t('in_boxes1')])
    n0 = g.t('in_boxes0').shape[0]
    for ci, (ratios, shake, ms) in enumerate(CFGS):
        p, v = f.fine_proposals(cu(boxes), ratios, shake, ms, hw)
        U = p.shape[0] // boxes.shape[0]
        exp_p = np.concatenate([g[f'out_c{ci}_prop0'], g[f'out_c{ci}_prop1']])
        exp_v = np.concatenate([g[f'out_c{ci}_valid0'], g[f'out_c{ci}_valid1']]).reshape(-1)
        close(p, exp_p, rtol=1e-6, atol=1e-4)
        assert np.array_equal(v.cpu().numpy().astype(bool), exp_v), ci
        assert U * n0 == g[f'out_c{ci}_prop0'].shape[0]
    pos, _ = f.fine_proposals(cu(boxes), *CFGS[0], hw)
    poff, _ = f.make_offsets([n0, boxes.shape[0] - n0], DEV)
    neg, ok = f.negative_proposals(cu(g.t('in_u')), pos, poff, hw)
    close(neg, np.concatenate([g['out_neg0'], g['out_neg1']]), atol=1e-5)
    assert np.array_equal(ok.cpu().numpy().astype(bool), np.concatenate([g['out_negw0'], g['out_negw1']]))


@pytest.mark.parametrize('name', ['u1x25_k1', 'u9x45_k3', 'full_k1'])
def test_mil_bags(name):
    g = load_golden('mil_' + name)
    f = F()
    N, U1, U2, C = g['in_cls'].shape
    cls = cu(g.t('in_cls')).requires_grad_(True)
    ins = cu(g.t('in_ins')).requires_grad_(True)
    neg = cu(g.t('in_neg_cls')).requires_grad_(True)
    valid = cu(g.t('in_valid')).reshape(-1)
    lab = cu(g.t('in_labels'))
    lab_b = lab[:, None].repeat(1, U1).reshape(-1)
    s, nv = f.mil_bag_loss_sum(cls.reshape(N * U1, U2, C), ins.reshape(N * U1, U2, C), valid, lab_b)
    ns = nv.clamp(min=1.0)
    l = s / ns + f.mil_neg_loss_sum(neg, cu(g.t('in_neg_w'))) / ns
    close(l, g['out_loss'])
    gc, gi, gn = torch.autograd.grad(l, [cls, ins, neg])
    close(gc, g['out_grad_cls'], atol=1e-7)
    close(gi, g['out_grad_ins'], atol=1e-7)
    close(gn, g['out_grad_neg'], atol=1e-7)
    m = f.mil_bag_select(cls, ins, valid, lab, cu(g.t('in_bags')), cu(g.t('in_pseudo')), U1, U2, int(g['topk']),
                         float(g['beta']), (800, 800))
    close(m, g['out_merged'], atol=1e-2)


def _rois(gen, K, B, size):
    c = torch.rand(K, 2, generator=gen) * size
    wh = torch.exp(torch.randn(K, 2, generator=gen) * 0.8 + np.log(14.))
    bi = torch.randint(0, B, (K, 1), generator=gen).float()
    return torch.cat([bi, c - wh / 2, c + wh / 2], 1)


@pytest.mark.parametrize('channels_last', [False, True])
def test_roi_align_vs_oracle(channels_last):
    f = F()
    gen = torch.Generator().manual_seed(7)
    B, C, H, W = 2, 64, 25, 25
    feat = torch.randn(B, C, H, W, generator=gen)
    rois = _rois(gen, 96, B, 200)
    rois[0, 1:] = torch.tensor([-30., -30., -20., -20.])      # entirely outside
    rois[1, 1:] = torch.tensor([150., 150., 260., 260.])      # crosses the border, big (grid > 1)
    rois[2, 1:] = torch.tensor([50., 50., 50., 50.])          # zero size
    fr = feat.clone().requires_grad_(True)
    out_ref = R.roi_align(fr, rois, 7, 0.125)
    wgt = torch.randn(out_ref.shape, generator=gen)
    (out_ref * wgt).sum().backward()
    fg = cu(feat)
    if channels_last:
        fg = fg.contiguous(memory_format=torch.channels_last)
    fg.requires_grad_(True)
    out = f.roi_align(fg, cu(rois), 7, 0.125)
    close(out, out_ref, atol=1e-5)
    (out * cu(wgt)).sum().backward()
    close(fg.grad, fr.grad, atol=1e-4)


def test_roi_align_full_size_properties():
    """BASELINE size (K = 5000 RoIs, [2,256,100,100]): constant map -> constant; linearity."""
    f = F()
    gen = torch.Generator().manual_seed(9)
    B, C, H, W = 2, 256, 100, 100
    rois = cu(_rois(gen, 5000, B, 800))
    rois[:, 1:3].clamp_(min=8)
    rois[:, 3:5].clamp_(max=790)
    rois[:, 3:5] = torch.maximum(rois[:, 3:5], rois[:, 1:3] + 1)
    const = torch.full((B, C, H, W), 1.75, device=DEV).contiguous(memory_format=torch.channels_last)
    out = f.roi_align(const, rois, 7, 0.125)
    assert out.shape == (5000, 256, 7, 7)
    close(out, torch.full_like(out, 1.75), atol=1e-5)
    a = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    b = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    lhs = f.roi_align(2 * a + 3 * b, rois, 7, 0.125)
    rhs = 2 * f.roi_align(a, rois, 7, 0.125) + 3 * f.roi_align(b, rois, 7, 0.125)
    close(lhs, rhs, atol=1e-4)
    # both layouts agree
    close(f.roi_align(a.contiguous(), rois, 7, 0.125), f.roi_align(a, rois, 7, 0.125), atol=1e-5)


def test_ema_sgd_norm():
    g = load_golden('ema')
    f = F()
    t = cu(g.t('in_teacher')).clone()
    f.ema_update_(t, cu(g.t('in_student')), float(g['alpha']))
    close(t, g['out_teacher'], rtol=1e-6, atol=1e-7)
    gen = torch.Generator().manual_seed(3)
    n = 1_000_003
    p, gr, m = (torch.randn(n, generator=gen) for _ in range(3))
    gr *= 0.05
    split = 900_000
    sq = f.grad_sqnorm(cu(gr))
    close(sq, (gr.double() ** 2).sum().float().reshape(1), rtol=1e-4)
    coef = min(1.0, 35.0 / (float(sq.sqrt()) + 1e-6))
    lr = torch.tensor([0.005], device=DEV)
    pg, mg = cu(p).clone(), cu(m).clone()
    f.sgd_step_(pg, cu(gr), mg, split, lr, 0.9, 1e-4, 2.0, 0.0, sq, 35.0, False)
    pw, mw = R.sgd_momentum_step(p[:split], gr[:split], m[:split], 0.005, 0.9, 1e-4, coef)
    pb, mb = R.sgd_momentum_step(p[split:], gr[split:], m[split:], 0.01, 0.9, 0.0, coef)
    close(pg, torch.cat([pw, pb]), rtol=1e-5, atol=1e-6)
    close(mg, torch.cat([mw, mb]), rtol=1e-5, atol=1e-6)
    # pt_sgd_step_groups: the parameter groups of a full paramwise_cfg (configs/baselines/aitodv2_yolof_r50_1x.py:70-71:
    # backbone lr x 1/3, norm decay x 0, plus a bias group) - four segments, the last one ending off a multiple of 4
    import ctypes
    from point_teacher_amd import hip
    ends, lr_m, wd_m = [400_000, 700_000, 900_000, n], [1.0, 1.0 / 3, 1.0, 2.0], [1.0, 1.0, 0.0, 0.0]
    pg, mg = cu(p).clone(), cu(m).clone()
    tables = ((ctypes.c_int64 * 4)(*ends), hip.host_floats(lr_m), hip.host_floats(wd_m), 4)
    f.sgd_step_groups_(pg, cu(gr), mg, tables, lr, 0.9, 1e-4, sq, 35.0, False)
    exp_p, exp_m, lo = [], [], 0
    for hi, a, b in zip(ends, lr_m, wd_m):
        q, mm = R.sgd_momentum_step(p[lo:hi], gr[lo:hi], m[lo:hi], 0.005 * a, 0.9, 1e-4 * b, coef)
        exp_p.append(q); exp_m.append(mm); lo = hi
    close(pg, torch.cat(exp_p), rtol=1e-5, atol=1e-6)
    close(mg, torch.cat(exp_m), rtol=1e-5, atol=1e-6)
    # first step: the momentum buffer is created from the (clipped, decayed) gradient
    pg, mg = cu(p).clone(), cu(m).clone()
    f.sgd_step_groups_(pg, cu(gr), mg, tables, lr, 0.9, 1e-4, sq, 35.0, True)
    q, mm = R.sgd_momentum_step(p[:400_000], gr[:400_000], torch.zeros(400_000), 0.005, 0.9, 1e-4, coef)
    close(pg[:400_000], q, rtol=1e-5, atol=1e-6)
    close(mg[:400_000], mm, rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError, match='PT_MAX_PARAM_GROUPS'):
        hip.call('pt_sgd_step_groups', pg, cu(gr), mg, n, (ctypes.c_int64 * 9)(*range(0, 36, 4)), hip.host_floats([1.0] * 9),
                 hip.host_floats([1.0] * 9), 9, lr, 0.9, 1e-4, sq, 35.0, 0)


def test_nms_and_rotated():
    f = F()
    gen = torch.Generator().manual_seed(11)
    N = 3000
    c = torch.rand(N, 2, generator=gen) * 400
    wh = torch.rand(N, 2, generator=gen) * 30 + 4
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1)
    scores = torch.rand(N, generator=gen)
    ids = torch.randint(0, 8, (N,), generator=gen)
    dets, keep = f.nms(cu(boxes), cu(scores), 0.5)
    assert torch.equal(keep.cpu(), R.nms(boxes, scores, 0.5))
    close(dets[:, 4], scores[keep.cpu()], rtol=0, atol=0)
    _, keep = f.batched_nms(cu(boxes), cu(scores), cu(ids), dict(type='nms', iou_threshold=0.5))
    exp = torch.cat([torch.nonzero(ids == k).reshape(-1)[R.nms(boxes[ids == k], scores[ids == k], 0.5)] for k in range(8)])
    assert sorted(keep.cpu().tolist()) == sorted(exp.tolist())
    M = 300
    rb = torch.cat([torch.rand(M, 2, generator=gen) * 200, torch.rand(M, 2, generator=gen) * 40 + 3,
                    (torch.rand(M, 1, generator=gen) - 0.5) * np.pi], 1)
    iou = f.box_iou_rotated(cu(rb[:60]), cu(rb[60:140]))
    close(iou, R.box_iou_rotated(rb[:60], rb[60:140]), atol=2e-4)
    rs = torch.rand(M, generator=gen)
    _, keep = f.nms_rotated(cu(rb), cu(rs), 0.05)
    assert torch.equal(keep.cpu(), R.nms_rotated(rb, rs, 0.05))


def test_nms_large_candidate_sets():
    """nms_pre x classes candidates of the shipped test_cfgs (up to 3000 x 8) exceed the two-word scan:
    N = 20000 goes through the eight-word scan and still equals the greedy oracle; above 32768 the
    entry point refuses loudly."""
    f = F()
    gen = torch.Generator().manual_seed(12)
    N = 20000
    c = torch.rand(N, 2, generator=gen) * 1500
    wh = torch.rand(N, 2, generator=gen) * 30 + 4
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1)
    scores = torch.rand(N, generator=gen)
    _, keep = f.nms(cu(boxes), cu(scores), 0.5)
    assert torch.equal(keep.cpu(), R.nms(boxes, scores, 0.5))
    with pytest.raises(RuntimeError, match='above 32768'):
        f.nms(cu(torch.zeros(32769, 4)), cu(torch.zeros(32769)), 0.5)


def test_fill_quads():
    f = F()
    gen = torch.Generator().manual_seed(13)
    obb = torch.cat([torch.rand(40, 2, generator=gen) * 180 + 10, torch.rand(40, 2, generator=gen) * 40 + 2,
                     (torch.rand(40, 1, generator=gen) - 0.5) * np.pi], 1)
    quads = R.obb2poly_le90(obb)
    alive = torch.rand(40, generator=gen) > 0.2
    img = torch.zeros(3, 200, 200, device=DEV)
    f.fill_quads_(img, cu(quads), cu(alive), 255.0)
    exp = R.fill_convex_quads(quads[alive].numpy().astype(np.int32), 200, 200)
    assert np.array_equal((img[0] == 255).cpu().numpy(), exp.astype(bool))
    assert torch.equal(img[0], img[2])


def test_roi_align_bag_fast_path():
    """Bags of jittered tiny boxes (group = 25) take the register-resident small-footprint kernels;
    bags with a large member or spanning two images fall back to the generic kernel inside the same
    call.  Both must equal the oracle (forward and backward)."""
    f = F()
    gen = torch.Generator().manual_seed(21)
    B, C, H, W = 2, 64, 40, 40
    feat = torch.randn(B, C, H, W, generator=gen)
    n_gt, U = 12, 25
    c = torch.rand(n_gt, 2, generator=gen) * 280 + 20
    wh = torch.exp(torch.randn(n_gt, 2, generator=gen) * 0.5 + np.log(11.))
    wh[3] = torch.tensor([150., 90.])                      # one big object -> generic path for its bag
    c[5] = torch.tensor([1.0, 318.0])                      # on the border -> clamped taps
    base = torch.cat([c - wh / 2, c + wh / 2], 1)
    props, _ = R.fine_proposals(base, [1.0, 1.2, 1.3, 0.8, 0.7], None, 4, (320, 320))
    bi = (torch.arange(n_gt).repeat_interleave(U) % 2).float()[:, None]
    bi[7 * U + 3] = 1 - bi[7 * U + 3]                      # a bag that spans two images -> fallback
    rois = torch.cat([bi, props], 1)
    fr = feat.clone().requires_grad_(True)
    out_ref = R.roi_align(fr, rois, 7, 0.125)
    wgt = torch.randn(out_ref.shape, generator=gen)
    (out_ref * wgt).sum().backward()
    fg = cu(feat).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = f.roi_align(fg, cu(rois), 7, 0.125, 0, True, U)
    close(out, out_ref, atol=1e-5)
    (out * cu(wgt)).sum().backward()
    close(fg.grad, fr.grad, atol=1e-4)
    # group hint must not change the result
    out1 = f.roi_align(fg.detach(), cu(rois), 7, 0.125, 0, True, 1)
    close(out1, out, rtol=1e-5, atol=1e-5)


def test_roi_align_backward_run_union_path():
    """Enough bags (K / 5 >= 512) for runs of 5: a run whose members span 6 ... 48 feature pixels takes the backward's path R
    (roi_align.hip: the members' blocks and weights together in LDS, one atomic per pixel of the footprints' UNION, 64-channel
    slices); tiny bags stay on the register path, members beyond 48 pixels and a bag spanning two images fall back member by
    member - all inside one launch.  Checked against the oracle's autograd on a channel subset (the op is channel-wise) and, on
    every channel, against the same op with group = 1 (per-RoI paths, themselves checked against the oracle above)."""
    f = F()
    gen = torch.Generator().manual_seed(77)
    B, C, H, W = 2, 128, 60, 60
    n, U = 108, 25
    feat = torch.randn(B, C, H, W, generator=gen)
    c = torch.rand(n, 2, generator=gen) * 400 + 40
    wh = torch.exp(torch.randn(n, 2, generator=gen) * 0.5 + np.log(11.))
    wh[:30] = torch.rand(30, 2, generator=gen) * 290 + 60            # 7 ... 44 feature pixels: path R
    wh[30] = torch.tensor([450., 120.])                              # one axis beyond 48 pixels: its run falls back
    wh[31] = torch.tensor([20., 300.])
    c[32] = torch.tensor([3.0, 470.0]); wh[32] = torch.tensor([150., 180.])   # clamped at two borders
    base = torch.cat([c - wh / 2, c + wh / 2], 1)
    props, _ = R.fine_proposals(base, [1.0, 1.2, 1.3, 0.8, 0.7], None, 4, (480, 480))
    bi = (torch.arange(n).repeat_interleave(U) % 2).float()[:, None]
    bi[5 * U + 7] = 1 - bi[5 * U + 7]                                # a large bag with one member in the other image
    rois = torch.cat([bi, props], 1)
    K = rois.shape[0]
    assert K // 5 >= 512
    wgt = torch.randn(K, C, 7, 7, generator=gen)
    fg = cu(feat).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = f.roi_align(fg, cu(rois), 7, 0.125, 0, True, U)
    (out * cu(wgt)).sum().backward()
    g25 = fg.grad.clone()
    fg.grad = None
    out1 = f.roi_align(fg, cu(rois), 7, 0.125, 0, True, 1)
    (out1 * cu(wgt)).sum().backward()
    close(out, out1, rtol=1e-5, atol=1e-5)
    close(g25, fg.grad, rtol=1e-4, atol=5e-4)
    sub = [0, 63, 64, 127]                                            # both 64-channel slices
    fr = feat[:, sub].clone().requires_grad_(True)
    out_ref = R.roi_align(fr, rois, 7, 0.125)
    (out_ref * wgt[:, sub]).sum().backward()
    close(out[:, sub], out_ref, rtol=1e-4, atol=2e-5)
    close(g25[:, sub], fr.grad, rtol=1e-4, atol=5e-4)


@pytest.mark.parametrize('C,H,W,group,out_size,sr', [(320, 24, 24, 45, 7, 0),      # two channel blocks (256 + 64); runs of 15
                                                       (6, 24, 24, 25, 7, 0),        # 6*49 floats: not a multiple of 4 -> scalar tile copies
                                                       (16, 120, 120, 7, 7, 0),      # map large enough for a > 104-pixel RoI: direct path C
                                                       (16, 24, 24, 4, 7, 2),        # fixed sampling_ratio
                                                       (16, 24, 24, 3, 5, 0)])       # out_size != 7: the generic per-RoI kernels
def test_roi_align_every_path(C, H, W, group, out_size, sr):
    """One launch per direction serves three paths (register footprint / separable / direct taps, roi_align.hip) and
    any channel count and run length: all of them against the oracle, forward and backward, K not a multiple of the
    run length, RoIs outside the map, zero-size RoIs, RoIs of 1, 20 and (when the map allows) 115 feature pixels."""
    f = F()
    gen = torch.Generator().manual_seed(31 + C)
    B = 2
    size = 8 * H
    feat = torch.randn(B, C, H, W, generator=gen)
    n_bag = 5
    c = torch.rand(n_bag, 2, generator=gen) * (size - 60) + 30
    wh = torch.exp(torch.randn(n_bag, 2, generator=gen) * 0.5 + np.log(11.))
    wh[1] = torch.tensor([160., 90.])                              # separable path
    if size > 800:
        wh[2] = torch.tensor([930., 900.]); c[2] = torch.tensor([size / 2, size / 2])   # > 104 feature pixels: path C
    base = torch.cat([c - wh / 2, c + wh / 2], 1)
    jit = torch.randn(n_bag, group, 4, generator=gen) * 1.5
    props = (base[:, None] + jit).reshape(-1, 4)
    bi = torch.arange(n_bag).repeat_interleave(group).float()[:, None] % B
    extra = torch.tensor([[0, -90., -90., -50., -40.], [1, 33., 41., 33., 41.], [1, size - 5., size - 9., size + 40., size + 30.]])
    rois = torch.cat([torch.cat([bi, props], 1), extra])           # K = n_bag * group + 3: the last run is partial
    fr = feat.clone().requires_grad_(True)
    out_ref = R.roi_align(fr, rois, out_size, 0.125, sr)
    wgt = torch.randn(out_ref.shape, generator=gen)
    (out_ref * wgt).sum().backward()
    fg = cu(feat).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = f.roi_align(fg, cu(rois), out_size, 0.125, sr, True, group)
    close(out, out_ref, rtol=1e-4, atol=2e-5)
    (out * cu(wgt)).sum().backward()
    close(fg.grad, fr.grad, rtol=1e-4, atol=2e-4)
    for other in (1, 16):                                          # the group hint only regroups the work
        o2 = f.roi_align(fg.detach(), cu(rois), out_size, 0.125, sr, True, other)
        close(o2, out, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('channels_last', [False, True])
@pytest.mark.parametrize('relu,with_res', [(True, False), (True, True), (False, False)])
def test_frozen_bn_epilogue(channels_last, relu, with_res):
    """pt_affine_relu_* == eval-mode BatchNorm (+ identity) (+ ReLU) of torch, values and gradients."""
    f = F()
    gen = torch.Generator().manual_seed(31)
    N, C, H, W = 2, 64, 12, 20
    x = torch.randn(N, C, H, W, generator=gen)
    res = torch.randn(N, C, H, W, generator=gen)
    bn = torch.nn.BatchNorm2d(C).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=gen) + 0.5); bn.bias.copy_(torch.randn(C, generator=gen))
        bn.running_mean.copy_(torch.randn(C, generator=gen)); bn.running_var.copy_(torch.rand(C, generator=gen) + 0.2)
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    y = bn(xr) + (rr if with_res else 0)
    y = torch.relu(y) if relu else y
    wgt = torch.randn(y.shape, generator=gen)
    (y * wgt).sum().backward()
    mf = torch.channels_last if channels_last else torch.contiguous_format
    xg = cu(x).contiguous(memory_format=mf).requires_grad_(True)
    rg = cu(res).contiguous(memory_format=mf).requires_grad_(True)
    sc = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).detach()
    sh = (bn.bias - bn.running_mean * sc).detach()
    out = f.affine_relu_(xg * 1.0, cu(sc), cu(sh), rg if with_res else None, relu)
    close(out, y, atol=1e-5)
    (out * cu(wgt)).sum().backward()
    close(xg.grad, xr.grad, atol=1e-5)
    if with_res:
        close(rg.grad, rr.grad, atol=1e-6)


@pytest.mark.parametrize('relu,with_res', [(True, False), (True, True), (False, False)])
def test_frozen_bn_epilogue_bf16(relu, with_res):
    """pt_affine_relu_*_bf16 == eval-mode BatchNorm (+ identity) (+ ReLU) on bf16 channels_last activations (arithmetic
    in fp32, one rounding on store): within bf16 resolution of the fp32 result, gradients likewise."""
    from point_teacher_amd import nn_modules
    gen = torch.Generator().manual_seed(35)
    N, C, H, W = 2, 64, 12, 20
    x = torch.randn(N, C, H, W, generator=gen)
    res = torch.randn(N, C, H, W, generator=gen)
    bn = torch.nn.BatchNorm2d(C).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=gen) + 0.5); bn.bias.copy_(torch.randn(C, generator=gen))
        bn.running_mean.copy_(torch.randn(C, generator=gen)); bn.running_var.copy_(torch.rand(C, generator=gen) + 0.2)
    for p in bn.parameters():
        p.requires_grad = False
    xb, rb = x.bfloat16().float(), res.bfloat16().float()              # the values the kernel sees
    xr, rr = xb.clone().requires_grad_(True), rb.clone().requires_grad_(True)
    y = bn(xr) + (rr if with_res else 0)
    y = torch.relu(y) if relu else y
    wgt = torch.randn(y.shape, generator=gen).bfloat16().float()
    (y * wgt).sum().backward()
    bng = bn.to(DEV)
    xg = cu(x).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rg = cu(res).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = nn_modules.conv_bn(xg * 1.0, torch.nn.Identity(), bng, relu, residual=rg if with_res else None)
    assert out.dtype == torch.bfloat16
    close(out.float(), y, rtol=1e-2, atol=2e-2)
    (out.float() * cu(wgt)).sum().backward()
    close(xg.grad.float(), xr.grad, rtol=1e-2, atol=2e-2)
    if with_res:
        close(rg.grad.float(), rr.grad, rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize('C,HW', [(64, (40, 52)), (256, (20, 30)), (1024, (9, 11)), (2048, (5, 7))])
@pytest.mark.parametrize('relu,with_res', [(True, False), (True, True), (False, False)])
def test_trainable_bn_epilogue(C, HW, relu, with_res):
    """pt_affine_relu_fwd + pt_affine_relu_bwd_train == eval-mode BatchNorm with a TRAINING affine (+ identity)
    (+ ReLU) of torch: values, input / residual gradients and the weight / bias gradients (config 5)."""
    from point_teacher_amd import nn_modules
    gen = torch.Generator().manual_seed(33)
    N, (H, W) = 3, HW
    x = torch.randn(N, C, H, W, generator=gen)
    res = torch.randn(N, C, H, W, generator=gen)
    bn = torch.nn.BatchNorm2d(C).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=gen) + 0.5); bn.bias.copy_(torch.randn(C, generator=gen))
        bn.running_mean.copy_(torch.randn(C, generator=gen)); bn.running_var.copy_(torch.rand(C, generator=gen) + 0.2)
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    y = bn(xr) + (rr if with_res else 0)
    y = torch.relu(y) if relu else y
    wgt = torch.randn(y.shape, generator=gen)
    (y * wgt).sum().backward()
    import copy
    bng = copy.deepcopy(bn).to(DEV)
    bng.weight.grad = bng.bias.grad = None
    xg = cu(x).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rg = cu(res).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ident = torch.nn.Identity()
    out = nn_modules.conv_bn(xg, ident, bng, relu, residual=rg if with_res else None)
    assert out.data_ptr() != xg.data_ptr()                       # x is kept for the weight gradient
    close(out, y, atol=1e-5)
    (out * cu(wgt)).sum().backward()
    close(xg.grad, xr.grad, atol=1e-5)
    if with_res:
        close(rg.grad, rr.grad, atol=1e-6)
    scale = float(bn.weight.grad.abs().max())
    close(bng.weight.grad, bn.weight.grad, rtol=1e-3, atol=1e-4 * scale)
    close(bng.bias.grad, bn.bias.grad, rtol=1e-3, atol=1e-4 * float(bn.bias.grad.abs().max()))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 256, (100, 100), (50, 50)), (3, 64, (25, 25), (13, 13)), (2, 8, (13, 9), (7, 5)),
                                   (1, 16, (7, 7), (4, 4)), (2, 32, (10, 12), (10, 12)), (1, 8, (19, 10), (10, 5))])
def test_upsample_add(shape, dtype):
    """pt_upsample_add_* == a + F.interpolate(b, size=a.shape[2:], mode='nearest') of torch (the FPN / PSAGG top-down step),
    values bit for bit and both gradients, on the pyramid's size pairs (even, odd: 25 <- 13, 13 <- 7, 7 <- 4, equal sizes)."""
    f = F()
    N, C, (Ha, Wa), (Hb, Wb) = shape
    gen = torch.Generator().manual_seed(61)
    a = torch.randn(N, C, Ha, Wa, generator=gen).to(dtype)
    b = torch.randn(N, C, Hb, Wb, generator=gen).to(dtype)
    w = torch.randn(N, C, Ha, Wa, generator=gen).to(dtype)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = ar + torch.nn.functional.interpolate(br, size=(Ha, Wa), mode='nearest')
    (ref.float() * w.float()).sum().backward()
    ag = cu(a).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bg = cu(b).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = f.upsample_add(ag, bg)
    assert 'UpsampleAdd' in type(out.grad_fn).__name__
    assert torch.equal(out.detach().cpu(), ref.detach())
    (out.float() * cu(w).float()).sum().backward()
    assert torch.equal(ag.grad.cpu(), ar.grad)
    if dtype == torch.float32:
        close(bg.grad, br.grad, atol=1e-5)
    else:                                        # torch sums the (up to 4) contributions in bf16, the kernel in fp32 then rounds
        close(bg.grad.float(), br.grad.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize('C,HW', [(256, (20, 30)), (64, (9, 11)), (1024, (5, 7))])
def test_conv_bias_relu_epilogue(C, HW):
    """ConvModule (conv + bias + ReLU, the FCOS towers): the fused epilogue (pt_affine_relu_fwd with unit scale in place,
    pt_affine_relu_bwd_train with x == NULL) == torch conv2d(bias) -> relu: values, input / weight / bias gradients."""
    import copy
    from point_teacher_amd.nn_modules import ConvModule
    torch.manual_seed(35)
    H, W = HW
    m = ConvModule(16, C, 3, padding=1, bias=True)
    with torch.no_grad():
        m.conv.bias.copy_(torch.randn(C) * 0.3)
    x = torch.randn(2, 16, H, W)
    wgt = torch.randn(2, C, H, W)
    xr = x.clone().requires_grad_(True)
    y = m(xr)                                                                   # CPU: plain torch path
    (y * wgt).sum().backward()
    mg = copy.deepcopy(m).to(DEV)
    mg.zero_grad()
    xg = cu(x).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = mg(xg)
    assert out.grad_fn is not None and 'BiasReLU' in type(out.grad_fn).__name__, type(out.grad_fn).__name__
    close(out, y.detach(), atol=2e-5)
    (out * cu(wgt)).sum().backward()
    close(xg.grad, xr.grad, atol=2e-5)
    close(mg.conv.weight.grad, m.conv.weight.grad, rtol=1e-3, atol=1e-4 * float(m.conv.weight.grad.abs().max()))
    close(mg.conv.bias.grad, m.conv.bias.grad, rtol=1e-3, atol=1e-4 * float(m.conv.bias.grad.abs().max()))


@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
@pytest.mark.parametrize('modulated,dg,stride,dil', [(True, 1, 1, 1), (True, 2, 2, 1), (False, 1, 1, 2), (True, 4, 1, 1)])
def test_deform_conv(modulated, dg, stride, dil, layout):
    """pt_deform_* + GEMM == the published (modulated) deformable convolution: values and every gradient against the
    torch-gather oracle; zero offsets and unit mask reduce to a plain convolution; samples far outside contribute 0.
    `nhwc`: channels_last input (the training layout) -> the wave-per-tap kernels pt_deform_*_cl (C / dg a multiple of 4;
    dg = 4 with C = 8 falls back to the NCHW kernels), output channels_last."""
    f = F()
    gen = torch.Generator().manual_seed(51)
    B, C, H, W, O, k = 2, 8, 13, 17, 6, 3
    pad = dil
    x = torch.randn(B, C, H, W, generator=gen)
    wgt = torch.randn(O, C, k, k, generator=gen) * 0.2
    bias = torch.randn(O, generator=gen) if modulated else None
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    off = torch.randn(B, 2 * dg * k * k, Ho, Wo, generator=gen) * 2.0
    off[0, 0, 0, 0] = -50.0                                        # a sample far outside the map
    off = off + 0.013                                               # keep samples off the integer lattice (floor kinks)
    mask = torch.rand(B, dg * k * k, Ho, Wo, generator=gen) if modulated else None
    xr, offr, wr = x.clone().requires_grad_(True), off.clone().requires_grad_(True), wgt.clone().requires_grad_(True)
    mr = mask.clone().requires_grad_(True) if modulated else None
    br = bias.clone().requires_grad_(True) if bias is not None else None
    ref = R.modulated_deform_conv2d(xr, offr, mr, wr, br, stride, pad, dil, dg)
    gout = torch.randn(ref.shape, generator=gen)
    (ref * gout).sum().backward()
    fmt = (lambda t: t.contiguous(memory_format=torch.channels_last)) if layout == 'nhwc' else (lambda t: t)
    xg, offg, wg = fmt(cu(x)).requires_grad_(True), fmt(cu(off)).requires_grad_(True), fmt(cu(wgt)).requires_grad_(True)
    mg = fmt(cu(mask)).requires_grad_(True) if modulated else None
    bg = cu(bias).requires_grad_(True) if bias is not None else None
    if modulated:
        out = f.modulated_deform_conv2d(xg, offg, mg, wg, bg, stride, pad, dil, 1, dg)
    else:
        out = f.deform_conv2d(xg, offg, wg, stride, pad, dil, 1, dg)
    if layout == 'nhwc' and (C // dg) % 4 == 0:
        assert out.is_contiguous(memory_format=torch.channels_last)
    close(out, ref, atol=1e-4)
    (out * cu(gout)).sum().backward()
    close(xg.grad, xr.grad, atol=1e-4)
    close(offg.grad, offr.grad, atol=2e-4)
    close(wg.grad, wr.grad, atol=2e-4)
    if modulated:
        close(mg.grad, mr.grad, atol=1e-4)
        close(bg.grad, br.grad, atol=1e-4)
    # zero offsets (+ unit mask) == plain convolution
    z = torch.zeros_like(off)
    if modulated:
        y0 = f.modulated_deform_conv2d(fmt(cu(x)), cu(z), cu(torch.ones_like(mask)), cu(wgt), cu(bias), stride, pad, dil, 1, dg)
    else:
        y0 = f.deform_conv2d(fmt(cu(x)), cu(z), cu(wgt), stride, pad, dil, 1, dg)
    close(y0, torch.nn.functional.conv2d(x, wgt, bias, stride, pad, dil), atol=1e-4)


@pytest.mark.parametrize('C,stride,dil,sigma', [(256, 1, 1, 1.5), (64, 2, 1, 3.0), (64, 1, 2, 0.3), (128, 1, 1, 6.0)])
def test_deform_conv_tower_shape(C, stride, dil, sigma):
    """Tower-like shapes (C -> C, 3x3, dg 1) on a 40x40 map, NHWC kernels (wave-per-tap gather; grad_x staged in LDS per 8x8
    tile with a 2-pixel halo, far samples - sigma 3 / 6 px offsets - through global atomics) against the NCHW kernels of
    the same library, which the oracle test above pins: values and all gradients."""
    f = F()
    gen = torch.Generator().manual_seed(52)
    B, H, W, O = 2, 40, 40, C
    pad = dil
    Ho = (H + 2 * pad - (dil * 2 + 1)) // stride + 1
    x = torch.randn(B, C, H, W, generator=gen)
    wgt = torch.randn(O, C, 3, 3, generator=gen) * 0.02
    bias = torch.randn(O, generator=gen)
    off = torch.randn(B, 18, Ho, Ho, generator=gen) * sigma + 0.013
    mask = torch.rand(B, 9, Ho, Ho, generator=gen)
    gout = torch.randn(B, O, Ho, Ho, generator=gen)
    res = []
    for cl in (False, True):
        fmt = (lambda t: t.contiguous(memory_format=torch.channels_last)) if cl else (lambda t: t)
        xs = [fmt(cu(t)).requires_grad_(True) for t in (x, off, mask, wgt)] + [cu(bias).requires_grad_(True)]
        out = f.modulated_deform_conv2d(xs[0], xs[1], xs[2], xs[3], xs[4], stride, pad, dil, 1, 1)
        (out * cu(gout)).sum().backward()
        res.append([out.detach()] + [t.grad for t in xs])
    for a, b in zip(res[0], res[1]):
        close(b, a, rtol=1e-3, atol=1e-3 * float(a.abs().max()))


def test_deform_conv_contraction_runs_on_the_own_matrix_kernels(monkeypatch):
    """Round-4 verdict (row DCN): the deformable convolution's contraction was a hipBLASLt GEMM.  For channel counts the matrix
    kernels tile (the towers' 256 -> 256, anchor_free_head.py:101-102) the gathered columns now become split planes and the
    forward product, d W / d bias and d col run on pt_conv_bf16x6 / pt_conv_wgrad_bf16x6 (weight planes in the transposed
    `mode 2` form for d col): values and all five gradients against the float64 torch-gather oracle, and the launches counted."""
    from point_teacher_amd import hip, planes as PL
    import point_teacher_amd.functional as PF
    f = F()
    PL.CENSUS.reset()
    gen = torch.Generator().manual_seed(53)
    B, C, H, W, O = 2, 128, 20, 18, 128
    x = torch.randn(B, C, H, W, generator=gen)
    wgt = torch.randn(O, C, 3, 3, generator=gen) * 0.03
    bias = torch.randn(O, generator=gen)
    off = torch.randn(B, 18, H, W, generator=gen) * 1.5 + 0.013
    mask = torch.rand(B, 9, H, W, generator=gen)
    gout = torch.randn(B, O, H, W, generator=gen) * 1e-3
    ref_in = [t.double().clone().requires_grad_(True) for t in (x, off, mask, wgt, bias)]
    ref = R.modulated_deform_conv2d(*ref_in, 1, 1, 1, 1)
    (ref * gout.double()).sum().backward()
    calls = []
    real = hip.call
    hook = lambda fn, *a: (calls.append(fn), real(fn, *a))[1]       # noqa: E731
    monkeypatch.setattr(hip, 'call', hook)
    monkeypatch.setattr(PF.hip, 'call', hook)
    monkeypatch.setattr(PL.hip, 'call', hook)
    cl = lambda t: cu(t).contiguous(memory_format=torch.channels_last)       # noqa: E731
    xs = [cl(x).requires_grad_(True), cl(off).requires_grad_(True), cl(mask).requires_grad_(True), cl(wgt).requires_grad_(True),
          cu(bias).requires_grad_(True)]
    out = f.modulated_deform_conv2d(*xs, 1, 1, 1, 1, 1)
    (out * cu(gout)).sum().backward()
    assert calls.count('pt_conv_bf16x6') == 2 and calls.count('pt_conv_wgrad_bf16x6') == 1 and 'pt_deform_im2col_cl' in calls, calls
    err = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())        # noqa: E731
    assert err(out.detach(), ref.detach()) < 2e-6
    for n, a, b in zip(('x', 'offset', 'mask', 'weight', 'bias'), xs, ref_in):
        e = err(a.grad, b.grad)
        print(n, f'{e:.2e}')
        assert e < 5e-6, (n, e)                                       # measured 1e-7 ... 7e-7 (fp32 bilinear gather + atomics in col2im)
    assert not PL.CENSUS.demoted


def test_dcn_on_last_conv_head():
    """`dcn_on_last_conv=True` (anchor_free_head.py:101-102): the last conv of both towers is a DCNv2 pack; with its
    zero-initialised offset predictor (mask = sigmoid(0) = 0.5) the tower equals the plain one at half amplitude, and
    gradients reach the offset predictor."""
    import point_teacher_amd as pta
    from point_teacher_amd.nn_modules import ModulatedDeformConv2dPack
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
    hc = dict(cfg.model['_model_']['bbox_head'])
    hc.update(dcn_on_last_conv=True, train_cfg=cfg.model['train_cfg'], test_cfg=cfg.model['test_cfg'])
    head = pta.build_from_cfg(hc, pta.HEADS).to(DEV)
    assert isinstance(head.cls_convs[3].conv, ModulatedDeformConv2dPack) and isinstance(head.reg_convs[3].conv, ModulatedDeformConv2dPack)
    assert isinstance(head.cls_convs[2].conv, torch.nn.Conv2d)
    assert float(head.cls_convs[3].conv.conv_offset.weight.abs().max()) == 0
    x = torch.randn(2, 256, 16, 16, device=DEV, requires_grad=True)
    cls, reg, ctr, pts = head([x])
    d = head.cls_convs[3].conv
    t = x
    for l in list(head.cls_convs)[:3]:
        t = l(t)
    plain = torch.relu(torch.nn.functional.conv2d(t, d.weight, d.bias, 1, 1) * 0.5 + d.bias.view(1, -1, 1, 1) * 0.5)
    close(head.conv_cls(plain), cls[0], atol=1e-4)
    (cls[0].square().mean() + reg[0].mean()).backward()
    assert d.conv_offset.weight.grad is not None and float(d.conv_offset.weight.grad.abs().max()) > 0
    assert x.grad is not None and torch.isfinite(x.grad).all()


def test_diff_iou_rotated():
    f = F()
    gen = torch.Generator().manual_seed(41)
    N = 200
    b2 = torch.cat([torch.rand(N, 2, generator=gen) * 100 + 50, torch.rand(N, 2, generator=gen) * 40 + 6,
                    (torch.rand(N, 1, generator=gen) - 0.5) * np.pi], 1)
    b1 = b2 + torch.cat([torch.randn(N, 2, generator=gen) * 4, torch.randn(N, 2, generator=gen) * 3,
                         torch.randn(N, 1, generator=gen) * 0.2], 1)
    b1[:, 2:4] = b1[:, 2:4].clamp(min=3)
    b1[0] = torch.tensor([500., 500., 10., 10., 0.1])              # disjoint -> IoU 0, zero gradient
    iou_ref, grad_ref = R.diff_iou_rotated_grad(b1, b2)
    x = cu(b1).requires_grad_(True)
    iou = f.diff_iou_rotated_2d(x[None], cu(b2)[None])[0]
    close(iou, iou_ref, atol=2e-4)
    close(iou, f.box_iou_rotated(cu(b1), cu(b2), aligned=True), atol=1e-5)
    wgt = torch.rand(N, generator=gen)
    (iou * cu(wgt)).sum().backward()
    close(x.grad, grad_ref * wgt[:, None], rtol=2e-2, atol=2e-3)
    # theta = 0: equals the axis-aligned IoU
    hb = torch.cat([b1[:, :4], torch.zeros(N, 1)], 1), torch.cat([b2[:, :4], torch.zeros(N, 1)], 1)
    close(f.diff_iou_rotated_2d(cu(hb[0])[None], cu(hb[1])[None])[0],
          R.bbox_overlaps(R.bbox_cxcywh_to_xyxy(hb[0][:, :4]), R.bbox_cxcywh_to_xyxy(hb[1][:, :4]), is_aligned=True), atol=1e-4)


@pytest.mark.parametrize('channels_last', [False, True])
def test_roi_align_rotated(channels_last):
    f = F()
    gen = torch.Generator().manual_seed(43)
    B, C, H, W = 2, 32, 30, 30
    feat = torch.randn(B, C, H, W, generator=gen)
    K = 40
    rois = torch.cat([torch.randint(0, B, (K, 1), generator=gen).float(), torch.rand(K, 2, generator=gen) * 220 + 10,
                      torch.rand(K, 2, generator=gen) * 60 + 4, (torch.rand(K, 1, generator=gen) - 0.5) * np.pi], 1)
    rois[0, 1:3] = torch.tensor([-40., -40.])                     # outside the map
    fr = feat.clone().requires_grad_(True)
    ref = R.roi_align_rotated(fr, rois, 7, 0.125, 2, True, True)
    wgt = torch.randn(ref.shape, generator=gen)
    (ref * wgt).sum().backward()
    fg = cu(feat)
    if channels_last:
        fg = fg.contiguous(memory_format=torch.channels_last)
    fg.requires_grad_(True)
    out = f.roi_align_rotated(fg, cu(rois), 7, 0.125, 2, True, True)
    close(out, ref, atol=1e-5)
    (out * cu(wgt)).sum().backward()
    close(fg.grad, fr.grad, atol=1e-4)
    # theta = 0, adaptive grid: identical to the axis-aligned RoIAlign
    r0 = rois.clone(); r0[:, 5] = 0
    xyxy = torch.cat([r0[:, :1], r0[:, 1:3] - r0[:, 3:5] / 2, r0[:, 1:3] + r0[:, 3:5] / 2], 1)
    close(f.roi_align_rotated(fg.detach(), cu(r0), 7, 0.125, 0, True, True), f.roi_align(fg.detach(), cu(xyxy), 7, 0.125),
          atol=1e-4)


def test_obb_building_blocks():
    """DistanceAnglePointCoder, rotated FCOS targets, RotatedIoULoss / DN_IoULoss, rbbox2roi,
    RotatedSingleRoIExtractor and multiclass_nms_rotated of the OBB variant against the oracle."""
    import point_teacher_amd as pta
    from point_teacher_amd import obb
    gen = torch.Generator().manual_seed(51)
    P = 300
    pts = torch.rand(P, 2, generator=gen) * 400
    dist = torch.cat([torch.rand(P, 4, generator=gen) * 30, (torch.rand(P, 1, generator=gen) - 0.5) * 4], 1)
    coder = pta.registry.build_bbox_coder(dict(type='DistanceAnglePointCoder', angle_version='le90'))
    dec = coder.decode(cu(pts), cu(dist))
    close(dec, R.distance2obb(pts, dist, 'le90'), atol=1e-4)
    # encode(decode(x)) round trip on the distances (angle already in range)
    d2 = dist.clone(); d2[:, 4] = (torch.rand(P, generator=gen) - 0.5) * 3.0
    rt = coder.encode(cu(pts), coder.decode(cu(pts), cu(d2)))
    close(rt[:, :4], d2[:, :4], atol=1e-3)
    G = 17
    boxes5 = torch.cat([torch.rand(G, 2, generator=gen) * 400, torch.rand(G, 2, generator=gen) * 50 + 4,
                        (torch.rand(G, 1, generator=gen) - 0.5) * np.pi], 1)
    gi = torch.randint(0, G + 1, (P,), generator=gen)
    tg, ang = obb.rotated_fcos_targets(cu(pts), cu(gi), cu(boxes5))
    tr, ar = R.rotated_targets_single(pts, gi, boxes5)
    close(tg, tr, atol=1e-3); close(ang, ar, atol=1e-6)
    N = 120
    tgt = torch.cat([torch.rand(N, 2, generator=gen) * 200 + 50, torch.rand(N, 2, generator=gen) * 40 + 8,
                     (torch.rand(N, 1, generator=gen) - 0.5) * np.pi], 1)
    pred = tgt + torch.cat([torch.randn(N, 2, generator=gen) * 3, torch.randn(N, 2, generator=gen) * 2,
                            torch.randn(N, 1, generator=gen) * 0.15], 1)
    w = torch.rand(N, generator=gen)
    l1 = pta.registry.build_loss(dict(type='RotatedIoULoss', loss_weight=1.0))
    l2 = pta.registry.build_loss(dict(type='DN_IoULoss', loss_weight=1.0, hyper=0.1))
    x = cu(pred).requires_grad_(True)
    v1 = l1(x, cu(tgt), weight=cu(w), avg_factor=10.0)
    exp1 = (-R.box_iou_rotated(pred, tgt, aligned=True).clamp(min=1e-6).log() * w).sum() / 10.0
    close(v1, exp1, rtol=2e-3)
    v2 = l2(x, cu(tgt), weight=cu(w), avg_factor=10.0)
    close(v2, (R.dn_iou_loss_values(pred, tgt, 0.1) * w).sum() / 10.0, rtol=2e-3)
    v2.backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0
    rois = obb.rbbox2roi([cu(tgt[:5]), cu(tgt[:0]), cu(tgt[5:9])])
    assert rois.shape == (9, 6) and rois[:, 0].tolist() == [0.] * 5 + [2.] * 4
    ext = pta.registry.build_roi_extractor(dict(type='RotatedSingleRoIExtractor', out_channels=16, featmap_strides=[8],
                                                roi_layer=dict(type='RoIAlignRotated', out_size=7, sample_num=2, clockwise=True)))
    feat = torch.randn(3, 16, 40, 40, generator=gen)
    out = ext([cu(feat)], rois)
    close(out, R.roi_align_rotated(feat, rois.cpu(), 7, 0.125, 2, True, True), atol=1e-5)
    scores = torch.rand(N, 9, generator=gen); scores[:, -1] = 0
    dets, labels = obb.multiclass_nms_rotated(cu(tgt), cu(scores), 0.3, dict(iou_thr=0.1), 50)
    assert dets.shape[1] == 6 and dets.shape[0] == labels.shape[0] <= 50
    assert bool((dets[:-1, 5] >= dets[1:, 5]).all())


@pytest.mark.parametrize('N,C,H,W,G,relu', [(2, 256, 150, 150, 32, True), (3, 64, 37, 41, 16, False), (1, 128, 9, 7, 32, True),
                                            (2, 512, 20, 20, 32, True)])
def test_group_norm_channels_last(N, C, H, W, G, relu):
    """pt_group_norm_cl_* == torch.nn.GroupNorm (+ ReLU) on a channels_last activation (the oriented head's tower norm): values and
    the three gradients against float64 torch, an input far from zero mean (the variance is accumulated in float64), pixel counts
    that are not multiples of the 256-pixel chunk, bit-identical repeats, and the result stays channels_last."""
    from point_teacher_amd import functional as PF
    gen = torch.Generator().manual_seed(N * C + H)
    x = (torch.randn(N, C, H, W, generator=gen) * 2.0 + 30.0 * torch.randn(1, C, 1, 1, generator=gen)).to(DEV)
    x = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gn = torch.nn.GroupNorm(G, C).to(DEV)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5, generator=None); gn.bias.normal_(0, 0.3)
    assert PF.group_norm_cl_ok(x, gn)
    y = PF.group_norm_cl(x, gn, relu)
    assert y.shape == x.shape and y.is_contiguous(memory_format=torch.channels_last)
    gy = torch.randn(N, C, H, W, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)
    gx, gw, gb = torch.autograd.grad(y, (x, gn.weight, gn.bias), gy)
    xr = x.detach().double().requires_grad_(True)
    wr, br = gn.weight.detach().double().requires_grad_(True), gn.bias.detach().double().requires_grad_(True)
    pre = torch.nn.functional.group_norm(xr, G, wr, br, gn.eps)
    ref = torch.relu(pre) if relu else pre
    assert float((y.detach().double() - ref).abs().max()) < 2e-5
    mask = (y.detach() > 0).double() if relu else 1.0               # the kernel's own kink (an element within rounding of 0)
    rx, rw, rb = torch.autograd.grad(pre, (xr, wr, br), gy.double() * mask)
    for mine, r, tol in ((gx, rx, 2e-5), (gw, rw, 2e-5), (gb, rb, 2e-5)):
        assert float((mine.double() - r).abs().max() / r.abs().max()) < tol
    lib = gn(x.detach())
    e_mine = float((y.detach().double() - ref).abs().max())
    e_lib = float(((torch.relu(lib) if relu else lib).double() - ref).abs().max())
    print(f'[{N}x{C}x{H}x{W} G={G}] group norm error vs float64: own {e_mine:.2e}, torch fp32 {e_lib:.2e}')
    y2 = PF.group_norm_cl(x, gn, relu)
    g2 = torch.autograd.grad(y2, (x, gn.weight, gn.bias), gy)
    assert torch.equal(y2, y) and all(torch.equal(a, b) for a, b in zip(g2, (gx, gw, gb)))


def test_conv_module_group_norm_routes_agree():
    """mmcv-style ConvModule(conv -> GroupNorm(32) -> ReLU) of the oriented head's towers: the channels_last GroupNorm kernels (and
    the split convolution they keep reachable) against the torch.nn.GroupNorm route (PT_GN_CL=0), values and all gradients; the
    output of the own route stays channels_last."""
    from point_teacher_amd import nn_modules as NM
    torch.manual_seed(5)
    m = NM.ConvModule(256, 256, 3, padding=1, bias='auto', norm_cfg=dict(type='GN', num_groups=32, requires_grad=True)).to(DEV).to(memory_format=torch.channels_last)
    with torch.no_grad():
        m.gn.weight.uniform_(0.5, 1.5); m.gn.bias.normal_(0, 0.2)
    x = torch.randn(2, 256, 96, 100, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn(2, 256, 96, 100, device=DEV).contiguous(memory_format=torch.channels_last)
    res = {}
    for own in (True, False):
        NM._GN_CL = own
        try:
            y = m(x)
            res[own] = (y.detach(), torch.autograd.grad(y, [x] + list(m.parameters()), gy))
        finally:
            NM._GN_CL = True
    assert res[True][0].is_contiguous(memory_format=torch.channels_last)
    assert float((res[True][0] - res[False][0]).abs().max()) < 2e-5
    for a, b in zip(res[True][1], res[False][1]):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-7


def test_roi_align_rotated_backward_pixel_sorted_path():
    """2 100 RoIs (groups of four per workgroup) of 0.5 ... 19 feature pixels per side at every angle: footprints of 9 ... ~700 pixels,
    i.e. all three backward routes of pt_roi_align_rotated_bwd in one launch - the small dense product (<= 48 px), the pixel-sorted
    entries (49 ... 128 px: one atomic per touched pixel) and the per-sample taps beyond.  Forward and backward against the oracle's
    autograd on a channel subset (the op is channel-wise), all 256 channels against each other through linearity."""
    f = F()
    gen = torch.Generator().manual_seed(47)
    B, C, H, W = 2, 256, 60, 60
    K = 2100
    feat = torch.randn(B, C, H, W, generator=gen)
    rois = torch.cat([torch.randint(0, B, (K, 1), generator=gen).float(), torch.rand(K, 2, generator=gen) * 400 + 40,
                      torch.rand(K, 2, generator=gen) * 150 + 4, (torch.rand(K, 1, generator=gen) - 0.5) * np.pi], 1)
    rois[5, 1:3] = torch.tensor([-90., -90.])                         # outside the map
    rois[6, 1:5] = torch.tensor([2., 477., 120., 90.])                # a medium footprint clipped by two borders
    wgt = torch.randn(K, C, 7, 7, generator=gen)
    fg = cu(feat).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = f.roi_align_rotated(fg, cu(rois), 7, 0.125, 2, True, True)
    (out * cu(wgt)).sum().backward()
    sub = [0, 31, 64, 255]
    fr = feat[:, sub].clone().requires_grad_(True)
    ref = R.roi_align_rotated(fr, rois, 7, 0.125, 2, True, True)
    (ref * wgt[:, sub]).sum().backward()
    close(out[:, sub], ref, atol=2e-5)
    close(fg.grad[:, sub], fr.grad, rtol=1e-4, atol=5e-4)
    # every channel does the same arithmetic: the gradient of channel c for the upstream gradient of channel 0 equals channel 0's
    g0 = fg.grad[:, 0].clone()
    fg.grad = None
    w2 = cu(wgt[:, :1]).expand(-1, C, -1, -1).contiguous()
    (f.roi_align_rotated(fg, cu(rois), 7, 0.125, 2, True, True) * w2).sum().backward()
    close(fg.grad, g0[:, None].expand(-1, C, -1, -1), rtol=1e-4, atol=5e-4)


@pytest.mark.parametrize('B,H,W,C', [(2, 37, 41, 64), (1, 8, 8, 4), (3, 100, 99, 64)])
def test_affine_relu_maxpool_equals_the_three_passes(B, H, W, C):
    """pt_affine_relu_maxpool_fwd (the frozen stem's norm1 -> relu -> maxpool, resnet.py:633-640) = the fused BatchNorm pass followed
    by torch's max_pool2d(3, 2, 1), bit for bit (same fp32 multiply, add, maximum)."""
    from point_teacher_amd import functional as F
    g = torch.Generator().manual_seed(B + H)
    x = torch.randn(B, C, H, W, generator=g).to('cuda:0').contiguous(memory_format=torch.channels_last)
    sc = (torch.randn(C, generator=g)).to('cuda:0')              # (negative scales included)
    sh = torch.randn(C, generator=g).to('cuda:0')
    want = torch.nn.functional.max_pool2d(torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 3, 2, 1)
    got = F.affine_relu_maxpool(x, sc, sh)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)
    # NaN propagates as in torch's relu + max_pool2d (round-4 advice: fmaxf dropped it)
    xn = x.clone()
    xn[0, 1, H // 2, W // 2] = float('nan')
    want = torch.nn.functional.max_pool2d(torch.relu(xn * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 3, 2, 1)
    got = F.affine_relu_maxpool(xn, sc, sh)
    assert bool(torch.isnan(want).any()) and torch.equal(torch.isnan(got), torch.isnan(want))
    assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(want))
