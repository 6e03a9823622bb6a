"""Pins the oriented-box oracle (oracle/ref_obb.py + the rotated helpers of ref_ops.py) to golden vectors
captured from the reference itself (oracle/gen_golden_obb.py -> tests/golden/obb_*.npz).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_obb as O
from oracle import ref_ops as R


def close(a, b, rtol=1e-5, atol=1e-5):
    torch.testing.assert_close(torch.as_tensor(a).double(), torch.as_tensor(b).double(), rtol=rtol, atol=atol)


def test_transforms():
    g = load_golden('obb_transforms')
    rb = g.t('in_rboxes')
    close(R.obb2poly_le90(rb), g.t('out_poly'))
    close(O.poly2obb_le90(g.t('out_poly')), g.t('out_back'))
    close(O.poly2obb_le90(g.t('in_quads')), g.t('out_quads_obb'))
    close(O.norm_angle_le90(g.t('in_angles')), g.t('out_norm_le90'))
    close(R.obb2xyxy(rb), g.t('out_obb2xyxy'))


def test_distance_angle_point_coder():
    g = load_golden('obb_coder')
    close(R.distance2obb(g.t('in_points'), g.t('in_distance'), 'le90'), g.t('out_decode'))
    close(O.obb2distance(g.t('in_points'), g.t('in_gt')), g.t('out_encode'), atol=1e-4)
    close(O.obb2distance(g.t('in_points'), g.t('in_gt'), 16.0, 0.1), g.t('out_encode_clamped'), atol=1e-4)


def test_mil_bag_construction():
    g = load_golden('obb_proposals')
    for tag, cfg in (('coarse', dict(base_ratios=[1.0], shake_ratio=None, min_scale=0)),
                     ('ext', dict(base_ratios=[1.0, 1.2, 1.3, 0.8, 0.6], shake_ratio=None, min_scale=4))):
        for i in range(2):
            b = g.t(f'in_boxes{i}')
            props, valid = O.fine_proposals_obb(b, cfg, (256, 256))
            close(props, g.t(f'out_{tag}_props{i}'), atol=1e-4)
            assert torch.equal(valid.reshape(-1), g.t(f'out_{tag}_valid{i}').reshape(-1))
            U = props.shape[0] // b.shape[0]
            close(b[:, None].repeat(1, U, 1).reshape(-1, 5), g.t(f'out_{tag}_ref{i}'))
            close(g.t(f'in_real{i}')[:, None].repeat(1, U, 1).reshape(-1, 5), g.t(f'out_{tag}_real{i}'))
    assert not bool(g.t('out_ext_valid0').all())          # the fixture contains bags that leave the image


def test_strong_augmentation_geometry():
    g = load_golden('obb_strong_aug')
    H, W = [int(v) for v in g['hw']]
    for i in range(4):
        gp, gl, pp, pl, pb = O.strong_augmentation_geometry_obb(
            g.t(f'in_gt_points{i}'), g.t(f'in_labels{i}'), g.t(f'in_pseudo_points{i}'), g.t(f'in_labels{i}'),
            g.t(f'in_pseudo_bboxes{i}'), str(g['flips'][i]), int(g['angles'][i]), float(g['scales'][i]), H, W)
        close(gp, g.t(f'out_gt_points{i}'), atol=1e-4)
        assert torch.equal(gl, g.t(f'out_gt_labels{i}'))
        close(pp, g.t(f'out_pseudo_points{i}'), atol=1e-4)
        assert torch.equal(pl, g.t(f'out_pseudo_labels{i}'))
        close(pb, g.t(f'out_pseudo_bboxes{i}'), atol=1e-4)
    assert sum(g.t(f'out_gt_points{i}').shape[0] for i in range(4)) < 32      # the filters really dropped something


@pytest.mark.parametrize('name', ['small', 'mid'])
def test_head_pseudo_boxes_and_targets(name):
    g = load_golden('obb_head_' + name)
    pts, cls, reg = g.t('in_points'), g.t('in_cls'), g.t('in_reg')
    gtb, lab = g.t('in_gt_bboxes'), g.t('in_gt_labels')
    pb, pp, ps, _, valid = O.generate_pseudo_single_obb(pts, cls, reg, gtb[:, :2].contiguous(), lab, gtb)
    close(pb, g.t('out_pseudo_bboxes'), atol=1e-4)
    close(pp, g.t('out_pseudo_points'), atol=1e-4)
    assert sorted(valid.tolist()) == sorted(g.t('out_valid').tolist())
    # synthetic-box targets (syn_assigner 3,3; every label 0)
    gi, gl = R.topk_assign(pts, gtb, torch.zeros(gtb.shape[0], dtype=torch.long), 3)
    labels = torch.full((pts.shape[0],), 9, dtype=torch.long)
    labels[gi != 0] = gl[gi != 0]
    assert torch.equal(labels, g.t('out_syn_labels'))
    t, a = R.rotated_targets_single(pts, gi, gtb)
    close(t, g.t('out_syn_bbox_targets'), atol=1e-4)
    close(a, g.t('out_syn_angle_targets'))
    # pseudo-box targets (assigner 1,1 for the class labels, pseudo_assigner 3,3 for the regression)
    pb2 = g.t('in_pseudo_bboxes2')
    gi, gl = R.topk_assign(pts, gtb[:, :2].contiguous(), lab, 1)
    lc = torch.full((pts.shape[0],), 9, dtype=torch.long)
    lc[gi != 0] = gl[gi != 0]
    assert torch.equal(lc, g.t('out_labels_cls'))
    gi, gl = R.topk_assign(pts, pb2, lab, 3)
    lr = torch.full((pts.shape[0],), 9, dtype=torch.long)
    lr[gi != 0] = gl[gi != 0]
    assert torch.equal(lr, g.t('out_labels_reg'))
    t, a = R.rotated_targets_single(pts, gi, pb2)
    close(t, g.t('out_bbox_targets'), atol=1e-4)
    close(a, g.t('out_angle_targets'))
    close(R.centerness_target(t[lr < 9]), g.t('out_centerness'), atol=1e-5)


def test_mil_selection_and_bag_loss():
    g = load_golden('obb_mil')
    cls = g.t('in_cls').clone().requires_grad_(True)
    ins = g.t('in_ins').clone().requires_grad_(True)
    neg = g.t('in_neg_cls').clone().requires_grad_(True)
    valid, labels = g.t('in_valid'), g.t('in_labels')
    counts = [int(c) for c in g['counts']]
    merged, o = [], 0
    for c in counts:            # per image: the clamp uses that image's (h, w) - identical here
        sl = slice(o, o + c)
        merged.append(O.mil_bag_select_obb(cls[sl], ins[sl], valid[o * 25:(o + c) * 25], labels[sl],
                                           g.t('in_bags')[o * 25:(o + c) * 25], g.t('in_pseudo')[sl], (200, 240), 3, 0.25))
        o += c
    close(torch.cat(merged), g.t('out_merged'), atol=1e-4)
    loss = O.mil_bag_loss_obb(cls, ins, valid, labels, neg, g.t('in_neg_w'))
    close(loss, g.t('out_loss'), rtol=1e-5)
    gr = torch.autograd.grad(loss, [cls, ins, neg])
    close(gr[0], g.t('out_grad_cls'), atol=1e-6)
    close(gr[1], g.t('out_grad_ins'), atol=1e-6)
    close(gr[2], g.t('out_grad_neg'), atol=1e-6)
    close(O.mil_bag_loss_obb(cls, ins, valid, labels, None, None), g.t('out_loss_pos_only'), rtol=1e-5)


def test_product_eval_decode_matches_reference():
    """The decode half of the oriented eval path (`TS_P2RBRotatedFCOSHead.get_bboxes(with_nms=False)`: per-level top-k by
    the best class score, DistanceAnglePointCoder.decode with clipping, rescale, background column) against what the
    reference's own get_bboxes hands to the rotated NMS (tests/golden/obb_eval_decode.npz, gen_golden_obb.py gen_eval_decode).
    Pure torch in the product, so it runs on the CPU; centerness is NOT applied (the reference passes ones as score factors)."""
    import os
    import point_teacher_amd as pta
    E = load_golden('obb_eval_decode')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = pta.Config.fromfile(os.path.join(root, 'point_teacher_amd', 'configs', 'obb', 'point_teacher', 'sodaa_fcos_pointteacher_1x.py'))
    hcfg = dict(cfg.model['_model_']['bbox_head'])
    hcfg.update(strides=[8, 16, 32], train_cfg=cfg.model['_model_'].get('train_cfg'), test_cfg=cfg.model['_model_'].get('test_cfg'))
    h = pta.registry.HEADS.build(hcfg)
    cls, reg, ang, ctr = ([E.t(f'in_{n}{i}') for i in range(3)] for n in ('cls', 'reg', 'ang', 'ctr'))
    pts = [E.t(f'in_points{i}') for i in range(3)]
    metas = [dict(img_shape=(150, 180, 3), scale_factor=np.array([1.5, 1.25, 1.5, 1.25], np.float32)),
             dict(img_shape=(160, 192, 3), scale_factor=np.array([1.0, 1.0, 1.0, 1.0], np.float32))]
    tc = dict(nms_pre=60, score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=100)
    for tag, rescale in (('plain', False), ('rescale', True)):
        res = h.get_bboxes(cls, reg, ang, ctr, pts, metas, cfg=tc, rescale=rescale, with_nms=False)
        for b in range(2):
            bb, sc = res[b]
            assert bb.shape == (150, 5) and sc.shape == (150, 10)
            np.testing.assert_allclose(bb.numpy(), E[f'out_{tag}_bboxes{b}'], rtol=1e-6, atol=1e-6)
            np.testing.assert_array_equal(sc.numpy(), E[f'out_{tag}_scores{b}'])
            assert (E[f'out_{tag}_factors{b}'] == 1).all()
