"""Whole-iteration parity of the ORIENTED variant against the REFERENCE ITSELF (round-2 verdict P1): tests/golden/ref_iter_obb.npz
holds two consecutive iterations (burn-in step 1, then step 2) of the reference's own `RotatedFCOS_TS.forward_train`
(OBB_TOD/mmrotate/models/detectors/rotated_fcos_teacher_student.py:135-304) around a REAL `TS_P2RBRotatedFCOSHead` built by its
own constructor from the reference's own config, run in the build container on seeded weights with every random draw logged
(oracle/gen_golden_obb_iter.py lists what ran unmodified and the absent mmcv / torchvision / cv2 ops bound to the oracle).

* CPU (`-m "not gpu"`): the oracle `oracle/ref_obb.py` reproduces the golden -> its COMPOSITION of the oriented iteration (order of
  EMA / generator / MIL / augmentation, which tensors feed which loss) is pinned, not only its pieces.
* GPU (`-m gpu`): the product reproduces the same golden from the same seeds and draws: every entry of the loss dict within 1e-3
  relative (north_star), the MIL stage's merged boxes, the synthetic rectangles, gradients within GRAD_TOL of the gradient norm in
  BOTH phases."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_model as M
from oracle import ref_obb as O
from oracle import ref_ops as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRAD_KEYS = ['bbox_head.conv_cls.weight', 'bbox_head.conv_reg.weight', 'bbox_head.conv_angle.weight', 'bbox_head.conv_centerness.weight',
             'bbox_head.reg_convs.3.conv.weight', 'bbox_head.reg_convs.0.gn.weight', 'bbox_head.cls_convs.1.gn.bias',
             'bbox_head.scales.0.scale', 'bbox_head.scale_angle.scale', 'bbox_head.fc_cls.0.weight', 'bbox_head.fc_ins.0.weight',
             'bbox_head.fc_reg.0.weight', 'bbox_head.shared_fcs_bag.0.1.weight', 'bbox_head.shared_fcs_reg.0.0.weight',
             'neck_agg.lateral_convs.4.conv.weight', 'neck.fpn_convs.0.conv.weight', 'backbone.layer4.2.conv3.weight',
             'backbone.layer3.0.bn2.weight', 'backbone.layer2.0.conv1.weight']
CAP = 4096
GRAD_TOL = 3e-3          # tests/test_reference_iteration.py GRAD_TOL: the fp32 conditioning of the reference's own iteration
# The GroupNorm towers and the angle branch are worse conditioned than that in this variant: the same oracle iteration evaluated
# in float64 instead of float32 moves these gradients by 5e-3 ... 4.8e-2 in the steady-state phase while every loss value moves
# by <= 1.4e-4 (profiles/r03/obb_grad_conditioning.txt; the rotated-IoU gradient, rounded nearest-neighbour rotation of the
# augmented image and `torch.round` of the rescale sit in front of them).  Measured on MI355X vs the golden: <= 4.5e-3.
GRAD_TOL_ILL = 2e-2
ILL_CONDITIONED = ('bbox_head.cls_convs.1.gn.bias', 'bbox_head.reg_convs.0.gn.weight', 'bbox_head.reg_convs.3.conv.weight',
                   'bbox_head.conv_angle.weight', 'bbox_head.scale_angle.scale')


def _condition(sd):
    """oracle/gen_golden_obb_iter.py condition_state (restated: tests do not import generator scripts)."""
    sd = dict(sd)
    sd['bbox_head.conv_reg.bias'] = torch.full_like(sd['bbox_head.conv_reg.bias'], 1.0)
    sd['bbox_head.conv_angle.bias'] = torch.full_like(sd['bbox_head.conv_angle.bias'], 0.2)
    for k in list(sd):
        if '.fc_cls.' in k:
            sd[k] = sd[k] * 0.1
        elif '.fc_ins.' in k and k.endswith('weight'):
            sd[k] = sd[k] * 10.0
        elif '.fc_reg.' in k and k.endswith('bias'):
            sd[k] = torch.zeros_like(sd[k])
    return sd


def _states(G):
    seed = int(G['seed'])
    return _condition(O.init_detector_state_obb(seed + 1, 1)), _condition(O.init_detector_state_obb(seed + 2, 1))


def _inputs(G):
    """make_inputs(seed + 3, size, counts) of the generator (torch's CPU generator: same image, same numbers)."""
    seed, size, counts = int(G['seed']) + 3, int(G['size']), [int(c) for c in G['counts']]
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(len(counts), 3, size, size, generator=g)
    boxes, labels = [], []
    for n in counts:
        c = torch.rand(n, 2, generator=g) * (size - 64) + 32 + 0.37
        wh = torch.exp(torch.randn(n, 2, generator=g) * 0.4 + np.log(14.0)).clamp(5, 36)
        a = (torch.rand(n, 1, generator=g) - 0.5) * np.pi * 0.98
        boxes.append(torch.cat([c, wh, a], 1))
        labels.append(torch.randint(0, 9, (n,), generator=g))
    assert torch.equal(img[:, :, ::37, ::41], G.t('in_img_probe')) and float(img.double().sum()) == pytest.approx(float(G['in_img_sum']), rel=1e-12)
    for i, b in enumerate(boxes):
        assert torch.equal(b, G.t(f'in_gt_bboxes{i}')) and torch.equal(labels[i], G.t(f'in_gt_labels{i}'))
    metas = [dict(ori_filename=f'g{i}.png', filename=f'g{i}.png', img_shape=(size, size, 3), scale_factor=np.ones(4, np.float32))
             for i in range(len(counts))]
    return img, boxes, labels, metas


def _inject(G, it, dev='cpu'):
    tag = f'it{it}_'
    inj = dict(neg0=G.t(tag + 'draw_neg0').to(dev),
               aug=([str(f) for f in G[tag + 'aug_flips']], [int(a) for a in G[tag + 'aug_angles']], [float(s) for s in G[tag + 'aug_scales']]))
    if tag + 'draw_syn0_cls' in G:
        names = ('cls', 'scale', 'x', 'y', 'wn', 'rn', 'a', 'boost', 'itv', 'itv2', 'dev')
        inj['syn'] = [{n: G.t(tag + f'draw_syn{i}_{n}').to(dev) for n in names} for i in range(2)]
    return inj


def _sample(g):
    flat = g.reshape(-1)
    return flat[::max(1, -(-flat.numel() // CAP))]


def _check_losses(got, G, it, rel):
    keys = [str(k) for k in G[f'it{it}_keys']] + ['loss']
    assert set(got.keys()) == set(keys), set(got.keys()) ^ set(keys)
    for k in keys:
        a, b = float(got[k]), float(G[f'it{it}_out_{k}'])
        assert abs(a - b) <= rel * max(abs(b), 1e-2) + 1e-6, (it, k, a, b)


def _check_grads(grads, G, it, rel):
    worst = {}
    for k in GRAD_KEYS:
        ref = G.t(f'it{it}_grad_{k}').double()
        nref = float(G[f'it{it}_gradnorm_{k}'])
        if nref == 0.0:                                   # a parameter the phase does not reach
            assert grads.get(k) is None or float(grads[k].abs().max()) == 0.0, (it, k)
            continue
        got = _sample(grads[k].detach().cpu()).double()
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        err = float((got - ref).norm() / (ref.norm() + 1e-30))
        worst[k] = err
        tol = max(rel, GRAD_TOL_ILL) if (k in ILL_CONDITIONED and rel >= GRAD_TOL) else rel
        assert err < tol, (it, k, err)
        assert abs(float(grads[k].double().norm()) - nref) <= tol * nref, (it, k)
    return worst


# ------------------------------------------------------------------------ CPU: oracle vs the reference --
def test_oracle_obb_iteration_vs_reference():
    G = load_golden('ref_iter_obb')
    torch.set_num_threads(8)
    sd_s, sd_t = _states(G)
    probe = [float(sd_s['backbone.layer3.2.conv2.weight'].sum()), float(sd_t['bbox_head.shared_fcs_bag.0.0.weight'][5, :64].sum()),
             float(sd_s['bbox_head.fc_ins.0.weight'].abs().sum())]
    assert probe == pytest.approx([float(v) for v in G['weight_probe']], rel=1e-6), 'seeded weights differ from the generator run'
    img, boxes, labels, _ = _inputs(G)
    points = [b[:, :2] for b in boxes]                                            # _point_ = 'center' (genrate_points :435-436)
    for it in range(2):
        inj = _inject(G, it)
        stats = {}
        inj['stats'] = stats
        sd_t = M.ema(sd_t, sd_s)                                                   # :144, before anything else
        params = {k: (v.clone().requires_grad_(True) if O.trainable_obb(k) else v) for k, v in sd_s.items()}
        step = O.forward_train_step1 if it == 0 else O.forward_train_step2         # count <= burn_in_step (= 0), :151
        ref, points = step(params, sd_t, img, boxes, labels, points, dict(O.MODEL_CFG), inj)
        ref['loss'] = M.total_loss(ref)
        _check_losses(ref, G, it, rel=5e-5)
        for i in range(2):
            torch.testing.assert_close(points[i].detach(), G.t(f'it{it}_out_points{i}'), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(stats['merged0'], G.t(f'it{it}_out_merged')[-stats['merged0'].shape[0]:], rtol=1e-5, atol=1e-3)
        # no pick of the reference's own top-3 of 25 is decided by rounding on this fixture (ref_obb.mil_select_margin)
        assert int(((stats['gap0'] < 2e-3) & (stats['shift0'] > 1e-3)).sum()) == 0
        live = [k for k in GRAD_KEYS if float(G[f'it{it}_gradnorm_{k}']) > 0]
        gr = dict(zip(live, torch.autograd.grad(ref['loss'], [params[k] for k in live], allow_unused=True)))
        _check_grads(gr, G, it, rel=5e-4)
        points = [p.detach() for p in points]


def test_oracle_obb_black_paper_vs_reference():
    """The oriented rectangle generator against what the reference's own genrate_syn / generate_black_paper produced inside the
    iteration: the NMS keep-set, the surviving boxes and every painted pixel (painted with the image maximum, :722)."""
    G = load_golden('ref_iter_obb')
    img, boxes, _, _ = _inputs(G)
    inj = _inject(G, 0)
    prior = torch.tensor(O.MODEL_CFG['shape_list'])
    for i in range(2):
        img_syn, obb5, keep, _ = R.generate_black_paper(img[i], boxes[i], prior, inj['syn'][i], oriented=True)
        assert torch.equal(keep, G.t(f'it0_out_syn_keep{i}'))
        torch.testing.assert_close(obb5, G.t(f'it0_out_syn_boxes{i}'), rtol=1e-6, atol=1e-4)
        mask_ref = np.unpackbits(G[f'it0_out_syn_mask{i}'], axis=-1)[:, :img.shape[-1]].astype(bool)
        painted = (img_syn[0] != img[i][0]).numpy() | (img_syn == img_syn.max()).all(0).numpy()
        assert obb5.shape[0] > 0 and np.array_equal(painted, mask_ref)
        assert float(img_syn.max()) == float(img[i].max())


# ------------------------------------------------------------------------ GPU: product vs the reference --
@pytest.mark.gpu
@pytest.mark.parametrize('layout', ['nchw', 'channels_last'])
def test_product_obb_iteration_vs_reference(layout):
    """`channels_last` = the training layout: the trunk (trainable eval-mode BatchNorm), necks and FC stacks plane-native."""
    import point_teacher_amd as pta
    dev = torch.device('cuda:0')
    G = load_golden('ref_iter_obb')
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'obb', 'point_teacher', 'sodaa_fcos_pointteacher_1x.py'))
    cfg.model['burn_in_step'] = 0
    torch.manual_seed(0)
    model = pta.build_detector(cfg.model)
    sd_s, sd_t = _states(G)
    miss = model.student.load_state_dict(sd_s, strict=False)
    assert not miss.unexpected_keys and all('num_batches_tracked' in k for k in miss.missing_keys), miss     # incl.: no fc_iou in this head
    model.teacher.load_state_dict(sd_t, strict=False)
    model = model.to(dev).train()
    img, boxes, labels, metas = _inputs(G)
    img_dev = img.to(dev)
    if layout == 'channels_last':
        model = model.to(memory_format=torch.channels_last)
        img_dev = img_dev.contiguous(memory_format=torch.channels_last)
    data = dict(img=img_dev, img_metas=metas, gt_bboxes=[b.to(dev) for b in boxes], gt_labels=[l.to(dev) for l in labels])
    seen = {}
    real_syn = model.genrate_syn

    def spy(*a, **k):
        r = real_syn(*a, **k)
        seen['img_syn'], seen['boxes'], seen['alive'] = r[0], r[2], r[3]
        return r
    model.genrate_syn = spy
    head = model.student.bbox_head
    real_sel = head.mil_bag_selection

    def spy_sel(*a, **k):
        r = real_sel(*a, **k)
        seen.setdefault('merged', []).append(torch.cat(r).detach().cpu())
        return r
    head.mil_bag_selection = spy_sel
    report, grads_by_it = {}, {}
    for it in range(2):
        model._inject = _inject(G, it, dev)
        seen.pop('merged', None)
        for p in model.student.parameters():
            p.grad = None
        out = model.train_step(data, None)
        out['loss'].backward()
        lv = out['log_vars'].materialize()
        _check_losses(lv, G, it, rel=1e-3)
        for i in range(2):
            torch.testing.assert_close(model.refined_gt_bboxes_point[metas[i]['ori_filename']].cpu(), G.t(f'it{it}_out_points{i}'),
                                       rtol=1e-5, atol=1e-3)
        ref_merged = G.t(f'it{it}_out_merged')
        got_merged = torch.cat(seen['merged'])
        torch.testing.assert_close(got_merged[-ref_merged.shape[0]:] if got_merged.shape[0] >= ref_merged.shape[0] else got_merged,
                                   ref_merged[-got_merged.shape[0]:], rtol=1e-4, atol=2e-3)
        grads = {k: p.grad for k, p in model.student.named_parameters() if k in GRAD_KEYS}
        grads_by_it[it] = {k: g.clone() for k, g in grads.items() if g is not None}
        report[it] = _check_grads(grads, G, it, rel=GRAD_TOL)
        if it == 0:            # the generator through the product: survivors in the reference's order, painted pixels
            for i in range(2):
                torch.testing.assert_close(seen['boxes'][i][seen['alive'][i]].cpu(), G.t(f'it0_out_syn_boxes{i}'), rtol=1e-6, atol=1e-3)
                mask_ref = np.unpackbits(G[f'it0_out_syn_mask{i}'], axis=-1)[:, :img.shape[-1]].astype(bool)
                im = seen['img_syn'][i].cpu()
                assert np.array_equal((im[0] != img[i][0]).numpy() | (im == im.max()).all(0).numpy(), mask_ref)
    assert model.count == 2
    print('gradient errors (relative to the sample norm):', {it: {k: f'{v:.1e}' for k, v in r.items()} for it, r in report.items()})
    # The five ill-conditioned tensors against the FLOAT64 evaluation of the same iteration (tests/golden/ref_iter_obb_fp64.npz,
    # oracle/gen_golden_obb_fp64.py): the golden above is the reference's own fp32 run, itself 5e-3 ... 5e-2 away from float64 on
    # these tensors - a 2e-2 bound against it cannot tell a 1 % kernel error from conditioning (round-3 verdict P2).  Against
    # float64 the product's fp32 gradients must not be further away than the reference's fp32 gradients are (x 1.5 + 3e-3: both
    # are single fp32 runs scattered around the float64 value).
    G64 = load_golden('ref_iter_obb_fp64')
    cmp64 = {}
    for it in range(2):
        for k in ILL_CONDITIONED:
            if f'it{it}_grad64_{k}' not in G64:
                continue
            r64 = G64.t(f'it{it}_grad64_{k}').double()
            ref32 = G.t(f'it{it}_grad_{k}').double()
            got = _sample(grads_by_it[it][k].detach().cpu()).double()
            e_prod, e_ref = float((got - r64).norm() / r64.norm()), float((ref32 - r64).norm() / r64.norm())
            cmp64[(it, k)] = (e_prod, e_ref)
            assert e_prod <= 1.5 * e_ref + GRAD_TOL, (it, k, e_prod, e_ref)
    print('ill-conditioned gradients vs float64 (product, reference fp32):', {f'{it}:{k}': (f'{a:.1e}', f'{b:.1e}') for (it, k), (a, b) in cmp64.items()})
