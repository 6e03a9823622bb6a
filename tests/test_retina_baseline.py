"""RetinaNet baseline (second entry): anchors, MaxIoU assignment, targets, loss.

CPU: oracle/ref_retina.py AND the product's AnchorGenerator against tests/golden/retina_baseline.npz (outputs of the
reference's own AnchorGenerator / MaxIoUAssigner / AnchorHead, oracle/gen_golden_retina.py).  GPU (-m gpu):
`pt_max_iou_assign` through the C ABI - assignment bit-exact in four assigner configurations, incl. the box that overlaps
no anchor (with min_pos_iou = 0 it claims every zero-IoU anchor, as in the reference) - `RetinaHead` targets exact, losses
within 1e-3 and gradients, then training iterations of the baseline config."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import load_golden  # noqa: E402
from oracle import ref_retina as RR  # noqa: E402

STRIDES = [8, 16, 32]
SIZES = [(20, 24), (10, 12), (5, 6)]
G = load_golden('retina_baseline')
NL, NB = 3, 3
MODES = {'all': dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0),
         'first': dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.2, gt_max_assign_all=False),
         'range': dict(pos_iou_thr=0.6, neg_iou_thr=(0.1, 0.4), min_pos_iou=0.3),
         'nolow': dict(pos_iou_thr=0.5, neg_iou_thr=0.4, match_low_quality=False)}


def _inputs(dev='cpu'):
    cls = [G.t(f'in_cls{i}').to(dev) for i in range(NL)]
    reg = [G.t(f'in_reg{i}').to(dev) for i in range(NL)]
    boxes = [G.t(f'in_gt_bboxes{b}').to(dev) for b in range(NB)]
    labels = [G.t(f'in_gt_labels{b}').to(dev) for b in range(NB)]
    return cls, reg, boxes, labels


def test_oracle_and_anchor_generator_match_reference():
    from point_teacher_amd.retina_baseline import AnchorGenerator
    cls, reg, boxes, labels = _inputs()
    gen = AnchorGenerator(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=STRIDES)
    lvl = RR.grid_anchors(SIZES, STRIDES, RR.octave_scales(), (0.5, 1.0, 2.0))
    mine = gen.grid_anchors(SIZES, 'cpu')
    pad = tuple(int(v) for v in G['in_pad_hw'])
    fl_o = RR.valid_flags(SIZES, STRIDES, pad, 9)
    fl_m = gen.valid_flags(SIZES, pad + (3,), 'cpu')
    for i in range(NL):
        np.testing.assert_array_equal(gen.base_anchors[i].numpy(), G[f'out_base{i}'])
        np.testing.assert_array_equal(lvl[i].numpy(), G[f'out_anchors{i}'])
        np.testing.assert_array_equal(mine[i].numpy(), G[f'out_anchors{i}'])
        np.testing.assert_array_equal(fl_o[i].numpy(), G[f'out_flags{i}'])
        np.testing.assert_array_equal(fl_m[i].numpy(), G[f'out_flags{i}'])
    flat = torch.cat(lvl)
    for tag, kw in MODES.items():
        for b in range(NB):
            asg, mo = RR.max_iou_assign(flat, boxes[b], **kw)
            np.testing.assert_array_equal(asg.numpy(), G[f'out_assign_{tag}_{b}'])
            np.testing.assert_array_equal(mo.numpy(), G[f'out_maxov_{tag}_{b}'])
    # the box no anchor overlaps (best IoU 0, min_pos_iou 0) claims every anchor whose IoU with it is 0 - unless a later box wins
    a0 = G['out_assign_all_0']
    assert (a0 == 4).sum() > 100 and (G['out_assign_first_0'] == 4).sum() == 0
    for t in cls + reg:
        t.requires_grad_(True)
    losses, lv_t, npos = RR.loss(cls, reg, boxes, labels, STRIDES, (160, 192), 8, MODES['all'])
    assert npos == int(G['out_num_total_pos'])
    for i in range(NL):
        np.testing.assert_array_equal(lv_t[i][0].numpy(), G[f'out_labels{i}'].reshape(-1))
        np.testing.assert_array_equal(lv_t[i][1].numpy(), G[f'out_label_weights{i}'].reshape(-1))
        np.testing.assert_allclose(lv_t[i][2].numpy(), G[f'out_bbox_targets{i}'].reshape(-1, 4), rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(lv_t[i][3].numpy(), G[f'out_bbox_weights{i}'].reshape(-1, 4))
        assert float(losses['loss_cls'][i].detach()) == pytest.approx(float(G[f'out_loss_cls{i}']), rel=1e-5)
        assert float(losses['loss_bbox'][i].detach()) == pytest.approx(float(G[f'out_loss_bbox{i}']), rel=1e-5)
    grads = torch.autograd.grad(sum(losses['loss_cls']) + sum(losses['loss_bbox']), cls + reg)
    for i in range(NL):
        np.testing.assert_allclose(grads[i].numpy(), G[f'out_grad_cls{i}'], rtol=1e-4, atol=1e-8)
        np.testing.assert_allclose(grads[NL + i].numpy(), G[f'out_grad_reg{i}'], rtol=1e-4, atol=1e-8)


def test_retinanet_config_builds():
    import point_teacher_amd as pta
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'baselines', 'aitodv2_retinanet_r50_1x.py'))
    model = pta.build_detector(cfg.model)
    assert type(model).__name__ == 'RetinaNet' and type(model.bbox_head).__name__ == 'RetinaHead'
    names = set(model.bbox_head.state_dict())
    assert {'cls_convs.3.conv.weight', 'reg_convs.0.conv.bias', 'retina_cls.bias', 'retina_reg.weight'} <= names and len(names) == 20
    assert model.bbox_head.retina_cls.out_channels == 72 and model.bbox_head.retina_reg.out_channels == 36
    assert model.neck.add_extra_convs == 'on_input' and cfg.optimizer.lr == 0.005


def _head(dev):
    from point_teacher_amd.retina_baseline import RetinaHead
    return RetinaHead(8, 4, stacked_convs=1, feat_channels=4,
                      anchor_generator=dict(type='AnchorGenerator', octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0],
                                            strides=STRIDES),
                      train_cfg=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0,
                                                   ignore_iof_thr=-1, gpu_assign_thr=512), allowed_border=-1, pos_weight=-1)).to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', list(MODES))
def test_max_iou_assign_kernel_vs_reference(tag):
    from point_teacher_amd import functional as PF
    dev = torch.device('cuda:0')
    _, _, boxes, _ = _inputs(dev)
    anchors = torch.cat([G.t(f'out_anchors{i}') for i in range(NL)]).to(dev)
    kw = dict(MODES[tag])
    off, _ = PF.make_offsets([int(b.shape[0]) for b in boxes], dev)
    inds, mo = PF.max_iou_assign(anchors, torch.cat(boxes), off, NB, kw.pop('pos_iou_thr'), kw.pop('neg_iou_thr'), **kw)
    for b in range(NB):
        np.testing.assert_array_equal(inds[b].cpu().numpy(), G[f'out_assign_{tag}_{b}'])      # index work: bit-exact
        np.testing.assert_array_equal(mo[b].cpu().numpy(), G[f'out_maxov_{tag}_{b}'])


@pytest.mark.gpu
def test_retina_head_targets_and_loss_vs_reference():
    dev = torch.device('cuda:0')
    cls, reg, boxes, labels = _inputs(dev)
    for t in cls + reg:
        t.requires_grad_(True)
    h = _head(dev)
    metas = [dict(img_shape=(160, 192, 3), pad_shape=(160, 192, 3))] * NB
    anchors = torch.cat(h.anchor_generator.grid_anchors(SIZES, dev))
    lab, lw, tg, bw, npos = h.get_targets(anchors, boxes, labels, metas, SIZES)
    assert int(npos) == int(G['out_num_total_pos'])
    n = [G[f'out_anchors{i}'].shape[0] for i in range(NL)]
    o = 0
    for i in range(NL):
        np.testing.assert_array_equal(lab[:, o:o + n[i]].cpu().numpy(), G[f'out_labels{i}'])
        np.testing.assert_array_equal(lw[:, o:o + n[i]].cpu().numpy(), G[f'out_label_weights{i}'])
        np.testing.assert_allclose(tg[:, o:o + n[i]].cpu().numpy(), G[f'out_bbox_targets{i}'], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(bw[:, o:o + n[i]].expand(-1, -1, 4).cpu().numpy(), G[f'out_bbox_weights{i}'])
        o += n[i]
    losses = h.loss(cls, reg, boxes, labels, metas)
    for i in range(NL):
        assert float(losses['loss_cls'][i].detach()) == pytest.approx(float(G[f'out_loss_cls{i}']), rel=1e-3)
        assert float(losses['loss_bbox'][i].detach()) == pytest.approx(float(G[f'out_loss_bbox{i}']), rel=1e-3)
    grads = torch.autograd.grad(sum(losses['loss_cls']) + sum(losses['loss_bbox']), cls + reg)
    for i in range(NL):
        for j, name in enumerate(('cls', 'reg')):
            g, w = grads[j * NL + i].cpu().numpy().ravel(), G[f'out_grad_{name}{i}'].ravel()
            np.testing.assert_allclose(g, w, rtol=2e-3, atol=1e-6 * np.abs(w).max())
    # a padded shape smaller than the maps switches anchors off through valid_flags
    pad = tuple(int(v) for v in G['in_pad_hw'])
    lab2, lw2, _, _, _ = h.get_targets(anchors, boxes, labels, [dict(img_shape=pad + (3,), pad_shape=pad + (3,))] * NB, SIZES)
    flags = torch.cat([G.t(f'out_flags{i}') for i in range(NL)]).to(dev)
    assert bool((lw2[:, ~flags] == 0).all()) and bool((lab2[:, ~flags] == 8).all())


@pytest.mark.gpu
def test_retinanet_trains_and_infers():
    import point_teacher_amd as pta
    import test_train_step_parity as T
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'baselines', 'aitodv2_retinanet_r50_1x.py'))
    model = pta.build_detector(cfg.model).to(dev)
    model.train()
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config)
    img, boxes, labels, metas = T._data(dev)
    img = (img - 110.0) / 58.0                                             # the config normalises its input
    data = dict(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in boxes], gt_labels=[l.to(dev) for l in labels])
    first = None
    for it in range(6):
        lv = trainer.step(data)['log_vars'].materialize()
        assert set(lv) == {'loss_cls', 'loss_bbox', 'loss'} and all(np.isfinite(v) for v in lv.values()), lv
        first = first or lv['loss']
    assert lv['loss'] < first
    model.eval()
    with torch.no_grad():
        res = model(return_loss=False, rescale=True, img=[data['img']], img_metas=[metas])
    assert len(res) == 2 and len(res[0]) == 8 and all(r.shape[1] == 5 for r in res[0])


@pytest.mark.gpu
def test_max_iou_assign_full_size():
    """BASELINE size: 120 087 anchors (800x800, five levels, nine per location) x 600 boxes per image, two images - against
    the reference algorithm written with torch ops on the device (the explicit [G, A] matrix the kernel never stores)."""
    from point_teacher_amd import functional as PF
    from point_teacher_amd.retina_baseline import AnchorGenerator
    from point_teacher_amd.synthetic import make_tile
    dev = torch.device('cuda:0')
    gen = AnchorGenerator(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128])
    anchors = torch.cat(gen.grid_anchors([(100, 100), (50, 50), (25, 25), (13, 13), (7, 7)], dev))
    assert anchors.shape[0] == 120087
    boxes = [make_tile(i, 800, 8, 600, seed=9, device=dev, max_objects=600)[1] for i in range(2)]
    off, _ = PF.make_offsets([int(b.shape[0]) for b in boxes], dev)
    inds, mo = PF.max_iou_assign(anchors, torch.cat(boxes), off, 2, 0.5, 0.4, 0.0, True, True)
    for b in range(2):
        g = boxes[b]
        a1 = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
        a2 = (anchors[:, 2] - anchors[:, 0]) * (anchors[:, 3] - anchors[:, 1])
        wh = (torch.min(g[:, None, 2:], anchors[None, :, 2:]) - torch.max(g[:, None, :2], anchors[None, :, :2])).clamp(min=0)
        ov = wh[..., 0] * wh[..., 1]
        ov = ov / torch.max(a1[:, None] + a2[None, :] - ov, ov.new_tensor([1e-6]))
        m, am = ov.max(0)
        gm = ov.max(1)[0]
        want = torch.full((anchors.shape[0],), -1, dtype=torch.long, device=dev)
        want[(m >= 0) & (m < 0.4)] = 0
        want[m >= 0.5] = am[m >= 0.5] + 1
        for i in range(g.shape[0]):
            want[ov[i] == gm[i]] = i + 1
        assert torch.equal(mo[b], m)
        assert torch.equal(inds[b].long(), want)
        assert int((inds[b] > 0).sum()) >= g.shape[0] * 0.5


@pytest.mark.gpu
def test_retina_eval_decode_vs_reference():
    """RetinaHead.get_bboxes(with_nms=False) - per-level top-k, `pt_delta2bbox_fwd` with clipping, rescale, background
    column - against the reference's own AnchorHead.get_bboxes output (tests/golden/retina_eval_decode.npz)."""
    E = load_golden('retina_eval_decode')
    dev = torch.device('cuda:0')
    h = _head(dev)
    cls = [E.t(f'in_cls{i}').to(dev) for i in range(NL)]
    reg = [E.t(f'in_reg{i}').to(dev) for i in range(NL)]
    metas = [dict(img_shape=(150, 180, 3), scale_factor=np.array([1.5, 1.25, 1.5, 1.25], np.float32)),
             dict(img_shape=(160, 192, 3), scale_factor=np.array([1.0, 1.0, 1.0, 1.0], np.float32))]
    cfg = dict(nms_pre=200, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100)
    for tag, rescale in (('plain', False), ('rescale', True)):
        res = h.get_bboxes(cls, reg, metas, cfg=cfg, rescale=rescale, with_nms=False)
        for b in range(2):
            bb, sc = res[b]
            assert bb.shape == E[f"out_{tag}_bboxes{b}"].shape == (200 + 200 + 200, 4)
            np.testing.assert_allclose(bb.cpu().numpy(), E[f'out_{tag}_bboxes{b}'], rtol=1e-5, atol=1e-4)
            np.testing.assert_allclose(sc.cpu().numpy(), E[f'out_{tag}_scores{b}'], rtol=1e-6, atol=1e-7)   # sigmoid on the GPU: 1 ulp


KNOWN_BOXES = [[0, 0, 10, 10], [10, 10, 20, 20], [5, 5, 15, 15], [32, 32, 38, 42]]
KNOWN_GTS = [[0, 0, 10, 9], [0, 10, 10, 19]]


def test_oracle_reference_known_answers():
    """The known answers of the reference's own assigner tests (HBB_TOD/tests/test_utils/test_assigner.py:15-36 and :66-82)."""
    b, g = torch.FloatTensor(KNOWN_BOXES), torch.FloatTensor(KNOWN_GTS)
    assert RR.max_iou_assign(b, g, 0.5, 0.5)[0].tolist() == [1, 0, 2, 0]
    assert RR.max_iou_assign(b, torch.empty(0, 4), 0.5, 0.5)[0].tolist() == [0, 0, 0, 0]


@pytest.mark.gpu
def test_assigner_reference_known_answers():
    """The same known answers through `MaxIoUAssigner.assign` / pt_max_iou_assign, plus the empty-prediction case (:85-106)."""
    from point_teacher_amd.retina_baseline import MaxIoUAssigner
    dev = torch.device('cuda:0')
    a = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5)
    b, g = torch.FloatTensor(KNOWN_BOXES).to(dev), torch.FloatTensor(KNOWN_GTS).to(dev)
    assert a.assign(b, g)[0].tolist() == [1, 0, 2, 0]
    assert a.assign(b, torch.empty(0, 4, device=dev))[0].tolist() == [0, 0, 0, 0]
    inds, mo = a.assign(torch.empty(0, 4, device=dev), g)
    assert len(inds) == 0 and len(mo) == 0


def test_anchor_generator_reference_known_answers():
    """The reference's own `test_retina_anchor` (HBB_TOD/tests/test_utils/test_anchor.py:455-552): base anchors of the
    RetinaNet generator, valid-flag counts at 640x640, anchors per location."""
    from point_teacher_amd.retina_baseline import AnchorGenerator
    gen = AnchorGenerator(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128])
    level0 = torch.Tensor([[-22.6274, -11.3137, 22.6274, 11.3137], [-28.5088, -14.2544, 28.5088, 14.2544],
                           [-35.9188, -17.9594, 35.9188, 17.9594], [-16.0000, -16.0000, 16.0000, 16.0000],
                           [-20.1587, -20.1587, 20.1587, 20.1587], [-25.3984, -25.3984, 25.3984, 25.3984],
                           [-11.3137, -22.6274, 11.3137, 22.6274], [-14.2544, -28.5088, 14.2544, 28.5088],
                           [-17.9594, -35.9188, 17.9594, 35.9188]])
    for i, base in enumerate(gen.base_anchors):                     # every level is level 0 scaled by 2^i (the reference lists all five)
        assert base.allclose(level0 * 2 ** i, rtol=1e-5, atol=1e-3)
    assert gen.base_anchors[4][2].allclose(torch.Tensor([-574.7006, -287.3503, 574.7006, 287.3503]), atol=1e-3)
    sizes = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
    flags = gen.valid_flags(sizes, (640, 640), 'cpu')
    assert [int(f.sum()) for f in flags] == [57600, 14400, 3600, 900, 225]
    assert gen.num_base_anchors == [9, 9, 9, 9, 9] and len(gen.grid_anchors(sizes, 'cpu')) == 5
