"""YOLOF baseline (configs/baselines/aitodv2_yolof_r50_1x.py): the product `yolof_baseline.py` against
tests/golden/yolof_baseline.npz - outputs of the reference's OWN `YOLOFHead` / `UniformAssigner` / coder / losses
(oracle/gen_golden_yolof.py) - and as a trainable detector under the flat optimizer's three parameter groups
(`norm_decay_mult=0`, `custom_keys={'backbone': lr_mult 1/3}`)."""
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    import point_teacher_amd as pta
    return pta, pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'baselines', 'aitodv2_yolof_r50_1x.py'))


def test_config_mirror_and_parameter_groups():
    """The mirror equals the reference's dictionary wherever the reference tree is present; the optimizer's paramwise_cfg
    yields three contiguous groups: plain, backbone (lr x 1/3, decay x 1 - the custom key wins over norm_decay_mult) and the
    BatchNorm layers of encoder and head (decay x 0)."""
    pta, cfg = _cfg()
    ref_path = '/root/reference/HBB_TOD/configs/baselines/aitodv2_yolof_r50_1x.py'
    if os.path.exists(ref_path):
        ref = pta.Config.fromfile(ref_path).to_dict()
        mine = cfg.to_dict()
        for k in ('model', 'optimizer', 'lr_config', 'optimizer_config', 'data', 'runner'):
            assert mine.get(k) == ref.get(k), k
    assert cfg.optimizer['paramwise_cfg'] == dict(norm_decay_mult=0., custom_keys={'backbone': dict(lr_mult=1. / 3)})
    from point_teacher_amd.runtime import FlatParams
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = pta.build_detector(cfg.model)
    names = [n for n, _ in model.named_parameters()]
    assert 'neck.dilated_encoder_blocks.3.conv2.bn.weight' in names and 'bbox_head.bbox_subnet.3.conv.weight' in names
    assert 'bbox_head.object_pred.bias' in names and 'neck.lateral_norm.bias' in names
    assert not any('conv.bias' in n for n in names if 'subnet' in n or 'encoder_blocks' in n)       # bias='auto' in front of a norm
    flat = FlatParams(model, paramwise_cfg=cfg.optimizer['paramwise_cfg'])
    third = float(1. / 3)
    assert flat.group_mults == [(1., 1.), (third, 1.), (1., 0.)]
    group = {n: sum(flat.slices[n][0] >= e for e in flat.group_ends) for n, p in model.named_parameters() if p.requires_grad}
    assert all(g == 1 for n, g in group.items() if n.startswith('backbone.'))
    assert all(g == 2 for n, g in group.items() if ('.bn.' in n or '_norm.' in n) and not n.startswith('backbone.'))
    assert group['bbox_head.cls_score.weight'] == group['bbox_head.cls_score.bias'] == group['neck.fpn_conv.bias'] == 0


@pytest.mark.gpu
def test_head_vs_reference_golden():
    """Implicit objectness, centre-clamped decode, UniformAssigner (incl. its last-write-wins scatter), focal + DIoU loss and
    their gradients against the reference's own outputs."""
    from point_teacher_amd.yolof_baseline import YOLOFHead, UniformAssigner, delta2bbox_ctr_clamp
    G = load_golden('yolof_baseline')
    dev = torch.device('cuda:0')
    H, W = (int(v) for v in G['in_img_hw'])
    cls, obj, reg = (G.t(k).to(dev) for k in ('in_cls', 'in_obj', 'in_reg'))
    reg.requires_grad_(True)
    head = YOLOFHead(8, 16, train_cfg=dict(assigner=dict(type='UniformAssigner', pos_ignore_thr=0.15, neg_ignore_thr=0.7),
                                            allowed_border=-1, pos_weight=-1), test_cfg=dict()).to(dev)
    N, _, h, w = cls.shape
    norm = head.implicit_objectness(cls.view(N, 3, 8, h, w), obj)
    torch.testing.assert_close(norm.detach().cpu(), G.t('out_normalized_cls'), rtol=1e-5, atol=1e-5)
    norm = norm.detach().requires_grad_(True)            # the golden's gradients are taken at the normalised scores (gen_golden_yolof.py)
    anchors = head.anchor_generator.grid_anchors([(h, w)], dev)[0]
    torch.testing.assert_close(anchors.cpu(), G.t('out_anchors'), rtol=0, atol=0)
    assigner = UniformAssigner(0.15, 0.7)
    for b in range(2):
        bp = reg[b].detach().permute(1, 2, 0).reshape(-1, 4)
        dec = delta2bbox_ctr_clamp(anchors, bp, 32)
        torch.testing.assert_close(dec.cpu(), G.t(f'out_decoded{b}'), rtol=1e-5, atol=1e-4)
        r = assigner.assign(dec, anchors, G.t(f'in_gt_bboxes{b}').to(dev), None, G.t(f'in_gt_labels{b}').to(dev))
        assert torch.equal(r['gt_inds'].cpu(), G.t(f'out_assign{b}')), b
        assert torch.equal(r['labels'].cpu(), G.t(f'out_assign_labels{b}'))
        assert torch.equal(r['pos_idx'].cpu(), G.t(f'out_pos_idx{b}'))
        torch.testing.assert_close(r['pos_predicted_boxes'].cpu(), G.t(f'out_pos_pred{b}'), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(r['target_boxes'].cpu(), G.t(f'out_pos_target{b}'), rtol=0, atol=0)
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3)) for _ in range(2)]
    losses = head.loss([norm], [reg], [G.t(f'in_gt_bboxes{b}').to(dev) for b in range(2)], [G.t(f'in_gt_labels{b}').to(dev) for b in range(2)], metas)
    assert abs(float(losses['loss_cls']) - float(G['out_loss_cls'])) <= 1e-3 * abs(float(G['out_loss_cls']))
    assert abs(float(losses['loss_bbox']) - float(G['out_loss_bbox'])) <= 1e-3 * abs(float(G['out_loss_bbox']))
    (losses['loss_cls'] + losses['loss_bbox']).backward()
    torch.testing.assert_close(norm.grad.cpu(), G.t('out_grad_cls'), rtol=2e-3, atol=1e-6)
    torch.testing.assert_close(reg.grad.cpu(), G.t('out_grad_reg'), rtol=2e-3, atol=1e-6)
    # an image without boxes: every anchor a negative, no regression term, finite
    losses = head.loss([norm.detach()], [reg.detach()], [G.t('in_gt_bboxes0').to(dev), torch.zeros(0, 4, device=dev)],
                       [G.t('in_gt_labels0').to(dev), torch.zeros(0, dtype=torch.long, device=dev)], metas)
    assert torch.isfinite(losses['loss_cls']) and torch.isfinite(losses['loss_bbox'])


@pytest.mark.gpu
def test_yolof_trains_and_infers():
    """The config through the flat-parameter Trainer (three parameter groups -> pt_sgd_step_groups; BatchNorm of encoder / head in
    training mode): finite losses, the backbone moves 3x slower than it would at lr x 1, norm layers are not decayed, no host
    synchronisation in a steady-state step, inference returns per-class detections."""
    from point_teacher_amd.synthetic import SyntheticTiles
    pta, cfg = _cfg()
    dev = torch.device('cuda:0')
    torch.manual_seed(2)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = pta.build_detector(cfg.model).to(dev)
    model.train()
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
    assert trainer.flat.group_mults == [(1., 1.), (float(1. / 3), 1.), (1., 0.)] and trainer.flat.teacher_flat is None
    data = SyntheticTiles(n=4, size=512, mean_objects=40, seed=5, device=dev)
    rm0 = model.neck.lateral_norm.running_mean.clone()
    s0 = trainer.flat.student_flat.clone()
    losses = []
    for it in range(6):
        if it == 5:
            torch.cuda.synchronize()
            torch.cuda.set_sync_debug_mode('error')
        try:
            out = trainer.step(data.batch(it, 2))
        finally:
            torch.cuda.set_sync_debug_mode('default')
        lv = out['log_vars'].materialize()
        assert set(lv) == {'loss_cls', 'loss_bbox', 'loss'} and all(v == v and abs(v) != float('inf') for v in lv.values()), lv
        losses.append(lv['loss'])
    assert not torch.equal(model.neck.lateral_norm.running_mean, rm0)                       # BatchNorm ran in training mode
    f = trainer.flat
    moved = (f.student_flat - s0).abs()
    assert float(moved[:f.group_ends[0]].max()) > 0 and float(moved[f.group_ends[0]:f.group_ends[1]].max()) > 0
    assert float(moved[f.frozen_start:].max()) == 0 and f.check_views()
    model.eval()
    with torch.no_grad():
        res = model.simple_test(data.batch(0, 2)['img'], [dict(img_shape=(512, 512, 3), scale_factor=np.ones(4, np.float32))] * 2)
    assert len(res) == 2 and len(res[0]) == 8 and all(r.shape[1] == 5 for r in res[0])
