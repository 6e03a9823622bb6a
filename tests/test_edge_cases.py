"""Ragged / empty inputs through the whole iteration on the GPU (the reference filters empty images in the dataset,
`filter_empty_gt=True`, and divides by the first image's object count in mil_bag_extensive; here an image without
objects - or a whole batch without any - must simply contribute nothing): every configuration runs two iterations,
emits the full key set and only finite values."""
import pytest
import torch

import test_obb_parity as TO
import test_train_step_parity as T

pytestmark = pytest.mark.gpu

KEYS = {'coarse_bboxes_iou', 'stage0_refine_bboxes_iou', 'stage0_loss_mil_bbox', 'stage0_loss_mil_bags',
        'stage0_coarse_bags_iou', 'stage0_refine_bags_iou', 'refined_points_distance', 'loss_cls', 'loss_bbox',
        'loss_centerness', 'loss'}


@pytest.mark.parametrize('variant', ['hbb', 'obb'])
@pytest.mark.parametrize('phase2', [True, False])
@pytest.mark.parametrize('counts', [(1, 0), (0, 0)])
def test_images_without_objects(variant, phase2, counts):
    dev = torch.device('cuda:0')
    mod = T if variant == 'hbb' else TO
    pta, cfg, model = mod._build(dev, phase2=phase2)
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config)
    img, boxes, labels, metas = mod._data(dev) if variant == 'hbb' else mod._data()
    boxes = [b[:n] for b, n in zip(boxes, counts)]
    labels = [l[:n] for l, n in zip(labels, counts)]
    data = dict(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in boxes], gt_labels=[l.to(dev) for l in labels])
    for _ in range(2):
        lv = trainer.step(data)['log_vars'].materialize()
    assert set(lv) == KEYS, set(lv) ^ KEYS
    bad = [k for k, v in lv.items() if v != v or abs(v) == float('inf')]
    assert not bad, (bad, lv)
    assert torch.isfinite(trainer.flat.student_flat).all() and torch.isfinite(trainer.flat.teacher_flat).all()


def test_iteration_with_dcn_on_last_conv():
    """The latent `dcn_on_last_conv=True` option (anchor_free_head.py:101-102; no shipped config sets it): both phases
    train through the DCNv2 tower convs (pt_deform_*), offsets receive gradients, everything stays finite."""
    import os
    dev = torch.device('cuda:0')
    import point_teacher_amd as pta
    from point_teacher_amd.nn_modules import ModulatedDeformConv2dPack
    from point_teacher_amd.synthetic import benchmark_init_
    torch.manual_seed(3)
    cfg = pta.Config.fromfile(os.path.join(T.ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
    cfg.model['_model_']['bbox_head']['dcn_on_last_conv'] = True
    cfg.model['burn_in_step'] = 0                      # iteration 0: phase 1, iteration 1: phase 2
    model = pta.build_detector(cfg.model).to(dev)
    benchmark_init_(model, phase2=True)
    model.train()
    assert isinstance(model.student.bbox_head.reg_convs[3].conv, ModulatedDeformConv2dPack)
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config)
    img, boxes, labels, metas = T._data(dev)
    data = dict(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in boxes], gt_labels=[l.to(dev) for l in labels])
    w0 = model.student.bbox_head.reg_convs[3].conv.conv_offset.weight.detach().clone()
    for _ in range(2):
        lv = trainer.step(data)['log_vars'].materialize()
        assert all(v == v and abs(v) != float('inf') for v in lv.values()), lv
    w1 = model.student.bbox_head.reg_convs[3].conv.conv_offset.weight.detach()
    assert float((w1 - w0).abs().max()) > 0            # the offset predictor trains
    assert torch.isfinite(trainer.flat.student_flat).all()


@pytest.mark.gpu
def test_pinned_upload_ring_survives_a_busy_stream():
    """Index uploads (offsets, RoI batch ids) are asynchronous copies out of a pinned ring; the host runs ahead of the GPU, so a
    staged slot must outlive the copy that reads it.  With the stream kept busy, more words than the whole ring are uploaded
    in small and large pieces; every device tensor must hold what was uploaded (the 100 % configuration - 60 750 batch ids per
    RoIAlign call - wrapped the unguarded ring inside one iteration and faulted)."""
    import numpy as np
    from point_teacher_amd import functional as Fn
    dev = torch.device('cuda:0')
    a = torch.randn(8192, 8192, device=dev)
    rng = np.random.default_rng(0)
    for _ in range(12):                                   # ~0.1 s of queued GEMMs: the copies below wait behind them
        a = (a @ a) * 1e-4
    srcs, outs = [], []
    for i in range(400):
        n = [7, 301, 5000, 60750][i % 4]
        v = rng.integers(0, 1 << 20, n).astype(np.int32)
        srcs.append(v)
        outs.append(Fn.upload_i32(v, dev) if i % 3 else Fn.upload_f32(v.astype(np.float32), dev))
    torch.cuda.synchronize()
    assert sum(s.size for s in srcs) > 4 * Fn._ring.words
    for s, o in zip(srcs, outs):
        assert np.array_equal(o.cpu().numpy().astype(np.int64), s.astype(np.int64))
    assert float(a.abs().sum()) >= 0
    # host targets get their own memory (no alias of the ring)
    h1 = Fn.upload_i32([1, 2, 3], 'cpu'); h2 = Fn.upload_i32([4, 5, 6], 'cpu')
    assert h1.tolist() == [1, 2, 3] and h2.tolist() == [4, 5, 6]
