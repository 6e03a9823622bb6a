"""Whole-iteration parity on the GPU: the product's `TS_P2B_FCOS.train_step` (HIP kernels +
MIOpen convs) against the CPU oracle `oracle/ref_model.py` on the SAME weights, inputs and
injected random draws.  Every entry of the loss dict within 1e-3 relative (north_star),
gradients of representative parameters within 1e-3 of the gradient's norm, in BOTH phases.
(The oracle's whole iteration is itself pinned to the reference: tests/test_reference_iteration.py.)"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_model as M
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(dev, phase2):
    import point_teacher_amd as pta
    from point_teacher_amd.synthetic import benchmark_init_
    torch.manual_seed(3)
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
    cfg.model['burn_in_step'] = -1 if phase2 else 10 ** 9
    model = pta.build_detector(cfg.model).to(dev)
    benchmark_init_(model, phase2=True)
    with torch.no_grad():      # moderate bag logits, as in a trained head (oracle/ref_model.damp_mil_logits explains why)
        for det in (model.student, model.teacher):
            for m in list(det.bbox_head.fc_cls) + list(det.bbox_head.fc_ins):
                m.weight.mul_(0.1); m.bias.mul_(0.1)
    model.train()
    return pta, cfg, model


def _syn_draws(G, seed, n_prior=6):
    """Per-object draws of the step-1 rectangle generator (TS_P2B_FCOS._inject['syn'][i] / ref_ops.generate_black_paper)."""
    g = torch.Generator().manual_seed(seed)
    d = dict(cls=torch.randint(0, n_prior, (G,), generator=g))
    for n in ('scale', 'x', 'y', 'a', 'boost', 'itv', 'dev'):
        d[n] = torch.rand(G, generator=g)
    for n in ('wn', 'rn'):
        d[n] = torch.randn(G, generator=g)
    d['itv2'] = d['itv']
    return d


def _data(dev, size=256, n_obj=(23, 17), seed=5):
    g = torch.Generator().manual_seed(seed)
    img = (torch.rand(2, 3, size, size, generator=g) * 90 + 60).round()
    boxes, labels = [], []
    for n in n_obj:
        c = torch.rand(n, 2, generator=g) * (size - 48) + 24
        # keep centres off the symmetric tie positions of the stride-8 grid
        c = c + 0.37
        wh = torch.exp(torch.randn(n, 2, generator=g) * 0.4 + np.log(14.0)).clamp(4, 40)
        boxes.append(torch.cat([c - wh / 2, c + wh / 2], 1))
        labels.append(torch.randint(0, 8, (n,), generator=g))
    metas = [dict(ori_filename=f't{i}.png', img_shape=(size, size, 3), scale_factor=np.ones(4, np.float32)) for i in range(2)]
    return img, boxes, labels, metas


def _layout(model, img, layout):
    """`channels_last` = the training layout (Trainer(channels_last=True), bench.py): trunk, necks, towers and FC stacks run
    plane-native on pt_conv_bf16x6 / pt_conv_wgrad_bf16x6; `nchw`: the 1x1 convolutions stay with the library."""
    if layout == 'channels_last':
        return model.to(memory_format=torch.channels_last), img.contiguous(memory_format=torch.channels_last)
    return model, img


LAYOUTS = pytest.mark.parametrize('layout', ['nchw', 'channels_last'])


def _strip(sd, prefix):
    return {k[len(prefix):]: v.detach().cpu().clone() for k, v in sd.items() if k.startswith(prefix)}


def _check(losses_gpu, losses_ref, keys=None):
    keys = keys or losses_ref.keys()
    for k in keys:
        a, b = float(losses_gpu[k]), float(losses_ref[k])
        assert abs(a - b) <= 1e-3 * max(abs(b), 1e-2) + 1e-5, (k, a, b)


def _rel(a, b):
    """|a - b| / |b| over the whole tensor (an element-wise relative test is meaningless for entries that cancel to ~0)."""
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def _check_grads(model, ref_loss, params, tol=1e-3):
    gr = torch.autograd.grad(ref_loss, [params[k] for k in GRAD_KEYS])
    gs = dict(model.student.named_parameters())
    errs = {k: _rel(gs[k].grad.cpu(), gref) for k, gref in zip(GRAD_KEYS, gr)}
    print('gradient errors |g - g_ref| / |g_ref|:', {k: f'{v:.1e}' for k, v in errs.items()})
    for k, e in errs.items():
        assert e < tol, (k, e)


GRAD_KEYS = ['bbox_head.conv_cls.weight', 'bbox_head.conv_reg.weight', 'bbox_head.reg_convs.3.conv.weight',
             'bbox_head.fc_cls.0.weight', 'bbox_head.fc_ins.0.weight', 'bbox_head.fc_reg.0.weight',
             'bbox_head.shared_fcs_bag.0.1.weight', 'neck_agg.lateral_convs.4.conv.weight',
             'neck.fpn_convs.0.conv.weight', 'backbone.layer4.2.conv3.weight', 'backbone.layer2.0.conv1.weight']


@LAYOUTS
def test_step2_loss_dict_and_grads(layout):
    dev = torch.device('cuda:0')
    pta, cfg, model = _build(dev, phase2=True)
    img, boxes, labels, metas = _data(dev)
    g = torch.Generator().manual_seed(11)
    neg_u = torch.rand(2, 4, 200, generator=g)
    aug = (['horizontal', 'diagonal'], [0.9, 1.1])
    model._inject = dict(neg0=neg_u.to(dev), aug=aug)
    sd_s0 = _strip(model.state_dict(), 'student.')
    sd_t0 = _strip(model.state_dict(), 'teacher.')
    model, img_dev = _layout(model, img.to(dev), layout)
    data = dict(img=img_dev, img_metas=metas, gt_bboxes=[b.to(dev) for b in boxes], gt_labels=[l.to(dev) for l in labels])
    out = model.train_step(data, None)
    out['loss'].backward()
    lv = out['log_vars'].materialize()
    # ---- oracle on the same weights --------------------------------------------
    torch.set_num_threads(8)
    params = {k: (v.clone().requires_grad_(True) if M.trainable(k) else v) for k, v in sd_s0.items()}
    sd_t = M.ema(sd_t0, sd_s0)                                     # the EMA runs first (:126)
    gp = [R.bbox_xyxy_to_cxcywh(b)[:, :2] for b in boxes]          # _point_ = 0.0 -> box centres
    cfgm = dict(M.MODEL_CFG)
    ref, _ = M.forward_train_step2(params, sd_t, img, boxes, labels, gp, cfgm, dict(neg0=neg_u, aug=aug))
    ref['loss'] = M.total_loss(ref)
    assert set(ref.keys()) == set(lv.keys()), set(ref.keys()) ^ set(lv.keys())
    _check(lv, ref)
    # teacher after EMA
    t_gpu = _strip(model.state_dict(), 'teacher.')
    for k in ('bbox_head.conv_cls.weight', 'backbone.layer3.1.conv2.weight', 'backbone.bn1.weight'):
        torch.testing.assert_close(t_gpu[k], sd_t[k], rtol=1e-6, atol=1e-7)
    _check_grads(model, ref['loss'], params)
    for n, p in model.teacher.named_parameters():
        assert p.grad is None, n                                     # the teacher never receives gradients


@LAYOUTS
def test_step1_loss_dict_and_grads(layout):
    """Burn-in step 1.  Product and oracle each run their OWN rectangle generator on the same injected draws
    (nothing the HIP path produced is handed to the oracle); loss dict 1e-3, gradients 1e-3 of their norm."""
    dev = torch.device('cuda:0')
    pta, cfg, model = _build(dev, phase2=False)
    img, boxes, labels, metas = _data(dev, seed=6)
    g = torch.Generator().manual_seed(12)
    neg_u = torch.rand(2, 4, 200, generator=g)
    aug = (['vertical', 'None'], [1.2, 0.8])
    syn = [_syn_draws(b.shape[0], 40 + i) for i, b in enumerate(boxes)]
    model._inject = dict(neg0=neg_u.to(dev), aug=aug, syn=[{k: v.to(dev) for k, v in d.items()} for d in syn])
    sd_s0 = _strip(model.state_dict(), 'student.')
    sd_t0 = _strip(model.state_dict(), 'teacher.')
    model, img_dev = _layout(model, img.to(dev), layout)
    data = dict(img=img_dev, img_metas=metas, gt_bboxes=[b.to(dev) for b in boxes], gt_labels=[l.to(dev) for l in labels])
    captured = {}
    orig = model.genrate_syn

    def spy(*a, **k):
        r = orig(*a, **k)
        captured['img_syn'], captured['boxes'], captured['alive'] = r[0], r[2], r[3]
        return r
    model.genrate_syn = spy
    out = model.train_step(data, None)
    out['loss'].backward()
    lv = out['log_vars'].materialize()
    params = {k: (v.clone().requires_grad_(True) if M.trainable(k) else v) for k, v in sd_s0.items()}
    sd_t = M.ema(sd_t0, sd_s0)
    gp = [R.bbox_xyxy_to_cxcywh(b)[:, :2] for b in boxes]
    ref, _ = M.forward_train_step1(params, sd_t, img, boxes, labels, gp, dict(M.MODEL_CFG), dict(neg0=neg_u, aug=aug, syn=syn))
    ref['loss'] = M.total_loss(ref)
    # the generator itself: same survivors in the same order, same hulls, same painted pixels
    prior = torch.tensor(M.SHAPE_LIST)
    for i in range(2):
        img_syn, hull, _, _ = R.generate_black_paper(img[i], boxes[i], prior, syn[i])
        assert hull.shape[0] > 0
        torch.testing.assert_close(captured['boxes'][i][captured['alive'][i]].cpu(), hull, rtol=1e-5, atol=1e-3)
        assert torch.equal(captured['img_syn'][i].cpu(), img_syn)
    assert set(ref.keys()) == set(lv.keys()), set(ref.keys()) ^ set(lv.keys())
    _check(lv, ref)
    _check_grads(model, ref['loss'], params)


def test_teacher_bn_affine_follows_ema():
    """The flat EMA kernel rewrites the teacher's BatchNorm weight/bias through raw pointers; the cached
    (scale, shift) of the fused BN epilogue must follow (nn_modules.refresh_bn_affines)."""
    dev = torch.device('cuda:0')
    pta, cfg, model = _build(dev, phase2=True)
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config)
    img = _data(dev)[0].to(dev)
    with torch.no_grad():
        before = model.extract_feat(img, model.teacher)[0].clone()               # fills the caches
        for m in model.student.backbone.modules():                               # a student whose BN differs from the teacher's
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.fill_(1.04)
                m.bias.fill_(0.01)
        for _ in range(3):
            model.update_teacher_model(model.teacher, model.student, 0.5)
        bn = model.teacher.backbone.layer3[1].bn2
        torch.testing.assert_close(bn.weight, torch.full_like(bn.weight, 1 + 0.04 * (1 - 0.5 ** 3)))
        fused = model.extract_feat(img, model.teacher)[0]
        for m in model.teacher.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.fuse_epilogue = False
        plain = model.extract_feat(img, model.teacher)[0]
    scale = float(plain.abs().max())
    assert float((fused - plain).abs().max()) < 1e-3 * scale          # the fused epilogue saw the EMA-updated affine
    assert float((before - plain).abs().max()) > 0.05 * scale         # ... which really changed the features
    del trainer


def test_full_size_iteration_properties():
    """BASELINE size (bs 2, 800x800, ~300 points/image, 88 M parameters): properties that do not need the oracle.
    (1) the teacher after the flat EMA kernel is exactly alpha*t + (1-alpha)*s of the previous state (fp32, same
    rounding as torch's mul_/add_); (2) the flat gradient the optimizer consumes equals the per-parameter autograd
    gradients of the same iteration (gather path) and unused parameters stay zero; (3) clip + SGD moves every
    trainable parameter by -lr*(clip*g + wd*w) on the first step and never touches frozen ones; (4) nothing in the
    iteration synchronises the host (checked with torch.cuda.set_sync_debug_mode)."""
    from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_
    dev = torch.device('cuda:0')
    import point_teacher_amd as pta
    torch.manual_seed(5)
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
    cfg.model['burn_in_step'] = -1
    model = pta.build_detector(cfg.model).to(dev)
    benchmark_init_(model, phase2=True)
    model.train()
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
    data = SyntheticTiles(n=2, size=800, mean_objects=300, seed=1, device=dev)
    f = trainer.flat
    batch = data.batch(0, 2)
    before = {n: p.detach().clone() for n, p in model.student.named_parameters()}
    trainer.step(batch)                 # warm-up iteration (allocator, MIOpen); the first step also settles the dead segment
    dead = sorted(f.dead)
    assert trainer.dead_known and len(dead) == 10 and all(n.split('.')[1] in ('shared_fcs', 'shared_fcs_refine', 'fc_iou') for n in dead)
    assert f.n_dead * 4 > 110e6 and f.grad_flat.numel() == f.mom_flat.numel() == f.n_train and f.check_views()
    for n in dead:                      # (0) round-2 verdict P3: never-used parameters are not decayed (torch.optim.SGD skips grad None)
        assert torch.equal(dict(model.student.named_parameters())[n], before[n]), n
    f.mom_flat.zero_()                  # so that the next step is a "first" step again: buf = 0.9*0 + d
    t0, s0 = f.teacher_flat.clone(), f.student_flat.clone()
    trainer.step(batch)
    t1, s1 = f.teacher_flat.clone(), f.student_flat.clone()
    torch.testing.assert_close(t1, t0 * 0.999 + s0 * (1 - 0.999), rtol=1e-6, atol=1e-8)        # (1) EMA ran BEFORE the update
    g = f.grad_flat.clone()
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    # (3) first SGD step: p1 = p0 - lr_g * (coef*g + wd_g*p0), bias group lr x2 / wd x0 (configs :212-215)
    lr = trainer.sched.lr_at(0)
    coef = min(1.0, 35.0 / (float(g.double().norm()) + 1e-6))
    nw = f.n_weights
    exp_w = s0[:nw] - lr * (coef * g[:nw] + 1e-4 * s0[:nw])
    exp_b = s0[nw:f.n_train] - 2 * lr * (coef * g[nw:f.n_train])
    torch.testing.assert_close(s1[:nw], exp_w, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(s1[nw:f.n_train], exp_b, rtol=1e-4, atol=1e-7)
    assert torch.equal(s1[f.n_train:], s0[f.n_train:])                                         # dead and frozen segments untouched
    assert f.frozen_start == f.n_train + f.n_dead
    # (2) gather path == autograd: rerun the same iteration by hand on the updated weights
    model._inject = dict(neg0=torch.rand(2, 4, 200, device=dev), aug=(['None', 'horizontal'], [1.0, 0.9]))
    f.zero_grad(); f.detach_grads()
    model.ema_alpha = 1.0
    out = model.train_step(dict(batch, img=batch['img'].contiguous(memory_format=torch.channels_last)), None)
    out['loss'].backward()
    per_param = {n: (p.grad.clone() if p.grad is not None else None) for n, p in f.order if p.requires_grad}
    f.gather_grads()
    unused = 0
    for n, p in f.order:
        if not p.requires_grad or n in f.dead:
            continue
        off, k = f.slices[n]
        seg = f.grad_flat[off:off + k]
        if per_param[n] is None:
            unused += 1
            assert float(seg.abs().max()) == 0, n
        else:
            ref = per_param[n].permute(0, 2, 3, 1).reshape(-1) if per_param[n].dim() == 4 else per_param[n].reshape(-1)
            assert torch.equal(seg, ref), n
    assert unused == 0 and f.check_views()            # shared_fcs / shared_fcs_refine / fc_iou are in the dead segment: every live parameter received a gradient
    assert all(dict(model.student.named_parameters())[n].grad is None for n in dead)
    # (4) no host synchronisation inside a steady-state iteration
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        model._inject = {}
        trainer.step(data.batch(1, 2))
    finally:
        torch.cuda.set_sync_debug_mode('default')


@pytest.mark.parametrize('percent', [30, 60, 100])
def test_full_size_noisy_point_configs(percent):
    """aitodv2_point_teacher_{30,100}% at BASELINE size (bs 2, 800x800, ~300 points/image): `_point_` > 0 (random initial
    points), lamda 0.5, U1 = 9 coarse boxes x U2 = 45 shaken boxes per object -> K = 2 x 75 x 405 = 60 750 RoIs per
    RoIAlign call, a 3.05 GB [K,256,7,7] block per branch (SURVEY section 7 "Memory at the 100 % config"; it simply
    fits - 288 GB of HBM - so nothing is chunked).  Both phases: finite loss dict with the full key set, the RoI count
    the reference would produce, points that move between visits (lamda 0.5), no host synchronisation."""
    from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_
    import point_teacher_amd as pta
    dev = torch.device('cuda:0')
    torch.manual_seed(7)
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', f'aitodv2_point_teacher_{percent}.py'))
    assert cfg.model['lamda'] == 0.5 and cfg.model['_point_'] == percent / 100 and cfg.model['num_training_burninstep2'] == 75
    cfg.model['burn_in_step'] = 1                                   # iterations 0, 1 = phase 1; 2, 3 = phase 2
    model = pta.build_detector(cfg.model).to(dev)
    benchmark_init_(model, phase2=True)
    model.train()
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
    data = SyntheticTiles(n=2, size=800, mean_objects=300, seed=2, device=dev)
    ks = []
    ext = model.student.bbox_head.bbox_roi_extractor
    # (RoI blocks leave as fp32 [K, C, 7, 7] or - into the FC stacks on the plane kernels - as planes.PlaneAct with the K RoIs as rows)
    hook = ext.register_forward_hook(lambda m, i, o: ks.append(int(o.H if type(o).__name__ == 'PlaneAct' else o.shape[0])))
    torch.cuda.reset_peak_memory_stats()
    keys = None
    pts_seen = []
    for it in range(4):
        if it == 3:
            torch.cuda.synchronize()
            torch.cuda.set_sync_debug_mode('error')                  # a steady-state phase-2 iteration may not touch the host
        try:
            out = trainer.step(data.batch(0, 2))                     # the SAME two tiles every time: the point dictionaries are reused
        finally:
            torch.cuda.set_sync_debug_mode('default')
        lv = out['log_vars'].materialize()
        assert all(v == v and abs(v) != float('inf') for v in lv.values()), (it, lv)
        keys = set(lv) if keys is None else keys
        assert set(lv) == keys
        pts_seen.append(torch.cat([v.clone() for _, v in sorted(model.refined_gt_bboxes_point.items())]))
    hook.remove()
    assert {'stage0_loss_mil_bbox', 'stage0_loss_mil_bags', 'loss_cls', 'loss_bbox', 'loss_centerness', 'refined_points_distance'} <= keys
    U = 9 * 45
    # every real-bag RoIAlign call sees the whole 60 750-box batch; the bag classifier's call also carries the 2 x 200 negative
    # proposals (round 5: they ride through the same RoIAlign / FC launches instead of a pass of their own)
    assert max(ks) == 2 * 75 * U + 400 and 2 * 75 * U in ks, (sorted(set(ks)), 2 * 75 * U)
    assert model.count == 4 and len(model.gt_bboxes_point) == 2
    first = torch.cat([v for _, v in sorted(model.gt_bboxes_point.items())])
    assert float((pts_seen[-1] - first).abs().max()) > 0.1           # lamda 0.5: refined points left the first-visit points ...
    assert float((pts_seen[-1] - pts_seen[-2]).abs().max()) > 1e-3  # ... and keep following the teacher between visits
    print(f'{percent} %: peak HBM {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB, RoIs per call {sorted(set(ks))}')


def test_runner_checkpoint_resume(tmp_path):
    """runner.Runner: log lines, checkpoint and a resume that continues in the same phase with the same counters,
    point dictionaries, momentum and weights (the reference loses `count` and the dictionaries)."""
    import json
    from point_teacher_amd.runner import Runner
    from point_teacher_amd.synthetic import SyntheticTiles
    dev = torch.device('cuda:0')
    pta, cfg, model = _build(dev, phase2=True)
    model.burn_in_step = 2                                   # iterations 0-2 are phase 1, then phase 2
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, iters_per_epoch=3)
    data = SyntheticTiles(n=4, size=256, mean_objects=20, seed=3, device=dev)
    runner = Runner(trainer, data.batch, str(tmp_path), max_epochs=2, iters_per_epoch=3, log_interval=1)
    assert runner.run(max_iters=3) == 3
    assert runner.epoch == 1 and os.path.exists(tmp_path / 'epoch_1.pth') and os.path.islink(tmp_path / 'latest.pth')
    lines = [json.loads(l) for l in open(tmp_path / 'log.json')]
    assert len(lines) == 3 and lines[0]['epoch'] == 1 and 'loss_cls' in lines[0] and lines[0]['lr'] > 0
    # a fresh process: new model, resume, and the state is the saved one
    pta2, cfg2, model2 = _build(dev, phase2=True)
    trainer2 = pta2.Trainer(model2, cfg2.optimizer, cfg2.optimizer_config, cfg2.lr_config, iters_per_epoch=3)
    runner2 = Runner(trainer2, data.batch, str(tmp_path), max_epochs=2, iters_per_epoch=3, log_interval=1)
    meta = runner2.resume(str(tmp_path / 'latest.pth'))
    assert meta['epoch'] == 1 and trainer2.iter == 3 and model2.count == 3 and runner2.epoch == 1
    assert set(model2.gt_bboxes_point) == set(model.gt_bboxes_point) and len(model2.gt_bboxes_point) == 4
    torch.testing.assert_close(trainer2.flat.student_flat, trainer.flat.student_flat, rtol=0, atol=0)
    torch.testing.assert_close(trainer2.flat.teacher_flat, trainer.flat.teacher_flat, rtol=0, atol=0)
    torch.testing.assert_close(trainer2.flat.mom_flat, trainer.flat.mom_flat, rtol=0, atol=0)
    assert trainer2.flat.check_views()
    assert runner2.run(max_iters=1) == 1 and model2.count == 4              # first phase-2 iteration after the resume


def test_eval_path_detections():
    """simple_test (fcos_p2b_teacher_student.py:276-298 -> get_bboxes :796-1005 -> multiclass_nms): the
    teacher's detections on the GPU equal a restatement from the oracle pieces (sigmoid scores x
    centerness, top-k, distance2bbox with clipping, class-aware greedy NMS)."""
    dev = torch.device('cuda:0')
    pta, cfg, model = _build(dev, phase2=True)
    with torch.no_grad():          # make some logits confident so that detections exist
        model.teacher.bbox_head.conv_cls.bias.fill_(-1.0)
    img, boxes, labels, metas = _data(dev, seed=9)
    model.eval()
    with torch.no_grad():
        res = model.simple_test(img.to(dev), metas, rescale=False)
    assert len(res) == 2 and len(res[0]) == 8 and all(r.shape[1] == 5 for r in res[0])
    sd_t = _strip(model.state_dict(), 'teacher.')
    with torch.no_grad():
        cls, reg, ctr = M.head_forward(sd_t, M.extract_feat(sd_t, img))
    pts = M.grid_points(img.shape[2] // 8, img.shape[3] // 8)
    tc = cfg.model['test_cfg']
    for b in range(2):
        sc = cls[b].permute(1, 2, 0).reshape(-1, 8).sigmoid()
        ct = ctr[b].permute(1, 2, 0).reshape(-1).sigmoid()
        bb = R.distance2bbox(pts, reg[b].permute(1, 2, 0).reshape(-1, 4), max_shape=metas[b]['img_shape'])
        per_cls = []
        for c in range(8):
            m = sc[:, c] > tc['score_thr']
            s = (sc[:, c] * ct)[m]
            keep = R.nms(bb[m], s, tc['nms']['iou_threshold'])
            per_cls.append((s[keep], bb[m][keep], torch.full((len(keep),), c)))
        all_s = torch.cat([p[0] for p in per_cls]); all_b = torch.cat([p[1] for p in per_cls])
        all_c = torch.cat([p[2] for p in per_cls])
        order = torch.argsort(all_s, descending=True, stable=True)[:tc['max_per_img']]   # bbox_nms.py:83-85
        all_s, all_b, all_c = all_s[order], all_b[order], all_c[order]
        assert all_s.numel() > 0
        for c in range(8):
            got = torch.from_numpy(res[b][c])
            rs, rb = all_s[all_c == c], all_b[all_c == c]
            assert abs(got.shape[0] - rs.numel()) <= 2, (b, c, got.shape, rs.numel())
            if got.shape[0] and rs.numel():
                # detections as a SET: a greedy NMS decision within rounding of the IoU threshold (MIOpen's split-K convolutions
                # accumulate with atomics, the scores move in the 7th digit from run to run) swaps single detections, so every
                # detection must have a partner (box within 0.05 px, score within 2e-3) except for a handful
                d = (got[:, None, :4] - rb[None, :, :].float()).abs().amax(-1)              # [n_got, n_ref]
                j = d.argmin(1)
                ok = (d[torch.arange(len(j)), j] <= 5e-2 + 1e-3 * rb[j].abs().amax(-1)) & \
                     ((got[:, 4] - rs[j]).abs() <= 2e-3 * rs[j] + 2e-4)
                assert int((~ok).sum()) <= max(2, got.shape[0] // 100), (b, c, int((~ok).sum()), got.shape[0])
                assert j[ok].unique().numel() >= int(ok.sum()) - 2                            # partners are distinct


@pytest.mark.parametrize('phase2', [False, True])
def test_bf16_backbone_fp32_head(phase2):
    """BASELINE configs[2] ("bf16 backbone + fp32 head"): with Trainer(autocast_dtype=bfloat16) the backbone / FPN / PSAGG
    convolutions run in bf16 while the dense head, the MIL head and the losses see and produce fp32.
    Checked against the ORACLE run with the same precision boundary (ref_model.bf16_backbone: bf16 operands and storage,
    fp32 accumulation, one rounding per fused epilogue): every entry of the loss dict within BF16_TOL.  bf16 keeps 8
    significant bits (one rounding = 4e-3 relative); product and oracle round the same quantities, but wherever two fp32
    accumulation orders straddle a rounding boundary one of them lands an ulp (0.4 - 0.8 %) away, the bias-carrying
    FPN / PSAGG convolutions round twice in the product (convolution, then bias add), and MIOpen's bf16 convolutions are
    not run-to-run reproducible on this stack: the same input twice moves the PSAGG features by 1.1 % (norm-wise;
    profiles/r02/bf16_accuracy.txt), the same size as the whole fp32 -> bf16 effect (1.6 %).  Hence BF16_TOL = 2 %
    (measured: <= 0.45 % on every key in both phases); the test also shows that it discriminates: summed over the keys the
    product is closer to the bf16 oracle than the fp32 oracle is."""
    BF16_TOL = 2e-2
    dev = torch.device('cuda:0')
    seen = {}
    pta, cfg, model = _build(dev, phase2=phase2)
    tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, autocast_dtype=torch.bfloat16)
    def trunk(m, i, o):            # plane-native under autocast: ONE bf16 plane per activation (planes.PlaneAct.np == 1)
        seen['backbone'] = torch.bfloat16 if (type(o[-1]).__name__ == 'PlaneAct' and o[-1].np == 1) else getattr(o[-1], 'dtype', None)
    hooks = [model.student.backbone.register_forward_hook(trunk),
             model.student.neck.fpn_convs[0].register_forward_hook(lambda m, i, o: seen.__setitem__('neck', o.dtype)),
             model.student.bbox_head.cls_convs[0].register_forward_hook(lambda m, i, o: seen.__setitem__('head', (i[0].dtype, o.dtype))),
             model.student.bbox_head.conv_reg.register_forward_hook(lambda m, i, o: seen.__setitem__('reg', o.dtype))]
    img, boxes, labels, metas = _data(dev, seed=6 if not phase2 else 5)
    g = torch.Generator().manual_seed(13)
    neg_u = torch.rand(2, 4, 200, generator=g)
    aug = (['horizontal', 'vertical'], [0.9, 1.1])
    inj = dict(neg0=neg_u, aug=aug)
    if not phase2:
        inj['syn'] = [_syn_draws(b.shape[0], 40 + i) for i, b in enumerate(boxes)]
    model._inject = {k: (v.to(dev) if torch.is_tensor(v) else ([{n: t.to(dev) for n, t in d.items()} for d in v] if k == 'syn' else v))
                     for k, v in inj.items()}
    sd_s0 = _strip(model.state_dict(), 'student.')
    sd_t0 = _strip(model.state_dict(), 'teacher.')
    data = dict(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in boxes], gt_labels=[l.to(dev) for l in labels])
    out = model.train_step(data, None)
    out['loss'].backward()
    lv = out['log_vars'].materialize()
    for h in hooks:
        h.remove()
    assert seen['backbone'] == torch.bfloat16 and seen['neck'] == torch.bfloat16
    assert seen['head'] == (torch.float32, torch.float32) and seen['reg'] == torch.float32
    assert all(v == v and abs(v) != float('inf') for v in lv.values()), lv
    gs = dict(model.student.named_parameters())
    assert all(torch.isfinite(gs[k].grad).all() for k in GRAD_KEYS)
    step = M.forward_train_step2 if phase2 else M.forward_train_step1
    sd_t = M.ema(sd_t0, sd_s0)
    gp = [R.bbox_xyxy_to_cxcywh(b)[:, :2] for b in boxes]
    with torch.no_grad():
        with M.bf16_backbone():
            ref16, _ = step(dict(sd_s0), sd_t, img, boxes, labels, gp, dict(M.MODEL_CFG), inj)
        ref32, _ = step(dict(sd_s0), sd_t, img, boxes, labels, gp, dict(M.MODEL_CFG), inj)
    assert set(ref16.keys()) == set(lv.keys()) - {'loss'}
    err = {k: abs(float(lv[k]) - float(ref16[k])) / max(abs(float(ref16[k])), 1e-2) for k in ref16}
    gap = {k: abs(float(ref32[k]) - float(ref16[k])) / max(abs(float(ref16[k])), 1e-2) for k in ref16}
    print('bf16 product vs bf16 oracle:', {k: f'{v:.1e}' for k, v in err.items()})
    print('fp32 oracle  vs bf16 oracle:', {k: f'{v:.1e}' for k, v in gap.items()})
    for k, e in err.items():
        # the refined boxes' IoU follows a top-k SELECTION of bag members: three runs of this test gave 1.1 / 1.3 / 2.2 % on it in phase 1
        # (the frozen stem's library bf16 kernels are not run-to-run reproducible: "same input twice" below moves the features by 1 %)
        assert e < (4e-2 if 'refine' in k else BF16_TOL), (k, e, float(lv[k]), float(ref16[k]))
    with torch.no_grad():                                   # the vendor kernels' own reproducibility, for the record
        x = img.to(dev).contiguous(memory_format=torch.channels_last)
        f1 = model.teacher.extract_feat(x)[0].float()
        f2 = model.teacher.extract_feat(x)[0].float()
    print('bf16 features, same input twice: relative difference %.2e' % float((f1 - f2).norm() / f1.norm()))
    assert tr.channels_last
    if phase2:      # (phase 1: the library stem's 1 % run-to-run noise is the size of the whole bf16 effect - no margin to discriminate)
        assert sum(err.values()) < sum(gap.values())


def test_full_size_bf16_backbone_vs_bf16_oracle():
    """BASELINE configs[2] at its real image size (800 x 800, bs 2, MIL on) against the ORACLE with the same precision boundary
    (ref_model.bf16_backbone), forward pass: every entry of the loss dict within 2 % (round-3 verdict item 5).  ~120 points per
    image keep the CPU oracle's MIL head (25 RoIs per point through three 12544 -> 1024 FC stacks) within a minute."""
    BF16_TOL = 2e-2
    dev = torch.device('cuda:0')
    pta, cfg, model = _build(dev, phase2=True)
    tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, autocast_dtype=torch.bfloat16, channels_last=True)
    seen = {}
    h = model.student.backbone.register_forward_hook(
        lambda m, i, o: seen.__setitem__('trunk', (type(o[-1]).__name__, getattr(o[-1], 'np', None))))
    img, boxes, labels, metas = _data(dev, size=800, n_obj=(127, 113), seed=9)
    g = torch.Generator().manual_seed(21)
    neg_u = torch.rand(2, 4, 200, generator=g)
    inj = dict(neg0=neg_u, aug=(['horizontal', 'None'], [0.9, 1.1]))
    model._inject = dict(neg0=neg_u.to(dev), aug=inj['aug'])
    sd_s0 = _strip(model.state_dict(), 'student.')
    sd_t0 = _strip(model.state_dict(), 'teacher.')
    data = dict(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in boxes], gt_labels=[l.to(dev) for l in labels])
    with torch.no_grad():
        out = model.train_step(data, None)
    lv = out['log_vars'].materialize()
    h.remove()
    assert seen['trunk'] == ('PlaneAct', 1)                    # the one-plane (bf16 operand) kernels ran the trainable stages
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    sd_t = M.ema(sd_t0, sd_s0)
    gp = [R.bbox_xyxy_to_cxcywh(b)[:, :2] for b in boxes]
    with torch.no_grad():
        with M.bf16_backbone():
            ref16, _ = M.forward_train_step2(dict(sd_s0), sd_t, img, boxes, labels, gp, dict(M.MODEL_CFG), inj)
    assert set(ref16.keys()) == set(lv.keys()) - {'loss'}
    err = {k: abs(float(lv[k]) - float(ref16[k])) / max(abs(float(ref16[k])), 1e-2) for k in ref16}
    print('800 x 800 bf16 product vs bf16 oracle:', {k: f'{v:.1e}' for k, v in err.items()})
    for k, e in err.items():
        assert e < BF16_TOL, (k, e, float(lv[k]), float(ref16[k]))
    del tr


@pytest.mark.parametrize('phase2', [False, True])
def test_shared_frozen_stem(phase2, monkeypatch):
    """Teacher and student hold the same frozen stem (both load the same pretrained backbone; conv1 / bn1 / layer1 never train,
    BatchNorm never updates): TS_P2B_FCOS evaluates the stem of the clean images once and skips the (identity) EMA of the frozen
    segment.  Same loss dict and gradients as the two-stem path (PT_SHARE_STEM=0) on the same weights, inputs and draws; a
    teacher whose frozen weights differ is detected and keeps its own stem."""
    dev = torch.device('cuda:0')
    img, boxes, labels, metas = _data(dev, seed=6 if not phase2 else 5)
    g = torch.Generator().manual_seed(13)
    inj = dict(neg0=torch.rand(2, 4, 200, generator=g).to(dev), aug=(['horizontal', 'vertical'], [0.9, 1.1]))
    if not phase2:
        inj['syn'] = [{n: t.to(dev) for n, t in _syn_draws(b.shape[0], 40 + i).items()} for i, b in enumerate(boxes)]
    data = dict(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in boxes], gt_labels=[l.to(dev) for l in labels])

    def run(share, perturb=False):
        monkeypatch.setenv('PT_SHARE_STEM', '1' if share else '0')
        pta, cfg, model = _build(dev, phase2=phase2)
        if perturb:
            with torch.no_grad():
                model.teacher.backbone.layer1[0].conv1.weight.mul_(1.001)
        tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
        model._inject = dict(inj)
        calls = []
        h = model.teacher.backbone.conv1.register_forward_hook(lambda m, i, o: calls.append(1))
        tr.flat.zero_grad(); tr.flat.detach_grads()
        out = model.train_step(dict(data, img=data['img'].contiguous(memory_format=torch.channels_last)), None)
        out['loss'].backward()
        h.remove()
        lv = out['log_vars'].materialize()
        return model._stem_shared, len(calls), lv, tr.flat.grad_flat.clone(), tr.flat.teacher_flat.clone(), tr.flat

    shared, calls, lv, grad, teacher, flat = run(True)
    assert shared is True and calls == 0                              # the teacher never ran its own stem
    shared0, calls0, lv0, grad0, teacher0, _ = run(False)
    assert shared0 is False and calls0 == 1
    for k in lv0:
        # (MIOpen's split-K convolutions accumulate with atomics: two runs of the SAME path already differ in the 6th digit)
        assert abs(lv[k] - lv0[k]) <= 1e-4 * max(abs(lv0[k]), 1e-2), (k, lv[k], lv0[k])
    assert _rel(grad, grad0) < 1e-3
    n = flat.n_train
    torch.testing.assert_close(teacher[:n], teacher0[:n], rtol=0, atol=0)           # trainable segment: the same EMA
    assert torch.equal(teacher[n:], flat.student_flat[n:])                         # frozen segment: still the student's, bit for bit
    torch.testing.assert_close(teacher0[n:], teacher[n:], rtol=1e-6, atol=1e-9)     # (the EMA of equal values only rounds)
    shared2, calls2, _, _, _, _ = run(True, perturb=True)
    assert shared2 is False and calls2 == 1


@LAYOUTS
@pytest.mark.parametrize('phase2', [False, True])
def test_full_size_vs_oracle(phase2, layout):
    """BASELINE size - bs 2, 800x800, 300 / 280 point annotations - against the CPU oracle on the same weights, inputs and draws:
    every entry of the loss dict within 1e-3 and the gradients of the representative parameters within 3e-3 of their norm (the
    conditioning of fp32 at this size, see tests/test_reference_iteration.py), in both phases; in phase 1 also the rectangle
    generator's survivors, hulls and painted pixels.  (The 256x256 tests above pin the same things at the size the reference's
    own iteration was recorded at.)"""
    dev = torch.device('cuda:0')
    torch.set_num_threads(max(torch.get_num_threads(), min(16, os.cpu_count() or 8)))
    pta, cfg, model = _build(dev, phase2=phase2)
    img, boxes, labels, metas = _data(dev, size=800, n_obj=(300, 280), seed=21)
    g = torch.Generator().manual_seed(14)
    neg_u = torch.rand(2, 4, 200, generator=g)
    aug = (['diagonal', 'horizontal'], [0.9, 1.1])
    inj = dict(neg0=neg_u, aug=aug)
    if not phase2:
        inj['syn'] = [_syn_draws(b.shape[0], 70 + i) for i, b in enumerate(boxes)]
    model._inject = {k: (v.to(dev) if torch.is_tensor(v) else ([{n: t.to(dev) for n, t in d.items()} for d in v] if k == 'syn' else v))
                     for k, v in inj.items()}
    sd_s0 = _strip(model.state_dict(), 'student.')
    sd_t0 = _strip(model.state_dict(), 'teacher.')
    model, img_dev = _layout(model, img.to(dev), layout)
    data = dict(img=img_dev, img_metas=metas, gt_bboxes=[b.to(dev) for b in boxes], gt_labels=[l.to(dev) for l in labels])
    captured = {}
    orig = model.genrate_syn

    def spy(*a, **k):
        r = orig(*a, **k)
        captured['img_syn'], captured['boxes'], captured['alive'] = r[0], r[2], r[3]
        return r
    model.genrate_syn = spy
    out = model.train_step(data, None)
    out['loss'].backward()
    lv = out['log_vars'].materialize()
    params = {k: (v.clone().requires_grad_(True) if M.trainable(k) else v) for k, v in sd_s0.items()}
    sd_t = M.ema(sd_t0, sd_s0)
    gp = [R.bbox_xyxy_to_cxcywh(b)[:, :2] for b in boxes]
    step = M.forward_train_step2 if phase2 else M.forward_train_step1
    ref, _ = step(params, sd_t, img, boxes, labels, gp, dict(M.MODEL_CFG), inj)
    ref['loss'] = M.total_loss(ref)
    if not phase2:
        prior = torch.tensor(M.SHAPE_LIST)
        for i in range(2):
            img_syn, hull, _, _ = R.generate_black_paper(img[i], boxes[i], prior, inj['syn'][i])
            assert hull.shape[0] > 20
            torch.testing.assert_close(captured['boxes'][i][captured['alive'][i]].cpu(), hull, rtol=1e-5, atol=1e-3)
            assert torch.equal(captured['img_syn'][i].cpu(), img_syn)
    assert set(ref.keys()) == set(lv.keys()), set(ref.keys()) ^ set(lv.keys())
    print({k: (round(float(lv[k]), 5), round(float(ref[k]), 5)) for k in ref})
    _check(lv, ref)
    _check_grads(model, ref['loss'], params, tol=3e-3)


def test_full_size_bf16_backbone_properties():
    """BASELINE configs[2] at its real size (round-2 verdict P2): bs 2, 800 x 800, ~300 points per image, bf16 backbone / FPN /
    PSAGG under autocast + fp32 dense head, MIL head and losses, both phases through `Trainer.step`.  Size-independent
    properties: the precision boundary sits where the config puts it (bf16 into the necks, fp32 into and out of the head, fp32 RoI
    blocks and FC stacks), the loss dict keeps one finite key set, every key of an iteration is within 10 % (bag loss: 15 %) of the SAME
    iteration run in fp32 on the same weights / inputs / draws (measured <= 5.3 % / 7 %), the fp32 master weights
    move and stay finite, and a steady-state iteration does not synchronise the host."""
    from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_
    import point_teacher_amd as pta
    dev = torch.device('cuda:0')

    def run(dtype):
        torch.manual_seed(5)
        cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
        cfg.model['burn_in_step'] = 0                                  # iteration 0 = phase 1, iterations 1, 2 = phase 2
        model = pta.build_detector(cfg.model).to(dev)
        benchmark_init_(model, phase2=True)
        model.train()
        tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True, autocast_dtype=dtype)
        data = SyntheticTiles(n=2, size=800, mean_objects=300, seed=1, device=dev)
        seen = {}
        hooks = [model.student.backbone.register_forward_hook(
                     lambda m, i, o: seen.__setitem__('trunk', (type(o[-1]).__name__, getattr(o[-1], 'np', None)))),
                 model.student.neck_agg.lateral_convs[0].register_forward_hook(lambda m, i, o: seen.__setitem__('psagg', o.dtype)),
                 model.student.bbox_head.reg_convs[0].register_forward_hook(lambda m, i, o: seen.__setitem__('head', (i[0].dtype, o.dtype))),
                 model.student.bbox_head.bbox_roi_extractor.register_forward_hook(lambda m, i, o: seen.__setitem__('roi', o.dtype)),
                 model.student.bbox_head.fc_reg[0].register_forward_hook(lambda m, i, o: seen.__setitem__('fc', (i[0].dtype, o.dtype)))]
        g = torch.Generator().manual_seed(17)
        lvs = []
        s0 = tr.flat.student_flat.clone()
        for it in range(3):
            model._inject = dict(neg0=torch.rand(2, 4, 200, generator=g).to(dev), aug=(['horizontal', 'None'], [0.9, 1.1]))
            if it == 0:
                model._inject['syn'] = [{k: v.to(dev) for k, v in _syn_draws(b.shape[0], 70 + i).items()}
                                        for i, b in enumerate(data.batch(0, 2)['gt_bboxes'])]
            if it == 2:
                torch.cuda.synchronize()
                torch.cuda.set_sync_debug_mode('error')
            try:
                out = tr.step(data.batch(0, 2))
            finally:
                torch.cuda.set_sync_debug_mode('default')
            lvs.append(out['log_vars'].materialize())
        for h in hooks:
            h.remove()
        assert torch.isfinite(tr.flat.student_flat).all() and not torch.equal(tr.flat.student_flat, s0)
        assert tr.flat.student_flat.dtype == torch.float32 and tr.flat.grad_flat.dtype == torch.float32        # fp32 master weights
        return seen, lvs

    seen16, lv16 = run(torch.bfloat16)
    assert seen16['trunk'] == ('PlaneAct', 1) and seen16['psagg'] == torch.bfloat16
    assert seen16['head'] == (torch.float32, torch.float32) and seen16['roi'] == torch.float32 and seen16['fc'] == (torch.float32, torch.float32)
    seen32, lv32 = run(None)
    # the trainable stages run plane-native (planes.PlaneAct between the layers, the convolution modules themselves are not called):
    # scaled fp16 x 2 planes of fp32 values (round 5; bf16 x 3 when the census demotes the group), or - under bf16 autocast - one bf16
    # plane per activation on the same kernels
    assert seen32['trunk'] == ('PlaneAct', 2)
    for it, (a, b) in enumerate(zip(lv16, lv32)):
        assert set(a) == set(b) and all(v == v and abs(v) != float('inf') for v in a.values()), (it, a)
        if it == 0:          # same weights, inputs and draws: later iterations start from weights that already differ by an update
            for k in a:
                # the bag loss sits on its log(1 - p + 1e-6) cliff at random initialisation (saturated class logits, DESIGN
                # section 4 "moderate bag logits"): one bf16 rounding of the features moves it by several per cent (measured 7 %)
                # ... and the other keys by up to 5.3 % (loss_centerness) at this random initialisation: 10 % here is a sanity
                # bound, the tight bf16 comparison is test_bf16_backbone_fp32_head (2 % against the bf16 ORACLE at 256 x 256)
                tol = 0.15 if 'mil_bags' in k else 0.10
                assert abs(a[k] - b[k]) <= tol * max(abs(b[k]), 1e-2), (it, k, a[k], b[k])
    assert set(lv16[0]) == set(lv16[1]) == set(lv16[2])


@pytest.mark.parametrize('phase2', [False, True])
def test_teacher_on_a_side_stream_gives_the_same_iteration(phase2):
    """`TS_P2B_FCOS.teacher_stream` (default on): the teacher pass runs on a second HIP stream next to the student's forward.  Same
    loss dict and the same student weights after two `Trainer.step`s as with the teacher inline (RoIAlign's float atomics make two
    runs agree to rounding, not bit for bit)."""
    dev = torch.device('cuda:0')

    def run(side):
        pta, cfg, model = _build(dev, phase2=phase2)
        model.teacher_stream = side
        tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
        img, boxes, labels, metas = _data(dev, seed=6 if not phase2 else 5)
        g = torch.Generator().manual_seed(13)
        lvs = []
        for it in range(2):
            inj = dict(neg0=torch.rand(2, 4, 200, generator=g).to(dev), aug=(['horizontal', 'vertical'], [0.9, 1.1]))
            if not phase2:
                inj['syn'] = [{n: t.to(dev) for n, t in _syn_draws(b.shape[0], 40 + i).items()} for i, b in enumerate(boxes)]
            model._inject = inj
            data = dict(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in boxes], gt_labels=[l.to(dev) for l in labels])
            lvs.append(tr.step(data)['log_vars'].materialize())
        used = model._side_stream is not None
        return lvs, tr.flat.student_flat.clone(), used
    a, wa, used_a = run(False)
    b, wb, used_b = run(True)
    assert not used_a and used_b
    for it, (x, y) in enumerate(zip(a, b)):
        assert set(x) == set(y)
        for k in x:
            # iteration 0: the same forward pass whichever stream the teacher ran on (fp32 reduction order aside: 1e-6 measured).
            # Iteration 1 starts from weights whose gradients went through RoIAlign's float atomics (run-to-run differences of 1e-3
            # on a loss key were measured between two INLINE runs): a sanity bound; a race between the streams shows in iteration 0
            tol = 1e-4 if it == 0 else 1e-2
            assert abs(x[k] - y[k]) <= tol * max(abs(x[k]), 1e-2), (it, k, x[k], y[k])
    assert float((wa - wb).abs().max()) <= 1e-3 * float(wa.abs().max())
