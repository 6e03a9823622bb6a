"""Host-side logic of the hot path that needs no GPU: branch selection of the dense head (burn-in step 1 evaluates only the tower
whose outputs are read), the split ResNet forward the shared frozen stem relies on, the executed-FLOP accounting of bench.py
and the host path of the index-upload helper."""
import importlib.util
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _head():
    import point_teacher_amd as pta
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
    hc = dict(cfg.model['_model_']['bbox_head'])
    hc.update(train_cfg=cfg.model['train_cfg'], test_cfg=cfg.model['test_cfg'])
    torch.manual_seed(0)
    return pta.build_from_cfg(hc, pta.HEADS)


def test_head_branches_are_the_same_numbers():
    """`branches='reg'` / `'cls'` skip a tower and return None for its outputs; what they do return equals the full forward."""
    head = _head().eval()
    x = [torch.randn(2, 256, 12, 12)]
    with torch.no_grad():
        cls, reg, ctr, pts = head(x)
        cls_r, reg_r, ctr_r, pts_r = head(x, branches='reg')
        cls_c, reg_c, ctr_c, pts_c = head(x, branches='cls')
    assert cls_r[0] is None and reg_c[0] is None and ctr_c[0] is None
    assert torch.equal(reg_r[0], reg[0]) and torch.equal(ctr_r[0], ctr[0]) and torch.equal(cls_c[0], cls[0])
    assert torch.equal(pts_r[0], pts[0]) and torch.equal(pts_c[0], pts[0])
    fc, fr, ft = head._flat(cls_r, reg_r, ctr_r)
    assert fc is None and fr.shape == (2, 144, 4) and ft.shape == (2, 144)
    fc, fr, ft = head._flat(cls_c, reg_c, ctr_c)
    assert fr is None and ft is None and fc.shape == (2, 144, head.cls_out_channels)
    # gradients: the skipped tower receives none, the evaluated one the same as in the full pass
    head.train()
    head.zero_grad()
    head(x, branches='reg')[1][0].sum().backward()
    assert head.cls_convs[0].conv.weight.grad is None and head.conv_cls.weight.grad is None
    g_reg = head.reg_convs[0].conv.weight.grad.clone()
    head.zero_grad()
    head(x)[1][0].sum().backward()
    torch.testing.assert_close(head.reg_convs[0].conv.weight.grad, g_reg)


def test_resnet_split_forward():
    """forward(x) == forward(None, stem=forward_stem(x)); the stem is exactly the frozen part (no parameter in it trains)."""
    from point_teacher_amd.nn_modules import ResNet
    torch.manual_seed(1)
    net = ResNet(50, frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='caffe').train()
    x = torch.randn(2, 3, 64, 64)
    with torch.no_grad():
        full = net(x)
        stem = net.forward_stem(x)
        split = net(None, stem=stem)
        rows = net(None, stem=(stem[0][:1], [o[:1] for o in stem[1]]))          # the teacher continues from the first rows
    assert len(full) == len(split) == 4
    for a, b, c in zip(full, split, rows):
        assert torch.equal(a, b)
        torch.testing.assert_close(a[:1], c, rtol=1e-3, atol=1e-4 * float(a.abs().max()))   # another batch size: another CPU conv kernel
    stem_params = [p for m in (net.conv1, net.bn1, net.layer1) for p in m.parameters()]
    assert stem_params and not any(p.requires_grad for p in stem_params)
    assert any(p.requires_grad for p in net.layer2.parameters())
    # nothing frozen: the "stem" is conv1 / bn1 / maxpool only and callers must not share it (TS_P2B_FCOS._shared_stem checks)
    free = ResNet(50, frozen_stages=-1, norm_eval=True)
    assert free.frozen_stages == -1 and any(p.requires_grad for p in free.conv1.parameters())


def test_iteration_flops_accounting():
    import point_teacher_amd as pta
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py')).to_dict()['model']
    f1, f1r = bench.iteration_flops('step1', cfg, 2, 800, 300), bench.iteration_flops('step1', cfg, 2, 800, 300, executed=False)
    f2, f2r = bench.iteration_flops('step2', cfg, 2, 800, 300), bench.iteration_flops('step2', cfg, 2, 800, 300, executed=False)
    trunk, stem, head = 60.2e9, 10.2e9, 47.5e9
    # the reference additionally runs the teacher's stem (both phases) and the forward of one unread tower per step-1 head pass
    assert f2r - f2 == pytest.approx(2.0 * 2 * stem, rel=1e-9)
    assert f1r - f1 == pytest.approx(2.0 * 2 * (stem + 2 * 0.5 * head), rel=1e-9)
    assert 3.8e12 < f1 < 4.0e12 and 3.0e12 < f2 < 3.2e12
    # size and batch scale the convolution part only as area x batch
    assert bench.iteration_flops('step2', cfg, 4, 800, 300) == pytest.approx(2 * f2, rel=1e-9)


def test_upload_helper_host_path():
    from point_teacher_amd import functional as Fn
    a = Fn.upload_i32([3, 1, 4], 'cpu')
    b = Fn.upload_f32(np.array([1.5, 2.5]), 'cpu')
    c = Fn.upload_i32([9, 9, 9], 'cpu')
    assert a.tolist() == [3, 1, 4] and a.dtype == torch.int32 and b.tolist() == [1.5, 2.5] and c.tolist() == [9, 9, 9]
    off, tot = Fn.make_offsets([2, 0, 5], 'cpu')
    assert off.tolist() == [0, 2, 2, 7] and tot == 7
    assert Fn._ring.seg_words * Fn._ring.SEG == Fn._ring.words
