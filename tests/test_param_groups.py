"""Optimizer parameter groups against mmcv's DefaultOptimizerConstructor rules (mmcv/runner/optimizer/default_constructor.py,
the constructor `optimizer = dict(type='SGD', paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))` selects): a parameter
NAMED `bias` takes bias_lr_mult / bias_decay_mult unless it belongs to a normalisation layer; norm weights AND biases keep
lr x 1 and decay x norm_decay_mult (1).  Checked on the flat layout [weights | biases | frozen] of runtime.FlatParams on the CPU
(no kernel runs), for the RetinaNet baseline config - the one shipped config that trains BatchNorm under a paramwise_cfg - and
the pretrained-checkpoint plumbing of the backbone."""
import os
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_norm_biases_are_not_bias_group():
    from point_teacher_amd.runtime import FlatParams

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(3, 8, 3, bias=True)
            self.bn = torch.nn.BatchNorm2d(8)
            self.gn = torch.nn.GroupNorm(2, 8)
            self.fc = torch.nn.Linear(8, 4)
            self.frozen = torch.nn.BatchNorm2d(8)
            for p in self.frozen.parameters():
                p.requires_grad = False
    net = Net()
    flat = FlatParams(net, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
    seg = {}
    for name, _ in flat.order:
        off, n = flat.slices[name]
        seg[name] = 'weights' if off < flat.n_weights else ('biases' if off < flat.n_train else 'frozen')
    assert seg['conv.bias'] == 'biases' and seg['fc.bias'] == 'biases'                   # bias_lr_mult / bias_decay_mult
    assert seg['bn.bias'] == 'weights' and seg['gn.bias'] == 'weights'                   # norm layers: lr x 1, decay x norm_decay_mult
    assert seg['bn.weight'] == 'weights' and seg['conv.weight'] == 'weights' and seg['fc.weight'] == 'weights'
    assert seg['frozen.weight'] == 'frozen' and seg['frozen.bias'] == 'frozen'
    assert flat.n_biases == 8 + 4
    assert flat.group_mults == [(1., 1.), (2., 0.)] and flat.group_ends == [flat.n_weights, flat.n_train]


def test_paramwise_cfg_of_the_yolof_baseline_and_dcn_rules():
    """`paramwise_cfg=dict(norm_decay_mult=0., custom_keys={'backbone': dict(lr_mult=1. / 3)})`
    (configs/baselines/aitodv2_yolof_r50_1x.py:70-71): a custom key wins over every other rule for the parameters whose name
    contains it; norm layers elsewhere take decay x 0; every distinct (lr_mult, decay_mult) pair is one contiguous group of
    the flat buffers.  DCN modules: their biases (and the conv_offset child's) take neither bias_lr_mult nor bias_decay_mult;
    conv_offset takes dcn_offset_lr_mult (mmcv DefaultOptimizerConstructor `is_dcn_module`)."""
    from point_teacher_amd.nn_modules import ModulatedDeformConv2dPack
    from point_teacher_amd.runtime import FlatParams, param_multipliers

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, bias=True), torch.nn.BatchNorm2d(8))
            self.neck = torch.nn.Sequential(torch.nn.Conv2d(8, 8, 3, bias=True), torch.nn.BatchNorm2d(8))
            self.dw = torch.nn.Conv2d(8, 8, 3, groups=8, bias=False)
            self.dcn = ModulatedDeformConv2dPack(8, 8, 3, padding=1)
    net = Net()
    pw = dict(norm_decay_mult=0., custom_keys={'backbone': dict(lr_mult=1. / 3)})
    m = param_multipliers(net, pw)
    third = float(1. / 3)
    assert m['backbone.0.weight'] == m['backbone.0.bias'] == m['backbone.1.weight'] == (third, 1.)   # the key wins, decay x 1
    assert m['neck.0.weight'] == (1., 1.) and m['neck.0.bias'] == (1., 1.)
    assert m['neck.1.weight'] == m['neck.1.bias'] == (1., 0.)
    flat = FlatParams(net, paramwise_cfg=pw)
    assert flat.group_mults == [(1., 1.), (third, 1.), (1., 0.)]
    for name, _ in flat.order:
        off = flat.slices[name][0]
        g = sum(off >= e for e in flat.group_ends)
        assert flat.group_mults[g] == m[name], name
    m2 = param_multipliers(net, dict(bias_lr_mult=2., bias_decay_mult=0., dwconv_decay_mult=0.5, dcn_offset_lr_mult=0.1))
    assert m2['neck.0.bias'] == (2., 0.) and m2['dw.weight'] == (1., 0.5)
    assert m2['dcn.bias'] == (1., 1.) and m2['dcn.weight'] == (1., 1.)
    assert m2['dcn.conv_offset.weight'] == (0.1, 1.) and m2['dcn.conv_offset.bias'] == (0.1, 1.)
    with pytest.raises(AssertionError, match='groups'):
        FlatParams(net, paramwise_cfg=dict(custom_keys={f'backbone.{i}': dict(lr_mult=0.1 * (i + 1)) for i in range(2)} |
                                           {f'neck.{i}': dict(lr_mult=0.01 * (i + 1)) for i in range(2)} |
                                           {'dw': dict(lr_mult=5.), 'dcn.weight': dict(lr_mult=6.), 'dcn.bias': dict(lr_mult=7.),
                                            'dcn.conv': dict(lr_mult=8.)}))


def test_dead_segment_and_relayout():
    """Trainable parameters without a gradient leave the live groups: they keep their values, have no gradient / momentum slot
    (`grad is None`, what torch.optim.SGD skips), and come back - with their values - when revived."""
    from point_teacher_amd.runtime import FlatParams
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 6), torch.nn.Linear(6, 7), torch.nn.Linear(7, 3))
    for p in net[2].parameters():
        p.requires_grad = False
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    flat = FlatParams(net, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
    n_all = flat.n_train
    flat.grad_flat.copy_(torch.arange(flat.n_train, dtype=torch.float32))
    flat.mom_flat.copy_(-torch.arange(flat.n_train, dtype=torch.float32))
    g0 = {n: p.grad.clone() for n, p in net.named_parameters() if p.requires_grad}
    assert flat.relayout({'1.weight', '1.bias'}) and not flat.relayout({'1.weight', '1.bias'})
    assert flat.n_train == n_all - 44 - 8 and flat.n_dead == 44 + 8 and flat.frozen_start == flat.n_train + flat.n_dead
    assert net[1].weight.grad is None and net[1].bias.grad is None and flat.check_views()
    for n, p in net.named_parameters():
        assert torch.equal(p, before[n]), n                                             # values survive
        off, k = flat.slices[n]
        assert p.data_ptr() == flat.student_flat.data_ptr() + 4 * off
        if p.requires_grad and n not in flat.dead:
            assert torch.equal(p.grad, g0[n]) and torch.equal(flat.mom_flat[off:off + k], -g0[n].reshape(-1))   # carried over
    assert [n for n, _ in flat.dead_params] == ['1.weight', '1.bias']
    net[1].weight.grad = torch.ones_like(net[1].weight)                                 # autograd reached it after all
    assert flat.take_revived() == ['1.weight'] and net[1].weight.grad is None
    assert flat.relayout({'1.bias'}) and flat.n_dead == 8
    off, k = flat.slices['1.weight']
    assert off < flat.n_weights and float(flat.grad_flat[off:off + k].abs().sum()) == 0 and torch.equal(net[1].weight, before['1.weight'])
    with pytest.raises(AssertionError, match='not trainable'):
        flat.relayout({'2.weight'})
    # the gradient that revives a parameter is kept for the update of that step (round-3 advice): take_revived(keep) hands it
    # over (summing several steps' worth), add_to_grad places it in the re-laid buffer
    net[1].bias.grad = torch.full_like(net[1].bias, 2.0)
    keep = {}
    assert flat.take_revived(keep) == ['1.bias'] and net[1].bias.grad is None and torch.equal(keep['1.bias'], torch.full((7,), 2.0))
    net[1].bias.grad = torch.full_like(net[1].bias, 0.5)
    assert flat.take_revived(keep) == ['1.bias'] and torch.equal(keep['1.bias'], torch.full((7,), 2.5))
    assert flat.relayout(set()) and flat.n_dead == 0
    flat.add_to_grad('1.bias', keep['1.bias'])
    off, k = flat.slices['1.bias']
    assert torch.equal(flat.grad_flat[off:off + k], torch.full((7,), 2.5)) and torch.equal(net[1].bias.grad, torch.full((7,), 2.5))


def test_revived_3x3_conv_gradient_lands_in_channels_last_order():
    """Round-4 advice: under channels_last the flat slot of a 4-D weight is stored as (O, KH, KW, I); the gradient that revives a
    dead 3x3 convolution arrives in logical [O, I, KH, KW] order and must be added through the permuted view, not the raw slice."""
    from point_teacher_amd.runtime import FlatParams
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Conv2d(4, 6, 3, padding=1), torch.nn.Conv2d(6, 5, 3, padding=1), torch.nn.Conv2d(5, 2, 1))
    flat = FlatParams(net, channels_last=True, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.), dead={'1.weight', '1.bias'})
    assert net[1].weight.grad is None and net[1].weight.is_contiguous(memory_format=torch.channels_last)
    g = torch.randn(5, 6, 3, 3)
    net[1].weight.grad = g.clone()
    keep = {}
    assert flat.take_revived(keep) == ['1.weight']
    assert flat.relayout({'1.bias'})
    flat.add_to_grad('1.weight', keep['1.weight'])
    assert torch.equal(net[1].weight.grad, g), 'the kept gradient must appear, element for element, in the parameter\'s .grad view'
    off, n = flat.slices['1.weight']
    assert torch.equal(flat.grad_flat[off:off + n], g.permute(0, 2, 3, 1).reshape(-1))       # (O, KH, KW, I) storage order
    flat.add_to_grad('1.weight', keep['1.weight'])                                           # and it accumulates
    assert torch.equal(net[1].weight.grad, 2 * g)
    g1 = torch.randn(2, 5, 1, 1)                                                             # 1x1: both orders coincide
    flat.grad_flat.zero_()
    flat.add_to_grad('2.weight', g1)
    assert torch.equal(net[2].weight.grad, g1)


def test_retinanet_baseline_config_groups():
    import point_teacher_amd as pta
    from point_teacher_amd.runtime import FlatParams
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'baselines', 'aitodv2_retinanet_r50_1x.py'))
    assert cfg.optimizer.paramwise_cfg == dict(bias_lr_mult=2., bias_decay_mult=0.)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = pta.build_detector(cfg.model)
    flat = FlatParams(model, paramwise_cfg=cfg.optimizer.paramwise_cfg)
    bn_bias = [n for n, p in model.named_parameters() if p.requires_grad and '.bn' in n and n.endswith('.bias')]
    assert bn_bias, 'this config trains BatchNorm affines'
    for n in bn_bias:
        assert flat.slices[n][0] < flat.n_weights, n
    head_bias = [n for n, p in model.named_parameters() if n.startswith('bbox_head') and n.endswith('.bias')]
    assert head_bias and all(flat.n_weights <= flat.slices[n][0] < flat.n_train for n in head_bias)


def test_pretrained_init_cfg_is_honoured(tmp_path, monkeypatch):
    """backbone init_cfg=dict(type='Pretrained', checkpoint='open-mmlab://detectron/resnet50_caffe') (every config): the
    file is looked up under $PT_PRETRAINED_DIR; found -> loaded, not found -> a loud warning and `pretrained_loaded` False."""
    from point_teacher_amd.nn_modules import ResNet
    monkeypatch.setenv('PT_PRETRAINED_DIR', str(tmp_path))
    with pytest.warns(RuntimeWarning, match='RANDOMLY initialised'):
        r = ResNet(50, style='caffe', init_cfg=dict(type='Pretrained', checkpoint='open-mmlab://detectron/resnet50_caffe'))
    assert r.pretrained_loaded is False
    torch.manual_seed(1)
    donor = ResNet(50, style='caffe')
    assert donor.pretrained_loaded is None
    with torch.no_grad():
        donor.layer3[2].conv2.weight.fill_(0.125)
    torch.save(dict(state_dict={'backbone.' + k: v for k, v in donor.state_dict().items()}),
               tmp_path / 'open-mmlab_detectron_resnet50_caffe.pth')
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        r2 = ResNet(50, style='caffe', init_cfg=dict(type='Pretrained', checkpoint='open-mmlab://detectron/resnet50_caffe'))
    assert r2.pretrained_loaded is True
    assert torch.equal(r2.layer3[2].conv2.weight, donor.layer3[2].conv2.weight)
    assert torch.equal(r2.conv1.weight, donor.conv1.weight)
