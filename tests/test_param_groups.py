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
    flat = FlatParams(net)
    seg = {}
    for name, _ in flat.order:
        off, n = flat.slices[name]
        seg[name] = 'weights' if off < flat.n_weights else ('biases' if off < flat.n_train else 'frozen')
    assert seg['conv.bias'] == 'biases' and seg['fc.bias'] == 'biases'                   # bias_lr_mult / bias_decay_mult
    assert seg['bn.bias'] == 'weights' and seg['gn.bias'] == 'weights'                   # norm layers: lr x 1, decay x norm_decay_mult
    assert seg['bn.weight'] == 'weights' and seg['conv.weight'] == 'weights' and seg['fc.weight'] == 'weights'
    assert seg['frozen.weight'] == 'frozen' and seg['frozen.bias'] == 'frozen'
    assert flat.n_biases == 8 + 4


def test_retinanet_baseline_config_groups():
    import point_teacher_amd as pta
    from point_teacher_amd.runtime import FlatParams
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'baselines', 'aitodv2_retinanet_r50_1x.py'))
    assert cfg.optimizer.paramwise_cfg == dict(bias_lr_mult=2., bias_decay_mult=0.)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = pta.build_detector(cfg.model)
    flat = FlatParams(model)
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
