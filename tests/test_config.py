"""The registry/config boundary: our loader parses the reference's own config files (when the
reference tree is present - never on the GPU box) to the same dictionary as the shipped mirrors."""
import os

import pytest

from point_teacher_amd.registry import Config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MINE = os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher')
REF = '/root/reference/HBB_TOD/configs/point_teacher'


@pytest.mark.parametrize('p', [0, 30, 60, 100])
def test_shipped_config_loads(p):
    cfg = Config.fromfile(os.path.join(MINE, f'aitodv2_point_teacher_{p}.py'))
    assert cfg.model.type == 'TS_P2B_FCOS'
    assert cfg.model._model_.type == 'Student_FCOS'
    assert cfg.model._model_.bbox_head.type == 'TS_P2BFCOSHead'
    assert cfg.optimizer.type == 'SGD' and cfg.optimizer.lr == 0.005 and cfg.optimizer.momentum == 0.9
    assert cfg.optimizer_config == dict(grad_clip=dict(max_norm=35, norm_type=2))      # _delete_ honoured
    assert cfg.lr_config.warmup == 'constant' and cfg.lr_config.warmup_iters == 10000
    assert cfg.data.train.type == 'AITODDataset'                                         # merged with the base
    assert len(cfg.model.train_cfg.fine_proposal_cfg) == 2


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present (GPU box)')
@pytest.mark.parametrize('p', [0, 30, 60, 100])
def test_reference_config_loads_and_matches_mirror(p):
    ref = Config.fromfile(os.path.join(REF, f'aitodv2_point_teacher_{p}%.py')).to_dict()
    mine = Config.fromfile(os.path.join(MINE, f'aitodv2_point_teacher_{p}.py')).to_dict()
    assert ref == mine


OBB_MINE = os.path.join(ROOT, 'point_teacher_amd', 'configs', 'obb', 'point_teacher', 'sodaa_fcos_pointteacher_1x.py')
OBB_REF = '/root/reference/OBB_TOD/configs/point teacher/sodaa_fcos_pointteacher_1x.py'


def test_shipped_obb_config_loads_and_builds():
    import point_teacher_amd as pta
    cfg = Config.fromfile(OBB_MINE)
    assert cfg.model.type == 'RotatedFCOS_TS' and cfg.model._model_.type == 'RotatedFCOS_Student'
    assert cfg.model._model_.bbox_head.type == 'TS_P2RBRotatedFCOSHead'
    assert cfg.optimizer == dict(type='SGD', lr=0.005, momentum=0.9, weight_decay=0.0001)      # base merged with lr override
    assert cfg.lr_config.warmup == 'linear' and cfg.lr_config.warmup_iters == 500
    assert cfg.data.train.type == 'SODAADOTADataset' and cfg.data.samples_per_gpu == 2
    model = pta.build_detector(cfg.model)             # construction is host-only (no kernel runs)
    head = model.student.bbox_head
    names = dict(head.named_parameters())
    assert 'conv_angle.weight' in names and 'scale_angle.scale' in names and 'cls_convs.0.gn.weight' in names
    assert 'cls_convs.0.conv.bias' not in names       # bias='auto' with GroupNorm
    assert all(not p.requires_grad for p in model.teacher.parameters())
    bn = dict(model.student.backbone.named_parameters())
    assert bn['layer2.0.bn1.weight'].requires_grad and not bn['layer1.0.bn1.weight'].requires_grad


@pytest.mark.skipif(not os.path.isfile(OBB_REF), reason='reference tree not present (GPU box)')
def test_reference_obb_config_loads_and_matches_mirror():
    assert Config.fromfile(OBB_REF).to_dict() == Config.fromfile(OBB_MINE).to_dict()


BASE_REF = '/root/reference/HBB_TOD/configs/baselines'
BASE_MINE = os.path.join(ROOT, 'point_teacher_amd', 'configs', 'baselines')


@pytest.mark.skipif(not os.path.isdir(BASE_REF), reason='reference tree not present (GPU box)')
@pytest.mark.parametrize('name', ['aitodv2_fcos_r50_1x.py', 'aitodv2_retinanet_r50_1x.py'])
def test_reference_baseline_config_loads_and_matches_mirror(name):
    """Row N4: the reference's own baseline configs parse (their absolute `_base_` paths remapped) to the shipped mirrors."""
    assert Config.fromfile(os.path.join(BASE_REF, name)).to_dict() == Config.fromfile(os.path.join(BASE_MINE, name)).to_dict()
