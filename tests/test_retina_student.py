"""Row N4 as SURVEY words it - another student architecture UNDER THE SAME TEACHER: `TS_P2BRetinaHead` (anchor-based, no
centerness, L1 on deltas) inside `TS_P2B_FCOS` with the EMA teacher, pseudo-box generation, MIL refinement, augmentation and
burn-in schedule unchanged.  The reference has no such head to compare with (parity unpinned): property tests."""
import os
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_retina_0.py')


def test_config_builds_and_differs_only_in_the_head():
    import point_teacher_amd as pta
    a = pta.Config.fromfile(CFG).to_dict()
    b = pta.Config.fromfile(CFG.replace('_retina_0', '_0')).to_dict()
    ha, hb = a['model']['_model_'].pop('bbox_head'), b['model']['_model_'].pop('bbox_head')
    a['detector'].pop('bbox_head'), b['detector'].pop('bbox_head')
    assert a == b and ha['type'] == 'TS_P2BRetinaHead' and hb['type'] == 'TS_P2BFCOSHead'
    assert ha['bbox_roi_extractor'] == hb['bbox_roi_extractor'] and ha['loss_cls'] == hb['loss_cls']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = pta.build_detector(pta.Config.fromfile(CFG).model)
    names = dict(model.student.bbox_head.named_parameters())
    assert names['retina_cls.weight'].shape[0] == 3 * 8 and names['retina_reg.weight'].shape[0] == 3 * 4
    assert 'conv_centerness.weight' not in names and 'shared_fcs_bag.0.0.weight' in names        # the MIL head is the parent's
    assert [n for n, _ in model.student.named_parameters()] == [n for n, _ in model.teacher.named_parameters()]


@pytest.mark.gpu
def test_retina_student_trains_under_the_teacher():
    import point_teacher_amd as pta
    from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_
    dev = torch.device('cuda:0')
    torch.manual_seed(4)
    cfg = pta.Config.fromfile(CFG)
    cfg.model['burn_in_step'] = 1                       # iterations 0, 1: phase 1; 2, 3: phase 2
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = pta.build_detector(cfg.model).to(dev)
    benchmark_init_(model, phase2=True)
    model.train()
    head = model.student.bbox_head
    tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
    data = SyntheticTiles(n=4, size=256, mean_objects=20, seed=3, device=dev)
    # dense outputs: every (location, base anchor) is a point; distances decode back to the anchors at zero deltas
    with torch.no_grad():
        feat = model.student.extract_feat(data.batch(0, 2)['img'])
        cls, dist, ctr, pts = head(feat)
        assert cls[0].shape == (2, 8, 32, 32 * 3) and dist[0].shape == (2, 4, 32, 96) and pts[0].shape == (32 * 32 * 3, 2)
        head.retina_reg.weight.zero_(); head.retina_reg.bias.zero_()
        _, d0, _, _ = head(feat)
        anchors, centres = head._anchors(32, 32, dev)
        box = torch.stack((centres[:, 0] - d0[0][0, 0].reshape(-1), centres[:, 1] - d0[0][0, 1].reshape(-1),
                           centres[:, 0] + d0[0][0, 2].reshape(-1), centres[:, 1] + d0[0][0, 3].reshape(-1)), -1)
        torch.testing.assert_close(box, anchors, rtol=1e-5, atol=1e-4)
        torch.nn.init.normal_(head.retina_reg.weight, 0, 0.01)
    t0 = tr.flat.teacher_flat.clone()
    keys = None
    for it in range(4):
        out = tr.step(data.batch(it, 2))
        lv = out['log_vars'].materialize()
        assert all(v == v and abs(v) != float('inf') for v in lv.values()), (it, lv)
        keys = set(lv) if keys is None else keys
        assert set(lv) == keys and lv['loss_centerness'] == 0.0 and lv['loss_cls'] > 0 and lv['loss_bbox'] >= 0
        if it == 0:
            g = tr.flat.grad_flat
            for n in ('bbox_head.retina_cls.weight', 'bbox_head.retina_reg.weight', 'bbox_head.fc_ins.0.weight', 'backbone.layer3.0.conv1.weight'):
                off, k = tr.flat.slices[n]
                assert float(g[off:off + k].abs().sum()) > 0, n
    assert {'stage0_loss_mil_bbox', 'stage0_loss_mil_bags', 'loss_cls', 'loss_bbox', 'loss_centerness'} <= keys
    assert model.count == 4 and not torch.equal(tr.flat.teacher_flat, t0)                 # the EMA teacher follows the student
    assert all(p.grad is None for p in model.teacher.parameters())
    model.eval()
    with torch.no_grad():
        model.teacher.bbox_head.retina_cls.bias.fill_(-1.0)
        res = model.simple_test(data.batch(0, 1)['img'], data.batch(0, 1)['img_metas'])
    assert len(res) == 1 and len(res[0]) == 8 and all(isinstance(r, np.ndarray) and r.shape[1] == 5 for r in res[0])
    assert sum(r.shape[0] for r in res[0]) > 0
