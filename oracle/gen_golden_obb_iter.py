#!/usr/bin/env python3
"""Whole-iteration golden of the ORIENTED variant (config 5): the reference's OWN `RotatedFCOS_TS.forward_train`
(OBB_TOD/mmrotate/models/detectors/rotated_fcos_teacher_student.py:135-304: both burn-in phases, `genrate_syn` ->
`generate_black_paper`, `forward_mil_head_burn_in_step1/2`, `update_points`, `strong_augmentation`) around a REAL
`TS_P2RBRotatedFCOSHead` instance built by its own constructor from the reference's own config file, its
`RotatedSingleRoIExtractor`, `RotatedIoULoss` / `DN_IoULoss`, `DistanceAnglePointCoder`, the HBB fork's assigners and
costs, run on the CPU in this container on seeded weights and inputs with every random draw logged.
TEST INFRASTRUCTURE - runs only here (/root/reference is read by path, nothing of it is copied); writes
tests/golden/ref_iter_obb.npz.

    python oracle/gen_golden_obb_iter.py

What is NOT the reference (absent from this image) and is bound to the oracle's restatement instead - every one of
them a "parity unpinned" op of DESIGN section 4:
  mmcv.ops.RoIAlignRotated     -> oracle.ref_ops.roi_align_rotated          mmcv.ops.nms_rotated -> oracle.ref_ops.nms_rotated
  mmcv.ops.box_iou_rotated     -> oracle.ref_ops.box_iou_rotated (behind the reference's own `rbbox_overlaps`)
  mmcv.ops.diff_iou_rotated_2d -> oracle.ref_obb._RotIoU (fp64 polygon clipping, central-difference gradient)
  torchvision...functional.rotate -> oracle.ref_obb.rotate_image_nearest   cv2.fillPoly -> oracle.ref_ops.fill_convex_quads
  mmcv's ConvModule / BaseModule (compositions of torch modules, _ref_loader.py) and ResNet-50 / FPN / PSAGG, which need
  mmcv to instantiate: `oracle.ref_obb.extract_feat` on the same weights.
As in gen_golden_iter.py, `torch.topk` is served by a stable top-k (the 5th / 6th L1-nearest grid points of a generic point
tie structurally) and `img.cpu()` inside `genrate_syn` copies.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_loader as L  # noqa: E402
from gen_golden_iter import DrawLog, grad_sample  # noqa: E402
from oracle import ref_model as M  # noqa: E402
from oracle import ref_obb as O  # noqa: E402
from oracle import ref_ops as R  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
torch.set_num_threads(8)

GRAD_KEYS = ['bbox_head.conv_cls.weight', 'bbox_head.conv_reg.weight', 'bbox_head.conv_angle.weight', 'bbox_head.conv_centerness.weight',
             'bbox_head.reg_convs.3.conv.weight', 'bbox_head.reg_convs.0.gn.weight', 'bbox_head.cls_convs.1.gn.bias',
             'bbox_head.scales.0.scale', 'bbox_head.scale_angle.scale', 'bbox_head.fc_cls.0.weight', 'bbox_head.fc_ins.0.weight',
             'bbox_head.fc_reg.0.weight', 'bbox_head.shared_fcs_bag.0.1.weight', 'bbox_head.shared_fcs_reg.0.0.weight',
             'neck_agg.lateral_convs.4.conv.weight', 'neck.fpn_convs.0.conv.weight', 'backbone.layer4.2.conv3.weight',
             'backbone.layer3.0.bn2.weight', 'backbone.layer2.0.conv1.weight']


class OracleRoIAlignRotated(nn.Module):
    """Stands where mmcv.ops.RoIAlignRotated would (absent; parity unpinned)."""

    def __init__(self, out_size=None, spatial_scale=1.0, sample_num=0, aligned=True, clockwise=False, output_size=None,
                 sampling_ratio=None):
        super().__init__()
        out_size = out_size if out_size is not None else output_size
        sample_num = sample_num if sampling_ratio is None else sampling_ratio
        self.out_size = out_size if isinstance(out_size, int) else out_size[0]
        self.output_size = (self.out_size, self.out_size)
        self.spatial_scale, self.sample_num, self.aligned, self.clockwise = spatial_scale, sample_num, aligned, clockwise

    def forward(self, feat, rois):
        return R.roi_align_rotated(feat, rois, self.out_size, self.spatial_scale, self.sample_num, self.aligned, self.clockwise)


def install_obb_iteration():
    """The oriented files the iteration needs on top of _ref_loader.install_obb, with the absent native ops bound."""
    captured = {}
    L.install()
    ops = sys.modules['mmcv.ops']

    def nms_rotated(dets, scores, thr, labels=None):
        keep = R.nms_rotated(dets[:, :5], scores, thr)
        captured.setdefault('keep', []).append(keep.clone())
        return torch.cat([dets[keep, :5], scores[keep, None]], 1), keep

    def diff_iou_rotated_2d(b1, b2):
        return O._RotIoU.apply(b1[0], b2[0])[None]

    def box_iou_rotated(b1, b2, mode='iou', aligned=False, clockwise=True):
        assert mode == 'iou'
        return R.box_iou_rotated(b1, b2, aligned=aligned)
    ops.nms_rotated, ops.diff_iou_rotated_2d, ops.box_iou_rotated = nms_rotated, diff_iou_rotated_2d, box_iou_rotated
    ops.RoIAlignRotated = OracleRoIAlignRotated
    sys.modules['mmcv'].ops = ops
    if not hasattr(sys.modules['mmcv.utils'], 'to_2tuple'):
        sys.modules['mmcv.utils'].to_2tuple = lambda v: v if isinstance(v, tuple) else (v, v)
    L.install_obb()
    imp = importlib.import_module
    mr = os.path.join(L.OBB, 'mmrotate')
    md = os.path.join(L.HBB, 'mmdet', 'models')
    # the real point generator of the anchor-free heads (install_obb stubbed the package before rotated_anchor_free_head imported it)
    L._pkg('mmdet.core.anchor', os.path.join(L.HBB, 'mmdet', 'core', 'anchor'))
    sys.modules.pop('mmdet.core.anchor.point_generator', None)
    imp('mmdet.core.anchor.builder')
    hbb_pg = imp('mmdet.core.anchor.point_generator').MlvlPointGenerator

    class MlvlPointGenerator(hbb_pg):
        """The OBB tree ran against an upstream mmdet whose `grid_priors` also takes `dtype`; the HBB fork vendors the older
        signature (core/anchor/point_generator.py:77).  Same points ((i + 0.5) * stride), cast to the requested dtype."""

        def grid_priors(self, featmap_sizes, dtype=torch.float32, device='cuda', with_stride=False):
            return [p.to(dtype) for p in super().grid_priors(featmap_sizes, device=device, with_stride=with_stride)]
    sys.modules['mmrotate.models.dense_heads.rotated_anchor_free_head'].MlvlPointGenerator = MlvlPointGenerator
    # `h_bbox_coder=dict(type='DistancePointBBoxCoder')` of the head: an upstream-mmdet class the HBB fork does not vendor; the
    # head builds it (:202) and never calls it - a placeholder that refuses to be used
    coders = imp('mmrotate.core.bbox.builder').ROTATED_BBOX_CODERS

    @coders.register_module()
    class DistancePointBBoxCoder:                               # noqa: F811
        def __init__(self, clip_border=True):
            self.clip_border = clip_border

        def encode(self, *a, **k):
            raise AssertionError('h_bbox_coder is not on the Point-Teacher path')
        decode = encode
    # the reference's own rbbox_overlaps (core/bbox/iou_calculators/rotate_iou2d_calculator.py) over the bound box_iou_rotated
    L._pkg('mmrotate.core.bbox.iou_calculators', os.path.join(mr, 'core', 'bbox', 'iou_calculators'))
    imp('mmrotate.core.bbox.iou_calculators.builder')
    calc = imp('mmrotate.core.bbox.iou_calculators.rotate_iou2d_calculator')
    ic = sys.modules['mmrotate.core.bbox.iou_calculators']
    ic.rbbox_overlaps, ic.build_iou_calculator = calc.rbbox_overlaps, imp('mmrotate.core.bbox.iou_calculators.builder').build_iou_calculator
    # losses: the reference's rotated_iou_loss.py (RotatedIoULoss, DN_IoULoss) and the HBB fork's registry entries
    imp('mmrotate.models.losses.rotated_iou_loss')
    # RoI extractor: the real RotatedSingleRoIExtractor over mmdet's BaseRoIExtractor
    L._pkg('mmdet.models.roi_heads', os.path.join(md, 'roi_heads'))
    L._pkg('mmdet.models.roi_heads.roi_extractors', os.path.join(md, 'roi_heads', 'roi_extractors'))
    imp('mmdet.models.roi_heads.roi_extractors.base_roi_extractor')
    L._pkg('mmrotate.models.roi_heads', os.path.join(mr, 'models', 'roi_heads'))
    L._pkg('mmrotate.models.roi_heads.roi_extractors', os.path.join(mr, 'models', 'roi_heads', 'roi_extractors'))
    imp('mmrotate.models.roi_heads.roi_extractors.rotate_single_level_roi_extractor')
    # what the extractor's forward asks the (never executed) package __init__ for: the version helpers of mmrotate/__init__.py:9-25,
    # with an mmcv version inside the range that file asserts (1.5.3 ... 1.8.0) and != 1.4.5
    root = sys.modules['mmrotate']
    root.digit_version = lambda v: [int(x) for x in v.split('.') if x.isdigit()]
    root.mmcv_version = root.digit_version('1.7.0')
    ops.RiRoIAlignRotated = type('RiRoIAlignRotated', (), {})
    # the detector file; its plotting / torchvision / cv2 names
    syn = imp('mmrotate.models.detectors.syn_images_generator_v2')
    for name in ('mmrotate.models.detectors.single_stage', 'mmrotate.models.detectors.base'):
        if name not in sys.modules:
            m = L._pkg(name, stub=True)
            m.RotatedSingleStageDetector = m.BaseDetector = type('Base', (nn.Module,), {})
    dab = sys.modules['mmrotate.models.detectors.data_augument_bank']
    sys.modules['mmcv'].ConfigDict = dict
    det = imp('mmrotate.models.detectors.rotated_fcos_teacher_student')

    def fill_poly(mask, polys, value):
        for q in polys:
            m = R.fill_convex_quads(np.asarray(q).reshape(1, 8), mask.shape[0], mask.shape[1])
            mask[m == 1] = value
    syn.nms_rotated = nms_rotated
    syn.cv2 = types.SimpleNamespace(fillPoly=fill_poly)
    syn.TF = types.SimpleNamespace(rotate=lambda img, angle, **kw: O.rotate_image_nearest(img, angle))
    syn.rbbox_overlaps = calc.rbbox_overlaps
    det.rbbox_overlaps = calc.rbbox_overlaps
    hd = imp('mmrotate.models.dense_heads.rotated_fcos_head_p2rb_ts')
    hd.rbbox_overlaps = calc.rbbox_overlaps
    # the oriented head was written against an upstream mmdet whose _expand_onehot_labels takes `ignore_index` (:1270); the HBB
    # fork's copy (the only mmdet source in the reference) has the older three-argument signature - same one-hot (gen_golden_obb.py)
    hbb_onehot = hd._expand_onehot_labels
    hd._expand_onehot_labels = lambda l, w, c, ignore_index=None: hbb_onehot(l, w, c)
    return det, syn, hd, dab, captured


class FakeDetector(nn.Module):
    """What RotatedFCOS_TS needs of `RotatedFCOS_Student`: `.extract_feat` and `.bbox_head`; the head is the reference's."""

    def __init__(self, sd, head, train):
        super().__init__()
        self.names = [k for k in sd if not k.startswith('bbox_head.') and M.is_param(k)]
        self.params = nn.ParameterList([nn.Parameter(sd[k].clone(), requires_grad=train and O.trainable_obb(k)) for k in self.names])
        self.bufs = {k: v.clone() for k, v in sd.items() if not k.startswith('bbox_head.') and not M.is_param(k)}
        self.bbox_head = head
        missing = head.load_state_dict({k[len('bbox_head.'):]: v for k, v in sd.items() if k.startswith('bbox_head.')}, strict=True)
        if not train:
            for p in head.parameters():
                p.requires_grad = False

    def sd(self):
        d = dict(zip(self.names, self.params))
        d.update(self.bufs)
        return d

    def extract_feat(self, img):
        return (O.extract_feat(self.sd(), img),)

    def named_grads(self):
        out = {n: p.grad for n, p in zip(self.names, self.params)}
        out.update({'bbox_head.' + n: p.grad for n, p in self.bbox_head.named_parameters()})
        return out


def condition_state(sd):
    """Trained-like magnitudes (tests/test_obb_parity.py `_build` applies the same to the product): regression deltas small,
    a positive distance / angle bias, damped class logits and spread instance logits, so that no top-3 of 25 bag scores is
    decided by rounding (ref_obb.mil_select_margin is asserted on the run below)."""
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


def build_reference_model(det, syn, mc, seed_s, seed_t):
    from mmrotate.models.builder import build_head
    hc = dict(mc['_model_']['bbox_head'])
    hc.update(train_cfg=mc['train_cfg'], test_cfg=mc['test_cfg'])
    TS = det.RotatedFCOS_TS
    m = TS.__new__(TS)
    nn.Module.__init__(m)
    m.student = FakeDetector(condition_state(O.init_detector_state_obb(seed_s, mc['num_stages'])), build_head(dict(hc)), True)
    m.teacher = FakeDetector(condition_state(O.init_detector_state_obb(seed_t, mc['num_stages'])), build_head(dict(hc)), False)
    m.train_cfg, m.test_cfg = mc['train_cfg'], mc['test_cfg']
    m.angle_version = mc['angle_version']
    m.gt_bboxes_point, m.refined_gt_bboxes_point = {}, {}                # :78-80
    m.count, m.ema_alpha = 0, mc['ema_alpha']
    m.epoch, m.epoch_dict, m.max_epoch = 0, {}, 12
    m.lamda = mc['lamda']
    m.pattern, m.prior_size = syn.load_basic_shape(mc['shape_list'])
    m.scale_ratio, m.filter_score, m.burn_in_step, m.alpha = 1.0, mc['filter_score'], mc['burn_in_step'], mc['alpha']
    m.num_stages, m.num_refine = mc['num_stages'], 500
    m.num_training_burninstep1, m.num_training_burninstep2 = mc['num_training_burninstep1'], mc['num_training_burninstep2']
    m._point_ = mc['_point_']
    m.fine_proposal_cfg = list(mc['train_cfg']['fine_proposal_cfg'])
    m.fine_proposal_extensive_cfg = list(mc['train_cfg']['fine_proposal_extensive_cfg'])
    return m


def make_inputs(seed, size, counts):
    """Normalised pixels (the SODA-A config subtracts the mean, sodaa_fcos_pointteacher_1x.py:185-186) and oriented boxes off
    the symmetric tie positions of the stride-8 grid."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(len(counts), 3, size, size, generator=g)
    boxes, labels = [], []
    for n in counts:
        c = torch.rand(n, 2, generator=g) * (size - 64) + 32 + 0.37
        wh = torch.exp(torch.randn(n, 2, generator=g) * 0.4 + np.log(14.0)).clamp(5, 36)
        a = (torch.rand(n, 1, generator=g) - 0.5) * np.pi * 0.98
        boxes.append(torch.cat([c, wh, a], 1))
        labels.append(torch.randint(0, 9, (n,), generator=g))
    return img, boxes, labels


class ObbDrawLog(DrawLog):
    """+ np.random.randint (the rotation angle of strong_augmentation, syn_images_generator_v2.py:263)."""

    def __enter__(self):
        super().__enter__()
        self._np_randint = np.random.randint

        def n_randint(lo, hi=None, size=None):
            v = int(self.np.randint(lo, hi))
            self.log.append((self._caller(), 'np.randint', v))
            return v
        np.random.randint = n_randint
        return self

    def __exit__(self, *exc):
        np.random.randint = self._np_randint
        super().__exit__(*exc)


def parse_iteration(log, counts, phase1):
    """One iteration's draw log -> the product's injection layout (RotatedFCOS_TS._inject)."""
    inj = {}
    B = len(counts)
    if phase1:
        cls = log.take('synthesis_single')            # torch.randint of genrate_syn :421
        bp = log.take('generate_black_paper')
        assert len(cls) == B
        syn, p = [], 0
        for i in range(B):
            G = counts[i]
            d = dict(cls=cls[i][1], scale=bp[p][1]); p += 1
            assert d['scale'].shape[0] == G
            for n in ('x', 'y', 'wn', 'rn', 'a', 'boost', 'itv', 'itv2', 'dev'):
                d[n] = torch.zeros(G)
            adj = 2
            for j in range(G):
                assert bp[p][0] == 'rand' and bp[p][1].numel() == 2
                d['x'][j], d['y'][j] = bp[p][1]; p += 1
                assert bp[p][0] == 'randn'; d['wn'][j] = bp[p][1][0]; p += 1
                assert bp[p][0] == 'randn'; d['rn'][j] = bp[p][1][0]; p += 1
                assert bp[p][0] == 'rand'; d['a'][j] = bp[p][1][0]; p += 1
                assert bp[p][0] == 'np.random'; d['boost'][j] = bp[p][1]; p += 1
                if d['boost'][j] < 0.2 and adj > 0:   # :674
                    adj -= 1
                    u_itv, u_dev = bp[p][1][0], bp[p + 1][1][0]; p += 2
                    d['itv'][j] = d['itv2'][j] = u_itv
                    d['dev'][j] = u_dev
            syn.append(d)
        assert p == len(bp), (p, len(bp))
        inj['syn'] = syn
    d = log.take('gen_negative_proposals')
    assert len(d) == 5 * B, len(d)
    inj['neg0'] = torch.stack([torch.stack([d[5 * i + j][1] for j in range(5)]) for i in range(B)])
    d = log.take('strong_augmentation')
    assert len(d) == 3 * B and d[0][0] == 'choice' and d[1][0] == 'np.randint' and d[2][0] == 'np.uniform', [e[0] for e in d]
    inj['aug'] = ([d[3 * i][1] for i in range(B)], [int(d[3 * i + 1][1]) for i in range(B)],
                  [float(np.around(d[3 * i + 2][1], 1)) for i in range(B)])
    assert not log.log, [e[:2] for e in log.log]
    return inj


def run(size=256, counts=(11, 8), seed=300):
    import point_teacher_amd as pta                 # only its config reader: the reference's OWN config file is parsed
    cfg = pta.Config.fromfile(os.path.join(L.OBB, 'configs', 'point teacher', 'sodaa_fcos_pointteacher_1x.py'))
    cfg.model['burn_in_step'] = 0                   # iteration 0 = burn-in step 1, iteration 1 = step 2
    mc = cfg.model.to_dict() if hasattr(cfg.model, 'to_dict') else dict(cfg.model)
    det, syn, hd, dab, captured = install_obb_iteration()
    torch.manual_seed(seed)
    model = build_reference_model(det, syn, mc, seed_s=seed + 1, seed_t=seed + 2)
    img, boxes, labels = make_inputs(seed + 3, size, counts)
    metas = [dict(ori_filename=f'g{i}.png', filename=f'g{i}.png', img_shape=(size, size, 3)) for i in range(len(counts))]
    out = dict(size=np.array(size), seed=np.array(seed), counts=np.array(counts), in_img_sum=img.double().sum(), in_img_probe=img[:, :, ::37, ::41].clone(),   # (pixels are make_inputs(seed + 3): regenerated by the tests)
               weight_probe=np.array([float(model.student.sd()['backbone.layer3.2.conv2.weight'].sum()),
                                      float(model.teacher.bbox_head.shared_fcs_bag[0][0].weight[5, :64].sum()),
                                      float(model.student.bbox_head.fc_ins[0].weight.abs().sum())]))
    for i, (b, l) in enumerate(zip(boxes, labels)):
        out[f'in_gt_bboxes{i}'], out[f'in_gt_labels{i}'] = b, l
    real_syn = model.genrate_syn

    def spy_syn(*a, **k):                           # `.cpu()` copies, as it does for a device tensor (gen_golden_iter.py)
        real_cpu = torch.Tensor.cpu
        torch.Tensor.cpu = lambda self, *aa, **kk: self.clone()
        try:
            r = real_syn(*a, **k)
        finally:
            torch.Tensor.cpu = real_cpu
        captured['syn_out'] = r
        return r
    model.genrate_syn = spy_syn
    head = model.student.bbox_head
    real_sel = head.mil_bag_selection

    def spy_sel(*a, **k):
        r = real_sel(*a, **k)
        captured.setdefault('merged', []).append(torch.cat([t.detach() for t in r]))
        return r
    head.mil_bag_selection = spy_sel
    for it in range(2):
        for p in model.student.parameters():
            p.grad = None
        captured.clear()
        with ObbDrawLog(seed + 10 + it) as log:
            losses = model.forward_train(img.clone(), metas, [b.clone() for b in boxes], [l.clone() for l in labels])
        inj = parse_iteration(log, counts, phase1=(it == 0))
        total = sum(v.mean() for k, v in losses.items() if 'loss' in k)       # base.py _parse_losses
        total.backward()
        tag = f'it{it}_'
        for k, v in losses.items():
            out[tag + 'out_' + k] = v.detach().mean()
        out[tag + 'out_loss'] = total.detach()
        out[tag + 'keys'] = np.array(list(losses.keys()))
        grads = model.student.named_grads()
        for k in GRAD_KEYS:
            s, stride = grad_sample(grads[k])
            out[tag + 'grad_' + k] = s
            out[tag + 'gradnorm_' + k] = grads[k].double().norm().float()
        for i in range(len(counts)):
            out[tag + f'out_points{i}'] = model.refined_gt_bboxes_point[metas[i]['ori_filename']].detach()
        out[tag + 'out_merged'] = torch.cat(captured['merged'])              # the MIL stage's merged boxes (top-3 of 25)
        if 'syn' in inj:
            for i, d in enumerate(inj['syn']):
                for n, v in d.items():
                    out[tag + f'draw_syn{i}_{n}'] = v
            for i, k in enumerate(captured['keep']):
                out[tag + f'out_syn_keep{i}'] = k
            img_syn, _, syn_boxes = captured['syn_out']
            for i in range(len(counts)):
                out[tag + f'out_syn_boxes{i}'] = syn_boxes[i]
                assert float(img_syn[i].max()) == float(img[i].max())          # painted with the image maximum (:722)
                out[tag + f'out_syn_mask{i}'] = np.packbits((img_syn[i, 0] != img[i, 0]).numpy() | ((img_syn[i] == img_syn[i].max()).all(0)).numpy(), axis=-1)
        out[tag + 'draw_neg0'] = inj['neg0']
        out[tag + 'aug_flips'] = np.array(inj['aug'][0])
        out[tag + 'aug_angles'] = np.array(inj['aug'][1], np.int64)
        out[tag + 'aug_scales'] = np.array(inj['aug'][2], np.float64)
        print(f'  obb iteration {it}:', {k: round(float(v.mean()), 5) for k, v in losses.items()})
    assert model.count == 2
    path = os.path.join(OUT, 'ref_iter_obb.npz')
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    print(f'  wrote {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB)')


if __name__ == '__main__':
    run()
