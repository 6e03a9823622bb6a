#!/usr/bin/env python3
"""Whole-iteration goldens: the reference's OWN `TS_P2B_FCOS.forward_train` (both burn-in phases, the step-1
rectangle generator, the MIL stage, `update_points`, strong augmentation, every loss) run on the CPU in this
container, on seeded weights and inputs, with every random draw logged.  TEST INFRASTRUCTURE - runs only here
(/root/reference is read by path, nothing of it is copied); writes tests/golden/ref_iter_*.npz.

    python oracle/gen_golden_iter.py            # ref_iter_p0.npz, ref_iter_p100.npz

What runs from the reference, unmodified (files under /root/reference/HBB_TOD/mmdet):
  models/detectors/fcos_p2b_teacher_student.py  forward_train :116-139, forward_train_burn_in_step1 :141-211,
      forward_train_burn_in_step2 :213-252, update_teacher_model :254-257, update_points :266-274,
      forward_mil_head_burn_in_step1/2 :365-466, genrate_syn :469-502, genrate_points :504-519
  models/detectors/syn_images_generator_v2.py   generate_black_paper :591-690, strong_augmentation :24-132,
      MIL_gen_proposals_from_cfg :134-145, gen_negative_proposals :234-259, fine_proposals_from_cfg :262-324
  models/dense_heads/fcos_head_p2b_ts.py        a REAL `TS_P2BFCOSHead` instance built by its own constructor from the
      config's `bbox_head` dictionary: forward, get_pseudo_bbox, loss, loss_pseudo, MIL_head_burn_in_step1/2, ...
  models/roi_heads/roi_extractors/*             SingleRoIExtractor (real), core/bbox/** assigners, costs, coders,
  models/losses/**                              FocalLoss (CPU branch), DIoU / DN-DIoU, BCE
What is NOT the reference (absent from this image) and is bound to the oracle's restatement instead:
  mmcv.ops.RoIAlign -> oracle.ref_model.roi_align_vec, mmcv.ops.nms_rotated -> oracle.ref_ops.nms_rotated,
  cv2.fillPoly -> oracle.ref_ops.fill_convex_quads (all three "parity unpinned"), mmcv's ConvModule / BaseModule
  (compositions of torch modules, _ref_loader.py) and the ResNet-50 / FPN / PSAGG modules, which need mmcv to
  instantiate and are compositions of torch convolutions: `oracle.ref_model.extract_feat` on the same weights.
The draws (python `random`, numpy, CPU torch) are made by seeded generators through logging wrappers and stored
in the layout of the product's injection hooks (`TS_P2B_FCOS._inject`).
One unspecified behaviour is fixed for the run: `torch.topk` may return tied elements in any order, and on the stride-8
grid the 5th / 6th L1-nearest points of a generic point tie structurally (8+a+b is reached from two neighbours; about
half of them tie exactly in float32), so the FUSETopkAssigner's candidate set (fuse_topk_assigner.py:97) is not a
function of its inputs.  The assigners' `torch.topk` calls are served by a STABLE top-k (lowest index among ties) -
one of the results torch may legally return, and the rule the oracle and the kernels implement.
"""
import importlib
import os
import random
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
from oracle import ref_model as M  # noqa: E402
from oracle import ref_ops as R  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
torch.set_num_threads(8)
MIL_LOGIT_SCALE = 0.1          # ref_model.damp_mil_logits: moderate bag logits, as in a trained head

GRAD_KEYS = ['bbox_head.conv_cls.weight', 'bbox_head.conv_reg.weight', 'bbox_head.conv_centerness.weight',
             'bbox_head.cls_convs.0.conv.weight', 'bbox_head.reg_convs.3.conv.weight', 'bbox_head.scales.0.scale',
             'bbox_head.fc_cls.0.weight', 'bbox_head.fc_ins.0.weight', 'bbox_head.fc_reg.0.weight',
             'bbox_head.shared_fcs_bag.0.1.weight', 'bbox_head.shared_fcs_reg.0.0.weight',
             'neck_agg.lateral_convs.4.conv.weight', 'neck.fpn_convs.0.conv.weight', 'neck.lateral_convs.2.conv.bias',
             'backbone.layer4.2.conv3.weight', 'backbone.layer3.0.downsample.0.weight', 'backbone.layer2.0.conv1.weight']


def grad_sample(g, cap=4096):
    """What the golden keeps of one gradient tensor: everything when small, else every `stride`-th element."""
    flat = g.reshape(-1)
    stride = max(1, -(-flat.numel() // cap))
    return flat[::stride].clone(), stride


# ------------------------------------------------------------------ draws --
class DrawLog:
    """Seeded generators behind torch.rand / randn / randint, np.random.random / uniform and random.choice; every
    call is logged with the name of the calling reference function."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.np = np.random.RandomState(seed)
        self.py = random.Random(seed)
        self.log = []
        self._saved = None

    def _caller(self):
        return sys._getframe(2).f_code.co_name

    def __enter__(self):
        self._saved = (torch.rand, torch.randn, torch.randint, np.random.random, np.random.uniform, random.choice, torch.topk)
        rand, randn, randint = self._saved[:3]

        def stable_topk(x, k, dim=-1, largest=True, sorted=True):
            v, i = torch.sort(x, dim=dim, descending=largest, stable=True)
            return v.narrow(dim, 0, k), i.narrow(dim, 0, k)
        torch.topk = stable_topk

        def t_rand(*size, device=None, **kw):
            size = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
            v = rand(*size, generator=self.g)
            self.log.append((self._caller(), 'rand', v.clone()))
            return v

        def t_randn(*size, device=None, **kw):
            v = randn(*size, generator=self.g)
            self.log.append((self._caller(), 'randn', v.clone()))
            return v

        def t_randint(lo, hi, size, **kw):
            v = randint(lo, hi, size, generator=self.g)
            self.log.append((self._caller(), 'randint', v.clone()))
            return v

        def n_random():
            v = float(self.np.random_sample())
            self.log.append((self._caller(), 'np.random', v))
            return v

        def n_uniform(a, b):
            v = float(self.np.uniform(a, b))
            self.log.append((self._caller(), 'np.uniform', v))
            return v

        def p_choice(seq):
            v = self.py.choice(seq)
            self.log.append((self._caller(), 'choice', v))
            return v
        torch.rand, torch.randn, torch.randint = t_rand, t_randn, t_randint
        np.random.random, np.random.uniform, random.choice = n_random, n_uniform, p_choice
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn, torch.randint, np.random.random, np.random.uniform, random.choice, torch.topk = self._saved

    def take(self, caller):
        out = [(k, v) for c, k, v in self.log if c == caller]
        self.log = [e for e in self.log if e[0] != caller]
        return out


def parse_iteration(log, counts, dense_n, n_neg, first_visit, phase1):
    """Turn one iteration's draw log into the product's injection layout (TS_P2B_FCOS._inject)."""
    inj = {}
    B = len(counts)
    if first_visit:                                   # genrate_points :504-519 -> random_point_in_quadrilateral
        d = log.take('random_point_in_quadrilateral')
        assert len(d) == 2 * B
        inj['point_u'] = [(d[2 * i][1], d[2 * i + 1][1]) for i in range(B)]
    if phase1:
        cls = log.take('synthesis_single')            # torch.randint of genrate_syn :473
        bp = log.take('generate_black_paper')
        assert len(cls) == B
        syn, p = [], 0
        for i in range(B):
            G = counts[i]
            d = dict(cls=cls[i][1], scale=bp[p][1]); p += 1
            assert d['scale'].shape[0] == G
            names = ('x', 'y', 'wn', 'rn', 'a', 'boost', 'itv', 'itv2', 'dev')
            for n in names:
                d[n] = torch.zeros(G)
            adj = 2
            for j in range(G):
                assert bp[p][0] == 'rand' and bp[p][1].numel() == 2
                d['x'][j], d['y'][j] = bp[p][1]; p += 1
                assert bp[p][0] == 'randn'; d['wn'][j] = bp[p][1][0]; p += 1
                assert bp[p][0] == 'randn'; d['rn'][j] = bp[p][1][0]; p += 1
                assert bp[p][0] == 'rand'; d['a'][j] = bp[p][1][0]; p += 1
                assert bp[p][0] == 'np.random'; d['boost'][j] = bp[p][1]; p += 1
                if d['boost'][j] < 0.2 and adj > 0:   # :640
                    adj -= 1
                    u_itv, u_dev = bp[p][1][0], bp[p + 1][1][0]; p += 2
                    d['itv'][j] = d['itv2'][j] = u_itv
                    d['dev'][j] = u_dev
            syn.append(d)
        assert p == len(bp), (p, len(bp))
        inj['syn'] = syn
    d = log.take('gen_negative_proposals')
    assert len(d) == 4 * B, 'an image ended without synthetic rectangles (the reference then drops the MIL branch, :368-372): pick another seed'
    inj['neg0'] = torch.stack([torch.stack([d[4 * i + j][1] for j in range(4)]) for i in range(B)])
    d = log.take('strong_augmentation')
    assert len(d) == 2 * B
    inj['aug'] = ([d[2 * i][1] for i in range(B)], [float(np.around(d[2 * i + 1][1], 1)) for i in range(B)])
    assert not log.log, [e[:2] for e in log.log]
    return inj


# ------------------------------------------------------- reference objects --
class OracleRoIAlign(nn.Module):
    """Stands where mmcv.ops.RoIAlign would (absent; parity unpinned): the oracle's restatement."""

    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True, use_torchvision=False):
        super().__init__()
        assert sampling_ratio == 0 and pool_mode == 'avg' and aligned
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale = spatial_scale

    def forward(self, feat, rois):
        return M.roi_align_vec(feat, rois, self.output_size[0], self.spatial_scale)


def install_iteration():
    L.install()
    mmcv_ops = sys.modules['mmcv.ops']
    mmcv_ops.RoIAlign = OracleRoIAlign
    L._pkg('mmdet.core.visualization', stub=True)
    md = os.path.join(L.HBB, 'mmdet', 'models')
    L._pkg('mmdet.models.roi_heads', os.path.join(md, 'roi_heads'))
    L._pkg('mmdet.models.roi_heads.roi_extractors', os.path.join(md, 'roi_heads', 'roi_extractors'))
    importlib.import_module('mmdet.models.roi_heads.roi_extractors.base_roi_extractor')
    importlib.import_module('mmdet.models.roi_heads.roi_extractors.single_level_roi_extractor')
    det = importlib.import_module('mmdet.models.detectors.fcos_p2b_teacher_student')
    syn = importlib.import_module('mmdet.models.detectors.syn_images_generator_v2')
    captured = {}

    def nms_rotated(dets, scores, thr):
        keep = R.nms_rotated(dets, scores, thr)
        captured.setdefault('keep', []).append(keep.clone())
        return None, keep

    def fill_poly(mask, polys, value):
        for q in polys:
            m = R.fill_convex_quads(np.asarray(q).reshape(1, 8), mask.shape[0], mask.shape[1])
            mask[m == 1] = value
    syn.nms_rotated = nms_rotated
    syn.cv2 = types.SimpleNamespace(fillPoly=fill_poly)
    return det, syn, captured


class FakeDetector(nn.Module):
    """What TS_P2B_FCOS needs of `Student_FCOS` (fcos_student.py:9-50): `.extract_feat` and `.bbox_head`.  The head
    is the reference's; backbone + necks are the oracle's functional torch composition over these parameters."""

    def __init__(self, sd, head, train):
        super().__init__()
        self.names = [k for k in sd if not k.startswith('bbox_head.') and M.is_param(k)]
        self.params = nn.ParameterList([nn.Parameter(sd[k].clone(), requires_grad=train and M.trainable(k)) for k in self.names])
        self.bufs = {k: v.clone() for k, v in sd.items() if not k.startswith('bbox_head.') and not M.is_param(k)}
        self.bbox_head = head
        head.load_state_dict({k[len('bbox_head.'):]: v for k, v in sd.items() if k.startswith('bbox_head.')}, strict=True)
        if not train:
            for p in head.parameters():
                p.requires_grad = False

    def sd(self):
        d = dict(zip(self.names, self.params))
        d.update(self.bufs)
        return d

    def extract_feat(self, img):
        return (M.extract_feat(self.sd(), img),)

    def named_grads(self):
        out = {n: p.grad for n, p in zip(self.names, self.params)}
        out.update({'bbox_head.' + n: p.grad for n, p in self.bbox_head.named_parameters()})
        return out


def build_reference_model(det, cfg, seed_s, seed_t):
    from mmdet.models.builder import build_head
    mc = cfg.model.to_dict() if hasattr(cfg.model, 'to_dict') else dict(cfg.model)
    hc = dict(mc['_model_']['bbox_head'])
    hc.update(train_cfg=mc['train_cfg'], test_cfg=mc['test_cfg'])
    TS = det.TS_P2B_FCOS
    m = TS.__new__(TS)
    nn.Module.__init__(m)
    m.student = FakeDetector(M.damp_mil_logits(M.init_detector_state(seed_s, mc['num_stages']), MIL_LOGIT_SCALE), build_head(dict(hc)), True)
    m.teacher = FakeDetector(M.damp_mil_logits(M.init_detector_state(seed_t, mc['num_stages']), MIL_LOGIT_SCALE), build_head(dict(hc)), False)
    m.train_cfg, m.test_cfg = mc['train_cfg'], mc['test_cfg']
    m.gt_bboxes_point, m.refined_gt_bboxes_point = {}, {}                # :66-88
    m.count, m.ema_alpha = 0, mc['ema_alpha']
    m.epoch, m.epoch_dict, m.max_epoch = 0, {}, 12
    m.lamda = mc['lamda']
    syn = importlib.import_module('mmdet.models.detectors.syn_images_generator_v2')
    m.pattern, m.prior_size = syn.load_basic_shape(mc['shape_list'])
    m.scale_ratio, m.filter_score, m.burn_in_step, m.alpha = 1.0, mc['filter_score'], mc['burn_in_step'], mc['alpha']
    m.num_stages, m.num_refine = mc['num_stages'], 500
    m.num_training_burninstep1, m.num_training_burninstep2 = mc['num_training_burninstep1'], mc['num_training_burninstep2']
    m._point_ = mc['_point_']
    m.fine_proposal_cfg = list(mc['train_cfg']['fine_proposal_cfg'])
    m.fine_proposal_extensive_cfg = list(mc['train_cfg']['fine_proposal_extensive_cfg'])
    return m


def make_inputs(seed, size, counts):
    g = torch.Generator().manual_seed(seed)
    img = (torch.rand(len(counts), 3, size, size, generator=g) * 90 + 60).round()      # < 255: painted pixels are identifiable
    boxes, labels = [], []
    for n in counts:
        c = torch.rand(n, 2, generator=g) * (size - 48) + 24 + 0.37       # off the symmetric tie positions of the stride-8 grid
        wh = torch.exp(torch.randn(n, 2, generator=g) * 0.4 + np.log(14.0)).clamp(4, 40)
        boxes.append(torch.cat([c - wh / 2, c + wh / 2], 1))
        labels.append(torch.randint(0, 8, (n,), generator=g))
    return img, boxes, labels


def run(percent, size, counts, seed):
    import point_teacher_amd as pta                 # only its config reader: the reference's OWN config file is parsed
    cfg = pta.Config.fromfile(os.path.join(L.HBB, 'configs', 'point_teacher', f'aitodv2_point_teacher_{percent}%.py'))
    cfg.model['burn_in_step'] = 0                   # iteration 0 = burn-in step 1, iteration 1 = step 2 (:133)
    det, syn, captured = install_iteration()
    torch.manual_seed(seed)
    model = build_reference_model(det, cfg, seed_s=seed + 1, seed_t=seed + 2)
    img, boxes, labels = make_inputs(seed + 3, size, counts)
    metas = [dict(ori_filename=f'g{i}.png', img_shape=(size, size, 3)) for i in range(len(counts))]
    out = dict(percent=np.array(percent), size=np.array(size), seed=np.array(seed), in_img=img.to(torch.uint8),
               mil_logit_scale=np.array(MIL_LOGIT_SCALE),
               weight_probe=np.array([float(model.student.sd()['backbone.layer3.2.conv2.weight'].sum()),
                                      float(model.teacher.bbox_head.shared_fcs_bag[0][0].weight[5, :64].sum())]))
    for i, (b, l) in enumerate(zip(boxes, labels)):
        out[f'in_gt_bboxes{i}'], out[f'in_gt_labels{i}'] = b, l
    dense_n = int(len(model.pattern) / 2)
    real_syn = model.genrate_syn

    def spy_syn(*a, **k):
        # genrate_syn paints a HOST COPY of each image (`img.cpu()`, :481-482; the model lives on a GPU there).  Run on
        # host tensors `.cpu()` would alias the training image and the rectangles would leak into the teacher / MIL /
        # augmentation inputs, so for the duration of the call `.cpu()` copies, as it does for a device tensor.
        real_cpu = torch.Tensor.cpu
        torch.Tensor.cpu = lambda self, *aa, **kk: self.clone()
        try:
            r = real_syn(*a, **k)
        finally:
            torch.Tensor.cpu = real_cpu
        captured['syn_out'] = r
        return r
    model.genrate_syn = spy_syn
    for it in range(2):
        for p in model.student.parameters():
            p.grad = None
        captured.clear()
        with DrawLog(seed + 10 + it) as log:
            losses = model.forward_train(img.clone(), metas, [b.clone() for b in boxes], [l.clone() for l in labels])
        inj = parse_iteration(log, counts, dense_n, 200, first_visit=(it == 0), phase1=(it == 0))
        total = sum(v.mean() for k, v in losses.items() if 'loss' in k)       # base.py:197-198
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
        out[tag + 'grad_total_norm'] = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values() if g is not None)).float()
        for i in range(len(counts)):
            out[tag + f'out_points{i}'] = model.refined_gt_bboxes_point[metas[i]['ori_filename']].detach()
        # the injected draws, in the product's layout
        if 'point_u' in inj:
            for i, (ux, uy) in enumerate(inj['point_u']):
                out[tag + f'draw_point_u{i}'] = torch.stack([ux, uy])
        if 'syn' in inj:
            for i, d in enumerate(inj['syn']):
                for n, v in d.items():
                    out[tag + f'draw_syn{i}_{n}'] = v
            for i, k in enumerate(captured['keep']):
                out[tag + f'out_syn_keep{i}'] = k                     # indices into [real objects | rectangles] that survive the NMS
            img_syn, _, syn_boxes = captured['syn_out']
            for i in range(len(counts)):
                out[tag + f'out_syn_boxes{i}'] = syn_boxes[i]
            assert float(img.max()) < 255 and torch.equal(img, out['in_img'].float())       # the input image stayed intact
            out[tag + 'out_syn_mask'] = np.packbits((img_syn[:, 0] == 255).numpy(), axis=-1)
            assert bool(((img_syn == 255) == (img_syn[:, :1] == 255)).all()) and bool((img_syn[img_syn != 255] == img[img_syn != 255]).all())
        out[tag + 'draw_neg0'] = inj['neg0']
        out[tag + 'aug_flips'] = np.array(inj['aug'][0])
        out[tag + 'aug_scales'] = np.array(inj['aug'][1], np.float64)
        print(f'  p{percent} iteration {it}:', {k: round(float(v.mean()), 5) for k, v in losses.items()})
    assert model.count == 2
    path = os.path.join(OUT, f'ref_iter_p{percent}.npz')
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    print(f'  wrote {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB)')


if __name__ == '__main__':
    which = sys.argv[1:] or ['0', '100']
    if '0' in which:
        run(0, 256, (12, 9), seed=100)
    if '100' in which:
        run(100, 256, (8, 6), seed=200)
