"""Student_FCOS and TS_P2B_FCOS - the teacher->student detector of Point-Teacher, mirroring
/root/reference/HBB_TOD/mmdet/models/detectors/{fcos_student,fcos_p2b_teacher_student,
single_stage,base}.py.  Same registry names, constructor keywords, `train_step` /
`forward(return_loss=...)` contract and per-iteration order of operations (Appendix A of
SURVEY.md); the per-image python loops, CPU round trips and host synchronisations of the
reference are replaced by batched launches into libpt_hip.so."""
from collections import OrderedDict

import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import functional as F
from . import planes as PL
from .core import bbox2result, bbox_overlaps, bbox_xyxy_to_cxcywh, mean0
from .proposals import (MIL_gen_proposals_from_cfg, gen_negative_proposals, generate_black_paper_batch,
                        load_basic_shape, random_point_in_quadrilateral, strong_augmentation_images,
                        strong_augmentation_masked)
from .nn_modules import refresh_bn_affines
def img_is_cuda(t):
    return torch.is_tensor(t) and t.is_cuda


from .registry import DETECTORS, build_backbone, build_detector, build_head, build_neck


def _first_images(t, B):
    """The first B images of a batched stem output: a view of a tensor, or - plane activations (planes.PlaneAct, the plane-native
    frozen stage) - a copy of their rows behind a zero row of their own (the 3 x 3 taps of the consumer read it)."""
    if isinstance(t, PL.PlaneAct):
        return PL.first_images(t, B)
    return t[:B]


class LazyLogVars(OrderedDict):
    """log_vars whose values stay on the device until somebody reads them.  The reference
    calls `.item()` on ~12 scalars every iteration (base.py:201-207); here that cost is paid
    only when a logger actually looks (every `log_config.interval` iterations)."""

    def materialize(self):
        keys = list(self.keys())
        if not keys:
            return {}
        vals = torch.stack([super(LazyLogVars, self).__getitem__(k).detach().float().reshape(()) for k in keys])
        if dist.is_available() and dist.is_initialized():
            vals = vals / dist.get_world_size()
            dist.all_reduce(vals)                      # ONE coalesced all-reduce instead of one per key
        host = vals.cpu().tolist()
        out = OrderedDict(zip(keys, host))
        from . import planes as PL
        if PL.CENSUS.mode >= 2:            # PT_F16_CENSUS=2, the debugging census of the fp16 planes: per site saturated / sub-0.125
            PL.CENSUS.poll(sync=True)      # elements and the largest stored magnitude since the last read (this rank's)
            out.update(PL.CENSUS.log_vars())
        return out

    def __getitem__(self, k):
        v = super().__getitem__(k)
        return v.item() if isinstance(v, torch.Tensor) else v


class BaseDetector(nn.Module):
    """detectors/base.py: forward dispatch, _parse_losses, train_step."""

    def __init__(self, init_cfg=None):
        super().__init__()
        self.fp16_enabled = False

    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        """base.py:156-173"""
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def forward_test(self, imgs, img_metas, **kwargs):
        if isinstance(imgs, torch.Tensor):
            imgs, img_metas = [imgs], [img_metas]
        assert len(imgs) == 1, 'test-time augmentation is not on the Point-Teacher path'
        return self.simple_test(imgs[0], img_metas[0], **kwargs)

    def _parse_losses(self, losses):
        """base.py:175-208: every key containing 'loss' is summed; all keys are logged."""
        log_vars = LazyLogVars()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value if value.dim() == 0 else value.mean()      # (.mean() of a scalar is one more launch)
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        terms = [v for k, v in OrderedDict.items(log_vars) if 'loss' in k]
        # (python's sum() is one launch per term - it starts with `0 + term` -; one stack + one sum here.  The summation order of the
        #  total differs from the reference's left-to-right one by rounding only; every term's gradient is 1 either way)
        loss = terms[0] if len(terms) == 1 else torch.stack([t.reshape(()) for t in terms]).sum()
        log_vars['loss'] = loss
        return loss, log_vars

    def train_step(self, data, optimizer=None):
        """base.py:210-243"""
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))


@DETECTORS.register_module()
class Student_FCOS(BaseDetector):
    """detectors/fcos_student.py:8-50 over single_stage.py:10-60 (backbone -> neck -> neck_agg -> head)."""

    def __init__(self, backbone, neck=None, neck_agg=None, bbox_head=None, roi_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        self.backbone = build_backbone(backbone)
        if neck is not None:
            self.neck = build_neck(neck)
        if neck_agg is not None:
            self.neck_agg = build_neck(neck_agg)
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.roi_head = build_head(roi_head) if roi_head is not None else None
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    @property
    def with_neck_agg(self):
        return hasattr(self, 'neck_agg') and self.neck_agg is not None

    backbone_autocast = None      # torch.bfloat16 = BASELINE configs[2]: bf16 backbone / necks, fp32 head (set by the Trainer)

    def _plane_trunk(self):
        """The trainable stages hand their outputs to the neck as split planes (planes.PlaneAct): no fp32 copy in between."""
        return (self.with_neck and getattr(self.neck, 'accepts_planes', False) and getattr(self.backbone, 'plane_capable', False)
                and self.backbone_autocast in (None, torch.bfloat16))

    def backbone_stem(self, img):
        """The frozen stem of the backbone (ResNet.forward_stem) under this detector's autocast setting."""
        if self.backbone_autocast is None:
            return self.backbone.forward_stem(img, planes=True) if self._plane_trunk() else self.backbone.forward_stem(img)
        with torch.autocast('cuda', dtype=self.backbone_autocast):
            return self.backbone.forward_stem(img)

    def extract_feat(self, img, stem=None):
        """`stem`: the output of `backbone_stem` for `img` (then `img` is not read)."""
        def run():
            if self._plane_trunk():
                x = self.backbone(img, planes=True) if stem is None else self.backbone(None, stem=stem, planes=True)
            else:
                x = self.backbone(img) if stem is None else self.backbone(None, stem=stem)
            if self.with_neck:
                x = self.neck(x)
            if self.with_neck_agg:
                x = self.neck_agg(x)
            return x
        if self.backbone_autocast is None:
            return run()
        with torch.autocast('cuda', dtype=self.backbone_autocast):
            x = run()
        return tuple(f.float() for f in x)          # the dense head, the MIL head and every loss run in fp32

    def forward_dummy(self, img):
        return self.bbox_head(self.extract_feat(img))

    def simple_test(self, img, img_metas, rescale=False):
        res = self.bbox_head.simple_test(self.extract_feat(img), img_metas, rescale=rescale)
        return [bbox2result(b, l, self.bbox_head.num_classes) for b, l in res]


@DETECTORS.register_module()
class TS_P2B_FCOS(BaseDetector):
    """detectors/fcos_p2b_teacher_student.py:36-519."""

    # Box-format hooks: everything below that depends on xyxy boxes goes through these, so that the
    # oriented twin (obb_detectors.RotatedFCOS_TS) only swaps them.
    _epoch_key = 'ori_filename'
    _mil_gen = staticmethod(MIL_gen_proposals_from_cfg)
    _neg_gen = staticmethod(gen_negative_proposals)

    @staticmethod
    def _aligned_iou(a, b):
        return bbox_overlaps(a, b, is_aligned=True)

    @staticmethod
    def _cxcywh(boxes):
        return bbox_xyxy_to_cxcywh(boxes)

    def _strong_aug_images(self, img):
        return strong_augmentation_images(img, params=self._inject.get('aug'))

    def _strong_aug(self, img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes, params=None, imgs=None):
        return strong_augmentation_masked(img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes,
                                          params=params if params is not None else self._inject.get('aug'), imgs=imgs)

    def _prior_on(self, dev):
        """The shape priors on the device (uploaded once; a per-iteration `.to(device)` is a blocking copy)."""
        if getattr(self, '_prior_dev', None) is None or self._prior_dev.device != dev:
            self._prior_dev = self.prior_size.to(dev)
        return self._prior_dev

    def _black_paper(self, imgs, gt_bboxes, imgsize, draws):
        return generate_black_paper_batch(imgs, gt_bboxes, self._prior_on(imgs[0].device), range(int(len(self.pattern) / 2)),
                                          imgsize, draws=draws)

    def _initial_points(self, gt_bboxes, u):
        return random_point_in_quadrilateral(gt_bboxes, self._point_, *(u if u is not None else (None, None)))

    def _to_results(self, dets, labels, num_classes):
        return bbox2result(dets, labels, num_classes)

    def __init__(self, _model_, _point_='random', num_stages=2, num_refine=500, num_training_burninstep1=512,
                 num_training_burninstep2=512, ema_alpha=0.999, filter_score=0.8, burn_in_step=10000, lamda=1.0,
                 alpha=[0.1, 1.0],
                 shape_list=[[20, 20, 0.5, 0.5], [30, 120, 0.5, 0.5], [10, 20, 0.5, 0.5], [20, 50, 0.5, 0.5],
                             [30, 20, 0.5, 0.5]],
                 MIL_head=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        self.teacher = build_detector(_model_, train_cfg, test_cfg)       # two independently initialised copies (:60-61)
        self.student = build_detector(_model_, train_cfg, test_cfg)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.gt_bboxes_point, self.refined_gt_bboxes_point = {}, {}
        self.count, self.ema_alpha = 0, ema_alpha
        self.epoch, self.epoch_dict, self.max_epoch = 0, {}, 12
        self.lamda = lamda
        self.pattern, self.prior_size = load_basic_shape(shape_list)
        self.scale_ratio, self.filter_score, self.burn_in_step, self.alpha = 1.0, filter_score, burn_in_step, alpha
        self.num_stages, self.num_refine = num_stages, num_refine
        self.num_training_burninstep1, self.num_training_burninstep2 = num_training_burninstep1, num_training_burninstep2
        self._point_ = _point_
        if self.train_cfg is not None:
            self.fine_proposal_cfg = list(self.train_cfg['fine_proposal_cfg'])
            self.fine_proposal_extensive_cfg = list(self.train_cfg['fine_proposal_extensive_cfg'])
        # Set by runtime.FlatParams: (teacher_flat, student_flat) so the EMA is ONE launch.  The
        # teacher never receives gradients (the reference defines freeze() :110-113 but never calls
        # it, which only wastes optimizer/DDP work); here its parameters are frozen explicitly.
        self._flat = None
        for p in self.teacher.parameters():
            p.requires_grad = False
        for m in self.teacher.modules():      # the EMA moves the teacher's BN affines: their fused (scale, shift) cache
            if isinstance(m, nn.BatchNorm2d):  # must follow (nn_modules._bn_key / refresh_bn_affines)
                m._affine_dynamic = True
        # hooks for tests: inject the random draws of one iteration
        self._inject = {}
        self._stem_shared = None          # decided at the first iteration (`_shared_stem`)
        # ... and again after every route that can change frozen weights: any load_state_dict into this module or a child
        # (pre-hook below), the Trainer's re-layouts, and once per epoch (`update_epoch`) for in-place edits nobody announced
        self._register_load_state_dict_pre_hook(self._forget_stem_decision)
        self.student.backbone._register_load_state_dict_pre_hook(self._forget_stem_decision)
        self.teacher.backbone._register_load_state_dict_pre_hook(self._forget_stem_decision)
        self.point_stamp = {}             # image key -> iteration of its last refinement (checkpoint merges keep the newest)
        self.teacher_stream = os.environ.get('PT_TEACHER_STREAM', '1') == '1'     # the teacher pass on a second HIP stream (_teacher_fork)
        self._side_stream = None
        self._teacher_graph = None                 # graphs.GraphedNoGrad of _teacher_trunk (PT_TEACHER_GRAPH=0 turns it off)
        self.teacher_graph_off = False

    # the reference keeps count/point dictionaries as plain attributes, so they are lost on resume
    # (SURVEY section 5); persisting them is a documented deviation.
    def _forget_stem_decision(self, *args, **kw):
        self._stem_shared = None

    def _recheck_stem_async(self):
        """Compare the frozen segments of teacher and student again WITHOUT stalling the host: the verdict lands in pinned
        memory behind an event and is read by a later `_shared_stem()` once the event has completed."""
        if not self._stem_shared or self._flat is None or not self._flat[0].is_cuda:
            return
        n = self._flat_frozen_start
        differ = (self._flat[0][n:] != self._flat[1][n:]).any().to(torch.uint8)
        host = torch.empty((), dtype=torch.uint8).pin_memory()
        host.copy_(differ, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._stem_check = (ev, host)

    def get_extra_state(self):
        return dict(count=self.count, epoch=self.epoch, point_stamp=dict(self.point_stamp),
                    gt_bboxes_point={k: v.cpu() for k, v in self.gt_bboxes_point.items()},
                    refined_gt_bboxes_point={k: v.cpu() for k, v in self.refined_gt_bboxes_point.items()})

    def set_extra_state(self, state):
        if not state:
            return
        dev = next(self.parameters()).device
        self.count, self.epoch = state.get('count', 0), state.get('epoch', 0)
        self.point_stamp = dict(state.get('point_stamp', {}))
        self.gt_bboxes_point = {k: v.to(dev) for k, v in state.get('gt_bboxes_point', {}).items()}
        self.refined_gt_bboxes_point = {k: v.to(dev) for k, v in state.get('refined_gt_bboxes_point', {}).items()}

    def extract_feat(self, img, model=None, stem=None):
        return model.extract_feat(img) if stem is None else model.extract_feat(None, stem=stem)

    def _shared_stem(self):
        """True when the teacher provably computes the same frozen stem as the student: the parameters live in the Trainer's flat
        buffers, the backbone freezes its stem (frozen_stages >= 0) and every frozen parameter and BatchNorm buffer of the two
        models is bit-equal - the state of a run whose two detectors loaded the same pretrained backbone (the configs'
        init_cfg).  Then the EMA of the frozen segment is the identity and is skipped (alpha*t + (1-alpha)*t rounds away from
        t), and the stem of the clean images is evaluated once per iteration (`_student_inputs`).  PT_SHARE_STEM=0: off."""
        chk = getattr(self, '_stem_check', None)
        if chk is not None and chk[0].query():
            self._stem_check = None
            if int(chk[1]):
                self._stem_shared = None      # the frozen weights diverged: decide again (and stop sharing)
        if self._stem_shared is None:
            ok = False
            sb = getattr(self.student, 'backbone', None)
            if (self._flat is not None and getattr(self, '_flat_frozen_start', None) is not None and hasattr(sb, 'forward_stem')
                    and getattr(sb, 'frozen_stages', -1) >= 0 and os.environ.get('PT_SHARE_STEM', '1') != '0'):
                n = self._flat_frozen_start
                ok = bool(torch.equal(self._flat[0][n:], self._flat[1][n:]))
                ok = ok and all(torch.equal(a, b) for a, b in zip(self.teacher.backbone.buffers(), sb.buffers()))
            self._stem_shared = ok
        return self._stem_shared

    def forward_dummy(self, img, model=None):
        return model.forward_dummy(img)

    def freeze(self, model):
        model.eval()
        for p in model.parameters():
            p.requires_grad = False

    # ------------------------------------------------------------------ training --
    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
        """:116-139"""
        img = img.to(torch.float)
        num_img = len(img_metas)
        self.update_teacher_model(self.teacher, self.student, self.ema_alpha)
        for det in (self.student, self.teacher):       # BatchNorm terms the weight planes fold in: current before the first convolution
            if img.is_cuda and det._plane_trunk():
                det.backbone.refresh_plane_terms()
        self.update_epoch(num_img, img_metas)
        gt_points, img_list, img = self.genrate_points(num_img, img, img_metas, gt_bboxes)
        if self.count <= self.burn_in_step:
            losses = self.forward_train_burn_in_step1(num_img, img, img_list, img_metas, gt_bboxes, gt_points, gt_labels,
                                                      gt_bboxes_ignore, img.device)
        else:
            losses = self.forward_train_burn_in_step2(num_img, img, img_list, img_metas, gt_bboxes, gt_points, gt_labels,
                                                      gt_bboxes_ignore, img.device)
        self.count += 1
        return losses

    def update_teacher_model(self, teacher_model, student_model, ema_decay=0.999):
        """:254-257: over parameters() only (buffers untouched), at the START of the iteration."""
        with torch.no_grad():
            if self._flat is not None:
                n = self._flat_frozen_start if self._shared_stem() else self._flat[0].numel()
                F.ema_update_(self._flat[0][:n], self._flat[1][:n], ema_decay)
            else:   # parameters not flattened yet (unit tests / CPU construction): per-tensor form
                F.PARAM_EPOCH[0] += 1
                for t, s in zip(teacher_model.parameters(), student_model.parameters()):
                    t.data.mul_(ema_decay).add_(s.data, alpha=1 - ema_decay)
            refresh_bn_affines(teacher_model)     # the fused BN epilogue of the teacher follows the EMA

    def update_epoch(self, num_img, img_metas):
        """:259-264"""
        if img_metas[0][self._epoch_key] in self.epoch_dict:
            self.epoch += 1
            self.epoch_dict = {}
            self._recheck_stem_async()    # in-place edits of frozen weights nobody announced: re-verified once per epoch
        for i in range(num_img):
            self.epoch_dict[img_metas[i][self._epoch_key]] = 1

    def update_points(self, num_img, img_metas, pseudo_bboxes):
        """:266-274"""
        out = []
        for i in range(num_img):
            pc = self._cxcywh(pseudo_bboxes[i])[:, :2]
            oc = self.gt_bboxes_point[img_metas[i]['ori_filename']]
            rc = (1 - self.lamda) * pc + self.lamda * oc
            out.append(rc)
            self.refined_gt_bboxes_point[img_metas[i]['ori_filename']] = rc
            self.point_stamp[img_metas[i]['ori_filename']] = self.count
        return out

    def genrate_points(self, num_img, img, img_metas, gt_bboxes):
        """:504-519"""
        gt_points, img_list = [], []
        for i in range(num_img):
            name = img_metas[i]['ori_filename']
            if name in self.refined_gt_bboxes_point:
                gt_points.append(self.refined_gt_bboxes_point[name])
            else:
                u = self._inject.get('point_u')
                pts = self._initial_points(gt_bboxes[i], u[i] if u is not None else None)
                gt_points.append(pts)
                self.gt_bboxes_point[name] = pts
            img_list.append(img[i])
        return gt_points, img_list, img

    def _teacher_trunk(self, img, stem):
        """The fixed-shape part of the teacher pass: backbone, necks and the dense head's maps (no ground truth involved)."""
        feat = self.extract_feat(img, self.teacher, stem=stem)
        return self.teacher.bbox_head(feat)

    def _teacher_graph_state(self):
        """What a captured teacher trunk depends on besides its inputs (graphs.GraphedNoGrad): the teacher itself and its mode, the
        registrations of the weight-plane cache, the persistent BatchNorm affine buffers, the fp16 groups the census demoted."""
        t = self.teacher
        buf = getattr(t, '_bn_affine_buf', None)
        return (id(t), t.training, F._CONV_W.generation, buf[0].data_ptr() if buf is not None else 0,
                tuple(sorted(PL.CENSUS.demoted)), torch.is_autocast_enabled(), getattr(t, 'backbone_autocast', None))

    def _teacher_pseudo(self, img, img_metas, img_list, gt_points, gt_labels, gt_bboxes, gt_bboxes_ignore, stem=None):
        with torch.no_grad():
            if self._teacher_graph is None:
                from .graphs import GraphedNoGrad
                self._teacher_graph = GraphedNoGrad(self._teacher_trunk, self._teacher_graph_state, name='the teacher pass')
            if self.teacher_graph_off:    # (a measurement that brackets single launches with events)
                outs = self._teacher_trunk(img, stem)
            elif stem is not None:        # (the images are not read then: keep them out of the segment's static inputs)
                outs = self._teacher_graph(None, stem)
            else:
                outs = self._teacher_graph(img, None)
            return self.teacher.bbox_head.get_pseudo_bbox(*outs, gt_points, gt_labels, gt_bboxes, self.filter_score,
                                                          img_metas, img_list, gt_bboxes_ignore)

    def _teacher_fork(self, *args, **kw):
        """Start the teacher pass on a second HIP stream (`teacher_stream`, PT_TEACHER_STREAM=0 turns it off; bench.py turns it off in the
        steps whose launches it brackets with HIP events, so that co-scheduled teacher kernels do not stretch them; -0.3 ... -0.7 ms per
        iteration): it has no gradient and nothing of the student's
        batched pass depends on it until the MIL stage, so its small-batch kernels (B = 2 at 25x25 / 50x50 leave most of the 256
        CUs idle) overlap the student's forward.  -> a callable that joins the stream and returns the pseudo boxes."""
        if not self.teacher_stream or not img_is_cuda(args[0]):
            res = self._teacher_pseudo(*args, **kw)
            return lambda: res
        main = torch.cuda.current_stream()
        F.refresh_conv_weight_planes()             # both streams read the weight planes of this parameter epoch: made here, once
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream()
        side = self._side_stream
        side.wait_stream(main)
        stem = kw.get('stem')
        if stem is not None:                       # views of a tensor the main stream allocated and will free
            for t in [stem[0]] + list(stem[1]):
                (t.t if isinstance(t, PL.PlaneAct) else t).record_stream(side)
        with torch.cuda.stream(side):
            res = self._teacher_pseudo(*args, **kw)

        def record(obj):                           # every tensor of the (nested) result is used by the main stream from here on
            if torch.is_tensor(obj):
                obj.record_stream(main)
            elif isinstance(obj, (list, tuple)):
                for o in obj:
                    record(o)
            elif isinstance(obj, dict):
                for o in obj.values():
                    record(o)

        def join():
            main.wait_stream(side)
            record(res)
            return res
        return join

    def _refined_points_distance(self, gt_points, gt_bboxes):
        real = self._cxcywh(torch.cat(gt_bboxes, dim=0))
        # sqrt(x ** 2) of the reference (:291) is |x| exactly in binary floating point (no overflow at pixel scale)
        return mean0((torch.cat(gt_points) - real[:, :2]).abs() / (real[:, 2:4] / 2).abs())

    def _student_inputs(self, img, extra=None):
        """The batch of ONE student pass, [clean images | extra (synthetic) images | strongly augmented images], and - when the
        teacher shares the student's frozen stem (`_shared_stem`) - the stem of that batch, whose first rows are the teacher's.
        The augmented PIXELS only depend on the input image and the draws (the boxes / points are transformed later, after the
        MIL stage, with the same draws).  -> (aug_pre, parts, batch, stem | None, teacher stem | None)"""
        params, aug_imgs = self._strong_aug_images(img)
        img_aug = torch.stack(aug_imgs, dim=0)
        parts = [img] + ([extra] if extra is not None else []) + [img_aug]
        if img.is_contiguous(memory_format=torch.channels_last) and not img.is_contiguous():
            parts = [p.contiguous(memory_format=torch.channels_last) for p in parts]
        batch = torch.cat(parts, dim=0)
        stem = t_stem = None
        if self._shared_stem():
            with torch.no_grad():                      # no trainable parameter, no input that requires a gradient
                stem = self.student.backbone_stem(batch)
            B = img.shape[0]
            memo = {}                                  # (the last frozen stage's output is both `stem[0]` and an entry of `stem[1]`)

            def first(o):
                if id(o) not in memo:
                    memo[id(o)] = _first_images(o, B)
                return memo[id(o)]
            t_stem = (first(stem[0]), [first(o) for o in stem[1]])
        return (params, aug_imgs), parts, batch, stem, t_stem

    def _student_passes(self, img, extra=None, inputs=None):
        """ONE batched student pass over the batch of `_student_inputs`.  The reference runs the groups as separate passes
        (:146, :191 / :226, :243); every layer of the student is per-sample (BatchNorm in eval mode, GroupNorm), so the batched
        pass computes the same features with a third of the launches and one gradient accumulation per parameter.
        Returns (aug_pre, [feature tuples per group], features of the whole batch)."""
        aug_pre, parts, batch, stem, _ = inputs if inputs is not None else self._student_inputs(img, extra)
        feat_all = self.extract_feat(batch, self.student, stem=stem)
        groups, o = [], 0
        for p in parts:
            groups.append(tuple(f[o:o + p.shape[0]] for f in feat_all))
            o += p.shape[0]
        return aug_pre, groups, feat_all

    def _student_aug_loss(self, img, img_metas, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes,
                          gt_bboxes_ignore, aug_pre=None, feat_aug=None, branches='both'):
        params, imgs = aug_pre if aug_pre is not None else (None, None)
        aug = self._strong_aug(img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes, params=params,
                               imgs=imgs)
        img_aug, img_aug_list, gp, gl, pp, pl, pb, gv, pv = aug
        outs = self.student.bbox_head(feat_aug if feat_aug is not None else self.extract_feat(img_aug, self.student),
                                      branches=branches)
        return self.student.bbox_head.loss_pseudo(*outs, gp, gl, pp, pl, pb, [None] * len(img_metas), img_metas,
                                                  img_aug_list, self.count <= self.burn_in_step, gt_bboxes_ignore,
                                                  gt_valid=gv, pseudo_valid=pv)

    def forward_train_burn_in_step2(self, num_img, img, img_list, img_metas, gt_bboxes, gt_points, gt_labels,
                                    gt_bboxes_ignore, device):
        """:213-252"""
        losses = {}
        inputs = self._student_inputs(img)
        teacher = self._teacher_fork(img, img_metas, img_list, gt_points, gt_labels, gt_bboxes, gt_bboxes_ignore,
                                     stem=inputs[4])
        aug_pre, (feat, feat_aug), _ = self._student_passes(img, inputs=inputs)
        del inputs
        pb_c, pp_c, pl_c, _, _ = teacher()
        mil_feat = self.student.bbox_head.forward_mil(feat)
        pb_r, pp_r, mil_losses = self.forward_mil_head_burn_in_step2(num_img, pb_c, pp_c, pl_c, gt_bboxes, img_metas,
                                                                     mil_feat)
        if mil_losses is not None:
            losses.update(mil_losses)
        del feat, mil_feat
        gt_points = self.update_points(num_img, img_metas, pb_r)
        losses['refined_points_distance'] = self._refined_points_distance(gt_points, gt_bboxes)
        lc, lb, lt = self._student_aug_loss(img, img_metas, gt_points, gt_labels, pp_r, pl_c, pb_r, gt_bboxes_ignore,
                                            aug_pre=aug_pre, feat_aug=feat_aug)
        losses['loss_cls'], losses['loss_bbox'], losses['loss_centerness'] = lc, lb, lt
        return losses

    def forward_mil_head_burn_in_step2(self, num_img, pseudo_bboxes, pseudo_points, pseudo_labels, gt_bboxes, img_metas,
                                       x_ori):
        """:425-466"""
        n = self.num_training_burninstep2
        losses = {}
        pb_t = [b[:n] for b in pseudo_bboxes]          # (the reference clones these, :430-433; nothing below writes into them)
        gb_t = [b[:n] for b in gt_bboxes]
        pp_t = [p[:n] for p in pseudo_points]
        pl_t = [l[:n] for l in pseudo_labels]
        refined_b = [b.clone() for b in pseudo_bboxes]
        refined_p = [p.clone() for p in pseudo_points]
        gb_cat = torch.cat(gb_t, dim=0)
        losses['coarse_bboxes_iou'] = mean0(self._aligned_iou(torch.cat(pb_t, dim=0), gb_cat))
        head = self.student.bbox_head
        for stage in range(self.num_stages):
            cfg = self.fine_proposal_cfg[stage]
            pr, pv, pref, preal = self._mil_gen(pp_t, pb_t, cfg, gb_t, img_meta=img_metas)
            neg, negw = self._neg_gen(pp_t, cfg, pr, img_meta=img_metas, uniforms=self._inject.get(f'neg{stage}'))
            mil_loss, pb_t = head.MIL_head_burn_in_step2(x_ori, img_metas, pr, pv, pref, preal, neg, negw, pb_t, pl_t,
                                                         self.fine_proposal_extensive_cfg[stage], stage)
            losses[f'stage{stage}_refine_bboxes_iou'] = mean0(self._aligned_iou(torch.cat(pb_t, dim=0), gb_cat))
            mil_loss[f'stage{stage}_loss_mil_bbox'] = mil_loss[f'stage{stage}_loss_mil_bbox'] * self.alpha[0]
            mil_loss[f'stage{stage}_loss_mil_bags'] = mil_loss[f'stage{stage}_loss_mil_bags'] * self.alpha[1]
            losses.update(mil_loss)
        for i in range(num_img):
            k = pb_t[i].shape[0]
            refined_b[i][:k] = pb_t[i]
            refined_p[i][:k] = self._cxcywh(pb_t[i])[:, :2]
        return refined_b, refined_p, losses

    # --------------------------------------------------------------- burn-in step 1 --
    def genrate_syn(self, num_img, img_list, gt_bboxes, gt_labels):
        """:469-502 on the device: (img_syn [B,C,H,W], list, syn boxes list, alive masks list)."""
        C, H, W = img_list[0].shape
        res = self._black_paper(list(img_list[:num_img]), list(gt_bboxes[:num_img]), min(H, W), self._inject.get('syn'))
        imgs, boxes, alive = [r[0] for r in res], [r[1] for r in res], [r[2] for r in res]
        return torch.stack(imgs, dim=0), imgs, boxes, alive

    def forward_train_burn_in_step1(self, num_img, img, img_list, img_metas, gt_bboxes, gt_points, gt_labels,
                                    gt_bboxes_ignore, device):
        """:141-211.  The synthetic boxes are carried as fixed-size lists + `alive` masks."""
        losses = {}
        img_syn, _, syn_boxes, syn_alive = self.genrate_syn(num_img, img_list, gt_bboxes, gt_labels)
        head = self.student.bbox_head
        inputs = self._student_inputs(img, extra=img_syn)
        teacher = self._teacher_fork(img, img_metas, img_list, gt_points, gt_labels, gt_bboxes, gt_bboxes_ignore,
                                     stem=inputs[4])
        aug_pre, (feat_ori, feat_syn, feat_aug), _ = self._student_passes(img, inputs=inputs)
        del inputs
        mil_syn, mil_ori = head.forward_mil(feat_syn), head.forward_mil(feat_ori)
        outs_syn = head(feat_syn, branches='reg')          # `loss` reads the regression / centerness outputs only
        loss_syn = head.loss(*outs_syn, syn_boxes, img_metas, gt_bboxes_ignore, gt_valid=syn_alive)
        pb_c, pp_c, pl_c, _, _ = teacher()
        _, _, mil_losses = self.forward_mil_head_burn_in_step1(num_img, syn_boxes, syn_alive, pb_c, pp_c, pl_c, gt_bboxes,
                                                               img_metas, mil_syn, mil_ori, img)
        pb_r, pp_r = pb_c, pp_c                                   # the MIL output is discarded in step 1 (:187)
        if mil_losses is not None:
            losses.update(mil_losses)
            gt_points = self.update_points(num_img, img_metas, pb_r)
            losses['refined_points_distance'] = self._refined_points_distance(gt_points, gt_bboxes)
        lc, _, _ = self._student_aug_loss(img, img_metas, gt_points, gt_labels, pp_r, pl_c, pb_r, gt_bboxes_ignore,
                                          aug_pre=aug_pre, feat_aug=feat_aug, branches='cls')
        losses['loss_cls'] = lc
        losses['loss_bbox'], losses['loss_centerness'] = loss_syn
        return losses

    def _compact_alive(self, boxes, alive, n):
        """First `n` alive rows of `boxes` without a host round trip: (compacted [n,4], weight [n]).
        Rows beyond the number of alive boxes are a harmless dummy box with weight 0."""
        out = boxes.new_zeros((n + 1, boxes.shape[1]))
        out[:, 2:4] = 8.0
        w = boxes.new_zeros(n + 1)
        if boxes.shape[0]:
            rank = torch.cumsum(alive.int(), 0) - 1
            take = alive & (rank < n)
            idx = torch.where(take, rank, torch.full_like(rank, n)).long()   # everything else -> dump slot n
            out.index_copy_(0, idx, boxes)
            w.index_copy_(0, idx, take.to(w.dtype))
        return out[:n].contiguous(), w[:n].contiguous()

    def _compact_alive_batch(self, boxes, alive, n):
        """`_compact_alive` for a [B, N, 4] stack / [B, N] masks -> ([B, n, 4], [B, n]); row for row the same values."""
        B, N = alive.shape
        rank = torch.cumsum(alive.int(), 1) - 1
        take = alive & (rank < n)
        idx = torch.where(take, rank, n).long() + torch.arange(B, device=boxes.device).view(B, 1) * (n + 1)
        out = boxes.new_zeros((B * (n + 1), boxes.shape[2]))
        out[:, 2:4] = 8.0
        w = boxes.new_zeros(B * (n + 1))
        out.index_copy_(0, idx.reshape(-1), boxes.reshape(B * N, -1))
        w.index_copy_(0, idx.reshape(-1), take.reshape(-1).to(w.dtype))
        return out.view(B, n + 1, -1)[:, :n].contiguous(), w.view(B, n + 1)[:, :n].contiguous()

    def forward_mil_head_burn_in_step1(self, num_img, synthetic_bboxes, synthetic_alive, pseudo_bboxes, pseudo_points,
                                       pseudo_labels, gt_bboxes, img_metas, x_synthetic, x_ori, img):
        """:365-423.  Deviation (documented): when an image ends with zero synthetic boxes the
        reference drops the MIL loss keys for that iteration (a DDP hazard); here
        the keys are always present and such an image simply contributes zero-weight bags."""
        n = self.num_training_burninstep1
        losses = {}
        if len({(b.shape, a.shape) for b, a in zip(synthetic_bboxes, synthetic_alive)}) == 1 and synthetic_bboxes[0].shape[0]:
            # the fixed-size lists of the batch in one pass (the same per-image arithmetic on a [B, N] stack)
            st, sw = self._compact_alive_batch(torch.stack(list(synthetic_bboxes)), torch.stack(list(synthetic_alive)), n)
            syn_t, syn_w = list(st.unbind(0)), list(sw.unbind(0))
        else:
            syn_t, syn_w = zip(*[self._compact_alive(b, a, n) for b, a in zip(synthetic_bboxes, synthetic_alive)])
            syn_t, syn_w = list(syn_t), list(syn_w)
        syn_p = [self._cxcywh(b)[:, :2] for b in syn_t]
        pb_t = [b[:n] for b in pseudo_bboxes]
        gb_t = [b[:n] for b in gt_bboxes]
        pp_t = [p[:n] for p in pseudo_points]
        pl_t = [l[:n] for l in pseudo_labels]
        refined_b = [b.clone() for b in pseudo_bboxes]
        refined_p = [p.clone() for p in pseudo_points]
        gb_cat = torch.cat(gb_t, dim=0)
        losses['coarse_bboxes_iou'] = mean0(self._aligned_iou(torch.cat(pb_t, dim=0), gb_cat))
        head = self.student.bbox_head
        for stage in range(self.num_stages):
            cfg = self.fine_proposal_cfg[stage]
            ext = self.fine_proposal_extensive_cfg[stage]
            pr, pv, pref, preal = self._mil_gen(pp_t, pb_t, cfg, gb_t, img_meta=img_metas)
            spr, spv, spref, spreal = self._mil_gen(syn_p, syn_t, cfg, syn_t, img_meta=img_metas)
            neg, negw = self._neg_gen(pp_t, cfg, pr, img_meta=img_metas, uniforms=self._inject.get(f'neg{stage}'))
            U1 = len(cfg['base_ratios']) ** 2 * (1 + 4 * len(cfg['shake_ratio'] or []))
            U2 = len(ext['base_ratios']) ** 2 * (1 + 4 * len(ext['shake_ratio'] or []))
            bag_w = [w.repeat_interleave(U1 * U2) for w in syn_w]
            mil_loss, pb_t = head.MIL_head_burn_in_step1(x_ori, x_synthetic, img_metas, pr, pv, pref, preal, spr, spv,
                                                         spref, spreal, neg, negw, syn_t, pb_t, pl_t, ext, stage,
                                                         syn_bag_weight=bag_w)
            losses[f'stage{stage}_refine_bboxes_iou'] = mean0(self._aligned_iou(torch.cat(pb_t, dim=0), gb_cat))
            mil_loss[f'stage{stage}_loss_mil_bbox'] = mil_loss[f'stage{stage}_loss_mil_bbox'] * self.alpha[0]
            mil_loss[f'stage{stage}_loss_mil_bags'] = mil_loss[f'stage{stage}_loss_mil_bags'] * self.alpha[1]
            losses.update(mil_loss)
        for i in range(num_img):
            k = pb_t[i].shape[0]
            refined_b[i][:k] = pb_t[i]
            refined_p[i][:k] = self._cxcywh(pb_t[i])[:, :2]
        return refined_b, refined_p, losses

    # ------------------------------------------------------------------- inference --
    def simple_test(self, img, img_metas, rescale=False):
        """:276-298: inference uses the TEACHER."""
        img = img.to(torch.float)
        feat = self.extract_feat(img, self.teacher)
        res = self.teacher.bbox_head.simple_test(feat, img_metas, rescale=rescale)
        return [self._to_results(b, l, self.teacher.bbox_head.num_classes) for b, l in res]
