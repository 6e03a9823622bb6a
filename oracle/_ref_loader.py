"""Loader that imports individual Point-Teacher reference files BY PATH from
/root/reference so that golden vectors can be generated from the reference's
own arithmetic.

TEST INFRASTRUCTURE ONLY.  This module is used by ``oracle/gen_golden.py`` in
the build container.  Nothing here (and nothing under /root/reference) travels
to the GPU box: ``-m gpu`` tests, ``smoke()`` and ``bench.py`` never import it.

How it works
------------
* ``mmdet`` and its sub-packages are registered in ``sys.modules`` as *empty*
  package objects whose ``__path__`` points at the reference directories, so the
  heavy ``__init__.py`` files (which need mmcv-full, cv2, pycocotools ...) never
  execute, while ``import mmdet.core.bbox.transforms`` still finds the real file.
* ``mmcv`` is not installed.  The reference only uses it on this path for
  plumbing (identity decorators, the registry, ``Scale``); those few names are
  provided below.  No arithmetic of the reference is replaced: every number in
  the goldens comes out of the reference's own python/torch statements.
  ``mmcv.ops`` (RoIAlign, nms_rotated, CUDA focal loss) is NOT provided - the
  functions that need it are "parity unpinned" (see DESIGN.md).
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get('PT_REFERENCE_ROOT', '/root/reference')
HBB = os.path.join(REF_ROOT, 'HBB_TOD')


def reference_available():
    return os.path.isdir(os.path.join(HBB, 'mmdet'))


class _Stub(types.ModuleType):
    """A module whose unknown attributes resolve to ``None`` - used for names the
    reference imports at module-import time but never calls on this path
    (cv2, torchvision, mmcv.ops, test mixins)."""

    def __getattr__(self, item):
        if item.startswith('__'):
            raise AttributeError(item)
        return None


def _pkg(name, path=None, stub=False):
    m = (_Stub if stub else types.ModuleType)(name)
    m.__path__ = [path] if path else []
    m.__package__ = name
    sys.modules[name] = m
    parent, _, child = name.rpartition('.')
    if parent:
        setattr(sys.modules[parent], child, m)
    return m


class _Registry:
    """The registration/lookup behaviour of mmcv.utils.Registry that the
    reference relies on: ``@REG.register_module()`` and ``REG.get(name)``."""

    def __init__(self, name, build_func=None, parent=None, scope=None):
        self._name = name
        self._module_dict = {}

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            self._module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def build(self, cfg, **kw):
        return _build_from_cfg(cfg, self, kw.get('default_args'))


def _build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop('type')
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f'{typ} is not in the {registry._name} registry')
    return cls(**args)


def _identity_decorator(*dargs, **dkw):
    # covers @force_fp32(apply_to=...), @auto_fp16(), @mmcv.jit(coderize=True)
    if len(dargs) == 1 and callable(dargs[0]) and not dkw:
        return dargs[0]

    def deco(f):
        return f
    return deco


def install():
    """Install the shims and the empty ``mmdet`` package tree (idempotent)."""
    if 'mmdet' in sys.modules and getattr(sys.modules['mmdet'], '_pt_shim', False):
        return
    if not reference_available():
        raise RuntimeError(f'reference tree not found under {REF_ROOT}')
    import torch
    import torch.nn as nn

    # ---- mmcv plumbing ---------------------------------------------------
    mmcv = _pkg('mmcv')
    mmcv.jit = _identity_decorator
    utils = _pkg('mmcv.utils')
    utils.Registry = _Registry
    utils.build_from_cfg = _build_from_cfg
    runner = _pkg('mmcv.runner')
    runner.force_fp32 = _identity_decorator
    runner.auto_fp16 = _identity_decorator

    class BaseModule(nn.Module):
        """mmcv.runner.BaseModule: an nn.Module whose constructor takes `init_cfg` (weight-init bookkeeping only)."""

        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg
    runner.BaseModule = BaseModule
    runner.OptimizerHook = object
    cnn = _pkg('mmcv.cnn')

    class Scale(nn.Module):
        def __init__(self, scale=1.0):
            super().__init__()
            self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

        def forward(self, x):
            return x * self.scale
    cnn.Scale = Scale

    class ConvModule(nn.Module):
        """mmcv.cnn.ConvModule for the cases this path builds: Conv2d -> (GN | BN) -> ReLU, bias='auto' = no bias
        under a norm layer, sub-module names `conv` / `gn` / `bn` as in mmcv (so checkpoints keep their keys).  A
        composition of torch modules - no arithmetic of the reference is restated."""

        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias='auto',
                     conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True, **kw):
            super().__init__()
            assert conv_cfg is None, 'plain convolutions only'
            self.with_norm = norm_cfg is not None
            if bias == 'auto':
                bias = not self.with_norm
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias=bias)
            self.norm_name = None
            if self.with_norm:
                if norm_cfg['type'] == 'GN':
                    self.gn, self.norm_name = nn.GroupNorm(norm_cfg['num_groups'], out_channels), 'gn'
                else:
                    assert norm_cfg['type'] == 'BN'
                    self.bn, self.norm_name = nn.BatchNorm2d(out_channels), 'bn'
            self.with_activation = act_cfg is not None
            assert act_cfg is None or act_cfg['type'] == 'ReLU'

        def forward(self, x):
            x = self.conv(x)
            if self.norm_name:
                x = getattr(self, self.norm_name)(x)
            return torch.relu(x) if self.with_activation else x
    cnn.ConvModule = ConvModule
    cnn.MODELS = _Registry('model')
    # arithmetic-free stubs: every name resolves to None (never called here)
    _pkg('mmcv.ops', stub=True)
    _pkg('mmcv.ops.nms', stub=True)
    _pkg('cv2', stub=True)
    _pkg('torchvision', stub=True)
    _pkg('torchvision.transforms', stub=True)
    _pkg('torchvision.transforms.functional', stub=True)

    # ---- empty mmdet package tree pointing at the reference dirs ----------
    md = os.path.join(HBB, 'mmdet')
    root = _pkg('mmdet', md)
    root._pt_shim = True
    _pkg('mmdet.utils', os.path.join(md, 'utils'), stub=True)
    _pkg('mmdet.core', os.path.join(md, 'core'), stub=True)
    for sub in [ 'core/bbox', 'core/bbox/iou_calculators',
                'core/bbox/match_costs', 'core/bbox/assigners',
                'core/bbox/coder', 'core/utils', 'core/post_processing',
                'models', 'models/losses', 'models/dense_heads',
                'models/detectors', 'models/utils']:
        _pkg('mmdet.' + sub.replace('/', '.'), os.path.join(md, sub))

    imp = importlib.import_module
    sys.modules['mmdet.utils'].util_mixins = imp('mmdet.utils.util_mixins')

    # core.bbox: real files, then re-export the names the path imports
    bb = sys.modules['mmdet.core.bbox']
    builder = imp('mmdet.core.bbox.builder')
    tr = imp('mmdet.core.bbox.transforms')
    ioub = imp('mmdet.core.bbox.iou_calculators.builder')
    iou2d = imp('mmdet.core.bbox.iou_calculators.iou2d_calculator')
    ic = sys.modules['mmdet.core.bbox.iou_calculators']
    ic.bbox_overlaps = iou2d.bbox_overlaps
    ic.build_iou_calculator = ioub.build_iou_calculator
    ic.BboxOverlaps2D = iou2d.BboxOverlaps2D
    mcb = imp('mmdet.core.bbox.match_costs.builder')
    mc = imp('mmdet.core.bbox.match_costs.match_cost')
    mcp = sys.modules['mmdet.core.bbox.match_costs']
    mcp.build_match_cost = mcb.build_match_cost
    for n in ('PointCost', 'FocalLossCost', 'InsiderCost'):
        setattr(mcp, n, getattr(mc, n))
    imp('mmdet.core.bbox.assigners.assign_result')
    imp('mmdet.core.bbox.assigners.base_assigner')
    ta = imp('mmdet.core.bbox.assigners.topk_assigner')
    fa = imp('mmdet.core.bbox.assigners.fuse_topk_assigner')
    imp('mmdet.core.bbox.coder.base_bbox_coder')
    dc = imp('mmdet.core.bbox.coder.delta_xywh_bbox_coder')
    for n in ('bbox2roi', 'distance2bbox', 'bbox2distance',
              'bbox_cxcywh_to_xyxy', 'bbox_xyxy_to_cxcywh', 'bbox2result'):
        setattr(bb, n, getattr(tr, n))
    bb.build_assigner = builder.build_assigner
    bb.build_bbox_coder = builder.build_bbox_coder
    bb.build_sampler = builder.build_sampler
    bb.bbox_overlaps = iou2d.bbox_overlaps

    # core.utils: misc.py imports mask structures -> give it an empty one
    _pkg('mmdet.core.mask', stub=True)
    _pkg('mmdet.core.mask.structures', stub=True)
    _pkg('mmdet.utils.contextmanagers', stub=True)
    misc = imp('mmdet.core.utils.misc')
    du = imp('mmdet.core.utils.dist_utils')
    core = sys.modules['mmdet.core']
    for n in ('bbox2roi', 'distance2bbox', 'bbox2distance',
              'bbox_cxcywh_to_xyxy', 'bbox_xyxy_to_cxcywh', 'bbox2result'):
        setattr(core, n, getattr(tr, n))
    core.multi_apply = misc.multi_apply
    core.reduce_mean = du.reduce_mean
    core.build_assigner = builder.build_assigner
    core.build_bbox_coder = builder.build_bbox_coder
    core.build_sampler = builder.build_sampler
    core.bbox_overlaps = iou2d.bbox_overlaps
    core.multiclass_nms = None      # needs mmcv.ops.batched_nms: unpinned

    # models: builder + losses + the head + the generator
    mb = imp('mmdet.models.builder')
    mu = sys.modules['mmdet.models.utils']
    mu.build_linear_layer = lambda cfg, *a, **k: nn.Linear(*a, **k)
    lu = imp('mmdet.models.losses.utils')
    ls = sys.modules['mmdet.models.losses']
    ls.weight_reduce_loss = lu.weight_reduce_loss
    ls.weighted_loss = lu.weighted_loss
    ls.reduce_loss = lu.reduce_loss
    imp('mmdet.models.losses.cross_entropy_loss')
    imp('mmdet.models.losses.focal_loss')
    imp('mmdet.models.losses.iou_loss')
    imp('mmdet.models.losses.smooth_l1_loss')
    imp('mmdet.models.detectors.data_augument_bank')
    imp('mmdet.models.detectors.syn_images_generator_v2')
    # anchor_free_head pulls a mixin + base head
    imp('mmdet.models.dense_heads.base_dense_head')
    imp('mmdet.models.dense_heads.dense_test_mixins')
    imp('mmdet.models.dense_heads.anchor_free_head')
    imp('mmdet.models.dense_heads.fcos_head_p2b_ts')


def ref(name):
    """Return a loaded reference module, e.g. ref('models.losses.iou_loss')."""
    install()
    return importlib.import_module('mmdet.' + name)


# ------------------------------------------------------------------------------
# Oriented-box tree (OBB_TOD/mmrotate): same technique.  The OBB tree imports the HBB fork's
# `mmdet` for assigners / match costs / coders / losses, so `install()` runs first.
# ------------------------------------------------------------------------------
OBB = os.path.join(REF_ROOT, 'OBB_TOD')


def install_obb():
    """Register an empty `mmrotate` package tree over OBB_TOD/mmrotate and load the files of the
    oriented path whose arithmetic is pure python/torch (idempotent)."""
    install()
    if 'mmrotate' in sys.modules and getattr(sys.modules['mmrotate'], '_pt_shim', False):
        return
    import torch.nn as nn
    mr = os.path.join(OBB, 'mmrotate')
    root = _pkg('mmrotate', mr)
    root._pt_shim = True
    core = _pkg('mmrotate.core', os.path.join(mr, 'core'), stub=True)
    for sub in ['core/bbox', 'core/bbox/coder', 'models', 'models/losses', 'models/dense_heads', 'models/detectors']:
        _pkg('mmrotate.' + sub.replace('/', '.'), os.path.join(mr, sub))
    _pkg('mmrotate.core.bbox.iou_calculators', stub=True)          # rbbox_overlaps -> mmcv.ops (absent)
    _pkg('mmrotate.core.visualization', stub=True)
    _pkg('mmrotate.core.visualization.palette', stub=True)
    _pkg('mmrotate.models.detectors.data_augument_bank', stub=True)  # plotting helpers (matplotlib/PIL)
    for n in ('matplotlib', 'matplotlib.pyplot', 'matplotlib.patches', 'matplotlib.collections', 'PIL'):
        if n not in sys.modules:
            _pkg(n, stub=True)
    # names the HBB shim did not need
    mmcv_cnn = sys.modules['mmcv.cnn']
    assert getattr(mmcv_cnn, 'ConvModule', None) is not None
    _pkg('mmdet.core.anchor', stub=True)
    _pkg('mmdet.core.anchor.point_generator', stub=True)
    _pkg('mmdet.models.roi_heads', stub=True)
    _pkg('mmdet.models.roi_heads.bbox_heads', stub=True)
    _pkg('mmdet.models.roi_heads.bbox_heads.bbox_head', stub=True)
    _pkg('mmdet.core.visualization', stub=True)
    _pkg('mmdet.core.visualization.image', stub=True)
    sys.modules['mmdet.core'].BaseBBoxCoder = importlib.import_module('mmdet.core.bbox.coder.base_bbox_coder').BaseBBoxCoder
    sys.modules['mmdet.models.dense_heads'].AnchorFreeHead = importlib.import_module(
        'mmdet.models.dense_heads.anchor_free_head').AnchorFreeHead
    ls = sys.modules['mmdet.models.losses']
    ls.accuracy = None
    imp = importlib.import_module
    tr = imp('mmrotate.core.bbox.transforms')
    bld = imp('mmrotate.core.bbox.builder')
    imp('mmrotate.core.bbox.coder.distance_angle_point_coder')
    core.build_bbox_coder = bld.build_bbox_coder
    core.build_assigner = bld.build_assigner
    core.build_sampler = bld.build_sampler
    core.obb2xyxy = tr.obb2xyxy
    core.rbbox2result = tr.rbbox2result
    core.rbbox2roi = tr.rbbox2roi
    imp('mmrotate.models.builder')
    imp('mmrotate.models.detectors.syn_images_generator_v2')
    imp('mmrotate.models.dense_heads.rotated_anchor_free_head')
    imp('mmrotate.models.dense_heads.rotated_fcos_head_p2rb_ts')


def install_obb_eval(iou_fn):
    """Load OBB_TOD/mmrotate/core/evaluation/eval_map.py with mmcv's box_iou_rotated replaced by `iou_fn`
    (the rotated IoU itself is parity-unpinned; everything else in the file is the reference's own numpy) and a
    serial stand-in for the multiprocessing pool (spawned workers would not see the shims)."""
    install_obb()
    import numpy as np
    if 'terminaltables' not in sys.modules:
        _pkg('terminaltables', stub=True)
    _pkg('mmdet.core.evaluation', os.path.join(HBB, 'mmdet', 'core', 'evaluation'))
    for n in ('mmdet.core.evaluation.bbox_overlaps', 'mmdet.core.evaluation.class_names'):
        _pkg(n, stub=True)
    sys.modules['mmcv.utils'].print_log = lambda *a, **k: None
    mean_ap = importlib.import_module('mmdet.core.evaluation.mean_ap')
    sys.modules['mmdet.core'].average_precision = mean_ap.average_precision
    sys.modules['mmcv.ops'].box_iou_rotated = iou_fn
    _pkg('mmrotate.core.evaluation', os.path.join(OBB, 'mmrotate', 'core', 'evaluation'))
    em = importlib.import_module('mmrotate.core.evaluation.eval_map')
    em.box_iou_rotated = iou_fn

    class _SerialPool:
        def starmap(self, f, it):
            return [f(*a) for a in it]

        def close(self):
            pass
    em.get_context = lambda kind: type('Ctx', (), {'Pool': staticmethod(lambda n: _SerialPool())})()
    em.print_map_summary = lambda *a, **k: None
    return em


def ref_obb(name):
    """Return a loaded OBB reference module, e.g. ref_obb('core.bbox.transforms')."""
    install_obb()
    return importlib.import_module('mmrotate.' + name)


def install_pipeline():
    """Load the HOST LOGIC of the data pipeline by path: the HBB transforms / samplers / COCO
    reader and the OBB RResize / RRandomFlip.  mmcv's image functions (imrescale, imflip, imnormalize, impad - cv2
    underneath, neither installed) are NOT provided: only the methods that never touch pixels are called by
    oracle/gen_golden_pipeline.py (box scaling / flipping, scale sampling, annotation parsing, index sampling)."""
    install_obb()
    if getattr(sys.modules['mmdet'], '_pt_pipeline', False):
        return
    sys.modules['mmdet']._pt_pipeline = True
    mmcv = sys.modules['mmcv']
    mmcv.is_list_of = lambda seq, typ: isinstance(seq, list) and all(isinstance(x, typ) for x in seq)   # type check only
    mmcv.is_str = lambda x: isinstance(x, str)
    sys.modules['mmcv.runner'].get_dist_info = lambda: (0, 1)
    sys.modules['mmcv.utils'].print_log = lambda *a, **k: None
    for n in ('terminaltables', 'aitodpycocotools', 'aitodpycocotools.mask', 'aitodpycocotools.coco',
              'aitodpycocotools.cocoeval', 'mmcv.parallel'):
        if n not in sys.modules:
            _pkg(n, stub=True)
    md = os.path.join(HBB, 'mmdet')
    if 'mmdet.core.evaluation' not in sys.modules:
        _pkg('mmdet.core.evaluation', os.path.join(md, 'core', 'evaluation'))
    sys.modules.pop('mmdet.core.evaluation.bbox_overlaps', None)
    importlib.import_module('mmdet.core.evaluation.bbox_overlaps')            # pure numpy
    _pkg('mmdet.datasets', os.path.join(md, 'datasets'))
    bld = _pkg('mmdet.datasets.builder')
    bld.PIPELINES, bld.DATASETS = _Registry('pipeline'), _Registry('dataset')
    _pkg('mmdet.datasets.pipelines', os.path.join(md, 'datasets', 'pipelines'))
    _pkg('mmdet.datasets.samplers', os.path.join(md, 'datasets', 'samplers'))
    _pkg('mmdet.datasets.api_wrappers', stub=True)
    tr = importlib.import_module('mmdet.datasets.pipelines.transforms')
    # mmrotate's transforms were written against a newer mmdet: class placeholders for the two names this tree lacks
    for n in ('Mosaic', 'RandomCrop'):
        if not hasattr(tr, n):
            setattr(tr, n, type(n, (), {}))
    importlib.import_module('mmdet.datasets.samplers.group_sampler')
    sys.modules['mmdet.datasets.pipelines'].Compose = importlib.import_module('mmdet.datasets.pipelines.compose').Compose
    mr = os.path.join(OBB, 'mmrotate')
    _pkg('mmrotate.datasets', os.path.join(mr, 'datasets'))
    rb = _pkg('mmrotate.datasets.builder')
    rb.ROTATED_PIPELINES, rb.ROTATED_DATASETS = bld.PIPELINES, bld.DATASETS
    _pkg('mmrotate.datasets.pipelines', os.path.join(mr, 'datasets', 'pipelines'))
    core = sys.modules['mmrotate.core']
    rtr = importlib.import_module('mmrotate.core.bbox.transforms')
    core.norm_angle, core.obb2poly_np, core.poly2obb_np = rtr.norm_angle, rtr.obb2poly_np, rtr.poly2obb_np
    importlib.import_module('mmrotate.datasets.pipelines.transforms')
