"""Registry / build_from_cfg / Config work-alikes: the plugin surface the reference's
configs address (mmcv.utils.Registry as used by HBB_TOD/mmdet/models/builder.py:6-14,
core/bbox/builder.py:3-5, core/bbox/match_costs/builder.py:3).

The same registry NAMES and constructor keywords resolve here, so
`configs/point_teacher/*.py` of the reference load unchanged (their `_base_` entries are
absolute paths on the author's machine and are remapped, see Config.fromfile).
"""
import copy
import os
import re


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def __contains__(self, key):
        return key in self._module_dict

    def __len__(self):
        return len(self._module_dict)

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if not force and key in self._module_dict:
                raise KeyError(f'{key} is already registered in {self._name}')
            self._module_dict[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    """mmcv.utils.build_from_cfg semantics: `type` selects the class, the rest are kwargs."""
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg and not (default_args and 'type' in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}')
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif isinstance(obj_type, type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    try:
        return obj_cls(**args)
    except Exception as e:
        raise type(e)(f'{obj_cls.__name__}: {e}')


# one MODELS registry under six names, exactly as mmdet/models/builder.py:6-14
MODELS = Registry('models')
BACKBONES = NECKS = ROI_EXTRACTORS = SHARED_HEADS = HEADS = LOSSES = DETECTORS = MODELS
BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
BBOX_CODERS = Registry('bbox_coder')
MATCH_COST = Registry('Match Cost')
IOU_CALCULATORS = Registry('IoU calculator')


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_roi_extractor(cfg):
    return ROI_EXTRACTORS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet/models/builder.py:49-59"""
    assert cfg.get('train_cfg') is None or train_cfg is None, 'train_cfg specified in both outer field and model field'
    assert cfg.get('test_cfg') is None or test_cfg is None, 'test_cfg specified in both outer field and model field'
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_bbox_coder(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


def build_match_cost(cfg, default_args=None):
    return build_from_cfg(cfg, MATCH_COST, default_args)


def build_iou_calculator(cfg, default_args=None):
    return build_from_cfg(cfg, IOU_CALCULATORS, default_args)


# --------------------------------------------------------------------------- Config
class ConfigDict(dict):
    """dict with attribute access (addict-style, as mmcv.ConfigDict)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _to_cfgdict(x):
    if isinstance(x, dict):
        return ConfigDict({k: _to_cfgdict(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return type(x)(_to_cfgdict(v) for v in x)
    return x


DELETE_KEY = '_delete_'
BASE_KEY = '_base_'


def _merge_a_into_b(a, b):
    """mmcv Config._merge_a_into_b: child `a` overrides base `b`; `_delete_=True` replaces."""
    b = dict(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and not v.get(DELETE_KEY, False):
            if not isinstance(b[k], dict):
                raise TypeError(f'{k}={v} in child config cannot inherit from base because {k} is a dict in the '
                                f'child config but is of type {type(b[k])} in base config.')
            b[k] = _merge_a_into_b(v, b[k])
        elif isinstance(v, dict):
            v = dict(v)
            v.pop(DELETE_KEY, None)
            b[k] = v
        else:
            b[k] = v
    return b


class Config:
    """`Config.fromfile(path)` -> attribute-style config, python-file configs with `_base_`
    inheritance and `_delete_` (the subset of mmcv.Config the Point-Teacher configs use).

    `_base_` entries of the reference are absolute paths on the author's machine
    (`/home/zhr/mmdet-rfla/configs/_base_/...`, OBB: `/home/zhr/SODA/configs/_base_/...`);
    anything containing `/configs/_base_/` that does not exist is looked up under
    `<config root>/_base_/` of the file being loaded, then under this package's configs/.
    """

    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', _to_cfgdict(cfg_dict or {}))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def _file2dict(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        with open(filename, 'r', encoding='utf-8') as f:
            src = f.read()
        ns = {'__file__': filename}
        exec(compile(src, filename, 'exec'), ns)            # configs are python, as in mmcv
        cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v)
               and not isinstance(v, type(os))}
        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            base = base if isinstance(base, (list, tuple)) else [base]
            merged = {}
            for b in base:
                bd = Config._file2dict(Config._resolve_base(b, filename))
                dup = merged.keys() & bd.keys()
                if dup:
                    raise KeyError(f'Duplicate key is not allowed among bases: {dup}')
                merged.update(bd)
            cfg = _merge_a_into_b(cfg, merged)
        return cfg

    @staticmethod
    def _resolve_base(b, filename):
        here = os.path.dirname(filename)
        cand = b if os.path.isabs(b) else os.path.join(here, b)
        if os.path.isfile(cand):
            return cand
        m = re.search(r'/configs/_base_/(.*)$', b.replace('\\', '/'))
        if m:
            rel = m.group(1)
            d = here
            for _ in range(4):                               # walk up to the configs/ root
                p = os.path.join(d, '_base_', rel)
                if os.path.isfile(p):
                    return p
                d = os.path.dirname(d)
            p = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'configs', '_base_', rel)
            if os.path.isfile(p):
                return p
        raise FileNotFoundError(f'_base_ entry {b} of {filename} cannot be resolved')

    @staticmethod
    def fromfile(filename):
        return Config(Config._file2dict(filename), filename=filename)

    def merge_from_dict(self, options):
        """--cfg-options style overrides: {'a.b.c': v}."""
        d = {}
        for full, v in options.items():
            cur = d
            keys = full.split('.')
            for k in keys[:-1]:
                cur = cur.setdefault(k, {})
            cur[keys[-1]] = v
        object.__setattr__(self, '_cfg_dict', _to_cfgdict(_merge_a_into_b(d, self._cfg_dict)))

    def to_dict(self):
        def plain(x):
            if isinstance(x, dict):
                return {k: plain(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return type(x)(plain(v) for v in x)
            return x
        return plain(self._cfg_dict)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, k, default=None):
        return self._cfg_dict.get(k, default)

    def __repr__(self):
        return f'Config (path: {self.filename}): {dict(self._cfg_dict)!r}'
