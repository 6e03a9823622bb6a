"""Backbone / neck / RoI modules with the reference's registry names, constructor keywords
and parameter names (so mmdet checkpoints load): ResNet (caffe style), FPN, PSAGG, Scale,
RoIAlign, SingleRoIExtractor.  The convolutions are plain torch (MIOpen on ROCm) - the
contraction work belongs on the matrix cores via the vendor library; everything specific to
Point-Teacher around them is in libpt_hip.so.

Citations: /root/reference/HBB_TOD/mmdet/models/backbones/resnet.py,
necks/fpn.py, necks/ps_fpn.py, roi_heads/roi_extractors/*.py.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import functional as F
from . import planes as PL
from .registry import BACKBONES, NECKS, ROI_EXTRACTORS


_SPLIT_CONV = os.environ.get('PT_SPLIT_CONV', '1') != '0'
_STEM_TAIL = os.environ.get('PT_STEM_TAIL', '1') != '0'            # 0: bn1 + ReLU and the max-pool as two passes
_PLANE_TRUNK = _SPLIT_CONV and os.environ.get('PT_PLANE_TRUNK', '1') != '0'     # 0: the round-3 routing (3x3 only, fp32 between layers)
_PLANE_BN_TRAIN = os.environ.get('PT_PLANE_BN_TRAIN', '1') != '0'               # 0: blocks with a trainable BatchNorm keep the round-3 routing
_PLANE_STEM = _PLANE_TRUNK and os.environ.get('PT_PLANE_STEM', '1') != '0'      # 0: the frozen stages (layer1) stay library convolutions + BatchNorm passes
_GN_CL = os.environ.get('PT_GN_CL', '1') != '0'


class ModulatedDeformConv2dPack(nn.Module):
    """mmcv.ops.ModulatedDeformConv2dPack (`conv_cfg=dict(type='DCNv2')`): a zero-initialised `conv_offset` predicts
    (dy, dx) offsets and a sigmoid mask per kernel tap, `pt_deform_*` does the sampling.  Parameter names as mmcv
    (`weight`, `bias`, `conv_offset.weight`, `conv_offset.bias`)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deform_groups=1,
                 bias=True):
        super().__init__()
        assert groups == 1
        k = kernel_size if isinstance(kernel_size, tuple) else (kernel_size, kernel_size)
        self.stride, self.padding, self.dilation, self.deform_groups, self.kernel_size = stride, padding, dilation, deform_groups, k
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *k))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        n = in_channels * k[0] * k[1]
        nn.init.uniform_(self.weight, -1.0 / n ** 0.5, 1.0 / n ** 0.5)                 # mmcv ModulatedDeformConv2d.init_weights
        self.conv_offset = nn.Conv2d(in_channels, deform_groups * 3 * k[0] * k[1], k, stride=stride, padding=padding,
                                     dilation=dilation, bias=True)
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)

    def forward(self, x):
        out = self.conv_offset(x)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        return F.modulated_deform_conv2d(x, offset, torch.sigmoid(mask), self.weight, self.bias, self.stride, self.padding,
                                         self.dilation, 1, self.deform_groups)


class DeformConv2dPack(nn.Module):
    """mmcv.ops.DeformConv2dPack (`conv_cfg=dict(type='DCN')`, no bias, no mask)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deform_groups=1,
                 bias=False):
        super().__init__()
        assert groups == 1 and not bias
        k = kernel_size if isinstance(kernel_size, tuple) else (kernel_size, kernel_size)
        self.stride, self.padding, self.dilation, self.deform_groups = stride, padding, dilation, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *k))
        nn.init.kaiming_uniform_(self.weight, nonlinearity='relu')
        self.conv_offset = nn.Conv2d(in_channels, deform_groups * 2 * k[0] * k[1], k, stride=stride, padding=padding,
                                     dilation=dilation, bias=True)
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)

    def forward(self, x):
        return F.deform_conv2d(x, self.conv_offset(x), self.weight, self.stride, self.padding, self.dilation, 1,
                               self.deform_groups)


_CONV_LAYERS = {'Conv2d': nn.Conv2d, 'DCN': DeformConv2dPack, 'DCNv2': ModulatedDeformConv2dPack}


class ConvModule(nn.Module):
    """The subset of mmcv.cnn.ConvModule on this path: conv (+GroupNorm) (+ReLU).  Parameter paths
    `.conv` and `.gn` (mmcv names the norm layer after its type); bias='auto' means "no bias when a
    norm layer follows" (mmcv/cnn/bricks/conv_module.py:113-116)."""
    plane_min_pixels = int(os.environ.get('PT_PLANE_MIN_PIXELS', '2048'))     # below: the library's kernels (launch-bound shapes)
    plane_group = 'neck'             # the census / fall-back group of its fp16 operands (planes.CENSUS); the head's towers say 'towers'

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, act=True, norm_cfg=None,
                 conv_cfg=None, dilation=1):
        super().__init__()
        if bias == 'auto':
            bias = norm_cfg is None
        conv_cls = _CONV_LAYERS[(conv_cfg or dict(type='Conv2d'))['type']]           # mmcv build_conv_layer
        self.conv = conv_cls(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation, bias=bias)
        self.with_activation = act
        self.with_norm = norm_cfg is not None
        self.gn = self.bn = None
        if self.with_norm:
            # GroupNorm on the Point-Teacher path; a trainable BatchNorm in TRAINING mode (batch statistics, running averages
            # updated) in the YOLOF baseline's encoder and head (row N4): mmcv names the layer after its type, `.gn` / `.bn`
            assert norm_cfg['type'] in ('GN', 'BN'), norm_cfg
            if norm_cfg['type'] == 'GN':
                self.gn = nn.GroupNorm(norm_cfg['num_groups'], out_channels)
            else:
                self.bn = nn.BatchNorm2d(out_channels)
            for p in (self.gn or self.bn).parameters():
                p.requires_grad = norm_cfg.get('requires_grad', True)
            if self.gn is None:
                del self.gn
            else:
                del self.bn

    def forward(self, x, out_planes=False):
        c = self.conv
        if isinstance(x, PL.PlaneAct) or out_planes:
            if (_PLANE_TRUNK and not self.with_norm and PL.plane_conv_ok(c) and (isinstance(x, PL.PlaneAct) or PL.dense_ok(x))):
                return PL.conv_module(x, c, relu=self.with_activation, out_planes=out_planes, group=self.plane_group)
            if isinstance(x, PL.PlaneAct):
                x = x.tensor()
            assert not out_planes, 'this convolution cannot emit planes'
        if (_PLANE_TRUNK and not self.with_norm and PL.plane_conv_ok(c) and PL.dense_ok(x)
                and x.shape[0] * x.shape[2] * x.shape[3] >= self.plane_min_pixels):
            # 1x1 / 3x3 convolutions of necks and towers (fp32 by the config): implicit GEMM on the bf16 matrix cores with split-bf16
            # operands and fp32 accumulation, bias (+ ReLU) in its epilogue, input / weight / bias gradients on the same kernels
            # (fp16 x 2 operands and three products unless the group fell back: 4-byte planes for the byte-bound 1x1s as well)
            return PL.conv_module(x, c, relu=self.with_activation, f16=PL.use_f16(self.plane_group), group=self.plane_group)
        if (_SPLIT_CONV and type(c) is nn.Conv2d and c.bias is not None and not self.with_norm and self.with_activation
                and F.split_conv3x3_ok(x, c)):
            # the dense head's tower convolutions (fp32 by the config): implicit GEMM on the bf16 matrix cores with split-bf16
            # operands and fp32 accumulation, bias + ReLU in its epilogue (csrc/gemm_split.hip).  PT_SPLIT_CONV=0: library.
            return F.split_conv3x3(x, c.weight, c.bias, True)
        if (_SPLIT_CONV and type(c) is nn.Conv2d and c.bias is not None and not self.with_norm and not self.with_activation
                and F.split_conv3x3_ok(x, c)):
            return F.split_conv3x3(x, c.weight, c.bias, False)             # FPN output convolutions (no activation)
        if (type(c) is nn.Conv2d and c.bias is not None and not self.with_norm and self.with_activation and x.is_cuda
                and not torch.is_autocast_enabled()):
            y = TF.conv2d(x, c.weight, None, c.stride, c.padding, c.dilation, c.groups)
            if F.bias_relu_ok(y, c.bias):
                return F.bias_relu_(y, c.bias, True)          # bias + ReLU in one pass; backward: grad_x + bias gradient in one
            return TF.relu(y + c.bias.view(1, -1, 1, 1), inplace=True)
        if (_PLANE_TRUNK and self.with_norm and c.bias is None and PL.plane_conv_ok(c) and PL.dense_ok(x) and PL.use_f16(self.plane_group)
                and x.shape[0] * x.shape[2] * x.shape[3] >= self.plane_min_pixels):
            # towers with GroupNorm (config 5), round 5: the convolution alone on fp16 x 2 operands / three MFMA products (fp32 map in,
            # split once, fp32 map out; input / weight gradients on the same kernels), then the norm
            x = PL.conv_module(x, c, relu=False, f16=True, group=self.plane_group)
        elif _SPLIT_CONV and type(c) is nn.Conv2d and c.bias is None and self.with_norm and F.split_conv3x3_ok(x, c):
            x = F.split_conv3x3(x, c.weight, None, False)     # ... on bf16 x 3 operands / six products (the group fell back, PT_F16_FC=0)
        else:
            x = c(x)
        if self.with_norm:
            if hasattr(self, 'gn') and _GN_CL and F.group_norm_cl_ok(x, self.gn):
                # GroupNorm + ReLU on the channels_last activation itself (csrc/group_norm.hip): no NCHW round trip, and the next
                # tower convolution still sees channels_last.  PT_GN_CL=0: torch.nn.GroupNorm.
                return F.group_norm_cl(x, self.gn, self.with_activation)
            x = self.gn(x) if hasattr(self, 'gn') else self.bn(x)
        return TF.relu(x, inplace=True) if self.with_activation else x


class Scale(nn.Module):
    """mmcv.cnn.Scale"""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


# ------------------------------------------------------------------------ ResNet --
def resolve_checkpoint(name):
    """A local file for a checkpoint name.  Plain paths are used as they are; model-zoo names (`open-mmlab://...`,
    `torchvision://...`, http URLs: there is no network here) are looked up as `<dir>/<name with ':' and '/' -> '_'>.pth`
    and `<dir>/<basename>.pth` under $PT_PRETRAINED_DIR (default ~/.cache/point_teacher_amd/pretrained)."""
    if os.path.isfile(name):
        return name
    root = os.environ.get('PT_PRETRAINED_DIR', os.path.join(os.path.expanduser('~'), '.cache', 'point_teacher_amd', 'pretrained'))
    flat = name.replace('://', '_').replace('/', '_').replace(':', '_')
    for cand in (flat, flat + '.pth', os.path.basename(name), os.path.basename(name) + '.pth'):
        path = os.path.join(root, cand)
        if os.path.isfile(path):
            return path
    return None


def load_pretrained(module, checkpoint, prefix=None):
    """mmcv's `Pretrained` initialiser for this path: load `checkpoint` into `module` (non-strict, `state_dict` /
    `model` wrappers and a `prefix` stripped).  When the file cannot be found the module KEEPS its random initialisation
    and a loud warning says so - training from it does not reproduce the reference's recipe."""
    import warnings
    path = resolve_checkpoint(checkpoint)
    if path is None:
        warnings.warn(f'pretrained checkpoint {checkpoint!r} not found (set PT_PRETRAINED_DIR or pass a file path): '
                      f'{type(module).__name__} stays RANDOMLY initialised', RuntimeWarning, stacklevel=2)
        return False
    sd = torch.load(path, map_location='cpu', weights_only=False)
    for k in ('state_dict', 'model'):
        if isinstance(sd, dict) and k in sd and isinstance(sd[k], dict):
            sd = sd[k]
    pre = (prefix.rstrip('.') + '.') if prefix else None
    if pre is None and any(k.startswith('backbone.') for k in sd) and not any(k.startswith('backbone.') for k in module.state_dict()):
        pre = 'backbone.'
    if pre:
        sd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    res = module.load_state_dict(sd, strict=False)
    own = set(module.state_dict())
    if not own & set(sd):
        warnings.warn(f'{path}: no parameter name matches {type(module).__name__}', RuntimeWarning, stacklevel=2)
        return False
    missing = [k for k in res.missing_keys if 'num_batches_tracked' not in k]
    if missing:
        warnings.warn(f'{path}: {len(missing)} parameters of {type(module).__name__} not in the checkpoint, e.g. {missing[:3]}',
                      RuntimeWarning, stacklevel=2)
    return True


def _bn_key(bn):
    """Everything the (scale, shift) of an eval-mode BatchNorm depends on.  Modules flagged
    `_affine_dynamic` (the teacher's: the flat EMA kernel rewrites their weight/bias through raw
    pointers, which autograd's version counters do not see) also key on functional.PARAM_EPOCH."""
    return (bn.weight._version, bn.bias._version, bn.running_var._version, bn.running_mean._version,
            bn.weight.data_ptr(), bn.weight.device, F.PARAM_EPOCH[0] if getattr(bn, '_affine_dynamic', False) else 0)


def _bn_affine(bn):
    """(scale, shift) of an eval-mode BatchNorm with a non-trainable affine, cached until its tensors change."""
    cache = getattr(bn, '_affine', None)
    ver = _bn_key(bn)
    if cache is None or cache[0] != ver:
        with torch.no_grad():
            sc = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).float().contiguous()
            cache = (ver, sc, (bn.bias - bn.running_mean * sc).float().contiguous())
        bn._affine = cache
    return cache[1], cache[2]


def refresh_bn_affines(model, trainable=False):
    """Recompute the cached (scale, shift) of EVERY eval-mode BatchNorm of `model` whose affine is frozen (`trainable=False`) or
    trains (`trainable=True`: OBB config 5 - eval statistics, trainable gamma / beta; also caches rstd for d gamma) with a handful
    of launches (4 concatenations + 4 element-wise ops into persistent buffers) instead of four tiny kernels per layer.  Called
    after the teacher EMA and, for trainable affines, once per parameter update; the per-layer triples are views of persistent
    buffers (the plane kernels' weight-plane cache holds pointers to the scales)."""
    key = (bool(trainable), model.training)
    cached = model.__dict__.setdefault('_bn_lists', {}).get(key)
    allb = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)] if cached is None else cached[0]
    state = tuple((m.training, m.weight.requires_grad) for m in allb)
    if cached is None or cached[1] != state:         # (the module walk is 0.3 ms of host time: once per train() / eval() state)
        cached = model._bn_lists[key] = (allb, state, [m for m in allb if not m.training and bool(m.weight.requires_grad) == trainable])
    bns = cached[2]
    if not bns:
        return
    eps = bns[0].eps
    assert all(b.eps == eps for b in bns)
    n = sum(b.num_features for b in bns)
    name = '_bn_affine_buf_t' if trainable else '_bn_affine_buf'
    buf = getattr(model, name, None)
    if buf is None or buf[0].numel() != n or buf[0].device != bns[0].weight.device:
        buf = tuple(torch.empty(n, dtype=torch.float32, device=bns[0].weight.device) for _ in range(3))
        setattr(model, name, buf)
    with torch.no_grad():
        w = torch.cat([b.weight.detach().float() for b in bns])
        var = torch.cat([b.running_var.float() for b in bns])
        torch.rsqrt(var + eps, out=buf[2])
        torch.mul(w, buf[2], out=buf[0])
        mean = torch.cat([b.running_mean.float() for b in bns])
        bias = torch.cat([b.bias.detach().float() for b in bns])
        torch.sub(bias, mean * buf[0], out=buf[1])
    o = 0
    for b in bns:
        c = b.num_features
        if trainable:
            b._affine_dynamic = True                 # the fused SGD kernel rewrites gamma / beta through raw pointers
        b._affine = (_bn_key(b), buf[0][o:o + c], buf[1][o:o + c])
        b._rstd = buf[2][o:o + c]
        o += c


def conv_bn(x, conv, bn, relu, residual=None):
    """conv -> BatchNorm (-> + residual) (-> ReLU).  Every BatchNorm on this path is in eval mode
    with a frozen affine (norm_eval=True, requires_grad=False, resnet.py:647-658), i.e. a
    per-channel affine map; BN, the residual add and the ReLU then run as ONE pass over the
    activation in each direction (pt_affine_relu_fwd/bwd) instead of three."""
    if (_SPLIT_CONV and residual is None and type(conv) is nn.Conv2d and conv.bias is None and not bn.training
            and not bn.weight.requires_grad and getattr(bn, 'fuse_epilogue', True) and F.split_conv3x3_ok(x, conv)):
        # the 3x3 convolution of a bottleneck with its frozen BatchNorm and ReLU as the epilogue of the implicit GEMM on the bf16
        # matrix cores (split-bf16 operands, fp32 accumulation): no separate pass over the activation at all
        sc, sh = _bn_affine(bn)
        return F.split_conv3x3(x, conv.weight, sh, relu, scale=sc)
    if _SPLIT_CONV and type(conv) is nn.Conv2d and conv.bias is None and F.split_conv3x3_ok(x, conv):
        # a BatchNorm that is not a frozen affine map (config 5: eval-mode statistics, trainable affine): the convolution alone on
        # the bf16 matrix cores, the norm by the passes below
        y = F.split_conv3x3(x, conv.weight, None, False)
    else:
        y = conv(x)
    fused = (not bn.training and not bn.weight.requires_grad and y.dtype == torch.float32 and y.is_cuda
             and getattr(bn, 'fuse_epilogue', True) and y.numel() % 4 == 0
             and (y.shape[1] % 4 == 0 if (y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous())
                  else (y.is_contiguous() and (y.shape[2] * y.shape[3]) % 4 == 0)))
    if fused:
        sc, sh = _bn_affine(bn)
        return F.affine_relu_(y, sc, sh, residual, relu)
    C = y.shape[1]
    if (not bn.training and not bn.weight.requires_grad and y.dtype == torch.bfloat16 and y.is_cuda and y.dim() == 4
            and getattr(bn, 'fuse_epilogue', True) and C % 8 == 0
            and y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous()):
        sc, sh = _bn_affine(bn)                       # bf16 autocast backbone (configs[2]): same epilogue, bf16 in / out
        return F.affine_relu_bf16_(y, sc, sh, residual, relu)
    if (not bn.training and bn.weight.requires_grad and y.dtype == torch.float32 and y.is_cuda and y.dim() == 4
            and getattr(bn, 'fuse_epilogue', True) and C % 4 == 0 and (256 % (C // 4) == 0 or (C // 4) % 256 == 0)
            and y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous()):
        # eval-mode BatchNorm whose affine trains (config 5): fused forward, one-pass backward with the
        # per-channel reductions of the weight / bias gradients (pt_affine_relu_bwd_train)
        return F.bn_eval_relu(y, bn, residual, relu)
    y = bn(y)
    if residual is not None:
        y = y + residual
    return TF.relu(y, inplace=True) if relu else y


class Bottleneck(nn.Module):
    """backbones/resnet.py:96-303; `caffe` style puts the stride on conv1 (:153-158)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch'):
        super().__init__()
        assert style in ('pytorch', 'caffe')
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=s1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=s2, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def plane_ok(self):
        """Eval-mode BatchNorms (frozen affine, or all of them trainable - OBB config 5), one stride in the block (on conv1 -
        `caffe` - or on conv2 - `pytorch`) matched by the downsample, no dilation, channel counts the weight-gradient tiles take,
        channels_last fp32 weights on the device: the block can run plane-native (planes._BottleneckP)."""
        if not (_PLANE_TRUNK and PL.autocast_ok()):
            return False
        m = self._modules
        key = (self.training, m['bn1'].training, m['bn1'].weight.requires_grad, m['conv1'].weight.data_ptr(), _PLANE_BN_TRAIN,
               m['bn1'].__dict__.get('fuse_epilogue', True))
        c = self.__dict__.get('_plane_ok_cache')
        if c is None or c[0] != key:                    # (the module walk below is ~40 us of host time per block and pass)
            c = self.__dict__['_plane_ok_cache'] = (key, self._plane_ok_static())
        return c[1]

    def _plane_ok_static(self):
        bns = [self.bn1, self.bn2, self.bn3] + ([self.downsample[1]] if self.downsample is not None else [])
        convs = [self.conv1, self.conv2, self.conv3] + ([self.downsample[0]] if self.downsample is not None else [])
        s1, s2 = self.conv1.stride[0], self.conv2.stride[0]
        train = [bool(b.weight.requires_grad) for b in bns]
        return (_PLANE_TRUNK and all((not b.training) and getattr(b, 'fuse_epilogue', True) and b.weight.requires_grad == b.bias.requires_grad
                                     for b in bns)
                and (not any(train) or (all(train) and _PLANE_BN_TRAIN))
                and self.conv2.dilation == (1, 1) and s1 in (1, 2) and s2 in (1, 2) and s1 * s2 in (1, 2)
                and (self.downsample is None or self.downsample[0].stride[0] == s1 * s2)
                and (self.downsample is not None or s1 * s2 == 1)
                # the weight-gradient kernel tiles 128 channels; a FROZEN block (layer1: 64 channels) has no weight gradient and only
                # needs whole 32-channel k-steps
                and all(((c.in_channels % 128 == 0 and c.out_channels % 128 == 0) or
                         (not c.weight.requires_grad and not any(train) and c.in_channels % 32 == 0 and c.out_channels % 32 == 0))
                        and c.weight.is_cuda and c.weight.dtype == torch.float32 and F._ConvWeightPlanes.ok(c.weight) for c in convs))

    def forward_planes(self, x):
        """x: planes.PlaneAct (a ReLU output) or the fp32 channels_last output of the frozen stem -> PlaneAct."""
        is_p = isinstance(x, PL.PlaneAct)
        B, Cin, H, W = x.shape
        s1, s2 = self.conv1.stride[0], self.conv2.stride[0]
        ds = self.downsample
        train = bool(self.bn1.weight.requires_grad)
        bns = {'1': self.bn1, '2': self.bn2, '3': self.bn3}
        if ds is not None:
            bns['d'] = ds[1]
        terms = {}
        for k, bn in bns.items():
            sc, sh = _bn_affine(bn)
            terms[k] = (sc, sh, getattr(bn, '_rstd', None) if train else None, bn.running_mean if train else None)
            assert not train or terms[k][2] is not None, 'trainable BatchNorm terms are refreshed by ResNet.forward'
        if ds is None:
            terms['d'] = (None, None, None, None)
        p = self.conv1.out_channels
        cfg = PL.BottleneckCfg(B=B, H=H, W=W, Cin=Cin, planes=p, s1=s1, s2=s2, x_planes=is_p, x_relu=bool(is_p and x.relu),
                               bn=terms, has_ds=ds is not None, bn_train=train, name=getattr(self, 'plane_name', 'block'))
        gb = [(bn.weight, bn.bias) if train else (None, None) for bn in (self.bn1, self.bn2, self.bn3)]
        gb.append((ds[1].weight, ds[1].bias) if (train and ds is not None) else (None, None))
        t = PL._BottleneckP.apply(x.t if is_p else x, self.conv1.weight, self.conv2.weight, self.conv3.weight,
                                  ds[0].weight if ds is not None else None, *[t_ for pair in gb for t_ in pair], cfg)
        Ho, Wo = PL.out_hw(H, W, 1, s1 * s2, 0)
        return PL.PlaneAct(t, B, Ho, Wo, 4 * p, True)

    def forward(self, x):
        if isinstance(x, PL.PlaneAct):
            if self.plane_ok():
                return self.forward_planes(x)
            x = x.tensor()
        identity = x
        out = conv_bn(x, self.conv1, self.bn1, True)
        out = conv_bn(out, self.conv2, self.bn2, True)
        if self.downsample is not None:
            identity = conv_bn(x, self.downsample[0], self.downsample[1], False)
        return conv_bn(out, self.conv3, self.bn3, True, residual=identity)


@BACKBONES.register_module()
class ResNet(nn.Module):
    """backbones/resnet.py:305-658 for depth 50/101/152 (Bottleneck)."""
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth, in_channels=3, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(0, 1, 2, 3), style='pytorch', frozen_stages=-1, norm_cfg=dict(type='BN', requires_grad=True),
                 norm_eval=True, zero_init_residual=True, pretrained=None, init_cfg=None, **unused):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        blocks = self.arch_settings[depth][:num_stages]
        self.out_indices, self.frozen_stages, self.norm_eval = out_indices, frozen_stages, norm_eval
        self.conv1 = nn.Conv2d(in_channels, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = 64
        self.res_layers = []
        for i, nb in enumerate(blocks):
            planes = 64 * 2 ** i
            layers = []
            for j in range(nb):
                stride = strides[i] if j == 0 else 1
                ds = None
                if j == 0 and (stride != 1 or inplanes != planes * 4):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
                layers.append(Bottleneck(inplanes, planes, stride, dilations[i], ds, style))
                layers[-1].plane_name = f'layer{i + 1}.{j}'            # (the census' name of the block's tensors)
                inplanes = planes * 4
            name = f'layer{i + 1}'
            self.add_module(name, nn.Sequential(*layers))
            self.res_layers.append(name)
        if not norm_cfg.get('requires_grad', True):
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    for p in m.parameters():
                        p.requires_grad = False
        self._freeze_stages()
        self.init_cfg = init_cfg if init_cfg is not None else (dict(type='Pretrained', checkpoint=pretrained) if pretrained else None)
        self.pretrained_loaded = None           # None: no checkpoint asked for; True / False: found and loaded / not found
        self.init_weights()

    def init_weights(self):
        """Kaiming / constant init (resnet.py:612-628), then the `Pretrained` init_cfg of the configs
        (`open-mmlab://detectron/resnet50_caffe`): see load_pretrained."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        cfg = self.init_cfg
        if isinstance(cfg, dict) and cfg.get('type') == 'Pretrained' and cfg.get('checkpoint'):
            self.pretrained_loaded = load_pretrained(self, cfg['checkpoint'], prefix=cfg.get('prefix'))

    def _freeze_stages(self):
        """resnet.py:612-628"""
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def forward_stem(self, x, planes=False):
        """conv1 / bn1 / maxpool and the frozen stages: the part of the network no gradient reaches and no optimizer step
        changes (frozen_stages >= 0).  -> (activation, outputs collected so far); `forward(None, stem=...)` continues from it.
        A teacher that holds the same frozen weights as its student (both load the same pretrained file and the stem never
        trains) computes the same stem on the same image: TS_P2B_FCOS evaluates it once."""
        x = self._stem_head(x)
        outs = []
        # the frozen stages plane-native too (round 5): H2 planes (4 bytes per element) between their convolutions, BatchNorm + ReLU +
        # identity in the epilogues - no library convolution + separate BatchNorm pass per layer (layer1 at 200 x 200 is bound by
        # its bytes: ~3.1 GB per pass of six images instead of ~5.5); fp32 mode only (under bf16 autocast the library path stays)
        plane = bool(planes and _PLANE_STEM and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
                     and PL.use_f16('trunk') and not x.requires_grad)
        for i in range(max(self.frozen_stages, 0)):
            layer = getattr(self, self.res_layers[i])
            if plane:
                for blk in layer:
                    if isinstance(x, PL.PlaneAct) or (PL.dense_ok(x) and blk.plane_ok()):
                        x = blk(x) if isinstance(x, PL.PlaneAct) else blk.forward_planes(x)
                    else:
                        x = blk(x)
            else:
                x = layer(x)
            if i in self.out_indices:
                outs.append(x)
        return x, outs

    def _stem_head(self, x):
        """maxpool(relu(bn1(conv1(x)))) (resnet.py:633-640).  Frozen (frozen_stages >= 0, eval-mode BatchNorm, no gradient into the
        image): the convolution, then norm + ReLU + max-pool in ONE pass over its output (pt_affine_relu_maxpool_fwd) instead of an
        in-place BatchNorm pass and a pooling pass."""
        c, bn, mp = self.conv1, self.bn1, self.maxpool
        if (_STEM_TAIL and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled() and not bn.training
                and not bn.weight.requires_grad and not c.weight.requires_grad and not x.requires_grad
                and getattr(bn, 'fuse_epilogue', True) and mp.kernel_size == 3 and mp.stride == 2 and mp.padding == 1
                and mp.dilation == 1 and not mp.ceil_mode):
            y = c(x)
            if y.shape[1] % 4 == 0 and y.is_contiguous(memory_format=torch.channels_last) and y.numel() > 0:
                sc, sh = _bn_affine(bn)
                return F.affine_relu_maxpool(y, sc, sh)
            return mp(TF.relu(bn(y)))
        return mp(conv_bn(x, c, bn, True))

    plane_capable = True

    def refresh_plane_terms(self):
        """gamma * rstd, beta - mean * gamma * rstd and rstd of every TRAINABLE eval-mode BatchNorm (config 5), once per parameter
        epoch.  The input-gradient weight planes fold gamma * rstd in (functional._ConvWeightPlanes reads these buffers when it
        re-splits all weights at the first convolution of an epoch - which may be the TEACHER's), so a detector calls this for its
        student right after the optimizer step / EMA, before any convolution of the iteration; `forward` calls it as well."""
        if _PLANE_BN_TRAIN and getattr(self, '_bn_terms_epoch', None) != (F.PARAM_EPOCH[0], self.training):
            refresh_bn_affines(self, trainable=True)
            self._bn_terms_epoch = (F.PARAM_EPOCH[0], self.training)

    def forward(self, x, stem=None, planes=False):
        """`planes=True` (a caller whose neck reads planes.PlaneAct, i.e. FPN): the trainable stages run plane-native - their
        outputs are PlaneActs - as soon as a block qualifies (Bottleneck.plane_ok); otherwise fp32 tensors as ever."""
        x, outs = self.forward_stem(x, planes=planes) if stem is None else stem
        outs = list(outs)
        if planes:
            self.refresh_plane_terms()
        for i in range(max(self.frozen_stages, 0), len(self.res_layers)):
            layer = getattr(self, self.res_layers[i])
            if planes:
                for blk in layer:
                    if not isinstance(x, PL.PlaneAct) and PL.dense_ok(x) and blk.plane_ok():
                        # enter plane mode: the frozen stem's output is read where the block's stride samples it; an input that takes
                        # a gradient goes through a differentiable split first
                        x = blk.forward_planes(PL.to_planes(x, f16=PL.use_f16('trunk'), group='trunk', carrier=False) if x.requires_grad else x)
                    else:
                        x = blk(x)                            # (a PlaneAct stays plane-native while the blocks qualify)
            else:
                x = layer(x)
            if i in self.out_indices:
                if isinstance(x, PL.PlaneAct) and i + 1 < len(self.res_layers):
                    x, o = PL.fan_out(x)                      # two consumers (the next stage, the neck): exact gradient addition
                    outs.append(o)
                else:
                    outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        """resnet.py:647-658: re-freeze and force every BN to eval when norm_eval."""
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self


# --------------------------------------------------------------------------- FPN --
def _xavier_uniform_(module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


@NECKS.register_module()
class FPN(nn.Module):
    """necks/fpn.py:10-202 (no norm / no activation, as in the Point-Teacher configs)."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode='nearest'), init_cfg=None):
        super().__init__()
        assert isinstance(in_channels, list) and norm_cfg is None and conv_cfg is None
        self.in_channels, self.out_channels, self.num_outs = in_channels, out_channels, num_outs
        self.num_ins = len(in_channels)
        self.relu_before_extra_convs = relu_before_extra_convs
        self.upsample_cfg = dict(upsample_cfg)
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= len(in_channels) and num_outs == end_level - start_level
        self.start_level, self.end_level = start_level, end_level
        assert isinstance(add_extra_convs, (str, bool))
        if isinstance(add_extra_convs, str):
            assert add_extra_convs in ('on_input', 'on_lateral', 'on_output')
        elif add_extra_convs:
            add_extra_convs = 'on_input'
        self.add_extra_convs = add_extra_convs
        act = act_cfg is not None
        self.lateral_convs, self.fpn_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, act=act))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, act=act))
        extra_levels = num_outs - self.backbone_end_level + self.start_level
        if self.add_extra_convs and extra_levels >= 1:
            for i in range(extra_levels):
                cin = self.in_channels[self.backbone_end_level - 1] if (i == 0 and self.add_extra_convs == 'on_input') \
                    else out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1, act=act))
        _xavier_uniform_(self)

    accepts_planes = True

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        laterals = [l(inputs[i + self.start_level]) for i, l in enumerate(self.lateral_convs)]        # (PlaneAct inputs: ConvModule)
        n = len(laterals)
        for i in range(n - 1, 0, -1):
            if 'scale_factor' in self.upsample_cfg:
                laterals[i - 1] = laterals[i - 1] + TF.interpolate(laterals[i], **self.upsample_cfg)
            elif self.upsample_cfg.get('mode', 'nearest') == 'nearest' and len(self.upsample_cfg) <= 1:
                laterals[i - 1] = F.upsample_add(laterals[i - 1], laterals[i])        # one fused pass on NHWC maps
            else:
                laterals[i - 1] = laterals[i - 1] + TF.interpolate(laterals[i], size=laterals[i - 1].shape[2:],
                                                                   **self.upsample_cfg)
        outs = [self.fpn_convs[i](laterals[i]) for i in range(n)]
        if self.num_outs > len(outs):
            if not self.add_extra_convs:
                for i in range(self.num_outs - n):
                    outs.append(TF.max_pool2d(outs[-1], 1, stride=2))
            else:
                if self.add_extra_convs == 'on_input':
                    src = inputs[self.backbone_end_level - 1]
                elif self.add_extra_convs == 'on_lateral':
                    src = laterals[-1]
                else:
                    src = outs[-1]
                outs.append(self.fpn_convs[n](src))
                for i in range(n + 1, self.num_outs):
                    outs.append(self.fpn_convs[i](TF.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]))
        return tuple(outs)


@NECKS.register_module()
class PSAGG(nn.Module):
    """necks/ps_fpn.py:8-75: collapse the pyramid top-down into ONE stride-8 map."""

    def __init__(self, num_aggregation, in_channels, out_channels, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode='nearest'), init_cfg=None):
        super().__init__()
        assert norm_cfg is None and conv_cfg is None
        self.num_aggregation, self.in_channels, self.out_channels = num_aggregation, in_channels, out_channels
        self.upsample_cfg = dict(upsample_cfg)
        self.lateral_convs = nn.ModuleList()
        for i in range(num_aggregation):
            cout = in_channels if i != num_aggregation - 1 else out_channels
            self.lateral_convs.append(ConvModule(in_channels, cout, 1, act=act_cfg is not None))
        _xavier_uniform_(self)

    def forward(self, inputs):
        inputs = list(inputs)
        inputs[-1] = self.lateral_convs[0](inputs[-1])
        for i in range(self.num_aggregation):
            index = self.num_aggregation - i - 1
            if index != 0:
                if self.upsample_cfg.get('mode', 'nearest') == 'nearest' and len(self.upsample_cfg) <= 1:
                    summed = F.upsample_add(inputs[index - 1], inputs[index])            # one fused pass on NHWC maps
                elif 'scale_factor' in self.upsample_cfg:
                    summed = inputs[index - 1] + TF.interpolate(inputs[index], **self.upsample_cfg)
                else:
                    summed = inputs[index - 1] + TF.interpolate(inputs[index], size=inputs[index - 1].shape[2:], **self.upsample_cfg)
                inputs[index - 1] = self.lateral_convs[i + 1](summed)
        return tuple([inputs[0]])


# ---------------------------------------------------------------- RoI extraction --
class RoIAlign(nn.Module):
    """mmcv.ops.RoIAlign (constructor defaults of mmcv 1.x: sampling_ratio=0, pool_mode='avg',
    aligned=True), built at roi_extractors/base_roi_extractor.py:53-58."""

    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True,
                 use_torchvision=False):
        super().__init__()
        assert pool_mode == 'avg', "only pool_mode='avg' is on the Point-Teacher path"
        self.output_size = output_size if isinstance(output_size, int) else output_size[0]
        self.spatial_scale, self.sampling_ratio, self.aligned = float(spatial_scale), int(sampling_ratio), aligned

    def forward(self, input, rois, group=1, planes=False, tail=0):
        """planes=True: the result as planes.PlaneAct [1, K, 1, C * 49] - the split planes of `.flatten(1)` - when the kernel takes the
        shape (functional.roi_align_planes_ok); the caller feeds it to planes.linear."""
        if planes and _PLANE_TRUNK and rois.shape[0] > 0 and F.roi_align_planes_ok(input, rois, self.output_size):
            t = F.roi_align_planes(input, rois, self.spatial_scale, self.sampling_ratio, self.aligned, group, f16=PL.use_f16('fc'), tail=tail)
            return PL.PlaneAct(t, 1, rois.shape[0], 1, input.shape[1] * self.output_size ** 2, False, gcarrier=True)
        return F.roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned, group)


_ROI_LAYERS = {'RoIAlign': RoIAlign}


@ROI_EXTRACTORS.register_module()
class SingleRoIExtractor(nn.Module):
    """roi_extractors/single_level_roi_extractor.py:9-110 + base_roi_extractor.py (single level used)."""

    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56, init_cfg=None):
        super().__init__()
        cfg = dict(roi_layer)
        layer_cls = _ROI_LAYERS[cfg.pop('type')]
        self.roi_layers = nn.ModuleList([layer_cls(spatial_scale=1 / s, **cfg) for s in featmap_strides])
        self.out_channels, self.featmap_strides, self.finest_scale = out_channels, featmap_strides, finest_scale

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def map_roi_levels(self, rois, num_levels):
        scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
        lvls = torch.floor(torch.log2(scale / self.finest_scale + 1e-6))
        return lvls.clamp(min=0, max=num_levels - 1).long()

    def forward(self, feats, rois, roi_scale_factor=None, group=1, planes=False, tail=0):
        """`group`: how many consecutive RoIs overlap (one MIL bag) - a locality hint for the backward.
        `planes`: a single-level extractor may return planes.PlaneAct (see RoIAlign.forward)."""
        assert roi_scale_factor is None
        if len(feats) == 1:
            if rois.shape[0] == 0:
                o = self.roi_layers[0].output_size
                return feats[0].new_zeros(0, self.out_channels, o, o)
            if planes:
                return self.roi_layers[0](feats[0], rois, group, planes=True, tail=tail)
            return self.roi_layers[0](feats[0], rois, group)
        out_size = self.roi_layers[0].output_size
        roi_feats = feats[0].new_zeros(rois.size(0), self.out_channels, out_size, out_size)
        lvls = self.map_roi_levels(rois, len(feats))
        for i in range(len(feats)):
            inds = (lvls == i).nonzero(as_tuple=False).squeeze(1)
            if inds.numel() > 0:
                roi_feats[inds] = self.roi_layers[i](feats[i], rois[inds])
        return roi_feats
