"""CPU ORACLE of the whole Point-Teacher training iteration - TEST INFRASTRUCTURE ONLY.

A functional torch-CPU fp32 restatement of `TS_P2B_FCOS.train_step` (reference:
/root/reference/HBB_TOD/mmdet/models/detectors/fcos_p2b_teacher_student.py:116-252 and the
modules it drives) built ONLY from `torch.nn.functional` convolutions and the op-level oracle
`oracle/ref_ops.py`.  It shares no code with the product package: weights come in as a plain
`state_dict` (the parameter names are those of an mmdet checkpoint), randomness comes in as
explicit draws.  Used by tests (whole-loss-dict parity against the HIP path) and by
`bench.py` as the `cpu_baseline` ("port") leg.  The product never imports this file.

Pinning: the op-level pieces are pinned to the reference through tests/golden (see
ref_ops.py).  The network modules (ResNet/FPN/PSAGG/head convs) are compositions of torch
convolutions; their oracle is torch-CPU itself.  RoIAlign / rotated NMS / the
rasteriser remain "parity unpinned" (mmcv / cv2 absent).
"""
import math
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import ref_ops as R

NUM_CLASSES = 8
STRIDE = 8


# ------------------------------------------------------------------- state dict --
def _conv(sd, name, cout, cin, k, bias, std=None, gen=None):
    fan_out = cout * k * k
    w = torch.randn(cout, cin, k, k, generator=gen) * (std if std is not None else math.sqrt(2.0 / fan_out))
    sd[name + '.weight'] = w
    if bias:
        sd[name + '.bias'] = torch.zeros(cout)


def _bn(sd, name, c):
    sd[name + '.weight'] = torch.ones(c)
    sd[name + '.bias'] = torch.zeros(c)
    sd[name + '.running_mean'] = torch.zeros(c)
    sd[name + '.running_var'] = torch.ones(c)


def init_detector_state(seed=0, num_stages=1):
    """Random weights with the shapes/names of `Student_FCOS` (R50-caffe + FPN + PSAGG + head)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    _conv(sd, 'backbone.conv1', 64, 3, 7, False, gen=g)
    sd['backbone.conv1.weight'] *= 1.0 / 64.0            # same benchmark init as the product (synthetic.py)
    _bn(sd, 'backbone.bn1', 64)
    inpl = 64
    for li, nb in enumerate((3, 4, 6, 3)):
        planes = 64 * 2 ** li
        for j in range(nb):
            p = f'backbone.layer{li + 1}.{j}'
            _conv(sd, p + '.conv1', planes, inpl, 1, False, gen=g); _bn(sd, p + '.bn1', planes)
            _conv(sd, p + '.conv2', planes, planes, 3, False, gen=g); _bn(sd, p + '.bn2', planes)
            _conv(sd, p + '.conv3', planes * 4, planes, 1, False, gen=g); _bn(sd, p + '.bn3', planes * 4)
            if j == 0:
                _conv(sd, p + '.downsample.0', planes * 4, inpl, 1, False, gen=g); _bn(sd, p + '.downsample.1', planes * 4)
            inpl = planes * 4
    for i, cin in enumerate((512, 1024, 2048)):
        _conv(sd, f'neck.lateral_convs.{i}.conv', 256, cin, 1, True, std=math.sqrt(2.0 / (cin + 256)), gen=g)
        _conv(sd, f'neck.fpn_convs.{i}.conv', 256, 256, 3, True, std=math.sqrt(2.0 / (2304 * 2)), gen=g)
    for i in (3, 4):
        _conv(sd, f'neck.fpn_convs.{i}.conv', 256, 256, 3, True, std=math.sqrt(2.0 / (2304 * 2)), gen=g)
    for i in range(5):
        _conv(sd, f'neck_agg.lateral_convs.{i}.conv', 256, 256, 1, True, std=math.sqrt(2.0 / 512), gen=g)
    for br in ('cls_convs', 'reg_convs'):
        for i in range(4):
            _conv(sd, f'bbox_head.{br}.{i}.conv', 256, 256, 3, True, std=0.01, gen=g)
    _conv(sd, 'bbox_head.conv_cls', NUM_CLASSES, 256, 3, True, std=0.01, gen=g)
    sd['bbox_head.conv_cls.bias'] = torch.full((NUM_CLASSES,), -math.log((1 - 0.01) / 0.01))
    _conv(sd, 'bbox_head.conv_reg', 4, 256, 3, True, std=0.01, gen=g)
    _conv(sd, 'bbox_head.conv_centerness', 1, 256, 3, True, std=0.01, gen=g)
    sd['bbox_head.scales.0.scale'] = torch.tensor(1.0)

    def lin(name, cout, cin, scale=1.0):
        b = 1.0 / math.sqrt(cin)
        sd[name + '.weight'] = (torch.rand(cout, cin, generator=g) * 2 - 1) * b * scale
        sd[name + '.bias'] = (torch.rand(cout, generator=g) * 2 - 1) * b * (0.0 if scale != 1.0 else 1.0)
    for fam in ('shared_fcs', 'shared_fcs_refine'):
        lin(f'bbox_head.{fam}.0', 1024, 12544); lin(f'bbox_head.{fam}.1', 1024, 1024)
    for s in range(num_stages):
        for fam in ('shared_fcs_bag', 'shared_fcs_reg'):
            lin(f'bbox_head.{fam}.{s}.0', 1024, 12544); lin(f'bbox_head.{fam}.{s}.1', 1024, 1024)
        lin(f'bbox_head.fc_cls.{s}', NUM_CLASSES, 1024); lin(f'bbox_head.fc_ins.{s}', NUM_CLASSES, 1024)
        lin(f'bbox_head.fc_reg.{s}', 4, 1024, scale=0.01); lin(f'bbox_head.fc_iou.{s}', 1, 1024)
    return sd


def damp_mil_logits(sd, scale):
    """Scale the bag classifier's last layers (fc_cls / fc_ins) in place.  At random initialisation their logits have a
    standard deviation of ~10 (unnormalised 0-255 pixels under a random ResNet), i.e. every sigmoid saturates and the
    bag loss sits on its `log(1 - p + 1e-6)` cliff, where one fp32 ulp of `p` moves the value by per cents; a trained
    head has moderate logits.  Whole-iteration fixtures use a damped head so that they compare arithmetic, not the
    amplification of rounding noise."""
    for k in sd:
        if '.fc_cls.' in k or '.fc_ins.' in k:
            sd[k] = sd[k] * scale
    return sd


def is_param(name):
    return not (name.endswith('running_mean') or name.endswith('running_var') or name.endswith('num_batches_tracked'))


def trainable(name):
    """What the reference trains: everything except BN affine (norm_cfg requires_grad=False) and
    the frozen stem + layer1 (frozen_stages=1), resnet.py:612-628."""
    if not is_param(name):
        return False
    if '.bn' in name or 'downsample.1' in name:
        return False
    if name.startswith('backbone.conv1') or name.startswith('backbone.layer1.'):
        return False
    return True


# ------------------------------------------------------------------------ modules --
def _bn_eval(sd, p, x):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'], False,
                        0.0, 1e-5)


def resnet50_caffe(sd, x, pre='backbone'):
    """backbones/resnet.py (style='caffe': stride on the first 1x1), BN in eval mode."""
    x = F.max_pool2d(F.relu(_bn_eval(sd, pre + '.bn1', F.conv2d(x, sd[pre + '.conv1.weight'], None, 2, 3))), 3, 2, 1)
    outs = []
    for li, nb in enumerate((3, 4, 6, 3)):
        for j in range(nb):
            p = f'{pre}.layer{li + 1}.{j}'
            s = 2 if (j == 0 and li > 0) else 1
            idt = x
            o = F.relu(_bn_eval(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'], None, s)))
            o = F.relu(_bn_eval(sd, p + '.bn2', F.conv2d(o, sd[p + '.conv2.weight'], None, 1, 1)))
            o = _bn_eval(sd, p + '.bn3', F.conv2d(o, sd[p + '.conv3.weight']))
            if j == 0:
                idt = _bn_eval(sd, p + '.downsample.1', F.conv2d(x, sd[p + '.downsample.0.weight'], None, s))
            x = F.relu(o + idt)
        outs.append(x)
    return outs


def _c(sd, p, x, stride=1, pad=0):
    return F.conv2d(x, sd[p + '.conv.weight'], sd[p + '.conv.bias'], stride, pad)


def fpn_psagg(sd, feats):
    """necks/fpn.py:151-202 (start_level 1, extra convs on output, relu before the 2nd extra) then
    necks/ps_fpn.py:56-75."""
    lat = [_c(sd, f'neck.lateral_convs.{i}', feats[i + 1]) for i in range(3)]
    for i in (2, 1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
    outs = [_c(sd, f'neck.fpn_convs.{i}', lat[i], 1, 1) for i in range(3)]
    outs.append(_c(sd, 'neck.fpn_convs.3', outs[-1], 2, 1))
    outs.append(_c(sd, 'neck.fpn_convs.4', F.relu(outs[-1]), 2, 1))
    x = list(outs)
    x[4] = _c(sd, 'neck_agg.lateral_convs.0', x[4])
    for i in range(5):
        idx = 4 - i
        if idx != 0:
            x[idx - 1] = _c(sd, f'neck_agg.lateral_convs.{i + 1}',
                            x[idx - 1] + F.interpolate(x[idx], size=x[idx - 1].shape[2:], mode='nearest'))
    return x[0]


def extract_feat(sd, img):
    if _BF16[0]:
        return extract_feat_bf16(sd, img)
    return fpn_psagg(sd, resnet50_caffe(sd, img))


# ----------------------------------------------------- BASELINE configs[2]: bf16 backbone / necks, fp32 head --
_BF16 = [False]


class bf16_backbone:
    """`with bf16_backbone():` every extract_feat of the oracle models a bf16 backbone + FPN + PSAGG (below)."""

    def __enter__(self):
        self.old, _BF16[0] = _BF16[0], True

    def __exit__(self, *exc):
        _BF16[0] = self.old


class _RoundBF16(torch.autograd.Function):
    """Round to bfloat16 (the storage type between layers); straight gradient, as a dtype cast has."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _q(x):
    return _RoundBF16.apply(x)


def extract_feat_bf16(sd, img):
    """The backbone + FPN + PSAGG with bf16 operands and bf16 storage between layers, fp32 accumulation inside every
    convolution (what a bf16 MFMA convolution does) and fp32 arithmetic inside the fused BatchNorm / residual / ReLU
    epilogue with ONE rounding at its end (point_teacher_amd/nn_modules.conv_bn, pt_affine_relu_*_bf16); the result is
    widened to fp32 for the dense head, the MIL head and the losses (Student_FCOS.extract_feat).  The reference has
    no such mode of its own: mmdet's fp16 hooks (`auto_fp16` on extract_feat, `force_fp32` on the head's loss / target
    methods, fcos_head_p2b_ts.py:355,377) place the precision boundary at the same spot."""
    def conv(p, x, stride=1, pad=0, bias=True):
        b = sd[p + '.bias'] if (bias and p + '.bias' in sd) else None
        return _q(F.conv2d(x, _q(sd[p + '.weight']), b, stride, pad))

    def bn_act(p, y, relu, res=None):
        sc = sd[p + '.weight'] * torch.rsqrt(sd[p + '.running_var'] + 1e-5)
        sh = sd[p + '.bias'] - sd[p + '.running_mean'] * sc
        y = y * sc[None, :, None, None] + sh[None, :, None, None]
        if res is not None:
            y = y + res
        return _q(F.relu(y) if relu else y)
    pre = 'backbone'
    x = bn_act(pre + '.bn1', conv(pre + '.conv1', _q(img), 2, 3, bias=False), True)
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, nb in enumerate((3, 4, 6, 3)):
        for j in range(nb):
            p = f'{pre}.layer{li + 1}.{j}'
            s = 2 if (j == 0 and li > 0) else 1
            idt = x
            o = bn_act(p + '.bn1', conv(p + '.conv1', x, s, bias=False), True)
            o = bn_act(p + '.bn2', conv(p + '.conv2', o, 1, 1, bias=False), True)
            if j == 0:
                idt = bn_act(p + '.downsample.1', conv(p + '.downsample.0', x, s, bias=False), False)
            x = bn_act(p + '.bn3', conv(p + '.conv3', o, bias=False), True, res=idt)
        outs.append(x)
    lat = [conv(f'neck.lateral_convs.{i}.conv', outs[i + 1]) for i in range(3)]
    for i in (2, 1):
        lat[i - 1] = _q(lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest'))
    f = [conv(f'neck.fpn_convs.{i}.conv', lat[i], 1, 1) for i in range(3)]
    f.append(conv('neck.fpn_convs.3.conv', f[-1], 2, 1))
    f.append(conv('neck.fpn_convs.4.conv', F.relu(f[-1]), 2, 1))
    f[4] = conv('neck_agg.lateral_convs.0.conv', f[4])
    for i in range(5):
        idx = 4 - i
        if idx != 0:
            f[idx - 1] = conv(f'neck_agg.lateral_convs.{i + 1}.conv',
                              _q(f[idx - 1] + F.interpolate(f[idx], size=f[idx - 1].shape[2:], mode='nearest')))
    return f[0].float()


def head_forward(sd, feat):
    """fcos_head_p2b_ts.py:326-353 with norm_on_bbox / centerness_on_reg (the configs)."""
    c = r = feat
    for i in range(4):
        c = F.relu(_c(sd, f'bbox_head.cls_convs.{i}', c, 1, 1))
        r = F.relu(_c(sd, f'bbox_head.reg_convs.{i}', r, 1, 1))
    cls = F.conv2d(c, sd['bbox_head.conv_cls.weight'], sd['bbox_head.conv_cls.bias'], 1, 1)
    reg = F.conv2d(r, sd['bbox_head.conv_reg.weight'], sd['bbox_head.conv_reg.bias'], 1, 1)
    ctr = F.conv2d(r, sd['bbox_head.conv_centerness.weight'], sd['bbox_head.conv_centerness.bias'], 1, 1)
    reg = (reg * sd['bbox_head.scales.0.scale']).float().clamp(min=0) * STRIDE
    return cls, reg, ctr


def grid_points(h, w):
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing='ij')
    return torch.stack((xs.reshape(-1) * STRIDE, ys.reshape(-1) * STRIDE), -1) + STRIDE // 2


# ------------------------------------------------------- vectorised RoIAlign (CPU) --
def roi_align_vec(feat, rois, out_size=7, spatial_scale=0.125, chunk=1024):
    """Same arithmetic as ref_ops.roi_align (aligned=True, adaptive grid), vectorised over all
    RoIs that share a sampling grid so that the full-size workload (K=5000) runs in seconds."""
    B, C, H, W = feat.shape
    K = rois.shape[0]
    out = feat.new_zeros((K, C, out_size, out_size))
    if K == 0:
        return out
    flat = feat.permute(0, 2, 3, 1).reshape(B * H * W, C)
    b = rois[:, 0].long()
    sw = rois[:, 1] * spatial_scale - 0.5
    sh = rois[:, 2] * spatial_scale - 0.5
    rw = (rois[:, 3] * spatial_scale - 0.5) - sw
    rh = (rois[:, 4] * spatial_scale - 0.5) - sh
    bw, bh = rw / out_size, rh / out_size
    gw = torch.ceil(rw / out_size).long()
    gh = torch.ceil(rh / out_size).long()
    key = gh * 100000 + gw
    pieces, order = [], []
    for kv in torch.unique(key).tolist():
        sel = (key == kv).nonzero().reshape(-1)
        g_h, g_w = kv // 100000, kv % 100000
        if g_h <= 0 or g_w <= 0:
            pieces.append(feat.new_zeros((sel.numel(), C, out_size, out_size))); order.append(sel)
            continue
        for s0 in range(0, sel.numel(), max(1, chunk // (g_h * g_w))):
            s = sel[s0:s0 + max(1, chunk // (g_h * g_w))]
            ph = torch.arange(out_size, dtype=torch.float32)
            iy = torch.arange(g_h, dtype=torch.float32)
            ix = torch.arange(g_w, dtype=torch.float32)
            ys = (sh[s, None, None] + ph[None, :, None] * bh[s, None, None]
                  + (iy[None, None, :] + 0.5) * bh[s, None, None] / g_h).reshape(len(s), -1)
            xs = (sw[s, None, None] + ph[None, :, None] * bw[s, None, None]
                  + (ix[None, None, :] + 0.5) * bw[s, None, None] / g_w).reshape(len(s), -1)

            def prep(v, L):
                oob = (v < -1.0) | (v > L)
                v = v.clamp(min=0)
                lo = v.floor().long()
                hc = lo >= L - 1
                lo = torch.where(hc, torch.full_like(lo, L - 1), lo)
                hi = torch.where(hc, lo, lo + 1)
                v = torch.where(hc, lo.float(), v)
                l = v - lo.float()
                return lo, hi, l, 1 - l, oob
            ylo, yhi, ly, hy, yo = prep(ys, H)
            xlo, xhi, lx, hx, xo = prep(xs, W)
            base = (b[s] * H * W)[:, None, None]
            m = ((~yo)[:, :, None] & (~xo)[:, None, :]).float()

            def g(yy, xx, wy, wx):
                idx = base + yy[:, :, None] * W + xx[:, None, :]
                return flat[idx.reshape(-1)].reshape(len(s), ys.shape[1], xs.shape[1], C) * (wy[:, :, None] * wx[:, None, :] * m)[..., None]
            val = g(ylo, xlo, hy, hx) + g(ylo, xhi, hy, lx) + g(yhi, xlo, ly, hx) + g(yhi, xhi, ly, lx)
            val = val.reshape(len(s), out_size, g_h, out_size, g_w, C).sum((2, 4)) / max(g_h * g_w, 1)
            pieces.append(val.permute(0, 3, 1, 2)); order.append(s)
    idx = torch.cat(order)
    return out.index_copy(0, idx, torch.cat(pieces))


# ------------------------------------------------------------------- train step --
def _fc(sd, name, x):
    return F.linear(x, sd[name + '.weight'], sd[name + '.bias'])


def _fc_stack(sd, fam, stage, x):
    x = F.relu(_fc(sd, f'bbox_head.{fam}.{stage}.0', x))
    return F.relu(_fc(sd, f'bbox_head.{fam}.{stage}.1', x))


def mil_stage(sd, feat, stage, cfg, ext_cfg, pb_t, gb_t, pl_t, img_hw, neg_u, alpha, topk=1, beta=0.25,
              syn=None):
    """One MIL stage of burn-in step 2 (fcos_p2b_teacher_student.py:439-462 +
    fcos_head_p2b_ts.py:1182-1344).  `syn` = (x_syn feat, boxes list) switches the regression
    branch to the synthetic bags (step 1, :1279-1316)."""
    losses = {}
    B = len(pb_t)

    def bags_of(boxes_list):
        props = [R.fine_proposals(b, cfg['base_ratios'], cfg['shake_ratio'], cfg['min_scale'], img_hw) for b in boxes_list]
        ext = [R.fine_proposals(p, ext_cfg['base_ratios'], ext_cfg['shake_ratio'], ext_cfg['min_scale'], img_hw)
               for p, _ in props]
        return props, ext
    U1 = len(cfg['base_ratios']) ** 2 * (1 + 4 * len(cfg['shake_ratio'] or []))
    U2 = len(ext_cfg['base_ratios']) ** 2 * (1 + 4 * len(ext_cfg['shake_ratio'] or []))

    def reg_branch(x, boxes_list, real_list):
        _, ext = bags_of(boxes_list)
        bags = torch.cat([e[0] for e in ext])
        valid = torch.cat([e[1] for e in ext]).reshape(-1)
        ref = torch.cat([b[:, None].repeat(1, U1 * U2, 1).reshape(-1, 4) for b in boxes_list])
        real = torch.cat([b[:, None].repeat(1, U1 * U2, 1).reshape(-1, 4) for b in real_list])
        rois = torch.cat([torch.cat([torch.full((e[0].shape[0], 1), float(i)), e[0]], 1) for i, e in enumerate(ext)])
        f = roi_align_vec(x, rois).flatten(1)
        pred = R.delta2bbox(bags, _fc(sd, f'bbox_head.fc_reg.{stage}', _fc_stack(sd, 'shared_fcs_reg', stage, f)),
                            (img_hw[0], img_hw[1], 3))
        loss = R.iou_loss_module(pred, ref, weight=valid.float(), avg_factor=float(pred.shape[0]), kind='dn', hyper=0.2)
        return loss, pred.detach(), valid, bags, real, [e[0].shape[0] for e in ext]
    if syn is None:
        loss_bbox, refined, valid, bags, real, sizes = reg_branch(feat, pb_t, gb_t)
    else:
        loss_bbox, _, _, _, _, _ = reg_branch(syn[0], syn[1], syn[1])
        with torch.no_grad():
            pass
        _, refined, valid, bags, real, sizes = reg_branch(feat, pb_t, gb_t)
    losses[f'stage{stage}_loss_mil_bbox'] = loss_bbox * alpha[0]
    losses[f'stage{stage}_coarse_bags_iou'] = R.bbox_overlaps(bags, real, is_aligned=True).mean()
    losses[f'stage{stage}_refine_bags_iou'] = R.bbox_overlaps(refined, real, is_aligned=True).mean()
    # classifier on the refined bags
    ref_list = torch.split(refined, sizes)
    rois = torch.cat([torch.cat([torch.full((r.shape[0], 1), float(i)), r], 1) for i, r in enumerate(ref_list)])
    f = _fc_stack(sd, 'shared_fcs_bag', stage, roi_align_vec(feat, rois).flatten(1))
    N = sum(b.shape[0] for b in pb_t)
    cls = _fc(sd, f'bbox_head.fc_cls.{stage}', f).view(N, U1, U2, -1)
    ins = _fc(sd, f'bbox_head.fc_ins.{stage}', f).view(N, U1, U2, -1)
    # negatives
    props = [R.fine_proposals(b, cfg['base_ratios'], cfg['shake_ratio'], cfg['min_scale'], img_hw)[0] for b in pb_t]
    negs = [R.negative_proposals(neg_u[i], props[i], img_hw) for i in range(B)]
    nrois = torch.cat([torch.cat([torch.full((n[0].shape[0], 1), float(i)), n[0]], 1) for i, n in enumerate(negs)])
    ncls = _fc(sd, f'bbox_head.fc_cls.{stage}', _fc_stack(sd, 'shared_fcs_bag', stage, roi_align_vec(feat, nrois).flatten(1)))
    labels = torch.cat(pl_t)
    losses[f'stage{stage}_loss_mil_bags'] = R.mil_bag_loss(cls, ins, valid[:, None], labels, ncls,
                                                           torch.cat([n[1] for n in negs])) * alpha[1]
    merged = R.mil_bag_select(cls, ins, valid[:, None], labels, refined, torch.cat(pb_t), img_hw, topk, beta)
    return losses, list(torch.split(merged, [b.shape[0] for b in pb_t]))


def forward_train_step2(sd_s, sd_t, img, gt_bboxes, gt_labels, gt_points, cfg, inject, origin_points=None):
    """fcos_p2b_teacher_student.py:213-252.  `inject`: dict(neg0=[B,4,n] uniforms, aug=(flips, scales)).
    `gt_points`: what genrate_points returns (:504-519: the refined points once an image has been seen);
    `origin_points`: the first-visit points `self.gt_bboxes_point[...]`, which update_points blends with EVERY
    time (:270-271) - they differ from `gt_points` from the second visit on when lamda < 1."""
    origin_points = gt_points if origin_points is None else origin_points
    B, _, H, W = img.shape
    img_hw = (H, W)
    pts = grid_points(H // STRIDE, W // STRIDE)
    losses = {}
    with torch.no_grad():
        cls, reg, ctr = head_forward(sd_t, extract_feat(sd_t, img))
        fc, fr, _ = R._flatten_head(cls, reg, ctr)
        pseudo = [R.generate_pseudo_single(pts, fc[i], fr[i], gt_points[i], gt_labels[i], gt_bboxes[i], 0.0) for i in range(B)]
    pb_c, pp_c = [p[0] for p in pseudo], [p[1] for p in pseudo]
    feat = extract_feat(sd_s, img)
    n = cfg['num_training_burninstep2']
    pb_t, gb_t, pl_t = [b[:n].clone() for b in pb_c], [b[:n] for b in gt_bboxes], [l[:n] for l in gt_labels]
    losses['coarse_bboxes_iou'] = R.bbox_overlaps(torch.cat(pb_t), torch.cat(gb_t), is_aligned=True).mean()
    for stage in range(cfg['num_stages']):
        ml, pb_t = mil_stage(sd_s, feat, stage, cfg['fine_proposal_cfg'][stage], cfg['fine_proposal_extensive_cfg'][stage],
                             pb_t, gb_t, pl_t, img_hw, inject[f'neg{stage}'], cfg['alpha'], cfg['top_k'], cfg['beta'])
        losses[f'stage{stage}_refine_bboxes_iou'] = R.bbox_overlaps(torch.cat(pb_t), torch.cat(gb_t), is_aligned=True).mean()
        losses.update(ml)
    pb_r = [b.clone() for b in pb_c]
    pp_r = [p.clone() for p in pp_c]
    for i in range(B):
        k = pb_t[i].shape[0]
        pb_r[i][:k] = pb_t[i]
        pp_r[i][:k] = R.bbox_xyxy_to_cxcywh(pb_t[i])[:, :2]
    # update_points (:266-274) - lamda blend; the caller keeps the dictionaries
    new_points = [(1 - cfg['lamda']) * R.bbox_xyxy_to_cxcywh(pb_r[i])[:, :2] + cfg['lamda'] * origin_points[i] for i in range(B)]
    real = R.bbox_xyxy_to_cxcywh(torch.cat(gt_bboxes))
    losses['refined_points_distance'] = (torch.sqrt((torch.cat(new_points) - real[:, :2]) ** 2)
                                         / torch.sqrt((real[:, 2:] / 2) ** 2)).mean()
    lc, lb, lt = _aug_loss(sd_s, img, new_points, gt_labels, pp_r, gt_labels, pb_r, inject['aug'], pts)
    losses['loss_cls'], losses['loss_bbox'], losses['loss_centerness'] = lc, lb, lt
    return losses, new_points


def _aug_loss(sd_s, img, gt_points, gt_labels, pp, pl, pb, aug, pts):
    flips, scales = aug
    B = img.shape[0]
    outs = [R.strong_augmentation_single(img[i], gt_points[i], gt_labels[i], pp[i], pl[i], pb[i], flips[i], scales[i])
            for i in range(B)]
    img_aug = torch.stack([o[0] for o in outs])
    cls, reg, ctr = head_forward(sd_s, extract_feat(sd_s, img_aug))
    return R.loss_pseudo(cls, reg, ctr, pts, [o[1] for o in outs], [o[2] for o in outs], [o[3] for o in outs],
                         [o[4] for o in outs], [o[5] for o in outs])


def forward_train_step1(sd_s, sd_t, img, gt_bboxes, gt_labels, gt_points, cfg, inject, origin_points=None):
    """fcos_p2b_teacher_student.py:141-211.  The synthetic images/boxes are INPUTS here
    (inject['img_syn'], inject['syn_boxes']) or are generated from inject['syn'] (per-image draw dictionaries,
    ref_ops.generate_black_paper).  `origin_points`: see forward_train_step2."""
    origin_points = gt_points if origin_points is None else origin_points
    B, _, H, W = img.shape
    img_hw = (H, W)
    pts = grid_points(H // STRIDE, W // STRIDE)
    losses = {}
    if 'syn' in inject:           # genrate_syn :469-502 with the draws of this iteration
        prior = torch.tensor(cfg['shape_list'], dtype=torch.float32)
        gen = [R.generate_black_paper(img[i], gt_bboxes[i], prior, inject['syn'][i]) for i in range(B)]
        img_syn, syn_boxes = torch.stack([g[0] for g in gen]), [g[1] for g in gen]
    else:
        img_syn, syn_boxes = inject['img_syn'], inject['syn_boxes']
    feat_all = extract_feat(sd_s, torch.cat([img_syn, img]))
    feat_syn, feat_ori = feat_all[:B], feat_all[B:]
    cls_s, reg_s, ctr_s = head_forward(sd_s, feat_syn)
    loss_bbox, loss_ctr = R.loss_synthetic(reg_s, ctr_s, pts, syn_boxes)
    with torch.no_grad():
        cls, reg, ctr = head_forward(sd_t, extract_feat(sd_t, img))
        fc, fr, _ = R._flatten_head(cls, reg, ctr)
        pseudo = [R.generate_pseudo_single(pts, fc[i], fr[i], gt_points[i], gt_labels[i], gt_bboxes[i], 0.0) for i in range(B)]
    pb_c, pp_c = [p[0] for p in pseudo], [p[1] for p in pseudo]
    n = cfg['num_training_burninstep1']
    if all(s.shape[0] > 0 for s in syn_boxes):
        syn_t = [s[:n] for s in syn_boxes]
        pb_t, gb_t, pl_t = [b[:n] for b in pb_c], [b[:n] for b in gt_bboxes], [l[:n] for l in gt_labels]
        losses['coarse_bboxes_iou'] = R.bbox_overlaps(torch.cat(pb_t), torch.cat(gb_t), is_aligned=True).mean()
        for stage in range(cfg['num_stages']):
            ml, pb_t = mil_stage(sd_s, feat_ori, stage, cfg['fine_proposal_cfg'][stage],
                                 cfg['fine_proposal_extensive_cfg'][stage], pb_t, gb_t, pl_t, img_hw,
                                 inject[f'neg{stage}'], cfg['alpha'], cfg['top_k'], cfg['beta'], syn=(feat_syn, syn_t))
            losses[f'stage{stage}_refine_bboxes_iou'] = R.bbox_overlaps(torch.cat(pb_t), torch.cat(gb_t), is_aligned=True).mean()
            losses.update(ml)
        new_points = [(1 - cfg['lamda']) * R.bbox_xyxy_to_cxcywh(pb_c[i])[:, :2] + cfg['lamda'] * origin_points[i] for i in range(B)]
        real = R.bbox_xyxy_to_cxcywh(torch.cat(gt_bboxes))
        losses['refined_points_distance'] = (torch.sqrt((torch.cat(new_points) - real[:, :2]) ** 2)
                                             / torch.sqrt((real[:, 2:] / 2) ** 2)).mean()
    else:
        new_points = gt_points
    lc, _, _ = _aug_loss(sd_s, img, new_points, gt_labels, pp_c, gt_labels, pb_c, inject['aug'], pts)
    losses['loss_cls'], losses['loss_bbox'], losses['loss_centerness'] = lc, loss_bbox, loss_ctr
    return losses, new_points


def total_loss(losses):
    """base.py:197-198"""
    return sum(v for k, v in losses.items() if 'loss' in k)


def ema(sd_t, sd_s, alpha=0.999):
    """fcos_p2b_teacher_student.py:254-257 (parameters only)."""
    return {k: (R.ema_update(v, sd_s[k].detach(), alpha) if is_param(k) else v) for k, v in sd_t.items()}


def sgd_clip_step(sd_s, grads, mom, lr, first, momentum=0.9, wd=1e-4, max_norm=35.0):
    """mmcv OptimizerHook(grad_clip) + SGD with paramwise bias_lr_mult=2 / bias_decay_mult=0."""
    names = [k for k in sd_s if trainable(k) and grads.get(k) is not None]
    tot = math.sqrt(sum(float((grads[k].double() ** 2).sum()) for k in names))
    coef = min(1.0, max_norm / (tot + 1e-6))
    out, new_mom = dict(sd_s), {}
    for k in names:
        bias = k.endswith('.bias')
        g = grads[k] * coef + (0.0 if bias else wd) * sd_s[k].detach()
        buf = g if first else momentum * mom[k] + g
        new_mom[k] = buf
        out[k] = (sd_s[k].detach() - (lr * 2 if bias else lr) * buf)
    return out, new_mom, tot


# ---------------------------------------------------------------- CPU baseline --
SHAPE_LIST = [[20, 20, 0.5, 0.5], [10, 20, 0.5, 0.5], [30, 80, 0.5, 0.5], [20, 50, 0.5, 0.5], [30, 120, 0.5, 0.5],
              [30, 40, 0.5, 0.5]]                      # configs/point_teacher/aitodv2_point_teacher_0%.py (shape_list)
MODEL_CFG = dict(num_stages=1, num_training_burninstep1=100, num_training_burninstep2=100, lamda=1.0, alpha=[0.01, 0.25],
                 top_k=1, beta=0.25, shape_list=SHAPE_LIST,
                 fine_proposal_cfg=[dict(base_ratios=[1.0], shake_ratio=None, min_scale=0)],
                 fine_proposal_extensive_cfg=[dict(base_ratios=[1.0, 1.2, 1.3, 0.8, 0.7], shake_ratio=None, min_scale=4)])


def model_cfg_from(model):
    """The keys the oracle reads, out of a `TS_P2B_FCOS` config dictionary (configs/point_teacher/*.py: plain data)."""
    tc = model['train_cfg']
    pick = lambda c: dict(base_ratios=c['base_ratios'], shake_ratio=c['shake_ratio'], min_scale=c['min_scale'])   # noqa: E731
    return dict(num_stages=model['num_stages'], num_training_burninstep1=model['num_training_burninstep1'],
                num_training_burninstep2=model['num_training_burninstep2'], lamda=model['lamda'], alpha=list(model['alpha']),
                top_k=model['_model_']['bbox_head']['top_k'], beta=model['_model_']['bbox_head']['beta'],
                shape_list=[list(s) for s in model['shape_list']], point=model['_point_'],
                fine_proposal_cfg=[pick(c) for c in tc['fine_proposal_cfg']],
                fine_proposal_extensive_cfg=[pick(c) for c in tc['fine_proposal_extensive_cfg']])


def _synthetic_batch(batch, size, objects, seed=7):
    g = torch.Generator().manual_seed(seed)
    img = (torch.rand(batch, 3, size, size, generator=g) * 80 + 70).round()
    boxes, labels = [], []
    for _ in range(batch):
        G = max(1, int(objects))
        c = torch.rand(G, 2, generator=g) * (size - 32) + 16
        wh = torch.exp(torch.randn(G, 2, generator=g) * 0.5 + math.log(12.0)).clamp(2, 64)
        boxes.append(torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, size))
        labels.append(torch.randint(0, NUM_CLASSES, (G,), generator=g))
    return img, boxes, labels


def effective_cpus():
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota).  The GPU boxes
    expose 256 logical CPUs but cap the container at 16; sizing thread pools by os.cpu_count()
    there oversubscribes 16x and the CPU leg takes minutes."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(workload='step1', batch=2, size=800, objects=300, iters=1):
    """Time the oracle's full iteration (forward + backward + clip + SGD + EMA) on the host
    cores for `iters` iterations after one untimed warm-up at a reduced size.  Returns the
    dict bench.py puts under "cpu_baseline"."""
    cores = effective_cpus()
    torch.set_num_threads(cores)
    import sys
    print(f'[cpu_baseline] timing {iters} oracle iteration(s) on {cores} host threads ...', file=sys.stderr, flush=True)
    g = torch.Generator().manual_seed(0)
    sd_s = init_detector_state(1)
    sd_t = init_detector_state(2)
    if workload == 'step2':
        for sd in (sd_s, sd_t):
            sd['bbox_head.conv_reg.bias'] = torch.ones(4)

    def one(size_, objects_):
        nonlocal sd_s, sd_t
        img, gb, gl = _synthetic_batch(batch, size_, objects_)
        gp = [R.bbox_xyxy_to_cxcywh(b)[:, :2] for b in gb]
        inject = dict(neg0=torch.rand(batch, 4, 200, generator=g), aug=(['horizontal', 'None'] * batch)[:batch],)
        inject['aug'] = (inject['aug'], ([0.9, 1.1] * batch)[:batch])
        params = {k: (v.clone().requires_grad_(True) if trainable(k) else v) for k, v in sd_s.items()}
        sd_t = ema(sd_t, sd_s)
        if workload == 'step1':
            # white rectangles of the size range the generator draws (the rasteriser itself is <1 % of the step)
            syn = [torch.cat([b[:, :2] - 10, b[:, :2] + 25], 1).clamp(0, size_ - 1)[: max(1, len(b) // 2)] for b in gb]
            img_syn = img.clone()
            inject.update(img_syn=img_syn, syn_boxes=syn)
            losses, _ = forward_train_step1(params, sd_t, img, gb, gl, gp, MODEL_CFG, inject)
        else:
            losses, _ = forward_train_step2(params, sd_t, img, gb, gl, gp, MODEL_CFG, inject)
        loss = total_loss(losses)
        names = [k for k in params if trainable(k)]
        gr = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
        sd_s, _, _ = sgd_clip_step(sd_s, dict(zip(names, gr)), {}, 0.005 / 3, True)
        return float(loss)
    one(256, 30)                                   # warm-up (allocator, thread pools)
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        one(size, objects)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return dict(value=round(1.0 / med, 5), unit='iters/s', cores=cores, kind='port',
                sample=f'median of {iters} full {workload} iteration(s) of oracle/ref_model.py (torch-CPU fp32, {cores} threads), '
                       f'bs {batch}, {size}x{size}, {objects} pts/img, after one 256x256 warm-up; '
                       f'{sum(times):.1f} s in all (min {times[0]:.1f} / max {times[-1]:.1f} s per iteration)')
