"""Plane-native convolutions of the trunk (csrc/gemm_split.hip, pt_conv_bf16x6 / pt_conv_wgrad_bf16x6).

Every fp32 convolution of backbone / necks / towers is an implicit GEMM on the bf16 matrix cores whose operands are three-term
bf16 splits (x = x0 + x1 + x2 exactly; six products, fp32 accumulation).  Instead of splitting every layer's fp32 input again, the
activations TRAVEL as row-major split planes `[3, (pixels + 1) * C]` (last row zeros): a convolution's epilogue applies the frozen
BatchNorm, the identity add and the ReLU and emits the planes its consumer stages straight into LDS; the backward emits gradient
planes the same way (ReLU mask of the producer applied in the input-gradient kernel's epilogue, BatchNorm scale folded into the
input-gradient weights and into the weight gradient's reduction, bias gradients formed by the weight-gradient kernel).

Reference call sites: the Bottleneck of HBB_TOD/mmdet/models/backbones/resnet.py:262-303 (`caffe` style: stride on conv1,
:153-158), FPN laterals / output convolutions necks/fpn.py:151-202, PSAGG necks/ps_fpn.py:56-75, the towers
dense_heads/anchor_free_head.py:198-219.

Gradient convention: the gradient of a plane tensor is a plane tensor of the same shape that is ALREADY multiplied by the ReLU mask
of the activation it belongs to (its consumer holds that activation as its input and applies the mask in its epilogue)."""
import torch

from . import functional as F
from . import hip

f32 = torch.float32
bf16 = torch.bfloat16


class PlaneAct:
    """An NHWC activation [B, H, W, C] as row-major split planes `t` = bf16 [3, (B*H*W + 1) * C]; `relu`: its values went through a
    ReLU (so `plane 0 > 0` is the mask its gradient needs)."""
    __slots__ = ('t', 'B', 'H', 'W', 'C', 'relu', 'gcarrier')
    dtype = f32                         # what the planes represent

    def __init__(self, t, B, H, W, C, relu, gcarrier=False):
        self.t, self.B, self.H, self.W, self.C, self.relu = t, B, H, W, C, relu
        self.gcarrier = gcarrier         # the gradient of `t` travels as fp32 inside a tensor of t's shape (functional._RoIAlignPlanes)

    @property
    def P(self):
        return self.B * self.H * self.W

    @property
    def np(self):
        """Planes per value: 3 (fp32 = x0 + x1 + x2) or 1 (a bf16 NHWC tensor with a zero row behind it: the bf16 trunk)."""
        return self.t.shape[0]

    def tensor(self):
        """The activation as a [B, C, H, W] channels_last tensor, differentiable: fp32 (np = 3, one pass) or bf16 (np = 1: a VIEW of the
        plane - the backward copies the gradient behind a zero row)."""
        return self.float() if self.np == 3 else _PlaneView.apply(self.t, self)

    @property
    def shape(self):
        return (self.B, self.C, self.H, self.W)

    @property
    def f16(self):
        """Two fp16 planes (value = h0 + h1): the operand format of the three-product kernels (functional.F16_FC)."""
        return self.t.dtype == torch.float16

    def float(self):
        """-> fp32 [B, C, H, W] channels_last tensor (exact: x0 + x1 + x2), differentiable."""
        assert not self.f16, 'fp16 planes feed the first FC layer only (planes.linear)'
        return _PlanesToF32.apply(self.t, self)


class _PlaneView(torch.autograd.Function):
    """One bf16 plane [1, (P + 1) * C] <-> the bf16 channels_last tensor [B, C, H, W] it is (a view, no launch); backward: the
    gradient is copied behind a zero row (and masked by the activation's ReLU, the gradient convention of this module)."""

    @staticmethod
    def forward(ctx, t, act):
        ctx.act = act
        ctx.save_for_backward(t if act.relu else None)
        return t[0, :act.P * act.C].view(act.B, act.H, act.W, act.C).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        act = ctx.act
        t, = ctx.saved_tensors
        gp, _, _ = split_nhwc(g.to(bf16).contiguous(memory_format=torch.channels_last), np=1)
        if act.relu:
            gp, _ = combine(gp, mask=t)
        return gp, None


class _F32ToPlanes(torch.autograd.Function):
    """fp32 channels_last [B, C, H, W] -> planes; backward: the gradient planes summed back to fp32."""

    @staticmethod
    def forward(ctx, x):
        t, _, _ = split_nhwc(x)
        ctx.shape = x.shape
        return t

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        _, of = combine(g.contiguous(), n=B * H * W * C, want_planes=False, want_f32=True)
        return of.view(B, H, W, C).permute(0, 3, 1, 2)


class _F32ToPlanes2(torch.autograd.Function):
    """One split of an fp32 map for TWO consumers (the two towers of the dense head): backward adds the two gradient plane sets and
    returns fp32 in one pass."""

    @staticmethod
    def forward(ctx, x):
        t, _, _ = split_nhwc(x)
        ctx.shape = x.shape
        return t.view_as(t), t.view_as(t)

    @staticmethod
    def backward(ctx, g1, g2):
        B, C, H, W = ctx.shape
        if g1 is None or g2 is None:
            g1, g2 = (g1 if g2 is None else g2), None
        _, of = combine(g1.contiguous(), g2.contiguous() if g2 is not None else None, n=B * H * W * C, want_planes=False, want_f32=True)
        return of.view(B, H, W, C).permute(0, 3, 1, 2)


class _F32ToF16Planes(torch.autograd.Function):
    """fp32 channels_last [B, C, H, W] -> two fp16 planes (value = h0 + h1) for ONE or TWO consumers whose input gradients come back
    as fp32 CARRIERS (tensors of the planes' shape holding the fp32 gradient in their head: planes._PlaneConv with x_gcarrier)."""

    @staticmethod
    def forward(ctx, x, two):
        t, _, _ = split_nhwc(x, f16=True)
        ctx.shape = x.shape
        return (t.view_as(t), t.view_as(t)) if two else t.view_as(t)

    @staticmethod
    def backward(ctx, g1, g2=None):
        B, C, H, W = ctx.shape
        n = B * H * W * C
        gs = [g.contiguous().view(-1).view(f32)[:n] for g in (g1, g2) if g is not None]
        g = gs[0] if len(gs) == 1 else gs[0] + gs[1]
        return g.view(B, H, W, C).permute(0, 3, 1, 2), None


def to_planes2(x, f16=False):
    """x as planes for two consumers -> (PlaneAct, PlaneAct) sharing one set of planes.  f16: two fp16 planes - the consumers run on
    three fp16 MFMA products (functional.F16_FC) and return their input gradients as fp32 carriers."""
    B, C, H, W = x.shape
    if f16:
        a, b = _F32ToF16Planes.apply(x, True)
        return PlaneAct(a, B, H, W, C, False, gcarrier=True), PlaneAct(b, B, H, W, C, False, gcarrier=True)
    a, b = _F32ToPlanes2.apply(x)
    return PlaneAct(a, B, H, W, C, False), PlaneAct(b, B, H, W, C, False)


class _Bf16ToPlane(torch.autograd.Function):
    """bf16 channels_last [B, C, H, W] -> one plane (a copy behind which the zero row sits); backward: the plane viewed as a tensor."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return split_nhwc(x, np=1)[0]

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        return g.contiguous()[0, :B * H * W * C].view(B, H, W, C).permute(0, 3, 1, 2)


def to_planes(x, f16=False):
    """Differentiable entry into plane mode (relu=False: the producer of x masks its own gradient): fp32 -> three planes, bf16 -> one;
    f16: fp32 -> two fp16 planes (see to_planes2)."""
    B, C, H, W = x.shape
    if f16 and x.dtype == f32:
        return PlaneAct(_F32ToF16Planes.apply(x, False), B, H, W, C, False, gcarrier=True)
    return PlaneAct((_Bf16ToPlane if x.dtype == bf16 else _F32ToPlanes).apply(x), B, H, W, C, False)


def _u16(t):
    """Device pointer of a 16-bit plane tensor (bf16, or fp16 planes: the header's uint16_t* is 16-bit storage)."""
    if t is None:
        return 0
    return hip.dptr(t.view(bf16) if t.dtype == torch.float16 else t, 'uint16_t')


def to_f16_planes(t):
    """Three bf16 planes -> (two fp16 planes of s * value, device tensor [1] holding 1 / s): s = the power of two that brings the
    tensor's largest magnitude into [512, 1024), chosen on the device (pt_planes_to_f16 with scale 0: gradient magnitudes follow the
    loss normalisation - 1e-3 per element at K = 5 000 RoIs, 1e-8 at K = 60 750 - and fp16 keeps 22 bits only above 0.125)."""
    out = torch.empty((2, t.shape[1]), dtype=torch.float16, device=t.device)
    buf = torch.empty((2 + 1024,), dtype=f32, device=t.device)
    hip.call('pt_planes_to_f16', t, t.shape[1], t.shape[1], 0.0, out.view(bf16), out.shape[1], buf, buf[2:])
    return out, buf[1:2]


_F16_SCALES = {}      # data_ptr of an fp16 gradient plane tensor -> device tensor [1] with 1 / its scale (from its producer to its consumer)


def out_hw(H, W, K, stride, pad):
    return (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1


def _new_planes(rows, C, device, zero=False, np=3):
    n = (rows + 1) * C
    return (torch.zeros if zero else torch.empty)((np, n), dtype=bf16, device=device)


_SPLITS = {}


def _splits(fn, *shape):
    """The launch's chunk count by the library's cost model (a pure function of the shape: asked once per shape)."""
    k = (fn,) + shape
    v = _SPLITS.get(k)
    if v is None:
        v = _SPLITS[k] = hip.call(fn, *shape)
    return v


def launch_conv(x_t, B, Hs, Ws, Cin, wp, Cout, K, stride, pad, scale=None, shift=None, res_planes=None, res_f32=None,
                mask_planes=None, relu=False, want_planes=False, want_f32=False, scatter=None, tile_rows=0, splits=None, f32_out=None,
                transposed_out=None, alpha=None, alpha_dev=None, out_f16=False):
    """pt_conv_bf16x6: x_t row-major planes of [B*Hs*Ws (+1), Cin]; wp SplitPlanes of the weight.  -> (planes or None, fp32 rows or
    None).  scatter = (H, W): a stride-2 input gradient placed at (2y, 2x) of a zeroed [B, H, W] grid."""
    Ho, Wo = out_hw(Hs, Ws, K, stride, pad) if transposed_out is None else transposed_out
    M = B * Ho * Wo
    rows = M if scatter is None else B * scatter[0] * scatter[1]
    dev = x_t.device
    f16 = x_t.dtype == torch.float16                          # fp16 x 2 operands, three products; the epilogue's planes stay bf16 x 3
    np_ = 3 if f16 else x_t.shape[0]
    if want_planes and out_f16:                               # the next layer's fp16 operand
        assert np_ == 3 and scatter is None
        out_p = torch.empty((2, (rows + 1) * Cout), dtype=torch.float16, device=dev)
    else:
        out_p = _new_planes(rows, Cout, dev, zero=scatter is not None, np=np_) if want_planes else None
    out_f = (torch.zeros if scatter is not None else torch.empty)((rows, Cout), dtype=f32, device=dev) if want_f32 else None
    if f32_out is not None:                                   # a caller-provided fp32 buffer of rows * Cout elements
        assert f32_out.numel() == rows * Cout and f32_out.dtype == f32 and scatter is None
        out_f = f32_out
    d = hip.STRUCTS['pt_conv_desc']()
    d.B, d.Hs, d.Ws, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad, d.relu = B, Hs, Ws, Cin, Cout, K, K, stride, pad, int(bool(relu))
    d.np = np_
    d.operand_f16 = int(f16)
    d.alpha = float(alpha) if alpha is not None else 0.0
    d.alpha_dev = hip.dptr(alpha_dev, 'float')
    assert f16 == (wp.planes.shape[0] == 2), 'fp16 activations go with fp16 weight planes'
    d.x_planes, d.x_plane_stride = _u16(x_t), x_t.shape[1]
    d.w_planes, d.w_plane_stride = hip.dptr(wp.planes, 'uint16_t'), wp.planes.shape[1]
    d.scale, d.shift = hip.dptr(scale, 'float'), hip.dptr(shift, 'float')
    if res_planes is not None:
        d.res_planes, d.res_plane_stride = hip.dptr(res_planes, 'uint16_t'), res_planes.shape[1]
    d.res_f32 = hip.dptr(res_f32, 'float')
    d.mask_planes = _u16(mask_planes)
    d.out_f32 = hip.dptr(out_f, 'float')
    if out_p is not None:
        d.out_planes, d.out_plane_stride = _u16(out_p), out_p.shape[1]
        d.out_f16 = int(out_p.dtype == torch.float16)
    if scatter is not None:
        d.scatter_stride, d.scatter_H, d.scatter_W = 2, scatter[0], scatter[1]
    d.tile_rows = int(tile_rows)
    if transposed_out is not None:            # the input gradient of a 3x3 stride-2 convolution onto its input grid (Ho, Wo)
        d.dstride, d.out_H, d.out_W = 2, Ho, Wo
        splits = 1
    S = _splits('pt_conv_bf16x6_splits', B, Hs, Ws, Cin, Cout, K, K, stride, pad, int(tile_rows)) if splits is None else int(splits)
    if S > 1:
        ws = torch.empty((S * M * Cout,), dtype=f32, device=dev)
        d.workspace, d.workspace_elems, d.splits = hip.dptr(ws, 'float'), ws.numel(), S
    else:
        d.splits = 1
    assert wp.rows == Cout and wp.k == K * K * Cin, (wp.rows, wp.k, Cout, K, Cin)
    assert x_t.shape[1] >= (B * Hs * Ws + 1) * Cin
    if mask_planes is not None:
        assert mask_planes.shape[1] >= rows * Cout
    if res_planes is not None:
        assert res_planes.shape[1] >= M * Cout
    if res_f32 is not None:
        assert res_f32.numel() >= M * Cout
    hip.call('pt_conv_bf16x6', d)
    return out_p, out_f


def wgrad_ok(Cin, Cout):
    return Cin % 128 == 0 and Cout % 128 == 0


def launch_wgrad(gy_t, x_t, B, Hs, Ws, Cin, Cout, K, stride, pad, row_scale=None, want_bias=False, bn=None, w=None, alpha=None,
                 alpha_dev=None):
    """pt_conv_wgrad_bf16x6 -> (dw as a channels_last [Cout, Cin, K, K] tensor, dbias or None).
    bn = (scale, rstd, mean) of a TRAINABLE eval-mode BatchNorm behind the convolution (w = its weight): -> (dw, dbeta, dgamma)
    through pt_bn_wgrad_finish."""
    if bn is not None:
        row_scale, want_bias = None, True
    Ho, Wo = out_hw(Hs, Ws, K, stride, pad)
    dev = gy_t.device
    S = _splits('pt_conv_wgrad_bf16x6_splits', B, Ho, Wo, K, K, Cin, Cout)
    n = Cout * K * K * Cin
    ws = torch.empty((S * (n + (Cout if want_bias else 0)),), dtype=f32, device=dev)
    dw = torch.empty((Cout, K, K, Cin), dtype=f32, device=dev)
    db = torch.empty((Cout,), dtype=f32, device=dev) if want_bias else None
    d = hip.STRUCTS['pt_conv_wgrad_desc']()
    d.B, d.Hs, d.Ws, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad, d.accumulate = B, Hs, Ws, Cin, Cout, K, K, stride, pad, 0
    f16 = gy_t.dtype == torch.float16
    d.np = 3 if f16 else gy_t.shape[0]
    d.operand_f16 = int(f16)
    d.alpha = float(alpha) if alpha is not None else 0.0
    d.alpha_dev = hip.dptr(alpha_dev, 'float')
    assert x_t.shape[0] == gy_t.shape[0] and x_t.dtype == gy_t.dtype
    d.gy_planes, d.gy_plane_stride = _u16(gy_t), gy_t.shape[1]
    d.x_planes, d.x_plane_stride = _u16(x_t), x_t.shape[1]
    d.dw, d.dbias, d.row_scale = hip.dptr(dw, 'float'), hip.dptr(db, 'float'), hip.dptr(row_scale, 'float')
    d.workspace, d.workspace_elems, d.splits = hip.dptr(ws, 'float'), ws.numel(), S
    assert gy_t.shape[1] >= (B * Ho * Wo + 1) * Cout and x_t.shape[1] >= (B * Hs * Ws + 1) * Cin
    hip.call('pt_conv_wgrad_bf16x6', d)
    if bn is not None:
        scale, rstd, mean = bn
        dg = torch.empty((Cout,), dtype=f32, device=dev)
        wr = w.detach().permute(0, 2, 3, 1)
        assert wr.is_contiguous()
        hip.call('pt_bn_wgrad_finish', dw, wr, Cout, K * K * Cin, scale, rstd, mean, db, dg)
        return dw.permute(0, 3, 1, 2), db, dg
    return dw.permute(0, 3, 1, 2), db


def split_nhwc(x, stride=1, np=None, f16=False):
    """channels_last [B, C, H, W] (fp32, or bf16 -> np = 1) -> row-major planes of the pixels (y * stride, x * stride)
    (pt_split_bf16x3_gather).  np: 3 (fp32 as x0 + x1 + x2) or 1 (one bf16 plane); default by the dtype."""
    B, C, H, W = x.shape
    rows = x.permute(0, 2, 3, 1)
    is16 = x.dtype == bf16
    np = (1 if is16 else 3) if np is None else np
    assert rows.is_contiguous() and x.dtype in (f32, bf16) and (np == 1 or not is16)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if f16:                                                     # two fp16 planes (value = h0 + h1)
        assert not is16
        t = torch.empty((2, (B * Ho * Wo + 1) * C), dtype=torch.float16, device=x.device)
        hip.call('pt_split_bf16x3_gather', rows, 0, C, B, H, W, C, stride, 2, t.view(bf16), t.shape[1])
        return t, Ho, Wo
    t = _new_planes(B * Ho * Wo, C, x.device, np=np)
    hip.call('pt_split_bf16x3_gather', rows, int(is16), C, B, H, W, C, stride, np, t, t.shape[1])
    return t, Ho, Wo


def combine(a, b=None, c=None, mask=None, n=None, want_planes=True, want_f32=False):
    """pt_planes_combine over the first n elements of each plane: split(mask * (a + b + c)) and / or its fp32 value."""
    n = n if n is not None else a.shape[1]
    out = torch.empty_like(a) if want_planes else None
    of = torch.empty((n,), dtype=f32, device=a.device) if want_f32 else None
    hip.call('pt_planes_combine', a, a.shape[1], b, b.shape[1] if b is not None else 0, c, mask, n, a.shape[0], out,
             out.shape[1] if out is not None else 0, of)
    assert out is None or n == a.shape[1], 'planes out: combine whole planes (the zero rows included)'
    return out, of


class _PlanesToF32(torch.autograd.Function):
    """planes -> fp32 [B, C, H, W] channels_last; backward: split of the (masked) fp32 gradient."""

    @staticmethod
    def forward(ctx, t, act):
        _, of = combine(t, n=act.P * act.C, want_planes=False, want_f32=True)
        ctx.act = act
        ctx.save_for_backward(t if act.relu else None)
        return of.view(act.B, act.H, act.W, act.C).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        act = ctx.act
        t, = ctx.saved_tensors
        rows = g.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(act.P, act.C)
        gp = F._split_rows(rows, relu_of=None)
        if act.relu:
            gp, _ = combine(gp, mask=t)
        return gp, None


# ------------------------------------------------------------------------------------------------ single convolution --
class ConvCfg:
    """Static description of one plane convolution call (not a tensor: autograd passes it through)."""
    __slots__ = ('B', 'H', 'W', 'Cin', 'Cout', 'K', 'stride', 'pad', 'relu', 'x_planes', 'x_relu', 'out_planes', 'scale', 'bias_grad', 'x_gcarrier', 'f16_out', 'x_f16')

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


class _PlaneConv(torch.autograd.Function):
    """y = [relu](conv(x, w) * scale + shift) for ONE consumer of x (stride 1 in the backward's input gradient).
    x: plane tensor (cfg.x_planes) or fp32 channels_last [B, Cin, H, W]; y: plane tensor (cfg.out_planes) or fp32 channels_last."""

    @staticmethod
    def forward(ctx, x, w, shift, cfg):
        c = cfg
        if c.x_planes:
            xt = x
        else:
            xt, _, _ = split_nhwc(x, f16=bool(c.x_f16))            # fp32 -> three planes (or two fp16 ones), bf16 -> one
        np_ = xt.shape[0]
        f16 = xt.dtype == torch.float16                            # fp16 x 2 operands (np_ == 2): three products, weights stored x 16
        assert not f16 or c.scale is None
        wp = F._conv_weight_planes(w, False, None, np_)
        as_planes = c.out_planes or np_ == 1                       # (the one-plane result IS the bf16 tensor)
        yp, yf = launch_conv(xt, c.B, c.H, c.W, c.Cin, wp, c.Cout, c.K, c.stride, c.pad, scale=c.scale, shift=shift, relu=c.relu,
                             want_planes=as_planes, want_f32=not as_planes, alpha=1.0 / F.F16_WEIGHT_SCALE if f16 else None,
                             out_f16=bool(f16 and c.f16_out))
        Ho, Wo = out_hw(c.H, c.W, c.K, c.stride, c.pad)
        ctx.cfg = c
        ctx.np = np_
        ctx.f16 = f16
        ctx.out_hw = (Ho, Wo)
        need_x_for_mask = c.x_planes and c.x_relu
        keep_y = c.relu and not c.out_planes
        ctx.save_for_backward(xt if (w.requires_grad or need_x_for_mask) else None, w, (yp if np_ == 1 else yf) if keep_y else None)
        if c.out_planes:
            return yp
        if np_ == 1:
            return yp[0, :c.B * Ho * Wo * c.Cout].view(c.B, Ho, Wo, c.Cout).permute(0, 3, 1, 2)
        return yf.view(c.B, Ho, Wo, c.Cout).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        c = ctx.cfg
        xt, w, yf = ctx.saved_tensors
        Ho, Wo = ctx.out_hw
        M = c.B * Ho * Wo
        np_ = ctx.np
        inv_s = None
        if c.out_planes:
            E = g.contiguous()
            if E.dtype == torch.float16:                           # an fp16 gradient chain: the scale its first link chose travels beside it
                inv_s = _F16_SCALES.pop(E.data_ptr())
        elif np_ == 1:                                             # a bf16 gradient tensor: behind a zero row, masked by the result's ReLU
            E, _, _ = split_nhwc(g.to(bf16).contiguous(memory_format=torch.channels_last), np=1)
            if yf is not None:
                E, _ = combine(E, mask=yf)
        else:
            rows = g.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(M, c.Cout)
            E = F._split_rows(rows, relu_of=yf)
        gx = gw = gb = None
        f16 = ctx.f16
        a_dx = a_dw = None
        if f16:
            # fp16 x 2 operands: the gradient enters as two fp16 planes of s * g, s a power of two chosen on the device from the
            # tensor's largest magnitude (unscaled, 1e-3 ... 1e-9 would sit in fp16's subnormal range); the kernels' alpha scales the
            # results back (powers of two: exact).  A chain of fp16 layers keeps the scale of its first link (saturating at 60 000)
            if E.dtype != torch.float16:
                E, inv_s = to_f16_planes(E)
            a_dx, a_dw = 1.0 / F.F16_WEIGHT_SCALE, 1.0
        if ctx.needs_input_grad[0]:
            assert c.stride == 1, 'the single-consumer plane convolution back-propagates stride 1 only'
            wd = F._conv_weight_planes(w, True, c.scale, np_)
            if c.x_planes and c.x_gcarrier:
                # the producer of x (RoIAlign) wants its gradient as fp32: it travels in the head of a tensor of x's shape
                gx = torch.empty((np_ if f16 else 3, (c.B * c.H * c.W + 1) * c.Cin), dtype=torch.float16 if f16 else bf16, device=E.device)
                launch_conv(E, c.B, Ho, Wo, c.Cout, wd, c.Cin, c.K, 1, c.K - 1 - c.pad,
                            f32_out=gx.view(-1).view(f32)[:c.B * c.H * c.W * c.Cin], alpha=a_dx, alpha_dev=inv_s)
            elif f16 and not c.x_planes:                           # an fp32 input split here: its gradient leaves as fp32
                _, gf = launch_conv(E, c.B, Ho, Wo, c.Cout, wd, c.Cin, c.K, 1, c.K - 1 - c.pad, want_f32=True, alpha=a_dx, alpha_dev=inv_s)
                gx = gf.view(c.B, c.H, c.W, c.Cin).permute(0, 3, 1, 2)
            elif f16:
                # the producer of x is another fp16 layer: its output gradient as fp16 planes of s * g, masked by x's ReLU
                gx, _ = launch_conv(E, c.B, Ho, Wo, c.Cout, wd, c.Cin, c.K, 1, c.K - 1 - c.pad, mask_planes=xt if c.x_relu else None,
                                    want_planes=True, out_f16=True, alpha=a_dx)
                _F16_SCALES[gx.data_ptr()] = inv_s
            else:
                xp_out = c.x_planes or np_ == 1
                gp, gf = launch_conv(E, c.B, Ho, Wo, c.Cout, wd, c.Cin, c.K, 1, c.K - 1 - c.pad,
                                     mask_planes=xt if (c.x_planes and c.x_relu) else None, want_planes=xp_out, want_f32=not xp_out)
                if c.x_planes:
                    gx = gp
                elif np_ == 1:
                    gx = gp[0, :c.B * c.H * c.W * c.Cin].view(c.B, c.H, c.W, c.Cin).permute(0, 3, 1, 2)
                else:
                    gx = gf.view(c.B, c.H, c.W, c.Cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw, gb = launch_wgrad(E, xt, c.B, c.H, c.W, c.Cin, c.Cout, c.K, c.stride, c.pad, row_scale=c.scale,
                                  want_bias=bool(c.bias_grad and ctx.needs_input_grad[2]), alpha=a_dw, alpha_dev=inv_s)
            if w.dim() == 2:
                gw = gw.reshape(w.shape)
        return gx, gw, gb, None


def plane_conv_ok(conv, x_channels_last_f32_or_planes=True):
    """A convolution pt_conv_bf16x6 / pt_conv_wgrad_bf16x6 take: 1x1 (pad 0) or 3x3 (pad 1), stride 1, one group, no dilation,
    channels multiples of 128 (the weight gradient's tile), channels_last fp32 weight."""
    k = conv.kernel_size
    return (type(conv) is torch.nn.Conv2d and k in ((1, 1), (3, 3)) and conv.stride == (1, 1) and conv.padding == ((k[0] - 1) // 2,) * 2
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels % 128 == 0 and conv.out_channels % 128 == 0
            and conv.weight.is_cuda and conv.weight.dtype == f32 and F._ConvWeightPlanes.ok(conv.weight)
            and autocast_ok())


def conv_module(x, conv, relu=False, out_planes=False, scale=None, shift=None, f16=False):
    """[relu](conv(x) (* scale) + bias) through the plane kernels.  x: PlaneAct or fp32 channels_last tensor.
    -> PlaneAct (out_planes) or fp32 channels_last tensor.  f16 (an fp32 tensor in, an fp32 tensor out): the operands as two fp16
    planes, three MFMA products (functional.F16_FC: the compute-bound 3x3 convolutions of the necks)."""
    is_p = isinstance(x, PlaneAct)
    B, Cin, H, W = x.shape
    k = conv.kernel_size[0]
    cfg = ConvCfg(B=B, H=H, W=W, Cin=Cin, Cout=conv.out_channels, K=k, stride=1, pad=conv.padding[0], relu=bool(relu), x_planes=is_p,
                  x_relu=bool(is_p and x.relu), out_planes=bool(out_planes), scale=scale, bias_grad=scale is None and conv.bias is not None,
                  x_gcarrier=bool(is_p and x.gcarrier), f16_out=bool(is_p and x.f16 and out_planes),      # an fp16 chain stays in fp16 planes
                  x_f16=bool(f16 and not is_p and not out_planes and scale is None and x.dtype == f32))
    sh = shift if shift is not None else conv.bias
    y = _PlaneConv.apply(x.t if is_p else x, conv.weight, sh, cfg)
    if out_planes:
        return PlaneAct(y, B, H, W, conv.out_channels, bool(relu))
    return y                                                      # fp32 (three-plane operands) or bf16 (one plane) [B, Cout, H, W]


def linear_ok(fc, rows):
    """torch.nn.Linear layers the plane kernels take as 1x1 convolutions over `rows` pixels (both widths multiples of 128)."""
    return (fc.in_features % 128 == 0 and fc.out_features % 128 == 0 and fc.weight.is_cuda and fc.weight.dtype == f32
            and fc.weight.is_contiguous() and rows > 0 and not torch.is_autocast_enabled())


def linear(x, fc, relu=False, out_planes=False):
    """[relu](x W^T + b) with the rows of x as the pixels of a 1x1 convolution (fcos_head_p2b_ts.py:1202-1236 FC stacks).
    x: PlaneAct [B=1, H=rows, W=1, C=in_features] or fp32 [rows, in_features] -> PlaneAct or fp32 [rows, out_features]."""
    is_p = isinstance(x, PlaneAct)
    rows = x.H if is_p else x.shape[0]
    cfg = ConvCfg(B=1, H=rows, W=1, Cin=fc.in_features, Cout=fc.out_features, K=1, stride=1, pad=0, relu=bool(relu), x_planes=is_p,
                  x_relu=bool(is_p and x.relu), out_planes=bool(out_planes), scale=None, bias_grad=fc.bias is not None,
                  x_gcarrier=bool(is_p and x.gcarrier),
                  # an fp32 block (the oriented head's RoIAlignRotated output) into a layer as large as the first FC layer: split
                  # into two fp16 planes here, three MFMA products (functional.F16_FC)
                  x_f16=bool(F.F16_FC and not is_p and fc.in_features >= 4096 and x.dtype == f32))
    cfg.f16_out = bool(out_planes and ((is_p and x.f16) or cfg.x_f16))      # the next layer of the stack stays on fp16 operands
    xin = x.t if is_p else x.view(1, rows, 1, fc.in_features).permute(0, 3, 1, 2)
    y = _PlaneConv.apply(xin, fc.weight, fc.bias, cfg)
    if out_planes:
        return PlaneAct(y, 1, rows, 1, fc.out_features, bool(relu))
    return y.permute(0, 2, 3, 1).reshape(rows, fc.out_features)


def f32_ok(x):
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == f32 and x.dim() == 4 and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and x.numel() > 0)


def bf16_ok(x):
    """A bf16 channels_last map (the autocast trunk of BASELINE configs[2]) the one-plane kernels take."""
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == bf16 and x.dim() == 4 and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and x.numel() > 0)


def dense_ok(x):
    """fp32 outside autocast (three-plane operands), bf16 inside a bf16 autocast region (one plane)."""
    if torch.is_autocast_enabled():
        return torch.get_autocast_dtype('cuda') == bf16 and bf16_ok(x)
    return f32_ok(x)


def autocast_ok():
    """No autocast, or bf16 autocast (one-plane operands)."""
    return (not torch.is_autocast_enabled()) or torch.get_autocast_dtype('cuda') == bf16


# ------------------------------------------------------------------------------------------------------- fan-out --
class _FanOut(torch.autograd.Function):
    """One plane activation, two consumers: the backward adds the two (already masked) gradient plane sets exactly
    (sum of six planes in fp32, split again) instead of autograd's bf16 addition of planes."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t), t.view_as(t)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None or g2 is None:
            return g1 if g2 is None else g2
        out, _ = combine(g1.contiguous(), g2.contiguous())          # (the zero rows add up to the zero row)
        return out


def fan_out(act):
    a, b = _FanOut.apply(act.t)
    return (PlaneAct(a, act.B, act.H, act.W, act.C, act.relu), PlaneAct(b, act.B, act.H, act.W, act.C, act.relu))


# ---------------------------------------------------------------------------------------------------- Bottleneck --
class BottleneckCfg:
    """B, H, W, Cin, planes: the block's input and width; s1 / s2: strides of conv1 / conv2 (`caffe`: (s, 1), `pytorch`: (1, s));
    bn[i] for i in ('1', '2', '3', 'd'): (scale, shift, rstd, mean) of the eval-mode BatchNorm behind conv i; bn_train: their
    affines take gradients."""
    __slots__ = ('B', 'H', 'W', 'Cin', 'planes', 's1', 's2', 'x_planes', 'x_relu', 'bn', 'has_ds', 'bn_train')

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


class _BottleneckP(torch.autograd.Function):
    """backbones/resnet.py:262-303 with eval-mode BatchNorms (frozen affine - the HBB configs - or trainable affine - OBB config 5),
    stride on conv1 (`caffe`, :153-158) or on conv2 (`pytorch`), plane-native:
        y1 = relu(bn1(conv1(x)))   y2 = relu(bn2(conv2(y1)))   out = relu(bn3(conv3(y2)) + identity)
    identity = x, or bn_d(conv_d(x)) in a stage's first block.  x: plane tensor (block input, a ReLU output) or - first block of the
    first trainable stage - the fp32 channels_last output of the frozen stem.  Four (three) launches forward; backward three
    input-gradient launches (the identity's gradient and the ReLU masks ride in their epilogues), the weight gradients with the
    BatchNorm scale in their reduction - or, for a trainable affine, with d gamma / d beta formed from the raw weight gradient and the
    bias sums of the same launch (pt_bn_wgrad_finish): no pass over an activation for either."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, wd, g1, b1, g2, b2, g3, b3, gd, bd, cfg):
        c = cfg
        p, s1, s2 = c.planes, c.s1, c.s2
        st = s1 * s2
        H1, W1 = out_hw(c.H, c.W, 1, st, 0)                       # the block's output grid
        if c.x_planes:
            xs, xH, xW, xst = x, c.H, c.W, 1                      # dense planes of x; strided reads map through them
        elif s1 > 1:
            xs, xH, xW = split_nhwc(x, s1)                        # `caffe`: only the stride's pixels are ever read (fp32 -> 3 planes, bf16 -> 1)
            xst = s1
        else:
            xs, xH, xW = split_nhwc(x)[0], c.H, c.W
            xst = 1
        np_ = xs.shape[0]
        r1 = s1 // xst                                            # strides on the planes at hand
        rd = st // xst
        Ha, Wa = out_hw(xH, xW, 1, r1, 0)                         # conv1's output grid

        def cw(w_, dgrad, sc=None):
            return F._conv_weight_planes(w_, dgrad, sc, np_)
        S = c.bn
        # conv2 (3x3: the block's compute-bound product) on fp16 x 2 operands / three MFMA products (functional.F16_FC): conv1's
        # epilogue writes y1 as two fp16 planes, conv2's weights are fp16 planes of 16 w (alpha = 1 / 16 in front of its BatchNorm)
        h16 = bool(F.F16_BLOCK3 and np_ == 3)
        y1, _ = launch_conv(xs, c.B, xH, xW, c.Cin, cw(w1, False), p, 1, r1, 0, scale=S['1'][0], shift=S['1'][1], relu=True, want_planes=True,
                            out_f16=h16)
        y2, _ = launch_conv(y1, c.B, Ha, Wa, p, F._conv_weight_planes(w2, False, None, 2) if h16 else cw(w2, False), p, 3, s2, 1,
                            scale=S['2'][0], shift=S['2'][1], relu=True, want_planes=True, alpha=1.0 / F.F16_WEIGHT_SCALE if h16 else None)
        if c.has_ds and np_ == 1:                                 # bf16 trunk: the identity travels as bf16 too
            idn, _ = launch_conv(xs, c.B, xH, xW, c.Cin, cw(wd, False), 4 * p, 1, rd, 0, scale=S['d'][0], shift=S['d'][1], want_planes=True)
            out, _ = launch_conv(y2, c.B, H1, W1, p, cw(w3, False), 4 * p, 1, 1, 0, scale=S['3'][0], shift=S['3'][1], res_planes=idn, relu=True,
                                 want_planes=True)
        elif c.has_ds:
            _, idn = launch_conv(xs, c.B, xH, xW, c.Cin, cw(wd, False), 4 * p, 1, rd, 0, scale=S['d'][0], shift=S['d'][1], want_f32=True)
            out, _ = launch_conv(y2, c.B, H1, W1, p, cw(w3, False), 4 * p, 1, 1, 0, scale=S['3'][0], shift=S['3'][1], res_f32=idn, relu=True,
                                 want_planes=True)
        else:
            out, _ = launch_conv(y2, c.B, H1, W1, p, cw(w3, False), 4 * p, 1, 1, 0, scale=S['3'][0], shift=S['3'][1], res_planes=xs, relu=True,
                                 want_planes=True)
        ctx.cfg, ctx.geo = c, (H1, W1, Ha, Wa, xH, xW, r1, rd)
        ctx.save_for_backward(xs, y1, y2, w1, w2, w3, wd)
        return out

    @staticmethod
    def backward(ctx, E):
        c = ctx.cfg
        H1, W1, Ha, Wa, xH, xW, r1, rd = ctx.geo
        xs, y1, y2, w1, w2, w3, wd = ctx.saved_tensors
        p, s2 = c.planes, c.s2
        S = c.bn
        E = E.contiguous()
        np_ = E.shape[0]

        def cw(w_, dgrad, sc=None):
            return F._conv_weight_planes(w_, dgrad, sc, np_)
        need_x = ctx.needs_input_grad[0]
        gw = {}                                                   # conv -> (dw, dbeta, dgamma)

        def wgrad(tag, w, gy, xin, Hs_, Ws_, Ci, Co, K, stride, pad, need, alpha_dev=None):
            a = 1.0 if alpha_dev is not None else None
            if not need:
                gw[tag] = (None, None, None)
            elif c.bn_train:
                gw[tag] = launch_wgrad(gy, xin, c.B, Hs_, Ws_, Ci, Co, K, stride, pad, bn=(S[tag][0], S[tag][2], S[tag][3]), w=w,
                                       alpha=a, alpha_dev=alpha_dev)
            else:
                gw[tag] = (launch_wgrad(gy, xin, c.B, Hs_, Ws_, Ci, Co, K, stride, pad, row_scale=S[tag][0], alpha=a,
                                        alpha_dev=alpha_dev)[0], None, None)
        ni = ctx.needs_input_grad
        # conv3: gradient of y2 (masked by y2's ReLU), weight gradient
        E2, _ = launch_conv(E, c.B, H1, W1, 4 * p, cw(w3, True, S['3'][0]), p, 1, 1, 0, mask_planes=y2, want_planes=True)
        wgrad('3', w3, E, y2, H1, W1, p, 4 * p, 1, 1, 0, ni[3])
        # conv2 (a stride on it makes its input gradient a transposed convolution onto y1's grid).  y1 in fp16 planes (forward): the
        # product runs on fp16 x 2 operands - E2 converted with a power-of-two scale chosen on the device (two small launches against
        # half of a 3x3 product's MFMA work), weights x 16; the epilogue's alpha undoes both, E1 leaves as bf16 x 3 planes
        h16 = y1.dtype == torch.float16
        E2c, inv_s, a2 = E2, None, None
        if h16:
            E2c, inv_s = to_f16_planes(E2)
            a2 = 1.0 / F.F16_WEIGHT_SCALE
        wd2 = F._conv_weight_planes(w2, True, S['2'][0], 2) if h16 else cw(w2, True, S['2'][0])
        E1, _ = launch_conv(E2c, c.B, H1, W1, p, wd2, p, 3, 1, 1, mask_planes=y1, want_planes=True, alpha=a2, alpha_dev=inv_s,
                            transposed_out=(Ha, Wa) if s2 != 1 else None)
        wgrad('2', w2, E2c, y1, Ha, Wa, p, p, 3, s2, 1, ni[2], alpha_dev=inv_s)
        # conv1 (+ downsample) weights
        wgrad('1', w1, E1, xs, xH, xW, c.Cin, p, 1, r1, 0, ni[1])
        if c.has_ds:
            wgrad('d', wd, E, xs, xH, xW, c.Cin, 4 * p, 1, rd, 0, ni[4])
        else:
            gw['d'] = (None, None, None)
        gx = None
        if need_x:
            assert c.x_planes, 'a trainable stage behind an fp32 input is not plane-native'
            mask = xs if c.x_relu else None
            if not c.has_ds:
                gx, _ = launch_conv(E1, c.B, Ha, Wa, p, cw(w1, True, S['1'][0]), c.Cin, 1, 1, 0, res_planes=E, mask_planes=mask, want_planes=True)
            elif r1 == rd:
                # `caffe` (or no stride): conv1 and the downsample read the same pixels - the downsample path's gradient (fp32, coarse
                # grid) is added in conv1's input-gradient epilogue, the sum is scattered to the stride's pixels of a zeroed map
                _, gdx = launch_conv(E, c.B, H1, W1, 4 * p, cw(wd, True, S['d'][0]), c.Cin, 1, 1, 0, want_f32=True)
                gx, _ = launch_conv(E1, c.B, Ha, Wa, p, cw(w1, True, S['1'][0]), c.Cin, 1, 1, 0, res_f32=gdx, mask_planes=mask, want_planes=True,
                                    scatter=(c.H, c.W) if r1 == 2 else None)
            else:
                # `pytorch`: conv1 reads every pixel, the downsample the stride's - its gradient is scattered into a zeroed fp32 map
                # that conv1's (dense) input-gradient epilogue adds
                _, gdx = launch_conv(E, c.B, H1, W1, 4 * p, cw(wd, True, S['d'][0]), c.Cin, 1, 1, 0, want_f32=True, scatter=(c.H, c.W))
                gx, _ = launch_conv(E1, c.B, Ha, Wa, p, cw(w1, True, S['1'][0]), c.Cin, 1, 1, 0, res_f32=gdx, mask_planes=mask, want_planes=True)
        g = gw
        return (gx, g['1'][0], g['2'][0], g['3'][0], g['d'][0], g['1'][2], g['1'][1], g['2'][2], g['2'][1], g['3'][2], g['3'][1],
                g['d'][2], g['d'][1], None)
