"""Plane-native convolutions of the trunk (csrc/gemm_split.hip, pt_conv_bf16x6 / pt_conv_wgrad_bf16x6 / pt_planes_mix).

Every fp32 convolution of backbone / necks / towers / FC stacks is an implicit GEMM on the matrix cores whose operands are split
planes.  Two formats carry an fp32 value (round 5):

* **H2** - the default: two fp16 planes with a power-of-two scale, `torch.float16 [2, (pixels + 1) * C + 8]`: rows of pixels, a zero last
  row, and behind it (element `(pixels + 1) * C` of plane 0) an fp32 word - the TAIL - holding 1 / s; value = (h0 + h1) / s, 22
  significant bits, three MFMA products per fp32 product, 4 bytes per element.  Forward activations are stored unscaled (s = 1); a
  GRADIENT entering the plane region is scaled by the power of two that brings its largest magnitude into [128, 256) (chosen on
  the device, pt_planes_mix PT_SCALE_AUTO) and every input-gradient launch of the chain hands that scale on (the operations are
  linear: `out_inv_scale_src`); two chains that meet (a stage output feeding the next stage and an FPN lateral) are added exactly and
  continue with the smaller scale (PT_SCALE_MERGE); weight gradients multiply the tails back in (`alpha_dev`).
* **B3** - three bf16 planes `torch.bfloat16 [3, (pixels + 1) * C]` (x = x0 + x1 + x2 exactly, six products, 6 bytes): the fall-back
  with fp32's full range.  (B1: one bf16 plane = the bf16 autocast trunk of BASELINE configs[2].)

fp16 has 5 exponent bits: every H2 tensor a kernel writes is watched by a CENSUS (`Census`: saturated elements and the largest
stored magnitude per site, conditional atomics - free in the steady state) that the Trainer polls without synchronising; a site
that saturates (|stored| > 60 000) or whose largest stored magnitude falls below 0.25 (less than 23 bits against the tensor's maximum)
demotes its GROUP - 'trunk', 'neck', 'towers', 'fc' to B3 operands, a '*_grad' group to a fresh scale at every link - for the
rest of the run, with a warning.

Reference call sites: the Bottleneck of HBB_TOD/mmdet/models/backbones/resnet.py:262-303 (`caffe` style: stride on conv1,
:153-158), FPN laterals / output convolutions necks/fpn.py:151-202, PSAGG necks/ps_fpn.py:56-75, the towers
dense_heads/anchor_free_head.py:198-219, the MIL FC stacks dense_heads/fcos_head_p2b_ts.py:1202-1256.

Gradient convention: the gradient of a plane tensor is a plane tensor of the same shape and format that is ALREADY multiplied by the
ReLU mask of the activation it belongs to (its consumer holds that activation as its input and applies the mask in its epilogue)."""
import os
import warnings

import torch

from . import functional as F
from . import hip

f32 = torch.float32
bf16 = torch.bfloat16
f16 = torch.float16

FMT_F32, FMT_B1, FMT_H2, FMT_B3 = 0, 1, 2, 3          # PT_FMT_* of include/pt_hip.h
SCALE_ONE, SCALE_AUTO, SCALE_MERGE = 0, 1, 2          # PT_SCALE_*
H2_TAIL = 8                                           # elements behind the zero row of an H2 plane (the first two = the fp32 1 / scale)


# ----------------------------------------------------------------------------------------------- range census / policy --
class Census:
    """Device int32 [MAX, 4] (one row per site: saturated count, bits of the largest stored magnitude, and - mode 2 - non-zero
    elements below 0.125, elements written) + the fall-back policy it drives.  `ptr(group, name)` is what a launch receives."""
    MAX = 1024
    LOW = 0.25                                         # a non-zero tensor whose largest STORED magnitude is below this has lost bits: its
                                                       # absolute error 2^-25 is more than 2^-23 of the tensor's own maximum

    def __init__(self):
        self.buf = None
        self.index = {}                                # (group, name) -> row
        self.names = []
        self.demoted = {}                              # group -> reason
        self.mode = int(os.environ.get('PT_F16_CENSUS', '1'))         # 0: off, 1: saturation + maximum, 2: + tiny / total counts
        self.auto_fallback = os.environ.get('PT_F16_FALLBACK', '1') != '0'
        self._pending = None                           # (pinned host copy, event) of an asynchronous poll
        self._pin = None
        self.last = {}                                 # site -> dict of the last completed poll
        self.history = {}                              # site -> [saturated total, largest stored magnitude, smallest non-zero per-poll maximum]

    def _ensure(self, device):
        if self.buf is None or self.buf.device != device:
            self.buf = torch.zeros((self.MAX, 4), dtype=torch.int32, device=device)
            self._pending = None

    def ptr(self, group, name, device):
        if not self.mode:
            return 0
        self._ensure(device)
        k = (group, name)
        i = self.index.get(k)
        if i is None:
            if len(self.names) >= self.MAX:
                return 0
            i = self.index[k] = len(self.names)
            self.names.append(k)
        return self.buf.data_ptr() + 16 * i

    def ok(self, group):
        return group not in self.demoted

    def demote(self, group, reason):
        if group not in self.demoted:
            self.demoted[group] = reason
            warnings.warn(f'fp16 operand group {group!r} falls back ({reason}): '
                          + ('every link of its gradient chains re-scales' if group.endswith('_grad') else 'bf16 x 3 operands from now on'),
                          RuntimeWarning, stacklevel=3)

    def reset(self):
        """Forget demotions and counts (tests; a new model)."""
        self.demoted.clear()
        self.last.clear()
        self.history.clear()
        self._pending = None
        if self.buf is not None:
            self.buf.zero_()

    def _digest(self, host):
        n = len(self.names)
        rows = host[:n].tolist()
        amax = host[:n, 1].contiguous().view(torch.float32).tolist()
        out = {}
        for (group, name), r, m in zip(self.names, rows, amax):
            if r[0] == 0 and r[1] == 0 and r[3] == 0:
                continue
            out[(group, name)] = dict(saturated=r[0], amax_stored=m, tiny=r[2], total=r[3])
            h = self.history.setdefault((group, name), [0, 0.0, float('inf')])
            h[0] += r[0]
            h[1] = max(h[1], m)
            if m > 0:
                h[2] = min(h[2], m)
            if self.auto_fallback:
                if r[0] > 0:
                    self.demote(group, f'{r[0]} elements of {name} beyond fp16 range')
                elif 0.0 < m < self.LOW:
                    self.demote(group, f'largest stored magnitude of {name} is {m:.3g} < {self.LOW}')
        self.last = out
        return out

    def poll(self, sync=False):
        """Read the counters and clear them.  sync=False (the Trainer, every `interval` steps): start an asynchronous copy into pinned
        memory and digest the PREVIOUS one if it has landed - no host wait; sync=True (tests, debugging): wait and return the digest."""
        if self.buf is None or not self.names:
            return {}
        if sync:
            self._pending = None
            host = self.buf.cpu()
            self.buf.zero_()
            return self._digest(host)
        if self._pending is not None:
            host, ev = self._pending
            if not ev.query():
                return None
            self._pending = None
            self._digest(host)
        if self._pin is None:
            self._pin = torch.empty(self.buf.shape, dtype=torch.int32, pin_memory=True)
        host = self._pin                               # (the previous copy has been digested above)
        host.copy_(self.buf, non_blocking=True)
        self.buf.zero_()
        ev = torch.cuda.Event()
        ev.record()
        self._pending = (host, ev)
        return self.last

    def log_vars(self):
        """Flat dict for `log_vars` under PT_F16_CENSUS=2 (the debugging census): per site saturated / tiny fractions and the maximum."""
        out = {}
        for (group, name), d in self.last.items():
            k = f'census/{group}/{name}'
            out[k + '/sat'] = float(d['saturated'])
            out[k + '/amax'] = float(d['amax_stored'])
            if d['total']:
                out[k + '/tiny_frac'] = d['tiny'] / d['total']
        return out


CENSUS = Census()


def use_f16(group):
    """fp16 x 2 operands for this group of layers?  The static switches (functional.F16_*; PT_F16_FC=0 turns all of them off: the
    strict 24-bit path) and the census' verdict so far."""
    flag = {'trunk': F.F16_TRUNK, 'neck': F.F16_NECK, 'towers': F.F16_TOWERS, 'fc': F.F16_FC}[group]
    return bool(flag) and CENSUS.ok(group)


def chain_ok(group):
    """One gradient scale per chain for this group (else: every link re-scales - the fall-back when a chain left fp16's range)."""
    return F.F16_CHAIN and CENSUS.ok(group + '_grad')


# ----------------------------------------------------------------------------------------------------------- formats --
def is_h2(t):
    return t.dtype == f16


def fmt_of(t):
    if t.dtype == f32:
        return FMT_F32
    if t.dtype == f16:
        return FMT_H2
    return FMT_B3 if t.shape[0] == 3 else FMT_B1


def n_of(t):
    """Elements per plane (zero row included, tail excluded)."""
    return t.shape[1] - H2_TAIL if t.dtype == f16 else t.shape[1]


def tail_ptr(t):
    """Device address of the fp32 1 / scale of an H2 tensor (0 for other formats)."""
    return t.data_ptr() + 2 * (t.shape[1] - H2_TAIL) if t.dtype == f16 else 0


def tail_value(t):
    """1 / scale of an H2 tensor as a python float (synchronises: tests only)."""
    n = t.shape[1] - H2_TAIL
    return float(t[0, n:n + 2].view(f32)[0])


def h2_new(rows, C, device, zero=False):
    n = (rows + 1) * C
    return (torch.zeros if zero else torch.empty)((2, n + H2_TAIL), dtype=f16, device=device)


def _new_planes(rows, C, device, zero=False, np=3):
    n = (rows + 1) * C
    return (torch.zeros if zero else torch.empty)((np, n), dtype=bf16, device=device)


def _p16(t):
    """Device pointer of a 16-bit plane tensor (None -> 0)."""
    if t is None:
        return 0
    assert t.dtype in (bf16, f16) and t.is_cuda and t.is_contiguous(), (t.dtype, t.shape)
    return t.data_ptr()


def planes_to_f32(t, rows, C):
    """The represented fp32 values [rows, C] of a plane tensor of any format (tests, debugging; exact)."""
    n = rows * C
    if t.dtype == f16:
        v = (t[0, :n].float() + t[1, :n].float()) * tail_value(t)
    elif t.shape[0] == 3:
        v = (t[0, :n].float() + t[1, :n].float()) + t[2, :n].float()
    else:
        v = t[0, :n].float()
    return v.view(rows, C)


class PlaneAct:
    """An NHWC activation [B, H, W, C] as row-major planes `t` (H2, B3 or B1); `relu`: its values went through a ReLU (so
    `plane 0 > 0` is the mask its gradient needs)."""
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
        """Planes per value: 3 (B3), 2 (H2) or 1 (a bf16 NHWC tensor with a zero row behind it: the bf16 trunk)."""
        return self.t.shape[0]

    def tensor(self):
        """The activation as a [B, C, H, W] channels_last tensor, differentiable: fp32 (one pass) or bf16 (np = 1: a VIEW of the
        plane - the backward copies the gradient behind a zero row)."""
        return _PlaneView.apply(self.t, self) if self.np == 1 else self.float()

    @property
    def shape(self):
        return (self.B, self.C, self.H, self.W)

    @property
    def f16(self):
        return self.t.dtype == f16

    def float(self):
        """-> fp32 [B, C, H, W] channels_last tensor (exact), differentiable."""
        assert not self.gcarrier, 'a carrier-gradient activation feeds plane kernels only'
        return _PlanesToF32.apply(self.t, self)


# ------------------------------------------------------------------------------------------------------ mix / split --
def mix(a, b=None, c=None, mask=None, relu_of=None, n=None, n_valid=None, out_fmt=None, scale=SCALE_ONE, want_f32=False, census=0):
    """pt_planes_mix: out = fmt(so * m * (a + b + c)).  a, b: plane tensors (any format) or fp32 tensors; -> (planes or None, fp32 or
    None).  n: elements per plane of the result (default: a's); n_valid: elements that read the sources (the rest zeros)."""
    fa = fmt_of(a)
    n = n if n is not None else n_of(a)
    n_valid = n if n_valid is None else n_valid
    dev = a.device
    out = None
    if out_fmt is not None:
        if out_fmt == FMT_H2:
            out = torch.empty((2, n + H2_TAIL), dtype=f16, device=dev)
        else:
            out = torch.empty((3 if out_fmt == FMT_B3 else 1, n), dtype=bf16, device=dev)
    of = torch.empty((n,), dtype=f32, device=dev) if want_f32 else None
    ws = torch.empty((1024,), dtype=f32, device=dev) if (out_fmt == FMT_H2 and scale == SCALE_AUTO) else None
    hip.call('pt_planes_mix', a.data_ptr(), fa, a.shape[1] if fa != FMT_F32 else 0, tail_ptr(a),
             b.data_ptr() if b is not None else 0, fmt_of(b) if b is not None else 0,
             (b.shape[1] if fmt_of(b) != FMT_F32 else 0) if b is not None else 0, tail_ptr(b) if b is not None else 0,
             c, _p16(mask), relu_of, n, n_valid, out.data_ptr() if out is not None else 0, out_fmt if out_fmt is not None else 0,
             out.shape[1] if out is not None else 0, scale, of, ws, census)
    return out, of


def combine(a, b=None, c=None, mask=None, n=None, want_planes=True, want_f32=False):
    """split(mask * (a + b + c)) in a's format and / or its fp32 value (two gradient plane sets of one activation; planes -> fp32)."""
    fa = fmt_of(a)
    n = n if n is not None else n_of(a)
    assert not want_planes or n == n_of(a), 'planes out: combine whole planes (the zero rows included)'
    return mix(a, b, c, mask=mask, n=n, out_fmt=fa if want_planes else None, scale=SCALE_MERGE, want_f32=want_f32)


def split_nhwc(x, stride=1, np=None, f16=False, census=0):
    """channels_last [B, C, H, W] (fp32, or bf16 -> np = 1) -> row-major planes of the pixels (y * stride, x * stride).
    f16: H2 (pt_split_gather_h2); else np = 3 (B3) or 1 (B1) (pt_split_bf16x3_gather)."""
    B, C, H, W = x.shape
    rows = x.permute(0, 2, 3, 1)
    is16 = x.dtype == bf16
    np = (1 if is16 else 3) if np is None else np
    assert rows.is_contiguous() and x.dtype in (f32, bf16) and (np == 1 or not is16)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if f16:
        assert not is16
        t = h2_new(B * Ho * Wo, C, x.device)
        hip.call('pt_split_gather_h2', rows, C, B, H, W, C, stride, t.data_ptr(), t.shape[1], census)
        return t, Ho, Wo
    t = _new_planes(B * Ho * Wo, C, x.device, np=np)
    hip.call('pt_split_bf16x3_gather', rows, int(is16), C, B, H, W, C, stride, np, t, t.shape[1])
    return t, Ho, Wo


def grad_to_planes(rows, h2, relu_of=None, mask=None, census=0):
    """An fp32 gradient [M, C] entering the plane region -> planes with a zero row: H2 with the power-of-two scale that brings its
    largest magnitude into [128, 256) (two launches: maximum, conversion), or B3.  relu_of (fp32 [M, C]) / mask (plane 0 of the
    activation's planes): the ReLU of the tensor the gradient belongs to."""
    M, C = rows.shape
    if h2:
        return mix(rows, mask=mask, relu_of=relu_of, n=(M + 1) * C, n_valid=M * C, out_fmt=FMT_H2, scale=SCALE_AUTO, census=census)[0]
    if mask is None:
        return F._split_rows(rows, relu_of=relu_of)
    return mix(rows, mask=mask, relu_of=relu_of, n=(M + 1) * C, n_valid=M * C, out_fmt=FMT_B3)[0]


def rescale_h2(t, mask=None, census=0):
    """A fresh power-of-two scale for an H2 (or B3) tensor -> H2 (the per-link fall-back of a gradient chain)."""
    return mix(t, mask=mask, out_fmt=FMT_H2, scale=SCALE_AUTO, census=census)[0]


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


class _PlanesToF32(torch.autograd.Function):
    """planes -> fp32 [B, C, H, W] channels_last; backward: the (masked) fp32 gradient back into the planes' format."""

    @staticmethod
    def forward(ctx, t, act):
        _, of = mix(t, n=act.P * act.C, want_f32=True)
        ctx.act = act
        ctx.h2 = is_h2(t)
        ctx.save_for_backward(t if act.relu else None)
        return of.view(act.B, act.H, act.W, act.C).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        act = ctx.act
        t, = ctx.saved_tensors
        rows = g.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(act.P, act.C)
        return grad_to_planes(rows, ctx.h2, mask=t if act.relu else None, census=CENSUS.ptr('trunk_grad', 'planes_to_f32', g.device) if ctx.h2 else 0), None


class _F32ToPlanes(torch.autograd.Function):
    """fp32 channels_last [B, C, H, W] -> planes (B3, or H2); backward: the gradient planes summed back to fp32."""

    @staticmethod
    def forward(ctx, x, h2):
        t, _, _ = split_nhwc(x, f16=h2, census=CENSUS.ptr('trunk', 'to_planes', x.device) if h2 else 0)
        ctx.shape = x.shape
        return t

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        _, of = mix(g.contiguous(), n=B * H * W * C, want_f32=True)
        return of.view(B, H, W, C).permute(0, 3, 1, 2), None


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
        _, of = mix(g1.contiguous(), g2.contiguous() if g2 is not None else None, n=B * H * W * C, want_f32=True)
        return of.view(B, H, W, C).permute(0, 3, 1, 2)


class _F32ToF16Planes(torch.autograd.Function):
    """fp32 channels_last [B, C, H, W] -> H2 planes for ONE or TWO consumers whose input gradients come back as fp32 CARRIERS (tensors
    of the planes' shape holding the fp32 gradient in their head: planes._PlaneConv with x_gcarrier)."""

    @staticmethod
    def forward(ctx, x, two, group):
        t, _, _ = split_nhwc(x, f16=True, census=CENSUS.ptr(group, 'input', x.device))
        ctx.shape = x.shape
        return (t.view_as(t), t.view_as(t)) if two else t.view_as(t)

    @staticmethod
    def backward(ctx, g1, g2=None):
        B, C, H, W = ctx.shape
        n = B * H * W * C
        gs = [g.contiguous().view(-1).view(f32)[:n] for g in (g1, g2) if g is not None]
        g = gs[0] if len(gs) == 1 else gs[0] + gs[1]
        return g.view(B, H, W, C).permute(0, 3, 1, 2), None, None


def to_planes2(x, f16=False, group='towers'):
    """x as planes for two consumers -> (PlaneAct, PlaneAct) sharing one set of planes.  f16: H2 planes - the consumers run on three
    fp16 MFMA products and return their input gradients as fp32 carriers."""
    B, C, H, W = x.shape
    if f16:
        a, b = _F32ToF16Planes.apply(x, True, group)
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


def to_planes(x, f16=False, group='towers', carrier=True):
    """Differentiable entry into plane mode (relu=False: the producer of x masks its own gradient): fp32 -> B3, bf16 -> B1;
    f16: fp32 -> H2, whose gradient returns as an fp32 carrier (`carrier`, a single consumer: see to_planes2) or as H2 planes."""
    B, C, H, W = x.shape
    if f16 and x.dtype == f32 and carrier:
        return PlaneAct(_F32ToF16Planes.apply(x, False, group), B, H, W, C, False, gcarrier=True)
    if x.dtype == bf16:
        return PlaneAct(_Bf16ToPlane.apply(x), B, H, W, C, False)
    return PlaneAct(_F32ToPlanes.apply(x, bool(f16)), B, H, W, C, False)


def first_images(act, B):
    """The first B images of a plane activation as a PlaneAct of their own (one copy pass: rows, a zero row, the tail) - no gradient
    (the frozen stem shared between the student's batched pass and the teacher)."""
    assert B <= act.B and not act.t.requires_grad
    if B == act.B:
        return act
    rows = B * act.H * act.W
    t, _ = mix(act.t, n=(rows + 1) * act.C, n_valid=rows * act.C, out_fmt=fmt_of(act.t), scale=SCALE_MERGE)
    return PlaneAct(t, B, act.H, act.W, act.C, act.relu)


def out_hw(H, W, K, stride, pad):
    return (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1


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
                transposed_out=None, alpha=None, x_inv=0, out_f16=None, chain=False, res_inv=0, census=0):
    """pt_conv_bf16x6: x_t row-major planes of [B*Hs*Ws (+1), Cin]; wp SplitPlanes of the weight (the same operand format).
    -> (planes or None, fp32 rows or None).  scatter = (H, W): a stride-2 input gradient placed at (2y, 2x) of a zeroed [B, H, W] grid.
    H2 operands: alpha defaults to 1 / PT_F16_WEIGHT_SCALE; x_inv (a device address, see tail_ptr) multiplies the operand's 1 / scale
    in - or, `chain`, it is left out and the output (H2) carries the operand's tail on.  out_f16: the result planes as H2 (default:
    the operands' format)."""
    Ho, Wo = out_hw(Hs, Ws, K, stride, pad) if transposed_out is None else transposed_out
    M = B * Ho * Wo
    rows = M if scatter is None else B * scatter[0] * scatter[1]
    dev = x_t.device
    h2 = x_t.dtype == f16                                     # fp16 x 2 operands, three products
    np_ = 3 if h2 else x_t.shape[0]
    if out_f16 is None:
        out_f16 = h2
    out_p = None
    if want_planes and out_f16:
        assert np_ == 3
        out_p = h2_new(rows, Cout, dev, zero=scatter is not None)
    elif want_planes:
        out_p = _new_planes(rows, Cout, dev, zero=scatter is not None, np=np_)
    out_f = (torch.zeros if scatter is not None else torch.empty)((rows, Cout), dtype=f32, device=dev) if want_f32 else None
    if f32_out is not None:                                   # a caller-provided fp32 buffer of rows * Cout elements
        assert f32_out.numel() == rows * Cout and f32_out.dtype == f32 and scatter is None
        out_f = f32_out
    d = hip.STRUCTS['pt_conv_desc']()
    d.B, d.Hs, d.Ws, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad, d.relu = B, Hs, Ws, Cin, Cout, K, K, stride, pad, int(bool(relu))
    d.np = np_
    d.operand_f16 = int(h2)
    if h2:
        d.alpha = float(alpha) if alpha is not None else 1.0 / F.F16_WEIGHT_SCALE
        if chain:
            assert out_p is None or out_p.dtype == f16
            d.out_inv_scale_src = tail_ptr(x_t)
        elif x_inv:
            d.alpha_dev = x_inv
    else:
        assert not chain and not x_inv
        d.alpha = float(alpha) if alpha is not None else 0.0
    assert h2 == (wp.planes.shape[0] == 2), 'fp16 activations go with fp16 weight planes'
    d.x_planes, d.x_plane_stride = _p16(x_t), x_t.shape[1]
    d.w_planes, d.w_plane_stride = hip.dptr(wp.planes, 'uint16_t'), wp.planes.shape[1]
    d.scale, d.shift = hip.dptr(scale, 'float'), hip.dptr(shift, 'float')
    if res_planes is not None:
        d.res_planes, d.res_plane_stride = _p16(res_planes), res_planes.shape[1]
        if res_planes.dtype == f16:
            d.res_f16 = 1
            d.res_alpha_dev = res_inv
        else:
            assert res_planes.shape[0] == np_
    d.res_f32 = hip.dptr(res_f32, 'float')
    d.mask_planes = _p16(mask_planes)
    d.out_f32 = hip.dptr(out_f, 'float')
    if out_p is not None:
        d.out_planes, d.out_plane_stride = _p16(out_p), out_p.shape[1]
        d.out_f16 = int(out_p.dtype == f16)
        if d.out_f16 and census:
            d.census, d.census_mode = census, CENSUS.mode
    if scatter is not None:
        d.scatter_stride, d.scatter_H, d.scatter_W = 2, scatter[0], scatter[1]
    d.tile_rows = int(tile_rows)
    if transposed_out is not None:            # the input gradient of a 3x3 stride-2 convolution onto its input grid (Ho, Wo)
        d.dstride, d.out_H, d.out_W = 2, Ho, Wo
        splits = 1
    if splits is None:                          # the library's choice for this descriptor (a pure function of the key: asked once)
        k = ('plan', B, Hs, Ws, Cin, Cout, K, stride, pad, int(tile_rows), np_, h2)
        S = _SPLITS.get(k)
        if S is None:
            S = _SPLITS[k] = hip.call('pt_conv_bf16x6_plan', d)
    else:
        S = int(splits)
    if S > 1:
        ws = torch.empty((S * M * Cout,), dtype=f32, device=dev)
        d.workspace, d.workspace_elems, d.splits = hip.dptr(ws, 'float'), ws.numel(), S
    else:
        d.splits = 1
    assert wp.rows == Cout and wp.k == K * K * Cin, (wp.rows, wp.k, Cout, K, Cin)
    assert n_of(x_t) >= (B * Hs * Ws + 1) * Cin
    if mask_planes is not None:
        assert n_of(mask_planes) >= rows * Cout
    if res_planes is not None:
        assert n_of(res_planes) >= M * Cout
    if res_f32 is not None:
        assert res_f32.numel() >= M * Cout
    hip.call('pt_conv_bf16x6', d)
    return out_p, out_f


def wgrad_ok(Cin, Cout):
    return Cin % 128 == 0 and Cout % 128 == 0


def launch_wgrad(gy_t, x_t, B, Hs, Ws, Cin, Cout, K, stride, pad, row_scale=None, want_bias=False, bn=None, w=None, alpha=None,
                 x_scaled=False):
    """pt_conv_wgrad_bf16x6 -> (dw as a channels_last [Cout, Cin, K, K] tensor, dbias or None).
    bn = (scale, rstd, mean) of a TRAINABLE eval-mode BatchNorm behind the convolution (w = its weight): -> (dw, dbeta, dgamma)
    through pt_bn_wgrad_finish.  H2 operands: the gradient's tail (1 / its scale) multiplies the result; x_scaled: the
    activations' tail too (forward activations are stored unscaled)."""
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
    h2 = gy_t.dtype == f16
    d.np = 3 if h2 else gy_t.shape[0]
    d.operand_f16 = int(h2)
    d.alpha = float(alpha) if alpha is not None else (1.0 if h2 else 0.0)
    if h2:
        d.alpha_dev = tail_ptr(gy_t)
        if x_scaled:
            d.alpha_dev2 = tail_ptr(x_t)
    assert x_t.shape[0] == gy_t.shape[0] and x_t.dtype == gy_t.dtype, 'the weight gradient multiplies two operands of ONE format'
    d.gy_planes, d.gy_plane_stride = _p16(gy_t), gy_t.shape[1]
    d.x_planes, d.x_plane_stride = _p16(x_t), x_t.shape[1]
    d.dw, d.dbias, d.row_scale = hip.dptr(dw, 'float'), hip.dptr(db, 'float'), hip.dptr(row_scale, 'float')
    d.workspace, d.workspace_elems, d.splits = hip.dptr(ws, 'float'), ws.numel(), S
    assert n_of(gy_t) >= (B * Ho * Wo + 1) * Cout and n_of(x_t) >= (B * Hs * Ws + 1) * Cin
    hip.call('pt_conv_wgrad_bf16x6', d)
    if bn is not None:
        scale, rstd, mean = bn
        dg = torch.empty((Cout,), dtype=f32, device=dev)
        wr = w.detach().permute(0, 2, 3, 1)
        assert wr.is_contiguous()
        hip.call('pt_bn_wgrad_finish', dw, wr, Cout, K * K * Cin, scale, rstd, mean, db, dg)
        return dw.permute(0, 3, 1, 2), db, dg
    return dw.permute(0, 3, 1, 2), db


# ------------------------------------------------------------------------------------------------ single convolution --
class ConvCfg:
    """Static description of one plane convolution call (not a tensor: autograd passes it through)."""
    __slots__ = ('B', 'H', 'W', 'Cin', 'Cout', 'K', 'stride', 'pad', 'relu', 'x_planes', 'x_relu', 'out_planes', 'scale', 'bias_grad', 'x_gcarrier',
                 'x_f16', 'group', 'name')

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


class _PlaneConv(torch.autograd.Function):
    """y = [relu](conv(x, w) * scale + shift) for ONE consumer of x (stride 1 in the backward's input gradient).
    x: plane tensor (cfg.x_planes) or fp32 / bf16 channels_last [B, Cin, H, W]; y: plane tensor (cfg.out_planes, in the operands'
    format) or a channels_last tensor."""

    @staticmethod
    def forward(ctx, x, w, shift, cfg):
        c = cfg
        dev = x.device
        if c.x_planes:
            xt = x
        else:
            # fp32 -> H2 (x_f16) or B3, bf16 -> B1
            xt, _, _ = split_nhwc(x, f16=bool(c.x_f16), census=CENSUS.ptr(c.group, c.name + ':x', dev) if c.x_f16 else 0)
        np_ = xt.shape[0]
        h2 = is_h2(xt)                                             # fp16 x 2 operands: three products, weights stored x 16
        assert not h2 or c.scale is None
        wp = F._conv_weight_planes(w, False, None, np_)
        as_planes = c.out_planes or np_ == 1                       # (the one-plane result IS the bf16 tensor)
        yp, yf = launch_conv(xt, c.B, c.H, c.W, c.Cin, wp, c.Cout, c.K, c.stride, c.pad, scale=c.scale, shift=shift, relu=c.relu,
                             want_planes=as_planes, want_f32=not as_planes,
                             census=CENSUS.ptr(c.group, c.name + ':y', dev) if (h2 and as_planes) else 0)
        Ho, Wo = out_hw(c.H, c.W, c.K, c.stride, c.pad)
        ctx.cfg = c
        ctx.np = np_
        ctx.h2 = h2
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
        h2 = ctx.h2
        dev = g.device
        gg = c.group + '_grad'
        if c.out_planes:
            E = g.contiguous()                                     # (H2: the scale its chain chose sits in its tail)
        elif np_ == 1:                                             # a bf16 gradient tensor: behind a zero row, masked by the result's ReLU
            E, _, _ = split_nhwc(g.to(bf16).contiguous(memory_format=torch.channels_last), np=1)
            if yf is not None:
                E, _ = combine(E, mask=yf)
        else:
            # an fp32 gradient enters the plane region: B3, or H2 with a power-of-two scale chosen on the device from its largest
            # magnitude (unscaled, 1e-3 ... 1e-9 would sit in fp16's subnormal range) - the scale of the whole chain behind it
            rows = g.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(M, c.Cout)
            E = grad_to_planes(rows, h2, relu_of=yf, census=CENSUS.ptr(gg, c.name + ':gy', dev) if h2 else 0)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            assert c.stride == 1, 'the single-consumer plane convolution back-propagates stride 1 only'
            wd = F._conv_weight_planes(w, True, c.scale, np_)
            inv = tail_ptr(E)
            pad_t = c.K - 1 - c.pad
            if c.x_planes and c.x_gcarrier:
                # the producer of x (RoIAlign, the split at a chain's head) wants its gradient as fp32: it travels in the head of a
                # tensor of x's shape
                gx = torch.empty_like(xt)
                launch_conv(E, c.B, Ho, Wo, c.Cout, wd, c.Cin, c.K, 1, pad_t,
                            f32_out=gx.view(-1).view(f32)[:c.B * c.H * c.W * c.Cin], x_inv=inv)
            elif not c.x_planes and np_ != 1:                      # an fp32 input split here: its gradient leaves as fp32
                _, gf = launch_conv(E, c.B, Ho, Wo, c.Cout, wd, c.Cin, c.K, 1, pad_t, want_f32=True, x_inv=inv)
                gx = gf.view(c.B, c.H, c.W, c.Cin).permute(0, 3, 1, 2)
            elif h2:
                # the producer of x is another H2 layer: its output gradient as H2 planes masked by x's ReLU, in the chain's scale
                mask = xt if c.x_relu else None
                if chain_ok(c.group):
                    gx, _ = launch_conv(E, c.B, Ho, Wo, c.Cout, wd, c.Cin, c.K, 1, pad_t, mask_planes=mask, want_planes=True, chain=True,
                                        census=CENSUS.ptr(gg, c.name + ':gx', dev))
                else:                                              # fall-back: the natural value as B3, then a fresh scale
                    gp, _ = launch_conv(E, c.B, Ho, Wo, c.Cout, wd, c.Cin, c.K, 1, pad_t, mask_planes=mask, want_planes=True, out_f16=False,
                                        x_inv=inv)
                    gx = rescale_h2(gp, census=CENSUS.ptr(gg, c.name + ':gx', dev))
            else:
                gp, gf = launch_conv(E, c.B, Ho, Wo, c.Cout, wd, c.Cin, c.K, 1, pad_t,
                                     mask_planes=xt if (c.x_planes and c.x_relu) else None, want_planes=True)
                if c.x_planes:
                    gx = gp
                else:                                              # np_ == 1: the bf16 tensor
                    gx = gp[0, :c.B * c.H * c.W * c.Cin].view(c.B, c.H, c.W, c.Cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw, gb = launch_wgrad(E, xt, c.B, c.H, c.W, c.Cin, c.Cout, c.K, c.stride, c.pad, row_scale=c.scale,
                                  want_bias=bool(c.bias_grad and ctx.needs_input_grad[2]))
            if w.dim() == 2:
                gw = gw.reshape(w.shape)
        elif c.bias_grad and ctx.needs_input_grad[2]:              # a trainable bias behind a frozen weight (round-4 advice)
            _, of = mix(E, n=M * c.Cout, want_f32=True)
            gb = of.view(M, c.Cout).sum(0)
        return gx, gw, gb, None


def plane_conv_ok(conv, x_channels_last_f32_or_planes=True):
    """A convolution pt_conv_bf16x6 / pt_conv_wgrad_bf16x6 take: 1x1 (pad 0) or 3x3 (pad 1), stride 1, one group, no dilation,
    channels multiples of 128 (the weight gradient's tile), channels_last fp32 weight."""
    k = conv.kernel_size
    return (type(conv) is torch.nn.Conv2d and k in ((1, 1), (3, 3)) and conv.stride == (1, 1) and conv.padding == ((k[0] - 1) // 2,) * 2
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels % 128 == 0 and conv.out_channels % 128 == 0
            and conv.weight.is_cuda and conv.weight.dtype == f32 and F._ConvWeightPlanes.ok(conv.weight)
            and autocast_ok())


def _site(conv_or_fc, tag=''):
    """A short, stable name of a layer for the census (shape-based: layers of one shape share a row)."""
    w = conv_or_fc.weight
    k = w.shape[2] if w.dim() == 4 else 1
    return f'{tag}{w.shape[1]}>{w.shape[0]}k{k}'


def conv_module(x, conv, relu=False, out_planes=False, scale=None, shift=None, f16=False, group='neck'):
    """[relu](conv(x) (* scale) + bias) through the plane kernels.  x: PlaneAct or fp32 channels_last tensor.
    -> PlaneAct (out_planes) or fp32 channels_last tensor.  f16 (an fp32 tensor in): the operands as H2 planes, three MFMA products;
    a PlaneAct input decides the operand format itself."""
    is_p = isinstance(x, PlaneAct)
    B, Cin, H, W = x.shape
    k = conv.kernel_size[0]
    cfg = ConvCfg(B=B, H=H, W=W, Cin=Cin, Cout=conv.out_channels, K=k, stride=1, pad=conv.padding[0], relu=bool(relu), x_planes=is_p,
                  x_relu=bool(is_p and x.relu), out_planes=bool(out_planes), scale=scale, bias_grad=scale is None and conv.bias is not None,
                  x_gcarrier=bool(is_p and x.gcarrier),
                  x_f16=bool(f16 and not is_p and scale is None and x.dtype == f32), group=group, name=_site(conv))
    sh = shift if shift is not None else conv.bias
    y = _PlaneConv.apply(x.t if is_p else x, conv.weight, sh, cfg)
    if out_planes:
        return PlaneAct(y, B, H, W, conv.out_channels, bool(relu))
    return y                                                      # fp32 (split operands) or bf16 (one plane) [B, Cout, H, W]


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
                  # into H2 planes here, three MFMA products
                  x_f16=bool(use_f16('fc') and not is_p and fc.in_features >= 4096 and x.dtype == f32), group='fc', name=_site(fc))
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
    """fp32 outside autocast (split operands), bf16 inside a bf16 autocast region (one plane)."""
    if torch.is_autocast_enabled():
        return torch.get_autocast_dtype('cuda') == bf16 and bf16_ok(x)
    return f32_ok(x)


def autocast_ok():
    """No autocast, or bf16 autocast (one-plane operands)."""
    return (not torch.is_autocast_enabled()) or torch.get_autocast_dtype('cuda') == bf16


# ------------------------------------------------------------------------------------------------------- fan-out --
class _FanOut(torch.autograd.Function):
    """One plane activation, two consumers: the backward adds the two (already masked) gradient plane sets exactly (fp32 sum of the
    planes, split again; two H2 chains with different scales continue with the smaller one) instead of autograd's 16-bit addition."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t), t.view_as(t)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None or g2 is None:
            return g1 if g2 is None else g2
        g1, g2 = g1.contiguous(), g2.contiguous()
        out, _ = mix(g1, g2, out_fmt=fmt_of(g1), scale=SCALE_MERGE,
                     census=CENSUS.ptr('trunk_grad', 'fan_out', g1.device) if is_h2(g1) else 0)          # (zero rows add up to the zero row)
        return out


def fan_out(act):
    a, b = _FanOut.apply(act.t)
    return (PlaneAct(a, act.B, act.H, act.W, act.C, act.relu), PlaneAct(b, act.B, act.H, act.W, act.C, act.relu))


# ---------------------------------------------------------------------------------------------------- Bottleneck --
class BottleneckCfg:
    """B, H, W, Cin, planes: the block's input and width; s1 / s2: strides of conv1 / conv2 (`caffe`: (s, 1), `pytorch`: (1, s));
    bn[i] for i in ('1', '2', '3', 'd'): (scale, shift, rstd, mean) of the eval-mode BatchNorm behind conv i; bn_train: their
    affines take gradients; name: the block's name for the census."""
    __slots__ = ('B', 'H', 'W', 'Cin', 'planes', 's1', 's2', 'x_planes', 'x_relu', 'bn', 'has_ds', 'bn_train', 'name')

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
    bias sums of the same launch (pt_bn_wgrad_finish): no pass over an activation for either.

    Formats: H2 throughout (group 'trunk': x, y1, y2, out and every gradient; 4 bytes per element, three products) - or B3 planes
    with, optionally (functional.F16_BLOCK3), conv2 alone on H2 operands; B1 under bf16 autocast."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, wd, g1, b1, g2, b2, g3, b3, gd, bd, cfg):
        c = cfg
        p, s1, s2 = c.planes, c.s1, c.s2
        st = s1 * s2
        dev = x.device
        H1, W1 = out_hw(c.H, c.W, 1, st, 0)                       # the block's output grid
        if c.x_planes:
            xs, xH, xW, xst = x, c.H, c.W, 1                      # dense planes of x; strided reads map through them
            trunk16 = is_h2(xs)
        else:
            trunk16 = x.dtype == f32 and use_f16('trunk')
            cen = CENSUS.ptr('trunk', c.name + ':x', dev) if trunk16 else 0
            if s1 > 1:
                xs, xH, xW = split_nhwc(x, s1, f16=trunk16, census=cen)      # `caffe`: only the stride's pixels are ever read
                xst = s1
            else:
                xs, xH, xW = split_nhwc(x, f16=trunk16, census=cen)[0], c.H, c.W
                xst = 1
        np_ = 3 if trunk16 else xs.shape[0]
        r1 = s1 // xst                                            # strides on the planes at hand
        rd = st // xst
        Ha, Wa = out_hw(xH, xW, 1, r1, 0)                         # conv1's output grid

        def cw(w_, dgrad=False, sc=None, np=None):
            return F._conv_weight_planes(w_, dgrad, sc, (2 if trunk16 else np_) if np is None else np)

        def cen(tag):
            return CENSUS.ptr('trunk', f'{c.name}:{tag}', dev)
        S = c.bn
        if trunk16:
            y1, _ = launch_conv(xs, c.B, xH, xW, c.Cin, cw(w1), p, 1, r1, 0, scale=S['1'][0], shift=S['1'][1], relu=True, want_planes=True,
                                census=cen('y1'))
            y2, _ = launch_conv(y1, c.B, Ha, Wa, p, cw(w2), p, 3, s2, 1, scale=S['2'][0], shift=S['2'][1], relu=True, want_planes=True,
                                census=cen('y2'))
            if c.has_ds:
                _, idn = launch_conv(xs, c.B, xH, xW, c.Cin, cw(wd), 4 * p, 1, rd, 0, scale=S['d'][0], shift=S['d'][1], want_f32=True)
                out, _ = launch_conv(y2, c.B, H1, W1, p, cw(w3), 4 * p, 1, 1, 0, scale=S['3'][0], shift=S['3'][1], res_f32=idn, relu=True,
                                     want_planes=True, census=cen('out'))
            else:
                out, _ = launch_conv(y2, c.B, H1, W1, p, cw(w3), 4 * p, 1, 1, 0, scale=S['3'][0], shift=S['3'][1], res_planes=xs, relu=True,
                                     want_planes=True, census=cen('out'))
        else:
            # B3 planes; conv2 (3x3: the block's compute-bound product) optionally on H2 operands / three MFMA products: conv1's
            # epilogue writes y1 as H2 planes, conv2's weights are fp16 planes of 16 w (alpha = 1 / 16 in front of its BatchNorm)
            h16 = bool(F.F16_BLOCK3 and np_ == 3 and CENSUS.ok('trunk'))
            y1, _ = launch_conv(xs, c.B, xH, xW, c.Cin, cw(w1), p, 1, r1, 0, scale=S['1'][0], shift=S['1'][1], relu=True, want_planes=True,
                                out_f16=h16, census=cen('y1') if h16 else 0)
            y2, _ = launch_conv(y1, c.B, Ha, Wa, p, cw(w2, np=2) if h16 else cw(w2), p, 3, s2, 1,
                                scale=S['2'][0], shift=S['2'][1], relu=True, want_planes=True, out_f16=False)
            if c.has_ds and np_ == 1:                                 # bf16 trunk: the identity travels as bf16 too
                idn, _ = launch_conv(xs, c.B, xH, xW, c.Cin, cw(wd), 4 * p, 1, rd, 0, scale=S['d'][0], shift=S['d'][1], want_planes=True)
                out, _ = launch_conv(y2, c.B, H1, W1, p, cw(w3), 4 * p, 1, 1, 0, scale=S['3'][0], shift=S['3'][1], res_planes=idn, relu=True,
                                     want_planes=True)
            elif c.has_ds:
                _, idn = launch_conv(xs, c.B, xH, xW, c.Cin, cw(wd), 4 * p, 1, rd, 0, scale=S['d'][0], shift=S['d'][1], want_f32=True)
                out, _ = launch_conv(y2, c.B, H1, W1, p, cw(w3), 4 * p, 1, 1, 0, scale=S['3'][0], shift=S['3'][1], res_f32=idn, relu=True,
                                     want_planes=True)
            else:
                out, _ = launch_conv(y2, c.B, H1, W1, p, cw(w3), 4 * p, 1, 1, 0, scale=S['3'][0], shift=S['3'][1], res_planes=xs, relu=True,
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
        dev = E.device
        trunk16 = is_h2(E)
        np_ = 3 if trunk16 else E.shape[0]
        assert trunk16 == is_h2(y2)

        def cw(w_, dgrad, sc=None, np=None):
            return F._conv_weight_planes(w_, dgrad, sc, (2 if trunk16 else np_) if np is None else np)

        def cen(tag):
            return CENSUS.ptr('trunk_grad', f'{c.name}:{tag}', dev)
        need_x = ctx.needs_input_grad[0]
        gw = {}                                                   # conv -> (dw, dbeta, dgamma)

        def wgrad(tag, w, gy, xin, Hs_, Ws_, Ci, Co, K, stride, pad, need):
            if not need:
                gw[tag] = (None, None, None)
            elif c.bn_train:
                gw[tag] = launch_wgrad(gy, xin, c.B, Hs_, Ws_, Ci, Co, K, stride, pad, bn=(S[tag][0], S[tag][2], S[tag][3]), w=w)
            else:
                gw[tag] = (launch_wgrad(gy, xin, c.B, Hs_, Ws_, Ci, Co, K, stride, pad, row_scale=S[tag][0])[0], None, None)
        ni = ctx.needs_input_grad
        if trunk16:
            # H2 throughout: E carries its chain's scale s in its tail; every input-gradient launch leaves the operand's 1 / s out of
            # alpha (its result IS s * gradient) and hands the tail on; the weight gradients multiply 1 / s back in.  Fall-back
            # ('trunk_grad' demoted): every link is written as B3 in natural units and re-scaled (two more launches per link).
            chain = chain_ok('trunk')

            def dgrad(tag, gy, Hs_, Ws_, Ci, w_, Co, K, pad, mask, **kw):
                if chain:
                    return launch_conv(gy, c.B, Hs_, Ws_, Ci, w_, Co, K, 1, pad, mask_planes=mask, chain=True, census=cen(tag), **kw)
                want_p = kw.pop('want_planes', False)
                kw.pop('res_inv', None)
                res = kw.get('res_planes')
                gp, gf = launch_conv(gy, c.B, Hs_, Ws_, Ci, w_, Co, K, 1, pad, mask_planes=mask, x_inv=tail_ptr(gy), want_planes=want_p, out_f16=False,
                                     res_inv=tail_ptr(res) if res is not None else 0, **kw)
                return (rescale_h2(gp, census=cen(tag)) if want_p else None), gf
            E2, _ = dgrad('E2', E, H1, W1, 4 * p, cw(w3, True, S['3'][0]), p, 1, 0, y2, want_planes=True)
            wgrad('3', w3, E, y2, H1, W1, p, 4 * p, 1, 1, 0, ni[3])
            E1, _ = dgrad('E1', E2, H1, W1, p, cw(w2, True, S['2'][0]), p, 3, 1, y1, want_planes=True,
                          transposed_out=(Ha, Wa) if s2 != 1 else None)
            wgrad('2', w2, E2, y1, Ha, Wa, p, p, 3, s2, 1, ni[2])
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
                    # E by-passes the block on the identity path: in chain mode the same chain, added raw; in the fall-back E1 was
                    # re-scaled and `dgrad` adds both terms in natural units (res_inv = E's tail)
                    gx, _ = dgrad('gx', E1, Ha, Wa, p, cw(w1, True, S['1'][0]), c.Cin, 1, 0, mask, res_planes=E, want_planes=True)
                else:
                    # the downsample path's gradient as fp32 in the chain's units (chain) or in natural units (fall-back: E and E1
                    # carry different scales), added in conv1's input-gradient epilogue; `caffe`: the sum is scattered to the
                    # stride's pixels of a zeroed map; `pytorch`: the downsample's gradient is scattered, conv1's is dense
                    sc_d = None if r1 == rd else (c.H, c.W)
                    if chain:
                        _, gdx = launch_conv(E, c.B, H1, W1, 4 * p, cw(wd, True, S['d'][0]), c.Cin, 1, 1, 0, want_f32=True, chain=True, scatter=sc_d)
                    else:
                        _, gdx = launch_conv(E, c.B, H1, W1, 4 * p, cw(wd, True, S['d'][0]), c.Cin, 1, 1, 0, want_f32=True, x_inv=tail_ptr(E),
                                             scatter=sc_d)
                    gx, _ = dgrad('gx', E1, Ha, Wa, p, cw(w1, True, S['1'][0]), c.Cin, 1, 0, mask, res_f32=gdx, want_planes=True,
                                  scatter=(c.H, c.W) if (r1 == rd and r1 == 2) else None)
        else:
            # conv3: gradient of y2 (masked by y2's ReLU), weight gradient
            E2, _ = launch_conv(E, c.B, H1, W1, 4 * p, cw(w3, True, S['3'][0]), p, 1, 1, 0, mask_planes=y2, want_planes=True)
            wgrad('3', w3, E, y2, H1, W1, p, 4 * p, 1, 1, 0, ni[3])
            # conv2 (a stride on it makes its input gradient a transposed convolution onto y1's grid).  y1 in H2 planes (forward): the
            # product runs on fp16 x 2 operands - E2 converted with a power-of-two scale chosen on the device (two small launches against
            # half of a 3x3 product's MFMA work), weights x 16; the epilogue's alpha undoes both, E1 leaves as B3 planes
            h16 = is_h2(y1)
            E2c = rescale_h2(E2, census=cen('E2')) if h16 else E2
            wd2 = cw(w2, True, S['2'][0], np=2) if h16 else cw(w2, True, S['2'][0])
            E1, _ = launch_conv(E2c, c.B, H1, W1, p, wd2, p, 3, 1, 1, mask_planes=y1, want_planes=True, out_f16=False,
                                x_inv=tail_ptr(E2c), transposed_out=(Ha, Wa) if s2 != 1 else None)
            wgrad('2', w2, E2c, y1, Ha, Wa, p, p, 3, s2, 1, ni[2])
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
