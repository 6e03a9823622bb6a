"""The fp16 x 2 ("H2") operand path at TRAINED-LIKE dynamic range, and its guard (round-4 verdict W1 / item 2).

fp16 has 5 exponent bits.  The plane kernels store forward activations unscaled, weights x 16, gradients x a power of two chosen on
the device; this file checks against float64 that (i) inputs spread log-uniformly over 1e-4 ... 1e4 and the reference's real
initialisation (dense_heads/fcos_head_p2b_ts.py:137-145: every Conv2d of the head ~ N(0, 0.01); torch's default Linear init for the
FC stacks; Kaiming for the backbone on raw 0 - 255 pixel statistics) stay fp32-class, (ii) the range census sees a tensor that leaves
fp16's range and the group falls back to bf16 x 3 operands, after which the result is right again."""
import warnings

import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _loguniform(shape, g, lo=1e-4, hi=1e4, signed=True):
    import math
    e = torch.rand(shape, generator=g) * (math.log10(hi) - math.log10(lo)) + math.log10(lo)
    v = 10.0 ** e
    if signed:
        v = v * (torch.randint(0, 2, shape, generator=g) * 2 - 1)
    return v.float()


def _cl(x):
    return x.to(DEV).contiguous(memory_format=torch.channels_last)


def _err(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.fixture(autouse=True)
def _fresh_census():
    from point_teacher_amd import planes as PL
    PL.CENSUS.reset()
    yield
    PL.CENSUS.reset()


def test_fc1_with_log_uniform_rows_vs_fp64():
    """The MIL head's first FC layer (fcos_head_p2b_ts.py:1202-1236: 12 544 -> 1 024) + the second one on H2 operands, RoI rows spread
    log-uniformly over 1e-4 ... 1e4 (both signs), torch's default Linear init: value and every gradient no further from float64 than 3 x
    the fp32 library's own error (or 1e-6 of the norm); nothing saturates, no group is demoted."""
    from point_teacher_amd import planes as PL
    g = torch.Generator().manual_seed(1)
    K, Cin = 2304, 12544
    torch.manual_seed(2)
    fc1, fc2 = torch.nn.Linear(Cin, 1024).to(DEV), torch.nn.Linear(1024, 1024).to(DEV)
    x = _loguniform((K, Cin), g).to(DEV).requires_grad_(True)
    gy = (torch.randn(K, 1024, generator=g) * 1e-5).to(DEV)                  # gradient-sized
    xa = PL.PlaneAct(PL._F32ToF16Planes.apply(x.view(1, K, 1, Cin).permute(0, 3, 1, 2), False, 'fc'), 1, K, 1, Cin, False, gcarrier=True)
    assert xa.f16
    y1 = PL.linear(xa, fc1, relu=True, out_planes=True)
    y = PL.linear(y1, fc2, relu=True)
    got = torch.autograd.grad(y, [x, fc1.weight, fc1.bias, fc2.weight, fc2.bias], gy)
    xd = x.detach().double().requires_grad_(True)
    yr = torch.relu(torch.relu(xd @ fc1.weight.double().t() + fc1.bias.double()) @ fc2.weight.double().t() + fc2.bias.double())
    want = torch.autograd.grad(yr, [xd, fc1.weight, fc1.bias, fc2.weight, fc2.bias], gy.double())
    xl = x.detach().clone().requires_grad_(True)
    yl = torch.relu(TF.linear(torch.relu(TF.linear(xl, fc1.weight, fc1.bias)), fc2.weight, fc2.bias))
    lib = torch.autograd.grad(yl, [xl, fc1.weight, fc1.bias, fc2.weight, fc2.bias], gy)
    e, el = _err(y, yr), _err(yl, yr)
    print(f'y: H2 {e:.3e} library fp32 {el:.3e}')
    assert e <= max(3 * el, 1e-6)
    for n, a, b, c in zip(('x', 'w1', 'b1', 'w2', 'b2'), got, want, lib):
        e, el = _err(a, b), _err(c, b)
        print(f'{n}: H2 {e:.3e} library fp32 {el:.3e}')
        assert e <= max(3 * el, 2e-6), (n, e, el)
    cen = PL.CENSUS.poll(sync=True)
    assert cen and all(d['saturated'] == 0 for d in cen.values()) and not PL.CENSUS.demoted, (cen, PL.CENSUS.demoted)


def test_tower_chain_with_the_reference_init_and_log_uniform_map_vs_fp64():
    """Four 3x3 convolutions + bias + ReLU with the reference's head init (every Conv2d ~ N(0, 0.01), zero bias:
    fcos_head_p2b_ts.py:137-145) on a map spread log-uniformly over 1e-4 ... 1e4, a 1e-6-sized output gradient: the chain's output
    and gradients against float64 within 3 x the fp32 library's error; the census stays clean."""
    from point_teacher_amd import planes as PL
    from point_teacher_amd.nn_modules import ConvModule
    g = torch.Generator().manual_seed(3)
    B, C, H, W = 2, 256, 40, 36
    torch.manual_seed(4)
    layers = [ConvModule(C, C, 3, padding=1).to(DEV).to(memory_format=torch.channels_last) for _ in range(4)]
    for l in layers:
        l.plane_group = 'towers'
        torch.nn.init.normal_(l.conv.weight, 0, 0.01)
        torch.nn.init.zeros_(l.conv.bias)
    x = _cl(_loguniform((B, C, H, W), g)).requires_grad_(True)
    gy = _cl(torch.randn(B, C, H, W, generator=g) * 1e-6)
    t = PL.to_planes(x, f16=True)
    for i, l in enumerate(layers):
        t = l(t, out_planes=i + 1 < len(layers))
    params = [p for l in layers for p in (l.conv.weight, l.conv.bias)]
    got = torch.autograd.grad(t, [x] + params, gy)

    def chain(xin, dt):
        v = xin
        for l in layers:
            v = torch.relu(TF.conv2d(v, l.conv.weight.to(dt), l.conv.bias.to(dt), padding=1))
        return v
    xd = x.detach().double().requires_grad_(True)
    rd = chain(xd, torch.float64)
    want = torch.autograd.grad(rd, [xd] + params, gy.double())
    xl = x.detach().clone().requires_grad_(True)
    rl = chain(xl, torch.float32)
    lib = torch.autograd.grad(rl, [xl] + params, gy)
    e, el = _err(t, rd), _err(rl, rd)
    print(f'out: H2 {e:.3e} library {el:.3e}')
    assert e <= max(3 * el, 1e-6)
    for n, a, b, c in zip(['x'] + [f'p{i}' for i in range(len(params))], got, want, lib):
        e, el = _err(a, b), _err(c, b)
        print(f'{n}: H2 {e:.3e} library {el:.3e}')
        # (ReLU decisions within rounding of zero differ between any fp32 evaluation and float64: both errors are ~1e-5 ... 1e-4 here)
        assert e <= max(3 * el, 5e-6), (n, e, el)
    cen = PL.CENSUS.poll(sync=True)
    assert all(d['saturated'] == 0 for d in cen.values()) and not PL.CENSUS.demoted, (cen, PL.CENSUS.demoted)
    grads = {k: d for k, d in cen.items() if k[0] == 'towers_grad'}
    assert grads and all(0.25 <= d['amax_stored'] < 30000 for d in grads.values()), grads     # the chain's scale holds along four links


def test_bottleneck_on_raw_pixel_statistics_with_kaiming_init_vs_fp64():
    """A layer2-style Bottleneck (resnet.py:262-303, Kaiming init :612-628, default frozen BatchNorm) on a stem-like input with raw
    0 - 255 pixel statistics (no 1 / 64 damping: activations in the hundreds): H2 planes throughout, output and gradients against
    float64 within 3 x the fp32 routing's error (PT_PLANE_TRUNK off = library convolutions)."""
    from point_teacher_amd import planes as PL, nn_modules as NM
    import torch.nn as nn
    g = torch.Generator().manual_seed(5)
    torch.manual_seed(6)
    blk = NM.Bottleneck(512, 128, 1, 1, None, 'caffe').to(DEV).to(memory_format=torch.channels_last)
    for m in blk.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        if isinstance(m, nn.BatchNorm2d):
            for p in m.parameters():
                p.requires_grad = False
    blk.eval()
    x = _cl(torch.relu(torch.randn(2, 512, 40, 36, generator=g)) * 180.0).requires_grad_(True)       # post-ReLU, O(100)
    xa = PL.to_planes(x, f16=True, group='trunk', carrier=False)
    xa.relu = True
    out = blk.forward_planes(xa)
    assert out.f16
    y = out.float()
    gy = _cl(torch.randn(y.shape, generator=g) * 1e-4)
    params = [blk.conv1.weight, blk.conv2.weight, blk.conv3.weight]
    got = torch.autograd.grad(y, params + [x], gy)

    def ref(xin, dt):
        def cb(v, conv, bn, relu, res=None):
            o = TF.conv2d(v, conv.weight.to(dt), None, conv.stride, conv.padding)
            sc = (bn.weight.to(dt) * torch.rsqrt(bn.running_var.to(dt) + bn.eps)).view(1, -1, 1, 1)
            o = o * sc + (bn.bias.to(dt).view(1, -1, 1, 1) - bn.running_mean.to(dt).view(1, -1, 1, 1) * sc)
            if res is not None:
                o = o + res
            return torch.relu(o) if relu else o
        o = cb(xin, blk.conv1, blk.bn1, True)
        o = cb(o, blk.conv2, blk.bn2, True)
        return cb(o, blk.conv3, blk.bn3, True, xin)
    xd = x.detach().double().requires_grad_(True)
    rd = ref(xd, torch.float64)
    want = torch.autograd.grad(rd, params + [xd], gy.double())
    xl = x.detach().clone().requires_grad_(True)
    rl = ref(xl, torch.float32)
    lib = torch.autograd.grad(rl, params + [xl], gy)
    e, el = _err(y, rd), _err(rl, rd)
    print(f'out (max {float(rd.abs().max()):.0f}): H2 {e:.3e} library {el:.3e}')
    assert e <= max(3 * el, 1e-6)
    for n, a, b, c in zip(('w1', 'w2', 'w3', 'x'), got, want, lib):
        if n == 'x':
            b, c = b * (x > 0), c * (x > 0)           # the producer's ReLU mask is applied by the block's epilogue (gradient convention)
        e, el = _err(a, b), _err(c, b)
        print(f'{n}: H2 {e:.3e} library {el:.3e}')
        assert e <= max(3 * el, 5e-6), (n, e, el)
    cen = PL.CENSUS.poll(sync=True)
    assert all(d['saturated'] == 0 for d in cen.values()) and not PL.CENSUS.demoted, (cen, PL.CENSUS.demoted)


def test_census_sees_saturation_and_the_group_falls_back():
    """A convolution whose result leaves fp16's range (inputs ~ 1e4 through weights ~ 1): the H2 output saturates at 60 000 - silently
    until round 4 - the census counts it, the poll demotes the group with a warning, and the same call then runs on bf16 x 3 operands
    and matches float64 again.  Under PT_F16_CENSUS=2 the counts reach log_vars."""
    from point_teacher_amd import planes as PL
    g = torch.Generator().manual_seed(7)
    B, C, H, W = 2, 128, 24, 20
    torch.manual_seed(8)
    conv = torch.nn.Conv2d(C, C, 3, padding=1).to(DEV).to(memory_format=torch.channels_last)
    torch.nn.init.normal_(conv.weight, 0, 0.5)
    x = _cl(torch.randn(B, C, H, W, generator=g) * 3e3)
    ref = TF.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
    assert float(ref.abs().max()) > 65504

    def run():
        assert PL.use_f16('neck') == ('neck' not in PL.CENSUS.demoted)
        t = PL.conv_module(x, conv, relu=False, out_planes=True, f16=PL.use_f16('neck'), group='neck')
        return t, PL.planes_to_f32(t.t, B * H * W, C).view(B, H, W, C).permute(0, 3, 1, 2)
    t, y = run()
    assert t.f16 and float(y.abs().max()) <= 60000.0                            # saturated, not inf
    assert _err(y, ref) > 1e-3
    PL.CENSUS.mode = 2
    try:
        t, y = run()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            cen = PL.CENSUS.poll(sync=True)
        key = [k for k in cen if k[0] == 'neck' and k[1].endswith(':y')]
        assert key and cen[key[0]]['saturated'] > 0 and cen[key[0]]['total'] == B * H * W * C, cen
        assert 'neck' in PL.CENSUS.demoted and any('falls back' in str(m.message) for m in w)
        lv = PL.CENSUS.log_vars()
        assert any(k.endswith('/sat') and v > 0 for k, v in lv.items()) and any(k.endswith('/tiny_frac') for k in lv)
    finally:
        PL.CENSUS.mode = 1
    t, y = run()                                                                # bf16 x 3 operands: fp32's range
    assert not t.f16 and _err(y, ref) < 1e-6


def test_a_gradient_chain_that_leaves_the_range_relinks():
    """A gradient that grows along a chain of H2 layers (weights of norm >> 1) outgrows the scale its first link chose:
    the census demotes '<group>_grad', every link then re-scales (two more launches per link) and the gradients match float64."""
    from point_teacher_amd import planes as PL
    from point_teacher_amd.nn_modules import ConvModule
    g = torch.Generator().manual_seed(9)
    B, C, H, W = 2, 128, 20, 18
    torch.manual_seed(10)
    layers = [ConvModule(C, C, 1, padding=0).to(DEV).to(memory_format=torch.channels_last) for _ in range(3)]
    for l in layers:
        l.plane_group = 'towers'
        torch.nn.init.normal_(l.conv.weight, 0, 1.0)                          # ||W|| ~ 11: the backward grows ~8 x per link
        torch.nn.init.zeros_(l.conv.bias)
    x = _cl(torch.rand(B, C, H, W, generator=g) * 4.0).requires_grad_(True)
    gy = _cl(torch.randn(B, C, H, W, generator=g) * 1e-5)

    def run():
        t = PL.to_planes(x, f16=True)
        for i, l in enumerate(layers):
            t = l(t, out_planes=i + 1 < len(layers))
        return torch.autograd.grad(t, [x, layers[0].conv.weight], gy)
    xd = x.detach().double().requires_grad_(True)
    v = xd
    for l in layers:
        v = torch.relu(TF.conv2d(v, l.conv.weight.double(), l.conv.bias.double()))
    want = torch.autograd.grad(v, [xd, layers[0].conv.weight], gy.double())
    run()
    first = PL.CENSUS.poll(sync=True)
    grew = max(d['amax_stored'] for k, d in first.items() if k[0] == 'towers_grad')
    print('largest stored gradient magnitude along the chain:', grew, 'demoted:', dict(PL.CENSUS.demoted))
    assert grew > 1500                                                          # entry in [128, 256): the chain grew > 6-fold
    PL.CENSUS.demote('towers_grad', 'test: chain outgrew its scale')
    assert not PL.chain_ok('towers')
    got = run()
    cen = PL.CENSUS.poll(sync=True)
    assert all(128.0 <= d['amax_stored'] < 256.0 for k, d in cen.items() if k[0] == 'towers_grad' and k[1].endswith(':gx')), cen
    for n, a, b in zip(('x', 'w0'), got, want):
        assert _err(a, b) < 5e-6, (n, _err(a, b))
