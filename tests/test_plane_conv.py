"""Plane-native convolutions (csrc/gemm_split.hip: pt_conv_bf16x6, pt_conv_wgrad_bf16x6, pt_split_bf16x3_gather, pt_planes_combine,
point_teacher_amd/planes.py): the Bottlenecks of backbones/resnet.py:262-303, FPN / PSAGG 1x1 and 3x3 convolutions
(necks/fpn.py:151-202, necks/ps_fpn.py:56-75) and the towers (anchor_free_head.py:198-219) with activations travelling as split planes.
Every product is checked against a float64 convolution; the bar is the fp32 library kernel's own error on the same inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _planes_to_f32(t, rows, C):
    from point_teacher_amd import planes as PL
    n = PL.n_of(t)
    assert float(t[:, rows * C:n].float().abs().max()) == 0, 'the zero row'
    return PL.planes_to_f32(t, rows, C)


def _rows(x):                                        # [B, C, H, W] -> [B*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])


def _nchw(rows, B, H, W):
    return rows.view(B, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize('B,H,W,Cin,Cout,K,stride', [
    (6, 100, 100, 128, 512, 1, 1),       # layer2 conv3
    (6, 100, 100, 512, 128, 1, 1),       # layer2 conv1
    (2, 50, 50, 1024, 256, 1, 1),        # teacher layer3 conv1
    (6, 25, 25, 2048, 512, 1, 1),        # layer4 conv1
    (3, 100, 100, 256, 128, 1, 2),       # layer2.0 conv1 read through the stride
    (2, 51, 37, 256, 512, 1, 2),         # odd sizes, stride 2
    (6, 25, 25, 512, 512, 3, 1),         # layer4 conv2
    (2, 33, 29, 128, 128, 3, 1),
    (2, 33, 29, 128, 64, 3, 2),          # 3x3 stride 2 (forward only)
    (1, 9, 7, 32, 8, 1, 1),              # narrow output
])
def test_conv_forward_epilogues_vs_fp64(B, H, W, Cin, Cout, K, stride):
    from point_teacher_amd import functional as F, planes as PL
    g = torch.Generator().manual_seed(B * H + Cin + K)
    pad = (K - 1) // 2
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, K, K, generator=g) * (2.0 / (Cin * K * K)) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    xt, _, _ = PL.split_nhwc(x)
    assert torch.equal(_planes_to_f32(xt, B * H * W, Cin), _rows(x))                     # planes are exact
    Ho, Wo = PL.out_hw(H, W, K, stride, pad)
    M = B * Ho * Wo
    res = torch.randn(M, Cout, generator=g).to(DEV)
    rt, _, _ = PL.split_nhwc(_nchw(res, B, Ho, Wo).contiguous(memory_format=torch.channels_last))
    wp = F._conv_weight_planes(w, False)
    ref0 = torch.nn.functional.conv2d(x.double(), w.double(), None, stride, pad)
    mag = torch.nn.functional.conv2d(x.double().abs(), w.double().abs(), None, stride, pad)
    lib = torch.nn.functional.conv2d(x, w, None, stride, pad)
    # plain product
    _, y = PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, stride, pad, want_f32=True, splits=1)
    e_mine = float(((_nchw(y, B, Ho, Wo).double() - ref0).abs() / mag).max())
    e_lib = float(((lib.double() - ref0).abs() / mag).max())
    print(f'[{B}x{H}x{W} {Cin}->{Cout} k{K} s{stride}] bf16x6 {e_mine:.3e}  fp32 library {e_lib:.3e}')
    assert e_mine <= max(e_lib, 2.0 ** -23) and e_mine < 3e-7
    # full epilogue: scale, shift, planes residual, fp32 residual, ReLU; planes and fp32 outputs agree exactly
    yp, yf = PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, stride, pad, scale=scale, shift=shift, res_planes=rt, res_f32=res, relu=True,
                            want_planes=True, want_f32=True)
    want = torch.relu(y * scale + shift + res + res)                      # (the library may split k here: another summation order)
    torch.testing.assert_close(yf, want, rtol=1e-5, atol=1e-5)
    assert torch.equal(_planes_to_f32(yp, M, Cout), yf)
    # every tile height gives the same numbers
    for rows in (96, 160, 256):
        d_y = PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, stride, pad, want_f32=True, tile_rows=rows, splits=1)[1]
        assert torch.equal(d_y, y)


@pytest.mark.parametrize('B,H,W,Cin,Cout,K,stride,S', [(2, 25, 25, 2048, 512, 1, 1, 4), (2, 25, 25, 512, 512, 3, 1, 5), (1, 13, 11, 256, 128, 3, 2, 7),
                                                    (2, 50, 50, 256, 256, 3, 1, 0)])
def test_split_k_convolution(B, H, W, Cin, Cout, K, stride, S):
    """Few output tiles, long reduce dimension (the teacher's batch, layer4): k is cut into chunks (also in the middle of a tap),
    parts are added in a fixed order and the epilogue runs in the finishing launch.  S = 0: the library's own choice."""
    from point_teacher_amd import functional as F, planes as PL, hip
    g = torch.Generator().manual_seed(Cin + K + S)
    pad = (K - 1) // 2
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, K, K, generator=g) * (2.0 / (Cin * K * K)) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    scale, shift = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    xt, _, _ = PL.split_nhwc(x)
    Ho, Wo = PL.out_hw(H, W, K, stride, pad)
    res = torch.randn(B * Ho * Wo, Cout, generator=g).to(DEV)
    wp = F._conv_weight_planes(w, False)
    if S == 0:
        assert hip.call('pt_conv_bf16x6_splits', B, H, W, Cin, Cout, K, K, stride, pad, 0) > 1
    kw = dict(scale=scale, shift=shift, res_f32=res, relu=True, want_planes=True, want_f32=True)
    p1, f1 = PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, stride, pad, splits=1, **kw)
    pS, fS = PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, stride, pad, splits=S if S else None, **kw)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), None, stride, pad) * scale.double().view(1, -1, 1, 1)
                     + shift.double().view(1, -1, 1, 1) + _nchw(res, B, Ho, Wo).double())
    e1 = float((_nchw(f1, B, Ho, Wo).double() - ref).abs().max() / ref.abs().max())
    eS = float((_nchw(fS, B, Ho, Wo).double() - ref).abs().max() / ref.abs().max())
    print(f'split-k {S}: rel err {eS:.3e} (unsplit {e1:.3e})')
    assert eS < 2e-6 and eS <= 2 * e1 + 1e-7
    assert torch.equal(_planes_to_f32(pS, B * Ho * Wo, Cout), fS)
    p2, f2 = PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, stride, pad, splits=S if S else None, **kw)
    assert torch.equal(f2, fS)                                        # deterministic


def test_gather_split_reads_the_stride_pixels():
    from point_teacher_amd import planes as PL
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 11, 9, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    t, Ho, Wo = PL.split_nhwc(x, 2)
    assert (Ho, Wo) == (6, 5)
    assert torch.equal(_planes_to_f32(t, 2 * 6 * 5, 64), _rows(x[:, :, ::2, ::2]))


@pytest.mark.parametrize('B,H,W,Cin,Cout,K', [(6, 100, 100, 512, 128, 1), (2, 25, 25, 512, 2048, 1), (2, 40, 36, 128, 128, 3), (6, 25, 25, 512, 512, 3)])
def test_input_gradient_with_mask_and_identity_vs_fp64(B, H, W, Cin, Cout, K):
    """dx = mask * (g (diag(scale) W) + identity gradient): the mode-1 weight planes with the BatchNorm scale folded in, the ReLU mask
    of x read from x's own planes, the by-passing gradient added from planes."""
    from point_teacher_amd import functional as F, planes as PL
    gen = torch.Generator().manual_seed(Cin + K)
    pad = (K - 1) // 2
    P = B * H * W
    x = torch.relu(torch.randn(P, Cin, generator=gen)).to(DEV)                 # a ReLU output: ~half zeros
    xt, _, _ = PL.split_nhwc(_nchw(x, B, H, W).contiguous(memory_format=torch.channels_last))
    w = (torch.randn(Cout, Cin, K, K, generator=gen) * (2.0 / (Cout * K * K)) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    scale = (torch.rand(Cout, generator=gen) + 0.5).to(DEV)
    gy = torch.randn(P, Cout, generator=gen).to(DEV)
    idn = torch.randn(P, Cin, generator=gen).to(DEV)
    gt, _, _ = PL.split_nhwc(_nchw(gy, B, H, W).contiguous(memory_format=torch.channels_last))
    it, _, _ = PL.split_nhwc(_nchw(idn, B, H, W).contiguous(memory_format=torch.channels_last))
    wd = F._conv_weight_planes(w, True, scale)
    gp, gf = PL.launch_conv(gt, B, H, W, Cout, wd, Cin, K, 1, K - 1 - pad, res_planes=it, mask_planes=xt, want_planes=True, want_f32=True)
    ref = torch.nn.grad.conv2d_input((B, Cin, H, W), w.double(), (_nchw(gy, B, H, W).double() * scale.double().view(1, -1, 1, 1)), 1, pad)
    ref = (_rows(ref) + idn.double()) * (x > 0)
    err = float((gf.double() - ref).abs().max() / ref.abs().max())
    print(f'[{B}x{H}x{W} {Cout}->{Cin} k{K}] input gradient rel err {err:.3e}')
    assert err < 2e-6
    assert torch.equal(_planes_to_f32(gp, P, Cin), gf)
    assert float(gf[x <= 0].abs().max()) == 0


def test_stride2_input_gradient_is_scattered():
    """The input gradient of a stride-2 1x1 convolution: computed on the coarse grid, added to the downsample path's fp32 gradient,
    masked by x at (2y, 2x) and written there; every other pixel keeps the zero the caller provided."""
    from point_teacher_amd import functional as F, planes as PL
    gen = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout = 2, 51, 38, 256, 128
    Ho, Wo = 26, 19
    x = torch.relu(torch.randn(B, Cin, H, W, generator=gen)).to(DEV).contiguous(memory_format=torch.channels_last)
    xt, _, _ = PL.split_nhwc(x)
    w = (torch.randn(Cout, Cin, 1, 1, generator=gen) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B * Ho * Wo, Cout, generator=gen).to(DEV)
    other = torch.randn(B * Ho * Wo, Cin, generator=gen).to(DEV)
    gt, _, _ = PL.split_nhwc(_nchw(gy, B, Ho, Wo).contiguous(memory_format=torch.channels_last))
    gp, gf = PL.launch_conv(gt, B, Ho, Wo, Cout, F._conv_weight_planes(w, True), Cin, 1, 1, 0, res_f32=other, mask_planes=xt,
                            want_planes=True, want_f32=True, scatter=(H, W))
    ref = torch.zeros(B, Cin, H, W, dtype=torch.float64, device=DEV)
    coarse = _nchw(gy.double() @ w.double().view(Cout, Cin) + other.double(), B, Ho, Wo)
    ref[:, :, ::2, ::2] = coarse
    ref = ref * (x > 0)
    got = _nchw(gf, B, H, W)
    assert float((got.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    assert torch.equal(_planes_to_f32(gp, B * H * W, Cin), gf)


@pytest.mark.parametrize('B,H,W,Cin,Cout,K,stride,bias', [
    (6, 100, 100, 128, 512, 1, 1, False), (6, 100, 100, 512, 128, 1, 1, True), (3, 100, 100, 256, 128, 1, 2, False),
    (2, 25, 25, 2048, 512, 1, 1, False), (2, 50, 50, 256, 256, 3, 1, True), (2, 17, 23, 128, 256, 3, 1, True), (1, 6, 5, 128, 128, 1, 1, True)])
def test_weight_and_bias_gradient_vs_fp64(B, H, W, Cin, Cout, K, stride, bias):
    from point_teacher_amd import planes as PL
    gen = torch.Generator().manual_seed(Cin + Cout + K)
    pad = (K - 1) // 2
    Ho, Wo = PL.out_hw(H, W, K, stride, pad)
    x = torch.randn(B, Cin, H, W, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, Cout, Ho, Wo, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)
    scale = (torch.rand(Cout, generator=gen) + 0.5).to(DEV)
    xt, _, _ = PL.split_nhwc(x)
    gt, _, _ = PL.split_nhwc(gy)
    dw, db = PL.launch_wgrad(gt, xt, B, H, W, Cin, Cout, K, stride, pad, row_scale=scale, want_bias=bias)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, K, K), gy.double(), stride, pad) * scale.double().view(-1, 1, 1, 1)
    lib = torch.nn.grad.conv2d_weight(x, (Cout, Cin, K, K), gy, stride, pad) * scale.view(-1, 1, 1, 1)
    mag = torch.nn.grad.conv2d_weight(x.double().abs(), (Cout, Cin, K, K), gy.double().abs(), stride, pad) * scale.double().view(-1, 1, 1, 1)
    e_mine = float(((dw.double() - ref).abs() / mag).max())
    e_lib = float(((lib.double() - ref).abs() / mag).max())
    print(f'[{B}x{H}x{W} {Cin}->{Cout} k{K} s{stride}] wgrad bf16x6 {e_mine:.3e}  fp32 library {e_lib:.3e}')
    assert dw.shape == (Cout, Cin, K, K)
    assert e_mine <= max(e_lib, 2.0 ** -23) and e_mine < 3e-7
    if bias:
        rb = gy.double().sum((0, 2, 3))
        assert float((db.double() - rb).abs().max() / gy.double().abs().sum((0, 2, 3)).max()) < 3e-7
    dw2, _ = PL.launch_wgrad(gt, xt, B, H, W, Cin, Cout, K, stride, pad, row_scale=scale, want_bias=bias)
    assert torch.equal(dw, dw2)                                                           # deterministic


def test_weight_planes_of_1x1_and_3x3_in_one_launch_with_folded_scale(monkeypatch):
    from point_teacher_amd import functional as F, hip
    g = torch.Generator().manual_seed(1)
    ws = [(torch.randn(o, i, k, k, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
          for o, i, k in ((128, 256, 1), (256, 128, 3), (512, 128, 1), (64, 32, 3))]
    sc = [(torch.rand(w.shape[0], generator=g) + 0.5).to(DEV) for w in ws]

    def expect(w, dgrad, s):
        wd = w if s is None else w * s.view(-1, 1, 1, 1)
        O, I, k = wd.shape[0], wd.shape[1], wd.shape[2]
        m = wd.flip(2, 3).permute(1, 2, 3, 0).reshape(I, k * k * O) if dgrad else wd.permute(0, 2, 3, 1).reshape(O, k * k * I)
        return F.split_bf16x3(m.contiguous())
    calls = []
    real = hip.call
    monkeypatch.setattr(hip, 'call', lambda fn, *a: (calls.append(fn), real(fn, *a))[1])
    F.PARAM_EPOCH[0] += 1
    for w, s in zip(ws, sc):
        for dgrad in (False, True):
            for use in (None, s):
                F._conv_weight_planes(w, dgrad, use)
    calls.clear()
    F.PARAM_EPOCH[0] += 1
    for w, s in zip(ws, sc):
        for dgrad in (False, True):
            for use in (None, s):
                got, ref = F._conv_weight_planes(w, dgrad, use), expect(w, dgrad, use)
                assert (got.rows, got.k) == (ref.rows, ref.k)
                assert torch.equal(got.planes, ref.planes), (w.shape, dgrad, use is not None)
    assert calls.count('pt_conv_weight_planes_batch') == 1
    # a torch in-place edit of a weight (load_state_dict, init) re-splits although no optimizer step bumped the epoch
    with torch.no_grad():
        ws[0].mul_(2.0)
    assert torch.equal(F._conv_weight_planes(ws[0], False).planes, expect(ws[0], False, None).planes)


def _ref_bottleneck(blk, x):
    """The block in float64 torch (eval-mode BatchNorm as an affine map)."""
    import torch.nn.functional as TF

    def cb(x, conv, bn, relu, res=None):
        y = TF.conv2d(x, conv.weight.double(), None, conv.stride, conv.padding)
        sc = (bn.weight.double() * torch.rsqrt(bn.running_var.double() + bn.eps)).view(1, -1, 1, 1)
        y = y * sc + (bn.bias.double().view(1, -1, 1, 1) - bn.running_mean.double().view(1, -1, 1, 1) * sc)
        if res is not None:
            y = y + res
        return torch.relu(y) if relu else y
    idn = x if blk.downsample is None else cb(x, blk.downsample[0], blk.downsample[1], False)
    out = cb(x, blk.conv1, blk.bn1, True)
    out = cb(out, blk.conv2, blk.bn2, True)
    return cb(out, blk.conv3, blk.bn3, True, idn)


def _make_block(inplanes, planes, stride, ds, seed, style='caffe', train_bn=False):
    from point_teacher_amd import nn_modules as NM
    import torch.nn as nn
    torch.manual_seed(seed)
    down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4)) if ds else None
    blk = NM.Bottleneck(inplanes, planes, stride, 1, down, style).to(DEV).to(memory_format=torch.channels_last)
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
            m.weight.requires_grad = train_bn
            m.bias.requires_grad = train_bn
    blk.eval()
    if train_bn:
        NM.refresh_bn_affines(blk, trainable=True)          # (ResNet.forward does this once per parameter update)
    return blk


@pytest.mark.parametrize('inplanes,planes,stride,ds,B,H,W,from_stem,style,train_bn', [
    (512, 128, 1, False, 2, 40, 36, False, 'caffe', False),       # identity block
    (256, 128, 2, True, 2, 41, 36, True, 'caffe', False),         # layer2.0: fp32 stem output read at the stride's pixels, no input gradient
    (512, 256, 2, True, 2, 40, 37, False, 'caffe', False),        # layer3.0: plane input, stride-2 input gradient scattered
    (512, 128, 1, True, 2, 20, 20, False, 'caffe', False),        # a first block without stride
    # OBB config 5: `pytorch` style (stride on the 3x3: its input gradient is a transposed convolution), eval-mode BatchNorm whose
    # affine trains (d gamma / d beta from the raw weight gradient and the bias sums)
    (512, 128, 1, False, 2, 40, 36, False, 'pytorch', True),
    (256, 128, 2, True, 2, 41, 36, True, 'pytorch', True),
    (512, 256, 2, True, 2, 40, 37, False, 'pytorch', True),
    (512, 256, 2, True, 2, 38, 38, False, 'caffe', True),
])
@pytest.mark.parametrize('mode', ['h2', 'h2_relink', 'b3_f16conv2', 'b3'])
def test_bottleneck_planes_vs_fp64(inplanes, planes, stride, ds, B, H, W, from_stem, style, train_bn, mode, monkeypatch):
    """The three operand modes of the trunk - the same bars:
    h2: every activation and gradient of the block as scaled fp16 x 2 planes, three MFMA products, one gradient scale for the chain
        (functional.F16_TRUNK, the default); h2_relink: the same with a fresh scale at every link (the census' fall-back for a chain
        that left fp16's range: PT_F16_CHAIN=0);
    b3_f16conv2: bf16 x 3 planes between the layers, the 3x3 alone on fp16 operands (PT_F16_TRUNK=0: round 4's default);
    b3: bf16 x 3 operands and six products everywhere (PT_F16_FC=0: the strict 24-bit path)."""
    from point_teacher_amd import functional as F, planes as PL
    PL.CENSUS.reset()
    monkeypatch.setattr(F, 'F16_TRUNK', mode.startswith('h2'))
    monkeypatch.setattr(F, 'F16_CHAIN', mode != 'h2_relink')
    monkeypatch.setattr(F, 'F16_BLOCK3', mode == 'b3_f16conv2')
    blk = _make_block(inplanes, planes, stride, ds, 11, style, train_bn)
    assert blk.plane_ok()
    g = torch.Generator().manual_seed(3)
    x = torch.relu(torch.randn(B, inplanes, H, W, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
    if from_stem:
        out = blk.forward_planes(x)
    else:
        x.requires_grad_(True)
        xa = PL.to_planes(x, f16=mode.startswith('h2'), group='trunk', carrier=False)
        xa.relu = True                              # (x is a ReLU output; the test plays its producer)
        out = blk.forward_planes(xa)
    assert out.f16 == mode.startswith('h2')
    y = out.float()
    xr = x.detach().double().requires_grad_(not from_stem)
    ref = _ref_bottleneck(blk, xr)
    assert y.shape == ref.shape
    assert float((y.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    gy = torch.randn(y.shape, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    params = [blk.conv1.weight, blk.conv2.weight, blk.conv3.weight] + ([blk.downsample[0].weight] if ds else [])
    names = ['w1', 'w2', 'w3'] + (['wd'] if ds else [])
    if train_bn:
        for n, bn in (('1', blk.bn1), ('2', blk.bn2), ('3', blk.bn3)) + ((('d', blk.downsample[1]),) if ds else ()):
            params += [bn.weight, bn.bias]
            names += ['gamma' + n, 'beta' + n]
    ins = params + ([] if from_stem else [x])
    got = torch.autograd.grad(y, ins, gy)
    want = torch.autograd.grad(ref, params + ([] if from_stem else [xr]), gy.double())
    for name, a, b in zip(names + ([] if from_stem else ['x']), got, want):
        if name == 'x':
            b = b * (x > 0)                        # the producer's ReLU mask is applied by the block's epilogue (gradient convention)
        err = float((a.double() - b).abs().max() / b.abs().max())
        print(name, f'{err:.3e}')
        assert err < 5e-6, (name, err)
    if mode.startswith('h2'):
        cen = PL.CENSUS.poll(sync=True)
        assert cen and all(d['saturated'] == 0 for d in cen.values()) and not PL.CENSUS.demoted, (cen, PL.CENSUS.demoted)


def test_resnet_fpn_planes_vs_fp64_and_the_fp32_routing(monkeypatch):
    """ResNet-50 (caffe, frozen stem + layer1, frozen eval-mode BatchNorms) + FPN + PSAGG: the plane-native trunk and the round-3
    routing (PT_PLANE_TRUNK off: library 1x1 convolutions, fp32 between layers) on the same weights against the same network in
    float64 - outputs and every parameter gradient; the plane trunk may not be further from float64 than the fp32 routing is."""
    import copy
    from point_teacher_amd import nn_modules as NM
    torch.manual_seed(0)
    bb = NM.ResNet(50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=False),
                   norm_eval=True, style='caffe').to(DEV).to(memory_format=torch.channels_last)
    fpn = NM.FPN([256, 512, 1024, 2048], 256, 5, start_level=1, add_extra_convs='on_output', relu_before_extra_convs=True).to(DEV).to(memory_format=torch.channels_last)
    agg = NM.PSAGG(5, 256, 256).to(DEV).to(memory_format=torch.channels_last)
    with torch.no_grad():
        bb.conv1.weight.mul_(1 / 64)
    bb.train()
    x = (torch.rand(2, 3, 256, 256, device=DEV) * 255).contiguous(memory_format=torch.channels_last)
    mods = (bb, fpn, agg)
    names = [f'{i}.{n}' for i, m in enumerate(mods) for n, p in m.named_parameters() if p.requires_grad]
    gy = None

    def run(mods, planes, xin):
        nonlocal gy
        monkeypatch.setattr(NM, '_PLANE_TRUNK', planes)
        params = [p for m in mods for p in m.parameters() if p.requires_grad]
        feats = mods[0](xin, planes=planes)
        if planes:
            from point_teacher_amd import planes as PL
            assert all(isinstance(f, PL.PlaneAct) for f in feats[1:]), [type(f) for f in feats]
        out = mods[2](mods[1](feats))[0]
        if gy is None:
            gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(9)).to(DEV).contiguous(memory_format=torch.channels_last)
        grads = torch.autograd.grad(out, params, gy.to(out.dtype), allow_unused=True)
        return out.detach(), grads
    o1, g1 = run(mods, True, x)
    o0, g0 = run(mods, False, x)
    m64 = tuple(copy.deepcopy(m).double() for m in mods)
    m64[0].train()
    o64, g64 = run(m64, False, x.double())
    e1, e0 = float((o1.double() - o64).abs().max() / o64.abs().max()), float((o0.double() - o64).abs().max() / o64.abs().max())
    print(f'output vs float64: plane trunk {e1:.3e}, fp32 routing {e0:.3e}')
    assert e1 <= max(2 * e0, 2e-6)
    # Gradients: one ReLU decision that differs between an fp32 and the float64 evaluation (a pre-activation within an ulp of zero;
    # ~1e7 activations here) moves every gradient upstream of it by 1e-4 ... 1e-2 - in either fp32 routing, on different elements.
    # The per-kernel and per-block tests above pin the arithmetic at 1e-6; this one pins the WIRING (identity / downsample
    # gradients, fan-out at the stage outputs, masks, bias gradients): a missing term is an O(1) error.
    d1s, d0s = [], []
    for n, a, b, r in zip(names, g1, g0, g64):
        assert (a is None) == (r is None), n
        if a is not None:
            d1s.append(float((a.double() - r).abs().max() / (r.abs().max() + 1e-300)))
            d0s.append(float((b.double() - r).abs().max() / (r.abs().max() + 1e-300)))
            print(f'{n:40s} plane {d1s[-1]:.3e}  fp32 {d0s[-1]:.3e}')
            assert d1s[-1] <= max(3 * d0s[-1], 2e-2), (n, d1s[-1], d0s[-1])
    med = sorted(d1s)[len(d1s) // 2]
    print(f'parameter-gradient error vs float64: plane trunk worst {max(d1s):.3e} median {med:.3e}, fp32 routing worst {max(d0s):.3e}')
    assert med < 1e-4


@pytest.mark.parametrize('f16', [False, True])
def test_roi_planes_and_fc_stack_vs_fp64(f16):
    """RoIAlign writing split planes + the FC stack as 1x1 convolutions over the RoIs (fcos_head_p2b_ts.py:1202-1236): the planes sum
    to pt_roi_align_fwd's block bit for bit; Linear -> ReLU -> Linear -> ReLU and every gradient (input map through the fp32
    carrier, weights, biases) against the same stack in float64.  f16: the first layer on fp16 x 2 operands and three MFMA products
    (RoI block as two fp16 planes within 2^-22, weights x 16, gradient planes x 2^12) - the SAME bars."""
    from point_teacher_amd import functional as F, planes as PL
    g = torch.Generator().manual_seed(4)
    B, C, H, W, K = 2, 256, 40, 36, 2500
    feat = torch.randn(B, C, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    cx, cy = torch.rand(K, generator=g) * (W * 8 - 40) + 20, torch.rand(K, generator=g) * (H * 8 - 40) + 20
    wh = torch.rand(K, 2, generator=g) * 60 + 4
    rois = torch.stack([torch.randint(0, B, (K,), generator=g).float(), cx - wh[:, 0] / 2, cy - wh[:, 1] / 2, cx + wh[:, 0] / 2, cy + wh[:, 1] / 2], 1).to(DEV)
    ref_block = F.roi_align(feat.detach(), rois, 7, 0.125, 0, True, 5)
    assert F.roi_align_planes_ok(feat, rois, 7)
    t = F.roi_align_planes(feat, rois, 0.125, 0, True, 5, f16=f16)
    if f16:
        n = (K + 1) * C * 49
        assert t.dtype == torch.float16 and tuple(t.shape) == (2, n + 8) and float(t[:, K * C * 49:n].abs().max()) == 0
        assert PL.tail_value(t.detach()) == 1.0                           # forward activations are stored unscaled
        val = t[0, :K * C * 49].detach().float() + t[1, :K * C * 49].detach().float()
        ref_flat = ref_block.flatten()
        assert float((val - ref_flat).abs().max()) <= 2.0 ** -22 * float(ref_flat.abs().max()) + 3e-8
    else:
        assert torch.equal(_planes_to_f32(t.detach(), K, C * 49), ref_block.flatten(1))
    torch.manual_seed(1)
    fc1, fc2 = torch.nn.Linear(C * 49, 1024).to(DEV), torch.nn.Linear(1024, 1024).to(DEV)
    x = PL.PlaneAct(t, 1, K, 1, C * 49, False, gcarrier=True)
    y1 = PL.linear(x, fc1, relu=True, out_planes=True)
    y = PL.linear(y1, fc2, relu=True)
    f64 = feat.detach().double().requires_grad_(True)
    blk = _roi_ref(f64, rois.double(), 0.125)
    y1r = torch.relu(blk @ fc1.weight.double().t() + fc1.bias.double())
    yr = torch.relu(y1r @ fc2.weight.double().t() + fc2.bias.double())
    assert float((y.detach().double() - yr.detach()).abs().max() / yr.detach().abs().max()) < 1e-5      # (the fp32 RoIAlign feeds 12544-term sums)
    # a ReLU whose pre-activation is within rounding of zero may decide differently in fp32 and float64 (2 x 2.5 M of them here);
    # the gradient of such a RoI then differs by a whole term - rows with any such element take no gradient in this comparison
    # (f16: the first layer hands TWO fp16 planes to the second, which stays on fp16 operands as well)
    assert y1.f16 == f16
    y1v = (y1.t[0, :K * 1024].float() + y1.t[1, :K * 1024].float()).view(K, 1024) if f16 else _planes_to_f32(y1.t.detach(), K, 1024)
    flip = ((y1v.detach() > 0) != (y1r.detach() > 0)).any(1) | ((y.detach() > 0) != (yr.detach() > 0)).any(1)
    print('rows with a ReLU decided differently in fp32 and float64:', int(flip.sum()))
    assert int(flip.sum()) <= 20
    gy = torch.randn(K, 1024, generator=g).to(DEV) * (~flip)[:, None]
    got = torch.autograd.grad(y, [feat, fc1.weight, fc1.bias, fc2.weight, fc2.bias], gy)
    want = torch.autograd.grad(yr, [f64, fc1.weight, fc1.bias, fc2.weight, fc2.bias], gy.double())
    for n, a, b in zip(('feat', 'w1', 'b1', 'w2', 'b2'), got, want):
        err = float((a.double() - b).abs().max() / b.abs().max())
        print(n, f'{err:.3e}')
        assert err < 5e-6, (n, err)


def _roi_ref(feat, rois, scale):
    """RoIAlign (aligned, adaptive grid) in plain differentiable torch float64: [K, C*49]."""
    import math
    B, C, H, W = feat.shape
    out = []
    for r in rois.tolist():
        b = int(r[0])
        x1, y1, x2, y2 = [v * scale - 0.5 for v in r[1:]]
        rw, rh = x2 - x1, y2 - y1
        gh, gw = max(int(math.ceil(rh / 7)), 1), max(int(math.ceil(rw / 7)), 1)
        ys = torch.tensor([y1 + ph * rh / 7 + (iy + .5) * rh / 7 / gh for ph in range(7) for iy in range(gh)], dtype=torch.float64, device=feat.device)
        xs = torch.tensor([x1 + pw * rw / 7 + (ix + .5) * rw / 7 / gw for pw in range(7) for ix in range(gw)], dtype=torch.float64, device=feat.device)

        def axis(v, L):
            valid = (v >= -1.0) & (v <= L)
            v = v.clamp(min=0)
            lo = v.floor().long().clamp(max=L - 1)
            hi = (lo + 1).clamp(max=L - 1)
            v = torch.where(lo >= L - 1, lo.double(), v)
            fr = v - lo.double()
            Wm = torch.zeros(v.numel(), L, dtype=torch.float64, device=feat.device)
            Wm.scatter_add_(1, lo[:, None], ((1 - fr) * valid)[:, None])
            Wm.scatter_add_(1, hi[:, None], (fr * valid)[:, None])
            return Wm
        Ay = axis(ys, H).view(7, gh, H).sum(1)
        Ax = axis(xs, W).view(7, gw, W).sum(1)
        o = torch.einsum('ph,chw,qw->cpq', Ay, feat[b], Ax) / (gh * gw)
        out.append(o.reshape(-1))
    return torch.stack(out)


# ------------------------------------------------------------------------------------------------ one plane: bf16 operands --
# BASELINE configs[2] (bf16 backbone / FPN / PSAGG under autocast): the same kernels with ONE plane per operand - bf16 in, one MFMA
# product, fp32 accumulation, the epilogue in fp32, one rounding to bf16 on the way out.
def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize('B,H,W,Cin,Cout,K,stride', [
    (6, 100, 100, 512, 128, 1, 1), (2, 50, 50, 1024, 256, 1, 1), (2, 51, 37, 256, 512, 1, 2), (6, 25, 25, 512, 512, 3, 1),
    (2, 33, 29, 128, 128, 3, 1), (2, 33, 29, 128, 64, 3, 2), (2, 25, 25, 2048, 512, 1, 1)])
def test_one_plane_conv_forward_is_the_fp32_accumulated_bf16_product(B, H, W, Cin, Cout, K, stride):
    from point_teacher_amd import functional as F, planes as PL
    g = torch.Generator().manual_seed(B * H + Cin + K + 1)
    pad = (K - 1) // 2
    x = _bf(torch.randn(B, Cin, H, W, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, K, K, generator=g) * (2.0 / (Cin * K * K)) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    shift = torch.randn(Cout, generator=g).to(DEV)
    xt, _, _ = PL.split_nhwc(x)
    assert xt.shape[0] == 1 and torch.equal(xt[0, :x.numel()].view(-1, Cin), _rows(x)) and float(xt[0, x.numel():].abs().max()) == 0
    Ho, Wo = PL.out_hw(H, W, K, stride, pad)
    M = B * Ho * Wo
    res = _bf(torch.randn(B, Cout, Ho, Wo, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
    rt, _, _ = PL.split_nhwc(res)
    wp = F._conv_weight_planes(w, False, None, 1)
    assert wp.planes.shape[0] == 1
    w16 = _bf(w)                                                        # the weight plane: round-to-nearest bf16 of the fp32 weight
    ref0 = torch.nn.functional.conv2d(x.double(), w16.double(), None, stride, pad)
    mag = torch.nn.functional.conv2d(x.double().abs(), w16.double().abs(), None, stride, pad)
    _, y = PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, stride, pad, want_f32=True, splits=1)
    e = float(((_nchw(y, B, Ho, Wo).double() - ref0).abs() / mag).max())
    print(f'[{B}x{H}x{W} {Cin}->{Cout} k{K} s{stride}] one plane, fp32 out: {e:.3e}')
    assert e < 3e-7                                                     # exact bf16 products, fp32 accumulation
    yp, yf = PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, stride, pad, scale=scale, shift=shift, res_planes=rt, relu=True,
                            want_planes=True, want_f32=True)
    want = torch.relu(y * scale + shift + _rows(res).float())
    torch.testing.assert_close(yf, want, rtol=1e-5, atol=1e-5)
    assert yp.shape[0] == 1 and torch.equal(yp[0, :M * Cout].view(M, Cout), _bf(yf)) and float(yp[0, M * Cout:].abs().max()) == 0
    S = PL.hip.call('pt_conv_bf16x6_splits', B, H, W, Cin, Cout, K, K, stride, pad, 0)
    if S > 1:                                                           # split-k launches give the same result up to the summation order
        _, ys = PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, stride, pad, want_f32=True)
        torch.testing.assert_close(ys, y, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('B,H,W,Cin,Cout,K,stride,bias', [(2, 50, 50, 256, 128, 1, 1, True), (2, 25, 25, 512, 512, 3, 1, False),
                                                       (2, 51, 37, 256, 512, 1, 2, False), (3, 20, 24, 128, 256, 3, 1, True)])
def test_one_plane_gradients_vs_fp64_of_the_bf16_operands(B, H, W, Cin, Cout, K, stride, bias):
    from point_teacher_amd import functional as F, planes as PL
    g = torch.Generator().manual_seed(Cin + Cout + K)
    pad = (K - 1) // 2
    Ho, Wo = PL.out_hw(H, W, K, stride, pad)
    x = _bf(torch.relu(torch.randn(B, Cin, H, W, generator=g))).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = _bf(torch.randn(B, Cout, Ho, Wo, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, K, K, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
    sc = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    xt, gt = PL.split_nhwc(x)[0], PL.split_nhwc(gy)[0]
    # weight (and bias) gradient: fp32 out, BatchNorm scale in the reduction
    dw, db = PL.launch_wgrad(gt, xt, B, H, W, Cin, Cout, K, stride, pad, row_scale=sc, want_bias=bias)
    xd = x.double().requires_grad_(True)
    wd_ = w.double().requires_grad_(True)
    y = torch.nn.functional.conv2d(xd, wd_, None, stride, pad)
    (y * gy.double() * sc.double().view(1, -1, 1, 1)).sum().backward()
    mag = float(wd_.grad.abs().max())
    assert float((dw.double() - wd_.grad).abs().max()) < 2e-6 * mag
    if bias:
        ref_b = gy.double().sum((0, 2, 3))
        assert float((db.double() - ref_b).abs().max()) < 1e-5 * float(ref_b.abs().max())
    if stride == 1:
        # input gradient with the scale folded into the (bf16) weights and the ReLU mask of x: one rounding of (w * scale), one of the result
        wdg = F._conv_weight_planes(w, True, sc, 1)
        gp, gf = PL.launch_conv(gt, B, Ho, Wo, Cout, wdg, Cin, K, 1, K - 1 - pad, mask_planes=xt, want_planes=True, want_f32=True)
        ws = _bf(w * sc.view(-1, 1, 1, 1)).double()
        ref = torch.nn.functional.conv_transpose2d(gy.double(), ws, None, 1, pad) * (x > 0)
        m2 = torch.nn.functional.conv_transpose2d(gy.double().abs(), ws.abs(), None, 1, pad)
        assert float(((_nchw(gf, B, H, W).double() - ref).abs() / m2.clamp(min=1e-30)).max()) < 3e-7
        assert torch.equal(gp[0, :gf.numel()].view_as(gf), _bf(gf))


def _r16(x):
    """Round to bf16 (round-to-nearest-even), keep float64."""
    return x.float().to(torch.bfloat16).double()


def test_one_plane_bottleneck_under_autocast_vs_fp64_with_the_same_roundings():
    """A Bottleneck (frozen BatchNorms) under bf16 autocast, plane-native with one plane per activation, against float64 arithmetic that
    rounds to bf16 at the SAME points: operands (activations, weights, weights with the folded BatchNorm scale in the input gradient),
    one rounding per fused epilogue, fp32 weight gradients.  (Against the unrounded block the result is within 0.3 % and the gradients
    within 2 - 5 % by norm - a rounded activation flips the ReLU decision of ~0.2 % of the entries, which alone is sqrt(0.002) of a
    gradient's norm; with the roundings reproduced only values that straddle a rounding boundary differ, by one bf16 ulp.)"""
    import torch.nn.functional as TF
    from point_teacher_amd import planes as PL
    blk = _make_block(512, 128, 1, False, 3)
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 40, 36
    x16 = torch.relu(torch.randn(B, 512, H, W, generator=g)).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, 512, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        assert blk.plane_ok()
        xin = x16.clone().requires_grad_(True)
        act = blk.forward_planes(PL.to_planes(xin))
        assert act.np == 1
        out = act.tensor()
    assert out.dtype == torch.bfloat16
    (out.float() * gy).sum().backward()

    def aff(bn):
        sc = bn.weight.double() * torch.rsqrt(bn.running_var.double() + bn.eps)
        return sc, bn.bias.double() - bn.running_mean.double() * sc
    (s1, h1), (s2, h2), (s3, h3) = aff(blk.bn1), aff(blk.bn2), aff(blk.bn3)
    w1, w2, w3 = blk.conv1.weight.double(), blk.conv2.weight.double(), blk.conv3.weight.double()
    v = lambda t: t.view(1, -1, 1, 1)
    x = x16.double()
    y1 = _r16(torch.relu(TF.conv2d(x, _r16(w1)) * v(s1) + v(h1)))
    y2 = _r16(torch.relu(TF.conv2d(y1, _r16(w2), padding=1) * v(s2) + v(h2)))
    o = _r16(torch.relu(TF.conv2d(y2, _r16(w3)) * v(s3) + v(h3) + x))
    E = _r16(gy) * (o > 0)
    E2 = _r16(TF.conv_transpose2d(E, _r16(w3 * s3.view(-1, 1, 1, 1))) * (y2 > 0))
    E1 = _r16(TF.conv_transpose2d(E2, _r16(w2 * s2.view(-1, 1, 1, 1)), padding=1) * (y1 > 0))
    gx = _r16(TF.conv_transpose2d(E1, _r16(w1 * s1.view(-1, 1, 1, 1))) + E)

    def wgrad(xin_, e, wshape, pad, sc):
        wv = torch.zeros(wshape, dtype=torch.float64, device=DEV, requires_grad=True)
        return torch.autograd.grad(TF.conv2d(xin_, wv, padding=pad), wv, e)[0] * sc.view(-1, 1, 1, 1)
    ref_g = {'conv3.weight': wgrad(y2, E, w3.shape, 0, s3), 'conv2.weight': wgrad(y1, E2, w2.shape, 1, s2), 'conv1.weight': wgrad(x, E1, w1.shape, 0, s1)}

    def rel(a, b):
        return float((a.detach().double() - b.double()).norm() / b.double().norm())
    errs = {n: rel(dict(blk.named_parameters())[n].grad, r) for n, r in ref_g.items()}
    print('one-plane bottleneck vs fp64 with the same roundings: out', rel(out, o), 'dx', rel(xin.grad, gx), errs)
    assert rel(out, o) < 1e-3 and rel(xin.grad, gx) < 3e-3
    for n, e in errs.items():
        assert dict(blk.named_parameters())[n].grad.dtype == torch.float32 and e < 3e-3, (n, e)


def test_weight_plane_refresh_survives_a_collection_in_the_middle(monkeypatch):
    """Weights of a discarded model sit in reference cycles; the cyclic collector may free them at any allocation - also between the
    refresh's liveness check and the table it builds (seen once in a full-suite run: `ref()` returned None).  The refresh holds
    strong references for its duration."""
    import gc
    from point_teacher_amd import functional as F
    holder = {'w': (torch.randn(128, 128, 1, 1, device=DEV) * 0.05).contiguous(memory_format=torch.channels_last)}
    keep = (torch.randn(128, 128, 3, 3, device=DEV) * 0.05).contiguous(memory_format=torch.channels_last)
    F._conv_weight_planes(holder['w'], False)
    want = F._conv_weight_planes(keep, False).planes.clone()
    real_zeros = F.np.zeros

    def zeros_and_collect(*a, **k):            # the allocation inside _build_table: drop the last outside reference, collect
        holder.pop('w', None)
        gc.collect()
        return real_zeros(*a, **k)
    monkeypatch.setattr(F.np, 'zeros', zeros_and_collect)
    F.PARAM_EPOCH[0] += 1
    F._CONV_W.table = None                      # (as after a registration: the next refresh rebuilds the table)
    got = F._conv_weight_planes(keep, False).planes
    assert torch.equal(got, want)
    monkeypatch.setattr(F.np, 'zeros', real_zeros)
    F.PARAM_EPOCH[0] += 1
    assert torch.equal(F._conv_weight_planes(keep, False).planes, want)      # the dead entry is dropped by the next refresh


@pytest.mark.parametrize('two', [False, True])
def test_fp16_operand_tower_chain_vs_fp64(two):
    """The dense head's towers on fp16 x 2 operands (functional.F16_FC; anchor_free_head.py:198-219): the map is split ONCE into two
    fp16 planes for one or both towers, three 3x3 convolutions + bias + ReLU pass fp16 planes to one another (three MFMA products
    each), the last writes fp32; backward: the output gradient becomes fp16 planes with a power-of-two scale chosen on the device,
    the chain keeps that scale, the input gradient returns as an fp32 carrier.  Everything against float64."""
    import torch.nn.functional as TF
    from point_teacher_amd import planes as PL
    from point_teacher_amd.nn_modules import ConvModule
    g = torch.Generator().manual_seed(7)
    B, C, H, W = 2, 128, 33, 29
    torch.manual_seed(3)
    towers = [[ConvModule(C, C, 3, padding=1).to(DEV).to(memory_format=torch.channels_last) for _ in range(3)] for _ in range(2 if two else 1)]
    x = torch.randn(B, C, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gys = [(torch.randn(B, C, H, W, generator=g) * 1e-4).to(DEV).contiguous(memory_format=torch.channels_last) for _ in towers]   # gradient-sized
    PL.CENSUS.reset()
    acts = PL.to_planes2(x, f16=True) if two else (PL.to_planes(x, f16=True),)
    for layers in towers:
        for l in layers:
            l.plane_group = 'towers'
    outs = []
    for act, layers in zip(acts, towers):
        assert act.f16 and act.gcarrier
        t = act
        for i, l in enumerate(layers):
            t = l(t, out_planes=i + 1 < len(layers))
            assert (i + 1 == len(layers)) or (t.f16 and t.np == 2)
        outs.append(t)
    params = [p for layers in towers for l in layers for p in (l.conv.weight, l.conv.bias)]
    got = torch.autograd.grad(outs, [x] + params, gys)
    xd = x.detach().double().requires_grad_(True)
    refs = []
    for layers in towers:
        t = xd
        for l in layers:
            t = torch.relu(TF.conv2d(t, l.conv.weight.double(), l.conv.bias.double(), padding=1))
        refs.append(t)
    want = torch.autograd.grad(refs, [xd] + params, [gy.double() for gy in gys])
    for o, r in zip(outs, refs):
        assert o.dtype == torch.float32 and float((o.double() - r).abs().max() / r.abs().max()) < 2e-6
    # ReLU decisions within rounding of zero differ between fp32 and float64 on a handful of elements: norm-wise bars
    # round 4: 2e-4 (the chain kept a scale in [512, 1024) chosen from plane values, ReLU flips un-masked); round 5, with the gradient
    # entering the chain from fp32 at a scale in [128, 256): measured 1.2e-7 ... 2.8e-7 - the bar is the bf16 x 3 path's 5e-6
    for n, a, b in zip(['x'] + [f'p{i}' for i in range(len(params))], got, want):
        err = float((a.double() - b.double()).norm() / b.double().norm())
        print(n, f'{err:.3e}')
        assert err < 5e-6, (n, err)
    cen = PL.CENSUS.poll(sync=True)                                    # the chain's scale: every gradient tensor's largest stored
    grads = {k: d for k, d in cen.items() if k[0] == 'towers_grad'}    # magnitude sits inside fp16's comfortable range
    assert grads and all(d['saturated'] == 0 and 0.25 <= d['amax_stored'] < 60000 for d in grads.values()), grads


def test_fp16_operand_single_convolution_vs_fp64():
    """An FPN output convolution (necks/fpn.py:175-177: 3x3, bias, no activation) on fp16 x 2 operands: fp32 map in, fp32 map out,
    input / weight / bias gradients from a gradient-sized output gradient, against float64."""
    import torch.nn.functional as TF
    from point_teacher_amd import planes as PL
    g = torch.Generator().manual_seed(9)
    B, C, H, W = 2, 256, 27, 31
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(C, C, 3, padding=1).to(DEV).to(memory_format=torch.channels_last)
    x = torch.randn(B, C, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = (torch.randn(B, C, H, W, generator=g) * 3e-6).to(DEV).contiguous(memory_format=torch.channels_last)
    y = PL.conv_module(x, conv, relu=False, f16=True)
    got = torch.autograd.grad(y, [x, conv.weight, conv.bias], gy)
    xd = x.detach().double().requires_grad_(True)
    yr = TF.conv2d(xd, conv.weight.double(), conv.bias.double(), padding=1)
    want = torch.autograd.grad(yr, [xd, conv.weight, conv.bias], gy.double())
    assert y.dtype == torch.float32 and float((y.double() - yr).abs().max() / yr.abs().max()) < 2e-6
    for n, a, b in zip(('x', 'w', 'b'), got, want):
        err = float((a.double() - b.double()).abs().max() / b.double().abs().max())
        assert err < 5e-6, (n, err)


def _h2_conv_ref(x, w, K):
    import torch.nn.functional as TF
    return TF.conv2d(x.double(), w.double(), None, 1, (K - 1) // 2)


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,Cin,Cout,K', [(2, 40, 37, 256, 256, 3), (1, 700, 1, 2048, 256, 1), (3, 50, 50, 128, 384, 1)])
def test_h2_every_tile_height_and_ring_depth_gives_the_same_product(B, H, W, Cin, Cout, K):
    """fp16 x 2 operands through every instantiation of the k-loop: 64 ... 256-row tiles = the two-stage form (64 - 128 rows, many
    workgroups), the deep-ring form of the same tiles (no more workgroups than CUs), the three / four-stage ring (160 rows and up),
    with and without a k-split.  The ring changes WHEN a stage is loaded, never what is multiplied: every variant gives the bits of
    the 64-row launch, and that launch is float64-close."""
    from point_teacher_amd import functional as F, planes as PL
    g = torch.Generator().manual_seed(Cin + K + H)
    pad = (K - 1) // 2
    x = torch.relu(torch.randn(B, Cin, H, W, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, K, K, generator=g) * (2.0 / (Cin * K * K)) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    xt = PL.split_nhwc(x, f16=True)[0]
    wp = F._conv_weight_planes(w, False, None, 2)
    ref = _h2_conv_ref(x, w, K)
    base = None
    for rows in (64, 96, 128, 160, 192, 224, 256):
        _, y = PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, 1, pad, want_f32=True, tile_rows=rows, splits=1)
        if base is None:
            base = y
            err = float((_nchw(y, B, H, W).double() - ref).abs().max() / ref.abs().max())
            assert err < 2e-6, err
        else:
            assert torch.equal(y, base), rows
    # a k-split: a different summation order (the chunks are added by the finish launch), not a different product
    KB = K * K * Cin // 32
    if KB >= 32:
        _, ys = PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, 1, pad, want_f32=True, tile_rows=64, splits=2)
        assert float((ys.double() - base.double()).abs().max() / base.double().abs().max()) < 1e-6
    # the library's own plan: the workspace it asks for is the one the launch takes
    _, ya = PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, 1, pad, want_f32=True)
    assert float((ya.double() - base.double()).abs().max() / base.double().abs().max()) < 1e-6


@pytest.mark.gpu
def test_h2_many_workgroups_take_the_two_stage_form_and_agree():
    """More tiles than CUs at 64 - 128 rows (the two-stage form with two workgroups per CU) against the 160-row ring on the same
    operands: a layer2-sized 3x3."""
    from point_teacher_amd import functional as F, planes as PL
    g = torch.Generator().manual_seed(77)
    B, H, W, Cin, Cout, K = 4, 100, 100, 128, 128, 3
    x = torch.relu(torch.randn(B, Cin, H, W, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, K, K, generator=g) * 0.03).to(DEV).contiguous(memory_format=torch.channels_last)
    xt = PL.split_nhwc(x, f16=True)[0]
    wp = F._conv_weight_planes(w, False, None, 2)
    ys = [PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, 1, 1, want_f32=True, tile_rows=r, splits=1)[1] for r in (64, 128, 160)]
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    ref = _h2_conv_ref(x, w, K)
    assert float((_nchw(ys[0], B, H, W).double() - ref).abs().max() / ref.abs().max()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,Cin,Cout,K', [(3, 64, 64, 128, 256, 1), (5000, 1, 1, 1024, 128, 1), (2, 30, 30, 128, 128, 3)])
def test_weight_gradient_ring_and_sliced_reduction_vs_fp64(B, H, W, Cin, Cout, K):
    """pt_conv_wgrad_bf16x6 on fp16 x 2 operands: the LDS ring, the flat row walk of B = rows x 1 x 1 pixels (the FC stacks) and the
    sliced reduction of a small result behind many chunks (S >= 8) against float64; the chunk count only changes the summation order."""
    from point_teacher_amd import hip, planes as PL
    g = torch.Generator().manual_seed(B + Cin + K)
    pad = (K - 1) // 2
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, Cout, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    xt, gt = PL.split_nhwc(x, f16=True)[0], PL.split_nhwc(gy, f16=True)[0]
    want = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, K, K), gy.double(), padding=pad)
    wb = gy.double().sum((0, 2, 3))
    M = B * H * W
    n = Cout * K * K * Cin
    outs = []
    for S in (1, 3, 8, 24):
        if S * 4 * 32 > M:
            continue
        ws = torch.empty(S * (n + Cout), device=DEV)
        dw, db = torch.empty(n, device=DEV), torch.empty(Cout, device=DEV)
        d = hip.STRUCTS['pt_conv_wgrad_desc']()
        d.B, d.Hs, d.Ws, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad = B, H, W, Cin, Cout, K, K, 1, pad
        d.np, d.operand_f16, d.alpha = 3, 1, 1.0
        d.alpha_dev, d.alpha_dev2 = PL.tail_ptr(gt), PL.tail_ptr(xt)
        d.gy_planes, d.gy_plane_stride = PL._p16(gt), gt.shape[1]
        d.x_planes, d.x_plane_stride = PL._p16(xt), xt.shape[1]
        d.dw, d.dbias = hip.dptr(dw, 'float'), hip.dptr(db, 'float')
        d.workspace, d.workspace_elems, d.splits = hip.dptr(ws, 'float'), ws.numel(), S
        hip.call('pt_conv_wgrad_bf16x6', d)
        got = dw.view(Cout, K, K, Cin).permute(0, 3, 1, 2).double()
        err = float((got - want).abs().max() / want.abs().max())
        eb = float((db.double() - wb).abs().max() / wb.abs().max())
        assert err < 2e-6 and eb < 2e-6, (S, err, eb)
        outs.append(got)
    assert len(outs) >= 2


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,Cin,Cout,K', [(2, 60, 50, 64, 64, 3), (3, 41, 37, 256, 64, 1), (6, 200, 200, 64, 64, 1)])
def test_h2_64_column_tiles_vs_fp64_and_the_128_column_form(B, H, W, Cin, Cout, K):
    """Results of <= 64 channels (the frozen layer1) take 64-column tiles at 64 / 128 rows: same products as the 128-column form of
    another tile height (bit-identical), float64-close, epilogue with BatchNorm terms, identity planes and ReLU included."""
    from point_teacher_amd import functional as F, planes as PL
    g = torch.Generator().manual_seed(Cin + Cout + K + H)
    pad = (K - 1) // 2
    x = torch.relu(torch.randn(B, Cin, H, W, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, K, K, generator=g) * (2.0 / (Cin * K * K)) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    r = torch.relu(torch.randn(B, Cout, H, W, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    xt, rt = PL.split_nhwc(x, f16=True)[0], PL.split_nhwc(r, f16=True)[0]
    wp = F._conv_weight_planes(w, False, None, 2)
    kw = dict(scale=sc, shift=sh, res_planes=rt, relu=True, want_planes=True, want_f32=True, splits=1)
    outs = [PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, 1, pad, tile_rows=rows, **kw) for rows in (64, 128, 96, 160)]
    for p, f in outs[1:]:                      # 64 / 128 rows: 64 columns; 96 / 160 rows: the 128-column kernels
        n = (B * H * W + 1) * Cout               # (behind the zero row: the fp32 tail word, then uninitialised padding)
        assert torch.equal(f, outs[0][1]) and torch.equal(p[:, :n], outs[0][0][:, :n]) and PL.tail_value(p) == PL.tail_value(outs[0][0])
    ref = torch.relu(_h2_conv_ref(x, w, K) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + r.double())
    err = float((_nchw(outs[0][1], B, H, W).double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err
    pa, fa = PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, 1, pad, **dict(kw, splits=None))     # the library's own choice
    assert torch.equal(fa, outs[0][1])
