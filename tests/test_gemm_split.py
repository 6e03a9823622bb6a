"""fp32 GEMM on the bf16 matrix cores (csrc/gemm_split.hip, `pt_split_bf16x3` + `pt_gemm_bf16x6_nt`): the MIL FC stacks
(fcos_head_p2b_ts.py:1202-1236, :1240-1256) as six bf16 MFMA products per fp32 product with fp32 accumulation.
Checked against a float64 reference; the bar is the fp32 library GEMM's own error on the same inputs (round-2 verdict, item 2)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _err(y, ref):
    """max |y - ref| / (|a| @ |b|^T): error in units of the sum of the products' magnitudes - what an fp32 chain's bound is quoted in."""
    return float(((y.double() - ref).abs() / _err.scale).max())


def test_split_planes_sum_back_exactly():
    """x0 + x1 + x2 == x bit for bit (8 + 8 + 8 significant bits, each step rounded to nearest); both orientations; padding is zero."""
    from point_teacher_amd import functional as F
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)
    x = (torch.randn(70, 45, generator=g) * torch.logspace(-6, 6, 45)).to(dev)
    x[3, 7] = 0.0
    x[5, 9] = 1e-30
    sp = F.split_bf16x3(x)
    assert (sp.rows, sp.k) == (70, 45) and sp.planes.shape == (3, 5 * 2 * 512) and sp.planes.dtype == torch.bfloat16
    p = sp.dense()                                                    # [3, 80, 64]: un-blocked, un-swizzled
    s = p[0] + p[1] + p[2]                                            # exact in fp32: the terms do not overlap
    assert torch.equal(s[:70, :45], x) and float(s[70:].abs().max()) == 0 and float(s[:, 45:].abs().max()) == 0
    assert torch.equal(p[0][:70, :45], x.to(torch.bfloat16).float())  # the leading term is the round-to-nearest bf16 of x
    st = F.split_bf16x3(x, transpose=True)
    assert (st.rows, st.k) == (45, 70)
    pt = st.dense()                                                   # [3, 48, 96]
    s2 = pt[0] + pt[1] + pt[2]
    assert torch.equal(s2[:45, :70], x.t()) and float(s2[45:].abs().max()) == 0 and float(s2[:, 70:].abs().max()) == 0
    v = x[:, 3:40]                                                    # a strided view (row stride 45, unaligned start)
    pv = F.split_bf16x3(v).dense()
    assert torch.equal((pv[0] + pv[1] + pv[2])[:70, :37], v)


@pytest.mark.parametrize('M,N,K', [(5000, 1024, 12544), (5000, 1024, 1024), (400, 1024, 12544), (1024, 12544, 5000), (333, 200, 96),
                                   (97, 130, 40)])
def test_gemm_matches_fp64_at_least_as_well_as_the_fp32_library(M, N, K):
    from point_teacher_amd import functional as F
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ref = a.double() @ b.double().t()
    _err.scale = a.double().abs() @ b.double().abs().t()
    y = F.gemm_bf16x6_nt(F.split_bf16x3(a), F.split_bf16x3(b))
    lib = a @ b.t()
    e_mine, e_lib = _err(y, ref), _err(lib, ref)
    print(f'[{M}x{N}x{K}] bf16x6 {e_mine:.3e}  fp32 library {e_lib:.3e}  (units of sum |a||b|)')
    assert e_mine <= max(e_lib, 2.0 ** -24), (e_mine, e_lib)
    assert e_mine < 3e-7
    # every tile height gives the same numbers up to the k-order inside a tile (none: the k loop is the same) -> identical
    for rows in (96, 128, 160, 192, 224, 256):
        y2 = F.gemm_bf16x6_nt(F.split_bf16x3(a), F.split_bf16x3(b), tile_rows=rows)
        assert torch.equal(y2, y), rows
    # epilogue: bias + ReLU
    yb = F.gemm_bf16x6_nt(F.split_bf16x3(a), F.split_bf16x3(b), bias=bias, relu=True)
    torch.testing.assert_close(yb, torch.relu(y + bias), rtol=0, atol=0)
    # the transposed splits bring the other two products of a Linear to the same kernel
    if M * N <= 6e6:
        gy = torch.randn(M, N, generator=g).to(dev)
        gx = F.gemm_bf16x6_nt(F.split_bf16x3(gy), F.split_bf16x3(b, transpose=True))            # gy @ b
        rx = gy.double() @ b.double()
        assert float((gx.double() - rx).abs().max() / rx.abs().max()) < 5e-6
        gw = F.gemm_bf16x6_nt(F.split_bf16x3(gy, transpose=True), F.split_bf16x3(a, transpose=True))   # gy^T @ a
        rw = gy.double().t() @ a.double()
        assert float((gw.double() - rw).abs().max() / rw.abs().max()) < 5e-6


def test_split_linear_autograd_matches_torch():
    from point_teacher_amd import functional as F
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    x = torch.randn(777, 12544, device=dev, requires_grad=True)
    fc = torch.nn.Linear(12544, 1024).to(dev)
    y = F.split_linear(x, fc.weight, fc.bias, relu=True)
    ref = torch.relu(torch.nn.functional.linear(x.double(), fc.weight.double(), fc.bias.double()))
    assert float((y.double() - ref).abs().max()) < 2e-5
    gy = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, (x, fc.weight, fc.bias), gy)
    rx, rw, rb = torch.autograd.grad(ref, (x, fc.weight, fc.bias), gy.double())
    for mine, r in ((gx, rx), (gw, rw), (gb, rb)):
        assert float((mine.double() - r).abs().max() / r.abs().max()) < 2e-6


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 100, 100, 256, 256), (3, 37, 41, 64, 128), (6, 100, 100, 256, 256)])
def test_conv3x3_bf16x6_matches_fp64_at_least_as_well_as_the_library(B, H, W, Cin, Cout):
    """The tower shape (3x3, stride 1, pad 1, NHWC) as an implicit GEMM on the split-bf16 kernel: values against a float64
    convolution (bar: the fp32 library kernel's error on the same inputs), the zero padding at every image border, bias + ReLU
    epilogue, and the three gradients through the autograd Function (the weight gradient: pt_conv3x3_wgrad_bf16x6_nhwc where
    both channel counts are multiples of 128, else the library's)."""
    from point_teacher_amd import functional as F
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(B * H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1).to(dev).to(memory_format=torch.channels_last)
    with torch.no_grad():
        conv.bias.normal_(0, 0.5)
    y = F.split_conv3x3(x, conv.weight, conv.bias, relu=False)
    assert y.shape == (B, Cout, H, W) and y.is_contiguous(memory_format=torch.channels_last)
    ref = torch.nn.functional.conv2d(x.detach().double(), conv.weight.detach().double(), conv.bias.detach().double(), padding=1)
    scale = torch.nn.functional.conv2d(x.detach().double().abs(), conv.weight.detach().double().abs(), None, padding=1) + conv.bias.detach().abs().double().view(1, -1, 1, 1)
    lib = torch.nn.functional.conv2d(x.detach(), conv.weight.detach(), conv.bias.detach(), padding=1)
    e_mine = float(((y.detach().double() - ref).abs() / scale).max())
    e_lib = float(((lib.double() - ref).abs() / scale).max())
    print(f'[{B}x{H}x{W} {Cin}->{Cout}] bf16x6 conv {e_mine:.3e}  fp32 library {e_lib:.3e}  (units of sum |x||w|)')
    assert e_mine <= max(e_lib, 2.0 ** -23) and e_mine < 3e-7
    # borders: the first / last rows and columns see zeros outside
    for sl in ((slice(None), slice(None), 0), (slice(None), slice(None), H - 1), (slice(None), slice(None), slice(None), 0),
               (slice(None), slice(None), slice(None), W - 1)):
        assert float((y.detach().double()[sl] - ref[sl]).abs().max()) < 1e-4
    yr = F.split_conv3x3(x, conv.weight, conv.bias, relu=True)
    torch.testing.assert_close(yr.detach(), torch.relu(y.detach()), rtol=0, atol=0)
    gy = torch.randn(B, Cout, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    gx, gw, gb = torch.autograd.grad(yr, (x, conv.weight, conv.bias), gy)
    xr = x.detach().double().requires_grad_(True)
    wr, br = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    # (the ReLU mask of the kernel's own output: an element within one rounding of 0 may fall on either side of the kink, and the
    # reference must differentiate the same piecewise-linear function)
    mask = (yr.detach() > 0).double()
    rx, rw, rb = torch.autograd.grad(torch.nn.functional.conv2d(xr, wr, br, padding=1), (xr, wr, br), gy.double() * mask)
    for mine, r in ((gx, rx), (gw, rw), (gb, rb)):
        assert float((mine.double() - r).abs().max() / r.abs().max()) < 5e-6


def test_conv3x3_bf16x6_with_frozen_batchnorm_epilogue():
    """conv -> frozen eval-mode BatchNorm -> ReLU (a ResNet bottleneck's conv2 / bn2, backbones/resnet.py:262-303) with the
    BatchNorm's (scale, shift) as the implicit GEMM's epilogue, through `nn_modules.conv_bn`: values and the gradients of the
    input and the weight against torch in float64."""
    from point_teacher_amd import nn_modules as NM
    dev = torch.device('cuda:0')
    torch.manual_seed(4)
    conv = torch.nn.Conv2d(128, 128, 3, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(128).to(dev).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 2.0)
    for p in bn.parameters():
        p.requires_grad = False
    x = torch.randn(2, 128, 70, 72, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    calls = []
    orig = NM.F.split_conv3x3
    NM.F.split_conv3x3 = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        y = NM.conv_bn(x, conv, bn, relu=True)
    finally:
        NM.F.split_conv3x3 = orig
    assert calls, 'the bottleneck shape did not take the implicit-GEMM path'
    xr, wr = x.detach().double().requires_grad_(True), conv.weight.detach().double().requires_grad_(True)
    sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).double()
    pre = torch.nn.functional.conv2d(xr, wr, None, padding=1) * sc.view(1, -1, 1, 1) + (bn.bias - bn.running_mean * sc.float()).double().view(1, -1, 1, 1)
    assert float((y.detach().double() - torch.relu(pre)).abs().max()) < 2e-5
    gy = torch.randn_like(y)
    gx, gw = torch.autograd.grad(y, (x, conv.weight), gy)
    rx, rw = torch.autograd.grad(pre, (xr, wr), gy.double() * (y.detach() > 0).double())
    for mine, r in ((gx, rx), (gw, rw)):
        assert float((mine.double() - r).abs().max() / r.abs().max()) < 5e-6


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 100, 100, 256, 256), (1, 37, 41, 128, 256), (2, 20, 17, 256, 128), (6, 100, 100, 128, 128)])
def test_conv3x3_wgrad_bf16x6_matches_fp64_at_least_as_well_as_the_library(B, H, W, Cin, Cout):
    """dW[o, c, ky, kx] = sum_p gy[p, o] x[p + shift, c] with the pixel index as the reduce dimension (transposed LDS reads of the
    row-major planes): against float64 in units of sum |gy||x| (bar: the fp32 library kernel), borders of every image, pixel
    counts that are not multiples of the 32-pixel k-step, rows narrower than a k-step, any number of pixel chunks, and bit-identical
    repeats (fixed summation order, no atomics)."""
    from point_teacher_amd import functional as F
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(B * H + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, Cout, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(Cout, Cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    P = B * H * W
    xp = F._split_rows(x.permute(0, 2, 3, 1).reshape(P, Cin))
    gp = F._split_rows(gy.permute(0, 2, 3, 1).reshape(P, Cout))
    dw = F._conv3x3_wgrad(gp, xp, B, H, W, Cin, Cout)
    assert dw.shape == (Cout, Cin, 3, 3) and dw.is_contiguous(memory_format=torch.channels_last)
    ref = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    scale = torch.ops.aten.convolution_backward(gy.double().abs(), x.double().abs(), w.double(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                [False, True, False])[1]
    lib = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    e_mine = float(((dw.double() - ref).abs() / scale).max())
    e_lib = float(((lib.double() - ref).abs() / scale).max())
    print(f'[{B}x{H}x{W} {Cin}->{Cout}] bf16x6 wgrad {e_mine:.3e}  fp32 library {e_lib:.3e}  (units of sum |gy||x|)')
    assert e_mine <= max(e_lib, 2.0 ** -23) and e_mine < 3e-7
    assert float((dw.double() - ref).abs().max() / ref.abs().max()) < 5e-6
    assert torch.equal(F._conv3x3_wgrad(gp, xp, B, H, W, Cin, Cout), dw)
    n = Cout * 9 * Cin
    for S in (1, 3, 8):
        ws = torch.empty((S * n,), device=dev)
        d2 = torch.empty((Cout, 3, 3, Cin), device=dev)
        F.hip.call('pt_conv3x3_wgrad_bf16x6_nhwc', gp, gp.shape[1], xp, xp.shape[1], d2, ws, S * n, B, H, W, Cin, Cout, S)
        assert float((d2.permute(0, 3, 1, 2).double() - ref).abs().max() / ref.abs().max()) < 5e-6, S


def test_conv3x3_weight_gradient_library_fallback_agrees():
    """PT_SPLIT_WGRAD=0 (or channel counts that are not multiples of 128) keeps the library's weight gradient behind the same
    autograd Function: both routes give the same three gradients, with and without the fused ReLU / BatchNorm scale."""
    from point_teacher_amd import functional as F
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(11)
    x = torch.randn(2, 128, 70, 66, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    conv = torch.nn.Conv2d(128, 256, 3, padding=1).to(dev).to(memory_format=torch.channels_last)
    scale = (torch.rand(256, generator=g) + 0.5).to(dev)
    gy = torch.randn(2, 256, 70, 66, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    res = {}
    for own in (True, False):
        F._SPLIT_WGRAD = own
        try:
            y1 = F.split_conv3x3(x, conv.weight, conv.bias, relu=True)
            y2 = F.split_conv3x3(x, conv.weight, conv.bias, relu=True, scale=scale)
            res[own] = (torch.autograd.grad(y1, (x, conv.weight, conv.bias), gy), torch.autograd.grad(y2, (x, conv.weight), gy))
        finally:
            F._SPLIT_WGRAD = True
    for a, b in zip(res[True][0] + res[True][1], res[False][0] + res[False][1]):
        assert float((a - b).abs().max()) <= 5e-6 * float(b.abs().max())


def test_conv_weight_planes_are_refreshed_in_one_launch_per_parameter_epoch():
    """Channels_last 3x3 weights register with the batched plane cache: both forms of every registered weight equal the planes of
    the per-weight split, they follow the weights after PARAM_EPOCH moves (and only then), a freed weight drops out, and a weight
    in another layout takes the per-weight path."""
    from point_teacher_amd import functional as F
    dev = torch.device('cuda:0')
    torch.manual_seed(2)
    ws = [torch.randn(o, i, 3, 3, device=dev).contiguous(memory_format=torch.channels_last) for o, i in ((256, 256), (128, 64), (512, 128))]

    def expect(w, dgrad):
        O, I = w.shape[:2]
        m = (w.flip(2, 3).permute(1, 2, 3, 0).reshape(I, 9 * O) if dgrad else w.permute(0, 2, 3, 1).reshape(O, 9 * I))
        return F.split_bf16x3(m.contiguous())

    F.PARAM_EPOCH[0] += 1
    for w in ws:
        for dgrad in (False, True):
            got, ref = F._conv_weight_planes(w, dgrad), expect(w, dgrad)
            assert (got.rows, got.k) == (ref.rows, ref.k) and torch.equal(got.planes, ref.planes)
    calls = []
    orig = F.hip.call
    F.hip.call = lambda fn, *a: (calls.append(fn), orig(fn, *a))[1]
    try:
        with torch.no_grad():
            for w in ws:
                w.mul_(1.5)
        # a torch in-place edit moves the weight's version counter: the planes follow without an optimizer step (round-3 advice:
        # load_state_dict / init / copy_ must not leave stale planes) - still ONE launch for all six (weight, form) pairs
        for w in ws:
            for dgrad in (False, True):
                assert torch.equal(F._conv_weight_planes(w, dgrad).planes, expect(w, dgrad).planes)
        assert calls.count('pt_conv_weight_planes_batch') == 1
        calls.clear()
        F._conv_weight_planes(ws[0], False)                            # same epoch, same version: the cached planes, no launch
        assert 'pt_conv_weight_planes_batch' not in calls
        F.PARAM_EPOCH[0] += 1                                          # raw-pointer updates (fused SGD / EMA) announce themselves
        for w in ws:
            for dgrad in (False, True):
                assert torch.equal(F._conv_weight_planes(w, dgrad).planes, expect(w, dgrad).planes)
        assert calls.count('pt_conv_weight_planes_batch') == 1
    finally:
        F.hip.call = orig
    n = len(F._CONV_W.ent)
    del ws[2], w
    F.PARAM_EPOCH[0] += 1
    F._conv_weight_planes(ws[0], True)
    assert len(F._CONV_W.ent) == n - 2
    plain = torch.randn(128, 64, 3, 3, device=dev)                    # NCHW-contiguous: per-weight path
    assert not F._ConvWeightPlanes.ok(plain)
    assert torch.equal(F._conv_weight_planes(plain, True).planes, expect(plain, True).planes)


def test_f16x3_experiment_matches_float64_inside_fp16_range():
    """EXPERIMENT (not on the training path): the same GEMM from fp16 x 2 operands with three fp16 MFMA products
    (pt_split_f16x2 + pt_gemm_f16x3_nt).  With operands of O(1) magnitude its error against float64 is the six-product kernel's
    (both below the fp32 library's); operands of 1e-4 magnitude lose it unless scaled by a power of two first."""
    from point_teacher_amd import functional as F, hip
    dev = 'cuda:0'
    M, N, K = 777, 300, 1000
    g = torch.Generator().manual_seed(4)

    def f16_gemm(a, b):
        def split(x):
            R, C = x.shape
            n = ((R + 15) // 16) * ((C + 31) // 32) * 512
            p = torch.empty((2, n), dtype=torch.bfloat16, device=dev)          # 16-bit storage: the header's uint16_t*
            hip.call('pt_split_f16x2', x, x.stride(0), R, C, p, n)
            return p
        ap, bp = split(a), split(b)
        out = torch.empty(a.shape[0], b.shape[0], device=dev)
        hip.call('pt_gemm_f16x3_nt', ap, ap.shape[1], bp, bp.shape[1], out, out.stride(0), None, a.shape[0], b.shape[0], a.shape[1], 0, 0)
        return out

    def err(out, a, b):
        ref = a.double() @ b.double().t()
        return float(((out.double() - ref).abs() / (a.double().abs() @ b.double().abs().t())).max())
    a = torch.randn(M, K, generator=g).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    e3, e6, el = err(f16_gemm(a, b), a, b), err(F.gemm_bf16x6_nt(F.split_bf16x3(a), F.split_bf16x3(b)), a, b), err(a @ b.t(), a, b)
    print(f'f16x3 {e3:.2e}  bf16x6 {e6:.2e}  fp32 library {el:.2e}')
    assert e3 < 3e-7 and e3 <= max(el, 2.0 ** -22)
    small = a * 1e-4
    assert err(f16_gemm(small, b), small, b) > 10 * e3                          # h1 of 1e-4-sized values is subnormal: the range limit
    scaled = f16_gemm(small * 2.0 ** 13, b) * 2.0 ** -13                         # a power-of-two scale restores it
    assert err(scaled, small, b) < 3e-7
