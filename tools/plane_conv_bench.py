"""Per-shape timing of the plane-native convolution kernels on the shapes of the R50-FPN-PSAGG trunk (bs 2 -> student batch 6 in
phase 1, teacher batch 2): forward (with the epilogue the network uses), input gradient, weight gradient; HIP events, median of 20,
fp32-equivalent TFLOP/s (2 M N K / time) and algorithmic GB/s.  usage: python tools/plane_conv_bench.py [B=6] [np1]
(np1: bf16 operands, one plane - the trunk of BASELINE configs[2])"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from point_teacher_amd import functional as F, planes as PL  # noqa: E402

dev = 'cuda:0'
NP = 1 if 'np1' in sys.argv[1:] else 3


def timeit(fn, n=20, reps=10):
    """GPU time of one call: `reps` calls captured into a HIP graph (the host cost of a call - descriptor, allocations - is tens of
    microseconds, more than the small kernels take), median over n replays."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)
    ts.sort()
    return ts[len(ts) // 2]


def planes_of(B, H, W, C, relu=True):
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    if relu:
        x = torch.relu(x)
    if NP == 1:
        x = x.to(torch.bfloat16)
    return PL.split_nhwc(x)[0], x


def bench(name, B, H, W, Cin, Cout, K, stride=1, res=None):
    pad = (K - 1) // 2
    Ho, Wo = PL.out_hw(H, W, K, stride, pad)
    M = B * Ho * Wo
    xt, x = planes_of(B, H, W, Cin)
    w = (torch.randn(Cout, Cin, K, K, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    wp, wd = F._conv_weight_planes(w, False, None, NP), F._conv_weight_planes(w, True, sc, NP)
    rt = planes_of(B, Ho, Wo, Cout)[0] if res == 'planes' else None
    gt, _ = planes_of(B, Ho, Wo, Cout, relu=False)
    flops = 2.0 * M * Cin * Cout * K * K
    t_f = timeit(lambda: PL.launch_conv(xt, B, H, W, Cin, wp, Cout, K, stride, pad, scale=sc, shift=sh, res_planes=rt, relu=True, want_planes=True))
    by_f = (M * Cin * 2 * NP + M * Cout * 2 * NP * (2 if rt is not None else 1))
    lib = torch.nn.functional.conv2d
    wl = w.to(x.dtype)
    t_l = timeit(lambda: lib(x, wl, None, stride, pad))
    line = f'{name:34s} M={M:6d} {Cin:4d}->{Cout:4d} k{K} s{stride}  fwd {t_f:7.1f} us {flops / t_f / 1e6:6.1f} TF {by_f / t_f / 1e3:6.0f} GB/s (lib {t_l:6.1f} us)'
    if stride == 1:
        t_d = timeit(lambda: PL.launch_conv(gt, B, Ho, Wo, Cout, wd, Cin, K, 1, K - 1 - pad, mask_planes=xt, want_planes=True))
        line += f' | dgrad {t_d:7.1f} us {flops / t_d / 1e6:6.1f} TF'
    if Cin % 128 == 0 and Cout % 128 == 0:
        t_w = timeit(lambda: PL.launch_wgrad(gt, xt, B, H, W, Cin, Cout, K, stride, pad, row_scale=sc))
        line += f' | wgrad {t_w:7.1f} us {flops / t_w / 1e6:6.1f} TF'
    print(line, flush=True)


B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
print(f'batch {B}')
for (nm, H, Cin, p) in (('layer2', 100, 512, 128), ('layer3', 50, 1024, 256), ('layer4', 25, 2048, 512)):
    bench(f'{nm} conv1 1x1', B, H, H, Cin, p, 1)
    bench(f'{nm} conv2 3x3', B, H, H, p, p, 3)
    bench(f'{nm} conv3 1x1 + identity', B, H, H, p, Cin, 1, res='planes')
bench('layer2.0 conv1 (compact)', B, 100, 100, 256, 128, 1)
bench('layer2.0 downsample', B, 100, 100, 256, 512, 1)
bench('layer3.0 conv1 s2', B, 100, 100, 512, 256, 1, 2)
bench('layer3.0 downsample s2', B, 100, 100, 512, 1024, 1, 2)
bench('layer4.0 conv1 s2', B, 50, 50, 1024, 512, 1, 2)
bench('layer4.0 downsample s2', B, 50, 50, 1024, 2048, 1, 2)
for (nm, H, Cin) in (('lateral3', 100, 512), ('lateral4', 50, 1024), ('lateral5', 25, 2048)):
    bench(nm, B, H, H, Cin, 256, 1)
for H in (100, 50, 25):
    bench(f'fpn 3x3 @{H}', B, H, H, 256, 256, 3)
    bench(f'psagg 1x1 @{H}', B, H, H, 256, 256, 1)
