"""Tile-height sweep of pt_conv_bf16x6 on the byte-bound 1x1 shapes of the trunk with H2 (scaled fp16 x 2) operands: forward with the
Bottleneck's epilogue (BatchNorm, identity planes, ReLU, H2 out) and the input gradient (mask, chain).  Buffers rotate over R
independent sets (> the 256 MB of Infinity Cache) so that every call reads cold bytes, as in the live iteration.
usage: python tools/h2_tile_sweep.py [b3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from point_teacher_amd import functional as F, planes as PL  # noqa: E402

dev = 'cuda:0'
H2 = 'b3' not in sys.argv[1:]
R = 4


def timeit(fns, n=12, reps=3):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for f in fns:
                f()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / (reps * len(fns)))
    ts.sort()
    return ts[len(ts) // 2]


def planes(B, H, W, C):
    x = torch.relu(torch.randn(B, C, H, W, device=dev)).contiguous(memory_format=torch.channels_last)
    return PL.split_nhwc(x, f16=H2)[0]


def sweep(name, B, H, W, Cin, Cout, res, mask, wgrad=False):
    M = B * H * W
    w = (torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    wp = F._conv_weight_planes(w, False, None, 2 if H2 else 3)
    xs = [planes(B, H, W, Cin) for _ in range(R)]
    rs = [planes(B, H, W, Cout) for _ in range(R)] if (res or mask) else [None] * R
    eb = 4 if H2 else 6
    byts = eb * (M * Cin + M * Cout * (1 + int(res))) + 2 * M * Cout * int(mask) + eb * Cin * Cout
    out = []
    for rows in (64, 96, 128, 160, 192, 256):
        fns = [(lambda x=x, r=r: PL.launch_conv(x, B, H, W, Cin, wp, Cout, 1, 1, 0, scale=sc, shift=sh, res_planes=r if res else None,
                                                mask_planes=r if mask else None, relu=not mask, want_planes=True, tile_rows=rows, splits=1))
               for x, r in zip(xs, rs)]
        us = timeit(fns)
        out.append((rows, us))
    best = min(out, key=lambda t: t[1])
    auto = timeit([(lambda x=x, r=r: PL.launch_conv(x, B, H, W, Cin, wp, Cout, 1, 1, 0, scale=sc, shift=sh, res_planes=r if res else None,
                                                    mask_planes=r if mask else None, relu=not mask, want_planes=True)) for x, r in zip(xs, rs)])
    print(f'{name:34s} M={M:6d} {Cin:5d}->{Cout:5d} {byts / 1e6:7.1f} MB  ' + ' '.join(f'{r}:{u:6.1f}' for r, u in out) +
          f'  | auto {auto:6.1f} us = {byts / auto / 1e6:5.2f} TB/s, best {best[0]} {best[1]:6.1f} us = {byts / best[1] / 1e6:5.2f} TB/s', flush=True)


print('operands:', 'H2 (fp16 x 2)' if H2 else 'B3 (bf16 x 3)')
for B in (() if ('k3' in sys.argv[1:] or 'b4' in sys.argv[1:]) else (6, 2)):
    sweep(f'layer2 conv3 +res (B={B})', B, 100, 100, 128, 512, True, False)
    sweep(f'layer2 conv3^T +mask', B, 100, 100, 512, 128, False, True)
    sweep(f'layer2 conv1 (B={B})', B, 100, 100, 512, 128, False, False)
    sweep(f'layer2 conv1^T +res+mask', B, 100, 100, 128, 512, True, True)
    sweep(f'layer3 conv3 +res (B={B})', B, 50, 50, 256, 1024, True, False)
    sweep(f'layer3 conv1 (B={B})', B, 50, 50, 1024, 256, False, False)
    sweep(f'layer3 conv1^T +res+mask', B, 50, 50, 256, 1024, True, True)
    sweep(f'layer4 conv3 +res (B={B})', B, 25, 25, 512, 2048, True, False)
    sweep(f'layer4 conv1 (B={B})', B, 25, 25, 2048, 512, False, False)


def sweep_k(name, B, H, W, Cin, Cout, K, mask=False, planes_out=True):
    """Compute-bound members: 3x3 convolutions / FC layers (rows of a 1x1 over RoIs)."""
    M = B * H * W
    w = (torch.randn(Cout, Cin, K, K, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    sh = torch.randn(Cout, device=dev)
    wp = F._conv_weight_planes(w, False, None, 2 if H2 else 3)
    xs = [planes(B, H, W, Cin) for _ in range(2)]
    ms = [planes(B, H, W, Cout) for _ in range(2)] if mask else [None] * 2
    out = []
    fl = 2.0 * M * Cin * Cout * K * K
    for rows in (64, 96, 128, 160, 192, 224, 256):
        fns = [(lambda x=x, m=m: PL.launch_conv(x, B, H, W, Cin, wp, Cout, K, 1, (K - 1) // 2, shift=sh, mask_planes=m, relu=not mask,
                                                want_planes=planes_out, want_f32=not planes_out, tile_rows=rows, splits=1)) for x, m in zip(xs, ms)]
        out.append((rows, timeit(fns, reps=2)))
    auto = timeit([(lambda x=x, m=m: PL.launch_conv(x, B, H, W, Cin, wp, Cout, K, 1, (K - 1) // 2, shift=sh, mask_planes=m, relu=not mask,
                                                    want_planes=planes_out, want_f32=not planes_out)) for x, m in zip(xs, ms)], reps=2)
    best = min(out, key=lambda t: t[1])
    print(f'{name:30s} M={M:6d} {Cin:5d}->{Cout:5d} k{K} ' + ' '.join(f'{r}:{u:6.1f}' for r, u in out) +
          f' | auto {auto:6.1f} us = {fl / auto / 1e6:5.1f} TF, best {best[0]} {best[1]:6.1f} us = {fl / best[1] / 1e6:5.1f} TF', flush=True)


if 'k3' in sys.argv[1:]:
    sweep_k('tower 3x3 (B=2)', 2, 100, 100, 256, 256, 3)
    sweep_k('tower 3x3 dgrad +mask (B=2)', 2, 100, 100, 256, 256, 3, mask=True)
    sweep_k('tower 3x3 (B=6)', 6, 100, 100, 256, 256, 3)
    sweep_k('layer2 3x3 (B=6)', 6, 100, 100, 128, 128, 3)
    sweep_k('layer3 3x3 (B=6)', 6, 50, 50, 256, 256, 3)
    sweep_k('layer3 3x3 (B=2)', 2, 50, 50, 256, 256, 3)
    sweep_k('layer4 3x3 (B=6)', 6, 25, 25, 512, 512, 3)
    sweep_k('FC1 5000 rows', 1, 5000, 1, 12544, 1024, 1)
    sweep_k('FC1 dgrad 5000 rows', 1, 5000, 1, 1024, 12544, 1, planes_out=False)
    sweep_k('FC2 5000 rows', 1, 5000, 1, 1024, 1024, 1)

if 'b4' in sys.argv[1:]:            # the phase-2 student batch (clean + augmented = 4 images)
    sweep('layer3 conv1 (B=4)', 4, 50, 50, 1024, 256, False, False)
    sweep('layer3 conv3^T +mask (B=4)', 4, 50, 50, 1024, 256, False, True)
    sweep('layer3 conv3 +res (B=4)', 4, 50, 50, 256, 1024, True, False)
    sweep('layer2 conv1 (B=4)', 4, 100, 100, 512, 128, False, False)
    sweep('layer2 conv3^T +mask (B=4)', 4, 100, 100, 512, 128, False, True)
    sweep('layer2 conv3 +res (B=4)', 4, 100, 100, 128, 512, True, False)
    sweep('layer4 conv1 (B=4)', 4, 25, 25, 2048, 512, False, False)
    sweep('layer4 conv3 +res (B=4)', 4, 25, 25, 512, 2048, True, False)
