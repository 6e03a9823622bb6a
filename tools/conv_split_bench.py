"""bf16x6 implicit-GEMM 3x3 convolution vs the library (MIOpen) on the dense head's tower shape, forward, input gradient and weight gradient:
time (HIP events, median of 20).  python tools/conv_split_bench.py > profiles/r03/conv_bf16x6_vs_library.txt"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_teacher_amd import functional as F  # noqa: E402

torch.backends.cudnn.benchmark = True


def timed(fn, n=20):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


dev = torch.device('cuda:0')
print('shape | library fwd ms (TF) | bf16x6 fwd ms incl. split (TF) [split alone] | library dgrad ms | bf16x6 dgrad ms incl. split | library wgrad ms | bf16x6 wgrad ms (TF)')
for B in (2, 4, 6):
    H = W = 100
    C = O = 256
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(C, O, 3, padding=1).to(dev).to(memory_format=torch.channels_last)
    gy = torch.randn(B, O, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * B * H * W * O * 9 * C
    t_lib = timed(lambda: torch.nn.functional.conv2d(x, conv.weight, conv.bias, padding=1))
    t_mine = timed(lambda: F.split_conv3x3(x, conv.weight, conv.bias, False))
    rows = x.permute(0, 2, 3, 1).reshape(-1, C)
    n = (rows.shape[0] + 1) * C
    xp = torch.empty((3, n), dtype=torch.bfloat16, device=dev)
    t_split = timed(lambda: F.hip.call('pt_split_bf16x3_rows', rows, C, rows.shape[0], C, None, None, None, xp, n))
    t_lib_d = timed(lambda: torch.ops.aten.convolution_backward(gy, x, conv.weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    wp = F._conv_weight_planes(conv.weight, True)
    grows = gy.permute(0, 2, 3, 1).reshape(-1, O)
    t_mine_d = timed(lambda: F._conv3x3_from_planes(F._split_rows(grows), grows.shape[0], O, B, H, W, wp, None, False))
    gp, xq = F._split_rows(grows), F._split_rows(rows)
    t_mine_w = timed(lambda: F._conv3x3_wgrad(gp, xq, B, H, W, C, O))
    t_lib_w = timed(lambda: torch.ops.aten.convolution_backward(gy, x, conv.weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    print(f'B={B} 100x100 256->256 | {t_lib:.3f} ({fl / t_lib / 1e9:.0f}) | {t_mine:.3f} ({fl / t_mine / 1e9:.0f}) [{t_split:.3f}] | {t_lib_d:.3f} | {t_mine_d:.3f} | library wgrad {t_lib_w:.3f} | bf16x6 wgrad (planes reused) {t_mine_w:.3f} ({fl / t_mine_w / 1e9:.0f})', flush=True)
