#!/usr/bin/env python3
"""1x1 stride-1 convolutions of the ResNet-50 bottlenecks / FPN laterals in NHWC: torch.conv2d (MIOpen) against the same product
written as a GEMM on the [B*H*W, C] view (hipBLASLt / rocBLAS, optionally TunableOp-tuned).  fwd and fwd+bwd, fp32 and bf16.
    python tools/conv1x1_bench.py [--batch 6] [--tune 1]"""
import argparse
import os
import sys
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=6)
ap.add_argument('--tune', type=int, default=0)
args = ap.parse_args()
if args.tune:
    os.environ.update(PYTORCH_TUNABLEOP_ENABLED='1', PYTORCH_TUNABLEOP_FILENAME='gpurun_out/tune1x1.csv')
import torch
import torch.nn.functional as F


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


torch.backends.cudnn.benchmark = True
SHAPES = [(200, 64, 256), (200, 256, 64), (100, 128, 512), (100, 512, 128), (50, 256, 1024), (50, 1024, 256), (25, 512, 2048), (25, 2048, 512),
          (100, 512, 256), (50, 1024, 256), (25, 2048, 256)]
B = args.batch
for dt in (torch.float32, torch.bfloat16):
    for hw, ci, co in SHAPES:
        x = torch.randn(B, ci, hw, hw, device='cuda', dtype=dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = (torch.randn(co, ci, 1, 1, device='cuda', dtype=dt) * 0.05).requires_grad_(True)
        xm = x.detach().permute(0, 2, 3, 1).reshape(-1, ci).requires_grad_(True)
        wm = w.detach().view(co, ci).requires_grad_(True)
        yc = F.conv2d(x, w)
        ym = F.linear(xm, wm)
        err = float((yc.permute(0, 2, 3, 1).reshape(-1, co).float() - ym.float()).abs().max() / yc.float().abs().max())
        g, gm = torch.randn_like(yc), torch.randn_like(ym)
        fl = 2.0 * B * hw * hw * ci * co
        t = [timed(lambda: F.conv2d(x, w)), timed(lambda: F.linear(xm, wm)),
             timed(lambda: torch.autograd.grad(yc, (x, w), g, retain_graph=True)), timed(lambda: torch.autograd.grad(ym, (xm, wm), gm, retain_graph=True))]
        print(f'{str(dt):15s} {hw:3d}^2 {ci:4d}->{co:<4d} fwd conv {t[0]:7.1f} us {fl / t[0] / 1e6:6.1f} TF | gemm {t[1]:7.1f} us {fl / t[1] / 1e6:6.1f} TF || '
              f'bwd conv {t[2]:7.1f} us {2 * fl / t[2] / 1e6:6.1f} TF | gemm {t[3]:7.1f} us {2 * fl / t[3] / 1e6:6.1f} TF   relerr {err:.1e}', flush=True)
