#!/usr/bin/env python3
"""Modulated deformable convolution (DCNv2) at the FCOS tower shape (3x3, 256 -> 256, 100x100, B = 2, deform_groups 1): the
NHWC path (pt_deform_*_cl + GEMM), the NCHW path and a plain convolution of the same shape, forward and forward + backward;
per-kernel HIP-event times of the hand-written halves with their algorithmic bytes.
    python tools/dcn_bench.py [--batch 2] [--hw 100]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_teacher_amd import functional as F, hip


def timed(fn, n=20, warm=5):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--hw', type=int, default=100)
    args = ap.parse_args()
    B, C, O, H = args.batch, 256, 256, args.hw
    dev = 'cuda'
    torch.manual_seed(0)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    x, w = torch.randn(B, C, H, H, device=dev), torch.randn(O, C, 3, 3, device=dev) * 0.02
    off, mask = torch.randn(B, 18, H, H, device=dev), torch.rand(B, 9, H, H, device=dev)
    bias = torch.zeros(O, device=dev)
    flops = 2.0 * B * H * H * C * O * 9
    for name, fmt in (('nhwc', cl), ('nchw', lambda t: t)):
        xs = [fmt(t).requires_grad_(True) for t in (x, off, mask, w)]
        y = F.modulated_deform_conv2d(*xs, bias, 1, 1, 1, 1, 1)
        g = fmt(torch.randn_like(y))
        tf = timed(lambda: F.modulated_deform_conv2d(*xs, bias, 1, 1, 1, 1, 1))
        tb = timed(lambda: torch.autograd.grad(F.modulated_deform_conv2d(*xs, bias, 1, 1, 1, 1, 1), xs, g))
        print(f'DCNv2 {name}: fwd {tf:7.1f} us ({flops / tf / 1e6:6.1f} TFLOP/s of the contraction)   fwd+bwd {tb:7.1f} us')
    xc, wc = cl(x).requires_grad_(True), cl(w).requires_grad_(True)
    yc = torch.nn.functional.conv2d(xc, wc, bias, 1, 1)
    gc = torch.randn_like(yc)
    tf = timed(lambda: torch.nn.functional.conv2d(xc, wc, bias, 1, 1))
    tb = timed(lambda: torch.autograd.grad(torch.nn.functional.conv2d(xc, wc, bias, 1, 1), (xc, wc), gc))
    print(f'plain conv (MIOpen, nhwc): fwd {tf:7.1f} us ({flops / tf / 1e6:6.1f} TFLOP/s)   fwd+bwd {tb:7.1f} us')
    # the hand-written halves alone
    L = H * H
    col = torch.empty(B * L, 9 * C, device=dev)
    geo = (B, C, H, H, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    xn, on, mn = (cl(t).permute(0, 2, 3, 1) for t in (x, off, mask))
    t1 = timed(lambda: hip.call('pt_deform_im2col_cl', xn, on, mn, *geo, col))
    by1 = col.numel() * 4 + x.numel() * 4 + off.numel() * 4 + mask.numel() * 4
    print(f'pt_deform_im2col_cl: {t1:6.1f} us, {by1 / 1e6:.0f} MB algorithmic (col written once, x / offsets / mask read once) -> {by1 / t1 / 1e6:.2f} TB/s = {by1 / t1 / 8e6:.2f} of the HBM peak')
    gcol = torch.randn_like(col)
    gx, go, gm = torch.zeros_like(cl(x)), torch.empty_like(cl(off)), torch.empty_like(cl(mask))
    t2 = timed(lambda: hip.call('pt_deform_col2im_cl', gcol, xn, on, mn, *geo, gx.permute(0, 2, 3, 1), go.permute(0, 2, 3, 1), gm.permute(0, 2, 3, 1)))
    by2 = gcol.numel() * 4 + 2 * x.numel() * 4 + 2 * (off.numel() + mask.numel()) * 4
    print(f'pt_deform_col2im_cl: {t2:6.1f} us, {by2 / 1e6:.0f} MB algorithmic (grad_col read once, x read, grad_x read-modify-write) -> {by2 / t2 / 1e6:.2f} TB/s = {by2 / t2 / 8e6:.2f} of the HBM peak')
    coln = torch.empty(B, C * 9, L, device=dev)
    t3 = timed(lambda: hip.call('pt_deform_im2col', x, off, mask, *geo, coln))
    print(f'pt_deform_im2col (NCHW): {t3:6.1f} us')


if __name__ == '__main__':
    main()
