#!/usr/bin/env python3
"""What the vendor convolution (torch -> MIOpen / CK) costs on the FCOS tower shape (3x3, 256 -> 256, stride 1, NHWC) per pyramid level:
forward, and backward (dgrad + wgrad), fp32 and bf16.  The bar for a hand-written kernel.
    python tools/conv_bench.py [--batch 2] [--size 800]"""
import argparse
import torch
import torch.nn.functional as F


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
    return a.elapsed_time(b) / n * 1e3            # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--size', type=int, default=800)
    ap.add_argument('--cin', type=int, default=256)
    ap.add_argument('--cout', type=int, default=256)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    levels = [-(-args.size // s) for s in (8, 16, 32, 64, 128)]
    for dt in (torch.float32, torch.bfloat16):
        tot_f = tot_b = tot_fl = 0.0
        for hw in levels + ['canvas']:
            h, w = (hw, hw) if hw != 'canvas' else (levels[0], levels[0] + 1 + levels[1])
            x = torch.randn(args.batch, args.cin, h, w, device='cuda', dtype=dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            wt = torch.randn(args.cout, args.cin, 3, 3, device='cuda', dtype=dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y = F.conv2d(x, wt, None, 1, 1)
            g = torch.randn_like(y)
            tf = timed(lambda: F.conv2d(x, wt, None, 1, 1))
            tb = timed(lambda: torch.autograd.grad(y, (x, wt), g, retain_graph=True))
            fl = 2.0 * args.batch * h * w * args.cin * args.cout * 9
            print(f'{str(dt):15s} B {args.batch} {h:4d}x{w:<4d} fwd {tf:8.1f} us {fl / tf / 1e6:7.1f} TF/s   bwd(dgrad+wgrad) {tb:8.1f} us {2 * fl / tb / 1e6:7.1f} TF/s')
            if hw != 'canvas':
                tot_f, tot_b, tot_fl = tot_f + tf, tot_b + tb, tot_fl + fl
        print(f'{str(dt):15s} 5 levels: fwd {tot_f:8.1f} us {tot_fl / tot_f / 1e6:7.1f} TF/s   bwd {tot_b:8.1f} us {2 * tot_fl / tot_b / 1e6:7.1f} TF/s')


if __name__ == '__main__':
    main()
