"""bf16x6 split GEMM vs the fp32 library GEMM on the MIL FC shapes (round-2 verdict item 2): time (HIP events, median of 20) and
error against float64.  python tools/gemm_bench.py > profiles/r03/gemm_bf16x6_vs_library.txt"""
import os
import sys
import statistics

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_teacher_amd import functional as F, hip  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


def main():
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    print('shape (M x N x K, out = A[M,K] B[N,K]^T) | library fp32 ms TF | bf16x6 gemm ms TF-equivalent | split A ms | split B ms | tile rows | err vs fp64 (library / bf16x6, units of sum|a||b|)')
    shapes = [('FC1 forward  K=5000', 5000, 1024, 12544), ('FC1 dgrad    K=5000', 5000, 12544, 1024), ('FC1 wgrad    K=5000', 1024, 12544, 5000),
              ('FC2 forward  K=5000', 5000, 1024, 1024), ('FC1 forward  K=400', 400, 1024, 12544),
              ('FC1 forward  K=60750', 60750, 1024, 12544), ('FC1 dgrad    K=60750', 60750, 12544, 1024), ('FC1 wgrad    K=60750', 1024, 12544, 60750)]
    if '--quick' in sys.argv:
        shapes = shapes[:3]
    for name, M, N, K in shapes:
        a = torch.randn(M, K, generator=g).to(dev)
        b = (torch.randn(N, K, generator=g) * 0.05).to(dev)
        flops = 2.0 * M * N * K
        t_lib = timed(lambda: a @ b.t())
        ap, bp = F.split_bf16x3(a), F.split_bf16x3(b)
        out = torch.empty(M, N, device=dev)
        t_g = timed(lambda: F.gemm_bf16x6_nt(ap, bp, out=out))
        t_sa = timed(lambda: F.split_bf16x3(a))
        t_sb = timed(lambda: F.split_bf16x3(b))
        rows = hip.call('pt_gemm_bf16x6_tile_rows', M, N)
        best = None
        for r in (96, 128, 160, 192, 224, 256):
            t = timed(lambda: F.gemm_bf16x6_nt(ap, bp, out=out, tile_rows=r), n=5)
            best = (t, r) if best is None or t < best[0] else best
        err = ''
        if M * N * K < 8e10:
            sub = slice(0, min(M, 512))
            ref = a[sub].double() @ b.double().t()
            sc = a[sub].double().abs() @ b.double().abs().t()
            err = f'{float(((a[sub] @ b.t()).double() - ref).abs().div(sc).max()):.2e} / {float((out[sub].double() - ref).abs().div(sc).max()):.2e}'
        print(f'{name}: {M}x{N}x{K} | {t_lib:.3f} {flops / t_lib / 1e9:.0f} | {t_g:.3f} {flops / t_g / 1e9:.0f} | {t_sa:.3f} | {t_sb:.3f} | auto {rows} (best {best[1]}: {best[0]:.3f}) | {err}', flush=True)
        del a, b, ap, bp, out


if __name__ == '__main__':
    main()
