"""EXPERIMENT: the GEMM from fp16 x 2 operands / three products (pt_gemm_f16x3_nt) against the shipped
bf16 x 3 operands / six products (pt_gemm_bf16x6_nt) and the fp32 library: time (median of 20, HIP events) and error against float64
in units of sum |a||b|, for operand magnitudes inside and outside fp16's comfortable range.
usage: python tools/gemm_f16_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_teacher_amd import functional as F, hip  # noqa: E402

dev = torch.device('cuda:0')


def split_f16(x):
    R, C = x.shape
    n = ((R + 15) // 16) * ((C + 31) // 32) * 512
    planes = torch.empty((2, n), dtype=torch.float16, device=x.device)
    hip.call('pt_split_f16x2', x, x.stride(0), R, C, planes.view(torch.bfloat16), n)      # (uint16_t* in the header = 16-bit storage)
    return planes


def gemm_f16(ap, bp, M, N, K, out):
    hip.call('pt_gemm_f16x3_nt', ap.view(torch.bfloat16), ap.shape[1], bp.view(torch.bfloat16), bp.shape[1], out, out.stride(0), None, M, N, K, 0, 0)
    return out


def med(fn, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


for (M, N, K) in ((5000, 1024, 12544), (5000, 1024, 1024), (1024, 12544, 5000), (20000, 256, 2304)):
    for (sa, sb, tag) in ((1.0, 0.05, 'activations ~ 1, weights ~ 0.05'), (1e-4, 0.05, 'gradient-like a ~ 1e-4'),
                          (1e-4 * 2 ** 13, 0.05 * 2 ** 4, 'the same, scaled by 2^13 / 2^4 before the split')):
        g = torch.Generator().manual_seed(0)
        a = (torch.randn(M, K, generator=g) * sa).to(dev)
        b = (torch.randn(N, K, generator=g) * sb).to(dev)
        rows = slice(0, 256)                               # float64 reference on a slab of rows
        ref = a[rows].double() @ b.double().t()
        mag = a[rows].double().abs() @ b.double().abs().t()
        out = torch.empty(M, N, device=dev)
        ap3, bp3 = F.split_bf16x3(a), F.split_bf16x3(b)
        ap2, bp2 = split_f16(a), split_f16(b)
        t6 = med(lambda: F.gemm_bf16x6_nt(ap3, bp3, out=out))
        e6 = float(((out[rows].double() - ref).abs() / mag).max())
        t3 = med(lambda: gemm_f16(ap2, bp2, M, N, K, out))
        e3 = float(((out[rows].double() - ref).abs() / mag).max())
        tl = med(lambda: torch.matmul(a, b.t(), out=out))
        el = float(((out[rows].double() - ref).abs() / mag).max())
        fl = 2.0 * M * N * K
        print(f'{M:6d} x {N:5d} x {K:5d}  {tag:48s} bf16x6 {t6 * 1e3:7.1f} us {fl / t6 / 1e9:6.1f} TF err {e6:.2e} | '
              f'f16x3 {t3 * 1e3:7.1f} us {fl / t3 / 1e9:6.1f} TF err {e3:.2e} | fp32 library {tl * 1e3:7.1f} us {fl / tl / 1e9:6.1f} TF err {el:.2e}', flush=True)
