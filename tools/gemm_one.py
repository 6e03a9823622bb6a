"""One GEMM shape a few times (for rocprofv3 --pmc passes): python tools/gemm_one.py M N K [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_teacher_amd import functional as F  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).to(dev)
b = (torch.randn(N, K, generator=g) * 0.05).to(dev)
ap, bp = F.split_bf16x3(a), F.split_bf16x3(b)
out = torch.empty(M, N, device=dev)
for _ in range(reps):
    F.gemm_bf16x6_nt(ap, bp, out=out)
torch.cuda.synchronize()
