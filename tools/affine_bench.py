"""Microbenchmark of the frozen-BN epilogue kernels on the activation shapes of the R50 trunk (B = 6, 800x800)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_teacher_amd import functional as F
dev = 'cuda'
shapes = [(6, 64, 400, 400), (6, 256, 200, 200), (6, 64, 200, 200), (6, 512, 100, 100), (6, 128, 100, 100), (6, 1024, 50, 50),
          (6, 2048, 25, 25)]
for cl in (True, False):
    for shp in shapes:
        B, C, H, W = shp
        x = torch.randn(shp, device=dev)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        res = torch.randn_like(x)
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        for with_res in (False, True):
            xs = [x.clone() for _ in range(12)]
            for t in xs[:2]:
                F.affine_relu_(t, sc, sh, res if with_res else None, True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for t in xs[2:]:
                F.affine_relu_(t, sc, sh, res if with_res else None, True)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            byt = x.numel() * 4 * (2 + with_res)
            print(f'cl={int(cl)} {str(shp):24s} res={int(with_res)} {us:8.1f} us  {byt / us / 1e6:6.2f} TB/s')
