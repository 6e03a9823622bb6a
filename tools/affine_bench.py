"""Microbenchmark of the frozen-BN epilogue kernels on the activation shapes of the R50 trunk (B = 6, 800x800)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_teacher_amd import functional as F
dev = 'cuda'
shapes = [(6, 64, 400, 400), (6, 256, 200, 200), (6, 64, 200, 200), (6, 512, 100, 100), (6, 128, 100, 100), (6, 1024, 50, 50)]
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

print('--- trainable-affine epilogue (config 5 shapes, B = 4, 1200x1200) ---')
import torch.nn as nn
from point_teacher_amd import nn_modules
for shp in [(4, 128, 150, 150), (4, 512, 150, 150), (4, 256, 75, 75), (4, 1024, 75, 75), (4, 512, 38, 38), (4, 2048, 38, 38)]:
    B, C, H, W = shp
    bn = nn.BatchNorm2d(C).to(dev).eval()
    for with_res in (False, True):
        x = torch.randn(shp, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        res = torch.randn(shp, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        g = torch.randn(shp, device=dev).contiguous(memory_format=torch.channels_last)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for it in range(12):
            ev[0].record()
            y = F.bn_eval_relu(x, bn, res if with_res else None, True)
            ev[1].record()
            y.backward(g)
            ev[2].record()
            torch.cuda.synchronize()
            if it >= 2:
                tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
        n = x.numel() * 4
        print(f'{str(shp):22s} res={int(with_res)} fwd {tf/10*1e3:7.1f} us ({n*(2+with_res)/(tf/10)/1e9:5.2f} TB/s)  '
              f'bwd {tb/10*1e3:7.1f} us ({n*(4+with_res)/(tb/10)/1e9:5.2f} TB/s)')
