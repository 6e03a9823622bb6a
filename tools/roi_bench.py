"""Micro-benchmark of the RoIAlign kernels on box distributions of the training loop."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from point_teacher_amd import functional as F

dev = torch.device('cuda:0')
torch.manual_seed(0)
B, C, H, W = 2, 256, 100, 100
feat = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)


def bags(n_gt, wh_fn, U=25, coarse=None):
    c = torch.rand(B * n_gt, 2, device=dev) * 700 + 50
    wh = wh_fn(B * n_gt)
    base = torch.cat([c - wh / 2, c + wh / 2], 1)
    if coarse is not None:          # the 100 % config: 9 coarse boxes per object, 45 shaken boxes per coarse box
        base, _ = F.fine_proposals(base, coarse, None, 0, (800, 800))
        props, _ = F.fine_proposals(base, [1.0, 1.3, 0.7], [0.1], 4, (800, 800))
        U = 9 * 45
    else:
        props, _ = F.fine_proposals(base, [1.0, 1.2, 1.3, 0.8, 0.7], None, 4, (800, 800))
    bi = torch.arange(B, device=dev).repeat_interleave(n_gt * U).float()[:, None]
    return torch.cat([bi, props], 1)


cases = {
    'step2 bags (12px objs)': (bags(100, lambda n: torch.exp(torch.randn(n, 2, device=dev) * 0.5 + math.log(12.)).clamp(2, 64)), 25),
    'step1 syn bags (hulls)': (bags(100, lambda n: (torch.rand(n, 2, device=dev) * 2 + 0.5) * torch.tensor([25., 55.], device=dev)), 25),
    '100 % cfg bags (U1 9 x U2 45)': (bags(75, lambda n: torch.exp(torch.randn(n, 2, device=dev) * 0.5 + math.log(12.)).clamp(2, 64),
                                           coarse=[1.0, 1.3, 0.8]), 45),
    'negatives (<=100px)': (torch.cat([torch.arange(B, device=dev).repeat_interleave(200).float()[:, None],
                                       (lambda x1: torch.cat([x1, x1 + torch.rand(400, 2, device=dev) * 100], 1))(torch.rand(400, 2, device=dev) * 640)], 1), 1),
}
for name, (rois, group) in cases.items():
    f = feat.clone().requires_grad_(True)
    out = F.roi_align(f, rois, 7, 0.125, 0, True, group)
    g = torch.randn_like(out)
    for _ in range(3):
        out = F.roi_align(f, rois, 7, 0.125, 0, True, group)
        out.backward(g)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    reps = 10
    for _ in range(reps):
        ev[0].record()
        out = F.roi_align(f, rois, 7, 0.125, 0, True, group)
        ev[1].record()
        out.backward(g)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    K = rois.shape[0]
    mb = K * C * 49 * 4 / 1e6
    print(f'{name:26s} K={K:5d} fwd {tf/reps*1e3:8.1f} us ({mb/(tf/reps)/1e3:6.2f} TB/s)  bwd(incl. zero+layout) {tb/reps*1e3:8.1f} us ({mb/(tb/reps)/1e3:6.2f} TB/s)')

# ---- oriented bags (config 5: 1200x1200 -> 150x150 map, out 7, sample_num 2, clockwise) ----
H2 = W2 = 150
feat2 = torch.randn(B, C, H2, W2, device=dev).contiguous(memory_format=torch.channels_last)
n_gt = 100
c = torch.rand(B * n_gt, 2, device=dev) * 1100 + 50
wh = torch.exp(torch.randn(B * n_gt, 2, device=dev) * 0.5 + math.log(12.)).clamp(2, 64)
base = torch.cat([c - wh / 2, c + wh / 2], 1)
props, _ = F.fine_proposals(base, [1.0, 1.2, 1.3, 0.8, 0.6], None, 4, (1200, 1200))
ang = (torch.rand(B * n_gt, device=dev) * math.pi - math.pi / 2).repeat_interleave(25)
bi = torch.arange(B, device=dev).repeat_interleave(n_gt * 25).float()
rr = torch.stack([bi, (props[:, 0] + props[:, 2]) / 2, (props[:, 1] + props[:, 3]) / 2, props[:, 2] - props[:, 0],
                  props[:, 3] - props[:, 1], ang], 1)
f = feat2.clone().requires_grad_(True)
out = F.roi_align_rotated(f, rr, 7, 0.125, 2, True, True)
g = torch.randn_like(out)
for _ in range(3):
    out = F.roi_align_rotated(f, rr, 7, 0.125, 2, True, True)
    out.backward(g)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(10):
    ev[0].record()
    out = F.roi_align_rotated(f, rr, 7, 0.125, 2, True, True)
    ev[1].record()
    out.backward(g)
    ev[2].record()
    torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
K = rr.shape[0]
mb = K * C * 49 * 4 / 1e6
print(f'{"oriented bags (12px objs)":26s} K={K:5d} fwd {tf/10*1e3:8.1f} us ({mb/(tf/10)/1e3:6.2f} TB/s)  bwd(incl. zero+layout) {tb/10*1e3:8.1f} us ({mb/(tb/10)/1e3:6.2f} TB/s)')
