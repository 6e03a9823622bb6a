"""Geometry and time of every axis-aligned RoIAlign call of one HBB iteration (diagnostics for csrc/roi_align.hip):
python tools/roi_stats.py [step1|step2].  For the backward it also prints, per run length, how many footprint pixels the
launch adds atomically today (per RoI) and how many a run-level union would add."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import point_teacher_amd as pta  # noqa: E402
from point_teacher_amd import hip  # noqa: E402
import point_teacher_amd.functional as PF  # noqa: E402
from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else 'step1'
dev = torch.device('cuda:0')
torch.manual_seed(1234)
cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
cfg.model['burn_in_step'] = 10 ** 9 if wl == 'step1' else -1
model = pta.build_detector(cfg.model).to(dev)
benchmark_init_(model, phase2=(wl == 'step2'))
model.train()
model.teacher_stream = False      # per-launch HIP-event timings: no co-scheduled teacher kernels on a second stream
tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
data = SyntheticTiles(n=8, size=800, mean_objects=300, seed=7, device=dev)
for it in range(3):
    tr.step(data.batch(it, 2))
rec = []
orig = hip.call


def spy(fn, *a):
    if fn in ('pt_roi_align_fwd', 'pt_roi_align_bwd'):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(fn, *a); e1.record()
        # fwd: (feat, rois, B, C, H, W, K, out, scale, sr, aligned, cl, group, out)   bwd: (g, rois, B, C, H, W, K, out, scale, ...)
        rec.append((fn, a[1].clone(), float(a[8]), int(a[12]), e0, e1, a))
        return r
    return orig(fn, *a)


hip.call = spy
PF.hip.call = spy
tr.step(data.batch(3, 2))
torch.cuda.synchronize()


def extent(lo, hi, L=100):
    """[first, last] feature pixel an aligned RoIAlign axis touches (taps at lo + (i + .5) * bin / grid, adaptive grid)."""
    size = hi - lo
    grid = np.maximum(np.ceil(size / 7.0), 1.0)
    first = lo + 0.5 * size / 7.0 / grid
    last = hi - 0.5 * size / 7.0 / grid
    a = np.clip(np.floor(np.maximum(first, 0.0)), 0, L - 1)
    b = np.clip(np.floor(np.maximum(last, 0.0)) + 1, 0, L - 1)
    return a.astype(np.int64), b.astype(np.int64)


for fn, rois, scale, group, e0, e1, args in rec:
    r = rois.cpu().numpy().astype(np.float64)
    K = r.shape[0]
    x0, x1 = extent(r[:, 1] * scale - 0.5, r[:, 3] * scale - 0.5)
    y0, y1 = extent(r[:, 2] * scale - 0.5, r[:, 4] * scale - 0.5)
    nx, ny = x1 - x0 + 1, y1 - y0 + 1
    F = nx * ny
    line = (f'{fn[-3:]} K={K:6d} group={group:3d} {e0.elapsed_time(e1) * 1e3:8.1f} us  footprint px median {np.median(F):.0f} '
            f'p90 {np.quantile(F, .9):.0f} max {F.max()}  side median {np.median(np.maximum(nx, ny)):.0f} p90 '
            f'{np.quantile(np.maximum(nx, ny), .9):.0f} max {np.maximum(nx, ny).max()}')
    print(line)
    if fn.endswith('bwd'):
        side1 = np.maximum(nx, ny)
        print('      share of the per-RoI pixel adds by RoI side: ' + ' '.join(
            f'<= {b}: {F[side1 <= b].sum() / max(F.sum(), 1) * 100:.0f}%' for b in (5, 8, 12, 16, 24, 32, 48, 64, 104)))
        for gs in (1, 5, 9, 15, 25, 45):
            if group % gs or K % gs:
                continue
            ux = x1.reshape(-1, gs).max(1) - x0.reshape(-1, gs).min(1) + 1
            uy = y1.reshape(-1, gs).max(1) - y0.reshape(-1, gs).min(1) + 1
            side = np.maximum(ux, uy)
            fits = {s: float((side <= s).mean()) for s in (5, 8, 10, 12, 16)}
            print(f'      run {gs:2d}: per-RoI pixel adds {F.sum():8d}  run-union adds {(ux * uy).sum():8d}  union side median '
                  f'{np.median(side):.0f} p90 {np.quantile(side, .9):.0f} max {side.max()}  runs with side <= s: '
                  + ' '.join(f'{s}:{v * 100:.0f}%' for s, v in fits.items()))


# ---- replay of the heaviest backward call with classes of RoIs replaced by a 1-pixel box at the bag's centre: which class costs the time
heavy = max((r for r in rec if r[0].endswith('bwd')), key=lambda r: r[4].elapsed_time(r[5]))
fn, rois, scale, group, _, _, args = heavy
r = rois.cpu().numpy().astype(np.float64)
x0, x1 = extent(r[:, 1] * scale - 0.5, r[:, 3] * scale - 0.5)
y0, y1 = extent(r[:, 2] * scale - 0.5, r[:, 4] * scale - 0.5)
side = torch.from_numpy(np.maximum(x1 - x0 + 1, y1 - y0 + 1)).to(dev)
cx, cy = (rois[:, 1] + rois[:, 3]) / 2, (rois[:, 2] + rois[:, 4]) / 2
tiny = torch.stack([rois[:, 0], cx - 2, cy - 2, cx + 2, cy + 2], 1)


def replay(keep, label):
    rr = torch.where(keep[:, None], rois, tiny).contiguous()
    a = list(args)
    a[1] = rr
    a[-1] = torch.zeros_like(args[-1])
    for _ in range(3):
        orig(fn, *a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        orig(fn, *a)
    e1.record()
    torch.cuda.synchronize()
    print(f'replay {label:34s} kept {int(keep.sum()):5d} RoIs  {e0.elapsed_time(e1) * 100:8.1f} us')


replay(side > 0, 'all')
replay(side <= 5, 'side <= 5 only')
replay(side <= 16, 'side <= 16 only')
replay(side <= 32, 'side <= 32 only')
replay(side <= 48, 'side <= 48 only')
replay(side > 48, 'side > 48 only')
replay((side > 16) & (side <= 48), '16 < side <= 48 only')
