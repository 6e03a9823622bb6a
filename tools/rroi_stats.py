"""Geometry and time of every RoIAlignRotated call of an oriented iteration (diagnostics for csrc/rotated.hip):
python tools/rroi_stats.py [step1|step2]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import point_teacher_amd as pta  # noqa: E402
from point_teacher_amd import hip  # noqa: E402
import point_teacher_amd.functional as PF  # noqa: E402
from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else 'step2'
dev = torch.device('cuda:0')
torch.manual_seed(1234)
cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'obb', 'point_teacher', 'sodaa_fcos_pointteacher_1x.py'))
cfg.model['burn_in_step'] = 10 ** 9 if wl == 'step1' else -1
model = pta.build_detector(cfg.model).to(dev)
benchmark_init_(model, phase2=(wl == 'step2'))
model.train()
model.teacher_stream = False      # per-launch HIP-event timings: no co-scheduled teacher kernels on a second stream
tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
data = SyntheticTiles(n=8, size=1200, mean_objects=300, seed=7, device=dev, oriented=True, num_classes=9)
WARM = int(sys.argv[2]) if len(sys.argv) > 2 else 3        # iterations before the recorded one (the bench times iterations 5 .. 25)
for it in range(WARM):
    tr.step(data.batch(it, 2))
rec = []
orig = hip.call


def spy(fn, *a):
    if fn in ('pt_roi_align_rotated_fwd', 'pt_roi_align_rotated_bwd'):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(fn, *a); e1.record()
        rec.append((fn, a[1].clone(), a[8], e0, e1))
        return r
    return orig(fn, *a)


hip.call = spy
PF.hip.call = spy
tr.step(data.batch(WARM, 2))
torch.cuda.synchronize()
for fn, rois, scale, e0, e1 in rec:
    w, h, th = rois[:, 3] * scale, rois[:, 4] * scale, rois[:, 5]
    ex = (w * th.cos().abs() + h * th.sin().abs()).floor() + 2
    ey = (w * th.sin().abs() + h * th.cos().abs()).floor() + 2
    F = ex * ey
    q = torch.quantile(F.float(), torch.tensor([0.5, 0.9, 0.99], device=F.device)).tolist()
    print(f'{fn[-3:]} K={rois.shape[0]:6d} {e0.elapsed_time(e1) * 1e3:8.1f} us  footprint px median {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} max {float(F.max()):.0f}'
          f'  > 48 px: {float((F > 48).float().mean()) * 100:.1f} %  w median {float(w.median()):.1f} h median {float(h.median()):.1f} feature px')
