"""Soak run: N iterations across the phase switch (burn-in step 1 -> step 2) with fresh tiles every iteration;
checks finiteness, a bounded allocator footprint and the size of the detector's per-image dictionaries."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import point_teacher_amd as pta
from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_

variant = sys.argv[1] if len(sys.argv) > 1 else 'hbb'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device('cuda:0')
cfgp = 'obb/point_teacher/sodaa_fcos_pointteacher_1x.py' if variant == 'obb' else 'point_teacher/aitodv2_point_teacher_0.py'
cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', cfgp))
cfg.model['burn_in_step'] = N // 2
torch.manual_seed(0)
model = pta.build_detector(cfg.model).to(dev)
benchmark_init_(model, phase2=True)
model.train()
tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
data = SyntheticTiles(n=16, size=512, mean_objects=60, seed=3, device=dev, oriented=variant == 'obb', num_classes=9 if variant == 'obb' else 8)
peak = []
for it in range(N):
    out = tr.step(data.batch(it, 2))
    if it % 10 == 9 or it == N - 1:
        lv = out['log_vars'].materialize()
        assert all(v == v and abs(v) != float('inf') for v in lv.values()), (it, lv)
        peak.append(torch.cuda.memory_allocated() / 2 ** 20)
        print(f'iter {it + 1:4d} phase {1 if model.count - 1 <= model.burn_in_step else 2} loss {lv["loss"]:10.4f} '
              f'allocated {peak[-1]:8.1f} MiB reserved {torch.cuda.memory_reserved() / 2 ** 20:8.1f} MiB dict {len(model.gt_bboxes_point)}', flush=True)
assert len(model.gt_bboxes_point) == 16 and len(model.refined_gt_bboxes_point) == 16
assert max(peak[-3:]) < 1.15 * max(peak[:3]) + 64, peak
assert torch.isfinite(tr.flat.student_flat).all() and torch.isfinite(tr.flat.teacher_flat).all()
print('soak ok')
