"""Debug helper: run a few iterations and print every log var per iteration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import point_teacher_amd as pta
from point_teacher_amd.synthetic import SyntheticTiles

wl = sys.argv[1] if len(sys.argv) > 1 else 'step2'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 800
dev = torch.device('cuda:0')
torch.manual_seed(1234)
cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd/configs/point_teacher/aitodv2_point_teacher_0.py'))
cfg.model['burn_in_step'] = 10 ** 9 if wl == 'step1' else -1
model = pta.build_detector(cfg.model).to(dev)
from point_teacher_amd.synthetic import benchmark_init_
benchmark_init_(model, phase2=(wl == 'step2'))
model.train()
tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config)
data = SyntheticTiles(n=4, size=size, mean_objects=300 if size == 800 else 30, seed=7, device=dev)
for it in range(4):
    out = tr.step(data.batch(it, 2))
    lv = out['log_vars'].materialize()
    print(it, {k: float(f'{v:.4g}') for k, v in lv.items()})
    print('   grad sqnorm', float(pta.functional.grad_sqnorm(tr.flat.grad_flat)), 'param finite',
          bool(torch.isfinite(tr.flat.student_flat).all()))
