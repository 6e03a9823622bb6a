"""cProfile of the host side of an iteration (bf16 phase 2 is host-bound: ~2 000 launches per iteration)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import point_teacher_amd as pta
from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
cfg.model['burn_in_step'] = -1
torch.manual_seed(1234)
model = pta.build_detector(cfg.model).to(dev)
benchmark_init_(model, phase2=True)
model.train()
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == 'bf16') else None
tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True, autocast_dtype=dt)
data = SyntheticTiles(n=8, size=800, mean_objects=300, seed=7, device=dev)
for it in range(6):
    tr.step(data.batch(it, 2))
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for it in range(10):
    tr.step(data.batch(it, 2))
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
from point_teacher_amd import planes as _PL
_PL.CENSUS.poll(sync=True)
print(f'un-profiled: host issue {t_issue * 100:.2f} ms/iter, with the GPU drained {t_all * 100:.2f} ms/iter; census demotions: {dict(_PL.CENSUS.demoted)}')
pr = cProfile.Profile()
pr.enable()
for it in range(10):
    tr.step(data.batch(it, 2))
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(30)
st.sort_stats('cumulative').print_stats(60)
