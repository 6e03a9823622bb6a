"""torch.profiler view of one steady-state iteration: which aten ops launch the element-wise / reduction kernels that are
not ours (profiles/r01/*_summary.txt lumps them as "torch elementwise/reduce").  usage: python tools/op_profile.py [step1|step2]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import point_teacher_amd as pta  # noqa: E402
from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

phase2 = len(sys.argv) > 1 and sys.argv[1] == 'step2'
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
cfg.model['burn_in_step'] = -1 if phase2 else 10 ** 9
torch.manual_seed(0)
model = pta.build_detector(cfg.model).to(dev)
benchmark_init_(model, phase2=phase2)
model.train()
tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
data = SyntheticTiles(n=8, size=800, mean_objects=300, seed=7, device=dev)
for it in range(6):
    tr.step(data.batch(it, 2))
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for it in range(N):
        tr.step(data.batch(it, 2))
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    cuda = getattr(e, 'self_device_time_total', None) or getattr(e, 'self_cuda_time_total', 0)
    if cuda > 0:
        rows.append((cuda / N / 1e3, e.count / N, e.key))
rows.sort(reverse=True)
print(f'{"ms/iter":>9} {"calls/iter":>10}  op')
for ms, cnt, key in rows[:45]:
    print(f'{ms:9.3f} {cnt:10.1f}  {key[:90]}')
