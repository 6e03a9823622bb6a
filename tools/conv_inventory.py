"""Where the library convolution time of one iteration goes, per SHAPE: torch.profiler with record_shapes over a steady-state
iteration; every aten convolution / convolution_backward / addmm / mm call grouped by its input shapes with its device time.
usage: python tools/conv_inventory.py [step1|step2] [obb]"""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import point_teacher_amd as pta  # noqa: E402
from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

phase2 = 'step2' in sys.argv[1:]
obb = 'obb' in sys.argv[1:]
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
if obb:
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'obb', 'point_teacher', 'sodaa_fcos_pointteacher_1x.py'))
    size = 1200
else:
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
    size = 800
cfg.model['burn_in_step'] = -1 if phase2 else 10 ** 9
torch.manual_seed(0)
model = pta.build_detector(cfg.model).to(dev)
benchmark_init_(model, phase2=phase2)
model.train()
tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
data = SyntheticTiles(n=8, size=size, mean_objects=300, seed=7, device=dev, oriented=True, num_classes=9) if obb else \
    SyntheticTiles(n=8, size=size, mean_objects=300, seed=7, device=dev)
for it in range(6):
    tr.step(data.batch(it, 2))
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for it in range(N):
        tr.step(data.batch(it, 2))
    torch.cuda.synchronize()
agg = defaultdict(lambda: [0.0, 0])
for e in prof.key_averages(group_by_input_shape=True):
    k = e.key
    if not any(s in k for s in ('convolution', 'addmm', 'aten::mm', 'miopen', 'cudnn')):
        continue
    dt = getattr(e, 'device_time_total', None) or getattr(e, 'cuda_time_total', 0)
    sdt = getattr(e, 'self_device_time_total', None) or getattr(e, 'self_cuda_time_total', 0)
    if sdt <= 0:
        continue
    shp = str([s for s in e.input_shapes if s])[:150]
    a = agg[(k, shp)]
    a[0] += sdt / N / 1e3
    a[1] += e.count / N
rows = sorted(((v[0], v[1], k[0], k[1]) for k, v in agg.items()), reverse=True)
tot = sum(r[0] for r in rows)
print(f'library matrix ops: {tot:.3f} ms / iteration (self device time), {"phase 2" if phase2 else "phase 1"}{" obb" if obb else ""}')
print(f'{"ms/iter":>9} {"calls":>6}  op  shapes')
for ms, cnt, key, shp in rows[:70]:
    print(f'{ms:9.3f} {cnt:6.1f}  {key[:44]:44s} {shp}')
