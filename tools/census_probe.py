"""Range census of the fp16 planes, iteration by iteration (synchronising): which sites saturate / sink, and what the losses do.
usage: python tools/census_probe.py [step1|step2] [iters] [nofallback]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import point_teacher_amd as pta  # noqa: E402
import point_teacher_amd.planes as PL  # noqa: E402
from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_  # noqa: E402

phase2 = 'step2' in sys.argv[1:]
iters = next((int(a) for a in sys.argv[1:] if a.isdigit()), 30)
if 'nofallback' in sys.argv[1:]:
    PL.CENSUS.auto_fallback = False
dev = torch.device('cuda:0')
cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
cfg.model['burn_in_step'] = -1 if phase2 else 10 ** 9
torch.manual_seed(1234)
model = pta.build_detector(cfg.model).to(dev)
benchmark_init_(model, phase2=phase2)
model.train()
tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, iters_per_epoch=5000, channels_last=True)
tr.census_interval = 10 ** 9
data = SyntheticTiles(n=16, size=800, mean_objects=300, seed=7, device=dev)
switch = next((int(a[7:]) for a in sys.argv[1:] if a.startswith('switch=')), None)      # flip to phase 2 after this many iterations
for it in range(iters):
    if switch is not None and it == switch:
        model.burn_in_step = -1
    out = tr.step(data.batch(it, 2))
    lv = out['log_vars'].materialize()
    if switch is not None and it < switch - 2:
        PL.CENSUS.poll(sync=True)
        continue
    cen = PL.CENSUS.poll(sync=True)
    sat = {f'{g}/{n}': d['saturated'] for (g, n), d in cen.items() if d['saturated']}
    am = sorted(((d['amax_stored'], f'{g}/{n}') for (g, n), d in cen.items()), reverse=True)
    lo = sorted(((d['amax_stored'], f'{g}/{n}') for (g, n), d in cen.items() if d['amax_stored'] > 0))
    wmax = float(tr.flat.student_flat.abs().max())
    print(f'it {it:3d} loss {lv.get("loss", float("nan")):10.4f} |w|max {wmax:8.3f} sites {len(cen)} saturated {sat} top {[(round(a, 1), n) for a, n in am[:4]]} '
          f'low {[(round(a, 3), n) for a, n in lo[:4]]} demoted {sorted(PL.CENSUS.demoted)}', flush=True)
