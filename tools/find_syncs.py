"""List every host-synchronising call site of a steady-state iteration (torch.cuda.set_sync_debug_mode('warn'))."""
import os, sys, traceback, warnings, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import point_teacher_amd as pta
from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_

dev = torch.device('cuda:0')
sites = collections.Counter()


def show(message, category, filename, lineno, file=None, line=None):
    if 'synchronizing' not in str(message):
        return
    st = [f for f in traceback.extract_stack() if '/point_teacher_amd/' in f.filename or f.filename.endswith('find_syncs.py')]
    key = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(st[-4:]))
    sites[key] += 1


warnings.showwarning = show
warnings.simplefilter('always')
for variant in ('hbb', 'obb'):
    for phase2 in (False, True):
        cfgp = ('obb/point_teacher/sodaa_fcos_pointteacher_1x.py' if variant == 'obb' else 'point_teacher/aitodv2_point_teacher_0.py')
        cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', cfgp))
        cfg.model['burn_in_step'] = -1 if phase2 else 10 ** 9
        model = pta.build_detector(cfg.model).to(dev)
        benchmark_init_(model, phase2=True)
        model.train()
        tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
        data = SyntheticTiles(n=4, size=256, mean_objects=20, seed=1, device=dev, oriented=variant == 'obb',
                              num_classes=9 if variant == 'obb' else 8)
        tr.step(data.batch(0, 2)); tr.step(data.batch(1, 2))
        torch.cuda.synchronize()
        sites.clear()
        torch.cuda.set_sync_debug_mode('warn')
        tr.step(data.batch(0, 2))
        torch.cuda.set_sync_debug_mode('default')
        print(f'== {variant} {"phase2" if phase2 else "phase1"}: {sum(sites.values())} synchronising calls')
        for k, v in sites.most_common():
            print(f'   {v:3d} x {k}')
