#!/usr/bin/env python3
"""Iterations/s of the supervised baselines of row N4 on one MI355X - `--model fcos` (configs/baselines/aitodv2_fcos_r50_1x.py)
`--model retinanet` (aitodv2_retinanet_r50_1x.py) or `--model faster_rcnn` (aitodv2_faster_rcnn_r50_1x.py): the same synthetic 800x800 tiles, batch 2, fp32 and flat-parameter
Trainer as bench.py uses for the Point-Teacher iteration."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(model_name='fcos', steps=20, warmup=5, batch=2):
    import point_teacher_amd as pta
    from point_teacher_amd.synthetic import SyntheticTiles
    dev = torch.device('cuda:0')
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(1234)
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'baselines', f'aitodv2_{model_name}_r50_1x.py'))
    model = pta.build_detector(cfg.model).to(dev)
    with torch.no_grad():
        model.backbone.conv1.weight.mul_(1.0 / 64.0)       # raw 0-255 tiles into a random stem (synthetic.benchmark_init_)
    model.train()
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, iters_per_epoch=5000, channels_last=True)
    data = SyntheticTiles(n=8, size=800, mean_objects=300, seed=7, device=dev)
    for it in range(warmup):
        trainer.step(data.batch(it, batch))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        out = trainer.step(data.batch(warmup + it, batch))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lv = out['log_vars'].materialize()
    print(json.dumps(dict(metric=f'train iters/sec (supervised {model_name} baseline, 800x800, ~300 boxes/img)', value=round(steps / dt, 3),
                          ms_per_step=round(dt / steps * 1e3, 2), batch=batch, dtype='f32', data='synthetic', loss=round(lv['loss'], 4))))


if __name__ == '__main__':
    main(sys.argv[sys.argv.index('--model') + 1] if '--model' in sys.argv else 'fcos')
