#!/usr/bin/env python3
"""Train a Point-Teacher config on MI355X (the counterpart of the reference's tools/train.py for this path).

    python tools/train.py <config.py> [--work-dir DIR] [--resume-from CKPT] [--synthetic N] [--max-iters K]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py <config.py> ...

When the config's `data.train` paths exist (or `--ann-file/--img-prefix` point at a dataset) the loop reads it through
`point_teacher_amd.datasets` (readers + pipelines + the fused GPU image preparation); otherwise it runs
on `--synthetic N` seeded tiles of the config's image size resident on the GPU."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config')
    ap.add_argument('--work-dir', default=None)
    ap.add_argument('--resume-from', default=None)
    ap.add_argument('--synthetic', type=int, default=64)
    ap.add_argument('--max-iters', type=int, default=None)
    ap.add_argument('--iters-per-epoch', type=int, default=None)
    ap.add_argument('--objects', type=int, default=300)
    ap.add_argument('--ann-file', default=None, help="override data.train.ann_file")
    ap.add_argument('--img-prefix', default=None, help="override data.train.img_prefix")
    ap.add_argument('--ori-ann-file', default=None, help="override data.train.ori_ann_file (SODA-A)")
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--load-from', default=None, help='checkpoint to initialise the whole detector from (cfg.load_from)')
    ap.add_argument('--benchmark-init', action='store_true',
                    help='scaled random initialisation of synthetic.benchmark_init_ (the default on synthetic tiles)')
    ap.add_argument('--allow-random-init', action='store_true',
                    help='train on a real dataset although the pretrained backbone of the config was not found')
    ap.add_argument('--cache-gb', type=float, default=48.0, help='decoded uint8 tiles kept in HBM between epochs (0 = off)')
    args = ap.parse_args()
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)          # RCCL
    import point_teacher_amd as pta
    from point_teacher_amd.runner import Runner
    from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_
    cfg = pta.Config.fromfile(args.config)
    torch.manual_seed(0)
    model = pta.build_detector(cfg.model).to(dev)        # ResNet honours init_cfg=dict(type='Pretrained', ...) when the file exists
    model.train()
    oriented = cfg.model.type == 'RotatedFCOS_TS'
    size = 1200 if oriented else 800
    tcfg = dict(cfg.data.train)
    for k, v in (('ann_file', args.ann_file), ('img_prefix', args.img_prefix), ('ori_ann_file', args.ori_ann_file)):
        if v is not None:
            tcfg[k] = v
    real = os.path.exists(str(tcfg.get('ann_file', '')))
    load_from = args.load_from or cfg.get('load_from')
    if load_from:
        from point_teacher_amd.nn_modules import load_pretrained
        assert load_pretrained(model, load_from), f'--load-from {load_from}: not found'
    loaded = [m.pretrained_loaded for m in model.modules() if getattr(m, 'pretrained_loaded', None) is not None]
    if real and not load_from and loaded and not all(loaded) and not args.allow_random_init:
        raise SystemExit('the pretrained backbone named by the config was not found (see the warning above): put it under '
                         '$PT_PRETRAINED_DIR, pass --load-from, or pass --allow-random-init to train from scratch')
    if hasattr(model, 'student') and (args.benchmark_init or (not real and not load_from and not any(loaded))):
        benchmark_init_(model)             # synthetic tiles + random weights: keep activations O(1), see its docstring
    if real:
        from point_teacher_amd.datasets import EpochBatches, build_dataloader, build_dataset
        dataset = build_dataset(tcfg)
        loader = build_dataloader(dataset, cfg.data.samples_per_gpu, cfg.data.workers_per_gpu, dist=world > 1, shuffle=True,
                                  seed=args.seed, device=dev, cache_bytes=int(args.cache_gb * (1 << 30)))
        ipe = args.iters_per_epoch or len(loader)
        batches = EpochBatches(loader)
        if rank == 0:
            print(f'{type(dataset).__name__}: {len(dataset)} images, {len(loader)} iterations per epoch per rank')
    else:
        ipe = args.iters_per_epoch or max(args.synthetic // (cfg.data.samples_per_gpu * world), 1)
        data = SyntheticTiles(n=max(args.synthetic // world, 2), size=size, mean_objects=args.objects, seed=0, device=dev,
                              rank=rank, world=world, oriented=oriented, num_classes=cfg.model.get('_model_', cfg.model)['bbox_head']['num_classes'])
        batches = data.batch
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, iters_per_epoch=ipe, channels_last=True)
    torch.backends.cudnn.benchmark = True
    work_dir = args.work_dir or os.path.join(ROOT, 'work_dirs', os.path.splitext(os.path.basename(args.config))[0])
    runner = Runner(trainer, batches, work_dir, cfg.runner.max_epochs, ipe, cfg.log_config.interval,
                    cfg.checkpoint_config.interval, cfg.data.samples_per_gpu)
    vcfg = dict(cfg.data.get('val', {}))
    if real and os.path.exists(str(vcfg.get('ann_file', ''))) and cfg.get('evaluation'):
        vcfg['test_mode'] = True
        vds = build_dataset(vcfg)
        vld = build_dataloader(vds, 1, cfg.data.workers_per_gpu, dist=world > 1, shuffle=False, device=dev)   # every rank tests its shard
        runner.register_eval(vds, vld, cfg.evaluation.get('interval', 1))
    if args.resume_from:
        meta = runner.resume(args.resume_from)
        if rank == 0:
            print(f'resumed from {args.resume_from}: {meta}')
    runner.run(max_iters=args.max_iters)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
