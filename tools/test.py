#!/usr/bin/env python3
"""Evaluate a checkpoint (the counterpart of the reference's tools/test.py for this path).

    python tools/test.py <config.py> <checkpoint.pth> [--synthetic N] [--iou-thr 0.5]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/test.py <config.py> <checkpoint.pth> ...

Runs `simple_test` of the TEACHER over the data (apis/test.py:16-66) and reports the reference's metrics: AI-TOD
COCO-style `bbox_mAP*` / `AR*` for the horizontal configs (datasets/aitod.py), DOTA-style `mAP` for the oriented one
(core/evaluation/eval_map.py).  When the config's `data.test` paths exist (or `--ann-file/--img-prefix` are given) the
validation set is read through `point_teacher_amd.datasets` and scored by the dataset's own `evaluate`; otherwise
`--synthetic N` seeded tiles stand in for it."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config')
    ap.add_argument('checkpoint', nargs='?', default=None)
    ap.add_argument('--synthetic', type=int, default=16)
    ap.add_argument('--objects', type=int, default=100)
    ap.add_argument('--iou-thr', type=float, default=0.5)
    ap.add_argument('--ann-file', default=None)
    ap.add_argument('--img-prefix', default=None)
    ap.add_argument('--ori-ann-file', default=None)
    args = ap.parse_args()
    rank, world, local = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:                                        # sharded evaluation (apis/test.py multi_gpu_test): RCCL for the gather
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)
    import point_teacher_amd as pta
    from point_teacher_amd.evaluation import AITODEvaluator, eval_rbbox_map, multi_gpu_test, single_gpu_test
    from point_teacher_amd.synthetic import SyntheticTiles
    cfg = pta.Config.fromfile(args.config)
    model = pta.build_detector(cfg.model).to(dev)
    if args.checkpoint:
        state = torch.load(args.checkpoint, map_location=dev, weights_only=False)
        model.load_state_dict(state['model'] if 'model' in state else state.get('state_dict', state))
    oriented = cfg.model.type == 'RotatedFCOS_TS'
    K = cfg.model.get('_model_', cfg.model)['bbox_head']['num_classes']      # TS wrappers nest the detector under `_model_`
    B = cfg.data.samples_per_gpu
    tcfg = dict(cfg.data.test)
    for k, v in (('ann_file', args.ann_file), ('img_prefix', args.img_prefix), ('ori_ann_file', args.ori_ann_file)):
        if v is not None:
            tcfg[k] = v
    if os.path.exists(str(tcfg.get('ann_file', ''))):
        from point_teacher_amd.datasets import build_dataloader, build_dataset
        tcfg['test_mode'] = True
        dataset = build_dataset(tcfg)
        loader = build_dataloader(dataset, 1, cfg.data.workers_per_gpu, dist=world > 1, shuffle=False, device=dev)
        results = multi_gpu_test(model, loader, len(dataset))        # world 1: the plain test loop
        if rank == 0:
            out = dataset.evaluate(results)          # the dataset's own protocol: AI-TOD COCO-style / SODA-A / DOTA-style mAP
            print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()})
        if world > 1:
            dist.destroy_process_group()
        return
    assert world == 1, 'the synthetic stand-in is a single-process check'

    data = SyntheticTiles(n=args.synthetic, size=1200 if oriented else 800, mean_objects=args.objects, seed=1, device=dev,
                          oriented=oriented, num_classes=K)
    results = single_gpu_test(model, lambda it: data.batch(it, B), args.synthetic // B)
    n = len(results)
    if oriented:
        anns = [dict(bboxes=data.items[i][1].cpu().numpy(), labels=data.items[i][2].cpu().numpy()) for i in range(n)]
        mean_ap, per_cls = eval_rbbox_map(results, anns, iou_thr=args.iou_thr)
        print({'mAP': round(mean_ap, 4), 'per_class_AP': [round(r['ap'], 4) for r in per_cls]})
    else:
        gts = [dict(bboxes=data.items[i][1].cpu().numpy(), labels=data.items[i][2].cpu().numpy()) for i in range(n)]
        out = AITODEvaluator(gts, K).evaluate(results)
        print({k: round(v, 4) for k, v in out.items() if isinstance(v, float)})


if __name__ == '__main__':
    main()
