"""Golden vectors for the HOST LOGIC of the data pipeline - TEST INFRASTRUCTURE ONLY.

Run in the build container:  python -m oracle.gen_golden_pipeline   ->  tests/golden/pipeline_flow.npz
Every array is the output of a method of the reference's own files, loaded by path through oracle/_ref_loader.py:

    HBB_TOD/mmdet/datasets/pipelines/transforms.py   Resize._resize_bboxes / _random_scale, RandomFlip.bbox_flip / __call__
    HBB_TOD/mmdet/datasets/samplers/group_sampler.py GroupSampler / DistributedGroupSampler index streams
    HBB_TOD/mmdet/datasets/coco.py                   CocoDataset._parse_ann_info
    OBB_TOD/mmrotate/datasets/pipelines/transforms.py RResize._resize_bboxes, RRandomFlip.bbox_flip

Pixel functions (mmcv / cv2) are not installed and are never called here: parity for them is unpinned
(oracle/ref_pipeline.py header)."""
import importlib
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _ref_loader  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'pipeline_flow.npz')


def inputs():
    """Seeded inputs shared with tests/test_data_pipeline.py (regenerated there, not stored twice)."""
    r = np.random.RandomState(20240611)
    xy = r.uniform(-20, 780, (40, 2)).astype(np.float32)
    wh = r.uniform(1, 90, (40, 2)).astype(np.float32)
    hbb = np.concatenate([xy, xy + wh], 1)
    obb = np.concatenate([r.uniform(0, 800, (40, 2)), r.uniform(2, 80, (40, 2)), r.uniform(-np.pi / 2, np.pi / 2, (40, 1))],
                         1).astype(np.float32)
    obb[::7, 4] = np.float32(np.pi / 2)                                     # the 'oc' rule's special case
    anns = []
    for i in range(60):
        x, y = r.uniform(-30, 790, 2)
        w, h = r.uniform(0.2, 60, 2)
        anns.append(dict(bbox=[float(x), float(y), float(w), float(h)], area=float(w * h) if i % 11 else 0.0,
                         category_id=int(r.randint(1, 11)), iscrowd=int(i % 13 == 0), id=i + 1, image_id=7,
                         **({'ignore': True} if i % 17 == 0 else {})))
    flag = (r.uniform(size=37) > 0.7).astype(np.uint8)
    return dict(hbb=hbb, obb=obb, anns=anns, flag=flag)


def main():
    _ref_loader.install_pipeline()
    tr = importlib.import_module('mmdet.datasets.pipelines.transforms')
    rtr = importlib.import_module('mmrotate.datasets.pipelines.transforms')
    gs = importlib.import_module('mmdet.datasets.samplers.group_sampler')
    inp = inputs()
    out = {}

    # Resize: box scaling with and without border clipping, for an exact and a fractional scale factor
    for tag, sf, shape, clip in (('a', [1.5, 1.5, 1.5, 1.5], (1200, 1200, 3), True), ('b', [0.731, 0.7295, 0.731, 0.7295], (584, 585, 3), True),
                                 ('c', [1.25, 1.25, 1.25, 1.25], (1000, 1000, 3), False)):
        t = tr.Resize(img_scale=(800, 800), bbox_clip_border=clip)
        res = dict(bbox_fields=['gt_bboxes'], gt_bboxes=inp['hbb'].copy(), scale_factor=np.array(sf, dtype=np.float32), img_shape=shape)
        t._resize_bboxes(res)
        out['resize_bboxes_' + tag] = res['gt_bboxes']
    # Resize: the scale draws
    np.random.seed(11)
    t = tr.Resize(img_scale=[(1333, 640), (1333, 800)], multiscale_mode='range')
    out['scale_range'] = np.array([(lambda d: (t._random_scale(d), d['scale'])[1])({}) for _ in range(8)])
    t = tr.Resize(img_scale=[(1333, 640), (1000, 600), (800, 800)], multiscale_mode='value')
    out['scale_value'] = np.array([(lambda d: (t._random_scale(d), d['scale'] + (d['scale_idx'],))[1])({}) for _ in range(8)])
    t = tr.Resize(img_scale=(800, 800), ratio_range=(0.5, 1.5))
    out['scale_ratio'] = np.array([(lambda d: (t._random_scale(d), d['scale'])[1])({}) for _ in range(8)])

    # RandomFlip: the three box rules and the direction draw of __call__ (no image field -> no pixel op)
    f = tr.RandomFlip(flip_ratio=0.5)
    for d in ('horizontal', 'vertical', 'diagonal'):
        out['flip_' + d] = f.bbox_flip(inp['hbb'], (800, 760, 3), d)
    np.random.seed(5)
    f = tr.RandomFlip(flip_ratio=[0.3, 0.3, 0.2], direction=['horizontal', 'vertical', 'diagonal'])
    draws, boxes = [], []
    for _ in range(12):
        res = dict(img_fields=[], bbox_fields=['gt_bboxes'], gt_bboxes=inp['hbb'][:5].copy(), img_shape=(800, 760, 3))
        f(res)
        draws.append({None: 0, 'horizontal': 1, 'vertical': 2, 'diagonal': 3}[res['flip_direction']])
        boxes.append(res['gt_bboxes'])
    out['flip_draws'], out['flip_call_boxes'] = np.array(draws), np.stack(boxes)

    # oriented: RResize and RRandomFlip
    t = rtr.RResize(img_scale=(1200, 1200))
    res = dict(bbox_fields=['gt_bboxes'], gt_bboxes=inp['obb'].copy(), scale_factor=np.array([1.5, 1.4975, 1.5, 1.4975], dtype=np.float32))
    t._resize_bboxes(res)
    out['rresize_bboxes'] = res['gt_bboxes']
    for ver in ('oc', 'le90'):
        f = rtr.RRandomFlip(flip_ratio=0.5, version=ver)
        for d in ('horizontal', 'vertical', 'diagonal'):
            out[f'rflip_{ver}_{d}'] = f.bbox_flip(inp['obb'], (1200, 1184, 3), d)

    # samplers
    ds = types.SimpleNamespace(flag=inp['flag'])
    np.random.seed(3)
    out['group_sampler'] = np.array(list(iter(gs.GroupSampler(ds, samples_per_gpu=2))))
    out['group_sampler_spg3'] = np.array(list(iter(gs.GroupSampler(ds, samples_per_gpu=3))))
    for world in (1, 2, 4):
        for rank in range(world):
            s = gs.DistributedGroupSampler(ds, samples_per_gpu=2, num_replicas=world, rank=rank, seed=7)
            s.set_epoch(3)
            out[f'dist_sampler_w{world}_r{rank}'] = np.array(list(iter(s)))

    # COCO annotation parsing (the method only reads self.cat_ids / self.cat2label)
    sys.modules['mmdet.core'].eval_recalls = None
    coco = importlib.import_module('mmdet.datasets.coco')
    cat_ids = [1, 2, 3, 4, 5, 6, 7, 8]
    me = types.SimpleNamespace(cat_ids=cat_ids, cat2label={c: i for i, c in enumerate(cat_ids)})
    ann = coco.CocoDataset._parse_ann_info(me, dict(width=800, height=800, filename='P0001.png'), inp['anns'])
    out['coco_bboxes'], out['coco_labels'], out['coco_ignore'] = ann['bboxes'], ann['labels'], ann['bboxes_ignore']

    np.savez_compressed(OUT, **out)
    print('wrote', OUT, {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
