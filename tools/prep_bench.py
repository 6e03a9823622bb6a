#!/usr/bin/env python3
"""Micro-benchmark of the N2 data path on one MI355X: `pt_image_prep` alone (HIP events, inputs resident) and the
loader end to end (PNG decode threads -> pinned upload -> render) on a generated AI-TOD style folder."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from point_teacher_amd import datasets as D  # noqa: E402


def kernel(dev, src_hw, dst_hw, norm, reps=200):
    r = np.random.RandomState(0)
    bgr = r.randint(0, 256, src_hw + (3,)).astype(np.uint8)
    im = D.LazyImage(bgr)
    if dst_hw != src_hw:
        im.resize((dst_hw[1], dst_hw[0]))
    if norm:
        im.normalize([123.675, 116.28, 103.53], [58.395, 57.12, 57.375], True)
    im.pad_to(-(-dst_hw[0] // 32) * 32, -(-dst_hw[1] // 32) * 32, 0)
    out = torch.empty((1, 3, im.shape[0], im.shape[1]), device=dev).contiguous(memory_format=torch.channels_last)
    dsrc = torch.from_numpy(bgr).to(dev)
    for _ in range(20):
        im.render(out[0], device_src=dsrc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        im.render(out[0], device_src=dsrc)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    byts = bgr.nbytes + out.numel() * 4
    return dict(src=src_hw, dst=dst_hw, normalize=norm, us_per_image=round(us, 2), algorithmic_bytes=byts,
                GBps=round(byts / us / 1e3, 1))


def loader(dev, n=64, size=800, workers=8, cache_bytes=0):
    from PIL import Image
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, 'img'))
    r = np.random.RandomState(1)
    base = (r.randint(0, 256, (size // 8, size // 8, 3)).astype(np.uint8)).repeat(8, 0).repeat(8, 1)   # compressible like real tiles
    images, anns = [], []
    for i in range(n):
        img = np.roll(base, i * 7, axis=1)
        Image.fromarray(img).save(os.path.join(tmp, 'img', f'{i}.png'))
        images.append(dict(id=i, file_name=f'{i}.png', width=size, height=size))
        for k in range(50):
            anns.append(dict(id=i * 50 + k + 1, image_id=i, category_id=1 + k % 8, bbox=[float(k * 9), float(k * 11), 12.0, 9.0], area=108.0, iscrowd=0))
    cats = [dict(id=k + 1, name=c, supercategory='x') for k, c in enumerate(D.AITODDataset.CLASSES)]
    json.dump(dict(images=images, annotations=anns, categories=cats), open(os.path.join(tmp, 'a.json'), 'w'))
    pipe = [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
            dict(type='Resize', img_scale=(800, 800), keep_ratio=True), dict(type='RandomFlip', flip_ratio=0.0),
            dict(type='Normalize', mean=[0.0, 0.0, 0.0], std=[1.0, 1.0, 1.0], to_rgb=False), dict(type='Pad', size_divisor=32),
            dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])]
    ds = D.build_dataset(dict(type='AITODDataset', ann_file=os.path.join(tmp, 'a.json'), img_prefix=os.path.join(tmp, 'img'), pipeline=pipe))
    ld = D.build_dataloader(ds, samples_per_gpu=2, workers_per_gpu=workers, dist=False, shuffle=True, device=dev, cache_bytes=cache_bytes)
    for _ in ld:                                   # warm the page cache / pinned pool
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    for b in ld:
        k += b['img'].shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(images=k, workers=workers, hbm_cache=bool(cache_bytes), seconds=round(dt, 3), images_per_s=round(k / dt, 1))


def main():
    dev = torch.device('cuda:0')
    out = dict(kernel=[kernel(dev, (800, 800), (800, 800), False), kernel(dev, (800, 800), (800, 800), True),
                       kernel(dev, (800, 800), (1200, 1200), True)],
               loader=[loader(dev, workers=w) for w in (2, 8)] + [loader(dev, workers=2, cache_bytes=8 << 30)])
    print(json.dumps(out))


if __name__ == '__main__':
    main()
