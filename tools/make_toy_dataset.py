#!/usr/bin/env python3
"""Write the seeded synthetic tiles of point_teacher_amd.synthetic to disk in the AI-TOD layout (PNG tiles + one COCO
json) so that the whole data path - reader, pipeline, fused GPU image preparation, loader, evaluator - can be exercised
without the real dataset:

    python tools/make_toy_dataset.py /tmp/toy --n 16
    python tools/train.py point_teacher_amd/configs/point_teacher/aitodv2_point_teacher_0.py \\
        --ann-file /tmp/toy/ann.json --img-prefix /tmp/toy/img --max-iters 8
    python tools/test.py  point_teacher_amd/configs/point_teacher/aitodv2_point_teacher_0.py \\
        --ann-file /tmp/toy/ann.json --img-prefix /tmp/toy/img
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CLASSES = ('airplane', 'bridge', 'storage-tank', 'ship', 'swimming-pool', 'vehicle', 'person', 'wind-mill')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('out')
    ap.add_argument('--n', type=int, default=16)
    ap.add_argument('--size', type=int, default=800)
    ap.add_argument('--objects', type=int, default=300)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    from PIL import Image
    from point_teacher_amd.synthetic import make_tile
    os.makedirs(os.path.join(args.out, 'img'), exist_ok=True)
    images, anns = [], []
    for i in range(args.n):
        img, boxes, labels, _ = make_tile(i, args.size, len(CLASSES), args.objects, args.seed)
        bgr = img.permute(1, 2, 0).numpy().astype(np.uint8)                # the tiles are defined in the order the net sees (BGR)
        Image.fromarray(bgr[:, :, ::-1].copy()).save(os.path.join(args.out, 'img', f'{i:05d}.png'))
        images.append(dict(id=i, file_name=f'{i:05d}.png', width=args.size, height=args.size))
        for b, l in zip(boxes.tolist(), labels.tolist()):
            w, h = b[2] - b[0], b[3] - b[1]
            anns.append(dict(id=len(anns) + 1, image_id=i, category_id=l + 1, bbox=[b[0], b[1], w, h], area=w * h, iscrowd=0))
    cats = [dict(id=k + 1, name=c, supercategory='object') for k, c in enumerate(CLASSES)]
    with open(os.path.join(args.out, 'ann.json'), 'w') as f:
        json.dump(dict(images=images, annotations=anns, categories=cats), f)
    print(f'{args.n} tiles, {len(anns)} boxes -> {args.out}')


if __name__ == '__main__':
    main()
