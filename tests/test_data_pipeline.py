"""Data pipeline: readers, transforms, samplers, the fused GPU image preparation and the loader.

CPU part: the oracle AND the product's host logic against tests/golden/pipeline_flow.npz (outputs of the reference's
own transforms / samplers / COCO reader, oracle/gen_golden_pipeline.py), the readers on generated AI-TOD / SODA-A
style folders, the decoder, the transform-order guard.
GPU part (-m gpu): `pt_image_prep` through the C ABI against oracle/ref_pipeline.py - BIT-EXACT (the resize is 8-bit
fixed point, the normalisation is one float subtract and one double multiply) - the device collate, and one training
iteration fed by the loader.  The pixel functions of the reference are mmcv/cv2 (not installed): that half of the
oracle is a restatement of the published algorithm (parity unpinned, see its header)."""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_pipeline as R                      # noqa: E402
from oracle.gen_golden_pipeline import inputs             # noqa: E402  (the seeded inputs; nothing of the reference is imported)
from conftest import load_golden                          # noqa: E402

import point_teacher_amd as pta                           # noqa: E402
from point_teacher_amd import datasets as D               # noqa: E402

G = load_golden('pipeline_flow')
INP = inputs()
DIRS = ('horizontal', 'vertical', 'diagonal')


# ------------------------------------------------------------------------------------------------ oracle vs the reference
def test_oracle_host_logic_matches_reference():
    hbb, obb = INP['hbb'], INP['obb']
    for tag, sf, shape, clip in (('a', [1.5] * 4, (1200, 1200, 3), True), ('b', [0.731, 0.7295, 0.731, 0.7295], (584, 585, 3), True),
                                 ('c', [1.25] * 4, (1000, 1000, 3), False)):
        np.testing.assert_array_equal(R.resize_bboxes(hbb, np.array(sf, np.float32), shape, clip), G['resize_bboxes_' + tag])
    for d in DIRS:
        np.testing.assert_array_equal(R.bbox_flip(hbb, (800, 760, 3), d), G['flip_' + d])
        for ver in ('oc', 'le90'):
            np.testing.assert_array_equal(R.rbbox_flip(obb, (1200, 1184, 3), d, ver), G[f'rflip_{ver}_{d}'])
    np.testing.assert_array_equal(R.rresize_bboxes(obb, np.array([1.5, 1.4975, 1.5, 1.4975], np.float32)), G['rresize_bboxes'])
    rng = np.random.RandomState(3)
    assert R.group_sampler_indices(INP['flag'], 2, rng) == G['group_sampler'].tolist()
    assert R.group_sampler_indices(INP['flag'], 3, rng) == G['group_sampler_spg3'].tolist()
    for world in (1, 2, 4):
        for rank in range(world):
            assert R.distributed_group_sampler_indices(INP['flag'], 2, world, rank, epoch=3, seed=7) == G[f'dist_sampler_w{world}_r{rank}'].tolist()
    b, l, ig = R.parse_coco_ann(dict(width=800, height=800), INP['anns'], list(range(1, 9)))
    np.testing.assert_array_equal(b, G['coco_bboxes'])
    np.testing.assert_array_equal(l, G['coco_labels'])
    np.testing.assert_array_equal(ig, G['coco_ignore'])


# ------------------------------------------------------------------------------------------------ product host logic
def test_resize_and_flip_boxes_match_reference():
    hbb, obb = INP['hbb'], INP['obb']
    for tag, sf, shape, clip in (('a', [1.5] * 4, (1200, 1200, 3), True), ('b', [0.731, 0.7295, 0.731, 0.7295], (584, 585, 3), True),
                                 ('c', [1.25] * 4, (1000, 1000, 3), False)):
        t = D.Resize(img_scale=(800, 800), bbox_clip_border=clip)
        res = dict(bbox_fields=['gt_bboxes'], gt_bboxes=hbb.copy(), scale_factor=np.array(sf, np.float32), img_shape=shape)
        t._resize_bboxes(res)
        np.testing.assert_array_equal(res['gt_bboxes'], G['resize_bboxes_' + tag])
    f = D.RandomFlip(flip_ratio=0.5)
    for d in DIRS:
        np.testing.assert_array_equal(f.bbox_flip(hbb, (800, 760, 3), d), G['flip_' + d])
    t = D.RResize(img_scale=(1200, 1200))
    res = dict(bbox_fields=['gt_bboxes'], gt_bboxes=obb.copy(), scale_factor=np.array([1.5, 1.4975, 1.5, 1.4975], np.float32))
    t._resize_bboxes(res)
    np.testing.assert_array_equal(res['gt_bboxes'], G['rresize_bboxes'])
    for ver in ('oc', 'le90'):
        f = D.RRandomFlip(flip_ratio=0.5, version=ver)
        for d in DIRS:
            np.testing.assert_array_equal(f.bbox_flip(obb, (1200, 1184, 3), d), G[f'rflip_{ver}_{d}'])


def test_random_draws_match_reference():
    """Same numpy seed -> the same scale and flip-direction stream as the reference's transforms."""
    np.random.seed(11)
    t = D.Resize(img_scale=[(1333, 640), (1333, 800)], multiscale_mode='range')
    got = []
    for _ in range(8):
        got.append(t._pick_scale()[0])
    np.testing.assert_array_equal(np.array(got), G['scale_range'])
    t = D.Resize(img_scale=[(1333, 640), (1000, 600), (800, 800)], multiscale_mode='value')
    got = []
    for _ in range(8):
        scale, idx = t._pick_scale()
        got.append(scale + (idx,))
    np.testing.assert_array_equal(np.array(got), G['scale_value'])
    t = D.Resize(img_scale=(800, 800), ratio_range=(0.5, 1.5))
    got = []
    for _ in range(8):
        got.append(t._pick_scale()[0])
    np.testing.assert_array_equal(np.array(got), G['scale_ratio'])
    np.random.seed(5)
    f = D.RandomFlip(flip_ratio=[0.3, 0.3, 0.2], direction=list(DIRS))
    draws, boxes = [], []
    for _ in range(12):
        res = dict(img_fields=[], bbox_fields=['gt_bboxes'], gt_bboxes=INP['hbb'][:5].copy(), img_shape=(800, 760, 3))
        f(res)
        draws.append({None: 0, 'horizontal': 1, 'vertical': 2, 'diagonal': 3}[res['flip_direction']])
        boxes.append(res['gt_bboxes'])
    np.testing.assert_array_equal(np.array(draws), G['flip_draws'])
    np.testing.assert_array_equal(np.stack(boxes), G['flip_call_boxes'])


def test_samplers_match_reference():
    ds = types.SimpleNamespace(flag=INP['flag'])
    np.random.seed(3)
    assert list(iter(D.GroupSampler(ds, samples_per_gpu=2))) == G['group_sampler'].tolist()
    assert list(iter(D.GroupSampler(ds, samples_per_gpu=3))) == G['group_sampler_spg3'].tolist()
    for world in (1, 2, 4):
        seen = []
        for rank in range(world):
            s = D.DistributedGroupSampler(ds, samples_per_gpu=2, num_replicas=world, rank=rank, seed=7)
            s.set_epoch(3)
            idx = list(iter(s))
            assert idx == G[f'dist_sampler_w{world}_r{rank}'].tolist()
            assert len(s) == len(idx)
            seen += idx
        assert set(seen) == set(range(len(INP['flag'])))            # the ranks cover the dataset
    s = D.DistributedSampler(list(range(7)), num_replicas=2, rank=1, shuffle=False)
    assert list(iter(s)) == [1, 3, 5, 0]


def test_coco_annotation_parsing_matches_reference():
    cat_ids = list(range(1, 9))
    me = types.SimpleNamespace(cat_ids=cat_ids, cat2label={c: i for i, c in enumerate(cat_ids)})
    ann = D.CocoDataset._parse_ann_info(me, dict(width=800, height=800, filename='P0001.png'), INP['anns'])
    np.testing.assert_array_equal(ann['bboxes'], G['coco_bboxes'])
    np.testing.assert_array_equal(ann['labels'], G['coco_labels'])
    np.testing.assert_array_equal(ann['bboxes_ignore'], G['coco_ignore'])
    assert ann['seg_map'] == 'P0001.png'


# ------------------------------------------------------------------------------------------------ generated datasets
def _write_png(path, arr_rgb, mode='RGB'):
    from PIL import Image
    im = Image.fromarray(arr_rgb)
    if mode != 'RGB':
        im = im.convert(mode)
    im.save(path)


def _make_aitod(tmp, n=5, sizes=((160, 160), (128, 192), (160, 160), (96, 96), (20, 200)), seed=0):
    r = np.random.RandomState(seed)
    os.makedirs(os.path.join(tmp, 'img'), exist_ok=True)
    images, anns, pix = [], [], {}
    aid = 1
    for i in range(n):
        h, w = sizes[i % len(sizes)]
        rgb = r.randint(0, 256, (h, w, 3)).astype(np.uint8)
        name = f'P{i:04d}.png'
        _write_png(os.path.join(tmp, 'img', name), rgb)
        pix[name] = rgb[:, :, ::-1].copy()
        images.append(dict(id=100 + i, file_name=name, width=w, height=h))
        for _ in range(0 if i == 2 else 6):                     # image 2 has no annotation -> filtered in train mode
            x, y = r.uniform(0, w - 12), r.uniform(0, h - 12)
            bw, bh = r.uniform(2, 11, 2)
            anns.append(dict(id=aid, image_id=100 + i, category_id=int(r.randint(1, 9)), bbox=[float(x), float(y), float(bw), float(bh)],
                             area=float(bw * bh), iscrowd=0))
            aid += 1
    cats = [dict(id=k + 1, name=c, supercategory='x') for k, c in enumerate(D.AITODDataset.CLASSES)]
    ann_file = os.path.join(tmp, 'ann.json')
    with open(ann_file, 'w') as f:
        json.dump(dict(images=images, annotations=anns, categories=cats), f)
    return ann_file, os.path.join(tmp, 'img'), pix


TRAIN_PIPELINE = [
    dict(type='LoadImageFromFile'),
    dict(type='LoadAnnotations', with_bbox=True),
    dict(type='Resize', img_scale=(240, 240), keep_ratio=True),
    dict(type='RandomFlip', flip_ratio=0.5),
    dict(type='Normalize', mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
    dict(type='Pad', size_divisor=32),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels']),
]
TEST_PIPELINE = [
    dict(type='LoadImageFromFile'),
    dict(type='MultiScaleFlipAug', img_scale=(240, 240), flip=False,
         transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'),
                     dict(type='Normalize', mean=[0.0, 0.0, 0.0], std=[1.0, 1.0, 1.0], to_rgb=False),
                     dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])]),
]


def _expected_sample(bgr, ann, meta):
    """The reference's pipeline on one image through the oracle, with the flip decision the product drew."""
    flip = meta['flip_direction'] if meta['flip'] else None
    chw, m = R.prepare_image(bgr, scale=(240, 240), keep_ratio=True, flip=flip,
                             norm=dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True), size_divisor=32)
    boxes = R.resize_bboxes(ann['bboxes'], m['scale_factor'], m['img_shape'])
    if flip:
        boxes = R.bbox_flip(boxes, m['img_shape'], flip)
    return chw, m, boxes


def test_aitod_reader_and_host_pipeline(tmp_path):
    ann_file, prefix, pix = _make_aitod(str(tmp_path))
    ds = D.build_dataset(dict(type='AITODDataset', ann_file=ann_file, img_prefix=prefix, pipeline=TRAIN_PIPELINE))
    # image 2 has no box (filter_empty_gt) and image 4 is 20 px high (min_size=32): both are dropped in train mode
    assert len(ds) == 3 and ds.img_ids == [100, 101, 103]
    assert ds.flag.tolist() == [0, 1, 0]                                 # width / height > 1 -> group 1
    np.random.seed(0)
    for i in range(len(ds)):
        s = ds[i]
        assert set(s) == {'img_metas', 'img', 'gt_bboxes', 'gt_labels'}
        meta, img = s['img_metas'].data, s['img'].data
        assert isinstance(img, D.LazyImage) and s['img'].stack
        assert set(meta) == {'filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor', 'flip',
                             'flip_direction', 'img_norm_cfg'}
        name = meta['ori_filename']
        np.testing.assert_array_equal(img.src, pix[name])                # PNG decodes losslessly, BGR order
        _, m, boxes = _expected_sample(pix[name], ds.get_ann_info(i), meta)
        assert meta['img_shape'] == m['img_shape'] and meta['pad_shape'] == m['pad_shape'] and meta['ori_shape'] == m['ori_shape']
        np.testing.assert_array_equal(meta['scale_factor'], m['scale_factor'])
        np.testing.assert_array_equal(s['gt_bboxes'].data.numpy(), boxes)
        assert s['gt_bboxes'].data.dtype == torch.float32 and s['gt_labels'].data.dtype == torch.int64
        assert img.shape == m['pad_shape'] and img.dtype == np.float32
    dt = D.build_dataset(dict(type='AITODDataset', ann_file=ann_file, img_prefix=prefix, pipeline=TEST_PIPELINE, test_mode=True))
    assert len(dt) == 5                                                  # test mode never filters
    s = dt[4]
    assert isinstance(s['img'], list) and len(s['img']) == 1 and isinstance(s['img'][0], D.LazyImage)
    assert s['img_metas'][0].data['flip'] is False and s['img_metas'][0].data['img_shape'] == (24, 240, 3)
    assert s['img_metas'][0].data['pad_shape'] == (32, 256, 3)


def test_decoder_modes(tmp_path):
    """`imfrombytes(flag='color')` always hands a 3-channel BGR uint8 array, whatever the file stores."""
    r = np.random.RandomState(1)
    rgb = r.randint(0, 256, (9, 13, 3)).astype(np.uint8)
    for mode in ('RGB', 'L', 'P', 'RGBA'):
        p = os.path.join(str(tmp_path), f'{mode}.png')
        _write_png(p, rgb, mode)
        with open(p, 'rb') as f:
            out = D.decode_image(f.read())
        assert out.dtype == np.uint8 and out.shape == (9, 13, 3) and out.flags['C_CONTIGUOUS']
        if mode in ('RGB', 'RGBA'):
            np.testing.assert_array_equal(out, rgb[:, :, ::-1])
        if mode == 'L':
            assert (out[..., 0] == out[..., 1]).all() and (out[..., 1] == out[..., 2]).all()
    with pytest.raises(NotImplementedError):
        D.decode_image(b'', flag='grayscale')


def test_transform_order_guard():
    """Orders the fused kernel cannot express fail loudly instead of rendering something else."""
    im = D.LazyImage(np.zeros((8, 8, 3), np.uint8))
    im.pad_to(32, 32, 0)
    with pytest.raises(NotImplementedError, match='normalize after pad'):
        im.normalize([0, 0, 0], [1, 1, 1], False)
    im = D.LazyImage(np.zeros((8, 8, 3), np.uint8))
    im.flip_('horizontal')
    im.flip_('vertical')                                                  # flips compose
    assert im.flip == 3
    with pytest.raises(NotImplementedError, match='resize after flip'):
        im.resize((16, 16))
    with pytest.raises(NotImplementedError):
        D.LoadImageFromFile(to_float32=True)
    with pytest.raises(RuntimeError, match='no CPU path'):
        D.LazyImage(np.zeros((8, 8, 3), np.uint8)).render(torch.zeros(3, 8, 8))
    with pytest.raises(RuntimeError, match='no CPU path'):
        D.collate_to_device([], 'cpu')


def test_oracle_resize_properties():
    """The fixed-point bilinear restatement: identity, constants, the 2x area route, and closeness to real-valued bilinear."""
    r = np.random.RandomState(2)
    img = r.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    np.testing.assert_array_equal(R.imresize_linear_u8(img, (53, 37)), img)
    const = np.full((20, 30, 3), 201, np.uint8)
    for size in ((45, 31), (17, 11), (30, 60)):
        assert (R.imresize_linear_u8(const, size) == 201).all()
    even = r.randint(0, 256, (40, 60, 3)).astype(np.uint8)
    a = even.astype(np.int64)
    np.testing.assert_array_equal(R.imresize_linear_u8(even, (30, 20)), ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2))
    up = R.imresize_linear_u8(img, (80, 56)).astype(np.float64)
    ys = np.clip((np.arange(56) + 0.5) * 37 / 56 - 0.5, 0, 36)
    xs = np.clip((np.arange(80) + 0.5) * 53 / 80 - 0.5, 0, 52)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, 36), np.minimum(x0 + 1, 52)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    f = img.astype(np.float64)
    ref = (f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx) * (1 - fy) + (f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx) * fy
    assert np.abs(up - ref).max() <= 1.0
    assert R.rescale_size((800, 800), (1200, 1200)) == (1200, 1200) and R.rescale_size((1000, 600), (800, 800)) == (800, 480)
    assert D.rescale_size((1000, 600), (800, 800)) == (800, 480) and D.rescale_size((333, 500), 1.5) == (500, 750)


def test_min_area_rect_and_sodaa_reader(tmp_path):
    r = np.random.RandomState(4)
    for _ in range(50):                                                   # a rotated rectangle is recovered
        cx, cy, w, h = r.uniform(50, 700), r.uniform(50, 700), r.uniform(4, 90), r.uniform(4, 90)
        a = r.uniform(-np.pi / 2, np.pi / 2)
        c, s = np.cos(a), np.sin(a)
        corners = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]]) @ np.array([[c, s], [-s, c]]) + [cx, cy]
        x, y, ww, hh, aa = D.poly2obb_np(corners.reshape(-1)[r.permutation(4).repeat(2) * 2 + np.tile([0, 1], 4)])
        wl, ws = max(w, h), min(w, h)
        al = a if w >= h else a + np.pi / 2
        al = (al + np.pi / 2) % np.pi - np.pi / 2
        assert abs(x - cx) < 1e-6 and abs(y - cy) < 1e-6 and abs(ww - wl) < 1e-6 and abs(hh - ws) < 1e-6
        assert min(abs(aa - al), np.pi - abs(aa - al)) < 1e-6 and -np.pi / 2 <= aa < np.pi / 2
    for _ in range(50):                                                   # arbitrary quads: product == oracle
        q = r.uniform(0, 100, (4, 2))
        (c1, s1, a1), (c2, s2, _) = D.min_area_rect(q), R.min_area_rect(q)
        assert abs(s1[0] * s1[1] - s2[0] * s2[1]) < 1e-6 * max(s2[0] * s2[1], 1)      # (a triangular hull has three equal minima)
        u = np.array([np.cos(np.radians(a1)), np.sin(np.radians(a1))])
        d = q - np.array(c1)
        assert (np.abs(d @ u) <= s1[0] / 2 + 1e-9).all() and (np.abs(d @ np.array([-u[1], u[0]])) <= s1[1] / 2 + 1e-9).all()
    assert D.poly2obb_np([0, 0, 1.5, 0, 1.5, 30, 0, 30]) is None          # a side under 2 px is dropped (transforms.py:406)
    # a SODA-A style folder: one json per patch; too-small and empty files are skipped like the reference does
    ann_dir, ori_dir = os.path.join(str(tmp_path), 'div'), os.path.join(str(tmp_path), 'raw')
    os.makedirs(ann_dir), os.makedirs(ori_dir)
    rect = lambda cx, cy, w, h: [cx - w / 2, cy - h / 2, cx + w / 2, cy - h / 2, cx + w / 2, cy + h / 2, cx - w / 2, cy + h / 2]   # noqa: E731
    json.dump(dict(annotations=[dict(poly=rect(100, 120, 40, 10), cat_id=2, trunc=0), dict(poly=rect(50, 60, 1, 30), cat_id=3, trunc=0),
                                dict(poly=rect(300, 310, 12, 36), cat_id=8, trunc=1)]), open(os.path.join(ann_dir, '00001__800__0___800.json'), 'w'))
    json.dump(dict(annotations=[]), open(os.path.join(ann_dir, '00001__800__800___0.json'), 'w'))
    open(os.path.join(ann_dir, '00002__800__0___0.json'), 'w').close()
    json.dump(dict(annotations=[dict(poly=rect(900, 920, 40, 10), category_id=2), dict(poly=list(range(12)), category_id=1)]),
              open(os.path.join(ori_dir, '00001.json'), 'w'))
    ds = D.build_dataset(dict(type='SODAADataset', ann_file=ann_dir, img_prefix='/nowhere', ori_ann_file=ori_dir, angle_version='le90',
                              pipeline=[dict(type='LoadAnnotations', with_bbox=True)]))
    assert len(ds) == 1 and ds.data_infos[0]['filename'] == '00001__800__0___800.jpg' and ds.flag.tolist() == [0]
    ann = ds.get_ann_info(0)
    np.testing.assert_allclose(ann['bboxes'], [[100, 120, 40, 10, 0], [300, 310, 36, 12, -np.pi / 2]], atol=1e-5)
    assert ann['labels'].tolist() == [2, 8] and ann['polygons'].shape == (2, 8) and ann['bboxes_ignore'].shape == (0, 5)
    assert ds.ori_img_ids == ['00001'] and ds.get_ori_ann_info(0)['bboxes'].shape == (1, 5)


def test_obb_config_pipeline_builds():
    """The shipped configs' pipeline lists build through the registries unchanged."""
    for rel in ('point_teacher/aitodv2_point_teacher_0.py', 'obb/point_teacher/sodaa_fcos_pointteacher_1x.py'):
        cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', rel))
        for split in ('train', 'test'):
            c = D.Compose(cfg.data[split]['pipeline'])
            assert len(c.transforms) >= 2
        assert cfg.data.train['type'] in D.DATASETS


# ------------------------------------------------------------------------------------------------ GPU
CASES = [
    # (src h, w), scale or None, keep_ratio, flip, normalise, size_divisor, channels_last
    ((61, 83), None, True, None, False, None, False),
    ((61, 83), (125, 125), True, None, True, 32, True),
    ((61, 83), (125, 125), True, 'horizontal', True, 32, False),
    ((97, 64), (300, 200), True, 'vertical', True, 32, True),
    ((97, 64), (50, 70), False, 'diagonal', True, 16, True),
    ((80, 120), (60, 40), True, 'horizontal', False, 32, True),               # exact 2x shrink -> the INTER_AREA route
    ((200, 200), (300, 300), True, None, True, 32, True),                     # the SODA-A ratio (800 -> 1200)
    ((333, 517), (731, 400), True, 'diagonal', True, 32, True),
    ((1, 1), (7, 5), False, None, True, 4, False),
]


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES)
def test_image_prep_bit_exact(case):
    (h, w), scale, keep, flip, norm, div, cl = case
    dev = torch.device('cuda:0')
    r = np.random.RandomState(h * 1000 + w)
    bgr = r.randint(0, 256, (h, w, 3)).astype(np.uint8)
    ncfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True) if norm else None
    want, meta = R.prepare_image(bgr, scale=scale, keep_ratio=keep, flip=flip, norm=ncfg, size_divisor=div, pad_val=0)
    im = D.LazyImage(bgr)
    if scale is not None:
        im.resize(D.rescale_size((w, h), scale) if keep else scale)
    if flip:
        im.flip_(flip)
    if norm:
        im.normalize(ncfg['mean'], ncfg['std'], True)
    if div:
        im.pad_to(int(np.ceil(im.shape[0] / div)) * div, int(np.ceil(im.shape[1] / div)) * div, 0)
    assert im.shape == meta['pad_shape']
    H, W = im.shape[0] + 3, im.shape[1] + 5                                  # a collate margin: must come back zero
    out = torch.full((1, 3, H, W), 7.0, device=dev).contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
    im.render(out[0])
    got = out[0].cpu().numpy()
    np.testing.assert_array_equal(got[:, :want.shape[1], :want.shape[2]], want.astype(np.float32))
    assert (got[:, want.shape[1]:, :] == 0).all() and (got[:, :, want.shape[2]:] == 0).all()
    if im.shape[1] % 4 == 0:           # exact-size channels-last target: the kernel's 16-byte store path
        out2 = torch.full((1, 3, im.shape[0], im.shape[1]), 7.0, device=dev).contiguous(memory_format=torch.channels_last)
        im.render(out2[0])
        np.testing.assert_array_equal(out2[0].cpu().numpy(), want.astype(np.float32))


@pytest.mark.gpu
def test_image_prep_pad_value_and_errors():
    dev = torch.device('cuda:0')
    from point_teacher_amd import hip
    bgr = np.arange(5 * 6 * 3, dtype=np.uint8).reshape(5, 6, 3)
    im = D.LazyImage(bgr)
    im.pad_to(8, 8, 114)
    out = torch.empty((3, 9, 9), device=dev)
    im.render(out)
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got[:, :5, :6], bgr.transpose(2, 0, 1).astype(np.float32))
    assert (got[:, 5:8, :8] == 114).all() and (got[:, :8, 6:8] == 114).all() and (got[:, 8] == 0).all() and (got[:, :, 8] == 0).all()
    src = torch.from_numpy(bgr).to(dev)
    with pytest.raises(RuntimeError, match='channels'):
        hip.call('pt_image_prep', src, 5, 6, 18, 1, 5, 6, 0, None, None, 0, 5, 6, 0.0, 5, 6, out.data_ptr(), 81, 9, 1)
    with pytest.raises(RuntimeError, match='padded'):
        hip.call('pt_image_prep', src, 5, 6, 18, 3, 5, 6, 0, None, None, 0, 4, 6, 0.0, 5, 6, out.data_ptr(), 81, 9, 1)
    with pytest.raises(RuntimeError, match='flip'):
        hip.call('pt_image_prep', src, 5, 6, 18, 3, 5, 6, 4, None, None, 0, 5, 6, 0.0, 5, 6, out.data_ptr(), 81, 9, 1)


@pytest.mark.gpu
def test_loader_batches_match_oracle_and_feed_a_training_step(tmp_path):
    """build_dataloader -> device batches == the reference's CPU pipeline + collate (through the oracle), then the batch
    goes straight into a Point-Teacher iteration; a second epoch reshuffles; test mode yields per-augmentation lists."""
    import test_train_step_parity as T
    dev = torch.device('cuda:0')
    ann_file, prefix, pix = _make_aitod(str(tmp_path), n=7, sizes=((160, 160), (128, 192), (160, 160), (96, 96), (160, 128)), seed=3)
    ds = D.build_dataset(dict(type='AITODDataset', ann_file=ann_file, img_prefix=prefix, pipeline=TRAIN_PIPELINE))
    np.random.seed(1)
    loader = D.build_dataloader(ds, samples_per_gpu=2, workers_per_gpu=2, dist=False, shuffle=True, device=dev)
    assert len(loader) == 3 and ds.flag.tolist() == [0, 1, 0, 0, 0, 1]       # groups of 4 and 2 -> batches never mix groups
    n_seen = 0
    by_name = {info['filename']: i for i, info in enumerate(ds.data_infos)}
    batches = list(loader)
    for batch in batches:
        img = batch['img']
        assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous(memory_format=torch.channels_last)
        exp = []
        for b, meta in enumerate(batch['img_metas']):
            chw, m, boxes = _expected_sample(pix[meta['ori_filename']], ds.get_ann_info(by_name[meta['ori_filename']]), meta)
            exp.append(chw)
            np.testing.assert_array_equal(batch['gt_bboxes'][b].cpu().numpy(), boxes)
            assert batch['gt_bboxes'][b].is_cuda and batch['gt_labels'][b].dtype == torch.int64
            n_seen += 1
        np.testing.assert_array_equal(img.cpu().numpy(), R.collate_images(exp))
    assert n_seen == 6
    # one Point-Teacher iteration fed by the loader (phase 1), finite losses.  AI-TOD tiles share one size (800x800) and the
    # hot path keeps the reference's one-size-per-batch assumption, so this loader reads a uniform folder
    uni = os.path.join(str(tmp_path), 'uniform')
    ann_u, prefix_u, _ = _make_aitod(uni, n=4, sizes=((160, 160),), seed=9)
    du = D.build_dataset(dict(type='AITODDataset', ann_file=ann_u, img_prefix=prefix_u, pipeline=TRAIN_PIPELINE))
    first = next(iter(D.build_dataloader(du, samples_per_gpu=2, workers_per_gpu=2, dist=False, shuffle=True, device=dev)))
    _, cfg, model = T._build(dev, phase2=False)
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config)
    lv = trainer.step(first)['log_vars'].materialize()
    assert all(v == v and abs(v) != float('inf') for v in lv.values()), lv
    # test mode: list per augmentation, sequential order, teacher inference + the AI-TOD evaluator on the results
    dt = D.build_dataset(dict(type='AITODDataset', ann_file=ann_file, img_prefix=prefix, pipeline=TEST_PIPELINE, test_mode=True))
    tl = D.build_dataloader(dt, samples_per_gpu=1, workers_per_gpu=1, dist=False, shuffle=False, device=dev)
    results = []
    model.eval()
    with torch.no_grad():
        for k, batch in enumerate(tl):
            assert isinstance(batch['img'], list) and batch['img'][0].shape[0] == 1
            assert batch['img_metas'][0][0]['ori_filename'] == dt.data_infos[k]['filename']
            results.extend(model(return_loss=False, rescale=True, img=batch['img'], img_metas=batch['img_metas']))
    assert len(results) == len(dt) == 7 and len(results[0]) == 8
    ev = dt.evaluate(results, device=dev)
    assert 'bbox_mAP' in ev and 'bbox_mAP_vt' in ev and len(ev['bbox_classwise']) == 8


@pytest.mark.gpu
def test_hbm_image_cache(tmp_path):
    """DeviceLoader(cache_bytes=...): the second epoch is served from HBM - the files can be gone - and renders the same
    pixels; a cache that is too small simply stops growing."""
    import shutil
    dev = torch.device('cuda:0')
    ann_file, prefix, pix = _make_aitod(str(tmp_path), n=6, sizes=((160, 160),), seed=5)
    pipe = [t for t in TRAIN_PIPELINE if t['type'] != 'RandomFlip']
    pipe.insert(3, dict(type='RandomFlip', flip_ratio=0.0))
    ds = D.build_dataset(dict(type='AITODDataset', ann_file=ann_file, img_prefix=prefix, pipeline=pipe))
    loader = D.build_dataloader(ds, samples_per_gpu=2, workers_per_gpu=2, dist=False, shuffle=False, device=dev, cache_bytes=1 << 30)
    first = {m['ori_filename']: b['img'][i].clone() for b in loader for i, m in enumerate(b['img_metas'])}
    assert len(first) == 5 and len(loader.cache) == 5 and loader.cache.hits == 0 and loader.cache.bytes == 5 * 160 * 160 * 3
    shutil.rmtree(prefix)                                                   # nothing left to decode
    second = {m['ori_filename']: b['img'][i].clone() for b in loader for i, m in enumerate(b['img_metas'])}
    assert loader.cache.hits == 5 and set(second) == set(first)
    for k in first:
        assert torch.equal(first[k], second[k])
    ann2, prefix2, _ = _make_aitod(os.path.join(str(tmp_path), 'b'), n=6, sizes=((160, 160),), seed=6)
    ds2 = D.build_dataset(dict(type='AITODDataset', ann_file=ann2, img_prefix=prefix2, pipeline=pipe))
    small = D.build_dataloader(ds2, samples_per_gpu=2, workers_per_gpu=1, dist=False, shuffle=False, device=dev, cache_bytes=2 * 160 * 160 * 3)
    for _ in small:
        pass
    assert len(small.cache) == 2


@pytest.mark.gpu
def test_runner_trains_from_loader_and_evaluates(tmp_path):
    """The whole outer loop on a generated AI-TOD folder: EpochBatches feeds the Runner from the device loader, the epoch
    ends with a checkpoint and an evaluation pass (mmdet's EvalHook) whose metrics land in log.json."""
    import test_train_step_parity as T
    from point_teacher_amd.runner import Runner
    dev = torch.device('cuda:0')
    ann_file, prefix, _ = _make_aitod(str(tmp_path), n=4, sizes=((160, 160),), seed=12)
    pipe = [dict(t) for t in TRAIN_PIPELINE]
    ds = D.build_dataset(dict(type='AITODDataset', ann_file=ann_file, img_prefix=prefix, pipeline=pipe))
    loader = D.build_dataloader(ds, samples_per_gpu=2, workers_per_gpu=2, dist=False, shuffle=True, device=dev, cache_bytes=1 << 28)
    dv = D.build_dataset(dict(type='AITODDataset', ann_file=ann_file, img_prefix=prefix, pipeline=TEST_PIPELINE, test_mode=True))
    vl = D.build_dataloader(dv, samples_per_gpu=1, workers_per_gpu=1, dist=False, shuffle=False, device=dev)
    _, cfg, model = T._build(dev, phase2=False)
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, iters_per_epoch=len(loader))
    work = os.path.join(str(tmp_path), 'work')
    runner = Runner(trainer, D.EpochBatches(loader), work, max_epochs=2, iters_per_epoch=len(loader), log_interval=1)
    runner.register_eval(dv, vl, interval=1, device=dev)
    assert runner.run() == 2 * len(loader) and runner.epoch == 2
    assert loader.cache.hits >= len(ds) and len(loader.cache) == len(ds)    # the second epoch came out of HBM (3 images padded to 2 batches)
    recs = [json.loads(l) for l in open(os.path.join(work, 'log.json'))]
    train, val = [r for r in recs if r['mode'] == 'train'], [r for r in recs if r['mode'] == 'val']
    assert len(train) == 2 * len(loader) and len(val) == 2 and 'bbox_mAP' in val[0] and val[1]['epoch'] == 2
    assert os.path.exists(os.path.join(work, 'epoch_2.pth')) and model.training


@pytest.mark.gpu
def test_image_prep_full_size_properties():
    """A 2160x3840 frame resized by a non-trivial factor (sizes the numpy oracle is not run at): flips commute with the
    rendering, normalisation is the affine map of the un-normalised rendering (same roundings), the resize of a constant
    is that constant and the pad / margin regions hold pad_val / 0."""
    dev = torch.device('cuda:0')
    r = np.random.RandomState(8)
    bgr = r.randint(0, 256, (2160, 3840, 3)).astype(np.uint8)
    size = D.rescale_size((3840, 2160), (2666, 1600))
    mean, std = np.array([123.675, 116.28, 103.53], np.float32), np.array([58.395, 57.12, 57.375], np.float32)

    def render(flip=None, norm=False, src=bgr):
        im = D.LazyImage(src)
        im.resize(size)
        if flip:
            im.flip_(flip)
        if norm:
            im.normalize(mean, std, True)
        im.pad_to(-(-size[1] // 32) * 32, -(-size[0] // 32) * 32, 3.0)
        out = torch.empty((1, 3, im.shape[0] + 32, im.shape[1]), device=dev).contiguous(memory_format=torch.channels_last)
        im.render(out[0])
        return out[0]
    base = render()
    h, w = size[1], size[0]
    assert float(base[:, :h, :w].min()) >= 0 and float(base[:, :h, :w].max()) <= 255 and bool((base[:, :h, :w] == base[:, :h, :w].round()).all())
    assert bool((base[:, h:-32, :] == 3.0).all()) and bool((base[:, :-32, w:] == 3.0).all()) and bool((base[:, -32:] == 0).all())
    for flip, dims in (('horizontal', [2]), ('vertical', [1]), ('diagonal', [1, 2])):
        assert torch.equal(render(flip)[:, :h, :w], torch.flip(base[:, :h, :w], dims))
    want = ((base[:, :h, :w].flip(0) - torch.from_numpy(mean).to(dev)[:, None, None]).double()
            * torch.from_numpy(1 / np.float64(std)).to(dev)[:, None, None]).float()
    assert torch.equal(render(norm=True)[:, :h, :w], want)
    const = render(src=np.full_like(bgr, 77))
    assert bool((const[:, :h, :w] == 77).all())
    # bilinear interpolation preserves the mean up to rounding
    assert abs(float(base[:, :h, :w].mean()) - float(bgr.mean())) < 0.5


@pytest.mark.gpu
def test_sodaa_folder_feeds_the_oriented_config(tmp_path):
    """Config 5 end to end on a generated SODA-A style folder (JPEG patches + one json per patch): SODAADOTADataset ->
    the config's own train pipeline at a reduced scale (RResize, RRandomFlip, Normalize, Pad) -> device loader -> one
    RotatedFCOS_TS iteration; then the test pipeline (DefaultFormatBundle inside MultiScaleFlipAug) -> teacher inference ->
    the dataset's DOTA-style evaluation."""
    from PIL import Image
    import test_obb_parity as TO
    dev = torch.device('cuda:0')
    r = np.random.RandomState(21)
    ann_dir, img_dir = os.path.join(str(tmp_path), 'ann'), os.path.join(str(tmp_path), 'img')
    os.makedirs(ann_dir), os.makedirs(img_dir)
    for k in range(4):
        name = f'{k:05d}__800__{k * 600}___0'
        Image.fromarray(r.randint(0, 256, (160, 160, 3)).astype(np.uint8)).save(os.path.join(img_dir, name + '.jpg'), quality=95)
        anns = []
        for _ in range(7):
            cx, cy, w, h, a = r.uniform(30, 130), r.uniform(30, 130), r.uniform(8, 30), r.uniform(4, 12), r.uniform(-1.4, 1.4)
            c, s = np.cos(a), np.sin(a)
            pts = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]]) @ np.array([[c, s], [-s, c]]) + [cx, cy]
            anns.append(dict(poly=[float(v) for v in pts.reshape(-1)], cat_id=int(r.randint(0, 9)), trunc=0))
        json.dump(dict(annotations=anns), open(os.path.join(ann_dir, name + '.json'), 'w'))
    cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'obb', 'point_teacher', 'sodaa_fcos_pointteacher_1x.py'))
    tp = [dict(t) for t in cfg.data.train['pipeline']]
    for t in tp:
        if t['type'] == 'RResize':
            t['img_scale'] = (240, 240)                                    # the config's 1200 at test size
    ds = D.build_dataset(dict(type=cfg.data.train['type'], ann_file=ann_dir, img_prefix=img_dir, pipeline=tp))
    assert type(ds).__name__ == 'SODAADOTADataset' and len(ds) == 4 and ds[0]['gt_bboxes'].data.shape[1] == 5
    loader = D.build_dataloader(ds, samples_per_gpu=2, workers_per_gpu=2, dist=False, shuffle=True, device=dev)
    batch = next(iter(loader))
    assert batch['img'].shape == (2, 3, 256, 256) and batch['gt_bboxes'][0].shape[1] == 5 and 'filename' in batch['img_metas'][0]
    sf = batch['img_metas'][0]['scale_factor']
    np.testing.assert_allclose(sf, [1.5, 1.5, 1.5, 1.5])
    # the rendered pixels = the oracle's pipeline on the decoded JPEG
    from PIL import Image as _I
    m0 = batch['img_metas'][0]
    bgr = np.asarray(_I.open(m0['filename']).convert('RGB'))[:, :, ::-1].copy()
    want, _ = R.prepare_image(bgr, scale=(240, 240), keep_ratio=True, flip=None,
                              norm=dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True), size_divisor=32)
    np.testing.assert_array_equal(batch['img'][0].cpu().numpy(), want)
    _, mcfg, model = TO._build(dev, phase2=False)
    trainer = pta.Trainer(model, mcfg.optimizer, mcfg.optimizer_config, mcfg.lr_config)
    lv = trainer.step(batch)['log_vars'].materialize()
    assert all(v == v and abs(v) != float('inf') for v in lv.values()), lv
    vp = [dict(t) for t in cfg.data.test['pipeline']]
    vp[1] = dict(vp[1], img_scale=(240, 240))
    dv = D.build_dataset(dict(type=cfg.data.test['type'], ann_file=ann_dir, img_prefix=img_dir, pipeline=vp, test_mode=True))
    vl = D.build_dataloader(dv, samples_per_gpu=1, workers_per_gpu=1, dist=False, shuffle=False, device=dev)
    model.eval()
    results = []
    with torch.no_grad():
        for b in vl:
            results.extend(model(return_loss=False, rescale=True, img=b['img'], img_metas=b['img_metas']))
    assert len(results) == 4 and len(results[0]) == 9 and all(r.shape[1] == 6 for r in results[0])
    ev = dv.evaluate(results, device=dev)
    assert set(ev) == {'mAP'} and 0.0 <= ev['mAP'] <= 1.0


def test_host_side_helpers():
    """CPU-only pieces of the loader: the HBM cache bookkeeping (with CPU tensors standing in), the epoch adapter and the
    lazy image's shape / dtype answers along the pipeline."""
    cache = D.DeviceImageCache(max_bytes=100)
    a, b = torch.zeros(60, dtype=torch.uint8), torch.zeros(60, dtype=torch.uint8)
    assert cache.get('a') is None and cache.misses == 1
    assert cache.put('a', a) and not cache.put('a', a) and not cache.put('b', b)         # duplicate key / over budget
    assert cache.get('a') is a and cache.hits == 1 and len(cache) == 1 and cache.bytes == 60

    class FakeLoader:
        def __init__(self):
            self.epochs = []

        def set_epoch(self, e):
            self.epochs.append(e)

        def __iter__(self):
            return iter([f'e{self.epochs[-1]}b0', f'e{self.epochs[-1]}b1'])
    fl = FakeLoader()
    nxt = D.EpochBatches(fl, start_epoch=3)
    assert [nxt(i, 2) for i in range(5)] == ['e3b0', 'e3b1', 'e4b0', 'e4b1', 'e5b0'] and fl.epochs == [3, 4, 5]

    im = D.LazyImage(np.zeros((30, 50, 3), np.uint8))
    assert im.shape == (30, 50, 3) and im.dtype == np.uint8
    im.resize((75, 45))
    assert im.shape == (45, 75, 3)
    im.flip_('horizontal')
    im.normalize([1, 2, 3], [4, 5, 6], True)
    assert im.dtype == np.float32 and im.shape == (45, 75, 3)
    im.pad_to(64, 96, dict(img=7))
    assert im.shape == (64, 96, 3) and im.pad == (64, 96, 7.0)
    c = im.copy()
    c.flip = 0
    assert im.flip == 1 and c.src is im.src                                  # augmentations share the pixels, not the state
    with pytest.raises(TypeError):
        D.LazyImage(np.zeros((4, 4), np.uint8))
    with pytest.raises(TypeError):
        D.LazyImage(torch.zeros(4, 4, 3, dtype=torch.uint8))                 # a cached image must live on the GPU


def test_reference_pipeline_known_answers():
    """The known answers the reference's own pipeline tests hold (HBB_TOD/tests/test_data/test_pipelines/test_transform/
    test_transform.py: test_resize :14-80, test_flip :82-181, test_pad :356-393, test_normalize :396-422,
    test_multi_scale_flip_aug :595-686), on a generated image of the size of their `color.jpg` (288x512 - the file itself is
    not in the reference snapshot; every expected number below depends on the size only)."""
    import copy
    build = lambda cfg: D.PIPELINES.build(cfg)                    # noqa: E731
    # constructor contracts
    for bad in (dict(type='Resize', img_scale=[1333, 800], keep_ratio=True),
                dict(type='Resize', img_scale=[(1333, 800), (1333, 600)], ratio_range=(0.9, 1.1), keep_ratio=True),
                dict(type='Resize', img_scale=[(1333, 800), (1333, 600)], keep_ratio=True, multiscale_mode='2333'),
                dict(type='RandomFlip', flip_ratio=1.5),
                dict(type='RandomFlip', flip_ratio=[0.7, 0.8], direction=['horizontal', 'vertical']),
                dict(type='RandomFlip', flip_ratio=[0.4, 0.5]),
                dict(type='RandomFlip', flip_ratio=1., direction='horizonta'),
                dict(type='Pad'),
                dict(type='MultiScaleFlipAug', scale_factor=1.0, img_scale=[(1333, 800)], transforms=[dict(type='Resize')]),
                dict(type='MultiScaleFlipAug', scale_factor=None, img_scale=None, transforms=[dict(type='Resize')]),
                dict(type='MultiScaleFlipAug', img_scale=[1333, 800], transforms=[dict(type='Resize')]),
                dict(type='MultiScaleFlipAug', img_scale=[(1333, 800)], flip_direction=1, transforms=[dict(type='Resize')])):
        with pytest.raises(AssertionError):
            build(bad)
    r = np.random.RandomState(0)
    pix = r.randint(0, 256, (288, 512, 3)).astype(np.uint8)

    def fresh(fields=('img',)):
        res = dict(img_shape=pix.shape, ori_shape=pix.shape, pad_shape=pix.shape, img_fields=list(fields))
        for f in fields:
            res[f] = D.LazyImage(pix)
        return res
    # test_resize: scale and scale_factor cannot both be set; keep_ratio=False takes the scale as is
    res = fresh()
    res['scale'], res['scale_factor'] = (1333, 800), 1.0
    with pytest.raises(AssertionError):
        build(dict(type='Resize', img_scale=(1333, 800), keep_ratio=True))(res)
    res = build(dict(type='Resize', img_scale=(1333, 800), keep_ratio=True))(fresh(('img', 'img2')))
    assert res['img'].shape == res['img2'].shape == (750, 1333, 3) and res['img_shape'] == (750, 1333, 3)
    res = build(dict(type='Resize', img_scale=(1280, 800), multiscale_mode='value', keep_ratio=False))(fresh())
    assert res['img_shape'] == (800, 1280, 3)
    # test_flip: two flips restore the image state; the drawn direction is recorded
    flip = build(dict(type='RandomFlip', flip_ratio=1.))
    res = dict(fresh(), scale_factor=1.0)
    res = flip(res)
    assert res['flip'] and res['flip_direction'] == 'horizontal' and res['img'].flip == 1
    res.pop('flip'), res.pop('flip_direction')
    res = build(dict(type='RandomFlip', flip_ratio=1.))(res)
    assert res['img'].flip == 0
    for cfg in (dict(type='RandomFlip', flip_ratio=0.9, direction=['horizontal', 'vertical', 'diagonal']),
                dict(type='RandomFlip', flip_ratio=[0.3, 0.3, 0.2], direction=['horizontal', 'vertical', 'diagonal'])):
        res = build(cfg)(dict(fresh(), scale_factor=1.0))
        assert res['img'].flip == ({None: 0, 'horizontal': 1, 'vertical': 2, 'diagonal': 3}[res['flip_direction']] if res['flip'] else 0)
    # test_pad: already divisible -> unchanged; after the resize -> the next multiples of 32
    pad = build(dict(type='Pad', size_divisor=32))
    res = pad(dict(fresh(), scale_factor=1.0))
    assert res['img'].shape == (288, 512, 3) and res['pad_shape'] == (288, 512, 3)
    res = pad(build(dict(type='Resize', img_scale=(1333, 800), keep_ratio=True))(fresh()))
    assert res['img'].shape == (768, 1344, 3) and res['img'].shape[0] % 32 == 0 and res['img'].shape[1] % 32 == 0
    # test_normalize: the recorded configuration (the arithmetic itself is checked on the GPU against the oracle)
    res = build(dict(type='Normalize', mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True))(fresh())
    assert res['img'].dtype == np.float32 and res['img_norm_cfg']['to_rgb'] is True
    np.testing.assert_allclose(res['img_norm_cfg']['mean'], [123.675, 116.28, 103.53], rtol=1e-6)
    # test_multi_scale_flip_aug
    ms = build(dict(type='MultiScaleFlipAug', img_scale=[(1333, 800), (1333, 640)], transforms=[dict(type='Resize', keep_ratio=True)]))
    out = ms(copy.copy(fresh()))
    assert len(out['img']) == 2 and out['img'][0].shape == (750, 1333, 3) and out['img_shape'][0] == (750, 1333, 3)
    assert out['img'][1].shape == (640, 1138, 3) and out['img_shape'][1] == (640, 1138, 3)
    ms = build(dict(type='MultiScaleFlipAug', scale_factor=[0.8, 1.0, 1.2], transforms=[dict(type='Resize', keep_ratio=False)]))
    out = ms(copy.copy(fresh()))
    assert [im.shape for im in out['img']] == [(230, 409, 3), (288, 512, 3), (345, 614, 3)]
    assert out['img_shape'] == [(230, 409, 3), (288, 512, 3), (345, 614, 3)]
    # the coco_detection test pipeline on the same image size (:668-685): shapes and the float32 scale factor
    ms = build(dict(type='MultiScaleFlipAug', img_scale=(1333, 800), flip=False,
                    transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'),
                                dict(type='Normalize', mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
                                dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']),
                                dict(type='Collect', keys=['img'])]))
    res = dict(fresh(), filename='color.jpg', ori_filename='color.jpg')
    out = ms(res)
    assert len(out['img']) == 1 and len(out['img_metas']) == 1 and isinstance(out['img_metas'][0], D.DataContainer)
    meta = out['img_metas'][0].data
    assert meta['ori_shape'] == (288, 512, 3) and meta['img_shape'] == (750, 1333, 3) and meta['pad_shape'] == (768, 1344, 3)
    assert meta['scale_factor'].tolist() == [2.603515625, 2.6041667461395264, 2.603515625, 2.6041667461395264]


def test_reference_dataset_known_answers(tmp_path):
    """The dataset tests the reference holds (HBB_TOD/tests/test_data/test_datasets/test_coco_dataset.py:10-58 duplicate
    annotation ids; test_custom_dataset.py:18-100 `classes` as tuple / list / file / None; test_common.py:335-360 empty-image
    filtering per class subset - on a generated stand-in for their missing `coco_sample.json`: 3 images, bus in two, car in one)."""
    def write(name, images, anns, cats):
        p = os.path.join(str(tmp_path), name)
        json.dump(dict(images=images, annotations=anns, categories=cats), open(p, 'w'))
        return p
    bad = write('dup.json', [dict(id=0, width=640, height=640, file_name='fake_name.jpg')],
                [dict(id=1, image_id=0, category_id=0, area=400, bbox=[50, 60, 20, 20], iscrowd=0),
                 dict(id=1, image_id=0, category_id=0, area=900, bbox=[100, 120, 30, 30], iscrowd=0)],
                [dict(id=0, name='car', supercategory='car')])
    with pytest.raises(AssertionError):
        D.CocoDataset(ann_file=bad, classes=('car',), pipeline=[])
    imgs = [dict(id=i, width=100, height=80, file_name=f'{i}.jpg') for i in range(3)]
    cats = [dict(id=1, name='bus', supercategory='v'), dict(id=2, name='car', supercategory='v')]
    anns = [dict(id=1, image_id=0, category_id=1, area=100, bbox=[10, 10, 10, 10], iscrowd=0),
            dict(id=2, image_id=1, category_id=1, area=100, bbox=[20, 10, 10, 10], iscrowd=0),
            dict(id=3, image_id=1, category_id=2, area=100, bbox=[30, 10, 10, 10], iscrowd=0)]
    sample = write('sample.json', imgs, anns, cats)
    for classes, expected in ((['bus'], 2), (['car'], 1), (['bus', 'car'], 2)):
        filtered = D.CocoDataset(ann_file=sample, img_prefix='', pipeline=[], classes=classes, filter_empty_gt=True)
        full = D.CocoDataset(ann_file=sample, img_prefix='', pipeline=[], classes=classes, filter_empty_gt=False)
        assert len(filtered) == len(filtered.img_ids) == expected and len(full) == len(full.img_ids) == 3
        assert filtered.CLASSES == classes and full.CLASSES == classes
    # `classes`: tuple, list, not a subset of the defaults, None (defaults), a file with one name per line
    mk = lambda c: D.AITODDataset(ann_file=sample, pipeline=[], classes=c, test_mode=True)      # noqa: E731
    assert mk(('bus', 'car')).CLASSES == ('bus', 'car') and mk(['bus', 'car']).CLASSES == ['bus', 'car'] and mk(['foo']).CLASSES == ['foo']
    assert mk(None).CLASSES == D.AITODDataset.CLASSES
    names = os.path.join(str(tmp_path), 'classes.txt')
    open(names, 'w').write('bus\ncar\n')
    assert mk(names).CLASSES == ['bus', 'car']
    with pytest.raises(ValueError):
        mk(0.5)
