"""Dataset readers of the Point-Teacher configs: the annotation side is host work in the
reference and stays host work here; the samples they emit carry `LazyImage`s that the batch collate renders on the GPU.

    CustomDataset     /root/reference/HBB_TOD/mmdet/datasets/custom.py:16-361
    CocoDataset       /root/reference/HBB_TOD/mmdet/datasets/coco.py:22-200
    AITODDataset      /root/reference/HBB_TOD/mmdet/datasets/aitod.py:17-247      (configs 1-4: AI-TOD-v2, COCO json)
    SODAADataset      /root/reference/OBB_TOD/mmrotate/datasets/sodaa.py:32-520   (config 5: SODA-A, one json per patch)
    SODAADOTADataset  /root/reference/OBB_TOD/mmrotate/datasets/sodaa_rewrite.py:52-378
"""
import glob
import json
import os
import os.path as osp
import re
import warnings
from collections import OrderedDict, defaultdict

import numpy as np

from ..registry import Registry, build_from_cfg
from .coco_api import COCO
from .pipelines import Compose

DATASETS = Registry('dataset')
ROTATED_DATASETS = DATASETS


def build_dataset(cfg, default_args=None):
    """datasets/builder.py build_dataset for the plain (non-wrapped) datasets the Point-Teacher configs use."""
    if isinstance(cfg, (list, tuple)) or cfg.get('type') in ('ConcatDataset', 'RepeatDataset', 'ClassBalancedDataset'):
        raise NotImplementedError('dataset wrappers are not used by the Point-Teacher configs')
    return build_from_cfg(dict(cfg), DATASETS, default_args)


class CustomDataset:
    CLASSES = None

    def __init__(self, ann_file, pipeline, classes=None, data_root=None, img_prefix='', seg_prefix=None, proposal_file=None,
                 test_mode=False, filter_empty_gt=True):
        self.ann_file, self.data_root, self.img_prefix, self.seg_prefix = ann_file, data_root, img_prefix, seg_prefix
        self.proposal_file, self.test_mode, self.filter_empty_gt = proposal_file, test_mode, filter_empty_gt
        self.CLASSES = self.get_classes(classes)
        if proposal_file is not None:
            raise NotImplementedError('proposal files are not on the Point-Teacher path')
        if self.data_root is not None:
            if not osp.isabs(self.ann_file):
                self.ann_file = osp.join(self.data_root, self.ann_file)
            if not (self.img_prefix is None or osp.isabs(self.img_prefix)):
                self.img_prefix = osp.join(self.data_root, self.img_prefix)
        self.data_infos = self.load_annotations(self.ann_file)
        self.proposals = None
        if not test_mode:
            valid_inds = self._filter_imgs()
            self.data_infos = [self.data_infos[i] for i in valid_inds]
            self._set_group_flag()
        self.pipeline = Compose(pipeline)

    def __len__(self):
        return len(self.data_infos)

    def load_annotations(self, ann_file):
        with open(ann_file) as f:
            return json.load(f)

    def get_ann_info(self, idx):
        return self.data_infos[idx]['ann']

    def get_cat_ids(self, idx):
        return self.data_infos[idx]['ann']['labels'].astype(np.int64).tolist()

    def pre_pipeline(self, results):
        results['img_prefix'] = self.img_prefix
        results['seg_prefix'] = self.seg_prefix
        results['proposal_file'] = self.proposal_file
        results['bbox_fields'] = []
        results['mask_fields'] = []
        results['seg_fields'] = []

    def _filter_imgs(self, min_size=32):
        if self.filter_empty_gt:
            warnings.warn('CustomDataset does not support filtering empty gt images.')
        return [i for i, info in enumerate(self.data_infos) if min(info['width'], info['height']) >= min_size]

    def _set_group_flag(self):
        """custom.py:163-173: aspect ratio > 1 -> group 1."""
        self.flag = np.zeros(len(self), dtype=np.uint8)
        for i in range(len(self)):
            info = self.data_infos[i]
            if info['width'] / info['height'] > 1:
                self.flag[i] = 1

    def _rand_another(self, idx):
        pool = np.where(self.flag == self.flag[idx])[0]
        return np.random.choice(pool)

    def __getitem__(self, idx):
        if self.test_mode:
            return self.prepare_test_img(idx)
        while True:
            data = self.prepare_train_img(idx)
            if data is None:
                idx = self._rand_another(idx)
                continue
            return data

    def prepare_train_img(self, idx):
        results = dict(img_info=self.data_infos[idx], ann_info=self.get_ann_info(idx))
        self.pre_pipeline(results)
        return self.pipeline(results)

    def prepare_test_img(self, idx):
        results = dict(img_info=self.data_infos[idx])
        self.pre_pipeline(results)
        return self.pipeline(results)

    @classmethod
    def get_classes(cls, classes=None):
        if classes is None:
            return cls.CLASSES
        if isinstance(classes, str):
            with open(classes) as f:
                return [line.rstrip('\n') for line in f]
        if isinstance(classes, (tuple, list)):
            return classes
        raise ValueError(f'Unsupported type {type(classes)} of classes.')


@DATASETS.register_module()
class CocoDataset(CustomDataset):

    def load_annotations(self, ann_file):
        self.coco = COCO(ann_file)
        self.cat_ids = self.coco.get_cat_ids(cat_names=self.CLASSES)
        self.cat2label = {cat_id: i for i, cat_id in enumerate(self.cat_ids)}
        self.img_ids = self.coco.get_img_ids()
        data_infos, total_ann_ids = [], []
        for i in self.img_ids:
            info = self.coco.load_imgs([i])[0]
            info['filename'] = info['file_name']
            data_infos.append(info)
            total_ann_ids.extend(self.coco.get_ann_ids(img_ids=[i]))
        assert len(set(total_ann_ids)) == len(total_ann_ids), f"Annotation ids in '{ann_file}' are not unique!"
        return data_infos

    def get_ann_info(self, idx):
        img_id = self.data_infos[idx]['id']
        ann_info = self.coco.load_anns(self.coco.get_ann_ids(img_ids=[img_id]))
        return self._parse_ann_info(self.data_infos[idx], ann_info)

    def get_cat_ids(self, idx):
        img_id = self.data_infos[idx]['id']
        return [ann['category_id'] for ann in self.coco.load_anns(self.coco.get_ann_ids(img_ids=[img_id]))]

    def _filter_imgs(self, min_size=32):
        """coco.py:96-122: drop images that are too small or (filter_empty_gt) hold no box of the wanted categories."""
        ids_with_ann = set(a['image_id'] for a in self.coco.anns.values())
        ids_in_cat = set()
        for class_id in self.cat_ids:
            ids_in_cat |= set(self.coco.cat_img_map[class_id])
        ids_in_cat &= ids_with_ann
        valid_inds, valid_img_ids = [], []
        for i, info in enumerate(self.data_infos):
            img_id = self.img_ids[i]
            if self.filter_empty_gt and img_id not in ids_in_cat:
                continue
            if min(info['width'], info['height']) >= min_size:
                valid_inds.append(i)
                valid_img_ids.append(img_id)
        self.img_ids = valid_img_ids
        return valid_inds

    def _parse_ann_info(self, img_info, ann_info):
        """coco.py:124-180 over the whole annotation table at once: an annotation survives when it is not flagged `ignore`, its
        box overlaps the image, has a positive `area` and at least 1 px of width and height, and its category is one of the
        dataset's; crowd boxes go to `bboxes_ignore`, the rest to `bboxes` / `labels` / `masks` (order preserved)."""
        anns = [a for a in ann_info if not a.get('ignore', False)]
        box = np.array([a['bbox'] for a in anns], dtype=np.float64).reshape(-1, 4)
        x1, y1, w, h = box.T
        overlap_w = np.clip(np.minimum(x1 + w, img_info['width']) - np.maximum(x1, 0), 0, None)
        overlap_h = np.clip(np.minimum(y1 + h, img_info['height']) - np.maximum(y1, 0), 0, None)
        keep = (overlap_w * overlap_h != 0) & (w >= 1) & (h >= 1)
        keep &= np.array([a['area'] > 0 and a['category_id'] in self.cat2label for a in anns], dtype=bool).reshape(-1)
        crowd = np.array([bool(a.get('iscrowd', False)) for a in anns], dtype=bool).reshape(-1)
        xyxy = np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
        real = np.flatnonzero(keep & ~crowd)
        return dict(bboxes=xyxy[real].reshape(-1, 4),
                    labels=np.array([self.cat2label[anns[i]['category_id']] for i in real], dtype=np.int64),
                    bboxes_ignore=xyxy[keep & crowd].reshape(-1, 4),
                    masks=[anns[i].get('segmentation', None) for i in real],
                    seg_map=img_info['filename'].replace('jpg', 'png'))

    @staticmethod
    def xyxy2xywh(bbox):
        b = bbox.tolist()
        return [b[0], b[1], b[2] - b[0], b[3] - b[1]]

    def _det2json(self, results):
        out = []
        for idx in range(len(self)):
            img_id = self.img_ids[idx]
            for label, bboxes in enumerate(results[idx]):
                for i in range(bboxes.shape[0]):
                    out.append(dict(image_id=img_id, bbox=self.xyxy2xywh(bboxes[i]), score=float(bboxes[i][4]),
                                    category_id=self.cat_ids[label]))
        return out

    def results2json(self, results, outfile_prefix):
        if not isinstance(results[0], list):
            raise TypeError('invalid type of results (only box detections are on the Point-Teacher path)')
        files = dict(bbox=f'{outfile_prefix}.bbox.json', proposal=f'{outfile_prefix}.bbox.json')
        with open(files['bbox'], 'w') as f:
            json.dump(self._det2json(results), f)
        return files

    def format_results(self, results, jsonfile_prefix=None, **kwargs):
        assert isinstance(results, list), 'results must be a list'
        assert len(results) == len(self), f'The length of results is not equal to the dataset len: {len(results)} != {len(self)}'
        tmp_dir = None
        if jsonfile_prefix is None:
            import tempfile
            tmp_dir = tempfile.TemporaryDirectory()
            jsonfile_prefix = osp.join(tmp_dir.name, 'results')
        return self.results2json(results, jsonfile_prefix), tmp_dir

    def coco_ground_truth(self):
        """What COCOeval._prepare reads for this dataset: per image, every annotation of the wanted categories with
        its json `area`, `iscrowd` and `ignore` (boxes as xyxy)."""
        gts = []
        for img_id in self.img_ids:
            anns = self.coco.load_anns(self.coco.get_ann_ids(img_ids=[img_id], cat_ids=self.cat_ids))
            b = np.array([a['bbox'] for a in anns], dtype=np.float64).reshape(-1, 4)
            gts.append(dict(bboxes=np.stack([b[:, 0], b[:, 1], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]], 1).astype(np.float32),
                            labels=np.array([self.cat2label[a['category_id']] for a in anns], dtype=np.int64),
                            area=np.array([a['area'] for a in anns], dtype=np.float32),
                            iscrowd=np.array([a.get('iscrowd', 0) for a in anns], dtype=bool),
                            ignore=np.array([a.get('ignore', 0) for a in anns], dtype=bool)))
        return gts


@DATASETS.register_module()
class AITODDataset(CocoDataset):
    CLASSES = ('airplane', 'bridge', 'storage-tank', 'ship', 'swimming-pool', 'vehicle', 'person', 'wind-mill')
    # aitod.py:64 overwrites the caller's `iou_thrs` with [0.25] (a leftover of the authors' analysis runs); the
    # class attribute keeps that behaviour switchable: set it to None for the AI-TOD protocol (0.50:0.05:0.95).
    FORCED_IOU_THRS = [0.25]

    def evaluate(self, results, metric='bbox', logger=None, jsonfile_prefix=None, classwise=True, classwise_lrp=True,
                 proposal_nums=(100, 300, 1500), iou_thrs=None, metric_items=None, with_lrp=False, device='cuda'):
        """aitod.py:23-247 for `metric='bbox'` through evaluation.AITODEvaluator (the GPU COCOeval of row N1)."""
        from ..evaluation import AITODEvaluator
        metrics = metric if isinstance(metric, list) else [metric]
        for m in metrics:
            if m not in ['bbox', 'segm', 'proposal', 'proposal_fast']:
                raise KeyError(f'metric {m} is not supported')
            if m != 'bbox':
                raise NotImplementedError(f"metric '{m}': the Point-Teacher configs evaluate metric='bbox'")
        if with_lrp:
            raise NotImplementedError('oLRP of the aitodpycocotools fork is not restated')
        if self.FORCED_IOU_THRS is not None:
            iou_thrs = self.FORCED_IOU_THRS
        assert len(results) == len(self)
        ev = AITODEvaluator(self.coco_ground_truth(), len(self.CLASSES), device=device, max_dets=proposal_nums, iou_thrs=iou_thrs)
        out = ev.evaluate(results)
        eval_results = OrderedDict((k, v) for k, v in out.items() if k.startswith('bbox_'))
        if metric_items is not None:
            keep = set('bbox_' + m for m in ([metric_items] if isinstance(metric_items, str) else metric_items))
            eval_results = OrderedDict((k, v) for k, v in eval_results.items() if k in keep)
        if classwise:
            prec = out['precision']                                  # [T, R, K, A, M]
            per = []
            for k, name in enumerate(self.CLASSES):
                p = prec[:, :, k, 0, -1]
                p = p[p > -1]
                per.append((name, float(p.mean()) if p.numel() else float('nan')))
            eval_results['bbox_classwise'] = per
        return eval_results


# ------------------------------------------------------------------------------------------ SODA-A (oriented)
def min_area_rect(points):
    """((cx, cy), (w, h), angle_deg) of the smallest enclosing rectangle - what cv2.minAreaRect returns up to the
    rectangle's own symmetries (the callers normalise to le90).  All candidate directions are evaluated at once."""
    q = np.unique(np.asarray(points, np.float64).reshape(-1, 2), axis=0)
    if len(q) == 1:
        return (q[0, 0], q[0, 1]), (0.0, 0.0), 0.0
    i, j = np.triu_indices(len(q), 1)                        # every point pair: a superset of the hull edges, one of
    e = q[j] - q[i]                                          # which carries a side of the smallest rectangle
    n = np.hypot(e[:, 0], e[:, 1])
    u = e / n[:, None]                                       # [E, 2]
    v = np.stack([-u[:, 1], u[:, 0]], 1)
    pu, pv = q @ u.T, q @ v.T                                # [P, E]
    w, h = pu.max(0) - pu.min(0), pv.max(0) - pv.min(0)
    k = int(np.argmin(w * h))
    c = u[k] * (pu[:, k].max() + pu[:, k].min()) / 2 + v[k] * (pv[:, k].max() + pv[:, k].min()) / 2
    return (c[0], c[1]), (w[k], h[k]), float(np.degrees(np.arctan2(u[k, 1], u[k, 0])))


def poly2obb_np(poly, version='le90', min_side=2):
    """mmrotate/core/bbox/transforms.py:393-418 (le90); `min_side=None` = SODAADataset.poly2obb (no size test)."""
    if version != 'le90':
        raise NotImplementedError(f"angle version '{version}': the SODA-A config uses le90")
    (x, y), (w, h), a = min_area_rect(np.array(poly).reshape((4, 2)))
    if min_side is not None and (w < min_side or h < min_side):
        return None
    a = a / 180 * np.pi
    if w < h:
        w, h = h, w
        a += np.pi / 2
    while not np.pi / 2 > a >= -np.pi / 2:
        if a >= np.pi / 2:
            a -= np.pi
        else:
            a += np.pi
    return x, y, w, h, a


def _load_sodaa_folder(ann_folder, angle_version, label_key, skip_long_polys=False):
    """The per-patch json reader shared by load_annotations / load_ori_annotations (sodaa.py:60-125, :139-208)."""
    data_infos = []
    for ann_file in glob.glob(ann_folder + '/*.json'):
        if os.path.getsize(ann_file) == 0:
            continue
        info = dict(filename=ann_file.replace('.json', '.jpg').split(os.sep)[-1], ann={})
        with open(ann_file, 'r') as f:
            annotations = json.load(f)['annotations']
        boxes, labels, polys = [], [], []
        for ann in annotations:
            poly = np.array(ann['poly'], dtype=np.float32)
            if skip_long_polys and len(poly) > 8:
                continue
            try:
                x, y, w, h, a = poly2obb_np(poly, angle_version)
            except Exception:                                # the reference's bare `except: continue` (also eats None)
                continue
            labels.append(int(ann[label_key]))
            if label_key == 'cat_id':
                int(ann['trunc'])                            # sodaa.py:168 - a patch annotation without `trunc` raises
            boxes.append([x, y, w, h, a])
            polys.append(poly)
        if boxes:
            info['ann'].update(bboxes=np.array(boxes, dtype=np.float32), labels=np.array(labels, dtype=np.int64),
                               polygons=np.array(polys, dtype=np.float32))
        else:
            info['ann'].update(bboxes=np.zeros((0, 5), dtype=np.float32), labels=np.array([], dtype=np.int64),
                               polygons=np.zeros((0, 8), dtype=np.float32))
        info['ann'].update(bboxes_ignore=np.zeros((0, 5), dtype=np.float32), labels_ignore=np.array([], dtype=np.int64),
                           polygons_ignore=np.zeros((0, 8), dtype=np.float32))
        data_infos.append(info)
    return data_infos


def _patch_origin(filename):
    """'<ori>__<x>___<y>.jpg' -> (ori, x, y)  (sodaa.py:262-266)"""
    return (filename.split('__')[0], int(filename.split('___')[0].split('__')[-1]), int(filename.split('___')[-1].split('.')[0]))


def merge_patch_detections(filenames, results, num_classes, iou_thr, device='cuda'):
    """sodaa.py:248-296 + _merge_func: shift every patch's boxes to whole-image coordinates, then rotated NMS per class
    (`pt_nms_rotated_sorted`).  -> [(ori_name, [ndarray[n,6] per class])]"""
    import torch
    from .. import functional as PF
    collector = defaultdict(list)
    for filename, result in zip(filenames, results):
        ori, x0, y0 = _patch_origin(filename)
        rows = []
        for i, res in enumerate(result):
            b = res[:, :-1].copy()
            b[..., :2] = b[..., :2] + np.array([x0, y0], dtype=np.float32)
            rows.append(np.concatenate([np.zeros((b.shape[0], 1)) + i, b, res[:, [-1]]], axis=1))
        collector[ori].append(np.concatenate(rows, axis=0))
    merged = []
    for ori, parts in collector.items():
        ld = np.concatenate(parts, axis=0)
        labels, dets = ld[:, 0], ld[:, 1:]
        per_cls = []
        for k in range(num_classes):
            d = dets[labels == k]
            if len(d) == 0:
                per_cls.append(np.empty((0, dets.shape[1]), dtype=np.float32))
                continue
            t = torch.from_numpy(d).to(device=device, dtype=torch.float32).contiguous()
            kept, _ = PF.nms_rotated(t[:, :5], t[:, 5], iou_thr)
            per_cls.append(kept.cpu().numpy())
        merged.append((ori, per_cls))
    return merged


@DATASETS.register_module()
class SODAADataset(CustomDataset):
    CLASSES = ('airplane', 'helicopter', 'small-vehicle', 'large-vehicle', 'ship', 'container', 'storage-tank', 'swimming-pool',
               'windmill')

    def __init__(self, ori_ann_file, angle_version='le90', **kwargs):
        self.angle_version = angle_version
        super().__init__(**kwargs)
        self.ori_data_infos = self.load_ori_annotations(ori_ann_file)
        self.cat_ids = {i: c for i, c in enumerate(self.CLASSES)}

    def load_ori_annotations(self, ori_ann_folder):
        infos = _load_sodaa_folder(ori_ann_folder, self.angle_version, 'category_id', skip_long_polys=True)
        self.ori_img_ids = [i['filename'].split(os.sep)[-1][:-4] for i in infos]
        return infos

    def get_ori_ann_info(self, idx):
        return self.ori_data_infos[idx]['ann']

    def load_annotations(self, ann_folder):
        infos = _load_sodaa_folder(ann_folder, self.angle_version, 'cat_id')
        self.img_ids = [i['filename'].split(os.sep)[-1][:-4] for i in infos]
        return infos

    def _filter_imgs(self):
        return [i for i, info in enumerate(self.data_infos) if info['ann']['labels'].size > 0]

    def _set_group_flag(self):
        self.flag = np.zeros(len(self), dtype=np.uint8)

    def merge_det(self, results, with_merge=True, nms_iou_thr=0.5, device='cuda', **kwargs):
        if not with_merge:
            return [(info.get('id'), r) for info, r in zip(self.data_infos, results)]
        return merge_patch_detections([i['filename'] for i in self.data_infos], results, len(self.CLASSES), nms_iou_thr, device)

    def evaluate(self, results, metric='mAP', logger=None, proposal_nums=(100, 300, 1000), iou_thr=None, scale_ranges=None,
                 metric_items=None, nproc=4, device='cuda', reference_ids=True):
        """sodaa.py:377-470: merge the patch detections into whole images (rotated NMS at 0.5), then the SODA-A protocol
        (`evaluation.SODAAEvaluator` = SODAAeval on the GPU) against the whole-image annotations."""
        from ..evaluation import SODAAEvaluator
        merged = self.merge_det(results, device=device)
        merge_idx = [self.ori_img_ids.index(name) for name, _ in merged]
        dets = [r for _, r in merged]
        if not isinstance(metric, str):
            assert len(metric) == 1
            metric = metric[0]
        if metric not in ['mAP']:
            raise KeyError(f'metric {metric} is not supported')
        annotations = [self.get_ori_ann_info(i) for i in merge_idx]
        # sodaa.py:401-403 only defines `iou_thrs` when iou_thr is None (a float argument raises NameError there)
        if iou_thr is not None:
            raise NotImplementedError('SODAADataset.evaluate: the reference only works with iou_thr=None (0.50:0.05:0.95)')
        ev = SODAAEvaluator(annotations, len(self.CLASSES), device=device, reference_ids=reference_ids)
        out = ev.evaluate(dets)
        prec = out['precision']                                              # [T, R, K, A, M]
        per = []
        for k, name in self.cat_ids.items():
            p = prec[:, :, k, 0, -1]
            p = p[p > -1]
            per.append((f'{name}', f'{float(p.mean()) if p.numel() else float("nan"):0.3f}'))
        if metric_items is None:
            metric_items = ['AP', 'AP_50', 'AP_75', 'AP_eS', 'AP_rS', 'AP_gS', 'AP_Normal']
        eval_results = {}
        for item in metric_items:
            eval_results[f'{metric}_{item}'] = float(f'{out[item]:.3f}')
        ap = out['stats'][:7]
        eval_results[f'{metric}_mAP_copypaste'] = ' '.join(f'{v:.3f}' for v in ap) + ' '
        eval_results['classwise'] = per
        return eval_results

    def evaluate_map(self, results, iou_thr=0.5, nms_iou_thr=0.5, device='cuda'):
        from ..evaluation import eval_rbbox_map
        merged = self.merge_det(results, nms_iou_thr=nms_iou_thr, device=device)
        idx = [self.ori_img_ids.index(name) for name, _ in merged]
        mean_ap, per = eval_rbbox_map([r for _, r in merged], [self.get_ori_ann_info(i) for i in idx], iou_thr=iou_thr, device=device)
        return dict(mAP=mean_ap, per_class=per)


@DATASETS.register_module()
class SODAADOTADataset(CustomDataset):
    CLASSES = SODAADataset.CLASSES
    FORCED_IOU_THR = 0.25            # sodaa_rewrite.py:206 overwrites the caller's iou_thr (None = honour the argument)

    def __init__(self, ann_file, pipeline, version='oc', difficulty=100, **kwargs):
        self.version, self.difficulty = version, difficulty
        super().__init__(ann_file, pipeline, **kwargs)

    def load_annotations(self, ann_folder):
        infos = _load_sodaa_folder(ann_folder, 'le90', 'cat_id')          # sodaa_rewrite.py:119 hard-codes 'le90'
        self.img_ids = [i['filename'].split(os.sep)[-1][:-4] for i in infos]
        return infos

    def _filter_imgs(self):
        return [i for i, info in enumerate(self.data_infos) if not self.filter_empty_gt or info['ann']['labels'].size > 0]

    def _set_group_flag(self):
        self.flag = np.zeros(len(self), dtype=np.uint8)

    def evaluate(self, results, metric='mAP', logger=None, proposal_nums=(100, 300, 1000), iou_thr=0.5, scale_ranges=None,
                 nproc=1, device='cuda'):
        """sodaa_rewrite.py:176-233"""
        from ..evaluation import eval_rbbox_map
        if self.FORCED_IOU_THR is not None:
            iou_thr = self.FORCED_IOU_THR
        if not isinstance(metric, str):
            assert len(metric) == 1
            metric = metric[0]
        if metric != 'mAP':
            raise KeyError(f'metric {metric} is not supported')
        if scale_ranges is not None:
            raise NotImplementedError('scale_ranges')
        assert isinstance(iou_thr, float)
        mean_ap, _ = eval_rbbox_map(results, [self.get_ann_info(i) for i in range(len(self))], iou_thr=iou_thr, device=device)
        return {'mAP': mean_ap}

    def merge_det(self, results, nproc=4, device='cuda'):
        """sodaa_rewrite.py:235-279 (patch origin from the `__x___y` suffix, NMS at 0.1)."""
        names = []
        for img_id in self.img_ids:
            x_y = re.findall(r'__\d+___\d+', img_id)
            x, y = re.findall(r'\d+', x_y[0])
            names.append(f"{img_id.split('__')[0]}__{x}___{y}.jpg")
        merged = merge_patch_detections(names, results, len(self.CLASSES), 0.1, device)
        return [m[0] for m in merged], [m[1] for m in merged]
