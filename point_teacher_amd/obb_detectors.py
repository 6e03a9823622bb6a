"""RotatedFCOS_Student and RotatedFCOS_TS - the oriented-box teacher->student detector of config 5
(`OBB_TOD/configs/point teacher/sodaa_fcos_pointteacher_1x.py`), mirroring
/root/reference/OBB_TOD/mmrotate/models/detectors/{rotated_fcos_student,
rotated_fcos_teacher_student,single_stage}.py.

The per-iteration order of operations is the one of TS_P2B_FCOS (the reference file is a
box-format edit of the HBB one), so the class only swaps the box-format hooks: boxes are
(cx,cy,w,h,a), IoUs are rotated, bags/negatives/augmentation come from obb_proposals and the
white rectangles are painted with the image maximum.
"""
import numpy as np
import torch

from .detectors import Student_FCOS, TS_P2B_FCOS
from .obb import rbbox_overlaps
from .obb_proposals import (MIL_gen_proposals_from_cfg, gen_negative_proposals, generate_black_paper_obb_batch,
                            obb2poly, strong_augmentation_images, strong_augmentation_masked)
from .registry import DETECTORS


def rbbox2result(bboxes, labels, num_classes):
    """core/bbox/transforms.py:54-70: list of per-class ndarray[n,6] = (cx,cy,w,h,a,score)."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 6), dtype=np.float32) for _ in range(num_classes)]
    bboxes = bboxes.detach().cpu().numpy()
    labels = labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


def random_point_in_quadrilateral(quads, version='le90', ux=None, uy=None, corner=None):
    """detectors/data_augument_bank.py:1705-1737: a random corner of the box shrunk to a random
    (w, h) fraction (rounded to 0.1 px)."""
    M = quads.shape[0]
    dev = quads.device
    ux = torch.rand(M, device=dev) if ux is None else ux.to(dev)
    uy = torch.rand(M, device=dev) if uy is None else uy.to(dev)
    corner = torch.randint(0, 4, (M,), device=dev) if corner is None else corner.to(dev)
    rq = quads.clone()
    rq[:, 2] = torch.round(ux.to(quads.dtype) * quads[:, 2] * 10) / 10.0
    rq[:, 3] = torch.round(uy.to(quads.dtype) * quads[:, 3] * 10) / 10.0
    poly = obb2poly(rq, version=version)
    r = torch.arange(M, device=dev)
    return torch.stack([poly[r, 2 * corner], poly[r, 2 * corner + 1]], dim=1)


@DETECTORS.register_module()
class RotatedFCOS_Student(Student_FCOS):
    """detectors/rotated_fcos_student.py:41-57 over single_stage.py (backbone -> neck -> neck_agg -> head)."""

    def simple_test(self, img, img_metas, rescale=False):
        res = self.bbox_head.simple_test(self.extract_feat(img), img_metas, rescale=rescale)
        return [rbbox2result(b, l, self.bbox_head.num_classes) for b, l in res]


@DETECTORS.register_module()
class RotatedFCOS_TS(TS_P2B_FCOS):
    """detectors/rotated_fcos_teacher_student.py:45-520."""

    _epoch_key = 'filename'                                   # :259-264 keys the epoch counter on 'filename'
    _mil_gen = staticmethod(MIL_gen_proposals_from_cfg)
    _neg_gen = staticmethod(gen_negative_proposals)

    def __init__(self, _model_, angle_version, _point_='random', num_stages=2, num_refine=500,
                 num_training_burninstep1=512, num_training_burninstep2=512, ema_alpha=0.999, filter_score=0.8,
                 burn_in_step=20000, lamda=1.0, alpha=[0.1, 1.0, 0.1],
                 shape_list=[[40, 40, 0.5, 0.5], [20, 40, 0.5, 0.5], [30, 120, 0.5, 0.5], [40, 100, 0.5, 0.5],
                             [60, 40, 0.5, 0.5]],
                 train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__(_model_, _point_=_point_, num_stages=num_stages, num_refine=num_refine,
                         num_training_burninstep1=num_training_burninstep1,
                         num_training_burninstep2=num_training_burninstep2, ema_alpha=ema_alpha,
                         filter_score=filter_score, burn_in_step=burn_in_step, lamda=lamda, alpha=alpha,
                         shape_list=shape_list, train_cfg=train_cfg, test_cfg=test_cfg, pretrained=pretrained,
                         init_cfg=init_cfg)
        self.angle_version = angle_version

    @staticmethod
    def _aligned_iou(a, b):
        return rbbox_overlaps(a, b, mode='iou', is_aligned=True)

    @staticmethod
    def _cxcywh(boxes):
        return boxes[:, :4]

    def _strong_aug_images(self, img):
        return strong_augmentation_images(img, params=self._inject.get('aug'))

    def _strong_aug(self, img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes, params=None, imgs=None):
        return strong_augmentation_masked(img, gt_points, gt_labels, pseudo_points, pseudo_labels, pseudo_bboxes,
                                          self.angle_version, params=params if params is not None else self._inject.get('aug'),
                                          imgs=imgs)

    def _black_paper(self, imgs, gt_bboxes, imgsize, draws):
        return generate_black_paper_obb_batch(imgs, gt_bboxes, self._prior_on(imgs[0].device), range(int(len(self.pattern) / 2)),
                                              imgsize, draws=draws)

    def _initial_points(self, gt_bboxes, u):
        """genrate_points :381-396"""
        if self._point_ == 'center':
            return gt_bboxes[:, :2]
        assert self._point_ == 'random'
        return random_point_in_quadrilateral(gt_bboxes, self.angle_version, *(u if u is not None else ()))

    def _to_results(self, dets, labels, num_classes):
        return rbbox2result(dets, labels, num_classes)
