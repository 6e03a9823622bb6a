"""The reference's only test vectors for the rotated-IoU op (OBB_TOD/tests/test_utils/test_overlaps.py:7-15): three degenerate /
huge predicted boxes (sides 1e-7 ... 1.81e7) against four ground-truth boxes, expected IoU 0 within 1e-3.  They pin the fp32
robustness of the oracle's polygon clipping (CPU) and of `pt_box_iou_rotated` (MI355X) where mmcv itself is absent."""
import pytest
import torch

PREDICT = [[903.34, 1034.4, 1.81e-7, 1e-7, -0.312],
           [903.34, 1034.4, 1e-7, 1e-3, -0.312],
           [903.34, 1034.4, 1.81e7, 1e7, -0.312]]
GT = [[2.1525e+02, 7.5750e+01, 3.3204e+01, 1.2649e+01, 3.2175e-01],
      [3.0013e+02, 7.7144e+02, 4.9222e+02, 3.1368e+02, -1.3978e+00],
      [8.4887e+02, 6.9989e+02, 4.6854e+02, 3.0743e+02, -1.4008e+00],
      [8.5250e+02, 7.0250e+02, 7.6181e+02, 3.8200e+02, -1.3984e+00]]
EXPECT = torch.zeros(3, 4)


def test_oracle_rotated_iou_of_degenerate_and_huge_boxes():
    from oracle import ref_ops as R
    iou = R.box_iou_rotated(torch.tensor(PREDICT), torch.tensor(GT))
    assert iou.shape == (3, 4) and torch.isfinite(iou).all()
    assert torch.allclose(iou, EXPECT, atol=1e-3)
    aligned = R.box_iou_rotated(torch.tensor(PREDICT), torch.tensor(GT[:3]), aligned=True)
    assert torch.allclose(aligned, torch.zeros(3), atol=1e-3)


@pytest.mark.gpu
def test_kernel_rotated_iou_of_degenerate_and_huge_boxes():
    from point_teacher_amd import functional as F
    from point_teacher_amd import ops
    p, g = torch.tensor(PREDICT, device='cuda'), torch.tensor(GT, device='cuda')
    iou = F.box_iou_rotated(p, g)
    assert iou.shape == (3, 4) and torch.isfinite(iou).all()
    assert torch.allclose(iou.cpu(), EXPECT, atol=1e-3)
    assert torch.allclose(ops.box_iou_rotated(p, g[:3], aligned=True).cpu(), torch.zeros(3), atol=1e-3)     # the mmcv-signature surface
    # the huge box CONTAINS every ground truth: its IoU is area(gt) / area(box) ~ 1e-9, not an artefact of a failed clip
    from oracle import ref_ops as R
    ref = R.box_iou_rotated(torch.tensor(PREDICT), torch.tensor(GT))
    assert torch.allclose(iou.cpu(), ref, atol=1e-6)
