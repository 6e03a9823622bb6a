"""Seeded synthetic AI-TOD-like tiles: the datasets are not available
here, so benchmarks and end-to-end tests run on images of the same shape and statistics:
800x800 uint8-valued float images (low-pass noise, mean ~110, sigma ~40), G ~ Poisson(300)
objects per image clipped to [1, 600], centres uniform in [16, 784]^2, w,h ~ LogNormal(ln 12,
0.5) clipped to [2, 64] px, labels uniform over 8 classes."""
import math

import numpy as np
import torch


def make_tile(idx, size=800, num_classes=8, mean_objects=300, seed=0, device='cpu', max_objects=600):
    g = torch.Generator().manual_seed(seed * 1_000_003 + idx)
    low = torch.randn(1, 3, size // 16 + 1, size // 16 + 1, generator=g)
    img = torch.nn.functional.interpolate(low, size=(size, size), mode='bilinear', align_corners=False)[0]
    img = (img * 40.0 + 110.0 + torch.randn(3, size, size, generator=g) * 6.0).clamp(0, 255).round()
    G = int(np.clip(torch.poisson(torch.tensor(float(mean_objects)), generator=g).item(), 1, max_objects))
    c = torch.rand(G, 2, generator=g) * (size - 32) + 16
    wh = torch.exp(torch.randn(G, 2, generator=g) * 0.5 + math.log(12.0)).clamp(2, 64)
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, size)
    labels = torch.randint(0, num_classes, (G,), generator=g)
    meta = dict(ori_filename=f'syn_{idx}.png', filename=f'syn_{idx}.png', img_shape=(size, size, 3),
                ori_shape=(size, size, 3), pad_shape=(size, size, 3), scale_factor=np.ones(4, np.float32), flip=False)
    return img.to(device), boxes.to(device), labels.to(device), meta


def make_tile_obb(idx, size=1200, num_classes=9, mean_objects=300, seed=0, device='cpu', max_objects=600):
    """SODA-A-like tile of config 5: the same low-pass-noise picture after the config's Normalize
    (mean [123.675, 116.28, 103.53], std [58.395, 57.12, 57.375]), oriented boxes (cx, cy, w, h, a) in the
    le90 convention (w >= h, a in [-pi/2, pi/2)), 9 classes."""
    img, boxes, _, meta = make_tile(idx, size, num_classes, mean_objects, seed, 'cpu', max_objects)
    g = torch.Generator().manual_seed(seed * 1_000_003 + idx + 77)
    mean = torch.tensor([123.675, 116.28, 103.53]).view(3, 1, 1)
    std = torch.tensor([58.395, 57.12, 57.375]).view(3, 1, 1)
    img = (img - mean) / std
    G = boxes.shape[0]
    c = (boxes[:, :2] + boxes[:, 2:]) / 2
    wh = (boxes[:, 2:] - boxes[:, :2]).clamp(min=2.0)
    a = torch.rand(G, generator=g) * math.pi - math.pi / 2
    rb = torch.stack([c[:, 0], c[:, 1], torch.max(wh[:, 0], wh[:, 1]), torch.min(wh[:, 0], wh[:, 1]), a], 1)
    labels = torch.randint(0, num_classes, (G,), generator=g)
    return img.to(device), rb.to(device), labels.to(device), meta


class SyntheticTiles:
    """A cycled dataset of `n` tiles resident on the device; `batch(i, B)` gives train_step's dict."""

    def __init__(self, n=64, size=800, num_classes=8, mean_objects=300, seed=0, device='cpu', rank=0, world=1,
                 oriented=False):
        make = make_tile_obb if oriented else make_tile
        self.items = [make(rank + world * k, size, num_classes, mean_objects, seed, device) for k in range(n)]

    def __len__(self):
        return len(self.items)

    def batch(self, it, B):
        sel = [self.items[(it * B + j) % len(self.items)] for j in range(B)]
        return dict(img=torch.stack([s[0] for s in sel]), img_metas=[s[3] for s in sel],
                    gt_bboxes=[s[1] for s in sel], gt_labels=[s[2] for s in sel])


def benchmark_init_(model, phase2=False):
    """Random-init stand-in for the pretrained state the recipe starts from.  There is no
    network here for `open-mmlab://detectron/resnet50_caffe`; a Kaiming-initialised stem fed
    raw 0-255 pixels produces activations in the hundreds and the first SGD steps diverge
    (in the reference as well).  Scaling the stem filters by 1/64 gives O(1) features, i.e. the
    regime real training runs in; it changes no shape and no amount of work.  With
    `phase2=True` the regression bias is set to +1 (8 px boxes): phase 2 only starts after
    4000 burn-in iterations, when the regression branch no longer emits the all-zero
    distances of a cold start (zero-area pseudo boxes make the centerness target 0.01/0).
    The MIL box-refinement layer `fc_reg` is scaled by 0.01 so that refined bags stay near the
    coarse boxes as they do in a trained model; torch's default Linear init makes it emit
    |dw| ~ 4 deltas, i.e. 60x larger boxes clipped to the whole image, which turns every RoI
    into a 100x100-pixel crop with a 15x15 sampling grid per bin (225x the realistic RoIAlign work).
    The teacher's backbone is a copy of the student's, as after loading the configs' pretrained checkpoint into both."""
    import torch
    with torch.no_grad():
        for m in (model.student, model.teacher):
            if not hasattr(m.bbox_head, 'conv_angle'):       # the OBB config feeds mean/std-normalised pixels
                m.backbone.conv1.weight.mul_(1.0 / 64.0)
            for fc in m.bbox_head.fc_reg:
                fc.weight.mul_(0.01)
                fc.bias.zero_()
            if phase2 and hasattr(m.bbox_head, 'conv_reg'):      # (an anchor-based head starts from its anchors: nothing to do)
                m.bbox_head.conv_reg.bias.fill_(1.0)
        # both detectors load the SAME pretrained backbone in a real run (init_cfg of the configs): same stand-in for both
        model.teacher.backbone.load_state_dict(model.student.backbone.state_dict())
    return model
