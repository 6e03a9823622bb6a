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


class SyntheticTiles:
    """A cycled dataset of `n` tiles resident on the device; `batch(i, B)` gives train_step's dict."""

    def __init__(self, n=64, size=800, num_classes=8, mean_objects=300, seed=0, device='cpu', rank=0, world=1):
        self.items = [make_tile(rank + world * k, size, num_classes, mean_objects, seed, device) for k in range(n)]

    def __len__(self):
        return len(self.items)

    def batch(self, it, B):
        sel = [self.items[(it * B + j) % len(self.items)] for j in range(B)]
        return dict(img=torch.stack([s[0] for s in sel]), img_metas=[s[3] for s in sel],
                    gt_bboxes=[s[1] for s in sel], gt_labels=[s[2] for s in sel])
