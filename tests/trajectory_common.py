"""Inputs and product set-up shared by tests/test_trajectory.py and oracle/gen_golden_trajectory.py (seeded, tiny)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, BURN_IN, SIZE = 20, 9, 256
CFG_PATH = os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py')
N_SAMPLES = 16384


def syn_draws(G, seed, n_prior=6):
    """Per-object draws of the step-1 rectangle generator (TS_P2B_FCOS._inject['syn'][i] / ref_ops.generate_black_paper)."""
    g = torch.Generator().manual_seed(seed)
    d = dict(cls=torch.randint(0, n_prior, (G,), generator=g))
    for n in ('scale', 'x', 'y', 'a', 'boost', 'itv', 'dev'):
        d[n] = torch.rand(G, generator=g)
    for n in ('wn', 'rn'):
        d[n] = torch.randn(G, generator=g)
    d['itv2'] = d['itv']
    return d


def tiles(n=6, size=SIZE, seed=21):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        img = (torch.rand(3, size, size, generator=g) * 90 + 60).round()
        k = int(torch.randint(14, 26, (1,), generator=g))
        c = torch.rand(k, 2, generator=g) * (size - 48) + 24 + 0.37
        wh = torch.exp(torch.randn(k, 2, generator=g) * 0.4 + np.log(14.0)).clamp(4, 40)
        out.append((img, torch.cat([c - wh / 2, c + wh / 2], 1), torch.randint(0, 8, (k,), generator=g),
                    dict(ori_filename=f'traj{i}.png', img_shape=(size, size, 3), scale_factor=np.ones(4, np.float32))))
    return out


def schedule(steps=STEPS):
    """Per iteration: (tile indices, neg_u, aug, is_step1, syn draws or None) - every random draw of the 20 iterations."""
    ts = tiles()
    g = torch.Generator().manual_seed(31)
    out = []
    for it in range(steps):
        sel = [(2 * it + j) % len(ts) for j in range(2)]
        neg_u = torch.rand(2, 4, 200, generator=g)
        aug = ([['horizontal', 'vertical', 'diagonal', 'None'][int(torch.randint(0, 4, (1,), generator=g))] for _ in range(2)],
               [round(float(torch.rand(1, generator=g)) * 0.4 + 0.8, 1) for _ in range(2)])
        step1 = it <= BURN_IN                                         # `count <= burn_in_step` (fcos_p2b_teacher_student.py:133)
        syn = [syn_draws(ts[i][1].shape[0], 100 * it + j) for j, i in enumerate(sel)] if step1 else None
        out.append((sel, neg_u, aug, step1, syn))
    return ts, out


def build_product(dev):
    """The detector in the state every trajectory starts from (seed 3, benchmark_init_, damped bag logits) -> (cfg, model)."""
    import point_teacher_amd as pta
    from point_teacher_amd.synthetic import benchmark_init_
    torch.manual_seed(3)
    cfg = pta.Config.fromfile(CFG_PATH)
    cfg.model['burn_in_step'] = BURN_IN
    model = pta.build_detector(cfg.model).to(dev)
    benchmark_init_(model, phase2=True)
    with torch.no_grad():      # moderate bag logits, as in a trained head (oracle/ref_model.damp_mil_logits explains why)
        for det in (model.student, model.teacher):
            for m in list(det.bbox_head.fc_cls) + list(det.bbox_head.fc_ins):
                m.weight.mul_(0.1)
                m.bias.mul_(0.1)
    model.train()
    return cfg, model


def sample_index(names, shapes, n=N_SAMPLES, seed=77):
    """Fixed random positions in the concatenation of the named tensors (row-major) -> (sorted int64 [n])."""
    total = sum(int(np.prod(shapes[k])) for k in names)
    g = torch.Generator().manual_seed(seed)
    return torch.sort(torch.randint(0, total, (n,), generator=g))[0]


def sample(sd, names, idx):
    """The values of `sd` at `idx` (see sample_index); tensors may live anywhere."""
    out = torch.empty(idx.numel(), dtype=torch.float32)
    off, lo = 0, 0
    idx_l = idx.tolist()
    import bisect
    for k in names:
        v = sd[k]
        n = v.numel()
        hi = bisect.bisect_left(idx_l, off + n, lo)
        if hi > lo:
            loc = idx[lo:hi] - off
            flat = v.detach().reshape(-1)
            out[lo:hi] = flat[loc.to(flat.device)].float().cpu()
        off, lo = off + n, hi
    return out
