"""Golden for tests/test_trajectory.py: 20 consecutive iterations of `oracle/ref_model.py` (EMA, forward, backward, clip + SGD with the
config's constant warm-up lr / 3) across the phase switch, from the product's seeded initial state, on 8 CPU threads.

TEST INFRASTRUCTURE (like everything under oracle/): the product never imports it.  Writes tests/golden/trajectory_oracle.npz: per
step the loss dict, the gradient norm, the distance travelled from the start, and the student / teacher weights at 16 384 fixed
random positions (tests/trajectory_common.sample_index).  The map is chaotic (tests/test_trajectory.py): regenerating with another
thread count changes late steps by the same amplification the test measures; 8 threads is part of the recipe.
usage: python oracle/gen_golden_trajectory.py   (~10 min on 8 cores)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref_model as M  # noqa: E402
from oracle import ref_ops as R  # noqa: E402
import trajectory_common as T  # noqa: E402


def main():
    torch.set_num_threads(8)
    cfg, model = T.build_product('cpu')
    lc = cfg.lr_config
    lr = cfg.optimizer['lr'] * lc['warmup_ratio']
    sd_s = {k[len('student.'):]: v.detach().clone() for k, v in model.state_dict().items() if k.startswith('student.')}
    sd_t = {k[len('teacher.'):]: v.detach().clone() for k, v in model.state_dict().items() if k.startswith('teacher.')}
    sd_0 = {k: v.clone() for k, v in sd_s.items()}
    names = [k for k in sd_s if M.trainable(k)]
    tnames = [k for k in sd_t if M.is_param(k)]
    idx = T.sample_index(names, {k: tuple(sd_s[k].shape) for k in names})
    tidx = T.sample_index(tnames, {k: tuple(sd_t[k].shape) for k in tnames}, seed=78)
    ts, sched = T.schedule()
    out = dict(lr=np.float64(lr), threads=np.int64(8), sample_idx=idx.numpy(), teacher_sample_idx=tidx.numpy(),
               student0=T.sample(sd_s, names, idx).numpy(), names=np.array(names), teacher_names=np.array(tnames))
    mom = {}
    keys = None
    for it, (sel, neg_u, aug, step1, syn) in enumerate(sched):
        t0 = time.time()
        img = torch.stack([ts[i][0] for i in sel])
        boxes, labels = [ts[i][1] for i in sel], [ts[i][2] for i in sel]
        inject = dict(neg0=neg_u, aug=aug)
        if step1:
            inject['syn'] = syn
        params = {k: (v.clone().requires_grad_(True) if M.trainable(k) else v) for k, v in sd_s.items()}
        sd_t = M.ema(sd_t, sd_s)
        gp = [R.bbox_xyxy_to_cxcywh(b)[:, :2] for b in boxes]
        fwd = M.forward_train_step1 if step1 else M.forward_train_step2
        ref, _ = fwd(params, sd_t, img, boxes, labels, gp, dict(M.MODEL_CFG), inject)
        ref['loss'] = M.total_loss(ref)
        gr = torch.autograd.grad(ref['loss'], [params[k] for k in names], allow_unused=True)
        sd_s, mom, gnorm = M.sgd_clip_step(sd_s, dict(zip(names, gr)), mom, lr, it == 0)
        ks = sorted(ref.keys())
        out[f'keys_{it}'] = np.array(ks)
        out[f'loss_{it}'] = np.array([float(ref[k]) for k in ks], dtype=np.float64)
        out[f'gnorm_{it}'] = np.float64(gnorm)
        out[f'travelled_{it}'] = np.float64(sum(float((sd_s[k].double() - sd_0[k].double()).pow(2).sum()) for k in names) ** 0.5)
        out[f'student_{it}'] = T.sample(sd_s, names, idx).numpy()
        out[f'teacher_{it}'] = T.sample(sd_t, tnames, tidx).numpy()
        print(f'it {it:2d} phase {1 if step1 else 2} loss {float(ref["loss"]):9.4f} |g| {gnorm:8.2f} travelled {float(out[f"travelled_{it}"]):.4f} '
              f'({time.time() - t0:.0f} s)', flush=True)
    dst = os.path.join(ROOT, 'tests', 'golden', 'trajectory_oracle.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes')


if __name__ == '__main__':
    main()
