"""N > 1 with the REAL model on the MI355X: two ranks (gloo rendezvous on 127.0.0.1, both on cuda:0 - the box has one card;
on a node each rank owns a GPU and the backend is RCCL) run `Trainer.step` of the Point-Teacher detector through both phases
on DIFFERENT data per rank, including the two cases that make ranks diverge in the reference: an image
without objects and an iteration in which no synthetic rectangle survives.  After three steps both ranks must hold bit-identical
student and teacher weights (same initial broadcast, same reduced gradients, same optimizer), the bucketed exchange must have
issued its collectives in the fixed order on both, and the loss-key set must be the same on every rank and iteration."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import point_teacher_amd as pta
        from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_
        dev = torch.device('cuda:0')
        torch.cuda.set_device(0)
        torch.manual_seed(50 + rank)                                  # DIFFERENT initial weights: the Trainer must broadcast rank 0's
        cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
        cfg.model['burn_in_step'] = 1                                 # steps 0, 1 = phase 1; step 2 = phase 2
        model = pta.build_detector(cfg.model).to(dev)
        benchmark_init_(model, phase2=True)
        model.train()
        tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
        assert tr.world == world and tr.exchange is not None
        data = SyntheticTiles(n=8, size=256, mean_objects=20, seed=11, device=dev, rank=rank, world=world)
        keys = []
        for it in range(3):
            batch = data.batch(it, 2)
            if rank == 1 and it == 0:                                 # an image without a single object on this rank only
                batch['gt_bboxes'][1] = batch['gt_bboxes'][1][:0]
                batch['gt_labels'][1] = batch['gt_labels'][1][:0]
            if rank == 0 and it == 1:                                 # every rectangle leaves the image: zero synthetic boxes
                draws = []
                for b in batch['gt_bboxes']:
                    G = b.shape[0]
                    d = {n: torch.full((G,), 0.5, device=dev) for n in ('x', 'y', 'a', 'boost', 'itv', 'itv2', 'dev')}
                    d.update(scale=torch.ones(G, device=dev), wn=torch.full((G,), 9.0, device=dev), rn=torch.full((G,), 9.0, device=dev),
                             cls=torch.full((G,), 4, dtype=torch.long, device=dev))       # prior (30, 120) at its largest: > 256 px
                    draws.append(d)
                model._inject = dict(syn=draws)
            else:
                model._inject = {}
            out = tr.step(batch)
            lv = out['log_vars'].materialize()
            assert all(v == v and abs(v) != float('inf') for v in lv.values()), (rank, it, lv)
            keys.append(sorted(lv))
            assert it == 0 or tr.exchange.issued == tr.exchange.issue_order      # (step 0 ends with a freshly built exchange)
            if it == 0:
                # round-2 verdict K1 / P3: the never-used MIL stacks left the live segment at the first step, on both ranks
                # alike (one bitmap all-reduce) - although rank 1 saw an image without objects
                dead = sorted(tr.flat.dead)
                assert tr.dead_known and len(dead) == 10, dead
                assert all(n.split('.')[1] in ('shared_fcs', 'shared_fcs_refine', 'fc_iou') for n in dead)
                assert tr.exchange.stats['bytes'] == 4 * tr.flat.n_train < 242e6
                dead_before = torch.cat([tr.flat.student_flat[tr.flat.slices[n][0]:tr.flat.slices[n][0] + tr.flat.slices[n][1]] for n in dead]).clone()
            else:
                # ... and the REAL backward now completes buckets while it runs: all but the first layers' buckets go out
                # before finish(), the first of them (MIL head) before the backbone's backward has produced anything
                st = tr.exchange.stats
                assert st['issued_during_backward'] >= max(st['buckets'] - 2, 4), st
        torch.cuda.synchronize()
        s, t = tr.flat.student_flat.cpu(), tr.flat.teacher_flat.cpu()
        both_s = [torch.empty_like(s) for _ in range(world)]
        both_t = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(both_s, s)
        dist.all_gather(both_t, t)
        assert torch.equal(both_s[0], both_s[1]), float((both_s[0] - both_s[1]).abs().max())
        assert torch.equal(both_t[0], both_t[1])
        assert torch.isfinite(s).all() and keys[0] == keys[1] == keys[2]
        dead_after = torch.cat([tr.flat.student_flat[tr.flat.slices[n][0]:tr.flat.slices[n][0] + tr.flat.slices[n][1]] for n in dead])
        assert torch.equal(dead_after, dead_before)                   # no weight decay, no momentum: torch.optim.SGD skips `grad is None`
        dlist = [None] * world
        dist.all_gather_object(dlist, dead)
        assert dlist[0] == dlist[1]
        klist = [None] * world
        dist.all_gather_object(klist, keys)
        assert klist[0] == klist[1]
        q.put((rank, 'ok'))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_rank_real_trainer_step():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    assert all(r[1] == 'ok' for r in res), res


def test_bench_two_ranks_end_to_end_line():
    """The driver's multi-GPU command - `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` - with N = 2 ranks sharing this box's one card over gloo (PT_DIST_BACKEND; on a node every
    rank owns a GPU and the backend is RCCL): rank setup, survey / timed / breakdown steps, barriers, max-over-ranks timing and the ONE
    rank-0 JSON line, so that the first RCCL run is not also the first run of this code path (round-3 verdict item 7).  Small
    tiles keep the CPU all-reduce of the 240 MB gradient buffer (seconds per step) out of a long test."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PT_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--size', '256',
           '--objects', '20', '--no-cpu-baseline', '--no-phase2', '--no-configs2']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                          # ONE line, printed by rank 0
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['higher_is_better'] is True
    assert d['config']['global_batch'] == 4 and d['config']['parallelism'] == 'dp2'
    assert d['value'] == pytest.approx(2 * 2 / (d['ms_per_step'] * 2 / 1e3), rel=1e-3)        # whole-job iterations / s over both ranks
    ex = d['exchange']
    assert ex['backend'] == 'gloo' and ex['buckets'] >= 4 and ex['bytes'] > 200e6 and ex['dead_bytes'] > 100e6
    assert ex['issued_during_backward'] >= ex['buckets'] - 2          # the all-reduces overlap the backward pass
    assert d['roofline']['bound'] in ('mfma', 'hbm') and d['roofline']['achieved'] > 0 and d['cpu_baseline'] is None
