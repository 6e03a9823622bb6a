"""The N > 1 path on CPU: two gloo ranks (127.0.0.1) run the pieces of the data-parallel
exchange that do not need a GPU - the chunked flat-gradient mean, the coalesced log-var
reduction, reduce_mean normalisers and the initial-state broadcast contract."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from point_teacher_amd.core import reduce_mean
        from point_teacher_amd.detectors import LazyLogVars
        from point_teacher_amd.runtime import GradReducer, StepLR
        torch.manual_seed(100 + rank)
        n = 1_000_003
        g = torch.randn(n)
        all_g = [torch.empty(n) for _ in range(world)]
        dist.all_gather(all_g, g)
        expect = sum(all_g) / world
        red = GradReducer(chunks=4, device=torch.device('cpu'))
        assert red.world == world and red.stream is None
        out = red.reduce_(g.clone())
        torch.testing.assert_close(out, expect, rtol=1e-6, atol=1e-6)
        # normalisers: every rank sees the mean of the per-rank counts
        npos = reduce_mean(torch.tensor(float(10 + 4 * rank)))
        assert float(npos) == pytest.approx(sum(10 + 4 * r for r in range(world)) / world)
        # log vars: ONE coalesced all-reduce, same key set and values on every rank
        lv = LazyLogVars()
        lv['loss_cls'] = torch.tensor(1.0 + rank)
        lv['coarse_bboxes_iou'] = torch.tensor(0.5 * (rank + 1))
        m = lv.materialize()
        assert list(m) == ['loss_cls', 'coarse_bboxes_iou']
        assert m['loss_cls'] == pytest.approx(sum(1.0 + r for r in range(world)) / world)
        assert m['coarse_bboxes_iou'] == pytest.approx(sum(0.5 * (r + 1) for r in range(world)) / world)
        # identical LR schedule on every rank (constant warm-up x1/3, steps [8,11])
        s = StepLR(0.005, [8, 11], 0.1, 'constant', 10000, 1.0 / 3, iters_per_epoch=5000)
        assert s.lr_at(0) == pytest.approx(0.005 / 3) and s.lr_at(10000) == pytest.approx(0.005)
        assert s.lr_at(8 * 5000) == pytest.approx(0.0005) and s.lr_at(11 * 5000) == pytest.approx(0.00005)
        q.put((rank, 'ok'))
    except Exception as e:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_exchange():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == 'ok' for r in res), res
