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
        from point_teacher_amd.runtime import StepLR
        # bucketed exchange overlapped with backward (runtime.BucketedGradExchange) on a toy teacher/student pair
        from point_teacher_amd.runtime import BucketedGradExchange, FlatParams

        class Pair(torch.nn.Module):
            def __init__(self):
                super().__init__()
                def net():
                    return torch.nn.Sequential(torch.nn.Linear(37, 64), torch.nn.ReLU(), torch.nn.Linear(64, 50),
                                               torch.nn.ReLU(), torch.nn.Linear(50, 3), torch.nn.Linear(3, 3))
                self.student, self.teacher = net(), net()
        torch.manual_seed(7)                                     # same weights on both ranks
        pair = Pair()
        for q_ in pair.student[5].parameters():                  # a layer that never receives a gradient
            pass
        flat = FlatParams(pair)
        ex = BucketedGradExchange(flat, n_buckets=3, device=None)
        assert len(ex.buckets) >= 2 and ex.buckets[-1][1] == flat.n_train
        torch.manual_seed(200 + rank)
        x = torch.randn(16, 37)
        for it in range(2):                                      # twice: hooks and views survive an iteration
            flat.zero_grad(); flat.detach_grads()
            ex.begin()
            pair.student[:5](x).pow(2).mean().backward()         # student[5] unused -> its segment must stay zero
            ex.finish()
            ref = Pair(); ref.load_state_dict(pair.state_dict())
            ref.student[:5](x).pow(2).mean().backward()
            mine = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ref.student.parameters()])
            gathered = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            expect_named = sum(gathered) / world
            o = 0
            for name, p in ref.student.named_parameters():
                off, n = flat.slices[name]
                torch.testing.assert_close(flat.grad_flat[off:off + n], expect_named[o:o + n], rtol=1e-5, atol=1e-6)
                o += n
            assert flat.check_views()
            assert ex.issued == ex.issue_order
        # ranks on which DIFFERENT parameters receive gradients (an empty image on one rank, a data-dependent branch): rank 0
        # skips the middle layer's path entirely, rank 1 uses it; arrival orders differ, yet both ranks must enqueue the same
        # sequence of all-reduces (a mismatch would hang or mix buckets) and end with the mean of what was produced
        flat.zero_grad(); flat.detach_grads()
        ex.begin()
        if rank == 0:
            (pair.student[4](torch.randn(16, 50)).pow(2).mean()).backward()      # only the last used layer
        else:
            pair.student[:5](x).pow(2).mean().backward()
        ex.finish()
        assert ex.issued == ex.issue_order
        per_rank = [torch.empty_like(flat.grad_flat) for _ in range(world)]
        dist.all_gather(per_rank, flat.grad_flat.clone())
        torch.testing.assert_close(per_rank[0], per_rank[1], rtol=0, atol=0)     # identical reduced gradient on both ranks
        off0, n0 = flat.slices['0.weight']
        assert float(flat.grad_flat[off0:off0 + n0].abs().sum()) > 0              # rank 1's contribution / world arrived
        # a detector without a teacher (the supervised baselines of row N4): same flat layout, same exchange
        torch.manual_seed(9)
        solo = torch.nn.Sequential(torch.nn.Linear(11, 20), torch.nn.ReLU(), torch.nn.Linear(20, 4))
        sflat = FlatParams(solo)
        assert not sflat.has_teacher and sflat.teacher_flat is None and not hasattr(solo, '_flat')
        sex = BucketedGradExchange(sflat, n_buckets=2, device=None)
        torch.manual_seed(300 + rank)
        xs = torch.randn(8, 11)
        sflat.zero_grad(); sflat.detach_grads()
        sex.begin()
        solo(xs).pow(2).mean().backward()
        sex.finish()
        gsum = sflat.grad_flat.clone()
        dist.all_reduce(gsum)                                    # every rank already holds the mean: the sum is world x it
        torch.testing.assert_close(gsum, sflat.grad_flat * world, rtol=1e-5, atol=1e-7)
        assert float(sflat.grad_flat.abs().sum()) > 0
        # data side: under an initialised group build_dataloader picks the distributed samplers with this rank, the ranks
        # split every epoch between them and agree on the permutation (samplers/group_sampler.py:106-140)
        import types
        import numpy as np
        from point_teacher_amd.datasets import DistributedGroupSampler, DistributedSampler, build_dataloader
        ds = types.SimpleNamespace(flag=(np.arange(21) % 3 == 0).astype(np.uint8), pipeline=None, __len__=lambda: 21)
        ds = type('DS', (), dict(flag=ds.flag, __len__=lambda self: 21, __getitem__=lambda self, i: i))()
        ld = build_dataloader(ds, samples_per_gpu=2, workers_per_gpu=1, dist=True, shuffle=True, seed=5, device='cpu')
        assert isinstance(ld.sampler, DistributedGroupSampler) and ld.sampler.rank == rank and ld.sampler.num_replicas == world
        ld.set_epoch(4)
        mine_idx = torch.tensor(list(iter(ld.sampler)))
        both = [torch.empty_like(mine_idx) for _ in range(world)]
        dist.all_gather(both, mine_idx)
        assert set(torch.cat(both).tolist()) == set(range(21)) and len(mine_idx) == len(ld.sampler) == 12
        for pair_ in torch.cat(both).view(-1, 2).tolist():          # a batch never mixes aspect-ratio groups
            assert ds.flag[pair_[0]] == ds.flag[pair_[1]]
        lt = build_dataloader(ds, samples_per_gpu=1, workers_per_gpu=1, dist=True, shuffle=False, device='cpu')
        assert isinstance(lt.sampler, DistributedSampler) and list(iter(lt.sampler)) == [i % 21 for i in range(rank, 22, world)]
        # sharded evaluation: rank r evaluated samples r, r + world, ... of a 7-sample set (padded to 8); rank 0 restores the order
        from point_teacher_amd.evaluation import collect_results
        part = [[np.full((1, 5), float(i % 7))] for i in range(rank, 8, world)]
        whole = collect_results(part, 7)
        if rank == 0:
            assert len(whole) == 7 and [float(r[0][0, 0]) for r in whole] == [0., 1., 2., 3., 4., 5., 6.]
        else:
            assert whole is None
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
