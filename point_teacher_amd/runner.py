"""The step after the path: an epoch-based run loop around `runtime.Trainer` with the three
hooks the Point-Teacher configs register - LR schedule (inside the Trainer), text / JSON logger and checkpoints -
replacing mmcv's EpochBasedRunner + TextLoggerHook + CheckpointHook for this path
(HBB_TOD/mmdet/apis/train.py:88-170, configs/_base_/default_runtime.py, schedules/schedule_1x.py).

Two defects of the reference's resume are fixed here: the checkpoint also carries the
detector's iteration counter and its per-image point dictionaries (`TS_P2B_FCOS.get_extra_state`), and the
optimizer state is the flat momentum buffer, so a resumed run continues bit-for-bit in the same phase.
Logging reads the device-side `LazyLogVars` only every `interval` iterations (one D2H per log line).
"""
import json
import os
import time

import torch
import torch.distributed as dist


def merge_point_states(parts):
    """One extra-state dict out of every rank's: the DistributedSampler reshuffles every epoch, so the same image can sit in
    several ranks' dictionaries with different ages - the entry with the NEWEST refinement stamp wins (`TS_P2B_FCOS.point_stamp`,
    iteration of the last `update_points`; entries without a stamp are older than any with one; ties: lowest rank).  First-visit
    points (`gt_bboxes_point`) are what `update_points` blends with: the earliest-stamped visit would be the reference's, but
    the reference keeps them per process too, so they follow the refined entry's rank."""
    merged = dict(parts[0])
    best = {}
    for r, part in enumerate(parts):
        stamps = part.get('point_stamp', {})
        for name in set(part.get('gt_bboxes_point', {})) | set(part.get('refined_gt_bboxes_point', {})):
            st = stamps.get(name, -1)
            if name not in best or st > best[name][0]:
                best[name] = (st, r)
    for k in ('gt_bboxes_point', 'refined_gt_bboxes_point'):
        merged[k] = {}
    merged['point_stamp'] = {}
    for name, (st, r) in best.items():
        for k in ('gt_bboxes_point', 'refined_gt_bboxes_point'):
            if name in parts[r].get(k, {}):
                merged[k][name] = parts[r][k][name]
            else:                                        # e.g. refined on another rank only: take it from whoever has it
                for part in parts:
                    if name in part.get(k, {}):
                        merged[k][name] = part[k][name]
                        break
        if st >= 0:
            merged['point_stamp'][name] = st
    return merged


class Runner:
    def __init__(self, trainer, batches, work_dir, max_epochs, iters_per_epoch, log_interval=50, checkpoint_interval=1,
                 batch_size=2):
        """`batches(it, batch_size)` -> the dict `train_step` consumes (e.g. synthetic.SyntheticTiles.batch)."""
        self.trainer, self.batches, self.work_dir = trainer, batches, work_dir
        self.max_epochs, self.iters_per_epoch = int(max_epochs), int(iters_per_epoch)
        self.log_interval, self.checkpoint_interval, self.batch_size = int(log_interval), int(checkpoint_interval), batch_size
        self.rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        self.epoch = 0
        if self.rank == 0:
            os.makedirs(work_dir, exist_ok=True)
        self._json = os.path.join(work_dir, 'log.json')
        self._eval = None

    # ------------------------------------------------------------------------ evaluation --
    def register_eval(self, dataset, loader, interval=1, **eval_kwargs):
        """mmdet's EvalHook (apis/train.py:140-155, `evaluation = dict(interval=12, metric='bbox')` in the configs): every
        `interval` epochs rank 0 runs the test loop over `loader` (a test-mode DeviceLoader) and `dataset.evaluate`."""
        self._eval = (dataset, loader, int(interval), eval_kwargs)

    def evaluate(self):
        """Every rank tests its shard (`build_dataloader(dist=True, shuffle=False)` under an initialised group), rank 0 gets
        the whole dataset's results (apis/test.py:68-171) and scores them; the others return None.  A rank never waits in a
        gradient all-reduce while another one evaluates."""
        from .evaluation import multi_gpu_test, single_gpu_test
        dataset, loader, _, kw = self._eval
        model = self.trainer.model
        was_training = model.training
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if world > 1:
            results = multi_gpu_test(model, loader, len(dataset))
        else:
            it = iter(loader)
            results = single_gpu_test(model, lambda _: next(it), len(loader))
        model.train(was_training)
        if results is None:
            return None
        metrics = dataset.evaluate(results, **kw)
        rec = dict(mode='val', epoch=self.epoch, iter=len(results))
        rec.update({k: v for k, v in metrics.items() if isinstance(v, (int, float, str))})
        with open(self._json, 'a') as f:
            f.write(json.dumps(rec) + '\n')
        print(f'Epoch(val) [{self.epoch}][{len(results)}]\t' + ', '.join(f'{k}: {v:.4f}' for k, v in rec.items() if isinstance(v, float)),
              flush=True)
        return metrics

    # ---------------------------------------------------------------------- checkpoints --
    def save_checkpoint(self, name=None):
        """CheckpointHook: epoch_{n}.pth + latest.pth (rank 0 only)."""
        extra = self._gather_point_state()                  # collective: every rank takes part
        if self.rank != 0:
            return None
        path = os.path.join(self.work_dir, name or f'epoch_{self.epoch}.pth')
        state = self.trainer.state_dict()
        if extra is not None:
            state['model']['_extra_state'] = extra
        state['meta'] = dict(epoch=self.epoch, iter=self.trainer.iter, time=time.strftime('%Y-%m-%d %H:%M:%S'),
                             loader_epoch=(self.batches.next_epoch() if hasattr(self.batches, 'next_epoch')
                                           else getattr(self.batches, 'epoch', None)))
        torch.save(state, path)
        latest = os.path.join(self.work_dir, 'latest.pth')
        if os.path.lexists(latest):
            os.remove(latest)
        os.symlink(os.path.basename(path), latest)
        return path

    def _gather_point_state(self):
        """The detector keeps per-image point dictionaries (first-visit and refined points, keyed by file name) for the
        images THIS rank has seen; a checkpoint written by rank 0 alone would lose the other shards' refinements and give
        them fresh random points after a resume.  All ranks contribute theirs (one all_gather_object per checkpoint)."""
        model = self.trainer.model
        if not hasattr(model, 'get_extra_state') or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return None
        mine = model.get_extra_state()
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, mine)
        return merge_point_states(parts)

    def resume(self, path):
        """--resume-from: model (incl. count / the point dictionaries of every rank), momentum, iteration, epoch, and the
        data loader's position: the sampler continues with the shuffle of the epoch the run was in (`EpochBatches.epoch`)."""
        state = torch.load(path, map_location=self.trainer.flat.student_flat.device, weights_only=False)
        self.trainer.load_state_dict(state)
        self.epoch = state['meta']['epoch']
        le = state['meta'].get('loader_epoch')
        if le is not None and hasattr(self.batches, 'epoch'):
            self.batches.epoch, self.batches._it = int(le), None
        return state['meta']

    # --------------------------------------------------------------------------- logging --
    def _log(self, it_in_epoch, log_vars, dt):
        vals = log_vars.materialize()                       # ONE coalesced all-reduce + one D2H
        if self.rank != 0:
            return
        lr = self.trainer.sched.lr_at(max(self.trainer.iter - 1, 0))
        rec = dict(mode='train', epoch=self.epoch + 1, iter=it_in_epoch, lr=lr, time=round(dt, 4),
                   **{k: round(v, 6) for k, v in vals.items()})
        with open(self._json, 'a') as f:
            f.write(json.dumps(rec) + '\n')
        body = ', '.join(f'{k}: {v:.4f}' for k, v in vals.items())
        print(f'Epoch [{self.epoch + 1}][{it_in_epoch}/{self.iters_per_epoch}]\tlr: {lr:.3e}, time: {dt:.3f}, {body}', flush=True)

    # ------------------------------------------------------------------------------ run --
    def run(self, max_iters=None):
        """Runs until `max_epochs` (or `max_iters` further iterations); returns the number of iterations done."""
        done = 0
        while self.epoch < self.max_epochs:
            start = self.trainer.iter - self.epoch * self.iters_per_epoch
            t0 = time.perf_counter()
            stop = False
            for i in range(start, self.iters_per_epoch):
                out = self.trainer.step(self.batches(self.trainer.iter, self.batch_size))
                done += 1
                if (i + 1) % self.log_interval == 0:
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / self.log_interval
                    self._log(i + 1, out['log_vars'], dt)
                    t0 = time.perf_counter()
                if max_iters is not None and done >= max_iters:
                    stop = True
                    break
            if self.trainer.iter - self.epoch * self.iters_per_epoch >= self.iters_per_epoch:   # the epoch is complete
                self.epoch += 1
                if self.epoch % self.checkpoint_interval == 0:
                    self.save_checkpoint()
                if self._eval is not None and self.epoch % self._eval[2] == 0:
                    self.evaluate()                         # every rank: sharded test loop
            if stop:
                break
        return done
