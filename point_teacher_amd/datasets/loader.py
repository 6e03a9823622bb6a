"""Samplers, the device-side collate and the prefetching loader.

    GroupSampler / DistributedGroupSampler   /root/reference/HBB_TOD/mmdet/datasets/samplers/group_sampler.py:11-148
    DistributedSampler                       /root/reference/HBB_TOD/mmdet/datasets/samplers/distributed_sampler.py
    build_dataloader                         /root/reference/HBB_TOD/mmdet/datasets/builder.py:88-148
    collate                                  mmcv.parallel.collate (third-party): stacked images are zero-padded to
                                             the largest H, W of the batch; everything else becomes a per-image list

The reference forks `workers_per_gpu` dataloader PROCESSES per GPU that decode, resize, normalise and pad on the CPU,
pickle float32 tensors back to the trainer, and `scatter` uploads them.  One process per GPU here: `workers_per_gpu`
THREADS decode (Pillow releases the GIL) and run the host-side box/meta arithmetic, the collate uploads the uint8
bytes from pinned memory on a side stream and renders each sample into the channels-last batch tensor with
`pt_image_prep`, one batch ahead of the training step; the step's stream waits on an event, never on the host.
"""
import math
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .pipelines import DataContainer, LazyImage


def get_dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return int(os.environ.get('RANK', 0)) if 'WORLD_SIZE' in os.environ else 0, int(os.environ.get('WORLD_SIZE', 1))


class GroupSampler:
    """group_sampler.py:11-49 (draws from numpy's global state, like the reference)."""

    def __init__(self, dataset, samples_per_gpu=1):
        assert hasattr(dataset, 'flag')
        self.dataset, self.samples_per_gpu = dataset, samples_per_gpu
        self.flag = dataset.flag.astype(np.int64)
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = sum(int(np.ceil(size / samples_per_gpu)) * samples_per_gpu for size in self.group_sizes)

    def __iter__(self):
        spg = self.samples_per_gpu
        indices = []
        for i, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            indice = np.where(self.flag == i)[0]
            assert len(indice) == size
            np.random.shuffle(indice)
            num_extra = int(np.ceil(size / spg)) * spg - len(indice)
            indices.append(np.concatenate([indice, np.random.choice(indice, num_extra)]))
        indices = np.concatenate(indices)
        indices = np.concatenate([indices[i * spg:(i + 1) * spg] for i in np.random.permutation(range(len(indices) // spg))])
        indices = indices.astype(np.int64).tolist()
        assert len(indices) == self.num_samples
        return iter(indices)

    def __len__(self):
        return self.num_samples


class DistributedGroupSampler:
    """group_sampler.py:52-148: every rank derives the same permutation from (seed + epoch) and takes its slice."""

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None, seed=0):
        _rank, _world = get_dist_info()
        self.num_replicas = _world if num_replicas is None else num_replicas
        self.rank = _rank if rank is None else rank
        self.dataset, self.samples_per_gpu, self.epoch = dataset, samples_per_gpu, 0
        self.seed = seed if seed is not None else 0
        assert hasattr(dataset, 'flag')
        self.flag = dataset.flag
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = sum(int(math.ceil(s * 1.0 / samples_per_gpu / self.num_replicas)) * samples_per_gpu
                               for s in self.group_sizes)
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch + self.seed)
        spg = self.samples_per_gpu
        indices = []
        for i, size in enumerate(self.group_sizes):
            if size > 0:
                indice = np.where(self.flag == i)[0]
                assert len(indice) == size
                indice = indice[list(torch.randperm(int(size), generator=g).numpy())].tolist()
                extra = int(math.ceil(size * 1.0 / spg / self.num_replicas)) * spg * self.num_replicas - len(indice)
                tmp = indice.copy()
                for _ in range(extra // size):
                    indice.extend(tmp)
                indice.extend(tmp[:extra % size])
                indices.extend(indice)
        assert len(indices) == self.total_size
        indices = [indices[j] for i in list(torch.randperm(len(indices) // spg, generator=g))
                   for j in range(i * spg, (i + 1) * spg)]
        offset = self.num_samples * self.rank
        indices = indices[offset:offset + self.num_samples]
        assert len(indices) == self.num_samples
        return iter(indices)

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class DistributedSampler:
    """distributed_sampler.py (shuffle=False is what build_dataloader uses it for): rank r takes indices r::world of the
    dataset padded by wrap-around to a multiple of world."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=False, seed=0):
        _rank, _world = get_dist_info()
        self.num_replicas = _world if num_replicas is None else num_replicas
        self.rank = _rank if rank is None else rank
        self.dataset, self.shuffle, self.seed, self.epoch = dataset, shuffle, seed if seed is not None else 0, 0
        self.num_samples = int(math.ceil(len(dataset) * 1.0 / self.num_replicas))
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch + self.seed)
            indices = torch.randperm(len(self.dataset), generator=g).tolist()
        else:
            indices = torch.arange(len(self.dataset)).tolist()
        indices = (indices * math.ceil(self.total_size / len(indices)))[:self.total_size]
        return iter(indices[self.rank:self.total_size:self.num_replicas])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class SequentialSampler:
    def __init__(self, dataset):
        self.n = len(dataset)

    def __iter__(self):
        return iter(range(self.n))

    def __len__(self):
        return self.n

    def set_epoch(self, epoch):
        pass


# ----------------------------------------------------------------------------------------------- collate on the device
def _unwrap(x):
    return x.data if isinstance(x, DataContainer) else x


def _render_batch(images, device, channels_last, stream, cache=None):
    """[LazyImage] -> float32 [B, 3, Hmax, Wmax] on `device`: one pinned uint8 upload (none for an image that comes out of
    the HBM cache) + one pt_image_prep per image."""
    H = max(im.shape[0] for im in images)
    W = max(im.shape[1] for im in images)
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    out = torch.empty((len(images), 3, H, W), dtype=torch.float32, device=device, memory_format=fmt)
    for b, im in enumerate(images):
        if isinstance(im.src, torch.Tensor):
            dsrc = im.src
        else:
            host = torch.from_numpy(im.src)
            pinned = torch.empty(host.shape, dtype=torch.uint8, pin_memory=True)
            pinned.copy_(host)
            dsrc = pinned.to(device, non_blocking=True)
            if cache is not None and im.cache_key is not None:
                cache.put(im.cache_key, dsrc)
        im.render(out[b], device_src=dsrc, stream=stream.cuda_stream)
    return out


def collate_to_device(samples, device, channels_last=True, stream=None, cache=None):
    """mmcv.parallel.collate + MMDataParallel.scatter for one GPU: a list of pipeline outputs -> the keyword
    arguments of `forward_train` / `forward_test`, already resident on `device`.
    train sample: {'img_metas': DC(meta), 'img': DC(LazyImage), 'gt_bboxes': DC(tensor), ...}
    test sample (MultiScaleFlipAug): {'img_metas': [DC(meta)...], 'img': [LazyImage | DC(LazyImage)...]} per augmentation."""
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('collate_to_device renders images with pt_image_prep: a CUDA/HIP device is required (no CPU path)')
    stream = stream or torch.cuda.current_stream(device)
    out = {}
    with torch.cuda.stream(stream):
        for key in samples[0]:
            vals = [s[key] for s in samples]
            if isinstance(vals[0], list):                                   # test mode: one entry per augmentation
                cols = []
                for a in range(len(vals[0])):
                    col = [_unwrap(v[a]) for v in vals]
                    cols.append(_render_batch(col, device, channels_last, stream, cache) if isinstance(col[0], LazyImage) else col)
                out[key] = cols
                continue
            data = [_unwrap(v) for v in vals]
            if isinstance(data[0], LazyImage):
                out[key] = _render_batch(data, device, channels_last, stream, cache)
            elif isinstance(data[0], torch.Tensor):
                out[key] = [d.to(device, non_blocking=True) for d in data]
            else:
                out[key] = data                                               # img_metas and other cpu_only payloads
    return out


class DeviceLoader:
    """Iterable over device-resident batches.  `len()` = batches per epoch; `set_epoch` reseeds a distributed sampler."""

    def __init__(self, dataset, sampler, batch_size, num_workers, device, channels_last=True, drop_last=False, prefetch=2,
                 cache_bytes=0):
        """cache_bytes > 0: keep the decoded uint8 tiles in HBM (pipelines.DeviceImageCache) up to that many bytes."""
        self.dataset, self.sampler, self.batch_size, self.device = dataset, sampler, int(batch_size), torch.device(device)
        self.cache = None
        if cache_bytes:
            from .pipelines import DeviceImageCache, LoadImageFromFile, MultiScaleFlipAug
            self.cache = DeviceImageCache(cache_bytes)

            def install(compose):
                for t in compose.transforms:
                    if isinstance(t, LoadImageFromFile):
                        t.cache = self.cache
                    elif isinstance(t, MultiScaleFlipAug):
                        install(t.transforms)
            install(dataset.pipeline)
        self.channels_last, self.drop_last, self.prefetch = channels_last, drop_last, max(int(prefetch), 1)
        self.num_workers = max(int(num_workers), 1)
        self._pool = ThreadPoolExecutor(max_workers=self.num_workers, thread_name_prefix='pt-data')
        self._stream = torch.cuda.Stream(self.device) if self.device.type == 'cuda' else None

    def __len__(self):
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def set_epoch(self, epoch):
        if hasattr(self.sampler, 'set_epoch'):
            self.sampler.set_epoch(epoch)

    def _index_batches(self):
        batch = []
        for i in self.sampler:
            batch.append(int(i))
            if len(batch) == self.batch_size:
                yield batch
                batch = []
        if batch and not self.drop_last:
            yield batch

    def _finish(self, futures):
        samples = [f.result() for f in futures]
        batch = collate_to_device(samples, self.device, self.channels_last, self._stream, self.cache)
        ready = torch.cuda.Event()
        ready.record(self._stream)
        return batch, ready

    def _hand_over(self, batch, ready):
        """The batch was produced on the loader's stream; make the consumer's stream wait for it and tell the caching
        allocator that the consumer uses these blocks (they were allocated from the loader stream's pool)."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready)

        def mark(v):
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(cur)
            elif isinstance(v, (list, tuple)):
                for x in v:
                    mark(x)
        for v in batch.values():
            mark(v)
        return batch

    def __iter__(self):
        pending, ready = [], []                   # batches being decoded by the threads / batches already rendered, oldest first
        for idx in self._index_batches():
            pending.append([self._pool.submit(self.dataset.__getitem__, i) for i in idx])
            if len(pending) > self.prefetch:
                ready.append(self._finish(pending.pop(0)))
            # render one batch ahead: while the consumer trains on batch k, batch k+1 is already uploaded and prepared on the
            # loader's stream (only when its decode is finished - otherwise hand batch k over first)
            if pending and not ready and all(f.done() for f in pending[0]):
                ready.append(self._finish(pending.pop(0)))
            while len(ready) > 1:
                yield self._hand_over(*ready.pop(0))
        while pending:
            ready.append(self._finish(pending.pop(0)))
            while len(ready) > 1:
                yield self._hand_over(*ready.pop(0))
        while ready:
            yield self._hand_over(*ready.pop(0))


def build_dataloader(dataset, samples_per_gpu, workers_per_gpu, num_gpus=1, dist=True, shuffle=True, seed=None, device=None,
                     channels_last=True, **kwargs):
    """builder.py:88-141 with the reference's sampler choice; one process drives one GPU, so `num_gpus` must be 1."""
    if num_gpus != 1:
        raise NotImplementedError('one process per GPU: launch with torch.distributed.run instead of num_gpus > 1')
    rank, world = get_dist_info()
    if dist:
        sampler = (DistributedGroupSampler(dataset, samples_per_gpu, world, rank, seed=seed) if shuffle
                   else DistributedSampler(dataset, world, rank, shuffle=False, seed=seed))
    else:
        sampler = GroupSampler(dataset, samples_per_gpu) if shuffle else SequentialSampler(dataset)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device())
    return DeviceLoader(dataset, sampler, samples_per_gpu, workers_per_gpu, device, channels_last=channels_last, **kwargs)


class EpochBatches:
    """Adapter for runner.Runner / Trainer loops, which ask for `batches(iteration, batch_size)`: walks the loader epoch
    after epoch (EpochBasedRunner.train: `data_loader.sampler.set_epoch(epoch)` then one pass)."""

    def __init__(self, loader, start_epoch=0):
        self.loader, self.epoch, self._it, self._served = loader, int(start_epoch), None, 0

    def next_epoch(self):
        """The loader epoch a resumed run must start with: `epoch` only advances when the NEXT call finds the iterator empty,
        so after the last batch of epoch e (where end-of-epoch checkpoints are written) it still reads e although e is used up
        - resuming with it would replay e's shuffle and shift every later epoch by one."""
        return self.epoch + 1 if (self._it is not None and self._served >= len(self.loader)) else self.epoch

    def __call__(self, iteration=None, batch_size=None):
        while True:
            if self._it is None:
                self.loader.set_epoch(self.epoch)
                self._it = iter(self.loader)
                self._served = 0
            try:
                b = next(self._it)
                self._served += 1
                return b
            except StopIteration:
                self._it = None
                self.epoch += 1
