"""Training runtime around `TS_P2B_FCOS.train_step`: flat parameter storage, the fused
grad-clip + SGD step, the teacher EMA, the data-parallel gradient exchange and the LR policy.

It replaces what the reference gets from mmcv (EpochBasedRunner + OptimizerHook +
DefaultOptimizerConstructor + MMDistributedDataParallel, driven by
HBB_TOD/mmdet/apis/train.py:73-170 and the `optimizer*`/`lr_config` entries of
configs/point_teacher/aitodv2_point_teacher_0%.py:212-223) with an MI355X-first layout:

* every student parameter lives in ONE flat fp32 buffer `[trainable weights | trainable
  biases | frozen]`, the teacher in a second buffer with the same order, gradients and
  momentum in two more.  EMA, gradient norm and the SGD update are one streaming kernel each
  (libpt_hip.so) instead of ~190 per-tensor launch pairs;
* data parallel = one process per GPU; the flat gradient buffer is all-reduced over RCCL in a
  few large chunks on a side stream (xGMI is point-to-point: large messages, few of them).
  The teacher is frozen and never enters the exchange (the reference all-reduces its 352 MB
  of never-written gradients as well);
* no host synchronisation anywhere in `step()`.
"""

import os

import torch
import torch.distributed as dist

from . import functional as F


class FlatParams:
    """Re-homes the parameters of `model.student` / `model.teacher` into flat buffers."""

    def __init__(self, model, channels_last=False):
        """channels_last=True stores 4-D (convolution) weights in [O,H,W,I] order inside the flat
        buffers and exposes them as channels_last-strided views, so MIOpen's NHWC kernels run
        without per-call layout transposes; the flat-buffer kernels are order-agnostic."""
        self.model = model
        self.channels_last = channels_last
        # a plain detector (the supervised FCOS baseline, row N4) has no teacher: its own parameters are the "student"
        self.has_teacher = hasattr(model, 'student') and hasattr(model, 'teacher')
        student = list((model.student if self.has_teacher else model).named_parameters())
        teacher = dict(model.teacher.named_parameters()) if self.has_teacher else None
        assert teacher is None or [n for n, _ in student] == list(teacher.keys()), 'teacher/student parameter lists differ'

        # mmcv DefaultOptimizerConstructor (optimizer/default_constructor.py): bias_lr_mult / bias_decay_mult apply to
        # parameters NAMED 'bias' of every module EXCEPT normalisation layers; norm weights and biases keep lr x 1 and take
        # weight_decay x norm_decay_mult (1 unless a config says otherwise) - i.e. they belong to the weights segment.
        root = model.student if self.has_teacher else model
        norm_owned = {id(p) for m in root.modules() if isinstance(m, (torch.nn.modules.batchnorm._BatchNorm, torch.nn.GroupNorm,
                                                                    torch.nn.LayerNorm, torch.nn.modules.instancenorm._InstanceNorm))
                      for p in m.parameters(recurse=False)}

        def is_bias(name, p):
            return (name.endswith('.bias') or name == 'bias') and id(p) not in norm_owned
        weights = [(n, p) for n, p in student if p.requires_grad and not is_bias(n, p)]
        biases = [(n, p) for n, p in student if p.requires_grad and is_bias(n, p)]
        frozen = [(n, p) for n, p in student if not p.requires_grad]
        self.order = weights + biases + frozen

        def padded(entries):            # 16-byte aligned segments so every view is float4-friendly
            return sum((p.numel() + 3) // 4 * 4 for _, p in entries)
        self.n_weights, self.n_biases, self.n_frozen = padded(weights), padded(biases), padded(frozen)
        self.n_train = self.n_weights + self.n_biases
        total = self.n_train + self.n_frozen
        dev = student[0][1].device
        self.student_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.teacher_flat = torch.zeros(total, dtype=torch.float32, device=dev) if self.has_teacher else None
        self.grad_flat = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.mom_flat = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.numel = sum(p.numel() for _, p in self.order)
        off = 0
        self.slices = {}
        self.train_params, self.grad_views = [], []
        with torch.no_grad():
            for name, p in self.order:
                n = p.numel()
                t = teacher[name] if teacher is not None else None
                if channels_last and p.dim() == 4:
                    O, I, KH, KW = p.shape

                    def view(buf):
                        return buf[off:off + n].view(O, KH, KW, I).permute(0, 3, 1, 2)
                    view(self.student_flat).copy_(p.data)
                    if t is not None:
                        view(self.teacher_flat).copy_(t.data)
                else:
                    def view(buf):
                        return buf[off:off + n].view(p.shape)
                    self.student_flat[off:off + n].copy_(p.data.reshape(-1))
                    if t is not None:
                        self.teacher_flat[off:off + n].copy_(t.data.reshape(-1))
                p.data = view(self.student_flat)
                if t is not None:
                    t.data = view(self.teacher_flat)
                if p.requires_grad:
                    p.grad = view(self.grad_flat)
                    self.train_params.append(p)
                    self.grad_views.append(p.grad)
                self.slices[name] = (off, n)
                off += (n + 3) // 4 * 4
        if self.has_teacher:
            model._flat = (self.teacher_flat, self.student_flat)
            model._flat_n_train = self.n_train          # [n_train, total) = the frozen parameters
            model._stem_shared = None

    def zero_grad(self):
        self.grad_flat.zero_()

    def detach_grads(self):
        """Let autograd hand over freshly computed gradients instead of adding them, one small launch per
        parameter, into the (zeroed) flat buffer: with `.grad = None` AccumulateGrad keeps the incoming tensor."""
        for p in self.train_params:
            p.grad = None

    def gather_grads(self):
        """Copy the gradients autograd left in `.grad` into the flat buffer with multi-tensor copies (segments of
        parameters that received none stay zero), then drop them."""
        dst, src = [], []
        for p, v in zip(self.train_params, self.grad_views):
            if p.grad is not None:
                dst.append(v)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.train_params, self.grad_views):
            p.grad = v

    def check_views(self):
        """Autograd must have accumulated in place: every .grad still aliases the flat buffer."""
        base = self.grad_flat.data_ptr()
        for name, p in self.order:
            if p.requires_grad:
                off, n = self.slices[name]
                if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                    return False
        return True


class BucketedGradExchange:
    """Mean of the flat gradient over the data-parallel ranks, OVERLAPPED with backward.

    The trainable segment of the flat gradient buffer is cut into `n_buckets` contiguous ranges on parameter
    boundaries.  A post-accumulate hook on every parameter counts arrivals; when the last gradient of a bucket
    has been produced the bucket is copied into the flat buffer with one multi-tensor copy and its all-reduce
    (RCCL over xGMI; a few large messages, xGMI being point-to-point) is issued on a side stream while autograd
    keeps computing the earlier layers.  `finish()` flushes buckets whose parameters received no gradient in
    this iteration (their segment stays zero) and joins the side stream.  gloo / CPU: same logic, synchronous.

    Collectives are ISSUED IN A FIXED ORDER (`issue_order`: buckets by the model position of their earliest parameter,
    latest first - the order backward completes them in when every parameter receives a gradient).  A completed bucket
    waits until every bucket ahead of it in that order has been issued, so two ranks on which gradients arrive in
    different orders, or on which a parameter receives no gradient at all (an empty image, a data-dependent branch),
    still enqueue the same sequence of equally sized all-reduces; `finish()` issues whatever is left in the same order
    and asserts the sequence."""

    def __init__(self, flat, n_buckets=6, device=None):
        self.flat = flat
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.stream = torch.cuda.Stream(device=device) if (device is not None and device.type == 'cuda') else None
        total = flat.n_train
        target = max(total // max(int(n_buckets), 1), 1)
        self.buckets = []                      # [start, end, [(param, view)]]
        cur, start, acc = [], 0, 0
        off = 0
        for p, v in zip(flat.train_params, flat.grad_views):
            n = (p.numel() + 3) // 4 * 4
            cur.append((p, v))
            off += n
            acc += n
            if acc >= target:
                self.buckets.append([start, off, cur])
                cur, start, acc = [], off, 0
        if cur:
            self.buckets.append([start, off, cur])
        assert off == total, (off, total)
        self.bucket_of = {id(p): b for b, (_, _, ps) in enumerate(self.buckets) for p, _ in ps}
        pos = {id(p): i for i, p in enumerate((flat.model.student if flat.has_teacher else flat.model).parameters())}
        first = [min(pos[id(p)] for p, _ in ps) for _, _, ps in self.buckets]
        self.issue_order = sorted(range(len(self.buckets)), key=lambda b: -first[b])
        self.left, self.done, self.active = [0] * len(self.buckets), [True] * len(self.buckets), False
        self.ready, self.next, self.issued = [False] * len(self.buckets), 0, []
        for p in flat.train_params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def begin(self):
        self.left = [len(ps) for _, _, ps in self.buckets]
        self.done = [False] * len(self.buckets)
        self.ready = [False] * len(self.buckets)
        self.next, self.issued = 0, []
        self.active = True

    def _on_grad(self, p):
        if not self.active:
            return
        b = self.bucket_of[id(p)]
        self.left[b] -= 1
        if self.left[b] == 0:
            self.ready[b] = True
            self._issue_ready()

    def _issue_ready(self):
        while self.next < len(self.issue_order) and self.ready[self.issue_order[self.next]]:
            self._flush(self.issue_order[self.next])
            self.next += 1

    def _flush(self, b):
        start, end, ps = self.buckets[b]
        dst, src = [], []
        for p, v in ps:
            if p.grad is not None:
                dst.append(v)
                src.append(p.grad)
        if dst:
            with torch.no_grad():
                torch._foreach_copy_(dst, src)
        for p, _ in ps:
            p.grad = None                      # the copy in the flat buffer is the gradient from here on
        g = self.flat.grad_flat[start:end]
        if self.world > 1:
            if self.stream is None:
                g.div_(self.world)
                dist.all_reduce(g)
            else:
                self.stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.stream):
                    g.div_(self.world)
                    dist.all_reduce(g)
        self.done[b] = True
        self.issued.append(b)

    def finish(self):
        while self.next < len(self.issue_order):            # buckets still waiting for a gradient that never came, in order
            self._flush(self.issue_order[self.next])
            self.next += 1
        assert self.issued == self.issue_order, (self.issued, self.issue_order)
        self.active = False
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        for p, v in zip(self.flat.train_params, self.flat.grad_views):
            p.grad = v


TUNED_GEMMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuned', 'gemm_gfx950.csv')


def enable_tuned_gemms(path=TUNED_GEMMS):
    """The MIL FC stacks are plain library GEMMs (12544 -> 1024 -> 1024 over K RoIs, forward / dgrad / wgrad).  PyTorch's TunableOp
    picks a rocBLAS / hipBLASLt solution per shape from a results file: `tuned/gemm_gfx950.csv` was recorded on an MI355X with this
    image's library versions (its Validator lines are checked by PyTorch; on a mismatch, or for shapes it does not hold, the
    library's default heuristic is used).  Nothing is tuned at run time.  The file is recorded by `tools/tune_gemms.sh`
    (PT_TUNE_GEMMS=1) WITH the numerical check on: some hipBLASLt solutions offered for fp32 compute with reduced-precision
    products (3e-3 off the default kernel's result) and a table recorded without the check selects them - see
    tests/test_tuned_gemms.py, which compares every shape of the table against the default solution.  PT_TUNED_GEMMS=0: off."""
    if not torch.cuda.is_available():
        return False
    import torch.cuda.tunable as tunable
    if os.environ.get('PT_TUNE_GEMMS', '0') == '1':                  # recording run
        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_numerical_check_tolerances(True, 1e-4, 1e-4)
        tunable.set_filename(os.environ.get('PT_TUNE_GEMMS_OUT', 'gemm_gfx950.csv'))
        return True
    if os.environ.get('PT_TUNED_GEMMS', '1') == '0' or not os.path.exists(path):
        return False
    tunable.enable(True)
    tunable.tuning_enable(False)
    return bool(tunable.read_file(path))


class StepLR:
    """mmcv StepLrUpdaterHook with warm-up, by_epoch=True (lr_config of the configs)."""

    def __init__(self, base_lr, step, gamma=0.1, warmup=None, warmup_iters=0, warmup_ratio=0.1, iters_per_epoch=1):
        self.base_lr, self.step, self.gamma = base_lr, list(step), gamma
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio
        self.iters_per_epoch = max(int(iters_per_epoch), 1)

    def lr_at(self, it):
        epoch = it // self.iters_per_epoch
        lr = self.base_lr * self.gamma ** sum(epoch >= s for s in self.step)
        if self.warmup is not None and it < self.warmup_iters:
            if self.warmup == 'constant':
                lr = lr * self.warmup_ratio
            elif self.warmup == 'linear':
                lr = lr * (1 - (1 - it / self.warmup_iters) * (1 - self.warmup_ratio))
            elif self.warmup == 'exp':
                lr = lr * self.warmup_ratio ** (1 - it / self.warmup_iters)
        return lr


class Trainer:
    """One object = model + flat storage + optimizer + (optional) data-parallel exchange."""

    def __init__(self, model, optimizer_cfg, optimizer_config=None, lr_config=None, iters_per_epoch=1000,
                 grad_chunks=4, autocast_dtype=None, channels_last=False):
        assert optimizer_cfg.get('type', 'SGD') == 'SGD', 'the Point-Teacher recipe is SGD'
        self.model = model
        if autocast_dtype is not None and not channels_last:
            # The bf16 backbone is only built for [B,H,W,C] activations (fused bf16 BatchNorm epilogue, MIOpen's NHWC
            # kernels).  MIOpen's NCHW bf16 path was measured 13 % away from a bf16-rounding oracle on the PSAGG
            # features (NHWC: 1.2 %, profiles/r02/bf16_accuracy.txt), so the layout is not left to the caller.
            channels_last = True
        self.flat = FlatParams(model, channels_last=channels_last)
        self.channels_last = channels_last
        self.momentum = optimizer_cfg.get('momentum', 0.0)
        self.weight_decay = optimizer_cfg.get('weight_decay', 0.0)
        pw = optimizer_cfg.get('paramwise_cfg', {}) or {}
        self.bias_lr_mult, self.bias_decay_mult = pw.get('bias_lr_mult', 1.0), pw.get('bias_decay_mult', 1.0)
        assert pw.get('norm_decay_mult', 1.0) == 1.0 and not pw.get('custom_keys'), \
            'paramwise_cfg: only bias_lr_mult / bias_decay_mult are on the Point-Teacher path (norm parameters decay like weights)'
        clip = (optimizer_config or {}).get('grad_clip') or {}
        assert clip.get('norm_type', 2) == 2
        self.max_norm = float(clip.get('max_norm', 0.0))
        lc = dict(lr_config or {})
        lc.pop('policy', None)
        self.sched = StepLR(optimizer_cfg['lr'], lc.get('step', []), lc.get('gamma', 0.1), lc.get('warmup'),
                            lc.get('warmup_iters', 0), lc.get('warmup_ratio', 0.1), iters_per_epoch)
        dev = self.flat.student_flat.device
        self.lr_t = torch.zeros(1, dtype=torch.float32, device=dev)
        self._lr_host = None
        self.iter = 0
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        # N > 1: bucketed exchange overlapped with backward; N == 1: nothing to exchange, one gather after backward
        self.exchange = BucketedGradExchange(self.flat, max(grad_chunks, 1) + 2, dev) if self.world > 1 else None
        # BASELINE configs[2] "bf16 backbone + fp32 head": autocast covers backbone / FPN / PSAGG only (Student_FCOS.extract_feat)
        self.autocast_dtype = autocast_dtype
        for m in model.modules():
            if hasattr(m, 'backbone') and hasattr(m, 'extract_feat'):
                m.backbone_autocast = autocast_dtype
        self.tuned_gemms = enable_tuned_gemms()
        self._broadcast_initial_state()

    def _broadcast_initial_state(self):
        if self.world > 1:                     # identical initial weights on every rank, as DDP does
            dist.broadcast(self.flat.student_flat, src=0)
            if self.flat.teacher_flat is not None:
                dist.broadcast(self.flat.teacher_flat, src=0)

    def _set_lr(self):
        lr = self.sched.lr_at(self.iter)
        if lr != self._lr_host:                # device scalar is rewritten only when the value changes
            self.lr_t.fill_(lr)
            self._lr_host = lr

    def step(self, data):
        """One training iteration: zero grads, train_step (EMA happens inside, at its start),
        backward, gradient exchange, clip + SGD.  Returns train_step's dict."""
        self._set_lr()
        self.flat.zero_grad()
        self.flat.detach_grads()
        if self.channels_last:
            data = dict(data, img=data['img'].contiguous(memory_format=torch.channels_last))
        out = self.model.train_step(data, None)     # reduced precision, if any, is scoped inside extract_feat
        if self.exchange is not None:
            self.exchange.begin()
            out['loss'].backward()
            self.exchange.finish()
        else:
            out['loss'].backward()
            self.flat.gather_grads()
        f = self.flat
        sq = F.grad_sqnorm(f.grad_flat) if self.max_norm > 0 else None
        F.sgd_step_(f.student_flat[:f.n_train], f.grad_flat, f.mom_flat, f.n_weights, self.lr_t, self.momentum,
                    self.weight_decay, self.bias_lr_mult, self.bias_decay_mult, sq, self.max_norm, self.iter == 0)
        self.iter += 1
        return out

    def state_dict(self):
        return dict(model=self.model.state_dict(), momentum=self.flat.mom_flat.clone(), iter=self.iter)

    def load_state_dict(self, sd):
        self.model.load_state_dict(sd['model'])
        if hasattr(self.model, '_stem_shared'):
            self.model._stem_shared = None           # re-decided from the loaded state
        self.flat.mom_flat.copy_(sd['momentum'])
        self.iter = sd['iter']
