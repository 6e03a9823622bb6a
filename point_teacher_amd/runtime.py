"""Training runtime around `TS_P2B_FCOS.train_step`: flat parameter storage, the fused
grad-clip + SGD step, the teacher EMA, the data-parallel gradient exchange and the LR policy.

It replaces what the reference gets from mmcv (EpochBasedRunner + OptimizerHook +
DefaultOptimizerConstructor + MMDistributedDataParallel, driven by
HBB_TOD/mmdet/apis/train.py:73-170 and the `optimizer*`/`lr_config` entries of
configs/point_teacher/aitodv2_point_teacher_0%.py:212-223) with an MI355X-first layout:

* every student parameter lives in ONE flat fp32 buffer `[parameter groups of the optimizer
  (weights | biases | ...) | never-used | frozen]`, the teacher in a second buffer with the same
  order, gradients and momentum of the live groups in two more.  EMA, gradient norm and the SGD update are one streaming kernel each
  (libpt_hip.so) instead of ~190 per-tensor launch pairs;
* data parallel = one process per GPU; the flat gradient buffer is all-reduced over RCCL in a
  few large chunks on a side stream (xGMI is point-to-point: large messages, few of them).
  The teacher is frozen and never enters the exchange (the reference all-reduces its 352 MB
  of never-written gradients as well);
* no host synchronisation anywhere in `step()`.
"""

import os
import time

import torch
import torch.distributed as dist

from . import functional as F


_NORMS = (torch.nn.modules.batchnorm._BatchNorm, torch.nn.GroupNorm, torch.nn.LayerNorm,
          torch.nn.modules.instancenorm._InstanceNorm)
MAX_GROUPS = 8                      # PT_MAX_PARAM_GROUPS of include/pt_hip.h


def param_multipliers(root, paramwise_cfg=None, prefix=''):
    """{parameter name: (lr_mult, decay_mult)} by the rules of mmcv's DefaultOptimizerConstructor.add_params
    (mmcv/runner/optimizer/default_constructor.py - the constructor every shipped `optimizer = dict(type='SGD', paramwise_cfg=...)`
    selects; un-vendored, restated from the published source):
      * `custom_keys`: the LONGEST key (ties: alphabetical) that is a substring of the full parameter name wins and sets
        lr_mult / decay_mult (1 when absent); nothing else applies to that parameter
        (configs/baselines/aitodv2_yolof_r50_1x.py:70-71: {'backbone': lr_mult 1/3});
      * a parameter NAMED `bias` takes bias_lr_mult unless it belongs to a normalisation layer or to a deformable-convolution
        module (incl. its `conv_offset` child); a Conv2d named `conv_offset` inside a DCN module takes dcn_offset_lr_mult (weight and bias);
      * decay: norm layers x norm_decay_mult, else depth-wise convolutions x dwconv_decay_mult, else a `bias` outside DCN
        modules x bias_decay_mult.
    `prefix` is what the reference's names carry in front (the optimizer is built over the whole wrapper, apis/train.py:88:
    'student.' for the teacher-student detectors) - custom keys see it."""
    pw = dict(paramwise_cfg or {})
    custom = pw.get('custom_keys', {}) or {}
    keys = sorted(sorted(custom.keys()), key=len, reverse=True)
    bias_lr, bias_wd = pw.get('bias_lr_mult', 1.), pw.get('bias_decay_mult', 1.)
    norm_wd, dw_wd, dcn_lr = pw.get('norm_decay_mult', 1.), pw.get('dwconv_decay_mult', 1.), pw.get('dcn_offset_lr_mult', 1.)
    from .nn_modules import DeformConv2dPack, ModulatedDeformConv2dPack
    out = {}

    def walk(module, mprefix, in_dcn):
        is_norm = isinstance(module, _NORMS)
        is_dw = isinstance(module, torch.nn.Conv2d) and module.in_channels == module.groups
        in_dcn = in_dcn or isinstance(module, (DeformConv2dPack, ModulatedDeformConv2dPack))
        for name, p in module.named_parameters(recurse=False):
            full = f'{mprefix}.{name}' if mprefix else name
            lr_m, wd_m = 1., 1.
            for k in keys:
                if k in prefix + full:
                    lr_m, wd_m = custom[k].get('lr_mult', 1.), custom[k].get('decay_mult', 1.)
                    break
            else:
                if name == 'bias' and not (is_norm or in_dcn):
                    lr_m = bias_lr
                if 'conv_offset' in mprefix and in_dcn and isinstance(module, torch.nn.Conv2d):
                    lr_m = dcn_lr
                if is_norm:
                    wd_m = norm_wd
                elif is_dw:
                    wd_m = dw_wd
                elif name == 'bias' and not in_dcn:
                    wd_m = bias_wd
            out[full] = (float(lr_m), float(wd_m))
        for cname, child in module.named_children():
            walk(child, f'{mprefix}.{cname}' if mprefix else cname, in_dcn)
    walk(root, '', False)
    return out


class FlatParams:
    """Re-homes the parameters of `model.student` / `model.teacher` into flat buffers

        [ group 0 | group 1 | ... | dead | frozen ]           (every segment 16-byte aligned)

    * group g = the live trainable parameters that share one (lr multiplier, decay multiplier) pair of the optimizer's
      paramwise_cfg (`param_multipliers`); group 0 is the plain (1, 1) group - the "weights" -, bias_lr_mult=2 / bias_decay_mult=0
      of the Point-Teacher configs makes group 1 the "biases".  Gradient and momentum buffers cover the groups only (`n_train`).
    * dead = trainable parameters that have never received a gradient (`shared_fcs`, `shared_fcs_refine`, `fc_iou` of
      TS_P2BFCOSHead are constructed, fcos_head_p2b_ts.py:147-181, and no forward reaches them: 27.8 M of the student's 88 M).
      torch.optim.SGD skips a parameter whose `.grad is None` - no weight decay, no momentum - and DDP has nothing to reduce for
      it; here they have no gradient / momentum storage, stay out of the gradient exchange and out of the fused clip+SGD launch.
      They still follow the EMA (update_teacher_model runs over parameters()).  The set is discovered by the Trainer at the
      first iteration (`relayout`); a dead parameter that later receives a gradient is revived the same way.
    * frozen = requires_grad False."""

    def __init__(self, model, channels_last=False, paramwise_cfg=None, dead=()):
        """channels_last=True stores 4-D (convolution) weights in [O,H,W,I] order inside the flat
        buffers and exposes them as channels_last-strided views, so MIOpen's NHWC kernels run
        without per-call layout transposes; the flat-buffer kernels are order-agnostic."""
        self.model = model
        self.channels_last = channels_last
        self.paramwise_cfg = dict(paramwise_cfg) if paramwise_cfg is not None else dict(bias_lr_mult=1., bias_decay_mult=1.)
        # a plain detector (the supervised FCOS baseline, row N4) has no teacher: its own parameters are the "student"
        self.has_teacher = hasattr(model, 'student') and hasattr(model, 'teacher')
        self.root = model.student if self.has_teacher else model
        self.mults = param_multipliers(self.root, self.paramwise_cfg, prefix='student.' if self.has_teacher else '')
        self.student_flat = self.teacher_flat = self.grad_flat = self.mom_flat = None
        self.slices = {}
        self._build(frozenset(dead))

    # ------------------------------------------------------------------------------------------------ layout --
    def _build(self, dead):
        model = self.model
        student = list(self.root.named_parameters())
        teacher = dict(model.teacher.named_parameters()) if self.has_teacher else None
        assert teacher is None or [n for n, _ in student] == list(teacher.keys()), 'teacher/student parameter lists differ'
        unknown = set(dead) - {n for n, p in student if p.requires_grad}
        assert not unknown, f'dead parameters that are not trainable parameters of the model: {sorted(unknown)[:4]}'
        keys = [(1.0, 1.0)]                                    # group 0 = the plain group, others in order of appearance
        for n, p in student:
            if p.requires_grad and n not in dead and self.mults[n] not in keys:
                keys.append(self.mults[n])
        assert len(keys) <= MAX_GROUPS, f'paramwise_cfg yields {len(keys)} (lr_mult, decay_mult) groups; the fused step holds {MAX_GROUPS}'
        groups = [[(n, p) for n, p in student if p.requires_grad and n not in dead and self.mults[n] == k] for k in keys]
        dead_l = [(n, p) for n, p in student if p.requires_grad and n in dead]
        frozen = [(n, p) for n, p in student if not p.requires_grad]
        self.dead = frozenset(dead)
        self.group_mults = keys
        self.order = [e for g in groups for e in g] + dead_l + frozen

        def padded(entries):            # 16-byte aligned segments so every view is float4-friendly
            return sum((p.numel() + 3) // 4 * 4 for _, p in entries)
        sizes = [padded(g) for g in groups]
        self.group_ends = [sum(sizes[:i + 1]) for i in range(len(sizes))]
        self.n_train = self.group_ends[-1]                                  # live trainable elements (gradient / momentum extent)
        self.n_weights, self.n_biases = sizes[0], self.n_train - sizes[0]   # the two groups of the Point-Teacher configs
        self.n_dead, self.n_frozen = padded(dead_l), padded(frozen)
        self.frozen_start = self.n_train + self.n_dead
        total = self.frozen_start + self.n_frozen
        dev = student[0][1].device
        old = (self.student_flat, self.teacher_flat, self.grad_flat, self.mom_flat, dict(self.slices))
        self.student_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.teacher_flat = torch.zeros(total, dtype=torch.float32, device=dev) if self.has_teacher else None
        self.grad_flat = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.mom_flat = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.numel = sum(p.numel() for _, p in self.order)
        off = 0
        self.slices = {}
        self.train_params, self.grad_views, self.dead_params = [], [], []
        self._group_tables = None
        with torch.no_grad():
            for name, p in self.order:
                n = p.numel()
                t = teacher[name] if teacher is not None else None
                if self.channels_last and p.dim() == 4:
                    O, I, KH, KW = p.shape

                    def view(buf):
                        return buf[off:off + n].view(O, KH, KW, I).permute(0, 3, 1, 2)
                else:
                    def view(buf):
                        return buf[off:off + n].view(p.shape)
                view(self.student_flat).copy_(p.data)
                if t is not None:
                    view(self.teacher_flat).copy_(t.data)
                p.data = view(self.student_flat)
                if t is not None:
                    t.data = view(self.teacher_flat)
                if p.requires_grad and name not in dead:
                    if old[2] is not None and name in old[4] and old[4][name][0] + n <= old[2].numel():
                        o0 = old[4][name][0]                # a re-layout keeps gradient and momentum of parameters that stay live
                        self.grad_flat[off:off + n].copy_(old[2][o0:o0 + n])
                        self.mom_flat[off:off + n].copy_(old[3][o0:o0 + n])
                    p.grad = view(self.grad_flat)
                    self.train_params.append(p)
                    self.grad_views.append(p.grad)
                elif p.requires_grad:
                    p.grad = None
                    self.dead_params.append((name, p))
                self.slices[name] = (off, n)
                off += (n + 3) // 4 * 4
        self.name_of = {id(p): n for n, p in self.order}
        if self.has_teacher:
            model._flat = (self.teacher_flat, self.student_flat)
            model._flat_frozen_start = self.frozen_start     # [frozen_start, total) = the frozen parameters
            model._stem_shared = None
        F.PARAM_EPOCH[0] += 1                                 # cached views of parameter storage (fused BN affines) are stale

    def relayout(self, dead):
        """Move parameters between the live groups and the dead segment (values, gradients and momentum of parameters that
        stay live are carried over; a revived parameter starts with zero gradient and momentum, as torch.optim.SGD creates
        its momentum buffer at the first step that sees a gradient).  -> True if the layout changed."""
        dead = frozenset(dead)
        if dead == self.dead:
            return False
        self._build(dead)
        return True

    def group_tables(self):
        """[host] arrays of pt_sgd_step_groups: ends, lr multipliers, decay multipliers."""
        if self._group_tables is None:
            import ctypes
            ends = (ctypes.c_int64 * len(self.group_ends))(*self.group_ends)
            self._group_tables = (ends, F.hip.host_floats([k[0] for k in self.group_mults]),
                                  F.hip.host_floats([k[1] for k in self.group_mults]), len(self.group_ends))
        return self._group_tables

    def zero_grad(self):
        self.grad_flat.zero_()

    def detach_grads(self):
        """Let autograd hand over freshly computed gradients instead of adding them, one small launch per
        parameter, into the (zeroed) flat buffer: with `.grad = None` AccumulateGrad keeps the incoming tensor."""
        for p in self.train_params:
            p.grad = None

    def take_revived(self, keep=None):
        """Names of dead parameters autograd produced a gradient for in the backward that just ran (host-side knowledge:
        `.grad` is no longer None - no synchronisation).  `keep` (a dict) receives / accumulates their gradient tensors, so that
        the revival applies them instead of dropping them (round-3 advice); without it they are dropped."""
        out = []
        for name, p in self.dead_params:
            if p.grad is not None:
                out.append(name)
                if keep is not None:
                    keep[name] = p.grad.detach() if name not in keep else keep[name] + p.grad.detach()
                p.grad = None
        return out

    def add_to_grad(self, name, g):
        """Add `g` (logical [O, I, KH, KW] order) to the flat gradient slice of the (live) parameter `name`.  Under channels_last the
        slot of a 4-D weight is stored as (O, KH, KW, I): the addition goes through the permuted view the parameter's `.grad` is."""
        off, n = self.slices[name]
        flat = self.grad_flat[off:off + n]
        if self.channels_last and g.dim() == 4:
            O, I, KH, KW = g.shape
            flat.view(O, KH, KW, I).permute(0, 3, 1, 2).add_(g.to(flat.dtype))
        else:
            flat.add_(g.reshape(-1).to(flat.dtype))

    def gather_grads(self, seen=None):
        """Copy the gradients autograd left in `.grad` into the flat buffer with multi-tensor copies (segments of
        parameters that received none stay zero), then drop them.  `seen` (a set) collects the names that had one."""
        dst, src = [], []
        for p, v in zip(self.train_params, self.grad_views):
            if p.grad is not None:
                dst.append(v)
                src.append(p.grad)
                if seen is not None:
                    seen.add(self.name_of[id(p)])
        if dst:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.train_params, self.grad_views):
            p.grad = v

    def check_views(self):
        """Autograd must have accumulated in place: every live .grad still aliases the flat buffer, dead ones hold none."""
        base = self.grad_flat.data_ptr()
        for name, p in self.order:
            if p.requires_grad and name not in self.dead:
                off, n = self.slices[name]
                if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                    return False
            elif p.requires_grad and p.grad is not None:
                return False
        return True


class BucketedGradExchange:
    """Mean of the flat gradient over the data-parallel ranks, OVERLAPPED with backward.

    The live trainable segment of the flat gradient buffer is cut into contiguous buckets on parameter boundaries, never
    across a parameter-group boundary (the bias group holds parameters of every depth of the network: mixed into a weight
    bucket it would hold that bucket back until the first layers' biases arrive); groups below `small` elements are one
    bucket each.  Dead parameters (FlatParams) are in no bucket: a bucket is complete when every one of ITS parameters has
    fired, and with the never-used MIL FC stacks inside, the buckets of the head - the first to be needed - never completed
    and every all-reduce went out after backward (round-2 verdict K1).
    A post-accumulate hook on every parameter counts arrivals; when the last gradient of a bucket
    has been produced the bucket is copied into the flat buffer with one multi-tensor copy and its all-reduce
    (RCCL over xGMI; a few large messages, xGMI being point-to-point) is issued on a side stream while autograd
    keeps computing the earlier layers.  `finish()` flushes buckets whose parameters received no gradient in
    this iteration (their segment stays zero) and joins the side stream.  gloo / CPU: same logic, synchronous.

    Collectives are ISSUED IN A FIXED ORDER (`issue_order`: buckets by the model position of their earliest parameter,
    latest first - the order backward completes them in when every parameter receives a gradient).  A completed bucket
    waits until every bucket ahead of it in that order has been issued, so two ranks on which gradients arrive in
    different orders, or on which a parameter receives no gradient at all (an empty image, a data-dependent branch),
    still enqueue the same sequence of equally sized all-reduces; `finish()` issues whatever is left in the same order
    and asserts the sequence.  `stats` = what the last iteration did (buckets, how many went out during backward, bytes)."""

    def __init__(self, flat, n_buckets=6, device=None, small=1 << 16, wire=None, force=False):
        """wire: 'fp32' (default) or 'bf16' (PT_GRAD_WIRE=bf16; SURVEY 8(e): 120 MB instead of 240 MB per step over xGMI): a bucket is
        rounded to bf16 on this rank, all-reduced in bf16, and widened back into the flat fp32 gradient - every rank receives the
        same reduced values, so the ranks stay bit-identical; the optimizer still steps fp32 master weights with fp32 momentum."""
        self.flat = flat
        self.force = bool(force)          # issue the collectives with ONE rank too (a 1-rank RCCL communicator: exercises the real path)
        self.wire = (wire or os.environ.get('PT_GRAD_WIRE', 'fp32')).lower()
        assert self.wire in ('fp32', 'bf16'), self.wire
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.stream = torch.cuda.Stream(device=device) if (device is not None and device.type == 'cuda') else None
        self.avg = None
        if (self.world > 1 or self.force) and dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl':
            self.avg = dist.ReduceOp.AVG                    # RCCL divides inside the collective: one launch less per bucket
        total = flat.n_train
        target = max(total // max(int(n_buckets), 1), 1)
        self.buckets = []                      # [start, end, [(param, view)]]
        group_of, g = [], 0
        off = 0
        for p in flat.train_params:
            while off >= flat.group_ends[g]:
                g += 1
            group_of.append(g)
            off += (p.numel() + 3) // 4 * 4
        gsize = [e - (flat.group_ends[i - 1] if i else 0) for i, e in enumerate(flat.group_ends)]
        # the buckets that hold the FIRST layers complete last and their all-reduce has nothing left to hide behind: the first
        # two buckets of a group are a quarter / half of the target size, so the exposed tail is small
        cur, start, acc, off, k = [], 0, 0, 0, 0
        for i, (p, v) in enumerate(zip(flat.train_params, flat.grad_views)):
            n = (p.numel() + 3) // 4 * 4
            cur.append((p, v))
            off += n
            acc += n
            last_of_group = i + 1 == len(group_of) or group_of[i + 1] != group_of[i]
            want = target // 4 if k == 0 else (target // 2 if k == 1 else target)
            if last_of_group or (acc >= want and gsize[group_of[i]] > small):
                self.buckets.append([start, off, cur])
                cur, start, acc, k = [], off, 0, (0 if last_of_group else k + 1)
        assert not cur and off == total, (off, total)
        self.bucket_of = {id(p): b for b, (_, _, ps) in enumerate(self.buckets) for p, _ in ps}
        pos = {id(p): i for i, p in enumerate(flat.root.parameters())}
        first = [min(pos[id(p)] for p, _ in ps) for _, _, ps in self.buckets]
        self.issue_order = sorted(range(len(self.buckets)), key=lambda b: -first[b])
        self.left, self.done, self.active = [0] * len(self.buckets), [True] * len(self.buckets), False
        self.ready, self.next, self.issued = [False] * len(self.buckets), 0, []
        self.seen = None
        self.stats = dict(buckets=len(self.buckets), issued_during_backward=0, bytes=(2 if self.wire == 'bf16' else 4) * total, wire=self.wire, overlap_ms=0.0)
        self._t_first = None
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in flat.train_params]

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def begin(self, seen=None):
        self.left = [len(ps) for _, _, ps in self.buckets]
        self.done = [False] * len(self.buckets)
        self.ready = [False] * len(self.buckets)
        self.next, self.issued = 0, []
        self.seen = seen
        self._in_backward = 0
        self._t_first = None
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
            self._in_backward += 1

    def _flush(self, b):
        start, end, ps = self.buckets[b]
        dst, src = [], []
        for p, v in ps:
            if p.grad is not None:
                dst.append(v)
                src.append(p.grad)
                if self.seen is not None:
                    self.seen.add(self.flat.name_of[id(p)])
        if dst:
            with torch.no_grad():
                torch._foreach_copy_(dst, src)
        for p, _ in ps:
            p.grad = None                      # the copy in the flat buffer is the gradient from here on
        g = self.flat.grad_flat[start:end]
        if self._t_first is None:
            self._t_first = time.perf_counter()

        def reduce_(t):
            if self.avg is not None:
                dist.all_reduce(t, op=self.avg)
            else:
                t.div_(self.world)
                dist.all_reduce(t)
        if self.world > 1 or self.force:
            if self.stream is None:
                if self.wire == 'bf16':
                    w16 = g.to(torch.bfloat16)
                    reduce_(w16)
                    g.copy_(w16)
                else:
                    reduce_(g)
            else:
                self.stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.stream):
                    if self.wire == 'bf16':
                        w16 = g.to(torch.bfloat16)          # (allocated and consumed on the side stream)
                        reduce_(w16)
                        g.copy_(w16)
                    else:
                        reduce_(g)
        self.done[b] = True
        self.issued.append(b)

    def finish(self):
        during = self._in_backward
        # host time between the first bucket's issue and the end of backward: the window the collectives had to hide in
        overlap_ms = (time.perf_counter() - self._t_first) * 1e3 if (self._t_first is not None and during) else 0.0
        while self.next < len(self.issue_order):            # buckets still waiting for a gradient that never came, in order
            self._flush(self.issue_order[self.next])
            self.next += 1
        assert self.issued == self.issue_order, (self.issued, self.issue_order)
        self.active = False
        self.stats = dict(buckets=len(self.buckets), issued_during_backward=during, bytes=(2 if self.wire == 'bf16' else 4) * self.flat.n_train,
                          wire=self.wire, overlap_ms=round(overlap_ms, 3))
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
                 grad_chunks=4, autocast_dtype=None, channels_last=False, force_exchange=False):
        assert optimizer_cfg.get('type', 'SGD') == 'SGD', 'the Point-Teacher recipe is SGD'
        self.model = model
        if autocast_dtype is not None and not channels_last:
            # The bf16 backbone is only built for [B,H,W,C] activations (fused bf16 BatchNorm epilogue, MIOpen's NHWC
            # kernels).  MIOpen's NCHW bf16 path was measured 13 % away from a bf16-rounding oracle on the PSAGG
            # features (NHWC: 1.2 %, profiles/r02/bf16_accuracy.txt), so the layout is not left to the caller.
            channels_last = True
        pw = optimizer_cfg.get('paramwise_cfg', {}) or {}
        self.flat = FlatParams(model, channels_last=channels_last, paramwise_cfg=pw)
        self.channels_last = channels_last
        self.momentum = optimizer_cfg.get('momentum', 0.0)
        self.weight_decay = optimizer_cfg.get('weight_decay', 0.0)
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
        self.n_buckets = max(grad_chunks, 1) + 2
        # force_exchange: run the bucketed exchange with a single rank as well (a 1-rank process group: the RCCL code path - AVG
        # all-reduce on the side stream, bf16 wire - executes on a one-GPU box; tests/test_rccl_single_rank.py)
        self.force_exchange = bool(force_exchange and dist.is_available() and dist.is_initialized())
        self.exchange = (BucketedGradExchange(self.flat, self.n_buckets, dev, force=self.force_exchange)
                         if (self.world > 1 or self.force_exchange) else None)
        # Parameters that never receive a gradient (FlatParams "dead"): discovered at the first step, agreed between the ranks
        # with one all-reduce of a bitmap, re-checked every `revive_interval` steps (N > 1; N == 1 sees a revival at once).
        self.dead_known = False
        self.revive_interval = 100
        self.census_interval = 10
        self._revived = set()
        self._revived_grads = {}              # name -> gradient(s) a dead parameter received since the last agreement point
        # BASELINE configs[2] "bf16 backbone + fp32 head": autocast covers backbone / FPN / PSAGG only (Student_FCOS.extract_feat)
        self.autocast_dtype = autocast_dtype
        for m in model.modules():
            if hasattr(m, 'backbone') and hasattr(m, 'extract_feat'):
                m.backbone_autocast = autocast_dtype
        self.tuned_gemms = enable_tuned_gemms()
        self._broadcast_initial_state()

    def _broadcast_initial_state(self):
        if self.world > 1 or self.force_exchange:   # identical initial weights on every rank, as DDP does
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
        seen = set() if not self.dead_known else None
        if self.exchange is not None:
            self.exchange.begin(seen)
            out['loss'].backward()
            self.exchange.finish()
        else:
            out['loss'].backward()
            self.flat.gather_grads(seen)
        self._settle_dead(seen)
        f = self.flat
        sq = F.grad_sqnorm(f.grad_flat) if self.max_norm > 0 else None
        F.sgd_step_groups_(f.student_flat[:f.n_train], f.grad_flat, f.mom_flat, f.group_tables(), self.lr_t, self.momentum,
                           self.weight_decay, sq, self.max_norm, self.iter == 0)
        self.iter += 1
        if self.iter % self.census_interval == 0:
            # range census of the fp16 planes (planes.CENSUS): an asynchronous copy now, digested at the next poll - no host wait; a
            # site that saturated or sank below fp16's comfortable range demotes its group to bf16 x 3 planes from the next step on
            from . import planes as PL
            PL.CENSUS.poll()
        return out

    # ------------------------------------------------------------------------------ never-used parameters --
    def _agree(self, names, universe):
        """Union over the ranks of a set of parameter names (one all-reduce of a bitmap + one read: only at the first step
        and every `revive_interval` steps)."""
        if self.world == 1:
            return set(names)
        bits = torch.zeros(len(universe), dtype=torch.int32)
        for i, n in enumerate(universe):
            if n in names:
                bits[i] = 1
        bits = bits.to(self.flat.student_flat.device)
        dist.all_reduce(bits, op=dist.ReduceOp.MAX)
        bits = bits.cpu()
        return {n for i, n in enumerate(universe) if int(bits[i])}

    def _settle_dead(self, seen):
        """torch.optim.SGD skips parameters whose `.grad is None` and DDP reduces nothing for them: the parameters no backward
        has ever reached (TS_P2BFCOSHead.shared_fcs / shared_fcs_refine / fc_iou: 31 % of the student) leave the live groups.
        First step: dead = trainable parameters without a gradient on ANY rank.  Later: a dead parameter that received a
        gradient is revived (N == 1: in the same step, before the update, WITH that gradient; N > 1: at the next agreement point,
        so that every rank re-lays its buffers in the same step - the latest gradient seen in between enters the update of the
        revival step, averaged over the ranks by one all-reduce)."""
        f = self.flat
        trainable = [n for n, p in f.order if p.requires_grad]
        if not self.dead_known:
            live = self._agree(seen | set(f.take_revived()), trainable)       # (first step: every trainable parameter is still live)
            self.dead_known = True
            self._relayout(set(trainable) - live)
            return
        if self.world == 1:
            self._revived.update(f.take_revived(self._revived_grads))
            if self._revived:
                self._relayout(set(f.dead) - self._revived)
                for n, g in self._revived_grads.items():           # the gradient that revived it takes part in THIS update
                    self.flat.add_to_grad(n, g)
                self._revived, self._revived_grads = set(), {}
            return
        # N > 1: only the LATEST gradient of a revived parameter is kept (round-4 advice: the sum of up to `revive_interval` steps'
        # gradients in one update would dominate that step's clip norm and shrink every other parameter's update)
        fresh = {}
        self._revived.update(f.take_revived(fresh))
        self._revived_grads.update(fresh)
        if (self.iter + 1) % self.revive_interval == 0 and f.dead:
            rev = self._agree(self._revived, sorted(f.dead))
            if rev:
                self._relayout(set(f.dead) - rev)
                # the latest gradient each rank saw since the last agreement point, averaged over the ranks by ONE all-reduce of
                # their concatenation (a rank that saw none contributes zeros), enters this update instead of being dropped
                names = sorted(rev)
                shapes = {n: dict(f.order)[n].shape for n in names}
                dev = self.flat.grad_flat.device
                parts = [(self._revived_grads[n].reshape(-1).float() if n in self._revived_grads
                          else torch.zeros(self.flat.slices[n][1], device=dev)) for n in names]
                buf = torch.cat(parts).contiguous()
                dist.all_reduce(buf)
                buf /= self.world
                o = 0
                for n in names:
                    cnt = self.flat.slices[n][1]
                    self.flat.add_to_grad(n, buf[o:o + cnt].view(shapes[n]))
                    o += cnt
            self._revived, self._revived_grads = set(), {}

    def _relayout(self, dead):
        if self.flat.relayout(dead) and self.exchange is not None:
            self.exchange.remove()
            self.exchange = BucketedGradExchange(self.flat, self.n_buckets, self.flat.student_flat.device, force=self.force_exchange)

    def state_dict(self):
        return dict(model=self.model.state_dict(), momentum=self.flat.mom_flat.clone(), iter=self.iter,
                    dead=sorted(self.flat.dead), dead_known=self.dead_known)

    def load_state_dict(self, sd):
        self.model.load_state_dict(sd['model'])
        if hasattr(self.model, '_stem_shared'):
            self.model._stem_shared = None           # re-decided from the loaded state
        # the flat momentum is stored in the layout of the run that wrote it: adopt its dead set first
        self._relayout(set(sd.get('dead', ())))
        self.dead_known = bool(sd.get('dead_known', False))
        self.flat.mom_flat.copy_(sd['momentum'])
        self.iter = sd['iter']
