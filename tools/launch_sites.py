"""Where do the launch-bound torch kernels of an iteration come from?  torch.profiler with python stacks: every device
kernel that is neither ours (pt::), nor a convolution / GEMM is charged to the innermost frame inside point_teacher_amd/.
usage: python tools/launch_sites.py [step1|step2] [bf16]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import point_teacher_amd as pta  # noqa: E402
from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

phase2 = 'step2' in sys.argv
bf16 = 'bf16' in sys.argv
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', 'aitodv2_point_teacher_0.py'))
cfg.model['burn_in_step'] = -1 if phase2 else 10 ** 9
torch.manual_seed(0)
model = pta.build_detector(cfg.model).to(dev)
benchmark_init_(model, phase2=phase2)
model.train()
tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True,
                 autocast_dtype=torch.bfloat16 if bf16 else None)
data = SyntheticTiles(n=8, size=800, mean_objects=300, seed=7, device=dev)
for it in range(6):
    tr.step(data.batch(it, 2))
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

VIEW = ('view', 'reshape', 'permute', 'expand', 'slice', 'select', 'unsqueeze', 'squeeze', 'transpose', 't.', 'detach', 'alias',
        'as_strided', 'split', 'unbind', 'empty', 'size', 'stride', 'is_', 'unflatten', '_unsafe_view', 'lift_fresh', 'sym_',
        'narrow', 'chunk', 'view_as', 'result_type', 'can_cast', '_local_scalar_dense', 'record_stream', 'numel', 'dim', 'set_')
sites, site_ops = collections.Counter(), collections.defaultdict(collections.Counter)


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, '__name__') else str(func)
        full = str(func)
        if not any(v in full.split('.')[-2] if full.count('.') >= 2 else v in full for v in VIEW):
            frame = None
            for fs in reversed(traceback.extract_stack(limit=40)):
                if 'point_teacher_amd' in fs.filename and not fs.filename.endswith(('hip.py', 'launch_sites.py')):
                    frame = f'{os.path.relpath(fs.filename, ROOT)}:{fs.lineno} {fs.name}'
                    break
            frame = frame or 'autograd engine / other'
            sites[frame] += 1
            site_ops[frame][full.replace('aten.', '')] += 1
        return func(*args, **(kwargs or {}))


with Sites():
    tr.step(data.batch(0, 2))
torch.cuda.synchronize()
print(f'aten ops that launch kernels, per call site, ONE iteration ({"step2" if phase2 else "step1"}): total {sum(sites.values())}')
for frame, n in sites.most_common(70):
    ops = ', '.join(f'{k}x{v}' for k, v in site_ops[frame].most_common(5))
    print(f'{n:5d}  {frame[:100]}  [{ops}]')

# ---- the launch-bound torch kernels by (aten op, input shapes): which of them are fat (feature-map sized)? ----
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for it in range(N):
        tr.step(data.batch(it, 2))
    torch.cuda.synchronize()
HEAVY = ('igemm', 'Cijk', 'ck::', '_ZN2ck', 'pt::', 'MIOpen', 'miopen', 'SubTensorOp', 'naive_conv', 'batched_transpose')
agg = collections.defaultdict(lambda: [0.0, 0])
tot_ms, tot_n = 0.0, 0
for e in prof.events():
    kernels = [k for k in getattr(e, 'kernels', []) if not any(h in k.name for h in HEAVY)]
    if not kernels:
        continue
    shapes = str([s for s in (e.input_shapes or []) if s])[:70]
    ms = sum(k.duration for k in kernels) / 1e3
    a = agg[(e.name, shapes)]
    a[0] += ms; a[1] += len(kernels)
    tot_ms += ms; tot_n += len(kernels)
print(f'\nlaunch-bound torch kernels: {tot_ms / N:.2f} ms / iteration in {tot_n / N:.0f} launches / iteration; by (op, shapes):')
for (name, shapes), (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f'{ms / N:8.3f} ms {n / N:7.1f} x  {name:32s} {shapes}')

# ---- GPU time per call site: HIP events around every dispatched op (one more iteration; the events add host time, not GPU time) ----
timed = collections.defaultdict(list)


class Timed(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        full = str(func)
        if any(v in full.split('.')[-2] if full.count('.') >= 2 else v in full for v in VIEW):
            return func(*args, **(kwargs or {}))
        frame = 'autograd engine / other'
        for fs in reversed(traceback.extract_stack(limit=40)):
            if 'point_teacher_amd' in fs.filename and not fs.filename.endswith(('hip.py', 'launch_sites.py')):
                frame = f'{os.path.relpath(fs.filename, ROOT)}:{fs.lineno} {fs.name}'
                break
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = func(*args, **(kwargs or {}))
        b.record()
        timed[frame].append((a, b, full.replace('aten.', '')))
        return out


with Timed():
    tr.step(data.batch(1, 2))
torch.cuda.synchronize()
rows = []
for frame, evs in timed.items():
    ops = collections.Counter(o for _, _, o in evs)
    rows.append((sum(a.elapsed_time(b) for a, b, _ in evs), len(evs), frame, ops))
print('\nGPU ms per call site (HIP events around each aten op; includes our own kernels when the site calls them through torch ops):')
for ms, n, frame, ops in sorted(rows, key=lambda r: -r[0])[:70]:
    print(f'{ms:8.3f} ms {n:5d} x  {frame[:90]:90s} [' + ', '.join(f'{k}x{v}' for k, v in ops.most_common(4)) + ']')
