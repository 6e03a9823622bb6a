#!/usr/bin/env python3
"""Benchmark of the Point-Teacher training iteration on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload step1|step2] [--dtype fp32|bf16]

One "step" = one full `train_step` (teacher EMA + teacher pass + student passes + losses +
backward + grad-clip + SGD) on B = 2 synthetic 800x800 tiles with ~300 point annotations per
image, per GPU.  Default workload = BASELINE.json configs[1]: burn-in phase 1, fp32, bs 2.
For N > 1 the driver launches this file under torch.distributed.run (one rank per GPU, RCCL).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='step1', choices=['step1', 'step2'])
    ap.add_argument('--variant', default='hbb', choices=['hbb', 'obb'],
                    help="hbb: AI-TOD-v2 configs (BASELINE configs[0-3]); obb: SODA-A oriented config (configs[4])")
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'])
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--size', type=int, default=None, help='tile side; default 800 (hbb) / 1200 (obb, RResize of the config)')
    ap.add_argument('--objects', type=int, default=300)
    ap.add_argument('--percent', type=int, default=0, choices=[0, 30, 60, 100])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--channels-last', type=int, default=1)
    ap.add_argument('--miopen-find', type=int, default=1)
    ap.add_argument('--fold-bn', type=int, default=0)
    ap.add_argument('--cpu-baseline-iters', type=int, default=1)
    ap.add_argument('--roofline-kernel', default='auto')
    return ap.parse_args()


# algorithmic HBM bytes per launch of the custom kernels
def algorithmic_bytes(name, shapes):
    if name in ('pt_roi_align_fwd', 'pt_roi_align_bwd', 'pt_roi_align_rotated_fwd', 'pt_roi_align_rotated_bwd'):
        K, C, o = shapes['K'], shapes['C'], shapes['out']
        return K * C * o * o * 4            # the [K,C,7,7] block written (fwd) / read (bwd); the map stays in L2/MALL
    if name in ('pt_affine_relu_fwd', 'pt_affine_relu_bwd', 'pt_affine_relu_bwd_train'):
        return shapes['n'] * 4 * shapes['streams']   # fp32 streams read + written per element (x, y, residual / g, y, gx, gres)
    if name == 'pt_ema_update':
        return shapes['n'] * 12             # read teacher+student, write teacher
    if name == 'pt_sgd_step':
        return shapes['n'] * 20             # read p,g,m; write p,m
    if name == 'pt_sqnorm_partial':
        return shapes['n'] * 4
    return None


def main():
    args = parse()
    obb = args.variant == 'obb'
    if args.size is None:
        args.size = 1200 if obb else 800
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU path)'
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)          # (rehearsals put several gloo ranks on one card)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        backend = os.environ.get('PT_DIST_BACKEND', 'nccl')     # 'nccl' is RCCL on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    import point_teacher_amd as pta
    from point_teacher_amd import hip
    from point_teacher_amd.synthetic import SyntheticTiles

    torch.backends.cudnn.benchmark = bool(args.miopen_find)      # MIOpen find mode
    torch.manual_seed(1234)           # same initial weights on every rank (also broadcast by the Trainer)
    if obb:
        cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'obb', 'point_teacher',
                                               'sodaa_fcos_pointteacher_1x.py'))
    else:
        cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher',
                                               f'aitodv2_point_teacher_{args.percent}.py'))
    # phase switch (fcos_p2b_teacher_student.py:133): count <= burn_in_step -> step 1
    cfg.model['burn_in_step'] = 10 ** 9 if args.workload == 'step1' else -1
    model = pta.build_detector(cfg.model).to(dev)
    from point_teacher_amd.synthetic import benchmark_init_
    benchmark_init_(model, phase2=(args.workload == 'step2'))     # see its docstring: same shapes, same work
    model.train()
    trainer = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, iters_per_epoch=5000,
                          autocast_dtype=torch.bfloat16 if args.dtype == 'bf16' else None,
                          channels_last=bool(args.channels_last))
    if args.fold_bn:
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.fold_into_conv = True
    data = SyntheticTiles(n=8, size=args.size, mean_objects=args.objects, seed=7, device=dev, rank=rank, world=world,
                          oriented=obb, num_classes=9 if obb else 8)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(args.warmup):
        trainer.step(data.batch(it, args.batch))
    barrier()

    # per-kernel HIP-event timing of the custom kernels inside the timed region (torch's current
    # stream is the stream every libpt_hip kernel is launched on)
    prof = {}
    orig_call = hip.call

    def timed_call(fn, *a):
        if fn.startswith('pt_') and fn not in ('pt_focal_nblocks', 'pt_sqnorm_nblocks', 'pt_abi_version', 'pt_affine_train_rows'):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig_call(fn, *a)
            e1.record()
            shp = None
            if fn in ('pt_roi_align_fwd', 'pt_roi_align_bwd', 'pt_roi_align_rotated_fwd', 'pt_roi_align_rotated_bwd'):
                shp = dict(K=a[6], C=a[3], out=a[7])
            elif fn == 'pt_affine_relu_fwd':
                shp = dict(n=a[4], streams=2 + (a[3] is not None))
            elif fn == 'pt_affine_relu_bwd':
                shp = dict(n=a[3], streams=1 + (a[1] is not None) + (a[7] is not None) + (a[8] is not None))
            elif fn == 'pt_affine_relu_bwd_train':
                shp = dict(n=a[4], streams=2 + (a[1] is not None) + (a[7] is not None) + (a[8] is not None))
            elif fn in ('pt_ema_update', 'pt_sgd_step', 'pt_sqnorm_partial'):
                shp = dict(n=a[2] if fn == 'pt_ema_update' else (a[3] if fn == 'pt_sgd_step' else a[1]))
            prof.setdefault(fn, []).append((e0, e1, shp))
            return r
        return orig_call(fn, *a)
    hip.call = timed_call
    import point_teacher_amd.functional as PF
    PF.hip.call = timed_call
    # what an (event, event) pair measures with NOTHING in between: the per-launch bias of the HIP-event timings below
    # (rocprofv3 reports the bare kernel time, which therefore lies between `avg_launch_us - overhead` and `avg_launch_us`)
    cal = []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        cal.append((e0, e1))
    torch.cuda.synchronize()
    cal = sorted(a.elapsed_time(b) for a, b in cal)
    event_overhead_us = cal[len(cal) // 2] * 1e3

    barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        out = trainer.step(data.batch(args.warmup + it, args.batch))
    barrier()
    dt = time.perf_counter() - t0
    hip.call = orig_call
    PF.hip.call = orig_call
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    log_vars = out['log_vars'].materialize()

    # ---- roofline of the dominant custom kernel -------------------------------------
    kern = {}
    for fn, evs in prof.items():
        ms = [e0.elapsed_time(e1) for e0, e1, _ in evs]
        byts = [algorithmic_bytes(fn, s) if s else None for _, _, s in evs]
        kern[fn] = dict(calls=len(ms), total_ms=sum(ms), avg_ms=sum(ms) / len(ms),
                        bytes=(sum(b for b in byts if b) / len(byts)) if all(b for b in byts) else None)
    timed = {k: v for k, v in kern.items() if v['bytes']}
    dom = args.roofline_kernel if args.roofline_kernel != 'auto' else max(timed, key=lambda k: timed[k]['total_ms'])
    d = timed[dom]
    achieved = d['bytes'] / (d['avg_ms'] * 1e-3) / 1e9
    traffic = None          # HBM bytes per launch from the committed PMC passes (same workload), else null
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01', 'pmc_traffic.json')))
        traffic = pmc.get(('obb_' if obb else '') + args.workload, {}).get(dom, {}).get('traffic')
    except Exception:
        pass
    roofline = dict(bound='hbm', kernel=dom, achieved=round(achieved, 1), peak=8000.0, unit='GB/s',
                    frac=round(achieved / 8000.0, 4), traffic=traffic, avg_launch_us=round(d['avg_ms'] * 1e3, 2),
                    event_pair_overhead_us=round(event_overhead_us, 2),
                    launches=d['calls'], bytes_per_launch=int(d['bytes']))

    if rank == 0:
        cpu_baseline = None
        if obb:
            cpu_baseline = dict(value=None, unit='iters/s', cores=None, kind='port',
                                sample='not timed for the oriented variant (its oracle, oracle/ref_obb.py, loops over RoIs '
                                       'in python); the CPU leg is reported on the default hbb workload')
        elif not args.no_cpu_baseline and world == 1:
            try:
                from oracle import ref_model
                cpu_baseline = ref_model.cpu_baseline(args.workload, args.batch, args.size, args.objects,
                                                      iters=args.cpu_baseline_iters)
            except Exception as e:  # the baseline is a reported side number; never fail the bench for it
                cpu_baseline = dict(value=None, unit='iters/s', cores=os.cpu_count(), kind='port',
                                    sample=f'failed: {type(e).__name__}: {e}')
        iters_s = args.steps * 1.0 / dt
        # BASELINE.md section 3 (0 % config, B = 2, 800x800); the conv towers dominate, so other tile sizes scale by area
        flops_iter = 3.3e12 * (args.batch / 2) * (args.size / 800.0) ** 2
        peak = 2.5e15 if args.dtype == 'bf16' else 157.3e12
        line = dict(
            metric=f'train iters/sec ({args.size}x{args.size}, ~{args.objects} pts/img)', value=round(iters_s * world, 4), unit='iters/s',
            n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
            higher_is_better=True, scaling='weak', vs_baseline=None,
            dtype='f32' if args.dtype == 'fp32' else 'bf16', data='synthetic',
            config=dict(workload=(f'sodaa_fcos_pointteacher_1x (oriented) ' if obb else f'aitodv2_point_teacher_{args.percent}% ')
                                 + f'{"burn-in phase 1" if args.workload == "step1" else "phase 2 (MIL on)"}, '
                                 f'R50-FPN-PSAGG + {"TS_P2RBRotatedFCOSHead" if obb else "TS_P2BFCOSHead"}, bs {args.batch}/GPU, {args.size}x{args.size}, '
                                 f'~{args.objects} pts/img, {"fp32" if args.dtype == "fp32" else "bf16 autocast convs + fp32 head"}',
                        global_batch=args.batch * world, parallelism=f'dp{world}', phase=args.workload),
            roofline=roofline, cpu_baseline=cpu_baseline,
            iteration=dict(flops=flops_iter, achieved_tflops=round(flops_iter * iters_s / 1e12, 2),
                           mfma_peak_tflops=peak / 1e12, frac=round(flops_iter * iters_s / peak, 4)),
            custom_kernels_ms_per_step={k: round(v['total_ms'] / args.steps, 3) for k, v in sorted(kern.items())},
            loss=round(log_vars.get('loss', float('nan')), 4))
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
