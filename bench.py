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
    ap.add_argument('--cpu-baseline-iters', type=int, default=3, help='timed oracle iterations (median is reported)')
    ap.add_argument('--no-phase2', action='store_true', help='skip the extra steady-state (phase 2) measurement')
    ap.add_argument('--no-configs2', action='store_true', help='skip the extra BASELINE configs[2] (bf16 backbone) measurement')
    ap.add_argument('--roofline-kernel', default='auto')
    ap.add_argument('--tiles', type=int, default=64, help='synthetic tiles per rank, cycled (SURVEY 8(d): 64, so the point dictionaries are exercised)')
    ap.add_argument('--no-strict', action='store_true', help='skip the extra strict-fp32 (bf16 x 3 operands everywhere) measurement')
    return ap.parse_args()


# algorithmic HBM bytes per launch of the custom kernels
def algorithmic_bytes(name, shapes):
    if name == 'pt_roi_align_fwd_planes':   # the same block written as three bf16 planes (6 B per element instead of 4)
        return shapes['K'] * shapes['C'] * 49 * 6 + shapes.get('footprint_px', 9 * shapes['K']) * shapes['C'] * 4
    if name == 'pt_roi_align_fwd_planes_f16':   # ... as two fp16 planes: 4 B per element, the fp32 block's size
        return shapes['K'] * shapes['C'] * 49 * 4 + shapes.get('footprint_px', 9 * shapes['K']) * shapes['C'] * 4
    if name in ('pt_roi_align_fwd', 'pt_roi_align_bwd', 'pt_roi_align_rotated_fwd', 'pt_roi_align_rotated_bwd'):
        K, C, o = shapes['K'], shapes['C'], shapes['out']
        # "K*256*49*4 B written (fwd) / read (bwd) + <= 9 feature pixels x 1 KB read per RoI": `footprint_px` is
        # sum_k min(footprint of RoI k, 9) - the members of a bag re-read the same L2-resident rows, charging every member's
        # whole footprint (round 2) put the line's bytes ABOVE the counter traffic (verdict K4)
        return K * C * o * o * 4 + shapes.get('footprint_px', 9 * K) * C * 4
    if name in ('pt_affine_relu_fwd', 'pt_affine_relu_bwd', 'pt_affine_relu_bwd_train'):
        return shapes['n'] * 4 * shapes['streams']   # fp32 streams read + written per element (x, y, residual / g, y, gx, gres)
    if name == 'pt_ema_update':
        return shapes['n'] * 12             # read teacher+student, write teacher
    if name in ('pt_sgd_step', 'pt_sgd_step_groups'):
        return shapes['n'] * 20             # read p,g,m; write p,m
    if name == 'pt_sqnorm_partial':
        return shapes['n'] * 4
    return None


def iteration_flops(workload, cfg_model, batch, size, objects, executed=True):
    """FLOPs of one iteration (1 MAC = 2 FLOP), SURVEY Appendix B / BASELINE.md section 3, counted per pass:
    forward = the layer's MACs, backward = dgrad + wgrad = 2x forward for every layer a gradient reaches.  The frozen stem
    (conv1 + layer1, 10.2 of the trunk's 60.2 GMAC / image) has no backward.  `executed=True` counts what THIS implementation
    runs: the teacher continues from the student's stem (no teacher stem), and in burn-in step 1 each student head pass
    evaluates one of the two towers (forward 0.5 + backward 1.0 head units).  `executed=False` counts what the reference runs
    (teacher stem; both towers forward, the read one backward: 2.0 head units per student head pass of step 1)."""
    a = (size / 800.0) ** 2
    trunk, stem, head = 60.2e9 * a, 10.2e9 * a, 47.5e9 * a     # GMAC per image: R50 + FPN + PSAGG; its frozen stem; dense head
    n = cfg_model['num_training_burninstep2' if workload == 'step2' else 'num_training_burninstep1']
    tc = cfg_model['train_cfg']
    U = 1
    for c in (tc['fine_proposal_cfg'][0], tc['fine_proposal_extensive_cfg'][0]):
        U *= len(c['base_ratios']) ** 2 * (1 + 4 * len(c['shake_ratio'] or []))
    K = batch * min(objects, n) * U
    neg = batch * tc['fine_proposal_cfg'][0]['gen_num_neg']
    fc = 12544 * 1024 + 1024 * 1024 + 1024 * 16                # MAC per RoI of one FC stack + its output layers
    student_trunk = trunk + 2 * (trunk - stem)                 # forward + backward of the trainable part
    teacher = (trunk - stem if executed else trunk) + head
    if workload == 'step2':
        conv = batch * (teacher + student_trunk + (student_trunk + 3 * head))       # teacher | clean (MIL only) | augmented
        mil = (3 * K + 3 * K + 3 * neg) * fc                   # regression and classifier branches, negatives; fwd + bwd
    else:
        half = (1.5 if executed else 2.0) * head               # one tower trains per pass; the other is (reference) forward only
        conv = batch * (teacher + (student_trunk + half) + student_trunk + (student_trunk + half))   # teacher | synthetic | clean | augmented
        mil = (3 * K + K + 3 * K + 3 * neg) * fc               # synthetic regression branch trains, the real one is forward only
    return 2.0 * (conv + mil)


def algorithmic_flops(name, shapes):
    """fp32-equivalent FLOPs of the matrix kernels (2 M N K); the kernels execute 6 bf16 MFMA products per fp32 product."""
    if name == 'pt_gemm_bf16x6_nt':
        return 2.0 * shapes['M'] * shapes['N'] * shapes['K']
    if name in ('pt_conv3x3_bf16x6_nhwc', 'pt_conv3x3_wgrad_bf16x6_nhwc'):
        return 2.0 * shapes['P'] * shapes['Cout'] * 9 * shapes['Cin']
    if name in ('pt_conv_bf16x6', 'pt_conv_wgrad_bf16x6'):         # M output pixels x Cout x (taps * Cin)
        return 2.0 * shapes['M'] * shapes['Cout'] * shapes['taps'] * shapes['Cin']
    return None


def conv_desc_shapes(d):
    """Shapes out of a pt_conv_desc / pt_conv_wgrad_desc (host struct); np: planes per operand (3: fp32 as six bf16 products,
    1: bf16 operands, one product - the bf16 trunk of BASELINE configs[2]); f16: fp32 as two fp16 terms, three products.
    abytes: the launch's ALGORITHMIC HBM bytes - every tensor it reads or writes once, as fp32 (4 B per element: activations at the
    pixels the convolution reads, weights, results, identity, the ReLU mask's activation); fbytes: the same tensors in the formats
    this implementation moves (6 B bf16 x 3 planes, 4 B fp16 x 2 planes or fp32, 2 B one plane / a mask's plane 0, plus the fp32
    partial tiles of split launches).  SURVEY 8(d): algorithmic bytes per launch x launches = the family's floor."""
    Ho = (d.Hs + 2 * d.pad - d.KH) // d.stride + 1
    Wo = (d.Ws + 2 * d.pad - d.KW) // d.stride + 1
    np_ = getattr(d, 'np', 3) or 3
    f16 = int(getattr(d, 'operand_f16', 0) or 0)
    ob = 4 if f16 else 2 * np_                                # bytes per operand element
    taps = d.KH * d.KW
    M = d.B * Ho * Wo
    wn = d.Cout * taps * d.Cin
    if hasattr(d, 'gy_planes'):                               # weight gradient: gy [M, Cout], x at the pixels read, dw (+ partial tiles)
        xe = (M if taps == 1 else d.B * d.Hs * d.Ws) * d.Cin
        S = max(int(d.splits), 1)
        ab = 4 * (M * d.Cout + xe + wn)
        fb = ob * (M * d.Cout + xe) + 4 * wn * (1 + 2 * S)   # (partials written and read once by the reduction)
    else:
        if getattr(d, 'dstride', 0) > 1:                      # transposed form: result on the [out_H, out_W] grid
            M = d.B * d.out_H * d.out_W
        xe = (M if taps == 1 else d.B * d.Hs * d.Ws) * d.Cin
        oe = M * d.Cout
        ab = 4 * (xe + wn)
        fb = ob * (xe + wn)
        if d.out_planes:
            ab += 4 * oe
            fb += (4 if d.out_f16 else 2 * np_) * oe
        if d.out_f32:
            ab += 0 if d.out_planes else 4 * oe               # (a second copy of the same result is format overhead)
            fb += 4 * oe
        if d.res_planes:
            ab += 4 * oe
            fb += (4 if d.res_f16 else 2 * np_) * oe
        if d.res_f32:
            ab += 4 * oe
            fb += 4 * oe
        if d.mask_planes:
            ab += 4 * oe
            fb += 2 * oe
        if int(d.splits) > 1:
            fb += 8 * int(d.splits) * oe
    return dict(M=d.B * Ho * Wo, Cin=d.Cin, Cout=d.Cout, taps=taps, np=np_, f16=f16, abytes=ab, fbytes=fb)


def executed_flops(name, shapes):
    """bf16 MFMA FLOPs a matrix launch executes: six products per fp32 product (three-plane operands), one for bf16 operands."""
    f = algorithmic_flops(name, shapes)
    if f is None:
        return None
    if shapes.get('f16'):                # fp16 x 2 operands (the MIL head's first FC layer): three fp16 MFMA products
        return f * 3.0
    return f * (1.0 if shapes.get('np', 3) == 1 else 6.0)


# the matrix kernels: every fp32 product of backbone / necks / towers / MIL FC stacks as six bf16 MFMA products (csrc/gemm_split.hip)
MFMA_FAMILY = ('pt_bf16x6 (conv + gemm + wgrad)', ('pt_conv_bf16x6', 'pt_conv_wgrad_bf16x6', 'pt_gemm_bf16x6_nt', 'pt_conv3x3_bf16x6_nhwc',
                                                  'pt_conv3x3_wgrad_bf16x6_nhwc'))
# their companion launches: fp32 -> split planes at the edges of the plane-native region, the once-per-update weight planes, the
# exact addition of gradient planes where an activation has two consumers (round-3 advice: report them beside the family)
SPLIT_FNS = ('pt_split_bf16x3', 'pt_split_bf16x3_rows', 'pt_split_bf16x3_gather', 'pt_conv_weight_planes_batch', 'pt_planes_combine',
             'pt_planes_mix', 'pt_split_gather_h2', 'pt_planes_to_f16')
BF16_PEAK_TFLOPS = 2500.0          # dense bf16 MFMA (MI355X_MICROARCH.md); 6 products per fp32 product -> 416.7 TFLOP/s fp32-equivalent ceiling

FAMILIES = {                                                    # op families for the roofline line
    'pt_roi_align': ('pt_roi_align_fwd', 'pt_roi_align_fwd_planes', 'pt_roi_align_fwd_planes_f16', 'pt_roi_align_bwd'),
    'pt_roi_align_rotated': ('pt_roi_align_rotated_fwd', 'pt_roi_align_rotated_bwd'),
    'pt_affine_relu': ('pt_affine_relu_fwd', 'pt_affine_relu_bwd', 'pt_affine_relu_bwd_train'),
    'pt_optimizer (ema + sqnorm + sgd)': ('pt_ema_update', 'pt_sqnorm_partial', 'pt_sgd_step', 'pt_sgd_step_groups'),
}


def cpu_info():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_function_timings():
    """Per-function CPU times of the oracle at BASELINE size: median of 5, milliseconds."""
    import statistics
    from oracle import ref_ops as R
    g = torch.Generator().manual_seed(0)
    ys, xs = torch.meshgrid(torch.arange(100.), torch.arange(100.), indexing='ij')
    pts = torch.stack((xs.reshape(-1) * 8, ys.reshape(-1) * 8), -1) + 4
    gt = torch.rand(300, 2, generator=g) * 780 + 10
    lab = torch.randint(0, 8, (300,), generator=g)
    cls = torch.randn(10000, 8, generator=g)
    box = torch.cat([pts, torch.rand(10000, 2, generator=g) * 30 + 4], 1)
    b4 = torch.cat([gt[:100] - 6, gt[:100] + 6], 1)
    c = torch.randn(200, 1, 25, 8, generator=g)

    def med(f):
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
        return round(statistics.median(ts), 2)
    return {'a7 TopkAssigner(3,3) P=10000 G=300': med(lambda: R.topk_assign(pts, gt, lab, 3)),
            'a6 FUSETopkAssigner P=10000 G=300': med(lambda: R.fuse_topk_assign(box, pts, cls, gt, lab)),
            'a13 bag generation 100 -> 2500 boxes': med(lambda: R.fine_proposals(b4, [1.0, 1.2, 1.3, 0.8, 0.7], None, 4, (800, 800))),
            'a16 bag loss [200,1,25,8]': med(lambda: R.mil_bag_loss(c, c * 0.5, torch.ones(5000, 1, dtype=torch.bool), lab[:200]))}


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
    data = SyntheticTiles(n=args.tiles, size=args.size, mean_objects=args.objects, seed=7, device=dev, rank=rank, world=world,
                          oriented=obb, num_classes=9 if obb else 8)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    trace = os.environ.get('PT_BENCH_TRACE') == '1'          # diagnostics: per-iteration loss dict on stderr (synchronises)

    def traced(it, out):
        if trace:
            lv = out['log_vars'].materialize()
            pts = [p for d in (model.gt_bboxes_point, model.refined_gt_bboxes_point) for p in d.values()] if hasattr(model, 'gt_bboxes_point') else []
            bad = sum(int((~torch.isfinite(p)).sum()) for p in pts)
            big = max([float(p.abs().max()) for p in pts if p.numel()] + [0.0])
            print(f'[trace] it {it} non-finite points {bad} max |coord| {big:.1f} ' + ' '.join(f'{k}={v:.4g}' for k, v in lv.items()), file=sys.stderr, flush=True)

    def settle(tr_, model_, base):
        """Un-timed extra steps (at most 6) until the teacher pass of this model replays as a HIP graph (graphs.py: the capture needs
        a stable signature - every weight form registered - plus two eager calls; its one-off cost, tens of ms of graph
        instantiation, must not fall into the timed region whatever --warmup says)."""
        for k in range(6):
            g = getattr(model_, '_teacher_graph', None)
            if g is not None and (g.disabled or g.replays > 0):
                break
            if g is None and k >= 1:                   # a model without a graphed teacher pass
                break
            tr_.step(data.batch(base + k, args.batch))

    for it in range(args.warmup):
        traced(it, trainer.step(data.batch(it, args.batch)))
    settle(trainer, model, 500)
    barrier()

    # HIP-event timing of the custom kernels (torch's current stream is the stream every libpt_hip kernel is launched on).
    # Inside the TIMED region only the launches of the dominant op family are bracketed (its `roofline.achieved` is measured live
    # there, as the contract asks); an event pair per launch is not free - ~350 pairs per iteration around EVERY custom call cost
    # the step 2 - 3 ms of stream bubbles (the un-instrumented `phase2` field ran below the instrumented stand-alone phase-2 line
    # by that much) - so the family is chosen in 2 un-timed survey steps and the per-kernel table comes from 3 un-timed steps
    # after the timed region.
    orig_call = hip.call
    import point_teacher_amd.functional as PF
    SKIP = ('pt_focal_nblocks', 'pt_sqnorm_nblocks', 'pt_abi_version', 'pt_affine_train_rows', 'pt_gemm_bf16x6_tile_rows',
            'pt_conv3x3_wgrad_bf16x6_splits', 'pt_group_norm_cl_workspace_bytes', 'pt_split_bf16x3_plane_elems', 'pt_conv_bf16x6_splits', 'pt_conv_bf16x6_plan',
            'pt_conv_wgrad_bf16x6_splits')
    import point_teacher_amd.planes as PPL

    def make_hook(prof, only=None):
        def timed_call(fn, *a):
            if fn.startswith('pt_') and fn not in SKIP and (only is None or fn in only):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = orig_call(fn, *a)
                e1.record()
                shp = None
                if fn in ('pt_roi_align_fwd', 'pt_roi_align_bwd'):
                    shp = dict(K=a[6], C=a[3], out=a[7], rois=a[1], scale=a[8], H=a[4], W=a[5])    # footprints are counted afterwards
                elif fn in ('pt_roi_align_fwd_planes', 'pt_roi_align_fwd_planes_f16'):
                    shp = dict(K=a[6], C=a[3], out=7, rois=a[1], scale=a[7], H=a[4], W=a[5])
                elif fn in ('pt_roi_align_rotated_fwd', 'pt_roi_align_rotated_bwd'):
                    shp = dict(K=a[6], C=a[3], out=a[7])
                elif fn == 'pt_affine_relu_fwd':
                    shp = dict(n=a[4], streams=2 + (a[3] is not None))
                elif fn == 'pt_affine_relu_bwd':
                    shp = dict(n=a[3], streams=1 + (a[1] is not None) + (a[7] is not None) + (a[8] is not None))
                elif fn == 'pt_affine_relu_bwd_train':
                    shp = dict(n=a[4], streams=2 + (a[1] is not None) + (a[7] is not None) + (a[8] is not None))
                elif fn in ('pt_conv_bf16x6', 'pt_conv_wgrad_bf16x6'):   # (the weight gradient includes its fixed-order reduction)
                    shp = conv_desc_shapes(a[0])
                elif fn == 'pt_gemm_bf16x6_nt':
                    shp = dict(M=a[7], N=a[8], K=a[9])
                elif fn == 'pt_conv3x3_bf16x6_nhwc':
                    shp = dict(P=a[8] * a[9] * a[10], Cin=a[11], Cout=a[12])
                elif fn == 'pt_conv3x3_wgrad_bf16x6_nhwc':       # (includes its fixed-order reduction of the pixel chunks)
                    shp = dict(P=a[7] * a[8] * a[9], Cin=a[10], Cout=a[11])
                elif fn in ('pt_ema_update', 'pt_sgd_step', 'pt_sgd_step_groups', 'pt_sqnorm_partial'):
                    shp = dict(n=a[2] if fn == 'pt_ema_update' else (a[3] if fn.startswith('pt_sgd_step') else a[1]))
                prof.setdefault(fn, []).append((e0, e1, shp))
                return r
            return orig_call(fn, *a)
        return timed_call

    live = [model]                      # the detector the hooks below act on (configs[2] builds a second one)
    side_stream = getattr(model, 'teacher_stream', False)

    def set_hook(h):
        hip.call = h
        PF.hip.call = h
        PPL.hip.call = h
        # steps whose launches are bracketed by HIP events run the teacher pass inline: on its side stream (the default) its kernels
        # would be co-scheduled with the bracketed ones and stretch them
        if hasattr(live[0], 'teacher_stream'):
            live[0].teacher_stream = side_stream and h is orig_call
            # ... and eagerly: the launches of a replayed HIP graph (graphs.py) cannot be bracketed one by one
            live[0].teacher_graph_off = h is not orig_call

    def footprint_px(r, scale, H, W):      # feature pixels an aligned RoI samples: the bilinear taps of its first and last sample
        x1, y1, x2, y2 = (r[:, i] * scale - 0.5 for i in (1, 2, 3, 4))
        nx = (x2.floor().clamp(0, W - 2) + 1) - x1.floor().clamp(0, W - 1) + 1
        ny = (y2.floor().clamp(0, H - 2) + 1) - y1.floor().clamp(0, H - 1) + 1
        return (nx.clamp(min=1) * ny.clamp(min=1)).clamp(max=9)

    def summarise(prof):
        """prof -> (per-function dict, per-family dict, the MFMA family or None)"""
        fp_total = []
        for fn, evs in prof.items():
            for _, _, shp in evs:
                if shp and 'rois' in shp:
                    fp_total.append((shp, footprint_px(shp.pop('rois'), shp['scale'], shp['H'], shp['W']).sum()))
        if fp_total:
            vals = torch.stack([v for _, v in fp_total]).cpu().tolist()
            for (shp, _), v in zip(fp_total, vals):
                shp['footprint_px'] = int(v)
        kern = {}
        for fn, evs in prof.items():
            ms = [e0.elapsed_time(e1) for e0, e1, _ in evs]
            byts = [algorithmic_bytes(fn, s) if s else None for _, _, s in evs]
            fl = [algorithmic_flops(fn, s) if s else None for _, _, s in evs]
            ex = [executed_flops(fn, s) if s else None for _, _, s in evs]
            kern[fn] = dict(calls=len(ms), total_ms=sum(ms), bytes=sum(b for b in byts if b) if all(b for b in byts) else None,
                            flops=sum(f for f in fl if f) if all(f for f in fl) else None,
                            exec_flops=sum(f for f in ex if f) if all(f for f in ex) else None,
                            abytes=sum((s or {}).get('abytes', 0) for _, _, s in evs), fbytes=sum((s or {}).get('fbytes', 0) for _, _, s in evs))
        fam = {}
        for name, members in FAMILIES.items():
            ks = [kern[m] for m in members if m in kern and kern[m]['bytes']]
            if ks:
                fam[name] = dict(calls=sum(k['calls'] for k in ks), total_ms=sum(k['total_ms'] for k in ks), bytes=sum(k['bytes'] for k in ks))
        # the matrix kernels of the path (fp32 products as six bf16 MFMA products): MFMA-bound, priced against the dense bf16 peak
        mk = [kern[m] for m in MFMA_FAMILY[1] if m in kern and kern[m].get('flops')]
        mfma = dict(calls=sum(k['calls'] for k in mk), total_ms=sum(k['total_ms'] for k in mk), flops=sum(k['flops'] for k in mk),
                    exec_flops=sum(k['exec_flops'] for k in mk), abytes=sum(k['abytes'] for k in mk),
                    fbytes=sum(k['fbytes'] for k in mk)) if mk else None
        if mfma:
            sk = [kern[m] for m in SPLIT_FNS if m in kern]
            mfma['split_ms'], mfma['split_calls'] = sum(k['total_ms'] for k in sk), sum(k['calls'] for k in sk)
        return kern, fam, mfma

    def dominant(fam, mfma):
        if args.roofline_kernel != 'auto':
            return args.roofline_kernel
        dom = max(fam, key=lambda k: fam[k]['total_ms'])
        return MFMA_FAMILY[0] if (mfma and mfma['total_ms'] > fam[dom]['total_ms']) else dom

    # survey (un-timed): which family dominates
    survey = {}
    set_hook(make_hook(survey))
    for it in range(2):
        trainer.step(data.batch(3000 + it, args.batch))
    torch.cuda.synchronize()
    set_hook(orig_call)
    _, sfam, smfma = summarise(survey)
    dom = dominant(sfam, smfma)
    dom_fns = set(MFMA_FAMILY[1] + SPLIT_FNS if dom == MFMA_FAMILY[0] else FAMILIES[dom])
    del survey

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

    # The family is ~280 launches per iteration now (every convolution of the trunk): an event pair around each of them in every
    # timed step costs the step ~1.8 ms of stream bubbles, so the timed region carries events on every EVENT_EVERY-th step
    # (5 of the default 20) - `achieved` is still measured live inside the timed region, on a sample of its steps.
    EVENT_EVERY = 4
    prof = {}
    family_hook = make_hook(prof, only=dom_fns)
    n_event_steps = 0
    barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        if it % EVENT_EVERY == 0:
            set_hook(family_hook)
            n_event_steps += 1
        out = trainer.step(data.batch(args.warmup + it, args.batch))
        if it % EVENT_EVERY == 0:
            set_hook(orig_call)
        traced(args.warmup + it, out)
    barrier()
    dt = time.perf_counter() - t0
    set_hook(orig_call)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    log_vars = out['log_vars'].materialize()

    # per-kernel table of ALL custom kernels: 3 un-timed steps after the timed region
    BREAKDOWN_STEPS = 3
    INSTR = (f'teacher pass on a second HIP stream except in instrumented steps; '
             f'timed region: HIP events around the launches of the dominant family (and its operand-format passes) on every '
             f'{EVENT_EVERY}th step ({n_event_steps} of {args.steps}); other families and custom_kernels_ms_per_step: '
             f'{BREAKDOWN_STEPS} un-timed steps after it with events around every custom call')
    full = {}
    set_hook(make_hook(full))
    for it in range(BREAKDOWN_STEPS):
        trainer.step(data.batch(4000 + it, args.batch))
    torch.cuda.synchronize()
    set_hook(orig_call)
    kern, fam, _ = summarise(full)

    # ---- roofline of the dominant custom OP FAMILY (forward + backward of one op count together), from the TIMED region ----
    _, tfam, mfma = summarise(prof)
    d = tfam[dom] if dom in tfam else mfma
    # HIP-event pairs include ~5 us of event overhead per launch (event_pair_overhead_us); `achieved` keeps it in
    # (conservative); rocprofv3's bare kernel time is shorter by about that much per launch
    achieved = d['bytes'] / (d['total_ms'] * 1e-3) / 1e9 if 'bytes' in d else None
    # HBM bytes from PMC passes: only when profiles/r02/pmc_traffic.json was taken with THIS kernel source (sha256 of the
    # .hip file) and this workload; otherwise null (never a number measured on other code)
    traffic = None
    try:
        import hashlib
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r05', 'pmc_traffic.json')))
        ent = pmc.get(('obb_' if obb else '') + args.workload + ('_bf16' if args.dtype == 'bf16' else ''), {}).get(dom)
        if ent:
            src = os.path.join(ROOT, 'point_teacher_amd', 'csrc', ent['source'])
            if hashlib.sha256(open(src, 'rb').read()).hexdigest() == ent['source_sha256']:
                traffic = ent['traffic_bytes_per_launch']
    except Exception:
        pass
    # the other families: from the breakdown steps (outside the timed region); the dominant one: from the timed region
    fams = {k: dict(ms_per_step=round(v['total_ms'] / BREAKDOWN_STEPS, 3), achieved_GBps=round(v['bytes'] / (v['total_ms'] * 1e-3) / 1e9, 1))
            for k, v in sorted(fam.items())}
    if dom in tfam:
        fams[dom] = dict(ms_per_step=round(tfam[dom]['total_ms'] / n_event_steps, 3),
                         achieved_GBps=round(tfam[dom]['bytes'] / (tfam[dom]['total_ms'] * 1e-3) / 1e9, 1))
    bmfma = summarise({k: v for k, v in full.items() if k in MFMA_FAMILY[1]})[2] if dom != MFMA_FAMILY[0] else mfma
    msteps = BREAKDOWN_STEPS if dom != MFMA_FAMILY[0] else n_event_steps
    if bmfma:
        ex = bmfma['exec_flops'] / (bmfma['total_ms'] * 1e-3) / 1e12
        fams[MFMA_FAMILY[0]] = dict(ms_per_step=round(bmfma['total_ms'] / msteps, 3), executed_bf16_TFLOPs=round(ex, 1),
                                    fp32_equivalent_TFLOPs=round(bmfma['flops'] / (bmfma['total_ms'] * 1e-3) / 1e12, 1))
    def mfma_roofline(m, steps_, **extra):
        # achieved = EXECUTED bf16 FLOPs (6 MFMA products per fp32 product) / HIP-event time, peak = dense bf16 MFMA (guide).
        # frac_algorithmic prices the ALGORITHMIC work (2 M N K fp32 FLOPs) against the same peak - SURVEY 8(d)'s fraction; the
        # scheme's own ceiling is peak / 6.  split_overhead_ms: the family's companion launches (operand-format passes), timed in the
        # same region; fp32_equivalent_incl_split charges them to the family as well.
        ppp = m['exec_flops'] / m['flops']                      # MFMA products per algorithmic product: 6 (fp32 operands) ... 1 (bf16 operands)
        ex = m['exec_flops'] / (m['total_ms'] * 1e-3) / 1e12
        incl = m['flops'] / ((m['total_ms'] + m.get('split_ms', 0.0)) * 1e-3) / 1e12
        return dict(bound='mfma', kernel=MFMA_FAMILY[0], achieved=round(ex, 1), peak=BF16_PEAK_TFLOPS, unit='TFLOP/s',
                    frac=round(ex / BF16_PEAK_TFLOPS, 4), frac_algorithmic=round(ex / ppp / BF16_PEAK_TFLOPS, 4),
                    products_per_algorithmic_product=round(ppp, 3),
                    scheme_ceiling_tflops=round(BF16_PEAK_TFLOPS / ppp, 1), fp32_equivalent_tflops=round(ex / ppp, 1),
                    fp32_equivalent_incl_split_tflops=round(incl, 1), fp32_mfma_peak_tflops=157.3,
                    # `peak` is the dense MFMA figure at the 2.4 GHz boost clock; under this family's load the shader clock measured
                    # 1.98 GHz (GRBM_GUI_ACTIVE / kernel time of FC1, profiles/r05/h2_pmc_fc1_ring.txt): a sustained peak of ~2 060
                    sustained_clock_ghz=1.98, frac_of_sustained_peak=round(ex / (BF16_PEAK_TFLOPS * 1.98 / 2.4), 4),
                    ms_per_step=round(m['total_ms'] / steps_, 3), split_overhead_ms=round(m.get('split_ms', 0.0) / steps_, 3),
                    split_launches_per_step=round(m.get('split_calls', 0) / steps_, 1),
                    avg_launch_us=round(m['total_ms'] / m['calls'] * 1e3, 2), event_pair_overhead_us=round(event_overhead_us, 2),
                    launches=m['calls'], flops_per_launch=int(m['exec_flops'] / m['calls']),
                    algorithmic_flops_per_launch=int(m['flops'] / m['calls']),
                    # bytes beside `traffic` (verdict r04 W3): the family's fp32-tensor floor and what its plane formats move, per launch
                    # (conv_desc_shapes); format / algorithmic = the byte overhead of the operand formats (1.5 for bf16 x 3, 1.0 for fp16 x 2)
                    algorithmic_bytes=int(m['abytes'] / m['calls']), format_bytes=int(m['fbytes'] / m['calls']),
                    algorithmic_bytes_per_step=int(m['abytes'] / steps_), format_bytes_per_step=int(m['fbytes'] / steps_),
                    hbm_floor_ms_per_step=round(m['abytes'] / steps_ / 8e12 * 1e3, 3), **extra)

    if dom == MFMA_FAMILY[0]:
        roofline = mfma_roofline(mfma, n_event_steps, traffic=traffic, families=fams, instrumentation=INSTR)
    else:
        roofline = dict(bound='hbm', kernel=dom, achieved=round(achieved, 1), peak=8000.0, unit='GB/s',
                        frac=round(achieved / 8000.0, 4), traffic=traffic,
                        avg_launch_us=round(d['total_ms'] / d['calls'] * 1e3, 2), event_pair_overhead_us=round(event_overhead_us, 2),
                        launches=d['calls'], bytes_per_launch=int(d['bytes'] / d['calls']), families=fams, instrumentation=INSTR)

    def fresh(workload, autocast=None):
        """A new detector + Trainer in the state the stand-alone `--workload <workload>` line starts from (same seed, same
        `benchmark_init_`).  Every sub-line below is timed on one of these: until round 4 the `phase2` field flipped the phase switch
        of the model the default workload had just trained for ~30 phase-1 iterations - its regression branch still emits the
        all-zero distances of a cold start, the centerness target becomes 0.01 / 0 and the weights are NaN from the second phase-2
        iteration on (profiles/r05/census_switch.txt): the driver's 25.46 ms against the stand-alone 22.8 ms was a NaN model."""
        torch.manual_seed(1234)
        c_ = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', 'obb', 'point_teacher', 'sodaa_fcos_pointteacher_1x.py') if obb else
                                 os.path.join(ROOT, 'point_teacher_amd', 'configs', 'point_teacher', f'aitodv2_point_teacher_{args.percent}.py'))
        c_.model['burn_in_step'] = 10 ** 9 if workload == 'step1' else -1
        m_ = pta.build_detector(c_.model).to(dev)
        live[0] = m_
        benchmark_init_(m_, phase2=(workload == 'step2'))
        m_.train()
        t_ = pta.Trainer(m_, c_.optimizer, c_.optimizer_config, c_.lr_config, iters_per_epoch=5000, autocast_dtype=autocast,
                         channels_last=bool(args.channels_last))
        return c_, m_, t_

    def final_loss(out_):
        """The sub-line's last loss; a non-finite one is REPORTED (null + a warning on stderr), never raised: the line must still be
        printed - but a timing behind it is of a diverged model (random labels on a random network do diverge for some seeds)."""
        v = out_['log_vars'].materialize().get('loss', float('nan'))
        if not (v == v and abs(v) != float('inf')):
            print('[bench] WARNING: a sub-line ended on a non-finite loss - its timing is of a diverged model', file=sys.stderr, flush=True)
            return None
        return round(v, 4)

    # ---- the steady-state phase as well (94 % of a 12-epoch run is phase 2) ----
    def timed_with_family(tr_, base):
        """args.steps timed steps of `tr_` with NO instrumentation (an event pair per matrix launch costs a host-bound iteration -
        the bf16 configuration - up to 25 % of its step), then 3 un-timed steps with HIP events around the matrix family (and its
        companions) for the roofline of this workload -> (seconds, roofline or None)."""
        barrier()
        t0_ = time.perf_counter()
        for it_ in range(args.steps):
            last_[0] = tr_.step(data.batch(base + it_, args.batch))
        barrier()
        d_ = torch.tensor([time.perf_counter() - t0_], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(d_, op=dist.ReduceOp.MAX)
        p_ = {}
        set_hook(make_hook(p_, only=set(MFMA_FAMILY[1] + SPLIT_FNS)))
        for it_ in range(3):
            tr_.step(data.batch(base + args.steps + it_, args.batch))
        torch.cuda.synchronize()
        set_hook(orig_call)
        m_ = summarise(p_)[2]
        return float(d_.item()), (mfma_roofline(m_, 3, timed_region=False) if m_ else None)

    last_ = [None]
    exchange_stats = (dict(trainer.exchange.stats, dead_bytes=4 * trainer.flat.n_dead, backend=dist.get_backend())
                      if trainer.exchange is not None else dict(buckets=0, issued_during_backward=0, bytes=0, dead_bytes=4 * trainer.flat.n_dead))
    tuned_table = bool(trainer.tuned_gemms)
    from point_teacher_amd import planes as PPLc

    def census_summary():
        PPLc.CENSUS.poll(sync=True)
        h = PPLc.CENSUS.history
        lo = [v[2] for v in h.values() if v[2] < float('inf')]
        return dict(mode=PPLc.CENSUS.mode, demoted=dict(PPLc.CENSUS.demoted), sites=len(h), saturated_elements=sum(v[0] for v in h.values()),
                    largest_stored=max([v[1] for v in h.values()] + [0.0]), smallest_site_maximum=min(lo) if lo else None, polls_every_steps=10)
    census = census_summary()
    phase2 = None
    if args.workload == 'step1' and not args.no_phase2:
        del trainer, model
        torch.cuda.empty_cache()
        PPLc.CENSUS.reset()
        _, model, trainer = fresh('step2')
        for it in range(max(args.warmup, 5)):
            trainer.step(data.batch(1000 + it, args.batch))
        settle(trainer, model, 1500)
        dt2, roof2 = timed_with_family(trainer, 2000)
        f2 = iteration_flops('step2', cfg.to_dict()['model'], args.batch, args.size, args.objects)
        phase2 = dict(workload='phase 2 (MIL on, steady state): a fresh model in the stand-alone `--workload step2` state, same inputs',
                      value=round(args.steps * world / dt2, 4), unit='iters/s',
                      ms_per_step=round(dt2 / args.steps * 1e3, 3), flops=f2, achieved_tflops=round(f2 * args.steps / dt2 / 1e12, 2),
                      roofline=roof2, loss=final_loss(last_[0]), f16_census=census_summary())

    # ---- BASELINE configs[2] (bf16 backbone / FPN / PSAGG + fp32 heads, two-phase = MIL on): timed by THIS run as well, so
    # that the driver's line carries it (round-2 verdict item 6); a fresh model + Trainer under bf16 autocast, phase 2, same inputs
    configs2 = None
    if args.workload == 'step1' and args.dtype == 'fp32' and not obb and not args.no_phase2 and not args.no_configs2:
        del trainer, model
        torch.cuda.empty_cache()
        PPLc.CENSUS.reset()
        cfg2, model, trainer = fresh('step2', autocast=torch.bfloat16)
        for it in range(max(args.warmup, 5)):
            trainer.step(data.batch(3000 + it, args.batch))
        settle(trainer, model, 3500)
        dt3, roof3 = timed_with_family(trainer, 4000)       # (fp32 heads: three products per product; bf16 trunk: one - the same kernels)
        f3 = iteration_flops('step2', cfg2.to_dict()['model'], args.batch, args.size, args.objects)
        configs2 = dict(workload=f'BASELINE configs[2]: aitodv2_point_teacher_{args.percent}% phase 2 (MIL on), bf16 backbone / FPN / PSAGG (autocast) + '
                                 f'fp32 dense head, MIL head and losses, bs {args.batch}/GPU, {args.size}x{args.size}',
                        value=round(args.steps * world / dt3, 4), unit='iters/s', ms_per_step=round(dt3 / args.steps * 1e3, 3),
                        steps=args.steps, dtype='bf16 backbone + f32 head', flops=f3, achieved_tflops=round(f3 * args.steps / dt3 / 1e12, 2),
                        roofline=roof3, loss=final_loss(last_[0]), f16_census=census_summary())

    # ---- the strictly 24-bit number (verdict r04 W1): the SAME default workload with every fp16-operand group switched off - bf16 x 3
    # planes and six MFMA products everywhere (PT_F16_FC=0) - timed by this run too, so the driver's line carries both
    strict = None
    if args.dtype == 'fp32' and not args.no_strict and PF.F16_FC:
        try:
            del trainer, model
        except NameError:
            pass
        torch.cuda.empty_cache()
        saved = {k: getattr(PF, k) for k in ('F16_FC', 'F16_TOWERS', 'F16_NECK', 'F16_NECK3', 'F16_TRUNK', 'F16_BLOCK3')}
        for k in saved:
            setattr(PF, k, False)
        _, model, trainer = fresh(args.workload)
        for it in range(max(args.warmup, 5)):
            trainer.step(data.batch(it, args.batch))
        settle(trainer, model, 5500)
        dt4, roof4 = timed_with_family(trainer, args.warmup)
        strict = dict(workload='the default workload with bf16 x 3 operands and six MFMA products in EVERY layer (PT_F16_FC=0): 24 significant bits per operand',
                      value=round(args.steps * world / dt4, 4), unit='iters/s', ms_per_step=round(dt4 / args.steps * 1e3, 3), steps=args.steps,
                      roofline=roof4, loss=final_loss(last_[0]))
        for k, v in saved.items():
            setattr(PF, k, v)

    if rank == 0:
        cpu_baseline = None
        if obb and not args.no_cpu_baseline and world == 1:
            try:       # the oriented oracle loops over RoIs in python: ONE iteration of a 1/8 sample (same object density), scaled
                from oracle import ref_obb
                cpu_baseline = ref_obb.cpu_baseline(args.workload, args.batch, args.size, args.objects, sample_div=8)
                cpu_baseline['cpu'] = cpu_info()
            except Exception as e:
                cpu_baseline = dict(value=None, unit='iters/s', cores=os.cpu_count(), kind='port', sample=f'failed: {type(e).__name__}: {e}')
        elif not args.no_cpu_baseline and args.cpu_baseline_iters > 0 and world == 1:
            try:
                from oracle import ref_model
                cpu_baseline = ref_model.cpu_baseline(args.workload, args.batch, args.size, args.objects,
                                                      iters=args.cpu_baseline_iters)
                cpu_baseline['cpu'] = cpu_info()
                cpu_baseline['per_function_ms'] = cpu_function_timings()
            except Exception as e:  # the baseline is a reported side number; never fail the bench for it
                cpu_baseline = dict(value=None, unit='iters/s', cores=os.cpu_count(), kind='port',
                                    sample=f'failed: {type(e).__name__}: {e}')
        iters_s = args.steps * 1.0 / dt
        # the oriented variant runs the same passes on the same trunk / towers (GroupNorm and the 1-channel angle convolution
        # add < 0.2 % of the head's MACs), at (1200 / 800)^2 the pixels and with its own bag sizes: counted per pass like HBB
        flops_iter = iteration_flops(args.workload, cfg.to_dict()['model'], args.batch, args.size, args.objects)
        flops_ref = iteration_flops(args.workload, cfg.to_dict()['model'], args.batch, args.size, args.objects, executed=False)
        peak = 2.5e15 if args.dtype == 'bf16' else 157.3e12
        line = dict(
            metric=f'train iters/sec ({args.size}x{args.size}, ~{args.objects} pts/img)', value=round(iters_s * world, 4), unit='iters/s',
            n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
            higher_is_better=True, scaling='weak', vs_baseline=None,
            # arithmetic type of the path; every fp32 product of the trainable trunk, the necks, the towers and the MIL FC stacks is
            # formed from six bf16 MFMA products - three fp16 ones in the MIL head's first FC layer - with fp32 accumulation
            # (csrc/gemm_split.hip: error vs float64 below the fp32
            # library kernels'; activations travel between the layers as exact three-term bf16 splits of their fp32 values)
            # NOT 24-bit operands by default (round-4 advice): the trainable trunk, necks, towers and MIL FC stacks multiply scaled
            # fp16 x 2 operands (22 significant bits per stored value, three MFMA products, fp32 accumulation) unless PT_F16_FC=0 or
            # the range census demoted a group; `strict_fp32` below is the bf16 x 3 (24-bit, six products) number of the same workload
            dtype=('f32' + ((' (fp32 tensors at the boundaries; trainable trunk / necks / towers / MIL FC stacks on split-operand MFMA with fp32 '
                             'accumulation: ' + ('scaled fp16 x 2 planes = 22-bit operands, 3 products' if PF.F16_FC else 'bf16 x 3 planes = 24-bit operands, 6 products')
                             + (f'; groups demoted to bf16 x 3 by the range census: {sorted(census["demoted"])}' if census['demoted'] else '') + ')')
                            if os.environ.get('PT_SPLIT_GEMM', '1') != '0' and os.environ.get('PT_SPLIT_CONV', '1') != '0' else ''))
            if args.dtype == 'fp32' else 'bf16', data='synthetic',
            config=dict(workload=(f'sodaa_fcos_pointteacher_1x (oriented) ' if obb else f'aitodv2_point_teacher_{args.percent}% ')
                                 + f'{"burn-in phase 1" if args.workload == "step1" else "phase 2 (MIL on)"}, '
                                 f'R50-FPN-PSAGG + {"TS_P2RBRotatedFCOSHead" if obb else "TS_P2BFCOSHead"}, bs {args.batch}/GPU, {args.size}x{args.size}, '
                                 f'~{args.objects} pts/img, {"fp32" if args.dtype == "fp32" else "bf16 autocast convs + fp32 head"}',
                        global_batch=args.batch * world, parallelism=f'dp{world}', phase=args.workload,
                        gemm_solution_table=tuned_table),
            roofline=roofline, cpu_baseline=cpu_baseline,
            iteration=dict(flops=flops_iter, flops_reference=flops_ref, achieved_tflops=round(flops_iter * iters_s / 1e12, 2),
                           mfma_peak_tflops=peak / 1e12, frac=round(flops_iter * iters_s / peak, 4)),
            phase2=phase2, configs2_bf16=configs2, strict_fp32=strict, f16_census=census,
            # N > 1: weak scaling - `value` = iterations/s of ONE rank x N, i.e. bs-2-equivalent iterations per second of the job
            # (every rank steps its own batch of 2; an 8-GPU line is 8 batches per step time, not 8x faster iterations)
            value_is='iterations/s of one rank x n_gpus (bs-%d-equivalent iterations/s; global batch %d per step)' % (args.batch, args.batch * world),
            # gradient exchange of the LAST timed step (N > 1): buckets, how many all-reduces were issued while backward was
            # still running, payload per step; the never-used MIL stacks are in neither (runtime.FlatParams "dead")
            exchange=exchange_stats,
            # all custom kernels, from the 3 un-timed steps after the timed region (HIP events around every call)
            custom_kernels_ms_per_step={k: round(v['total_ms'] / BREAKDOWN_STEPS, 3) for k, v in sorted(kern.items())},
            loss=(round(log_vars['loss'], 4) if (log_vars.get('loss') == log_vars.get('loss') and abs(log_vars.get('loss', 0.0)) != float('inf')) else None))
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
