"""Where the matrix family's time goes, per SHAPE: HIP events around every pt_conv_bf16x6 / pt_conv_wgrad_bf16x6 / pt_gemm_bf16x6_nt
launch of steady-state iterations, grouped by (entry point, M, Cin, Cout, taps, epilogue flavour); ms / iteration, launches, fp32-equivalent
TFLOP/s.  usage: python tools/family_breakdown.py [step1|step2] [obb]"""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import point_teacher_amd as pta  # noqa: E402
from point_teacher_amd import hip  # noqa: E402
import point_teacher_amd.functional as PF  # noqa: E402
import point_teacher_amd.planes as PPL  # noqa: E402
from point_teacher_amd.synthetic import SyntheticTiles, benchmark_init_  # noqa: E402

phase2 = 'step2' in sys.argv[1:]
obb = 'obb' in sys.argv[1:]
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
sub = ('obb', 'point_teacher', 'sodaa_fcos_pointteacher_1x.py') if obb else ('point_teacher', 'aitodv2_point_teacher_0.py')
cfg = pta.Config.fromfile(os.path.join(ROOT, 'point_teacher_amd', 'configs', *sub))
cfg.model['burn_in_step'] = -1 if phase2 else 10 ** 9
torch.manual_seed(0)
model = pta.build_detector(cfg.model).to(dev)
benchmark_init_(model, phase2=phase2)
model.train()
model.teacher_stream = False      # per-launch HIP-event timings: no co-scheduled teacher kernels on a second stream
tr = pta.Trainer(model, cfg.optimizer, cfg.optimizer_config, cfg.lr_config, channels_last=True)
data = SyntheticTiles(n=16, size=1200 if obb else 800, mean_objects=300, seed=7, device=dev, oriented=obb, num_classes=9 if obb else 8)
for it in range(6):
    tr.step(data.batch(it, 2))
torch.cuda.synchronize()
orig = hip.call
rec = []
groups = defaultdict(lambda: [0, 0.0, 0.0])
gkeys = {}


def hook(fn, *a):
    if fn in ('pt_conv_bf16x6', 'pt_conv_wgrad_bf16x6', 'pt_gemm_bf16x6_nt'):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(fn, *a)
        e1.record()
        if fn == 'pt_gemm_bf16x6_nt':
            key, fl = (fn, a[7], a[9], a[8], 1, ''), 2.0 * a[7] * a[8] * a[9]
        else:
            d = a[0]
            Ho, Wo = (d.Hs + 2 * d.pad - d.KH) // d.stride + 1, (d.Ws + 2 * d.pad - d.KW) // d.stride + 1
            if fn == 'pt_conv_bf16x6' and d.dstride > 1:
                Ho, Wo = d.out_H, d.out_W
            M = d.B * Ho * Wo
            flav = ''
            grp = 'FC' if (d.B == 1 and d.Ws == 1) else ('3x3' if d.KH == 3 else '1x1')
            if fn == 'pt_conv_bf16x6':
                flav = ('P' if d.out_planes else '') + ('F' if d.out_f32 else '') + ('+rp' if d.res_planes else '') + ('+rf' if d.res_f32 else '') + \
                       ('+m' if d.mask_planes else '') + (f' s{d.stride}' if d.stride > 1 else '') + (' T' if d.dstride > 1 else '') + \
                       (' sc' if d.scatter_stride else '') + (f' k{d.splits}' if d.splits > 1 else '')
            else:
                flav = (f's{d.stride} ' if d.stride > 1 else '') + ('b' if d.dbias else '') + f' S{d.splits}'
            flav = ('h2 ' if d.operand_f16 else ('b1 ' if d.np == 1 else 'b3 ')) + flav + (' >h2' if (fn == 'pt_conv_bf16x6' and d.out_f16) else '')
            key, fl = (fn, M, d.Cin, d.Cout, d.KH * d.KW, flav), 2.0 * M * d.Cin * d.Cout * d.KH * d.KW
            groups[grp][0] += 1
            gkeys[key] = grp
        rec.append((key, fl, e0, e1))
        return r
    return orig(fn, *a)


hip.call = PF.hip.call = PPL.hip.call = hook
model.teacher_graph_off = True          # (the launches of a replayed HIP graph cannot be bracketed one by one)
N = 3
for it in range(N):
    tr.step(data.batch(10 + it, 2))
torch.cuda.synchronize()
hip.call = PF.hip.call = PPL.hip.call = orig
agg = defaultdict(lambda: [0.0, 0, 0.0])
for key, fl, e0, e1 in rec:
    a = agg[key]
    a[0] += e0.elapsed_time(e1)
    a[1] += 1
    a[2] += fl
rows = sorted(((v[0] / N, v[1] / N, v[2] / v[0] / 1e9, k) for k, v in agg.items()), reverse=True)
tot = sum(r[0] for r in rows)
print(f'matrix family: {tot:.2f} ms / iteration in {sum(r[1] for r in rows):.0f} launches ({"phase 2" if phase2 else "phase 1"}{" obb" if obb else ""}); '
      f'{sum(v[2] for v in agg.values()) / N / tot / 1e9:.1f} TFLOP/s fp32-equivalent overall (HIP events incl. ~5 us per launch)')
for k, v in agg.items():
    if k in gkeys:
        groups[gkeys[k]][1] += v[0] / N
        groups[gkeys[k]][2] += v[2] / N
for gname, (cnt, ms, fl) in sorted(groups.items()):
    print(f'  group {gname:4s}: {ms:7.3f} ms / iteration, {cnt / N:6.1f} launches, {fl / ms / 1e9 if ms else 0:6.1f} TFLOP/s fp32-equivalent')
print(f'{"ms/iter":>8} {"calls":>6} {"TF":>6}  entry  M  Cin->Cout  taps  flavour')
for ms, n, tf, k in rows:
    print(f'{ms:8.3f} {n:6.1f} {tf:6.1f}  {k[0][3:]:22s} M={k[1]:6d} {k[2]:5d}->{k[3]:5d} t{k[4]} {k[5]}')
