"""Pixel-chunk (S) sweep of pt_conv_wgrad_bf16x6 on the step's weight-gradient shapes with H2 (scaled fp16 x 2) operands, reduce launch
included (what the step pays).  Buffers rotate over R sets so every call reads cold bytes.
usage: python tools/wgrad_sweep.py [b3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from point_teacher_amd import hip, planes as PL  # noqa: E402

dev = 'cuda:0'
H2 = 'b3' not in sys.argv[1:]
R = 3


def timeit(fns, n=12, reps=3):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for f in fns:
                f()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / (reps * len(fns)))
    ts.sort()
    return ts[len(ts) // 2]


def planes(B, H, W, C):
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    return PL.split_nhwc(x, f16=H2)[0]


def call(gy_t, x_t, B, H, W, Cin, Cout, K, S, ws, dw, db):
    d = hip.STRUCTS['pt_conv_wgrad_desc']()
    d.B, d.Hs, d.Ws, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad, d.accumulate = B, H, W, Cin, Cout, K, K, 1, K // 2, 0
    d.np, d.operand_f16, d.alpha = 3, int(H2), 1.0
    if H2:
        d.alpha_dev = PL.tail_ptr(gy_t)
    d.gy_planes, d.gy_plane_stride = PL._p16(gy_t), gy_t.shape[1]
    d.x_planes, d.x_plane_stride = PL._p16(x_t), x_t.shape[1]
    d.dw, d.dbias = hip.dptr(dw, 'float'), hip.dptr(db, 'float')
    d.workspace, d.workspace_elems, d.splits = hip.dptr(ws, 'float'), ws.numel(), S
    hip.call('pt_conv_wgrad_bf16x6', d)


def sweep(name, B, H, W, Cin, Cout, K, cands):
    M = B * H * W
    n = Cout * K * K * Cin
    gys = [planes(B, H, W, Cout) for _ in range(R)]
    xs = [planes(B, H, W, Cin) for _ in range(R)]
    auto = PL._splits('pt_conv_wgrad_bf16x6_splits', B, H, W, K, K, Cin, Cout)
    dw = torch.empty(n, device=dev)
    db = torch.empty(Cout, device=dev)
    out = []
    for S in sorted(set(cands + [auto])):
        if S * 4 * 32 > M + 127:
            continue
        ws = torch.empty(S * (n + Cout), device=dev)
        us = timeit([(lambda g=g, x=x: call(g, x, B, H, W, Cin, Cout, K, S, ws, dw, db)) for g, x in zip(gys, xs)])
        out.append((S, us))
    best = min(out, key=lambda t: t[1])
    fl = 2.0 * M * n
    a = dict(out)[auto]
    print(f'{name:28s} M={M:6d} {Cin:5d}->{Cout:5d} k{K} ' + ' '.join(f'S{s}:{u:6.1f}' for s, u in out) +
          f' | auto S{auto} {a:6.1f} us = {fl / a / 1e6:5.1f} TF, best S{best[0]} {best[1]:6.1f} us = {fl / best[1] / 1e6:5.1f} TF', flush=True)


if __name__ == '__main__':
    print('operands:', 'H2 (fp16 x 2)' if H2 else 'B3 (bf16 x 3)')
    sweep('FC1 5000 rows', 1, 50, 100, 12544, 1024, 1, [1, 2, 3, 4, 5, 6, 8, 13])
    sweep('FC1 5000 x 1 x 1', 5000, 1, 1, 12544, 1024, 1, [1, 2, 3, 4, 5, 6])
    sweep('FC2 5000 x 1 x 1', 5000, 1, 1, 1024, 1024, 1, [4, 8, 13, 16, 20, 26, 32])
    sweep('tower 3x3 (B=2)', 2, 100, 100, 256, 256, 3, [7, 10, 14, 18, 21, 28])
    sweep('tower 3x3 (B=6)', 6, 100, 100, 256, 256, 3, [7, 14, 21, 28, 42])
    sweep('layer3 3x3 (B=6)', 6, 50, 50, 256, 256, 3, [7, 10, 14, 21, 28])
    sweep('layer2 3x3 (B=6)', 6, 100, 100, 128, 128, 3, [14, 21, 28, 42, 56])
    sweep('layer4 3x3 (B=6)', 6, 25, 25, 512, 512, 3, [1, 2, 3, 4, 6])
    sweep('layer2 1x1 128->512 (B=6)', 6, 100, 100, 128, 512, 1, [32, 64, 96, 125])
    sweep('layer3 1x1 256->1024 (B=6)', 6, 50, 50, 256, 1024, 1, [8, 16, 24, 32, 48])
    sweep('layer3 1x1 1024->256 (B=6)', 6, 50, 50, 1024, 256, 1, [8, 16, 24, 32, 48])
    sweep('layer4 1x1 512->2048 (B=6)', 6, 25, 25, 512, 2048, 1, [2, 4, 6, 8, 12])
