"""Is the bf16x6 GEMM power- / clock-limited?  Runs the FC1 product (5 000 x 1 024 x 12 544) back to back for a few seconds while
sampling `rocm-smi` (socket power, shader clock) from a thread; then the same for a vendor bf16 GEMM of the same shape.
usage: python tools/power_probe.py"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from point_teacher_amd import functional as F  # noqa: E402

dev = torch.device('cuda:0')
M, N, K = 5000, 1024, 12544
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).to(dev)
b = (torch.randn(N, K, generator=g) * 0.05).to(dev)
ap, bp = F.split_bf16x3(a), F.split_bf16x3(b)
out = torch.empty(M, N, device=dev)
a16, b16 = a.bfloat16(), b.bfloat16().t().contiguous()


def sample(stop, rows):
    while not stop.is_set():
        try:
            t = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=10).stdout
            pw = re.findall(r'Power \(W\): ([0-9.]+)', t)
            sclk = re.findall(r'sclk clock level: \S+ \((\d+)Mhz\)', t)
            rows.append((time.time(), pw[:1], sclk[:1]))
        except Exception as e:
            rows.append((time.time(), str(e), None))
        time.sleep(0.3)


def run(name, fn, flops, seconds=4.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    stop, rows = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, rows))
    th.start()
    n, t0 = 0, time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = e0.elapsed_time(e1) / n
    print(f'{name}: {ms * 1e3:.1f} us / call, {flops / ms / 1e9:.1f} TFLOP/s (as counted); samples (power W, sclk MHz):',
          [(r[1], r[2]) for r in rows[1:-1]][:10], flush=True)


print(subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout[-900:])
run('idle-ish (sleep)', lambda: time.sleep(0.002), 1.0, 1.5)
run('bf16x6 GEMM 5000x1024x12544 (fp32-equivalent flops)', lambda: F.gemm_bf16x6_nt(ap, bp, out=out), 2.0 * M * N * K)
run('library bf16 GEMM same shape', lambda: torch.matmul(a16, b16), 2.0 * M * N * K)
