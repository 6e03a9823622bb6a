"""Summarise a rocprofv3 --kernel-trace --stats CSV: top kernels, ms per iteration."""
import csv, sys, re
path, iters = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot/1e6/iters:.2f} ms/iter over {iters:g} iters, {sum(int(r["Calls"]) for r in rows)/iters:.0f} launches/iter')
groups = {}
for r in rows:
    n = r['Name']
    key = ('pt::' + n.split('pt::')[1].split('(')[0].split('<')[0]) if 'pt::' in n else \
          ('MIOpen conv fwd/bwd (winograd/asm)' if 'miopenSp3AsmConv' in n or 'Winograd' in n.lower() else
           'MIOpen igemm' if n.startswith('igemm') else
           'Tensile GEMM (Cijk)' if n.startswith('Cijk') else
           'MIOpen BN infer' if 'BatchNorm' in n else
           'transpose (MIOpen layout)' if 'transpose' in n.lower() else
           'torch elementwise/reduce' if 'at::native' in n else
           'copy/fill' if 'copyBuffer' in n or 'fillBuffer' in n.lower() else n[:60])
    g = groups.setdefault(key, [0, 0.0])
    g[0] += int(r['Calls']); g[1] += float(r['TotalDurationNs'])
for k, (c, t) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f'{t/1e6/iters:9.3f} ms/iter {100*t/tot:6.2f}%  {c/iters:8.1f} calls/iter  avg {t/c/1e3:9.1f} us  {k}')
