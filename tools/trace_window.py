"""Steady-state kernel statistics from a rocprofv3 --kernel-trace CSV: only dispatches inside the
last N training iterations (iteration boundary = a pt::sgd_kernel dispatch) are aggregated, so
MIOpen's first-touch search kernels and other warm-up work do not pollute the per-iteration picture.
usage: trace_window.py <kernel_trace.csv> <n_last_iters> <out_stats.csv>"""
import csv, sys, collections
path, n_last, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
sgd_ends = [e for s, e, n in rows if 'pt::sgd_kernel' in n]
assert len(sgd_ends) > n_last, (len(sgd_ends), n_last)
t0, t1 = sgd_ends[-n_last - 1], sgd_ends[-1]
agg = collections.defaultdict(lambda: [0, 0, 10**18, 0])
busy = 0
for s, e, n in rows:
    if s >= t0 and e <= t1:
        a = agg[n]; d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        busy += d
tot = sum(a[1] for a in agg.values())
with open(out, 'w', newline='') as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3], 0])
print(f'window {(t1 - t0) / 1e6:.2f} ms over {n_last} iterations = {(t1 - t0) / 1e6 / n_last:.2f} ms/iter wall; kernel busy {busy / 1e6 / n_last:.2f} ms/iter')
