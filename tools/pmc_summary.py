"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per launch."""
import csv, sys, glob, collections
paths = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
want = sys.argv[2:] if len(sys.argv) > 2 else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in paths:
    for r in csv.DictReader(open(p)):
        n = r.get('Kernel_Name') or r.get('Kernel Name')
        if 'pt::' not in n:
            continue
        key = 'pt::' + n.split('pt::')[1].split('(')[0]
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in agg[k].items()})
