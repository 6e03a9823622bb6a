export TMPDIR=/tmp
out=gpurun_out/prof_wg
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-phase2 --no-configs2 --no-strict > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
tr=$(find $out -name '*kernel_trace.csv' | head -1)
python - $tr <<'P'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'wgrad' in n:
        key=(n[:60], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('LDS_Block_Size') )
        agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    v2=sorted(v)
    print(f'{sum(v):10.1f} us total  n={len(v):4d}  med {v2[len(v2)//2]:8.1f}  min {v2[0]:8.1f} max {v2[-1]:8.1f}', k)
P
rm -rf $out
