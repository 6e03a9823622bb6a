# usage (GPU box, repo root): bash tools/profile_aux.sh   -> rocprofv3 per-kernel stats of the N2 image preparation
# (tools/prep_bench.py); whole-run statistics include MIOpen's find pass, so the training benches are profiled with
# tools/profile_step.sh (steady-state window) instead
export TMPDIR=/tmp
for job in prep_bench; do
  out=gpurun_out/prof_$job
  rm -rf $out && mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python tools/$job.py > $out/run.json 2> $out/run.err || { tail -5 $out/run.err; exit 1; }
  st=$(find $out -name '*kernel_stats.csv' | head -1)
  python - "$st" > gpurun_out/${job}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot / 1e6:.2f} ms')
for r in rows[:25]:
    print(f"{float(r['TotalDurationNs']) / 1e6:9.3f} ms {100 * float(r['TotalDurationNs']) / tot:6.2f}% {int(r['Calls']):6d} calls avg {float(r['AverageNs']) / 1e3:9.2f} us  {r['Name'][:110]}")
PY
  cp $out/run.json gpurun_out/${job}_under_rocprof.json
  rm -rf $out
  cat gpurun_out/${job}_kernel_stats.txt | head -14
done
