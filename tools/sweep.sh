cd $GRAFT_REPO_ROOT
for cfg in "--fold-bn 0" "--fold-bn 1" "--fold-bn 1 --miopen-find 1" "--fold-bn 1 --channels-last 1" "--fold-bn 1 --channels-last 1 --miopen-find 1"; do
  for wl in step2; do
    echo "== $wl $cfg" >> gpurun_out/sweep.log
    timeout -k 10 400 python bench.py --steps 10 --warmup 4 --workload $wl --no-cpu-baseline $cfg 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['loss'])" >> gpurun_out/sweep.log 2>&1
  done
done
cat gpurun_out/sweep.log
