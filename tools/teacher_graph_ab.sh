for rep in 1 2; do
for g in 1 0; do
for wl in "--workload step2 --dtype bf16" "--workload step2"; do
PT_TEACHER_GRAPH=$g python bench.py $wl --no-cpu-baseline --no-phase2 --no-configs2 --no-strict --steps 30 --warmup 8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph=$g', '$wl', j['ms_per_step'])"
done; done; done
