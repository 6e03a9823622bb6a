# final loss of the default (phase-1) line: teacher graph on / off, five runs each
for rep in 1 2 3 4 5; do for g in 1 0; do
PT_TEACHER_GRAPH=$g python bench.py --no-cpu-baseline --no-phase2 --no-configs2 --no-strict --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph=$g', j['ms_per_step'], 'loss', j['loss'])"
done; done
