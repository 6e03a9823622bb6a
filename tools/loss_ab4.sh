# final loss of the default (phase-1) line with the teacher pass inline on the main stream (no side stream, no graph): eight runs
for rep in 1 2 3 4 5 6 7 8; do
PT_TEACHER_STREAM=0 PT_TEACHER_GRAPH=0 python bench.py --no-cpu-baseline --no-phase2 --no-configs2 --no-strict --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inline teacher', j['ms_per_step'], 'loss', j['loss'])"
done
