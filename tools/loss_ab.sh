# final loss of the stand-alone phase-2 line, three runs each with the teacher graph on / off (divergence check)
for rep in 1 2 3; do for g in 1 0; do
PT_TEACHER_GRAPH=$g python bench.py --workload step2 --no-cpu-baseline --no-phase2 --no-configs2 --no-strict --steps 25 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph=$g', j['ms_per_step'], 'loss', j['loss'], 'demoted', j['f16_census']['demoted'], 'largest', j['f16_census']['largest_stored'])"
done; done
