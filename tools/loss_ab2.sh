# final loss of the default (phase-1) line, several runs with the 64-column tiles on / off and the teacher graph on / off
for rep in 1 2 3 4; do for n in 1 0; do
PT_CONV_NARROW=$n python bench.py --no-cpu-baseline --no-phase2 --no-configs2 --no-strict --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('narrow=$n', j['ms_per_step'], 'loss', j['loss'], 'largest', j['f16_census']['largest_stored'])"
done; done
