# Round 5, measurement B: the oriented config 5 (both phases) and the 100 % config (both phases), stand-alone lines
common="--steps 16 --warmup 16 --no-cpu-baseline --no-phase2 --no-configs2 --no-strict --tiles 16"
python bench.py $common --variant obb --workload step2 > gpurun_out/r05_bench_obb_step2.json 2> gpurun_out/r05_bench_obb_step2.err
python bench.py $common --variant obb --workload step1 > gpurun_out/r05_bench_obb_step1.json 2> gpurun_out/r05_bench_obb_step1.err
python bench.py $common --percent 100 --workload step2 > gpurun_out/r05_bench_p100_step2.json 2> gpurun_out/r05_bench_p100_step2.err
python bench.py $common --percent 100 --workload step1 > gpurun_out/r05_bench_p100_step1.json 2> gpurun_out/r05_bench_p100_step1.err
python - <<P
import json
for n in ('obb_step2','obb_step1','p100_step2','p100_step1'):
    try:
        d=json.load(open(f'gpurun_out/r05_bench_{n}.json'))
        print(n, d['ms_per_step'], 'ms', d['value'], 'it/s loss', d['loss'], 'demoted', d['f16_census']['demoted'])
    except Exception as e:
        print(n, 'failed', e)
P
