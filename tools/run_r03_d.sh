python -m pytest tests/test_hip_ops.py tests/test_ops_surface.py tests/test_obb_parity.py tests/test_train_step_parity.py tests/test_hbb_product_vs_golden.py -q -x > gpurun_out/r03_gputest_d.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_gputest_d.log; tail -6 gpurun_out/r03_gputest_d.log
python tools/roi_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03_roi_bench.txt
for v in obb; do for w in step1 step2; do python bench.py --variant obb --workload $w --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r03_bench_obb_$w.json 2> gpurun_out/r03_bench_obb_$w.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r03_bench_obb_$w.json").read().strip().splitlines()[-1])
print("obb $w", d["ms_per_step"], {k:v for k,v in d["custom_kernels_ms_per_step"].items() if v>0.25})
PY
done; done
