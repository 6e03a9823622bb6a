# part C: the oriented config (its MIOpen searches make every process start slowly): bench lines, kernel trace
for w in step2 step1; do python bench.py --variant obb --workload $w --steps 8 --warmup 4 --no-cpu-baseline --no-phase2 > gpurun_out/r03_final_bench_obb_$w.json 2>/dev/null; tail -c 200 gpurun_out/r03_final_bench_obb_$w.json; echo; done
bash tools/profile_step.sh r03_obb_step2_fp32 --variant obb --workload step2 > /dev/null 2>&1; head -3 gpurun_out/r03_obb_step2_fp32_window.txt
