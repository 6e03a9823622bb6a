# part B: bf16 / oriented kernel traces, PMC traffic, the other workloads' bench lines, micro-benchmarks
bash tools/profile_step.sh r03_step2_bf16 --workload step2 --dtype bf16 > /dev/null 2>&1; head -3 gpurun_out/r03_step2_bf16_window.txt
bash tools/profile_step.sh r03_obb_step2_fp32 --variant obb --workload step2 > /dev/null 2>&1; head -3 gpurun_out/r03_obb_step2_fp32_window.txt
bash tools/pmc_pass.sh r03_step1 --no-configs2 > /dev/null 2>&1 && python tools/pmc_to_json.py r03_step1 step1
bash tools/pmc_pass.sh r03_step2 --workload step2 > /dev/null 2>&1 && python tools/pmc_to_json.py r03_step2 step2
bash tools/pmc_pass.sh r03_obb_step2 --variant obb --workload step2 > /dev/null 2>&1 && python tools/pmc_to_json.py r03_obb_step2 obb_step2
for w in step1 step2; do python bench.py --variant obb --workload $w --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r03_final_bench_obb_$w.json 2>/dev/null; done
python bench.py --workload step2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_final_bench_step2_fp32.json 2>/dev/null
for w in step1 step2; do python bench.py --workload $w --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_final_bench_${w}_bf16.json 2>/dev/null; done
for w in step1 step2; do python bench.py --workload $w --percent 100 --steps 8 --warmup 3 --no-cpu-baseline --no-phase2 > gpurun_out/r03_final_bench_p100_$w.json 2>/dev/null; done
python tools/gemm_bench.py > gpurun_out/r03_final_gemm_bench.txt 2>&1
python tools/conv_split_bench.py > gpurun_out/r03_final_conv_bench.txt 2>&1
python tools/roi_bench.py > gpurun_out/r03_final_roi_bench.txt 2>&1
python tools/rroi_stats.py step2 > gpurun_out/r03_final_rroi_stats_step2.txt 2>&1
python tools/rroi_stats.py step1 > gpurun_out/r03_final_rroi_stats_step1.txt 2>&1
PT_GEMM_VARIANT=0 bash tools/gemm_pmc.sh final 5000 1024 12544 > gpurun_out/r03_final_gemm_pmc.txt 2>&1
ls gpurun_out | grep r03_final | wc -l
