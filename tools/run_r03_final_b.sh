# part B: bf16 kernel trace, PMC traffic of the HBB workloads, the other HBB bench lines, micro-benchmarks
bash tools/profile_step.sh r03_step2_bf16 --workload step2 --dtype bf16 > /dev/null 2>&1; head -3 gpurun_out/r03_step2_bf16_window.txt
bash tools/pmc_pass.sh r03_step1 --no-configs2 > /dev/null 2>&1 && python tools/pmc_to_json.py r03_step1 step1
bash tools/pmc_pass.sh r03_step2 --workload step2 > /dev/null 2>&1 && python tools/pmc_to_json.py r03_step2 step2
python bench.py --workload step2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_final_bench_step2_fp32.json 2>/dev/null
for w in step1 step2; do python bench.py --workload $w --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_final_bench_${w}_bf16.json 2>/dev/null; done
for w in step1 step2; do python bench.py --workload $w --percent 100 --steps 8 --warmup 3 --no-cpu-baseline --no-phase2 > gpurun_out/r03_final_bench_p100_$w.json 2>/dev/null; done
python tools/gemm_bench.py > gpurun_out/r03_final_gemm_bench.txt 2>&1
python tools/conv_split_bench.py > gpurun_out/r03_final_conv_bench.txt 2>&1
python tools/roi_bench.py > gpurun_out/r03_final_roi_bench.txt 2>&1
python tools/roi_stats.py step1 > gpurun_out/r03_final_roi_stats_step1.txt 2>&1
ls gpurun_out | grep r03_final | wc -l
