# part B: kernel-trace summaries (steady-state windows) of the four workloads, stand-alone bench lines
export TMPDIR=/tmp
bash tools/profile_step.sh r04_step1_fp32 --no-configs2 > /dev/null 2>&1; head -3 gpurun_out/r04_step1_fp32_window.txt
bash tools/profile_step.sh r04_step2_fp32 --workload step2 > /dev/null 2>&1; head -3 gpurun_out/r04_step2_fp32_window.txt
bash tools/profile_step.sh r04_step2_bf16 --workload step2 --dtype bf16 > /dev/null 2>&1; head -3 gpurun_out/r04_step2_bf16_window.txt
bash tools/profile_step.sh r04_obb_step2_fp32 --variant obb --workload step2 > /dev/null 2>&1; head -3 gpurun_out/r04_obb_step2_fp32_window.txt
python bench.py --workload step2 --no-cpu-baseline > gpurun_out/r04_bench_step2_fp32.json 2> /dev/null
python bench.py --workload step2 --dtype bf16 --no-cpu-baseline > gpurun_out/r04_bench_step2_bf16.json 2> /dev/null
