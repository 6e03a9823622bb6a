# Final measurements of round 5, part B: rocprofv3 kernel traces (phase 1, phase 2, oriented phase 2) and the PMC traffic passes
bash tools/profile_step.sh r05_step1 --no-configs2 --no-strict > /dev/null 2>&1
bash tools/profile_step.sh r05_step2 --workload step2 --no-configs2 --no-strict > /dev/null 2>&1
bash tools/profile_step.sh r05_obb_step2 --variant obb --workload step2 --no-configs2 --no-strict --tiles 16 > /dev/null 2>&1
bash tools/profile_step.sh r05_step2_bf16 --workload step2 --dtype bf16 --no-configs2 --no-strict > /dev/null 2>&1
head -3 gpurun_out/r05_step1_window.txt; head -3 gpurun_out/r05_step2_window.txt; head -3 gpurun_out/r05_obb_step2_window.txt; head -3 gpurun_out/r05_step2_bf16_window.txt
bash tools/run_r05_pmc.sh > /dev/null 2>&1
ls gpurun_out/pmc_r05_*
