# part D: PMC traffic of the oriented phase 2, RoI geometry of the oriented iteration, SQ / LDS counters of the bf16x6 GEMM
bash tools/pmc_pass.sh r03_obb_step2 --variant obb --workload step2 > /dev/null 2>&1 && python tools/pmc_to_json.py r03_obb_step2 obb_step2 > /dev/null
python tools/rroi_stats.py step2 > gpurun_out/r03_final_rroi_stats_step2.txt 2>&1
python tools/rroi_stats.py step1 > gpurun_out/r03_final_rroi_stats_step1.txt 2>&1
bash tools/gemm_pmc.sh final 5000 1024 12544 > gpurun_out/r03_final_gemm_pmc.txt 2>&1; tail -20 gpurun_out/r03_final_gemm_pmc.txt
