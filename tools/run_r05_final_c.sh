# Final measurements of round 5, part C (after the teacher graph and the glue edits): default bench line, kernel traces, family breakdowns
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_final_bench_default.json 2> gpurun_out/r05_final_bench_default.err
tail -c 200 gpurun_out/r05_final_bench_default.json; echo
bash tools/profile_step.sh r05_step1 --no-configs2 --no-strict > /dev/null 2>&1
bash tools/profile_step.sh r05_step2 --workload step2 --no-configs2 --no-strict > /dev/null 2>&1
bash tools/profile_step.sh r05_step2_bf16 --workload step2 --dtype bf16 --no-configs2 --no-strict > /dev/null 2>&1
bash tools/profile_step.sh r05_obb_step2 --variant obb --workload step2 --no-configs2 --no-strict --tiles 16 > /dev/null 2>&1
head -2 gpurun_out/r05_step1_window.txt; head -2 gpurun_out/r05_step2_window.txt; head -2 gpurun_out/r05_step2_bf16_window.txt; head -2 gpurun_out/r05_obb_step2_window.txt
python tools/family_breakdown.py step1 > gpurun_out/r05_final_family_step1.txt 2>&1; grep "matrix family\|group" gpurun_out/r05_final_family_step1.txt
python tools/family_breakdown.py step2 > gpurun_out/r05_final_family_step2.txt 2>&1; grep "matrix family\|group" gpurun_out/r05_final_family_step2.txt
