# part C: oriented lines (with the CPU baseline of the oriented oracle), PMC traffic of the matrix family, per-shape breakdowns
export TMPDIR=/tmp
python bench.py --variant obb --workload step2 > gpurun_out/r04_bench_obb_step2_fp32.json 2> /dev/null
python bench.py --variant obb --workload step1 --no-phase2 --no-cpu-baseline > gpurun_out/r04_bench_obb_step1_fp32.json 2> /dev/null
bash tools/pmc_pass.sh r04_step1 --no-configs2 > /dev/null 2>&1 && python tools/pmc_to_json.py r04_step1 step1 gpurun_out/r04_pmc_traffic.json > /dev/null
bash tools/pmc_pass.sh r04_step2 --workload step2 > /dev/null 2>&1 && python tools/pmc_to_json.py r04_step2 step2 gpurun_out/r04_pmc_traffic.json > /dev/null
python tools/family_breakdown.py step1 > gpurun_out/r04_family_breakdown_step1.txt 2>&1
python tools/family_breakdown.py step2 > gpurun_out/r04_family_breakdown_step2.txt 2>&1
python tools/plane_conv_bench.py 6 > gpurun_out/r04_plane_conv_bench_b6.txt 2>&1
python tools/plane_conv_bench.py 2 np1 > gpurun_out/r04_plane_conv_bench_b2_np1.txt 2>&1
tail -3 gpurun_out/r04_plane_conv_bench_b6.txt
