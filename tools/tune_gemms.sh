# usage (GPU box, repo root): bash tools/tune_gemms.sh
# Records the best rocBLAS / hipBLASLt solution for every GEMM shape of the benchmarked workloads (PyTorch TunableOp) and
# copies the results file to point_teacher_amd/tuned/gemm_gfx950.csv, which runtime.enable_tuned_gemms reads.
export PT_TUNED_GEMMS=0 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tune.csv
cp point_teacher_amd/tuned/gemm_gfx950.csv gpurun_out/tune0.csv 2>/dev/null
for args in "--workload step1" "--workload step2" "--workload step2 --percent 100 --steps 4 --warmup 2" "--workload step1 --percent 100 --steps 4 --warmup 2" "--variant obb --workload step2 --steps 4 --warmup 2" "--variant obb --workload step1 --steps 4 --warmup 2"; do
  python bench.py $args --no-cpu-baseline --no-phase2 > gpurun_out/tune.out 2> gpurun_out/tune.err || tail -3 gpurun_out/tune.err
  tail -c 300 gpurun_out/tune.out; echo; wc -l gpurun_out/tune0.csv
done
cp gpurun_out/tune0.csv point_teacher_amd/tuned/gemm_gfx950.csv
