# usage (GPU box, repo root): bash tools/tune_gemms.sh
# Records the best rocBLAS / hipBLASLt solution THAT REPRODUCES THE DEFAULT SOLUTION'S RESULT (numerical check 1e-4) for every
# GEMM shape of the benchmarked workloads (PyTorch TunableOp) into point_teacher_amd/tuned/gemm_gfx950.csv.
export PT_TUNE_GEMMS=1
rm -f gpurun_out/tune_*.csv
i=0
for args in "--workload step1" "--workload step2" "--workload step2 --percent 100 --steps 4 --warmup 2" "--variant obb --workload step2 --steps 4 --warmup 2"; do
  i=$((i+1))
  PT_TUNE_GEMMS_OUT=gpurun_out/tune_$i.csv python bench.py $args --no-cpu-baseline --no-phase2 > gpurun_out/tune.out 2> gpurun_out/tune.err || tail -3 gpurun_out/tune.err
  tail -c 200 gpurun_out/tune.out; echo; wc -l gpurun_out/tune_$i.csv
done
# merge: validators of the first file + the union of the entries
(grep '^Validator' gpurun_out/tune_1.csv; cat gpurun_out/tune_*.csv | grep -v '^Validator' | sort -u -t, -k1,2) > point_teacher_amd/tuned/gemm_gfx950.csv
cp point_teacher_amd/tuned/gemm_gfx950.csv gpurun_out/gemm_gfx950.csv
wc -l point_teacher_amd/tuned/gemm_gfx950.csv
