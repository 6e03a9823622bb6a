# usage (GPU box, repo root): bash tools/profile_baseline.sh fcos|retinanet
# rocprofv3 kernel trace of tools/bench_baseline.py reduced to the steady-state window of the last 8 iterations.
model=$1
export TMPDIR=/tmp
out=gpurun_out/prof_base_$model
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python tools/bench_baseline.py --model $model > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
tr=$(find $out -name '*kernel_trace.csv' | head -1)
python tools/trace_window.py $tr 8 gpurun_out/${model}_baseline_kernel_stats.csv > gpurun_out/${model}_baseline_summary.txt
python tools/prof_summary.py gpurun_out/${model}_baseline_kernel_stats.csv 8 30 >> gpurun_out/${model}_baseline_summary.txt
rm -rf $out
head -24 gpurun_out/${model}_baseline_summary.txt
