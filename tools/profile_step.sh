# usage (on the GPU box, from the repo root): bash tools/profile_step.sh <tag> <bench args...>
# rocprofv3 kernel trace of bench.py, reduced to the steady-state window of the last 8 iterations.
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python bench.py --steps 10 --warmup 6 --no-cpu-baseline --no-phase2 "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
tr=$(find $out -name '*kernel_trace.csv' | head -1)
python tools/trace_window.py $tr 8 gpurun_out/${tag}_kernel_stats.csv > gpurun_out/${tag}_window.txt
python tools/prof_summary.py gpurun_out/${tag}_kernel_stats.csv 8 40 >> gpurun_out/${tag}_window.txt
cp $out/bench.json gpurun_out/${tag}_bench.json
rm -rf $out
cat gpurun_out/${tag}_window.txt
