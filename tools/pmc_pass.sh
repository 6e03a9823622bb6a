# usage (GPU box, repo root): bash tools/pmc_pass.sh <tag> <bench args...>
# two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) as the MI355X guide prescribes; summaries only are kept.
tag=$1; shift
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_${tag}_$ctr
  rm -rf $out && mkdir -p $out
  rocprofv3 --pmc $ctr --output-format csv -d $out -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-phase2 "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
  python tools/pmc_summary.py $out > gpurun_out/pmc_${tag}_$ctr.txt
  rm -rf $out
done
cat gpurun_out/pmc_${tag}_FETCH_SIZE.txt | head -60
