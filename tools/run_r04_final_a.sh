# Final measurements of round 4, part A (GPU box, repo root): full GPU suite, smoke, default bench, kernel-trace summaries
python -m pytest tests -m gpu -q > gpurun_out/r04_final_gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_final_gputest.log; tail -3 gpurun_out/r04_final_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_final_smoke.log 2>&1; tail -2 gpurun_out/r04_final_smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_final_bench_default.json 2> gpurun_out/r04_final_bench_default.err
tail -c 300 gpurun_out/r04_final_bench_default.json; echo
