# Final measurements of round 5, part A (GPU box, repo root): full GPU suite, smoke, default bench, family breakdowns
python -m pytest tests -m gpu -q > gpurun_out/r05_final_gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05_final_gputest.log; tail -3 gpurun_out/r05_final_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_final_smoke.log 2>&1; tail -2 gpurun_out/r05_final_smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_final_bench_default.json 2> gpurun_out/r05_final_bench_default.err
tail -c 300 gpurun_out/r05_final_bench_default.json; echo
python tools/family_breakdown.py step1 > gpurun_out/r05_final_family_step1.txt 2>&1; head -8 gpurun_out/r05_final_family_step1.txt | tail -4
python tools/family_breakdown.py step2 > gpurun_out/r05_final_family_step2.txt 2>&1; head -8 gpurun_out/r05_final_family_step2.txt | tail -4
