# Round 5, measurement A (GPU box, repo root): default bench line + family breakdown of both phases
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
tail -c 600 gpurun_out/r05_bench_default.json; echo
python tools/family_breakdown.py step1 > gpurun_out/r05_family_step1.txt 2>&1; head -8 gpurun_out/r05_family_step1.txt
python tools/family_breakdown.py step2 > gpurun_out/r05_family_step2.txt 2>&1; head -8 gpurun_out/r05_family_step2.txt
