# Round 5: HBM traffic of the dominant family from PMC counters (separate FETCH_SIZE / WRITE_SIZE passes, as the guide prescribes)
bash tools/pmc_pass.sh r05_step1 --no-configs2 --no-strict --tiles 8 > /dev/null 2>&1
bash tools/pmc_pass.sh r05_step2 --workload step2 --no-configs2 --no-strict --tiles 8 > /dev/null 2>&1
ls -la gpurun_out/pmc_r05_*; head -20 gpurun_out/pmc_r05_step1_FETCH_SIZE.txt | cut -c1-200
