bash tools/run_profiles.sh r02 > gpurun_out/r02_profiles.log 2>&1
tail -5 gpurun_out/r02_profiles.log
cat gpurun_out/r02/cfg2_pmc_summary.txt | head -30
cat gpurun_out/r02/cfg2_traffic.json gpurun_out/r02/cfg4_traffic.json gpurun_out/r02/cfg1_traffic.json 2>&1 | head -60
head -5 gpurun_out/r02/cfg2_kernel_stats.csv
