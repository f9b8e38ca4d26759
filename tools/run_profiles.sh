#!/bin/bash
# Round profile pass on the GPU box: bench lines + rocprofv3 kernel stats + PMC passes, written to gpurun_out/$1/.
# Usage (from the repo root, inside gpurun): bash tools/run_profiles.sh r01
set -u
R=$(pwd); OUT=$R/gpurun_out/${1:-r01}; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py"
# 1. headline line (with CPU baseline), same command the driver runs
$B > $OUT/cfg1_bench.json 2> $OUT/cfg1_bench.err
# 2. kernel trace + stats of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --no-cpu-baseline > $OUT/cfg1_bench_under_rocprof.json 2> $OUT/stats.err
# 3. PMC passes (separate runs, counters only)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- $B --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$c.err
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- $B --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq.err
python3 $R/tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq > $OUT/cfg1_pmc_summary.txt 2>&1
# 4. the other lines of DESIGN.md's table
$B --no-cpu-baseline --latency-waves 1 > $OUT/cfg1_throughput_kernel_bench.json 2>/dev/null
$B --no-cpu-baseline --dtype f64 > $OUT/cfg1_f64_bench.json 2>/dev/null
$B --no-cpu-baseline --batch 65536 --steps 3 --warmup 1 > $OUT/cfg1dims_b65536_bench.json 2>/dev/null
$B --no-cpu-baseline --workload cfg2 --steps 2 --warmup 1 > $OUT/cfg2_full_bench.json 2>/dev/null
$B --no-cpu-baseline --workload cfg4 --steps 1 --warmup 1 > $OUT/cfg4_f32_bench.json 2>/dev/null
$B --no-cpu-baseline --workload cfg4 --dtype f64 --steps 1 --warmup 1 > $OUT/cfg4_f64_bench.json 2>/dev/null
python3 $R/tools/bench_assemble.py 2>/dev/null | tail -1 > $OUT/f1_assemble_bench.json
python3 $R/tools/bench_hypotheses.py 2>/dev/null | tail -1 > $OUT/f2_hypotheses_bench.json
python3 $R/tools/bench_evaluate.py 4096 60 f32 2>/dev/null | tail -1 > $OUT/f3_evaluate_b4096.json
python3 $R/tools/bench_evaluate.py 256 60 f32 2>/dev/null | tail -1 > $OUT/f3_evaluate_b256.json
python3 $R/tools/bench_evaluate.py 1 60 f64 2>/dev/null | tail -1 > $OUT/f3_evaluate_b1_f64.json
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/cfg1_kernel_stats.csv
ls $OUT
