#!/bin/bash
# Counters of the fp64 N = 40 kernel (BASELINE configs[4], "fp64 vs fp32 tolerance sweep"): solve_coop_kernel<double,1,true,M>,
# M = 2 the general streamed table (9 values per entry), M = 1 the compressed one (5 values, axis-aligned ellipses), at a
# reduced batch (one launch each). Separate passes for FETCH_SIZE / WRITE_SIZE / the SQ counters (MI355X_MICROARCH.md).
# Usage (inside gpurun, repo root): bash tools/pmc_cfg4_f64.sh <out dir under gpurun_out> [batch]
set -u
R=$(pwd); OUT=$R/gpurun_out/${1:-r05/pmc_cfg4_f64}; BATCH=${2:-2048}; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --workload cfg4 --dtype f64 --batch $BATCH --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-accuracy"
for ax in -1 0; do
  tag=$([ $ax = 0 ] && echo compressed || echo general)
  $B --axis-aligned $ax > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/${tag}_$c -- $B --axis-aligned $ax > /dev/null 2> $OUT/${tag}_$c.err
  done
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/${tag}_sq -- $B --axis-aligned $ax > /dev/null 2> $OUT/${tag}_sq.err
  (python3 $R/tools/pmc_summary.py steps=1 $OUT/${tag}_FETCH_SIZE $OUT/${tag}_WRITE_SIZE; python3 $R/tools/pmc_summary.py steps=1 $OUT/${tag}_sq) > $OUT/pmc_summary_$tag.txt 2>&1
done
cat $OUT/bench_general.json $OUT/bench_compressed.json
cat $OUT/pmc_summary_general.txt $OUT/pmc_summary_compressed.txt
