R=$(pwd); OUT=$R/gpurun_out/tmp_pmc; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-accuracy --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_sq.err
python3 $R/tools/pmc_summary.py steps=1 $OUT/pmc_sq
