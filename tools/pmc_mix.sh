#!/bin/bash
# Dynamic instruction mix of the headline workload (one solve call): all instructions and the classes the SQ counts,
# plus issue-side cycle counters. Usage (inside gpurun, repo root): bash tools/pmc_mix.sh [out dir under gpurun_out]
R=$(pwd); OUT=$R/gpurun_out/${1:-pmc_mix}; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
B="python3 $R/bench.py --no-cpu-baseline --no-secondary --no-accuracy --steps 1 --warmup 0"
rocprofv3 --pmc SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_VSKIPPED --output-format csv -d $OUT/a -- $B > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -- $B > /dev/null 2> $OUT/b.err
python3 $R/tools/pmc_summary.py steps=1 $OUT/a $OUT/b
