#!/bin/bash
# Round profile pass on the GPU box: bench lines + rocprofv3 kernel stats + PMC passes, written to gpurun_out/$1/.
# Usage (from the repo root, inside gpurun): bash tools/run_profiles.sh r05
set -u
R=$(pwd); RN=${1:-r05}; OUT=$R/gpurun_out/$RN; mkdir -p $OUT
RNUM=$(echo $RN | sed 's/^r0*//; s/[^0-9].*$//')
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py"
LEAN="--no-cpu-baseline --no-secondary --no-accuracy"
# 1. headline line (secondary workloads, CPU baseline, accuracy table), same command the driver runs
$B > $OUT/cfg2_bench.json 2> $OUT/cfg2_bench.err
cp $R/bench_detail.json $OUT/cfg2_bench_detail.json 2>/dev/null
# 2. kernel trace + stats of the headline workload (same kernel, same batch)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B $LEAN --steps 3 > $OUT/cfg2_bench_under_rocprof.json 2> $OUT/stats.err
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/cfg2_kernel_stats.csv
# 3. PMC passes (separate runs, counters only)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- $B $LEAN --steps 2 --warmup 0 > /dev/null 2> $OUT/pmc_$c.err
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- $B $LEAN --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_sq.err
(python3 $R/tools/pmc_summary.py steps=2 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE; python3 $R/tools/pmc_summary.py steps=1 $OUT/pmc_sq) > $OUT/cfg2_pmc_summary.txt 2>&1
python3 $R/tools/make_traffic_json.py $RNUM cfg2 f32 65536 5928 20 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/cfg2_traffic.json 2 > /dev/null 2>> $OUT/pmc_sq.err
python3 $R/tools/kernel_stats_per_step.py $OUT/cfg2_kernel_stats.csv 4 > $OUT/cfg2_kernel_stats_per_step.txt 2>&1
# 3b. instruction cache of the FINAL headline kernel and the dynamic instruction mix
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $OUT/pmc_icache -- $B $LEAN --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_icache.err
python3 $R/tools/pmc_summary.py steps=1 $OUT/pmc_icache > $OUT/cfg2_icache_summary.txt 2>&1
(cd $R && bash tools/pmc_mix.sh $RN/pmc_mix) > $OUT/cfg2_pmc_instruction_mix.txt 2>&1
# 4. configs[4] traffic (the one configuration whose obstacle table is streamed from global memory) and configs[1]
for wl in cfg4:8192:40968:40 cfg1:1024:2778:20; do
  IFS=: read name batch np n <<< "$wl"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${name}_$c -- $B $LEAN --workload $name --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_${name}_$c.err
  done
  python3 $R/tools/make_traffic_json.py $RNUM $name f32 $batch $np $n $OUT/pmc_${name}_FETCH_SIZE $OUT/pmc_${name}_WRITE_SIZE $OUT/${name}_traffic.json 1 > /dev/null 2>> $OUT/pmc_sq.err
done
# 4b. the fp64 N = 40 kernel (configs[4] "fp64 vs fp32 tolerance sweep"): general / compressed streamed table, reduced batch
(cd $R && bash tools/pmc_cfg4_f64.sh $RN/pmc_cfg4_f64 2048) > $OUT/cfg4_f64_pmc.txt 2>&1
python3 - <<PY > $OUT/cfg4_f64_traffic.json 2>> $OUT/pmc_sq.err
import json, re
out = {"round": $RNUM, "workload": "cfg4", "dtype": "f64", "batch": 2048, "family": "toward_robot",
       "command": "tools/pmc_cfg4_f64.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_* (separate passes) -- python3 bench.py --workload cfg4 --dtype f64 --batch 2048 --steps 1 --warmup 0 [--axis-aligned -1]",
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; unit KB"}
for tag in ("general", "compressed"):
    t = open("$OUT/pmc_cfg4_f64/pmc_summary_%s.txt" % tag).read()
    g = lambda k: float(re.search(k + r"\s+([0-9.e+]+)", t).group(1))
    b = json.loads(open("$OUT/pmc_cfg4_f64/bench_%s.json" % tag).read().strip().splitlines()[-1])
    hbm = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024
    ms = b["roofline"]["kernel_ms"]
    out[tag] = {"kernel": b["roofline"]["kernel"], "solves_per_s": b["value"], "kernel_ms": ms, "psi_evals_per_solve": b["roofline"]["psi_evals_per_solve"],
                "FETCH_SIZE_KB": g("FETCH_SIZE"), "WRITE_SIZE_KB": g("WRITE_SIZE"), "hbm_bytes_per_launch": hbm,
                "hbm_TB_per_s": hbm / (ms * 1e-3) / 1e12, "frac_of_8TBps_peak": hbm / (ms * 1e-3) / 8e12,
                "SQ_INSTS_VALU": g("SQ_INSTS_VALU"), "SQ_WAIT_ANY_over_WAVE_CYCLES": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")}
print(json.dumps(out, indent=1))
PY
# 5. the next-row components
python3 $R/tools/bench_assemble.py 2>/dev/null | tail -1 > $OUT/f1_assemble_bench.json
python3 $R/tools/bench_hypotheses.py 2>/dev/null | tail -1 > $OUT/f2_hypotheses_bench.json
# (FAMILY=corridor: round 5's builder-designed family, comparable with the r05 records)
FAMILY=corridor python3 $R/tools/bench_evaluate.py 4096 60 f32 2>/dev/null | tail -1 > $OUT/f3_evaluate_b4096.json
FAMILY=corridor python3 $R/tools/bench_evaluate.py 1 60 f64 2>/dev/null | tail -1 > $OUT/f3_evaluate_b1_f64.json
FAMILY=corridor python3 $R/tools/bench_evaluate.py 65536 60 f32 2>/dev/null | tail -1 > $OUT/f3_evaluate_b65536.json
# ... and at BASELINE configs[2]'s dimensions: 4 pedestrians x 10 hypotheses (Ndynobs = 40), the closed loop itself
FAMILY=corridor python3 $R/tools/bench_evaluate.py 65536 60 f32 4 10 2>/dev/null | tail -1 > $OUT/f3_evaluate_b65536_4x10.json
# round 6: the REFERENCE's evaluation (scenario_0..2 on the warehouse map, 120-step cap), iteration caps only / with the
# reference's time cap as the evaluation budget
python3 $R/tools/bench_evaluate.py 65536 120 f32 4 10 2>/dev/null | tail -1 > $OUT/f3_evaluate_refscen_b65536_4x10.json
BUDGET=yaml python3 $R/tools/bench_evaluate.py 65536 120 f32 4 10 2>/dev/null | tail -1 > $OUT/f3_evaluate_refscen_b65536_4x10_budget.json
python3 $R/tools/solo_latency.py 2>/dev/null | grep "instance" > $OUT/solo_latency.txt
python3 $R/tools/kernel_resources.py > $OUT/kernel_resources.txt 2>/dev/null
python3 $R/tools/exp_three_per_wave.py > $OUT/exp_three_instances_per_wavefront.txt 2>/dev/null
# 6. the reference-scenario family of configs[2] under the kernel trace (same kernels as the headline, another distribution)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cl -- $B $LEAN --family refscen --steps 3 > $OUT/cfg2_refscen_bench_under_rocprof.json 2> $OUT/stats_cl.err
find $OUT/stats_cl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/cfg2_refscen_kernel_stats.csv
ls $OUT
