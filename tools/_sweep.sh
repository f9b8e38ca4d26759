mkdir -p gpurun_out/r2c
(python3 tools/prof_sections.py 1024 cfg1; python3 tools/prof_sections.py 4096 cfg2; python3 tools/prof_sections.py 512 cfg4) > gpurun_out/r2c/sections.txt 2>&1
for w in 1 2 3 4; do python3 bench.py --workload cfg2 --batch 16384 --steps 2 --warmup 1 --no-cpu-baseline --latency-waves $w 2>/dev/null | tail -1 >> gpurun_out/r2c/cfg2_b16384_waves.jsonl; done
cat gpurun_out/r2c/sections.txt
python3 - <<'PY'
import json
for l in open('gpurun_out/r2c/cfg2_b16384_waves.jsonl'):
    d = json.loads(l); print(d['config']['latency_waves'], round(d['value']), d['roofline']['kernel_ms'], d['roofline']['valu']['psi_evals_per_solve'], d['solver'])
PY
