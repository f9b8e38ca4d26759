mkdir -p gpurun_out/r2m
timeout 1700 python -m pytest tests/test_gpu_coop.py tests/test_gpu_parity.py tests/test_gpu_options.py -x -q 2>&1 | tail -15 > gpurun_out/r2m/tests.txt
cat gpurun_out/r2m/tests.txt
rm -f gpurun_out/r2m/*.jsonl
L="--no-cpu-baseline --no-secondary --no-accuracy"
python3 bench.py $L --workload cfg4 --steps 1 --warmup 1 2>/dev/null | tail -1 >> gpurun_out/r2m/cfg4.jsonl
python3 bench.py $L --workload cfg4 --steps 1 --warmup 1 --coop-waves 1 2>/dev/null | tail -1 >> gpurun_out/r2m/cfg4.jsonl
python3 bench.py $L --workload cfg4 --steps 1 --warmup 1 --coop-waves 2 2>/dev/null | tail -1 >> gpurun_out/r2m/cfg4.jsonl
python3 bench.py $L --workload cfg4 --steps 1 --warmup 1 --dtype f64 2>/dev/null | tail -1 >> gpurun_out/r2m/cfg4.jsonl
python3 bench.py $L --workload cfg2 --batch 16384 --steps 1 --warmup 1 --reg-table -1 --coop-waves 4 --latency-waves 1 2>/dev/null | tail -1 >> gpurun_out/r2m/cfg4.jsonl
python3 bench.py $L --workload cfg2 --batch 16384 --steps 1 --warmup 1 --reg-table -1 --coop-waves 3 --latency-waves 1 2>/dev/null | tail -1 >> gpurun_out/r2m/cfg4.jsonl
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2m/*.jsonl')):
  for l in open(f):
    d = json.loads(l); print(d['config']['workload'][:4], d['dtype'], d['config']['batch_per_gpu'], round(d['value']), round(d['roofline']['kernel_ms'],1), d['roofline']['kernel'][:60])
PY
