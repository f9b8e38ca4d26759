mkdir -p gpurun_out/r2f
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_latency_mode.py tests/test_gpu_options.py -x -q 2>&1 | tail -15 > gpurun_out/r2f/tests.txt
cat gpurun_out/r2f/tests.txt
rm -f gpurun_out/r2f/*.jsonl
for w in 1 2 3; do python3 bench.py --workload cfg2 --batch 16384 --steps 2 --warmup 1 --no-cpu-baseline --latency-waves $w 2>/dev/null | tail -1 >> gpurun_out/r2f/cfg2.jsonl; done
python3 bench.py --workload cfg2 --steps 1 --warmup 1 --no-cpu-baseline --latency-waves 1 2>/dev/null | tail -1 >> gpurun_out/r2f/cfg2.jsonl
python3 bench.py --workload cfg1 --batch 65536 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r2f/cfg1.jsonl
python3 bench.py --workload cfg1 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r2f/cfg1.jsonl
python3 bench.py --workload cfg1 --steps 5 --warmup 1 --no-cpu-baseline --latency-waves 1 2>/dev/null | tail -1 >> gpurun_out/r2f/cfg1.jsonl
python3 bench.py --workload cfg4 --batch 2048 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r2f/cfg4.jsonl
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2f/*.jsonl')):
  for l in open(f):
    d = json.loads(l); print(d['config']['workload'][:4], d['config']['batch_per_gpu'], d['config']['latency_waves'], round(d['value']), round(d['roofline']['kernel_ms'],1), d['roofline']['kernel'][:44], 'conv %.3f'%d['solver']['converged_frac'], 'inner %.0f'%d['solver']['inner_iters_mean'])
PY
