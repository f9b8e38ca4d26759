mkdir -p gpurun_out/r2k
( time python3 bench.py > gpurun_out/r2k/bench.json 2> gpurun_out/r2k/bench.err ) 2> gpurun_out/r2k/bench.time
tail -3 gpurun_out/r2k/bench.err; cat gpurun_out/r2k/bench.time
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r2k/bench.json'))
print({k: d[k] for k in ('metric','value','ms_per_step','dtype')}); print(d['config']); print(d['roofline']); print(d['solver'])
for s in d.get('secondary', []): print(s)
print(d.get('cpu_baseline'))
a = d.get('accuracy'); print(a['seconds'])
for r in a['rows']: print(json.dumps(r))
PY
timeout 900 python -m pytest tests/test_gpu_accuracy.py -x -q -s 2>&1 | tail -25
