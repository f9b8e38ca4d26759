mkdir -p gpurun_out/r2l
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2l/tests.txt
cat gpurun_out/r2l/tests.txt
python3 tools/bench_evaluate.py 1 60 f64 2>/dev/null | tail -1
