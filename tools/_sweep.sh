mkdir -p gpurun_out/r2j
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2j/tests.txt
cat gpurun_out/r2j/tests.txt
