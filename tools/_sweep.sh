timeout 900 python -m pytest tests/test_gpu_evaluate_reference.py tests/test_gpu_evaluate.py -x -q 2>&1 | tail -25
