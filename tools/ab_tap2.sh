python -m pytest tests/test_gpu_ops.py -m gpu -q -k "batched_input" 2>&1 | tail -1
python -m pytest tests -m gpu -x -q > gpurun_out/s25_tests.log 2>&1; grep -E "passed|failed" gpurun_out/s25_tests.log | head -3; grep -E "^E  |^FAILED" gpurun_out/s25_tests.log | head -10
