python -m pytest tests -m gpu -x -q > gpurun_out/s27_tests.log 2>&1; grep -E "passed|failed" gpurun_out/s27_tests.log | head -3; grep -E "^E  |^FAILED" gpurun_out/s27_tests.log | head -10
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "GPU:"; done
