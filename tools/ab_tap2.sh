python -m pytest tests -m gpu -x -q > gpurun_out/s18_tests.log 2>&1; grep -E "passed|failed" gpurun_out/s18_tests.log | head -3; grep -E "^E  " gpurun_out/s18_tests.log | head -10
python tools/layer_bench.py > gpurun_out/s18_layers.log 2>&1; tail -1 gpurun_out/s18_layers.log
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "GPU:"; done
