python -m pytest tests -m gpu -x -q > gpurun_out/s19_tests.log 2>&1; grep -E "passed|failed" gpurun_out/s19_tests.log | head -3; grep -E "^E  |^FAILED" gpurun_out/s19_tests.log | head -10
python tools/layer_bench.py > gpurun_out/s19_layers.log 2>&1; grep -E "discriminator.0.1|discriminator.1.0 |last_conv|TOTAL" gpurun_out/s19_layers.log | cut -c1-160
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "GPU:"; done
