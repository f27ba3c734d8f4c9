python -m pytest tests -m gpu -x -q > gpurun_out/s15_tests.log 2>&1; grep -E "passed|failed" gpurun_out/s15_tests.log | head -3; grep -E "^E  " gpurun_out/s15_tests.log | head -10
python tools/layer_bench.py > gpurun_out/s15_layers.log 2>&1; grep -E "encoder_blocks.2.conv|decoder_blocks.0.conv_trans|latent|TOTAL" gpurun_out/s15_layers.log | cut -c1-160
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "GPU:"; done
