python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
python tools/layer_bench.py --filter discriminator > gpurun_out/s20_layers.log 2>&1; grep -E "melgan|pqmf_discriminators.0|TOTAL" gpurun_out/s20_layers.log | cut -c1-40,105-160
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "GPU:"; done
