python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
python tools/layer_bench.py > gpurun_out/s31_layers.log 2>&1; tail -1 gpurun_out/s31_layers.log
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "GPU:"; done
