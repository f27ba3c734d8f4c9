timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "bf16" 2>&1 | tail -5
for i in 1 2 3; do timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-f32-leg 2>&1 | grep "GPU:"; done
