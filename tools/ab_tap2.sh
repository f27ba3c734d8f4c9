timeout 1500 python -m pytest tests/test_gpu_models.py -m gpu -x -q 2>&1 | grep -E "passed|failed|^E  " | head -8
for rep in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg --force-ddp 2>&1 | grep -E "GPU:|Error|error" | head -3
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg 2>&1 | grep -E "GPU:|Error|error" | head -3
done
