timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/mrstft_time.py 2>&1 | tail -1
timeout 600 python tools/layer_bench.py --filter G. 2>&1 | cut -c1-58,66-200 | grep -E "pointwise|TOTAL" | head -8
timeout 600 python tools/layer_bench.py --batch 64 --filter D. --math bf16 2>&1 | cut -c1-58,66-200 | grep -E "TOTAL"
for i in 1 2; do timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-f32-leg 2>&1 | grep "GPU:"; done
