timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "bf16" 2>&1 | tail -15
timeout 600 python tools/layer_bench.py --batch 64 --filter D. --math bf16 --min-gmacs 0.5 2>&1 | cut -c1-58,66-200 | tail -26
timeout 600 python tools/bf16_drift.py 2>&1 | tail -16
for m in f32 bf16; do EBEN_DISC_MATH=$m timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "GPU:"; done
