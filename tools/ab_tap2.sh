timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -x -q 2>&1 | grep -E "passed|failed|^E  " | head -5
timeout 600 python tools/phase_times.py 2>&1 | tail -14
for p in 0 1; do echo "== EBEN_PREPACK=$p"; for i in 1 2; do EBEN_PREPACK=$p timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-f32-leg 2>&1 | grep "GPU:"; done; done
