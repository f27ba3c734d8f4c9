timeout 1500 python -m pytest tests/test_gpu_models.py -m gpu -x -q 2>&1 | grep -E "passed|failed|^E  " | head -8
for rep in 1 2; do for v in 0 1; do echo "== EBEN_SPLIT_D_FWD=$v"; EBEN_SPLIT_D_FWD=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg 2>&1 | grep "GPU:"; done; done
timeout 600 python tools/phase_times.py 2>&1 | tail -13
