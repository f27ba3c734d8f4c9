R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "adam" 2>&1 | grep -E "passed|failed" | tail -2
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "golden or adam or optim or state" 2>&1 | grep -E "passed|failed" | tail -2
EBEN_DISC_MATH=bf16_bl python $R/tools/phase_times.py 2>/dev/null | tail -5
python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>&1 | grep -E "bench\] GPU" | tail -1
