R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "last_conv_gradient_norms" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
for v in "EBEN_FUSED_NORMS=1" "EBEN_FUSED_NORMS=0"; do echo "== $v"; for i in 1 2; do env $v python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>&1 >/dev/null | tail -1; done; env $v python tools/phase_times.py 2>&1 | grep -E "balancing|total"; done
