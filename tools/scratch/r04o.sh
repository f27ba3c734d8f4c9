R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r04o
timeout 900 python -m pytest tests/test_gpu_bl.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -2
for mn in 200000 50000 0; do echo "== EBEN_PACK3C_MIN=$mn"; for i in 1 2; do EBEN_PACK3C_MIN=$mn python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>&1 >/dev/null | tail -1; done; done
