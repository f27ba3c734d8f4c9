R=$GRAFT_REPO_ROOT
for v in "EBEN_SPLIT_MELGAN=0" "EBEN_SPLIT_MELGAN=1" "EBEN_SPLIT_MELGAN=0" "EBEN_SPLIT_MELGAN=1"; do echo "== $v"; env $v python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>/tmp/err.txt >/dev/null; tail -1 /tmp/err.txt; done
timeout 1500 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "golden or benchmarked or full_size or replayed or bitwise or bundle_layout" 2>&1 | grep -E "passed|failed|rror" | tail -3
