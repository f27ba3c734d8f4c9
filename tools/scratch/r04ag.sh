R=$GRAFT_REPO_ROOT
export EBEN_PR_MIN_STRIDE=2 EBEN_PR_MAX_DIL=3 EBEN_PR_MAX_ROWS=256
timeout 900 python -m pytest tests/test_gpu_bl.py -x -q -m gpu -k "phases_as_rows" 2>&1 | grep -E "passed|failed|rror|skipped" | tail -3
for rows in 64 128 256; do echo "== EBEN_PR_MAX_ROWS=$rows"; EBEN_PR_MAX_ROWS=$rows python $R/tools/layer_bench_bl.py --iters 20 2>&1 | grep -E "^pqmf|TOTAL" | awk -F'|' '{print $1 "|" $3}'; done
