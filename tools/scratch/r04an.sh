R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -3
timeout 1500 python -m pytest tests/test_gpu_models.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -5
