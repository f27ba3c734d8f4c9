R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "generator_direct_operand" 2>&1 | tail -15
python $R/tools/gen_fwd_timeline.py 2>/dev/null | grep -E "tap3|gc_|space_to|span"
