R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "generator_direct_operand" 2>&1 | tail -15
python $R/tools/gen_conv_bench.py
