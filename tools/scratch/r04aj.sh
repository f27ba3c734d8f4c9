R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "generator_direct_operand or split_bf16" 2>&1 | tail -3
python $R/tools/gen_conv_bench.py | tail -9
