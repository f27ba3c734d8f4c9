cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_bl.py -x -q -m gpu 2>&1 | tail -3
echo "== stats test, new kernels"; timeout 600 python -m pytest "tests/test_gpu_models.py::test_bf16_training_statistics_against_fp32" -q -m gpu -s 2>&1 | grep -E "mean D loss|passed|failed"
echo "== stats test, EBEN_BIG=0"; EBEN_BIG=0 timeout 600 python -m pytest "tests/test_gpu_models.py::test_bf16_training_statistics_against_fp32" -q -m gpu -s 2>&1 | grep -E "mean D loss|passed|failed"
echo "== stats test, EBEN_PR_BIG=0"; EBEN_PR_BIG=0 timeout 600 python -m pytest "tests/test_gpu_models.py::test_bf16_training_statistics_against_fp32" -q -m gpu -s 2>&1 | grep -E "mean D loss|passed|failed"
timeout 900 python tools/layer_bench_bl.py --filter melgan --iters 20 2>&1 | grep -v amdgpu.ids | awk -F'|' '{print $1 "|" $4}'
timeout 1200 python -m pytest tests -q -m gpu --deselect "tests/test_gpu_models.py::test_bf16_training_statistics_against_fp32" 2>&1 | tail -6
