cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_bl.py -q -m gpu 2>&1 | tail -15 > $O/r05o_tests.txt; cat $O/r05o_tests.txt
timeout 900 python tools/layer_bench_bl.py --iters 20 2>&1 | grep -v amdgpu.ids > $O/r05o_layers.txt; cat $O/r05o_layers.txt
