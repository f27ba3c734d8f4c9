cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_bl.py -x -q -m gpu 2>&1 | tail -6 > $O/r05m_tests.txt; cat $O/r05m_tests.txt
timeout 900 python tools/layer_bench_bl.py --filter melgan --iters 20 2>&1 | grep -v amdgpu.ids > $O/r05m_layers.txt; cat $O/r05m_layers.txt
