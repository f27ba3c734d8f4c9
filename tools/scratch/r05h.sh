cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_bl.py -x -q -m gpu 2>&1 | tail -4 > $O/r05h_tests.txt; cat $O/r05h_tests.txt
timeout 300 python tools/layer_bench_bl.py --filter melgan --iters 20 2>&1 | grep -v amdgpu.ids > $O/r05h_layers_melgan.txt; cat $O/r05h_layers_melgan.txt
