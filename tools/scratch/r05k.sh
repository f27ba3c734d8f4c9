cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_bl.py -x -q -m gpu 2>&1 | tail -4 > $O/r05k_tests.txt; cat $O/r05k_tests.txt
timeout 900 python tools/layer_bench_bl.py --iters 20 2>&1 | grep -v amdgpu.ids | awk -F'|' '{print $1 "|" $4}' > $O/r05k_layers_dw.txt; cat $O/r05k_layers_dw.txt
