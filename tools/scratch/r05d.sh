cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_bl.py -x -q -m gpu 2>&1 | tail -4 > $O/r05d_tests.txt; cat $O/r05d_tests.txt
timeout 300 python tools/layer_bench_bl.py --filter melgan --iters 20 2>&1 | grep -v amdgpu.ids > $O/r05d_layers_melgan.txt; cat $O/r05d_layers_melgan.txt
bash tools/l4_waits.sh r05d melgan.4 > $O/r05d_waits.log 2>&1; grep -E "^kernel|tap4" $O/r05d_l4_waits.txt | head -8
