cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_bl.py -x -q -m gpu 2>&1 | tail -4 > $O/r05l_tests.txt; cat $O/r05l_tests.txt
timeout 900 python tools/layer_bench_bl.py --iters 20 2>&1 | grep -v amdgpu.ids | awk -F'|' '{print $1 "|" $4}' > $O/r05l_layers_dw.txt; cat $O/r05l_layers_dw.txt
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05l_s -o p -- python $R/tools/layer_bench_bl.py --iters 5 > /dev/null 2>&1
cp "$(find $O/r05l_s -name '*kernel_stats.csv' | head -1)" $O/r05l_kernel_stats.csv; rm -rf $O/r05l_s
grep -i "bl_dw\|slab_reduce\|wn_bwd\|tail_dw\|head_dw" $O/r05l_kernel_stats.csv | cut -d, -f1-4,6-7 | cut -c1-150
