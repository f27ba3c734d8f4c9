cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_bl.py -x -q -m gpu 2>&1 | tail -4 > $O/r05c_tests.txt; cat $O/r05c_tests.txt
timeout 300 python tools/layer_bench_bl.py --filter melgan --iters 20 2>&1 | grep -v amdgpu.ids > $O/r05c_layers_melgan.txt; cat $O/r05c_layers_melgan.txt
for v in 1 4 5 8 16; do echo "== variant dbg$v"; EBEN_HIP_LIB=$R/vibravox_amd/lib/var/libeben_t4dbg$v.so timeout 200 python tools/layer_bench_bl.py --only melgan.4 --iters 20 2>&1 | grep "melgan.4"; done > $O/r05c_variants.txt 2>&1; cat $O/r05c_variants.txt
bash tools/l4_waits.sh r05c melgan.4 > $O/r05c_waits.log 2>&1; grep -E "^kernel|tap4" $O/r05c_l4_waits.txt
