cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_bl.py -x -q -m gpu 2>&1 | tail -4 > $O/r05g_tests.txt; cat $O/r05g_tests.txt
timeout 300 python tools/layer_bench_bl.py --filter melgan --iters 20 2>&1 | grep -v amdgpu.ids > $O/r05g_layers_melgan.txt; cat $O/r05g_layers_melgan.txt
for L in melgan.4 melgan.3; do for V in t4st; do echo "== stamps $L $V"; EBEN_BIG_MIN_KS_DX=1000000 EBEN_HIP_LIB=$R/vibravox_amd/lib/var/libeben_$V.so timeout 200 python tools/scratch/t4_stamps.py $L 2>&1 | tail -3; done; done > $O/r05g_stamps.txt 2>&1; cat $O/r05g_stamps.txt
