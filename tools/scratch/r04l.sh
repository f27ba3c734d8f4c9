mkdir -p gpurun_out/r04l
timeout 900 python -m pytest tests/test_gpu_bl.py -x -q -m gpu > gpurun_out/r04l/tests_bl.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r04l/tests_bl.log | tail -3
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv or residual or tap or gemm or stft" > gpurun_out/r04l/tests_ops.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r04l/tests_ops.log | tail -3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04l_s -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg > $O/r04l/stats_bench.json 2> $O/r04l/stats.err; cp "$(find $O/r04l_s -name "*kernel_stats.csv" | head -1)" $O/r04l/kernel_stats.csv; rm -rf $O/r04l_s
grep pack3 $O/r04l/kernel_stats.csv
cd $R; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg > gpurun_out/r04l/bench.json 2> gpurun_out/r04l/bench.err; tail -1 gpurun_out/r04l/bench.err
